"""Helper process of the multi-process GPU tests (tests/conftest.py starts it in pytest_sessionstart).

A process that has initialised the GPU must not start other programs (fork + exec from it takes the GPU box down), and the
one process of a `pytest -m gpu` run initialises the GPU long before it reaches a multi-rank test.  So this helper is started
FIRST, never touches a GPU itself (standard library only, no HIP, no torch) and starts the rank interpreters on request:

    request  (one JSON line on stdin):  {"nproc": 2, "argv": ["tests/dist_rank_worker.py", ...], "env": {...}, "timeout": 300}
    response (one JSON line on stdout): {"rc": 0, "outputs": ["<stdout+stderr of rank 0>", ...]}

Ranks get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / LIGHTGRAD_RCCL_ID_FILE like under lightgrad_amd.launch (whose
rank_environment is used).  On a failure or a timeout the remaining ranks are stopped by pid.  {"quit": true} ends the helper.
"""
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch_module():
    spec = importlib.util.spec_from_file_location("lightgrad_launch", os.path.join(ROOT, "lightgrad_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_job(launch, req):
    nproc, argv, timeout = int(req["nproc"]), list(req["argv"]), float(req.get("timeout", 300))
    workdir = tempfile.mkdtemp(prefix="lightgrad_ranks_")
    id_file = os.path.join(workdir, "rccl.id")
    port = launch._free_port()
    children, logs = [], []
    try:
        for rank in range(nproc):
            env = launch.rank_environment(rank, nproc, port, id_file)
            env.update({k: str(v) for k, v in req.get("env", {}).items()})
            for k, v in req.get("rank_env", {}).get(str(rank), {}).items():
                env[k] = str(v)
            log = open(os.path.join(workdir, "rank%d.log" % rank), "w+")
            logs.append(log)
            children.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT))
        deadline, rc, running = time.time() + timeout, 0, list(children)
        while running and rc == 0:
            for child in list(running):
                code = child.poll()
                if code is None:
                    continue
                running.remove(child)
                if code != 0 and not req.get("wait_for_all", False):
                    rc = code if code > 0 else 128 - code
            if running and time.time() > deadline:
                rc = 124
            if running and rc == 0:
                time.sleep(0.02)
        if req.get("wait_for_all", False) and rc == 0:
            codes = [c.returncode for c in children]
            rc = next((c if c > 0 else 128 - c for c in codes if c != 0), 0)
        launch._stop(children, 5.0)
        outputs = []
        for log in logs:
            log.flush()
            log.seek(0)
            outputs.append(log.read()[-20000:])
        return {"rc": rc, "outputs": outputs, "codes": [c.returncode for c in children]}
    finally:
        for log in logs:
            log.close()
        for name in os.listdir(workdir):
            try:
                os.remove(os.path.join(workdir, name))
            except OSError:
                pass
        try:
            os.rmdir(workdir)
        except OSError:
            pass


def main():
    launch = _launch_module()
    sys.stdout.write(json.dumps({"ready": True, "pid": os.getpid()}) + "\n")
    sys.stdout.flush()
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        if req.get("quit"):
            break
        try:
            resp = run_job(launch, req)
        except Exception as e:                       # the test sees the reason instead of a dead pipe
            resp = {"rc": 125, "outputs": [], "error": "%s: %s" % (type(e).__name__, e)}
        sys.stdout.write(json.dumps(resp) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
