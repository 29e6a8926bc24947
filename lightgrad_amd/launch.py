"""Single-node rank launcher: one OS process per GPU.

    python -m lightgrad_amd.launch --nproc 8 train.py --my --args      # or: spawn_ranks(8, ["train.py", ...])

Starts `nproc` fresh python interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the
variables `torch.distributed.run` exports, so a script works under either launcher) plus LIGHTGRAD_RCCL_ID_FILE, the
path rank 0 publishes the RCCL unique id under (dist.RcclCommunicator).  Rank 0 inherits this process's stdout; the
other ranks' stdout is sent to stderr, so a program that prints ONE machine-readable line on rank 0 (bench.py) yields
exactly that line.  If any rank fails, the rest are terminated (by pid - never by pattern) and the first non-zero
exit code is returned.

The launcher itself never initialises a GPU: it imports nothing but the standard library and makes no HIP call, so
the children - not this process - own the devices.  (The reference has no launcher and no distributed code.)
"""
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank: int, nproc: int, port: int, id_file: str, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LIGHTGRAD_RCCL_ID_FILE": id_file,
                "LIGHTGRAD_LAUNCHED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver stack
    env.setdefault("PYTHONUNBUFFERED", "1")
    return env


def spawn_ranks(nproc: int, argv, timeout: float = None, poll: float = 0.05, grace: float = 5.0) -> int:
    """run `python argv...` as ranks 0..nproc-1 of one job; returns the job's exit code (0 = every rank succeeded)"""
    assert nproc >= 1 and len(argv) >= 1
    port = _free_port()
    workdir = tempfile.mkdtemp(prefix="lightgrad_launch_")
    id_file = os.path.join(workdir, "rccl.id")
    children = []
    try:
        for rank in range(nproc):
            out = None if rank == 0 else sys.stderr          # only rank 0 may write to the job's stdout
            children.append(subprocess.Popen([sys.executable] + list(argv), env=rank_environment(rank, nproc, port, id_file),
                                             stdout=out, start_new_session=False))
        deadline = None if timeout is None else time.time() + timeout
        code = 0
        running = list(children)
        while running and code == 0:
            for child in list(running):
                rc = child.poll()
                if rc is None:
                    continue
                running.remove(child)
                if rc != 0:
                    code = rc if rc > 0 else 128 - rc        # killed by a signal: shell convention
                    sys.stderr.write("[lightgrad.launch] rank %d exited with %d; stopping the other ranks\n"
                                     % (children.index(child), rc))
                    break
            if deadline is not None and time.time() > deadline and running:
                sys.stderr.write("[lightgrad.launch] timeout after %.0f s; stopping %d rank(s)\n" % (timeout, len(running)))
                code = 124
            if running and code == 0:
                time.sleep(poll)
        return code
    finally:
        _stop(children, grace)
        for name in os.listdir(workdir):              # rendezvous files of the communicators (RCCL id, peer-window handles)
            try:
                os.remove(os.path.join(workdir, name))
            except OSError:
                pass
        try:
            os.rmdir(workdir)
        except OSError:
            pass


def _stop(children, grace: float) -> None:
    """terminate exactly the processes this launcher started that are still alive"""
    alive = [c for c in children if c.poll() is None]
    for c in alive:
        try:
            c.send_signal(signal.SIGTERM)
        except OSError:
            pass
    end = time.time() + grace
    for c in alive:
        try:
            c.wait(timeout=max(0.0, end - time.time()))
        except subprocess.TimeoutExpired:
            try:
                c.kill()
            except OSError:
                pass
            c.wait()


def main(args=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--nproc", type=int, required=True, help="ranks to start (one per GPU)")
    ap.add_argument("--timeout", type=float, default=None)
    ap.add_argument("script", nargs=argparse.REMAINDER, help="script and its arguments")
    ns = ap.parse_args(args)
    if not ns.script:
        ap.error("no script given")
    return spawn_ranks(ns.nproc, ns.script, timeout=ns.timeout)


if __name__ == "__main__":
    sys.exit(main())
