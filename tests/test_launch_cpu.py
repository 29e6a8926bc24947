"""The rank launcher and bench.py's self-launch (`python bench.py --gpus N` with no external launcher), exercised on
CPU: the parent must only spawn, rank 0's JSON line must be the only thing on stdout, a failing rank must take the
job down with its exit code.  The children run bench.py's `--dry-run` stand-in (CpuTensor + gloo): same rank
environment, same DataParallel exchange protocol, no GPU."""
import json
import os
import subprocess
import sys
import textwrap
import time
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LIGHTGRAD_RCCL_ID_FILE", "LIGHTGRAD_LAUNCHED"):
        env.pop(k, None)
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "expected exactly one line on stdout, got: %r" % (lines,)
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_self_launch_two_ranks_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 5
    ranks = out["ranks"]
    assert ranks["world_size"] == 2 and ranks["communicator_ranks"] == 2 and ranks["launcher"] == "lightgrad_amd.launch"
    # the ranks walked the whole chain of communicators (dist.open_communicators: attempt, vote, next form) down to the host one
    assert ranks["communicator"] == "GlooCommunicator"
    assert "no RCCL" in ranks["communicators_not_usable"]["rccl"] and "no peer windows" in ranks["communicators_not_usable"]["peer"]
    assert len(ranks["per_rank_steps_per_sec"]) == 2 and all(v > 0 for v in ranks["per_rank_steps_per_sec"])
    # aggregate = ranks x steps / slowest rank's time: never above the sum of the per-rank rates
    assert 0 < out["value"] <= sum(ranks["per_rank_steps_per_sec"]) * 1.001
    assert out["config"]["parallelism"] == "dp2"


@pytest.mark.timeout(300)
def test_bench_under_torch_distributed_run_still_works():
    """the driver's N > 1 command line: WORLD_SIZE is already set, bench.py must not spawn again"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["communicator_ranks"] == 2 and out["ranks"]["launcher"] == "external"


def test_single_rank_dry_run_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["ranks"]["communicator"] == "SingleProcess"


def test_launcher_environment_and_failure_propagation(tmp_path):
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("lightgrad_launch_t", os.path.join(ROOT, "lightgrad_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    # the launcher module itself pulls in nothing that could touch a GPU
    src = open(os.path.join(ROOT, "lightgrad_amd", "launch.py")).read()
    assert "import torch" not in src and "ctypes" not in src and "lightgrad_amd.autograd" not in src
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert os.environ["LIGHTGRAD_RCCL_ID_FILE"] and int(os.environ["MASTER_PORT"]) > 0
        open(os.path.join(sys.argv[1], "seen%d" % rank), "w").write(os.environ["LIGHTGRAD_RCCL_ID_FILE"])
        mode = sys.argv[2]
        if mode == "fail" and rank == 1:
            sys.exit(3)
        if mode == "fail":
            time.sleep(60)          # must be terminated by the launcher, not run to completion
        print("rank %d of %d" % (rank, world))
    """))
    assert launch.spawn_ranks(3, [str(script), str(tmp_path), "ok"]) == 0
    files = [(tmp_path / ("seen%d" % r)).read_text() for r in range(3)]
    assert len(set(files)) == 1                                  # one rendezvous file per job, shared by its ranks
    t0 = time.time()
    assert launch.spawn_ranks(3, [str(script), str(tmp_path), "fail"]) == 3
    assert time.time() - t0 < 30
