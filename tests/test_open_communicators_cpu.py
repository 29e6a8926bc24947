"""dist.open_communicators: every form of the gradient exchange is attempted under a timeout and the ranks VOTE through the
job's rendezvous files, so that no rank is left waiting on a form that does not work on its node (the reference
has no distributed code).  Two real rank processes, fake openers: one form fails on one rank and hangs on the other (what
ncclCommInitRank does when a peer falls out), one hangs everywhere, one works."""
import json
import os
import subprocess
import sys
import textwrap
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from lightgrad_amd.dist import open_communicators, Communicator

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    class Fake(Communicator):
        def __init__(self, name):
            self.name, self.rank, self.world_size, self.closed = name, rank, world, False
        def close(self):
            assert not getattr(self, "_abandoned", False), "an abandoned communicator must not be closed collectively"
            self.closed = True

    def fails_on_rank_1_and_hangs_elsewhere():
        if rank == 1:
            raise RuntimeError("no such device")
        time.sleep(3600)

    def hangs_everywhere():
        time.sleep(3600)

    def works_only_on_rank_0():
        if rank != 0:
            raise OSError("cannot map the window of my peer")
        return Fake("half")

    t0 = time.time()
    opened, why_not = open_communicators(rank, world, [("first", fails_on_rank_1_and_hangs_elsewhere), ("second", hangs_everywhere),
                                                       ("half", works_only_on_rank_0), ("skipped", None), ("good", lambda: Fake("good"))],
                                         timeout=1.0)
    print(json.dumps({"rank": rank, "opened": sorted(opened), "why_not": why_not, "seconds": time.time() - t0}))
    sys.stdout.flush()
    os._exit(0)              # abandoned helper threads are asleep for an hour
""")


@pytest.mark.timeout(120)
def test_forms_that_fail_or_hang_on_any_rank_are_dropped_by_all(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LIGHTGRAD_RCCL_ID_FILE=str(tmp_path / "job.id"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=100)
        assert p.returncode == 0, err[-3000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    for o in outs:
        assert o["opened"] == ["good"], o
        assert set(o["why_not"]) == {"first", "second", "half", "skipped"}
        assert "rank 1: RuntimeError: no such device" in o["why_not"]["first"] and "rank 0: no answer within 1 s" in o["why_not"]["first"]
        assert "rank 0: no answer" in o["why_not"]["second"] and "rank 1: no answer" in o["why_not"]["second"]
        assert o["why_not"]["half"] == "rank 1: OSError: cannot map the window of my peer"
        assert o["seconds"] < 30
    assert outs[0]["why_not"] == outs[1]["why_not"]                     # every rank decides alike, with the same words


def test_a_leftover_rendezvous_file_of_a_dead_process_is_not_taken_for_a_peer(tmp_path):
    """dist._exchange_blobs(accept=...): a file under the same name whose writer is gone counts as not yet written"""
    import threading
    import time
    sys.path.insert(0, ROOT)
    from lightgrad_amd.dist import _exchange_blobs
    prefix = str(tmp_path / "job.p2p.1")
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    stale = b"S" * 64 + ("%-16d" % dead.pid).encode()
    with open(prefix + ".1", "wb") as f:
        f.write(stale)

    def alive(data):
        try:
            os.kill(int(data[64:].decode()), 0)
        except ProcessLookupError:
            return False
        return True
    fresh = b"F" * 64 + ("%-16d" % os.getpid()).encode()

    def late_peer():
        time.sleep(0.3)
        tmp = prefix + ".1.tmp"
        with open(tmp, "wb") as f:
            f.write(fresh)
        os.replace(tmp, prefix + ".1")
    threading.Thread(target=late_peer).start()
    mine = b"M" * 64 + ("%-16d" % os.getpid()).encode()
    got = _exchange_blobs(0, 2, mine, prefix, timeout=10.0, accept=alive)
    assert got == [mine, fresh]
    with open(prefix + ".1", "wb") as f:
        f.write(stale)
    with pytest.raises(TimeoutError):
        _exchange_blobs(0, 2, mine, prefix, timeout=0.5, accept=alive)
