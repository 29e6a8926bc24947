import json
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# ---- multi-process GPU tests ------------------------------------------------------------------------------------------
# The rank processes of a multi-rank GPU test are started by a helper (tests/rank_spawner.py) that this session starts
# HERE, before any test - or fixture - has initialised the GPU: a process that holds the GPU must not fork + exec.
_spawner = None


def pytest_sessionstart(session):
    global _spawner
    import subprocess
    try:
        _spawner = subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "rank_spawner.py")], stdin=subprocess.PIPE,
                                    stdout=subprocess.PIPE, text=True, cwd=ROOT)
        assert json.loads(_spawner.stdout.readline()).get("ready")
    except Exception as e:                                   # reported by the tests that need it
        _spawner = "rank spawner did not start: %r" % (e,)


def pytest_sessionfinish(session, exitstatus):
    global _spawner
    if _spawner is not None and not isinstance(_spawner, str):
        try:
            _spawner.stdin.write(json.dumps({"quit": True}) + "\n")
            _spawner.stdin.flush()
            _spawner.wait(timeout=10)
        except Exception:
            _spawner.kill()                                  # exactly the process this session started
    _spawner = None


@pytest.fixture(scope="session")
def spawn_ranks():
    """spawn_ranks(nproc, argv, env=None, rank_env=None, timeout=300, wait_for_all=False) -> {"rc", "outputs", "codes"}:
    runs `python argv...` as ranks 0..nproc-1 in fresh interpreters started by the helper process"""
    if _spawner is None or isinstance(_spawner, str):
        pytest.fail(str(_spawner or "rank spawner not started"))

    def run(nproc, argv, env=None, rank_env=None, timeout=300, wait_for_all=False):
        req = {"nproc": nproc, "argv": list(argv), "env": env or {}, "rank_env": rank_env or {}, "timeout": timeout,
               "wait_for_all": wait_for_all}
        _spawner.stdin.write(json.dumps(req) + "\n")
        _spawner.stdin.flush()
        line = _spawner.stdout.readline()
        assert line, "the rank spawner died"
        return json.loads(line)
    return run


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def hip():
    """the HipTensor class, with the device library initialised - fails loudly if it cannot be"""
    from lightgrad_amd.autograd.hip import HipTensor, HipDevice
    from lightgrad_amd.autograd.hip import lib as hiplib
    hiplib.lib()       # raises HipError when the .so is missing or no GPU is visible: never a silent skip
    assert HipDevice.info()["arch"].startswith("gfx950"), HipDevice.info()
    return HipTensor
