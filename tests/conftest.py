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
