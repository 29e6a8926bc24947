"""Tensors that are not float32 on the CPU backend: the fixture recorded from the reference's CPU backend
(tests/golden/typed_ops.npz; oracle/gen_golden.py) bit for bit - CpuTensor is numpy like the reference's (cpu/tensor.py:45-46)."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from conftest import load_golden
from test_hip_typed import run_typed_cases

DTYPES = ["int16", "int32", "int64", "float64"]


@pytest.mark.parametrize("name", DTYPES)
def test_cpu_backend_reproduces_the_reference(name):
    run_typed_cases(CpuTensor, load_golden("typed_ops.npz"), name, exact_everywhere=True)


def test_astype_on_the_cpu_backend():
    a = CpuTensor.from_numpy(np.asarray([[-1.7, 2.2], [3.9, -0.1]], np.float32))
    np.testing.assert_array_equal(a.astype(np.int32).numpy(), np.asarray([[-1, 2], [3, 0]], np.int32))
    b = a.astype(np.float64)
    assert b.dtype == np.float64
    (b * 2.0).sum().backward()
    np.testing.assert_array_equal(a.grad.numpy(), np.full((2, 2), 2.0, np.float32))
    assert a.grad.dtype == np.float32
