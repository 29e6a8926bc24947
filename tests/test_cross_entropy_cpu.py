"""loss.cross_entropy (reference loss.py:14-24): the oracle and the CpuTensor form against fixtures produced by the
reference itself (oracle/gen_golden.py)."""
import os
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from oracle import np_oracle

CASES = ["n8_c10_i64", "n5_c3_i32", "n33_c130_i16", "n1_c1000_i64"]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "cross_entropy.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(golden, name):
    loss, grad = np_oracle.cross_entropy(golden[name + "/logits"], golden[name + "/labels"], golden[name + "/w"])
    np.testing.assert_allclose(loss, golden[name + "/loss"], rtol=1e-6)
    np.testing.assert_allclose(grad, golden[name + "/grad"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", CASES)
def test_cpu_backend_matches_reference(golden, name):
    y = CpuTensor.from_numpy(golden[name + "/logits"].copy())
    labels = CpuTensor.from_numpy(golden[name + "/labels"], requires_grad=False)
    loss = light.loss.cross_entropy(y, labels)
    (loss * CpuTensor.from_numpy(golden[name + "/w"], requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(loss.numpy(), golden[name + "/loss"], rtol=1e-6)
    np.testing.assert_allclose(y.grad.numpy(), golden[name + "/grad"], rtol=1e-5, atol=1e-8)


def test_trains_a_classifier():
    """a linear classifier on separable blobs reaches > 95 % accuracy with the loss"""
    rng = np.random.RandomState(0)
    centres = rng.uniform(-3, 3, (4, 6)).astype(np.float32)
    labels = rng.randint(0, 4, 256)
    x = centres[labels] + rng.normal(0, 0.3, (256, 6)).astype(np.float32)
    model = light.nn.Linear(6, 4)
    opt = light.optim.SGD(model.parameters(), lr=0.5)
    xt, lt = CpuTensor.from_numpy(x, requires_grad=False), CpuTensor.from_numpy(labels.astype(np.int64), requires_grad=False)
    first = None
    for _ in range(60):
        loss = light.loss.cross_entropy(model(xt), lt)
        first = loss.item() if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert loss.item() < 0.3 * first
    assert (model(xt).numpy().argmax(-1) == labels).mean() > 0.95
