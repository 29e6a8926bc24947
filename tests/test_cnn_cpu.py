"""CNN ops (conv in 1/2/3-d with strides, pad, pooling) on the CPU backend against fixtures recorded from the
reference (tests/golden/cnn_ops.npz) and the reference's own convolution gradcheck (test/test_cpu_tensor.py:38)."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from common import check_gradients
from conftest import load_golden

G = load_golden("cnn_ops.npz")
CASES = sorted({k.split("/")[0] for k in G.files})


def case_fn(name):
    if name.startswith("conv"):
        stride = int(name.split("_st")[1].split("_")[0])
        return lambda a, b: a.conv(b, strides=stride)
    return {"pad2": lambda a: a.pad(2), "pad_1_3": lambda a: a.pad((1, 3), value=0.5), "max_pool": lambda a: a.max_pool(),
            "min_pool_3x2": lambda a: a.min_pool(kernel=(3, 2)), "max_pool_2x3": lambda a: a.max_pool(kernel=(2, 3))}[name]


def run_case(T, name):
    n_in = len([k for k in G.files if k.startswith(name + "/in")])
    ts = [T.from_numpy(G["%s/in%d" % (name, i)].copy()) for i in range(n_in)]
    y = case_fn(name)(*ts)
    (y * T.from_numpy(G[name + "/w"], requires_grad=False)).backward(allow_fill=True)
    return y, ts


@pytest.mark.parametrize("name", CASES)
def test_matches_reference(name):
    y, ts = run_case(CpuTensor, name)
    assert y.shape == G[name + "/out"].shape
    np.testing.assert_allclose(y.numpy(), G[name + "/out"], rtol=1e-5, atol=1e-6)
    for i, t in enumerate(ts):
        np.testing.assert_allclose(t.grad.numpy(), G["%s/grad%d" % (name, i)], rtol=1e-5, atol=1e-5)


def test_convolution_gradcheck():
    np.random.seed(1234)
    check_gradients(CpuTensor, CpuTensor.conv, shapes=[(3, 2, 5, 5), (4, 2, 3, 3)], strides=1)
    check_gradients(CpuTensor, lambda x: CpuTensor.pad(x, padding=2), shapes=[(9, 11)])


def test_cnn_model_trains():
    import lightgrad_amd.nn as nn
    np.random.seed(0)

    class CNN(nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.c1 = nn.Conv2d(1, 4, kernelsize=3, bias=False, pad=0)
            self.c2 = nn.Conv2d(4, 8, kernelsize=3, pad=1)
            self.l1 = nn.Linear(3 * 3 * 8, 10)

        def forward(self, x):
            y = self.c1(x).max_pool().relu()
            y = self.c2(y).max_pool().relu()
            return self.l1(y.reshape(-1, 3 * 3 * 8))
    model = CNN()
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-2)
    x = CpuTensor.uniform(0, 1, (16, 1, 14, 14))
    t = CpuTensor.from_numpy(np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 16)])
    losses = []
    for _ in range(15):
        l = light.loss.mse(model(x), t)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    assert losses[-1] < 0.7 * losses[0]
