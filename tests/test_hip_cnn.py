"""CNN ops on the HIP path: conv / pad / pooling against the reference fixtures and the reference's own device
sweep (test/test_opencl_tensor.py:60-77: dims 1-3, sizes 3/6/9, strides 1-3, kernels 3/5/7, channels 1-3) vs the CPU
backend; gradients by numerical differentiation; a small CNN trained on both backends."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from common import check_gradients, compare_with_cpu
from test_cnn_cpu import G, CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_matches_reference(hip, name):
    y, ts = run_case(hip, name)
    assert y.shape == G[name + "/out"].shape
    exact = name.startswith(("pad", "max_pool", "min_pool"))
    if exact:
        np.testing.assert_array_equal(y.numpy(), G[name + "/out"])
    else:
        np.testing.assert_allclose(y.numpy(), G[name + "/out"], rtol=1e-5, atol=1e-5)
    for i, t in enumerate(ts):
        np.testing.assert_allclose(t.grad.numpy(), G["%s/grad%d" % (name, i)], rtol=1e-5, atol=1e-5)


def test_conv_sweep_vs_cpu(hip):
    np.random.seed(1337)
    n = 0
    for dim in (1, 2, 3):
        for shape in (3, 6, 9):
            for stride in (1, 2, 3):
                for kernel in (3, 5, 7):
                    for in_c in (1, 3):
                        for out_c in (1, 2):
                            if kernel > shape:
                                continue
                            cpu_k = CpuTensor.uniform(-1, 1, shape=(out_c, in_c) + (kernel,) * dim)
                            hip_k = cpu_k.hip()
                            compare_with_cpu(hip, lambda x: x.conv(hip_k if isinstance(x, hip) else cpu_k, strides=stride),
                                             shapes=[(2, in_c) + (shape,) * dim], rtol=1e-5, atol=1e-5)
                            n += 1
    assert n > 100


def test_conv_pool_pad_gradcheck(hip):
    np.random.seed(12)
    check_gradients(hip, hip.conv, shapes=[(2, 2, 5, 5), (3, 2, 3, 3)], strides=1, tol=2e-3)
    check_gradients(hip, lambda x, k: x.conv(k, strides=2), shapes=[(2, 1, 7, 6), (2, 1, 3, 2)], tol=2e-3)
    check_gradients(hip, lambda x: x.pad(2), shapes=[(5, 6)])
    check_gradients(hip, lambda x: x.max_pool(), shapes=[(2, 4, 6)])
    check_gradients(hip, lambda x: x.mean_pool(), shapes=[(2, 4, 5)])


def test_cnn_training_matches_cpu_backend(hip):
    import lightgrad_amd.nn as nn

    class CNN(nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.c1 = nn.Conv2d(1, 8, kernelsize=3, bias=False, pad=0)
            self.c2 = nn.Conv2d(8, 16, kernelsize=3, pad=0)
            self.l1 = nn.Linear(5 * 5 * 16, 10)

        def forward(self, x):                                   # the CNN of the reference's examples/mnist.py:14-22
            y = self.c1(x).max_pool().relu()
            y = self.c2(y).max_pool().relu()
            return self.l1(y.reshape(-1, 5 * 5 * 16))
    np.random.seed(0)
    cpu_model, hip_model = CNN(), CNN()
    start = [(n, p.numpy().copy()) for n, p in cpu_model.named_parameters()]
    hip_model.load_parameters(cpu_model.named_parameters())
    hip_model.map_parameters(lambda p: p.hip())
    x = np.random.uniform(0, 1, (8, 1, 28, 28)).astype(np.float32)
    t = np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 8)]
    out = {}
    for name, model, T in (("cpu", cpu_model, CpuTensor), ("hip", hip_model, hip)):
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-3)
        losses = []
        for _ in range(4):
            l = light.loss.mse(model(T.from_numpy(x)), T.from_numpy(t))
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(l.item())
        out[name] = (losses, [p.numpy() for p in model.parameters()])
    # yardstick: the same four steps in float64 (CpuTensor.default_dtype).  AdaBelief divides by the square root of a second
    # moment that starts near zero, so float32 rounding noise in a gradient is amplified in the first updates: the float32 CPU
    # backend itself ends 1e-4 .. 1e-3 (relative) from the float64 run - the HIP path must be about as close to it
    from common import float64_tape, rel_frobenius
    with float64_tape():
        ref_model = CNN()
        ref_model.load_parameters([(n, a.astype(np.float64)) for n, a in start])
        assert all(p.dtype == np.float64 for p in ref_model.parameters())
        opt = light.optim.AdaBelief(ref_model.parameters(), lr=1e-3)
        ref_losses = []
        for _ in range(4):
            l = light.loss.mse(ref_model(CpuTensor.from_numpy(x.astype(np.float64))), CpuTensor.from_numpy(t.astype(np.float64)))
            opt.zero_grad()
            l.backward()
            opt.step()
            ref_losses.append(l.item())
        ref_params = [p.numpy() for p in ref_model.parameters()]
    np.testing.assert_allclose(out["hip"][0], ref_losses, rtol=2e-5)
    for a, b, r in zip(out["hip"][1], out["cpu"][1], ref_params):
        e_hip, e_cpu = rel_frobenius(a, r), rel_frobenius(b, r)
        assert e_hip <= max(1e-5, 2 * e_cpu), (e_hip, e_cpu)
