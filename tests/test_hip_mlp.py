"""The MNIST-MLP training step on the HIP path against trajectories recorded from the reference
(tests/golden/mlp_*.npz) and against the numpy oracle; fused optimizer against the tape form."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from conftest import load_golden
import np_oracle as O
from test_cpu_backend import MLP, train
from common import mlp_trajectory_on_cpu, assert_as_close_to_float64_as_the_cpu_backend

pytestmark = pytest.mark.gpu


def make_opt(name, params, **kw):
    return {"adabelief": lambda: light.optim.AdaBelief(params, lr=1e-3, **kw), "adam": lambda: light.optim.Adam(params, lr=1e-3, **kw),
            "sgd": lambda: light.optim.SGD(params, lr=1e-4, momentum=0.9)}[name]()


@pytest.mark.parametrize("opt_name", ["adabelief", "adam", "sgd"])
def test_small_trajectory_vs_reference(hip, opt_name):
    g = load_golden("mlp_small_%s.npz" % opt_name)
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters({n: g["w0/" + n] for n in O.PARAM_ORDER})
    model.map_parameters(lambda p: p.hip())
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), g["labels"]] = 1
    losses, g0 = train(model, hip.from_numpy, g["x"], onehot, steps, make_opt(opt_name, model.parameters()))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        assert isinstance(p, hip)
        np.testing.assert_allclose(g0[n], g["g0/" + n], rtol=1e-5, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)      # two float32 runs: loose by nature ...
    # ... so the trained weights are judged by the float64 run of the same tape: as close to it as the reference's float32 result
    w0 = {n: g["w0/" + n] for n in O.PARAM_ORDER}
    _, ref64 = mlp_trajectory_on_cpu(w0, g["x"], onehot, steps, lambda params: make_opt(opt_name, params), np.float64)
    assert_as_close_to_float64_as_the_cpu_backend({n: p.numpy() for n, p in model.named_parameters()}, {n: g["wf/" + n] for n in O.PARAM_ORDER},
                                                  ref64, what=opt_name)


def test_full_size_trajectory_vs_reference(hip):
    """BASELINE config #3: 784 -> 512 -> 10, batch 1024, mse, AdaBelief(lr=1e-3), 5 steps"""
    g = load_golden("mlp_full_adabelief.npz")
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    w0, x, onehot, labels = O.synthetic_mlp_problem(seed, d_in, d_hid, d_out, batch)
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters(w0)
    model.map_parameters(lambda p: p.hip())
    losses, g0 = train(model, hip.from_numpy, x, onehot, steps, make_opt("adabelief", model.parameters()))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    sample = lambda a: a.reshape(-1)[::max(1, a.size // 64)][:64]   # noqa: E731
    # step-0 gradients, WHOLE arrays, against the oracle evaluated in float64 on the same float32 inputs (relative Frobenius, the
    # north star's 1e-5); the 64-element samples recorded from the reference (float32 itself) are compared at its own noise level
    _, g64, _ = O.mlp_loss_and_grads({k: v.astype(np.float64) for k, v in w0.items()}, x.astype(np.float64), onehot.astype(np.float64))
    for n, p in model.named_parameters():
        assert g64[n].dtype == np.float64
        e = np.linalg.norm(g0[n].astype(np.float64) - g64[n]) / np.linalg.norm(g64[n])
        assert e <= 1e-5, (n, e)
        e_ref = np.linalg.norm(g["g0sample/" + n].astype(np.float64) - sample(g64[n])) / np.linalg.norm(sample(g64[n]))
        np.testing.assert_allclose(sample(g0[n]), g["g0sample/" + n], rtol=0, atol=(e + e_ref + 1e-7) * 4 * np.abs(sample(g64[n])).max(), err_msg=n)
        np.testing.assert_allclose(sample(p.numpy()), g["wfsample/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
        w = p.numpy().astype(np.float64)
        np.testing.assert_allclose([w.sum(), np.abs(w).sum()], g["wfsum/" + n], rtol=1e-4, atol=1e-3)
    # the five-step trajectory against a float64 run of the same tape: WHOLE weight arrays, and the losses at 1e-5
    make = lambda params: make_opt("adabelief", params)      # noqa: E731
    losses64, ref64 = mlp_trajectory_on_cpu(w0, x, onehot, steps, make, np.float64)
    _, cpu32 = mlp_trajectory_on_cpu(w0, x, onehot, steps, make, np.float32)
    np.testing.assert_allclose(losses, losses64, rtol=1e-5)
    assert_as_close_to_float64_as_the_cpu_backend({n: p.numpy() for n, p in model.named_parameters()}, cpu32, ref64, what="full size")


def test_step_gradients_vs_oracle_on_fresh_seed(hip):
    w0, x, onehot, _ = O.synthetic_mlp_problem(123, 100, 64, 10, 256)
    loss_o, grads_o, dx_o = O.mlp_loss_and_grads(w0, x, onehot)
    model = MLP(100, 64, 10)
    model.load_parameters(w0)
    model.map_parameters(lambda p: p.hip())
    tx = hip.from_numpy(x)
    l = light.loss.mse(model(tx), hip.from_numpy(onehot))
    l.backward()
    np.testing.assert_allclose(l.item(), loss_o, rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), grads_o[n], rtol=1e-5, atol=1e-5, err_msg=n)
    np.testing.assert_allclose(tx.grad.numpy(), dx_o, rtol=1e-5, atol=1e-6)     # dX is computed for the input layer too


@pytest.mark.parametrize("opt_name", ["adabelief", "adam"])
def test_fused_optimizer_equals_tape_form(hip, opt_name):
    g = load_golden("mlp_small_%s.npz" % opt_name)
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), g["labels"]] = 1
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters({n: g["w0/" + n] for n in O.PARAM_ORDER})
    model.map_parameters(lambda p: p.hip())
    losses, _ = train(model, hip.from_numpy, g["x"], onehot, steps, make_opt(opt_name, model.parameters(), fused=True))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
    w0 = {n: g["w0/" + n] for n in O.PARAM_ORDER}
    _, ref64 = mlp_trajectory_on_cpu(w0, g["x"], onehot, steps, lambda params: make_opt(opt_name, params), np.float64)
    assert_as_close_to_float64_as_the_cpu_backend({n: p.numpy() for n, p in model.named_parameters()}, {n: g["wf/" + n] for n in O.PARAM_ORDER},
                                                  ref64, what="fused " + opt_name)
