"""hipGraph capture/replay of the training step and the graph-safe (device-resident step counter)
optimizer: replays must reproduce the reference trajectory exactly like the eager tape does."""
import numpy as np
import pytest
import lightgrad_amd as light
from conftest import load_golden
import np_oracle as O
from test_cpu_backend import MLP
from common import mlp_trajectory_on_cpu, assert_as_close_to_float64_as_the_cpu_backend


def judged_by_float64(model, g, opt_name, steps, what):
    """the trained weights against a float64 run of the same tape (tests/common.py): as close to it as the reference's float32 run"""
    onehot = np.zeros((int(g["config"][3]), int(g["config"][2])), np.float32)
    onehot[np.arange(onehot.shape[0]), g["labels"]] = 1
    cls = {"adabelief": light.optim.AdaBelief, "adam": light.optim.Adam}[opt_name]
    _, ref64 = mlp_trajectory_on_cpu({n: g["w0/" + n] for n in O.PARAM_ORDER}, g["x"], onehot, steps, lambda params: cls(params, lr=1e-3), np.float64)
    assert_as_close_to_float64_as_the_cpu_backend({n: p.numpy() for n, p in model.named_parameters()}, {n: g["wf/" + n] for n in O.PARAM_ORDER},
                                                  ref64, what=what)

pytestmark = pytest.mark.gpu


def build(hip, g, opt_name, **kw):
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters({n: g["w0/" + n] for n in O.PARAM_ORDER})
    model.map_parameters(lambda p: p.hip())
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), g["labels"]] = 1
    cls = {"adabelief": light.optim.AdaBelief, "adam": light.optim.Adam}[opt_name]
    opt = cls(model.parameters(), lr=1e-3, fused=True, **kw)
    x, t = hip.from_numpy(g["x"]), hip.from_numpy(onehot)

    def step():
        l = light.loss.mse(model(x), t)
        opt.zero_grad()
        l.backward()
        opt.step()
        return l
    return model, opt, step, steps


@pytest.mark.parametrize("opt_name", ["adabelief", "adam"])
def test_device_step_counter_equals_host_scalars(hip, opt_name):
    g = load_golden("mlp_small_%s.npz" % opt_name)
    model, opt, step, steps = build(hip, g, opt_name, device_step=True)
    losses = [step().item() for _ in range(steps)]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
    judged_by_float64(model, g, opt_name, steps, "device step counter")
    assert opt.t == steps * 4 and int(opt._step_counter.numpy()[0]) == steps


def test_graph_replay_reproduces_reference_trajectory(hip):
    from lightgrad_amd.autograd.hip import HipGraph, HipDevice
    g = load_golden("mlp_small_adabelief.npz")
    model, opt, step, steps = build(hip, g, "adabelief", device_step=True)
    losses = [step().item() for _ in range(2)]                 # eager warm-up: steps 0 and 1
    graph = HipGraph()
    with graph.capture():
        loss = step()                                          # recorded, not executed
    opt.t -= 4                                                 # the capture pass ran the python bookkeeping, not the kernels
    for _ in range(steps - 2):
        graph.replay()
        opt.on_graph_replay()
        losses.append(loss.item())                             # static output tensor of the graph
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
    judged_by_float64(model, g, "adabelief", steps, "graph replay")
    assert opt.t == steps * 4
    before = HipDevice.pool_stats()["in_use_bytes"]
    graph.destroy()
    del loss
    assert HipDevice.pool_stats()["in_use_bytes"] <= before


def test_capture_rejects_host_transfers(hip):
    from lightgrad_amd.autograd.hip import HipGraph, HipError
    a = hip.from_numpy(np.ones((4, 4), np.float32))
    graph = HipGraph()
    with pytest.raises(HipError, match="cannot be captured"):
        with graph.capture():
            (a + a).numpy()
    # the library left capture mode cleanly: normal work continues
    np.testing.assert_array_equal((a + a).numpy(), np.full((4, 4), 2, np.float32))
    g2 = HipGraph()
    with g2.capture():
        b = a * 3.0
    g2.replay()
    np.testing.assert_array_equal(b.numpy(), np.full((4, 4), 3, np.float32))


def test_graph_with_new_batches_through_async_upload(hip):
    """a captured step fed with a different batch per replay (upload_ into the static inputs) == the eager loop"""
    from lightgrad_amd.autograd.hip import HipGraph
    rng = np.random.RandomState(0)
    batches = [(rng.uniform(0, 1, (16, 20)).astype(np.float32), np.eye(10, dtype=np.float32)[rng.randint(0, 10, 16)]) for _ in range(6)]

    def make():
        np.random.seed(5)
        model = MLP(20, 16, 10).map_parameters(lambda p: p.hip())
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True)
        return model, opt
    # eager reference
    model_e, opt_e = make()
    eager = []
    for x, t in batches:
        l = light.loss.mse(model_e(hip.from_numpy(x)), hip.from_numpy(t))
        opt_e.zero_grad()
        l.backward()
        opt_e.step()
        eager.append(l.item())
    # graph: first two batches eagerly (warm-up), the rest by replay with uploads
    model_g, opt_g = make()
    xs, ts = hip.from_numpy(batches[0][0]), hip.from_numpy(batches[0][1])

    def step():
        l = light.loss.mse(model_g(xs), ts)
        opt_g.zero_grad()
        l.backward()
        opt_g.step()
        return l
    got = [step().item()]
    xs.upload_(batches[1][0])
    ts.upload_(batches[1][1])
    got.append(step().item())
    graph = HipGraph()
    with graph.capture():
        loss = step()
    opt_g.t -= 4
    for x, t in batches[2:]:
        xs.upload_(x)
        ts.upload_(t)
        graph.replay()
        opt_g.on_graph_replay()
        got.append(loss.item())
    np.testing.assert_allclose(got, eager, rtol=1e-6)
    for p, q in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(q.numpy(), p.numpy(), rtol=1e-6, atol=1e-7)


def test_graphed_step_helper(hip):
    from lightgrad_amd.autograd.hip import GraphedStep
    g = load_golden("mlp_small_adabelief.npz")
    model, opt, step, steps = build(hip, g, "adabelief", device_step=True)
    graphed = GraphedStep(step, optimizers=[opt], warmup=2)
    losses = [graphed().item() for _ in range(steps)]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
    judged_by_float64(model, g, "adabelief", steps, "GraphedStep")
    assert opt.t == steps * 4
    graphed.destroy()


def test_capture_without_warmup_allocates_inside_the_capture(hip):
    """relaxed capture mode: pool misses fall through to hipMalloc while capturing; the blocks stay pinned to the graph"""
    from lightgrad_amd.autograd.hip import HipGraph, HipDevice
    a = hip.from_numpy(np.ones((1000, 1003), np.float32))
    HipDevice.trim_pool()
    before = HipDevice.pool_stats()["hip_malloc_calls"]
    g = HipGraph()
    with g.capture():
        b = (a * 2.0 + 1.0).exp().sum()
    assert HipDevice.pool_stats()["hip_malloc_calls"] > before
    g.replay()
    np.testing.assert_allclose(b.item(), 1000 * 1003 * np.exp(3.0), rtol=1e-5)
    a.fill(0.0)                                   # new input values, same graph
    g.replay()
    np.testing.assert_allclose(b.item(), 1000 * 1003 * np.exp(1.0), rtol=1e-5)
