"""The optimizer's update applied by the kernels that make the gradients (optim.Adam.fuse_update_into_backward; include/lghip.h:
lg_adam_plan_* / lg_adam_epilogue_*): `p += compute_delta(p.grad, i)` of the reference (optim.py:10-13, :47-52) without a launch
of its own.  Same per-element arithmetic as the optimizer's launch, so everything here is compared BIT FOR BIT with it, and the
full-size trajectory against the fixture recorded from the reference."""
import numpy as np
import pytest
import lightgrad_amd as light
from conftest import load_golden
import np_oracle as O
from test_cpu_backend import MLP

pytestmark = pytest.mark.gpu


def setup(hip, w0, x, onehot, d_in, d_hid, d_out, opt_cls, in_backward, **kw):
    from lightgrad_amd.dist import DataParallel, SingleProcess
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters(w0)
    model.map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
    opt = opt_cls(model.parameters(), lr=1e-3, fused=True, device_step=True, **kw)
    dp.attach(opt)
    if in_backward:
        opt.fuse_update_into_backward()
    tx, tt = hip.from_numpy(x), hip.from_numpy(onehot)

    def step():
        loss = light.loss.mse(model(tx), tt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    return model, opt, step


def run(hip, in_backward, steps, opt_cls, dims=(64, 48, 10), batch=32, seed=3, **kw):
    w0, x, onehot, _ = O.synthetic_mlp_problem(seed, dims[0], dims[1], dims[2], batch)
    model, opt, step = setup(hip, w0, x, onehot, dims[0], dims[1], dims[2], opt_cls, in_backward, **kw)
    losses, launched = [], []
    for _ in range(steps):
        losses.append(step().item())
        launched.append(opt.last_step_launched_update)
    grads = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
    return losses, {n: p.numpy().copy() for n, p in model.named_parameters()}, grads, launched, opt


@pytest.mark.parametrize("opt_cls", [light.optim.AdaBelief, light.optim.Adam], ids=["adabelief", "adam"])
@pytest.mark.parametrize("dims,batch", [((64, 48, 10), 32), ((100, 70, 10), 200), ((784, 512, 10), 1024), ((40, 36, 24), 16)],
                         ids=["small", "ragged_tiles", "mnist_mlp", "no_head_kernel"])
def test_same_bits_as_the_update_launch(hip, opt_cls, dims, batch):
    steps = 5
    l_ref, w_ref, g_ref, _, opt_ref = run(hip, False, steps, opt_cls, dims, batch)
    l_new, w_new, g_new, launched, opt = run(hip, True, steps, opt_cls, dims, batch)
    np.testing.assert_array_equal(l_new, l_ref)
    for n in w_ref:
        np.testing.assert_array_equal(g_new[n], g_ref[n], err_msg=n)            # p.grad is still written
        np.testing.assert_array_equal(w_new[n], w_ref[n], err_msg=n)
    assert opt.t == opt_ref.t == steps * 4
    assert int(opt._backward_update.steps.numpy()[opt._backward_update.parity]) == steps
    # every gradient of this model is made by a GEMM (+ row sums) or by the head kernel: the step has no update launch at all
    assert launched == [False] * steps


def test_full_size_trajectory_vs_reference(hip):
    """BASELINE config #3 with the update inside the backward kernels: the fixture recorded from the reference"""
    g = load_golden("mlp_full_adabelief.npz")
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    w0, x, onehot, labels = O.synthetic_mlp_problem(seed, d_in, d_hid, d_out, batch)
    model, opt, step = setup(hip, w0, x, onehot, d_in, d_hid, d_out, light.optim.AdaBelief, True)
    losses = [step().item() for _ in range(steps)]
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    sample = lambda a: a.reshape(-1)[::max(1, a.size // 64)][:64]   # noqa: E731
    for n, p in model.named_parameters():
        np.testing.assert_allclose(sample(p.numpy()), g["wfsample/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
        w = p.numpy().astype(np.float64)
        np.testing.assert_allclose([w.sum(), np.abs(w).sum()], g["wfsum/" + n], rtol=1e-4, atol=1e-3)


def test_replayed_graph_of_two_steps_equals_eager_bits(hip):
    from lightgrad_amd.autograd.hip import HipGraph
    dims, batch = (784, 512, 10), 1024
    w0, x, onehot, _ = O.synthetic_mlp_problem(5, dims[0], dims[1], dims[2], batch)
    model_e, opt_e, step_e = setup(hip, w0, x, onehot, *dims, light.optim.AdaBelief, True)
    eager = [step_e().item() for _ in range(10)]
    model, opt, step = setup(hip, w0, x, onehot, *dims, light.optim.AdaBelief, True)
    losses = [step().item() for _ in range(2)]
    graph = HipGraph()
    with graph.capture():
        first = step()
        second = step()
    opt.t -= 8
    assert graph.kernel_count() == 6                           # 3 launches per step: forward GEMM, head forward (+ the head's input gradients), dW2 + dW1 + dx + loss
    for _ in range(4):
        graph.replay()
        opt.on_graph_replay(2)
        losses += [first.item(), second.item()]
    np.testing.assert_array_equal(losses, eager)
    for (n, p), (_, q) in zip(model.named_parameters(), model_e.named_parameters()):
        np.testing.assert_array_equal(p.numpy(), q.numpy(), err_msg=n)
    # and eager steps carry on from where the graph left off
    np.testing.assert_array_equal([step().item()], [step_e().item()])


def test_a_graph_with_an_odd_number_of_steps_is_refused(hip):
    from lightgrad_amd.autograd.hip import HipGraph
    w0, x, onehot, _ = O.synthetic_mlp_problem(5, 64, 48, 10, 32)
    model, opt, step = setup(hip, w0, x, onehot, 64, 48, 10, light.optim.AdaBelief, True)
    step()
    before = {n: p.numpy().copy() for n, p in model.named_parameters()}
    with pytest.raises(RuntimeError, match="EVEN number of steps"):
        with HipGraph().capture():
            step()
    opt.t -= 4
    for n, p in model.named_parameters():                      # nothing ran, and the parameters still point at their values
        np.testing.assert_array_equal(p.numpy(), before[n], err_msg=n)
    step()


def test_gradients_no_kernel_takes_are_applied_by_finish(hip):
    """a model whose parameters get their gradients from elementwise / reduction kernels: one launch applies all of them"""
    from lightgrad_amd.dist import DataParallel, SingleProcess

    def build(in_backward):
        np.random.seed(11)
        w = hip.from_numpy(np.random.uniform(-1, 1, (33, 7)).astype(np.float32))
        b = hip.from_numpy(np.random.uniform(-1, 1, (7,)).astype(np.float32))
        dp = DataParallel((w, b), SingleProcess(), flatten=True)
        opt = light.optim.AdaBelief((w, b), lr=1e-2, fused=True, device_step=True)
        dp.attach(opt)
        if in_backward:
            opt.fuse_update_into_backward()
        x = hip.from_numpy(np.random.uniform(-1, 1, (33, 7)).astype(np.float32), requires_grad=False)
        for _ in range(3):
            loss = ((w * x + b).tanh() ** 2).sum()
            opt.zero_grad()
            loss.backward()
            opt.step()
        return w.numpy(), b.numpy(), opt
    w0, b0, _ = build(False)
    w1, b1, opt = build(True)
    np.testing.assert_array_equal(w1, w0)
    np.testing.assert_array_equal(b1, b0)
    assert opt.last_step_launched_update is True


def test_a_weight_shared_by_two_layers_is_refused(hip):
    """the update rides with the FIRST kernel that writes the gradient: a second writer in the same step must not pass silently"""
    from lightgrad_amd.dist import DataParallel, SingleProcess
    from lightgrad_amd.autograd.hip import HipError
    np.random.seed(2)
    lin = light.nn.Linear(48, 48)
    lin.map_parameters(lambda p: p.hip())
    dp = DataParallel(lin.parameters(), SingleProcess(), flatten=True)
    opt = light.optim.AdaBelief(lin.parameters(), lr=1e-3, fused=True, device_step=True)
    dp.attach(opt)
    opt.fuse_update_into_backward()
    x = hip.from_numpy(np.random.uniform(-1, 1, (64, 48)).astype(np.float32))
    loss = lin(lin(x).tanh()).sum()
    opt.zero_grad()
    with pytest.raises(HipError, match="again after its optimizer update was applied"):
        loss.backward()
    opt._backward_update.disarm()
