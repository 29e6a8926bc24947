"""csrc/head.hip - the skinny output layer + loss in two launches - against numpy (float64) through the raw C ABI,
against the unfused tape on the CPU backend (itself pinned to the reference's fixtures), and the optimizer step
counter that rides along in the loss kernel."""
import ctypes
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from test_cpu_backend import MLP

pytestmark = pytest.mark.gpu


def _relu(a):
    with np.errstate(invalid="ignore"):
        return np.maximum(a, 0)


@pytest.mark.parametrize("rows,hidden,outs,relu,with_bias", [
    (1024, 512, 10, 1, True),          # the MNIST head
    (1, 4, 1, 0, False), (3, 8, 2, 1, True), (65, 36, 7, 1, True), (130, 100, 16, 0, True), (257, 1024, 16, 1, False),
    (4100, 64, 10, 1, True),           # more rows than the grid has waves: the row loop
])
def test_head_forward_c_abi(hip, rows, hidden, outs, relu, with_bias):
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(rows + hidden + outs)
    x = rng.uniform(-1, 1, (rows, hidden)).astype(np.float32)
    x[0, 0] = 0.0
    w = (rng.uniform(-1, 1, (outs, hidden)) / np.sqrt(hidden)).astype(np.float32)
    b = rng.uniform(-1, 1, (outs,)).astype(np.float32)
    t = rng.uniform(0, 1, (rows, outs)).astype(np.float32)
    tx, tw, tb, tt = (hip.from_numpy(a, requires_grad=False) for a in (x, w, b, t))
    y, err, row_loss, loss = hip.empty((rows, outs)), hip.empty((rows, outs)), hip.empty((rows,)), hip.empty(())
    L.check(lib.lg_head_fwd_f32(tx.ptr, hidden, relu, tw.ptr, tb.ptr if with_bias else None, tt.ptr, y.ptr, err.ptr, row_loss.ptr,
                                rows, hidden, outs))
    L.check(lib.lg_mse_finalize_f32(row_loss.ptr, rows, rows * outs, loss.ptr))
    a64 = (_relu(x) if relu else x).astype(np.float64)
    y_ref = a64 @ w.astype(np.float64).T + (b if with_bias else 0)
    e_ref = y_ref - t
    np.testing.assert_allclose(y.numpy(), y_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(err.numpy(), e_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(loss.item(), (e_ref ** 2).mean() / 2, rtol=1e-5)
    # err is exactly y + (-target) of the y that was written (one rounding, like the tape's `y - y_hat`)
    np.testing.assert_array_equal(err.numpy(), y.numpy() + (-t))
    # bit-reproducible from launch to launch
    first = (y.numpy().copy(), loss.numpy().copy())
    L.check(lib.lg_head_fwd_f32(tx.ptr, hidden, relu, tw.ptr, tb.ptr if with_bias else None, tt.ptr, y.ptr, err.ptr, row_loss.ptr,
                                rows, hidden, outs))
    L.check(lib.lg_mse_finalize_f32(row_loss.ptr, rows, rows * outs, loss.ptr))
    np.testing.assert_array_equal(y.numpy(), first[0])
    np.testing.assert_array_equal(loss.numpy(), first[1])
    # the backward launch finishes the same loss with its spare workgroup: same summation order, same bits
    loss2, dx = hip.empty(()), hip.empty((rows, hidden))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, relu, err.ptr, tw.ptr, dx.ptr, None, None, 0, None, 0, rows, hidden, outs, row_loss.ptr, loss2.ptr))
    np.testing.assert_array_equal(loss2.numpy(), first[1])
    assert lib.lg_head_bwd_f32(tx.ptr, hidden, relu, err.ptr, tw.ptr, dx.ptr, None, None, 0, None, 0, rows, hidden, outs, row_loss.ptr, None) != 0


@pytest.mark.parametrize("rows,hidden,outs,relu", [
    (1024, 512, 10, 1), (1, 4, 1, 0), (3, 8, 2, 1), (65, 36, 7, 1), (130, 100, 16, 0), (700, 1000, 3, 1), (5000, 40, 10, 1),
])
def test_head_backward_c_abi(hip, rows, hidden, outs, relu):
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(7 * rows + hidden + outs)
    x = rng.uniform(-1, 1, (rows, hidden)).astype(np.float32)
    x[0, 0] = 0.0                                                        # relu.backward passes the gradient at exactly 0
    if hidden > 1:
        x[0, 1] = -0.0
    g = rng.uniform(-1, 1, (rows, outs)).astype(np.float32)
    w = (rng.uniform(-1, 1, (outs, hidden)) / np.sqrt(hidden)).astype(np.float32)
    tx, tg, tw = (hip.from_numpy(a, requires_grad=False) for a in (x, g, w))
    dx, gpre = hip.empty((rows, hidden)), hip.empty((rows, hidden))
    dw0 = rng.uniform(-1, 1, (outs, hidden)).astype(np.float32)
    db0 = rng.uniform(-1, 1, (outs,)).astype(np.float32)
    g64, w64 = g.astype(np.float64), w.astype(np.float64)
    a64 = (_relu(x) if relu else x).astype(np.float64)
    dx_ref, dw_ref, db_ref = g64 @ w64, g64.T @ a64, g64.sum(axis=0)
    scale = np.abs(g64).T @ np.abs(a64) + 1e-30                          # what rounding errors are relative to
    for accumulate in (0, 1):
        dw, db = hip.from_numpy(dw0.copy(), requires_grad=False), hip.from_numpy(db0.copy(), requires_grad=False)
        L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, relu, tg.ptr, tw.ptr, dx.ptr, gpre.ptr if relu else None,
                                    dw.ptr, accumulate, db.ptr, accumulate, rows, hidden, outs, None, None))
        np.testing.assert_allclose(dx.numpy(), dx_ref, rtol=1e-5, atol=1e-6)
        if relu:
            np.testing.assert_array_equal(gpre.numpy(), dx.numpy() * (x >= 0))           # relu.backward: g * (t >= 0)
        got_dw = dw.numpy().astype(np.float64) - (dw0 if accumulate else 0)
        assert np.max(np.abs(got_dw - dw_ref) / scale) < 1e-5            # fp32 forward-error bound, north-star tolerance
        np.testing.assert_allclose(db.numpy().astype(np.float64) - (db0 if accumulate else 0), db_ref, rtol=1e-5, atol=1e-5 * max(1, rows / 64))
    # optional outputs: dW only / dx only; and bit-reproducibility
    dw_a, dw_b = hip.empty((outs, hidden)), hip.empty((outs, hidden))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, relu, tg.ptr, tw.ptr, None, None, dw_a.ptr, 0, None, 0, rows, hidden, outs, None, None))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, relu, tg.ptr, tw.ptr, dx.ptr, None, dw_b.ptr, 0, None, 0, rows, hidden, outs, None, None))
    np.testing.assert_array_equal(dw_a.numpy(), dw_b.numpy())
    dx_only = hip.empty((rows, hidden))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, relu, tg.ptr, tw.ptr, dx_only.ptr, None, None, 0, None, 0, rows, hidden, outs, None, None))
    np.testing.assert_array_equal(dx_only.numpy(), dx.numpy())


def test_head_c_abi_rejects_what_it_cannot_do(hip):
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    t = hip.zeros((8, 8), requires_grad=False)
    assert lib.lg_head_fwd_f32(t.ptr, 8, 0, t.ptr, None, t.ptr, t.ptr, t.ptr, t.ptr, 8, 8, 17) != 0          # > 16 outputs
    assert b"outs" in lib.lg_last_error()
    assert lib.lg_head_fwd_f32(t.ptr, 6, 0, t.ptr, None, t.ptr, t.ptr, t.ptr, t.ptr, 8, 6, 4) != 0           # hidden % 4
    assert lib.lg_head_fwd_f32(t.ptr + 4, 8, 0, t.ptr, None, t.ptr, t.ptr, t.ptr, t.ptr, 7, 8, 4) != 0       # misaligned x
    assert lib.lg_head_bwd_f32(t.ptr, 8, 0, t.ptr, t.ptr, t.ptr, t.ptr, None, 0, None, 0, 8, 8, 4, None, None) != 0   # gpre without relu
    assert lib.lg_head_bwd_f32(None, 8, 0, t.ptr, t.ptr, None, None, None, 0, None, 0, 8, 8, 4, None, None) != 0


@pytest.mark.parametrize("d_in,d_hid,d_out,batch", [(20, 16, 7, 33), (784, 512, 10, 1024), (12, 8, 16, 5), (30, 24, 1, 64)])
def test_tape_with_fused_head_equals_cpu_backend(hip, d_in, d_hid, d_out, batch):
    """Linear -> relu -> Linear(<= 16) -> mse: the tape must hand out the same values, ALL gradients included (the lazy relu
    output's own, which is dx WITHOUT the relu mask), as the unfused CPU backend; 3 launches carry forward head + loss,
    backward head, nothing else"""
    rng = np.random.RandomState(d_in + d_out)
    xn = rng.uniform(-1, 1, (batch, d_in)).astype(np.float32)
    tn = rng.uniform(0, 1, (batch, d_out)).astype(np.float32)
    res = {}
    for cls in (CpuTensor, hip):
        np.random.seed(11)
        model = MLP(d_in, d_hid, d_out)
        if cls is hip:
            model.map_parameters(lambda p: p.hip())
        x = cls.from_numpy(xn)
        pre = model.l1(x)
        h = pre.relu()
        y = model.l2(h)
        if cls is hip:
            assert y.is_lazy() and h.is_lazy()
        loss = light.loss.mse(y, cls.from_numpy(tn, requires_grad=False))
        if cls is hip:
            assert not y.is_lazy() and h.is_lazy()                       # the loss made y real in its own launch
            assert loss.is_lazy()                                        # ... and leaves its scalar to the backward launch
        loss.backward()
        if cls is hip:
            assert h.is_lazy() and not loss.is_lazy()
        res[cls] = [loss.numpy(), y.numpy(), y.grad.numpy(), h.grad.numpy(), pre.grad.numpy(), x.grad.numpy()] + \
                   [p.grad.numpy() for p in model.parameters()]
    for got, ref in zip(res[hip], res[CpuTensor]):
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)


def test_lazy_head_behaves_like_a_tensor_everywhere_else(hip):
    """a skinny Linear whose output is NOT fed to mse is computed by the plain kernel when somebody looks, snapshots its
    sources, and a second gradient contribution to the relu output bypasses the relu shortcut"""
    rng = np.random.RandomState(3)
    xn = rng.uniform(-1, 1, (9, 12)).astype(np.float32)
    np.random.seed(2)
    lin = light.nn.Linear(12, 4)
    w, b = lin.weight.numpy().copy(), lin.bias.numpy().copy()
    lin.map_parameters(lambda p: p.hip())
    x = hip.from_numpy(xn)
    y = lin(x)
    assert y.is_lazy()
    with light.no_grad():
        x += 1.0                                                         # y was defined from the OLD x
    np.testing.assert_allclose(y.numpy(), xn.astype(np.float64) @ w.T + b, rtol=1e-5, atol=1e-6)
    y2 = lin(x.relu())
    with light.no_grad():
        lin.weight.fill(0.0)                                             # ... and from the OLD weights
    np.testing.assert_allclose(y2.numpy(), _relu(xn + 1.0).astype(np.float64) @ w.T + b, rtol=1e-5, atol=1e-6)
    # elementwise / reduction / cross-entropy consumers of a lazy head
    lin2 = light.nn.Linear(12, 5)
    w2, b2 = lin2.weight.numpy().copy(), lin2.bias.numpy().copy()
    lin2.map_parameters(lambda p: p.hip())
    z = lin2(hip.from_numpy(xn))
    ref = xn.astype(np.float64) @ w2.T + b2
    np.testing.assert_allclose((z * 2.0).numpy(), 2 * ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lin2(hip.from_numpy(xn)).sum().numpy(), ref.sum(), rtol=1e-5)
    labels = rng.randint(0, 5, 9).astype(np.int64)
    ce = light.loss.cross_entropy(lin2(hip.from_numpy(xn)), hip.from_numpy(labels, requires_grad=False))
    p = np.exp(ref - ref.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    np.testing.assert_allclose(ce.item(), -np.log(p[np.arange(9), labels]).mean(), rtol=1e-5)
    # relu output used twice: its gradient is a sum, so relu.backward must run its own kernel
    res = {}
    for cls in (CpuTensor, hip):
        np.random.seed(5)
        l1, l2 = light.nn.Linear(12, 8), light.nn.Linear(8, 3)
        if cls is hip:
            l1.map_parameters(lambda p: p.hip())
            l2.map_parameters(lambda p: p.hip())
        xx = cls.from_numpy(xn)
        pre = l1(xx)
        h = pre.relu()
        out = l2(h)
        loss = light.loss.mse(out, cls.zeros((9, 3), requires_grad=False)) + (h * h).sum() * 0.01
        loss.backward()
        res[cls] = [loss.numpy(), pre.grad.numpy(), xx.grad.numpy()] + [q.grad.numpy() for q in list(l1.parameters()) + list(l2.parameters())]
    for got, ref in zip(res[hip], res[CpuTensor]):
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("loss_kind", ["head", "plain_mse", "other"])
def test_step_counter_rides_in_the_loss_kernel(hip, loss_kind):
    """device_step optimizers: the increment of the step counter is carried by the next training step's fused loss (no
    launch of its own); a forward pass under no_grad must not carry it, and without a carrier the optimizer flushes it.
    Pinned by equality with the host-scalar optimizer, whose bias corrections depend on the step number."""
    from lightgrad_amd.autograd.hip import HipTensor
    rng = np.random.RandomState(0)
    xn = rng.uniform(0, 1, (16, 12)).astype(np.float32)
    tn = rng.uniform(0, 1, (16, 4)).astype(np.float32)
    finals = []
    import gc
    gc.collect()                                                         # optimizers of earlier tests are gone
    for device_step in (False, True):
        np.random.seed(1)
        model = MLP(12, 8, 4).map_parameters(lambda p: p.hip())
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-2, fused=True, device_step=device_step)
        x, t = hip.from_numpy(xn), hip.from_numpy(tn, requires_grad=False)
        for step in range(4):
            y = model(x)
            if loss_kind == "head":
                loss = light.loss.mse(y, t)
            elif loss_kind == "plain_mse":
                loss = light.loss.mse(y * 1.0, t)                        # y is materialised by the multiplication: lg_mse_f32
            else:
                loss = ((y - t) ** 2).mean()                             # no fused loss at all: the optimizer flushes
            opt.zero_grad()
            loss.backward()
            opt.step()
            if device_step:
                with light.no_grad():
                    light.loss.mse(model(x), t)                          # evaluation pass: must not advance anything
                c = opt._step_counter.numpy()[0]
                assert c == step + 1, (c, step)                          # advanced by the optimizer step itself
        finals.append([p.numpy() for p in model.parameters()])
        if device_step:
            assert opt._step_counter.numpy()[0] == 4
    for a, b in zip(*finals):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("flat", [False, True], ids=["per_parameter_kernels", "flat_bucket_one_launch"])
@pytest.mark.parametrize("scenario", ["warm_whole_step", "unfused_loss_in_graph", "graph_fwd_bwd_eager_optimizer",
                                      "separate_graphs_fwd_bwd_first", "separate_graphs_optimizer_first",
                                      "capture_then_eager", "three_steps_per_graph", "three_steps_per_graph_unfused_loss"])
def test_step_counter_stays_right_across_graph_capture(hip, scenario, flat):
    """one increment of the device step counter per training step, whatever mix of eager steps, captures and replays - also
    with forward+backward and the optimizer in two SEPARATE graphs (round 2's carried increment counted twice there):
    the weights after 7 steps must equal those of the host-scalar optimizer (its bias corrections depend on the step)"""
    from lightgrad_amd.autograd.hip import HipGraph, HipTensor
    import gc
    gc.collect()
    rng = np.random.RandomState(0)
    xn = rng.uniform(0, 1, (16, 12)).astype(np.float32)
    tn = rng.uniform(0, 1, (16, 4)).astype(np.float32)
    total = 8 if "three_steps" in scenario else 7

    def build(device_step):
        np.random.seed(1)
        model = MLP(12, 8, 4).map_parameters(lambda p: p.hip())
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-2, fused=True, device_step=device_step)
        if flat and device_step:
            from lightgrad_amd.dist import DataParallel, SingleProcess
            DataParallel(model.parameters(), SingleProcess(), flatten=True).attach(opt)
        x, t = hip.from_numpy(xn), hip.from_numpy(tn, requires_grad=False)

        def fwd_bwd():
            y = model(x)
            loss = ((y - t) ** 2).mean() if "unfused_loss" in scenario else light.loss.mse(y, t)
            opt.zero_grad()
            loss.backward()
            return loss

        def step():
            loss = fwd_bwd()
            opt.step()
            return loss
        return model, opt, fwd_bwd, step

    model, opt, _, step = build(False)
    for _ in range(total):
        step()
    want = [p.numpy() for p in model.parameters()]

    model, opt, fwd_bwd, step = build(True)
    n_params = len(opt.parameters)
    for _ in range(2):
        step()
    done = 2
    g, g2 = HipGraph(), HipGraph()
    if scenario.startswith("separate_graphs"):
        if scenario.endswith("optimizer_first"):
            fwd_bwd()                                                    # gradients for the optimizer capture to read
            with g2.capture():
                opt.step()
            with g.capture():
                fwd_bwd()
        else:
            with g.capture():
                fwd_bwd()
            with g2.capture():
                opt.step()
        opt.t -= n_params
        while done < total:
            g.replay()
            g2.replay()
            opt.on_graph_replay()
            done += 1
    elif scenario == "graph_fwd_bwd_eager_optimizer":
        with g.capture():
            fwd_bwd()
        while done < total:
            g.replay()
            opt.step()
            done += 1
    elif "three_steps" in scenario:
        with g.capture():
            for _ in range(3):
                step()
        opt.t -= 3 * n_params
        while done < total:                                              # 2 eager + 2 x 3 replayed
            g.replay()
            opt.on_graph_replay(3)
            done += 3
    elif scenario == "capture_then_eager":
        with g.capture():
            step()
        opt.t -= n_params
        g.replay()
        opt.on_graph_replay()
        done += 1
        while done < total:                                              # back to the python tape after a replay
            step()
            done += 1
    else:
        with g.capture():
            step()
        opt.t -= n_params
        while done < total:
            g.replay()
            opt.on_graph_replay()
            done += 1
    assert opt._step_counter.numpy()[0] == total
    assert opt.t == total * n_params
    for a, b in zip([p.numpy() for p in model.parameters()], want):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    g.destroy()
    g2.destroy()


def test_loss_read_before_backward_equals_loss_finished_by_backward(hip):
    rng = np.random.RandomState(5)
    xn = rng.uniform(-1, 1, (40, 24)).astype(np.float32)
    tn = rng.uniform(0, 1, (40, 6)).astype(np.float32)
    values = []
    for read_first in (True, False):
        np.random.seed(8)
        model = MLP(24, 16, 6).map_parameters(lambda p: p.hip())
        loss = light.loss.mse(model(hip.from_numpy(xn)), hip.from_numpy(tn, requires_grad=False))
        assert loss.is_lazy()
        if read_first:
            values.append(loss.item())                                   # lg_mse_finalize_f32
            assert not loss.is_lazy()
        loss.backward()                                                  # must not overwrite / re-finish a loss that is real
        values.append(loss.item())
        grads = [p.grad.numpy() for p in model.parameters()]
        assert all(np.isfinite(g).all() and np.abs(g).sum() > 0 for g in grads)
    assert values[0] == values[1] == values[2]
    # evaluation only (no backward at all) under no_grad
    with light.no_grad():
        ev = light.loss.mse(model(hip.from_numpy(xn)), hip.from_numpy(tn, requires_grad=False))
    assert ev.item() == values[0]
    # a non-unit seed: mse.backward multiplies err by it (a new tensor), the loss is then finished on demand
    np.random.seed(8)
    model = MLP(24, 16, 6).map_parameters(lambda p: p.hip())
    loss = light.loss.mse(model(hip.from_numpy(xn)), hip.from_numpy(tn, requires_grad=False))
    total = loss * 3.0
    total.backward()
    np.testing.assert_allclose(total.item(), 3.0 * values[0], rtol=1e-6)
    for p, g in zip(model.parameters(), grads):
        np.testing.assert_allclose(p.grad.numpy(), 3.0 * g, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,hidden,outs,with_bias", [(1024, 512, 10, True), (3, 8, 2, True), (65, 36, 7, False), (257, 1024, 16, True), (4100, 64, 10, True)])
def test_forward_launch_writes_the_backward_tiles_ahead(hip, rows, hidden, outs, with_bias):
    """lg_head_fwd_grad_f32: dx = err @ W and g_pre = dx * (x >= 0) from the forward launch are lg_head_bwd_f32's for g = err, bit for bit;
    y / err / row sums are those of lg_head_fwd_f32"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(rows * 3 + hidden + outs)
    x = rng.uniform(-1, 1, (rows, hidden)).astype(np.float32)
    x[0, 0], x[0, 1] = 0.0, -0.0
    w = (rng.uniform(-1, 1, (outs, hidden)) / np.sqrt(hidden)).astype(np.float32)
    b = rng.uniform(-1, 1, (outs,)).astype(np.float32)
    t = rng.uniform(0, 1, (rows, outs)).astype(np.float32)
    tx, tw, tb, tt = (hip.from_numpy(a, requires_grad=False) for a in (x, w, b, t))
    bp = tb.ptr if with_bias else None
    y0, e0, r0 = hip.empty((rows, outs)), hip.empty((rows, outs)), hip.empty((rows,))
    L.check(lib.lg_head_fwd_f32(tx.ptr, hidden, 1, tw.ptr, bp, tt.ptr, y0.ptr, e0.ptr, r0.ptr, rows, hidden, outs))
    y1, e1, r1 = hip.empty((rows, outs)), hip.empty((rows, outs)), hip.empty((rows,))
    dx1, gp1 = hip.empty((rows, hidden)), hip.empty((rows, hidden))
    L.check(lib.lg_head_fwd_grad_f32(tx.ptr, hidden, 1, tw.ptr, bp, tt.ptr, y1.ptr, e1.ptr, r1.ptr, dx1.ptr, gp1.ptr, rows, hidden, outs))
    for a, c in ((y0, y1), (e0, e1), (r0, r1)):
        np.testing.assert_array_equal(a.numpy(), c.numpy())
    dx2, gp2 = hip.empty((rows, hidden)), hip.empty((rows, hidden))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, 1, e0.ptr, tw.ptr, dx2.ptr, gp2.ptr, None, 0, None, 0, rows, hidden, outs, None, None))
    np.testing.assert_array_equal(dx1.numpy(), dx2.numpy())
    np.testing.assert_array_equal(gp1.numpy(), gp2.numpy())
    ref = e0.numpy().astype(np.float64) @ w.astype(np.float64)
    np.testing.assert_allclose(dx1.numpy(), ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(gp1.numpy(), dx1.numpy() * (x >= 0))
    # one without the other, or without the relu: refused
    assert lib.lg_head_fwd_grad_f32(tx.ptr, hidden, 1, tw.ptr, bp, tt.ptr, y1.ptr, e1.ptr, r1.ptr, dx1.ptr, None, rows, hidden, outs) != 0
    assert lib.lg_head_fwd_grad_f32(tx.ptr, hidden, 0, tw.ptr, bp, tt.ptr, y1.ptr, e1.ptr, r1.ptr, dx1.ptr, gp1.ptr, rows, hidden, outs) != 0


@pytest.mark.parametrize("rows", [64, 1000, 1024])
def test_backward_tiles_alone_use_short_tiles(hip, rows):
    """lg_head_bwd_f32 without weight gradients (tiles only, 64 rows each) writes what the launch with slabs (256-row tiles) writes"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    hidden, outs = 512, 10
    rng = np.random.RandomState(rows)
    x = rng.uniform(-1, 1, (rows, hidden)).astype(np.float32)
    g = rng.uniform(-1, 1, (rows, outs)).astype(np.float32)
    w = rng.uniform(-1, 1, (outs, hidden)).astype(np.float32)
    tx, tg, tw = (hip.from_numpy(a, requires_grad=False) for a in (x, g, w))
    outs_a = [hip.empty((rows, hidden)) for _ in range(2)]
    outs_b = [hip.empty((rows, hidden)) for _ in range(2)]
    dw = hip.empty((outs, hidden))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, 1, tg.ptr, tw.ptr, outs_a[0].ptr, outs_a[1].ptr, None, 0, None, 0, rows, hidden, outs, None, None))
    L.check(lib.lg_head_bwd_f32(tx.ptr, hidden, 1, tg.ptr, tw.ptr, outs_b[0].ptr, outs_b[1].ptr, dw.ptr, 0, None, 0, rows, hidden, outs, None, None))
    for a, c in zip(outs_a, outs_b):
        np.testing.assert_array_equal(a.numpy(), c.numpy())


@pytest.mark.parametrize("rows,d_in,hidden,outs,expect", [
    (1024, 784, 512, 10, 1),          # the MNIST MLP: one launch
    (1000, 300, 512, 3, 1),           # a last row of tiles with 40 rows
    (64, 64, 96, 16, None), (4096, 128, 64, 10, None), (5000, 100, 512, 10, 2), (256, 784, 2048, 4, 2),
])
def test_hidden_layer_and_head_in_one_launch(hip, rows, d_in, hidden, outs, expect):
    """lg_gemm_bias_head_fwd_f32: the bits of lg_gemm_bias_f32 followed by lg_head_fwd_grad_f32, whether it chains the two in one
    launch (chain = 1: an experiment the tape does not use, measured slower) or not - and again when launched many times in a row
    (tickets and flags go back to zero)"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(rows + hidden)
    x = rng.uniform(-1, 1, (rows, d_in)).astype(np.float32)
    w1 = (rng.uniform(-1, 1, (hidden, d_in)) / np.sqrt(d_in)).astype(np.float32)
    b1 = rng.uniform(-0.1, 0.1, (hidden,)).astype(np.float32)
    w2 = (rng.uniform(-1, 1, (outs, hidden)) / np.sqrt(hidden)).astype(np.float32)
    b2 = rng.uniform(-1, 1, (outs,)).astype(np.float32)
    t = rng.uniform(0, 1, (rows, outs)).astype(np.float32)
    tx, tw1, tb1, tw2, tb2, tt = (hip.from_numpy(a, requires_grad=False) for a in (x, w1, b1, w2, b2, t))

    def outputs():
        return [hip.empty(s) for s in ((rows, hidden), (rows, outs), (rows, outs), (rows,), (rows, hidden), (rows, hidden))]
    pre0, y0, e0, r0, dx0, gp0 = ref = outputs()
    L.check(lib.lg_gemm_bias_f32(0, 1, rows, hidden, d_in, tx.ptr, d_in, 0, tw1.ptr, d_in, 0, pre0.ptr, hidden, 0, 1, tb1.ptr))
    L.check(lib.lg_head_fwd_grad_f32(pre0.ptr, hidden, 1, tw2.ptr, tb2.ptr, tt.ptr, y0.ptr, e0.ptr, r0.ptr, dx0.ptr, gp0.ptr, rows, hidden, outs))
    np.testing.assert_allclose(pre0.numpy(), x.astype(np.float64) @ w1.astype(np.float64).T + b1, rtol=1e-5, atol=1e-5)
    n = ctypes.c_int(0)

    def check_against(got, ref):
        # the chained launch runs its product on ITS tile (64x32, two K-groups), the plain product on whatever the cost model picks
        # (four K-groups for this shape): the same sums in another order.  The head's outputs are the head kernel's for THAT pre, bit for bit
        np.testing.assert_allclose(got[0].numpy(), ref[0].numpy(), rtol=1e-5, atol=2e-6)
        again = outputs()
        L.check(lib.lg_head_fwd_grad_f32(got[0].ptr, hidden, 1, tw2.ptr, tb2.ptr, tt.ptr, again[1].ptr, again[2].ptr, again[3].ptr, again[4].ptr, again[5].ptr,
                                         rows, hidden, outs))
        for a, b in zip(got[1:], again[1:]):
            if a is not None:
                np.testing.assert_array_equal(a.numpy(), b.numpy())
    for rep in range(7):
        chain = 0 if rep == 6 else 1
        got = outputs()
        for g in got:
            g.fill(np.nan)
        pre, y, e, r, dx, gp = got
        L.check(lib.lg_gemm_bias_head_fwd_f32(tx.ptr, d_in, tw1.ptr, d_in, tb1.ptr, pre.ptr, rows, hidden, d_in, 1, tw2.ptr, tb2.ptr, tt.ptr,
                                              y.ptr, e.ptr, r.ptr, dx.ptr, gp.ptr, outs, chain, ctypes.byref(n)))
        check_against(got, ref)
        assert n.value in (1, 2) and (expect is None or n.value == (expect if chain else 2)), n.value
    hip.synchronize() if hasattr(hip, "synchronize") else None
    # without the gradients ahead
    pre, y, e, r, _, _ = got = outputs()
    L.check(lib.lg_gemm_bias_head_fwd_f32(tx.ptr, d_in, tw1.ptr, d_in, tb1.ptr, pre.ptr, rows, hidden, d_in, 1, tw2.ptr, tb2.ptr, tt.ptr,
                                          y.ptr, e.ptr, r.ptr, None, None, outs, 1, None))
    check_against(got[:4] + [None, None], ref)
    assert lib.lg_gemm_bias_head_fwd_f32(tx.ptr, d_in, tw1.ptr, d_in, tb1.ptr, pre.ptr, rows, hidden, d_in, 1, tw2.ptr, tb2.ptr, tt.ptr,
                                         y.ptr, e.ptr, r.ptr, dx0.ptr, None, outs, 1, None) != 0
