"""Fused forms on the HIP path against the unfused tape and the reference fixtures:
nn.Linear as one op with the bias in the GEMM epilogue, loss.mse in one kernel, the multi-tensor
optimizer over flat buckets.  All must reproduce the reference trajectory."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from conftest import load_golden
import np_oracle as O
from test_cpu_backend import MLP

pytestmark = pytest.mark.gpu


def test_linear_op_equals_three_op_form(hip):
    rng = np.random.RandomState(0)
    for (batch, d_in, d_out) in [(32, 48, 20), (1024, 784, 512), (1024, 512, 10), (7, 5, 3)]:
        x, w, b = (rng.uniform(-1, 1, s).astype(np.float32) for s in [(batch, d_in), (d_out, d_in), (d_out,)])
        g = rng.uniform(-1, 1, (batch, d_out)).astype(np.float32)
        tx, tw, tb = hip.from_numpy(x), hip.from_numpy(w), hip.from_numpy(b)
        y = tx.linear(tw, tb)
        (y * hip.from_numpy(g, requires_grad=False)).backward(allow_fill=True)
        ux, uw, ub = hip.from_numpy(x), hip.from_numpy(w), hip.from_numpy(b)
        y2 = ux @ uw.T(1, 0) + ub
        (y2 * hip.from_numpy(g, requires_grad=False)).backward(allow_fill=True)
        np.testing.assert_allclose(y.numpy(), y2.numpy(), rtol=1e-6, atol=1e-6)
        ref = x.astype(np.float64) @ w.T + b
        np.testing.assert_allclose(y.numpy(), ref, rtol=1e-5, atol=1e-6 * d_in ** 0.5 * 4)
        # gradients: both forms against float64 products (relative Frobenius, the north star's 1e-5), and against each other within
        # what two different summation orders of K float32 products can differ by (the fused form pairs dW with dx in one launch and
        # splits K there; the three-op form does not)
        from common import rel_frobenius
        g64, x64, w64 = g.astype(np.float64), x.astype(np.float64), w.astype(np.float64)
        for (p, q), ref, k in [((tx, ux), g64 @ w64, d_out), ((tw, uw), g64.T @ x64, batch), ((tb, ub), g64.sum(0), batch)]:
            assert rel_frobenius(p.grad.numpy(), ref) <= 1e-5 and rel_frobenius(q.grad.numpy(), ref) <= 1e-5
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=1e-5, atol=4 * 2.0 ** -24 * k ** 0.5 * np.abs(ref).max())
        assert tw.grad.is_contiguous() and tw.grad.shape == w.shape
        np.testing.assert_allclose(tb.grad.numpy(), g.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    # no bias, batched input
    x3 = rng.uniform(-1, 1, (3, 5, 8)).astype(np.float32)
    w3 = rng.uniform(-1, 1, (4, 8)).astype(np.float32)
    np.testing.assert_allclose(hip.from_numpy(x3).linear(hip.from_numpy(w3)).numpy(), x3 @ w3.T, rtol=1e-5, atol=1e-5)


def test_fused_mse_equals_tape_expression(hip):
    rng = np.random.RandomState(1)
    for shape in [(8, 10), (1024, 10), (300, 700)]:
        y, t = rng.uniform(-1, 1, shape).astype(np.float32), rng.uniform(0, 1, shape).astype(np.float32)
        ty = hip.from_numpy(y)
        l = light.loss.mse(ty, hip.from_numpy(t))
        l.backward()
        cy = CpuTensor.from_numpy(y)
        lc = light.loss.mse(cy, CpuTensor.from_numpy(t))
        lc.backward()
        assert l.shape == ()
        np.testing.assert_allclose(l.item(), lc.item(), rtol=1e-6)
        np.testing.assert_array_equal(ty.grad.numpy(), cy.grad.numpy())          # err * 1.0: exact


def test_flat_buckets_single_launch_update_reproduces_reference(hip):
    from lightgrad_amd.dist import DataParallel, SingleProcess
    g = load_golden("mlp_small_adabelief.npz")
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters({n: g["w0/" + n] for n in O.PARAM_ORDER})
    model.map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
    opt = dp.attach(light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True, grad_scale=dp.grad_scale))
    for (n, p) in model.named_parameters():                                     # same objects, new home
        np.testing.assert_array_equal(p.numpy(), g["w0/" + n])
        assert p.data is dp.flat_parameters.data
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), g["labels"]] = 1
    x, t = hip.from_numpy(g["x"]), hip.from_numpy(onehot)
    losses = []
    for _ in range(steps):
        l = light.loss.mse(model(x), t)
        opt.zero_grad()
        l.backward()
        dp.sync_gradients()
        opt.step()
        losses.append(l.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.numpy(), g["wf/" + n], rtol=1e-4, atol=2e-6, err_msg=n)
    # two float32 runs are loose by nature: the float64 run of the same tape is the judge (tests/common.py)
    from common import mlp_trajectory_on_cpu, assert_as_close_to_float64_as_the_cpu_backend
    _, ref64 = mlp_trajectory_on_cpu({n: g["w0/" + n] for n in O.PARAM_ORDER}, g["x"], onehot, steps,
                                     lambda params: light.optim.AdaBelief(params, lr=1e-3), np.float64)
    assert_as_close_to_float64_as_the_cpu_backend({n: p.numpy() for n, p in model.named_parameters()}, {n: g["wf/" + n] for n in O.PARAM_ORDER},
                                                  ref64, what="flat buckets")
    assert opt.t == steps * 4


@pytest.mark.parametrize("name", ["n8_c10_i64", "n5_c3_i32", "n33_c130_i16", "n1_c1000_i64"])
def test_fused_cross_entropy_matches_reference_fixture(hip, name):
    g = load_golden("cross_entropy.npz")
    y = hip.from_numpy(g[name + "/logits"].copy())
    labels = hip.from_numpy(g[name + "/labels"], requires_grad=False)
    loss = light.loss.cross_entropy(y, labels)
    (loss * hip.from_numpy(g[name + "/w"], requires_grad=False)).backward(allow_fill=True)
    assert loss.shape == ()
    np.testing.assert_allclose(loss.numpy(), g[name + "/loss"], rtol=1e-5)
    np.testing.assert_allclose(y.grad.numpy(), g[name + "/grad"], rtol=1e-5, atol=1e-7)


def test_fused_cross_entropy_full_size_and_views(hip):
    """MNIST-sized logits, a transposed (non-dense) logits view, negative labels and the oracle at full size"""
    rng = np.random.RandomState(5)
    logits = rng.uniform(-6, 6, (1024, 10)).astype(np.float32)
    labels = rng.randint(0, 10, 1024).astype(np.int64)
    want_loss, want_grad = O.cross_entropy(logits, labels)
    y = hip.from_numpy(logits)
    loss = light.loss.cross_entropy(y, hip.from_numpy(labels, requires_grad=False))
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=1e-5)
    np.testing.assert_allclose(y.grad.numpy(), want_grad, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(y.grad.numpy().sum(-1), 0, atol=1e-6)      # each row of softmax - onehot sums to zero
    yt = hip.from_numpy(np.ascontiguousarray(logits.T))
    loss_t = light.loss.cross_entropy(yt.transpose(1, 0), hip.from_numpy(labels - 10, requires_grad=False))
    loss_t.backward()
    np.testing.assert_allclose(loss_t.item(), want_loss, rtol=1e-5)
    np.testing.assert_allclose(yt.grad.numpy(), want_grad.T, rtol=1e-5, atol=1e-8)
    with pytest.raises(AssertionError):
        light.loss.cross_entropy(y, hip.from_numpy(labels.astype(np.float32), requires_grad=False))


def test_lazy_zero_grad_equals_fill_then_accumulate(hip):
    """flat-bucket zero_grad only MARKS the gradients as zero: the first backward kernel overwrites, later ones add
    (a layer used twice), a parameter no gradient reaches reads as zeros and stays put.  Same trajectory as the
    per-parameter optimizer with its eager fill."""
    from lightgrad_amd.dist import DataParallel, SingleProcess

    class Net(light.nn.Module):
        def __init__(self):
            light.nn.Module.__init__(self)
            self.shared = light.nn.Linear(12, 12)
            self.head = light.nn.Linear(12, 5)
            self.unused = light.nn.Linear(3, 3)

        def forward(self, x):
            return self.head(self.shared(self.shared(x).relu()).relu())

    rng = np.random.RandomState(2)
    x_np, t_np = rng.uniform(-1, 1, (16, 12)).astype(np.float32), rng.uniform(0, 1, (16, 5)).astype(np.float32)
    results = []
    for flat in (False, True):
        np.random.seed(11)
        model = Net().map_parameters(lambda p: p.hip())
        x, t = hip.from_numpy(x_np, requires_grad=False), hip.from_numpy(t_np, requires_grad=False)
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-2, fused=True, device_step=flat)
        if flat:
            DataParallel(model.parameters(), SingleProcess(), flatten=True).attach(opt)
        for step in range(4):
            loss = light.loss.mse(model(x), t)
            opt.zero_grad()
            if flat:
                assert all(p._grad_zero_pending for p in model.parameters())
            loss.backward()
            if flat:
                assert not model.shared.weight._grad_zero_pending and model.unused.weight._grad_zero_pending
                if step == 0:
                    assert not model.unused.weight.grad.numpy().any()          # reading it writes the zeros
            opt.step()
            assert not any(p._grad_zero_pending for p in model.parameters())
        results.append({n: p.numpy() for n, p in model.named_parameters()})
        results[-1]["grad"] = model.shared.weight.grad.numpy()
    for name in results[0]:
        np.testing.assert_allclose(results[1][name], results[0][name], rtol=1e-5, atol=1e-7, err_msg=name)


def test_cross_entropy_vocabulary_sized_rows(hip):
    """the one-workgroup-per-row kernel (cols >= 4096): BERT's (1024, 30522) logits, an odd width with -inf entries
    (masked classes) and all label dtypes, against the oracle"""
    rng = np.random.RandomState(6)
    for (n, c), ldt in [((1024, 30522), np.int64), ((7, 4097), np.int32), ((3, 5000), np.int16)]:
        logits = rng.uniform(-8, 8, (n, c)).astype(np.float32)
        if c == 4097:
            logits[:, 100:200] = -np.inf
            logits[2, :] = -3.0                       # a constant row: uniform distribution
        labels = rng.randint(0, 99, n).astype(ldt) if c == 4097 else rng.randint(0, c, n).astype(ldt)
        with np.errstate(all="ignore"):
            want_loss, want_grad = O.cross_entropy(logits, labels.astype(np.int64))
        y = hip.from_numpy(logits)
        loss = light.loss.cross_entropy(y, hip.from_numpy(labels, requires_grad=False))
        loss.backward()
        np.testing.assert_allclose(loss.item(), want_loss, rtol=2e-5)
        np.testing.assert_allclose(y.grad.numpy(), want_grad, rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(y.grad.numpy().astype(np.float64).sum(-1), 0, atol=1e-6)


def test_gradient_of_a_reshaped_leaf_lands_in_the_leaf(hip):
    """`x.reshape(-1, d)` of a leaf that already owns a gradient (a static input re-used across steps): the first
    Linear adds dx straight into x.grad through the view; same values as reshape.backward + `x.grad += ...`, also when
    the view has a second consumer and after a zero_grad"""
    rng = np.random.RandomState(14)
    xn, tn = rng.uniform(-1, 1, (6, 2, 5)).astype(np.float32), rng.uniform(0, 1, (12, 3)).astype(np.float32)
    grads = {}
    for cls in (CpuTensor, hip):
        np.random.seed(5)
        lin = light.nn.Linear(5, 3)
        if cls is hip:
            lin.map_parameters(lambda p: p.hip())
        x, t = cls.from_numpy(xn), cls.from_numpy(tn, requires_grad=False)
        out = []
        for step in range(3):
            flat = x.reshape(-1, 5)
            loss = light.loss.mse(lin(flat), t) + (flat * flat).sum() * 0.01        # the view has two consumers
            loss.backward()
            out.append(x.grad.numpy().copy())
            if step == 1:
                x.zero_grad()
        grads[cls] = out
    for got, ref in zip(grads[hip], grads[CpuTensor]):
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert np.abs(grads[hip][1]).sum() > 1.5 * np.abs(grads[hip][0]).sum()          # second pass accumulated


def test_lazy_relu_is_folded_into_linear_and_invisible_elsewhere(hip):
    """relu of a dense tensor is lazy: Linear reads the pre-activation (forward operand and weight-gradient operand)
    and the relu kernel never runs; every other use makes it real.  Values, ALL gradients (the relu output's own
    included) and NaN / zero handling equal the CPU backend's."""
    rng = np.random.RandomState(17)
    xn = rng.uniform(-1, 1, (33, 20)).astype(np.float32)
    xn[0, 0], xn[1, 1] = 0.0, -0.0
    tn = rng.uniform(0, 1, (33, 7)).astype(np.float32)
    res = {}
    for cls in (CpuTensor, hip):
        np.random.seed(9)
        l1, l2 = light.nn.Linear(20, 16), light.nn.Linear(16, 7)
        if cls is hip:
            l1.map_parameters(lambda p: p.hip())
            l2.map_parameters(lambda p: p.hip())
        x = cls.from_numpy(xn)
        pre = l1(x)
        h = pre.relu()
        if cls is hip:
            assert h.is_lazy()
        y = l2(h)
        if cls is hip:
            assert h.is_lazy(), "Linear must not have materialised its relu input"
        loss = light.loss.mse(y, cls.from_numpy(tn, requires_grad=False))
        loss.backward()
        if cls is hip:
            assert h.is_lazy()
        res[cls] = [y.numpy(), h.grad.numpy(), pre.grad.numpy(), x.grad.numpy()] + [p.grad.numpy() for p in list(l1.parameters()) + list(l2.parameters())]
        res[cls].append(h.numpy())                                  # looking at it makes it real
        if cls is hip:
            assert not h.is_lazy()
    for got, ref in zip(res[hip], res[CpuTensor]):
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    # direct uses of a lazy relu: elementwise, reduction, view, numpy, second Linear after materialisation, NaN
    a = rng.uniform(-1, 1, (9, 12)).astype(np.float32)
    a[3, 3] = np.nan
    ta = hip.from_numpy(a)
    with np.errstate(all="ignore"):
        want = np.maximum(a, 0)
        np.testing.assert_array_equal(ta.relu().numpy(), want)
        np.testing.assert_array_equal((ta.relu() + 1.0).numpy(), want + 1.0)
        np.testing.assert_array_equal(ta.relu().transpose(1, 0).contiguous().numpy(), want.T)
        np.testing.assert_array_equal(ta.relu().reshape(3, 36).numpy(), want.reshape(3, 36))
        w = rng.uniform(-1, 1, (5, 12)).astype(np.float32)
        got = ta.relu().linear(hip.from_numpy(w)).numpy()
        ref = want.astype(np.float64) @ w.T
        np.testing.assert_allclose(got[np.arange(9) != 3], ref[np.arange(9) != 3], rtol=1e-5, atol=1e-5)
        assert np.isnan(got[3]).all()
    r = hip.from_numpy(a[:2]).relu()
    np.testing.assert_allclose(r.sum().item(), np.maximum(a[:2], 0).sum(), rtol=1e-5)


def test_lazy_relu_is_a_snapshot_of_its_source(hip):
    """the reference evaluates relu at once (cpu/ops.py:226); the lazy form must not let a later in-place change of
    its source show: every writer into existing storage computes the waiting relu first"""
    rng = np.random.RandomState(23)
    xn = rng.uniform(-1, 1, (17, 12)).astype(np.float32)
    for cls in (CpuTensor, hip):
        np.random.seed(4)
        l1 = light.nn.Linear(12, 8)
        if cls is hip:
            l1.map_parameters(lambda p: p.hip())
        pre = l1(cls.from_numpy(xn))
        before = pre.numpy().copy()
        h = pre.relu()
        with light.no_grad():
            pre += 1.0
        np.testing.assert_array_equal(h.numpy(), np.maximum(before, 0))
        np.testing.assert_allclose(pre.numpy(), before + 1.0, rtol=1e-6)
    # every kind of writer: in-place operators, fill, setitem, upload_, optimizer update of a parameter
    base = rng.uniform(-1, 1, (6, 10)).astype(np.float32)
    want = np.maximum(base, 0)
    writers = {
        "iadd": lambda w: w.__iadd__(1.0), "imul": lambda w: w.__imul__(-2.0), "fill": lambda w: w.fill(-3.0),
        "setitem": lambda w: w.__setitem__((slice(0, 2), slice(None)), 5.0),
        "setitem_view": lambda w: w.transpose(1, 0).__setitem__((0, slice(None)), -1.0),
        "upload": lambda w: w.upload_(np.full((6, 10), -7.0, np.float32)),
    }
    for name, write in writers.items():
        w = hip.from_numpy(base.copy())
        r = w.relu()
        assert r.is_lazy()
        with light.no_grad():
            write(w)
        assert not r.is_lazy(), name
        np.testing.assert_array_equal(r.numpy(), want, err_msg=name)
    # a parameter: r = w.relu(); optimizer step; r still relu(old w).  Linear folds the lazy relu in forward AND backward
    np.random.seed(5)
    l2 = light.nn.Linear(10, 4)
    l2.map_parameters(lambda p: p.hip())
    w = hip.from_numpy(base.copy())
    r = w.relu()
    loss = light.loss.mse(l2(r), hip.from_numpy(rng.uniform(0, 1, (6, 4)).astype(np.float32), requires_grad=False))
    loss.backward()
    opt = light.optim.AdaBelief([w], lr=0.1, fused=True)
    assert r.is_lazy()
    opt.step()
    np.testing.assert_array_equal(r.numpy(), want)
    assert np.abs(w.numpy() - base).max() > 1e-3


def test_multi_tensor_adam_beyond_64_parameters(hip):
    """the flat-bucket optimizer launch takes 64 parameters per launch; a model with more (tiny-BERT has 40+, anything
    bigger hundreds) goes in groups - every parameter must still see its own step number t = step * P + j + 1 (the
    reference's per-parameter `t`, optim.py:36/:48).  Pinned bit for bit against the one-parameter kernel."""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(12)
    sizes = [int(v) for v in rng.randint(1, 40, 150)]
    offsets = tuple(int(o) for o in np.concatenate([[0], np.cumsum(sizes)]))
    total = offsets[-1]
    p0, g0 = rng.uniform(-1, 1, total).astype(np.float32), rng.uniform(-1, 1, total).astype(np.float32)
    results = []
    slots = len(sizes) * 1                    # every parameter is shorter than 1024: one workgroup (and one step slot) each
    for form in ("flat_own_step_slots", "flat_shared_counter", "per_parameter"):
        p, g = hip.from_numpy(p0.copy(), requires_grad=False), hip.from_numpy(g0.copy(), requires_grad=False)
        m, v = hip.zeros((total,), requires_grad=False), hip.zeros((total,), requires_grad=False)
        counter = hip.from_numpy(np.asarray([3, 0] + [3] * slots, np.int64), requires_grad=False)
        for _ in range(2):
            if form == "flat_own_step_slots":          # the launch advances the step itself (private copy per workgroup)
                L.check(lib.lg_adam_multi_dev_f32(p.ptr, g.ptr, m.ptr, v.ptr, len(sizes), L.i64(offsets), 1e-2, 0.9, 0.999, 1e-8,
                                                  counter.ptr, slots, 0.5, 1))
                continue
            if form == "flat_shared_counter":
                L.check(lib.lg_adam_multi_dev_f32(p.ptr, g.ptr, m.ptr, v.ptr, len(sizes), L.i64(offsets), 1e-2, 0.9, 0.999, 1e-8,
                                                  counter.ptr, 0, 0.5, 1))
            else:
                for j, n in enumerate(sizes):
                    o = offsets[j] * 4
                    L.check(lib.lg_adam_step_dev_f32(p.ptr + o, g.ptr + o, m.ptr + o, v.ptr + o, n, 1e-2, 0.9, 0.999, 1e-8,
                                                     counter.ptr, len(sizes), j + 1, 0.5, 1))
            L.check(lib.lg_counter_add_i64(counter.ptr, 1))
        c = counter.numpy()
        assert c[0] == 5 and (form != "flat_own_step_slots" or np.all(c[2:] == 5)), c[:8]
        results.append((p.numpy(), m.numpy(), v.numpy()))
    for other in results[1:]:
        for a, b in zip(results[0], other):
            np.testing.assert_array_equal(a, b)
    assert np.abs(results[0][0] - p0).max() > 1e-3
    # too few step slots for the launch grid is an argument error, not a scribble
    assert lib.lg_adam_multi_dev_f32(p.ptr, g.ptr, m.ptr, v.ptr, len(sizes), L.i64(offsets), 1e-2, 0.9, 0.999, 1e-8, counter.ptr, 10, 0.5, 1) == -1
    assert b"step slots" in lib.lg_last_error()


def test_bias_free_linear_waits_for_its_bias_row(hip):
    """a large bias-free Linear stays unevaluated; `+ bias` right after it is one GEMM with the bias epilogue (BERT's decoder,
    reference bert.py:227).  Values and all three gradients against the CPU backend's two-node form; the product used a second
    time; the weight changed in place between the product and its first use (the product is a snapshot, cpu semantics)"""
    from lightgrad_amd import nn
    rng = np.random.RandomState(21)
    rows, d_in, d_out = 1024, 128, 1100                      # 1.1 M elements: above the threshold
    x, w, b = (rng.uniform(-1, 1, s).astype(np.float32) for s in [(rows, d_in), (d_out, d_in), (d_out,)])
    g = rng.uniform(-1, 1, (rows, d_out)).astype(np.float32)

    def run(T, second_use, poke):
        lin = nn.Linear(d_in, d_out, bias=False)
        lin.weight = T.from_numpy(w.copy())
        tx, tb = T.from_numpy(x), T.from_numpy(b)
        prod = lin(tx)
        if T is not CpuTensor:
            assert prod.is_lazy()
        if poke:
            wt = lin.weight
            with light.no_grad():
                wt *= 0.5                                     # in place; must not show in `prod`
        y = prod + tb
        if T is not CpuTensor:
            assert prod.is_lazy() != poke                     # evaluated only by the in-place writer
        if second_use:
            y = y + prod * 0.25
        (y * T.from_numpy(g, requires_grad=False)).backward(allow_fill=True)
        return y.numpy(), tx.grad.numpy(), lin.weight.grad.numpy(), tb.grad.numpy()

    for second_use, poke in [(False, False), (True, False), (False, True)]:
        want, got = run(CpuTensor, second_use, poke), run(hip, second_use, poke)
        for name, a, e in zip(["y", "dx", "dW", "db"], got, want):
            scale = np.abs(e).max()
            np.testing.assert_allclose(a, e, rtol=1e-5, atol=2e-6 * scale, err_msg="%s second_use=%s poke=%s" % (name, second_use, poke))
    # a small product is evaluated at once; so is one under no_grad that is read directly
    small = hip.from_numpy(x[:8]).linear(hip.from_numpy(w))
    assert not small.is_lazy()
    with light.no_grad():
        big = hip.from_numpy(x).linear(hip.from_numpy(w))
        np.testing.assert_allclose(big.numpy(), x @ w.T, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose((hip.from_numpy(x).linear(hip.from_numpy(w)) + hip.from_numpy(b)).numpy(), x @ w.T + b, rtol=1e-5, atol=1e-5)


def test_cross_entropy_row_held_in_registers_widths(hip):
    """widths on both sides of every variant of the one-pass kernel (1024 threads x 8 / 16 values, 512 x 60) and past
    its limit (two-pass kernel), bad labels included"""
    rng = np.random.RandomState(8)
    for c in [4096, 8161, 8162, 12000, 16353, 16354, 30522, 30689, 30690, 32769]:
        n = 5
        logits = rng.uniform(-6, 6, (n, c)).astype(np.float32)
        logits[1, c - 7:] = -np.inf
        labels = rng.randint(0, c - 8, n).astype(np.int32)
        labels[3] = -1 - labels[3]                                  # numpy-style negative label
        want_loss, want_grad = O.cross_entropy(logits, labels.astype(np.int64))
        y = hip.from_numpy(logits)
        loss = light.loss.cross_entropy(y, hip.from_numpy(labels, requires_grad=False))
        loss.backward()
        np.testing.assert_allclose(loss.item(), want_loss, rtol=2e-5, err_msg=str(c))
        np.testing.assert_allclose(y.grad.numpy(), want_grad, rtol=2e-5, atol=1e-9, err_msg=str(c))


def test_attention_products_take_the_layout_of_their_consumers(hip):
    """head-split attention (reference examples/bert.py:68-88): `probs @ v` lands in (b, s, h, d) memory order, so the head
    merge `transpose(0, 2, 1, 3).reshape(b, s, h*d)` is a view; the gradients of the head-split views q, k^T, v come out in
    the layout of those views, so their way back through transpose / reshape is a view as well.  Values against the CPU tape."""
    rng = np.random.RandomState(31)
    b, s, h, d = 3, 16, 2, 8
    xs = [rng.uniform(-1, 1, (b, s, h * d)).astype(np.float32) for _ in range(3)]
    gn = rng.uniform(-1, 1, (b, s, h * d)).astype(np.float32)

    def run(T):
        tq, tk, tv = (T.from_numpy(x) for x in xs)
        q = tq.reshape(b, s, h, d).transpose(0, 2, 1, 3)
        k = tk.reshape(b, s, h, d).transpose(0, 2, 3, 1)
        v = tv.reshape(b, s, h, d).transpose(0, 2, 1, 3)
        probs = (q @ k * 0.5).softmax(axis=-1)
        ctx_heads = probs @ v
        merged = ctx_heads.transpose(0, 2, 1, 3)
        if T is not CpuTensor:
            assert merged.is_contiguous() and not ctx_heads.is_contiguous()
        out = merged.reshape(b, s, h * d)
        (out * T.from_numpy(gn, requires_grad=False)).backward(allow_fill=True)
        if T is not CpuTensor:
            for leaf in (tq, tk, tv):
                assert leaf.grad.is_contiguous()
        return [out.numpy()] + [t.grad.numpy() for t in (tq, tk, tv)]

    for name, got, want in zip(["context", "dq", "dk", "dv"], run(hip), run(CpuTensor)):
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=name)
    # the general form: any dense permutation with a unit stride among the last two dims, against numpy
    from lightgrad_amd.autograd.hip import ops as H
    a, c = rng.uniform(-1, 1, (2, 3, 5, 7)).astype(np.float32), rng.uniform(-1, 1, (2, 3, 7, 4)).astype(np.float32)
    want = a @ c
    for perm in [(0, 1, 2, 3), (0, 2, 1, 3), (1, 0, 2, 3), (2, 0, 1, 3), (0, 1, 3, 2), (0, 3, 1, 2), (3, 1, 0, 2)]:
        # memory order `perm` (outermost first) of the (2, 3, 5, 4) result
        shape = want.shape
        strides, run_ = [0] * 4, 1
        for dim in reversed(perm):
            strides[dim] = run_
            run_ *= shape[dim]
        if 1 not in (strides[-1], strides[-2]):
            continue
        out = H._gemm(hip.from_numpy(a), hip.from_numpy(c), out_strides=tuple(strides))
        assert out.strides == tuple(strides) or out.is_contiguous()
        np.testing.assert_allclose(out.numpy(), want, rtol=1e-5, atol=1e-6, err_msg=str(perm))


def test_scaled_softmax_is_the_two_kernel_form_bit_for_bit(hip):
    rng = np.random.RandomState(32)
    for shape, scale in [((4, 2, 16, 16), 8.0 ** -1), ((5, 300), 0.3), ((3, 2500), 1.7)]:
        xn, gn = rng.uniform(-4, 4, shape).astype(np.float32), rng.uniform(-1, 1, shape).astype(np.float32)
        x1, x2 = hip.from_numpy(xn), hip.from_numpy(xn)
        y1 = x1.scaled_softmax(scale)
        y2 = (x2 * scale).softmax(axis=-1)
        (y1 * hip.from_numpy(gn, requires_grad=False)).backward(allow_fill=True)
        (y2 * hip.from_numpy(gn, requires_grad=False)).backward(allow_fill=True)
        np.testing.assert_array_equal(y1.numpy(), y2.numpy())
        np.testing.assert_array_equal(x1.grad.numpy(), x2.grad.numpy())
        want = O.softmax_forward(xn * np.float32(scale), axis=-1)
        np.testing.assert_allclose(y1.numpy(), want, rtol=1e-5, atol=1e-7)


def test_linear_with_residual_and_gradients_accumulated_in_the_input_product(hip):
    """a transformer-shaped block: h feeds three Linear layers and a residual (`dense(z, residual=h)`), so h's gradient has
    four contributions - on HipTensor the residual is added in the forward GEMM's epilogue and the second to fourth
    contributions in the epilogues of the input-gradient GEMMs (no separate add kernels); values and every gradient against
    the CPU backend's plain expressions"""
    from lightgrad_amd import nn
    rng = np.random.RandomState(41)
    b, s, d = 3, 16, 24
    xn, gn = rng.uniform(-1, 1, (b, s, d)).astype(np.float32), rng.uniform(-1, 1, (b, s, d)).astype(np.float32)
    weights = [rng.uniform(-0.3, 0.3, (d, d)).astype(np.float32) for _ in range(5)]
    biases = [rng.uniform(-0.3, 0.3, (d,)).astype(np.float32) for _ in range(5)]

    def run(T):
        layers = []
        for w, bias in zip(weights, biases):
            lin = nn.Linear(d, d)
            lin.weight, lin.bias = T.from_numpy(w.copy()), T.from_numpy(bias.copy())
            layers.append(lin)
        x = T.from_numpy(xn)
        h = layers[0](x)                                             # an intermediate: its gradient is built from 4 parts
        z = layers[1](h) * layers[2](h) + layers[3](h)
        out = layers[4](z, residual=h)
        (out * T.from_numpy(gn, requires_grad=False)).backward(allow_fill=True)
        grads = [x.grad.numpy(), h.grad.numpy()] + [p.grad.numpy() for lin in layers for p in (lin.weight, lin.bias)]
        return [out.numpy()] + grads

    for i, (got, want) in enumerate(zip(run(hip), run(CpuTensor))):
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6 * np.abs(want).max(), err_msg="result %d" % i)
    # the C entry point: (A @ B^T + bias) + addend, bit for bit the three-step form
    from lightgrad_amd.autograd.hip import ops as H
    a, w, bias, r = (rng.uniform(-1, 1, sh).astype(np.float32) for sh in [(70, 33), (50, 33), (50,), (70, 50)])
    fused = H._gemm(hip.from_numpy(a), H._swap_last(hip.from_numpy(w)), bias=hip.from_numpy(bias), addend=hip.from_numpy(r))
    plain = H._gemm(hip.from_numpy(a), H._swap_last(hip.from_numpy(w)), bias=hip.from_numpy(bias)) + hip.from_numpy(r)
    np.testing.assert_array_equal(fused.numpy(), plain.numpy())


def test_tied_table_looked_up_twice_keeps_call_order_in_the_gradient_queues(hip):
    """A weight used as an embedding table (looked up TWICE) and as the output projection, on a tape deep enough for the
    gradient group (>= 48 nodes), after a flat-bucket zero_grad: the projection's dW is queued as an OVERWRITING product, the
    two scatter-adds are queued behind it - and the second one (same table) forces an early flush of its queue.  The products
    must leave first (round 2 launched the first scatter-add ahead of the overwrite and lost it).  Against the CPU backend."""
    from lightgrad_amd import nn, CpuTensor
    from lightgrad_amd.dist import DataParallel, SingleProcess
    rng = np.random.RandomState(3)
    vocab, width, rows = 40, 16, 12
    w0 = (rng.uniform(-1, 1, (vocab, width)) / 4).astype(np.float32)
    ids1, ids2 = rng.permutation(vocab)[:rows].astype(np.int32), rng.permutation(vocab)[:rows].astype(np.int32)      # unique per lookup
    target = rng.uniform(0, 1, (rows, vocab)).astype(np.float32)

    class Tied(nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.out = nn.Linear(width, vocab, bias=False)

        def forward(self, a, b):
            table = self.out.weight
            h = table[a] + table[b]
            for _ in range(26):
                h = h * 1.0 + 0.0                       # 52 more tape nodes
            return self.out(h)

    grads = {}
    for cls in (CpuTensor, hip):
        model = Tied()
        model.load_parameters({"out.weight": w0})
        if cls is hip:
            model.map_parameters(lambda p: p.hip())
            opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True)
            DataParallel(model.parameters(), SingleProcess(), flatten=True).attach(opt)
        else:
            opt = light.optim.AdaBelief(model.parameters(), lr=1e-3)
        a, b, t = (cls.from_numpy(v, requires_grad=False) for v in (ids1, ids2, target))
        for _ in range(2):                              # the second pass starts from zero-PENDING gradients on the device
            loss = light.loss.mse(model(a, b), t)
            opt.zero_grad()
            loss.backward()
        grads[cls] = model.out.weight.grad.numpy().copy()
    scale = np.abs(grads[CpuTensor]).max()
    np.testing.assert_allclose(grads[hip], grads[CpuTensor], rtol=1e-5, atol=1e-6 * scale)
    assert np.abs(grads[CpuTensor][ids1]).max() > 0


def _mlp_pair(hip, seed, d_in=20, d_hid=32, d_out=5, batch=48):
    np.random.seed(seed)
    cpu = MLP(d_in, d_hid, d_out)
    w0 = [(n, p.numpy().copy()) for n, p in cpu.named_parameters()]
    dev = MLP(d_in, d_hid, d_out)
    dev.load_parameters(w0)
    dev.map_parameters(lambda p: p.hip())
    rng = np.random.RandomState(seed)
    x = rng.uniform(-1, 1, (batch, d_in)).astype(np.float32)
    t = rng.uniform(0, 1, (batch, d_out)).astype(np.float32)
    return cpu, dev, x, t


def _grads(model):
    return {n: p.grad.numpy().copy() for n, p in model.named_parameters()}


@pytest.mark.parametrize("scenario", ["plain", "seed_not_one", "weight_written_between", "input_written_between", "forward_twice",
                                      "no_zero_grad", "backward_inside_no_bucket", "loss_read_first", "x_without_grad"])
def test_head_gradients_written_ahead_and_riding_weight_gradient(hip, scenario):
    """`Linear -> relu -> Linear(<= 16) -> mse`: the forward launch writes the head's input gradients ahead, the head's weight
    gradient rides in the hidden layer's backward launch (ops._head_backward_riding).  Whatever happens between forward and
    backward, gradients and loss are those of the CPU backend"""
    from lightgrad_amd.autograd.hip import ops as H
    from lightgrad_amd.autograd.hip.tensor import HeldPair
    cpu, dev, x, t = _mlp_pair(hip, 31)

    def run(model, T):
        tx = T.from_numpy(x.copy(), requires_grad=scenario != "x_without_grad")
        tt = T.from_numpy(t, requires_grad=False)
        params = list(model.parameters())
        if scenario != "no_zero_grad":
            for p in params:
                p.zero_grad()
        if scenario == "forward_twice":
            light.loss.mse(model(tx), tt)                                 # a forward pass nobody differentiates
        loss = light.loss.mse(model(tx), tt)
        if scenario == "weight_written_between":
            with light.no_grad():                                         # the gradient is taken at the NEW weight: backward reads w
                model.l2.weight[...] = T.from_numpy(model.l2.weight.numpy() * 0.5, requires_grad=False)
        if scenario == "input_written_between":
            with light.no_grad():                                         # saved by Linear 1's node: its weight gradient sees zeros
                tx[...] = T.from_numpy(np.zeros_like(x), requires_grad=False)
        if scenario == "loss_read_first":
            float(loss.item())
        if scenario == "seed_not_one":
            (loss * 3.0).backward()
        else:
            loss.backward()
        if scenario == "no_zero_grad":
            loss2 = light.loss.mse(model(tx), tt)                         # a second pass ADDS to the gradients of the first
            loss2.backward()
        g = _grads(model)
        if tx.requires_grad:
            g["x"] = tx.grad.numpy().copy()
        return float(loss.item()), g

    from common import float64_tape, rel_frobenius
    with float64_tape():                                                 # the yardstick: the same tape in float64
        cpu.map_parameters(lambda p: CpuTensor.from_numpy(p.numpy().astype(np.float64)))
        ref_loss, ref = run(cpu, CpuTensor)
    assert not HeldPair.held
    got_loss, got = run(dev, hip)
    assert not HeldPair.held
    np.testing.assert_allclose(got_loss, ref_loss, rtol=1e-5)
    assert sorted(got) == sorted(ref)
    for n in ref:
        assert got[n].shape == ref[n].shape
        assert rel_frobenius(got[n], ref[n]) <= 1e-5, (scenario, n, rel_frobenius(got[n], ref[n]))


def test_riding_weight_gradient_is_one_launch_less_and_the_same_numbers(hip):
    """with and without the two peepholes (LIGHTGRAD_HEAD_RIDE / LIGHTGRAD_HEAD_GRAD_AHEAD): same loss bits, gradients within
    rounding (the head's dW comes from MFMA tiles instead of head_bwd's slabs), and a captured step has 4 kernels instead of 5"""
    from lightgrad_amd.autograd.hip import ops as H
    from lightgrad_amd.autograd.hip.graph import HipGraph
    results = {}
    for flags in ((True, True), (True, False), (False, True), (False, False)):
        saved = H._HEAD_RIDE, H._HEAD_GRAD_AHEAD
        H._HEAD_RIDE, H._HEAD_GRAD_AHEAD = flags
        try:
            _, dev, x, t = _mlp_pair(hip, 32, d_in=784, d_hid=512, d_out=10, batch=1024)
            opt = light.optim.AdaBelief(dev.parameters(), lr=1e-3, fused=True, device_step=True)
            tx, tt = hip.from_numpy(x), hip.from_numpy(t, requires_grad=False)
            box = {}

            def step():
                opt.zero_grad()
                loss = light.loss.mse(dev(tx), tt)
                loss.backward()
                box["loss"] = loss
                opt.step()
            step()                                                         # eager once (allocations)
            g = HipGraph()
            with g.capture():
                step()
            g.replay()
            opt.on_graph_replay()
            results[flags] = dict(kernels=g.kernel_count(), loss=float(box["loss"].item()),
                                  w=[p.numpy().copy() for p in dev.parameters()])
            g.destroy()
        finally:
            H._HEAD_RIDE, H._HEAD_GRAD_AHEAD = saved
    base = results[(False, False)]
    for flags, r in results.items():
        assert r["loss"] == base["loss"], flags
        for a, b in zip(r["w"], base["w"]):
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6, err_msg=str(flags))
    assert results[(True, True)]["kernels"] == base["kernels"] - 1, {k: v["kernels"] for k, v in results.items()}
    assert results[(True, False)]["kernels"] == base["kernels"], {k: v["kernels"] for k, v in results.items()}


def test_three_layer_mlp_backward_is_three_launches(hip):
    """Linear -> relu -> Linear -> relu -> Linear(10) -> mse: the hidden layer behind a lazy relu launches dW (+ db) and dx together,
    and the head's weight gradient rides with them; gradients against a float64 tape"""
    from lightgrad_amd.autograd.hip.graph import HipGraph
    from common import float64_tape, rel_frobenius
    dims, batch = (96, 128, 64, 10), 256

    class Net(light.nn.Module):
        def __init__(self):
            light.nn.Module.__init__(self)
            self.l1, self.l2, self.l3 = (light.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

        def forward(self, x):
            return self.l3(self.l2(self.l1(x).relu()).relu())
    np.random.seed(3)
    cpu = Net()
    w0 = [(n, p.numpy().copy()) for n, p in cpu.named_parameters()]
    dev = Net()
    dev.load_parameters(w0)
    dev.map_parameters(lambda p: p.hip())
    rng = np.random.RandomState(3)
    x = rng.uniform(-1, 1, (batch, dims[0])).astype(np.float32)
    t = rng.uniform(0, 1, (batch, dims[-1])).astype(np.float32)

    def run(model, T):
        tx, tt = T.from_numpy(x.copy()), T.from_numpy(t, requires_grad=False)
        for p in model.parameters():
            p.zero_grad()
        loss = light.loss.mse(model(tx), tt)
        loss.backward()
        g = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
        g["x"] = tx.grad.numpy().copy()
        return float(loss.item()), g
    with float64_tape():
        cpu.map_parameters(lambda p: CpuTensor.from_numpy(p.numpy().astype(np.float64)))
        ref_loss, ref = run(cpu, CpuTensor)
    got_loss, got = run(dev, hip)
    np.testing.assert_allclose(got_loss, ref_loss, rtol=1e-5)
    for n in ref:
        assert rel_frobenius(got[n], ref[n]) <= 1e-5, (n, rel_frobenius(got[n], ref[n]))
    # launches of one forward + backward pass: two products and the head's rows forward; backward: the head's weight gradient + the
    # second layer's two products (+ the loss) in one, relu.backward of the first hidden layer, the first layer's two products in one
    tx, tt = hip.from_numpy(x), hip.from_numpy(t, requires_grad=False)
    g = HipGraph()
    with g.capture():
        for p in dev.parameters():
            p.zero_grad()
        light.loss.mse(dev(tx), tt).backward()
    assert g.kernel_count() == 6, g.kernel_count()
    g.destroy()
