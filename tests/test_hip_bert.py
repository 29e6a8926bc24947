"""tiny-BERT and its row-wise fused ops on the HIP path: forward against the reference fixture, forward and
backward against this repo's CPU backend (which test_bert_cpu.py pins to the reference / to numerical
derivatives), fused softmax / LayerNorm / gelu / embedding against their composite definitions."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from common import check_gradients, float64_tape, rel_frobenius
from conftest import load_golden
from test_bert_cpu import bert, build_tiny, small_model

pytestmark = pytest.mark.gpu


def test_forward_matches_reference_fixture(hip):
    g = load_golden("bert_tiny_forward.npz")
    model = build_tiny().map_parameters(lambda p: p.hip())
    with light.no_grad():
        logits = model(hip.from_numpy(g["ids"], requires_grad=False)).numpy()
        masked = model(hip.from_numpy(g["ids"][:1], requires_grad=False),
                       attention_mask=hip.from_numpy(g["mask"], requires_grad=False)).numpy()
    np.testing.assert_allclose(logits[:, :, ::509], g["logits_sample"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(masked[:, ::8, ::509], g["logits_masked_sample"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(np.abs(logits).astype(np.float64).sum(), g["logits_digest"][1], rtol=1e-5)
    assert (logits.argmax(-1) == g["argmax"]).mean() > 0.995


def test_forward_backward_matches_cpu_backend(hip, monkeypatch):
    """full tiny config, batch 2: every parameter gradient of a scalar objective, measured against a FLOAT64 run of the same
    tape on the same parameter values (CpuTensor.default_dtype = float64) - the reference has no working BERT backward
    (SURVEY.md 3.4), so the yardstick is exact arithmetic, not another fp32 result.  Every HIP gradient must be within the
    north-star 1e-5 (relative Frobenius).  The query / key gradients are the hard ones: ~1e-7 in norm, the outcome of a
    contraction that cancels all of d(scores) but its row sums (condition ~150); round 1 accepted 5e-3 for them "because of
    cancellation" - the float64 yardstick showed the fp32 CPU backend at 1e-5 and HIP at 3e-4, i.e. a weakness of the
    fused softmax backward (an fp32 shift common to a row), fixed there (csrc/rowwise.hip) - now 5e-6."""
    g = load_golden("bert_tiny_forward.npz")
    cpu_model = build_tiny()
    hip_model = build_tiny().map_parameters(lambda p: p.hip())
    rng = np.random.RandomState(0)
    w = rng.uniform(-1, 1, (2, 128, 30522)).astype(np.float32)
    fwd = {}
    for model, T in ((cpu_model, CpuTensor), (hip_model, hip)):
        logits = model(T.from_numpy(g["ids"], requires_grad=False))
        fwd[T] = logits.numpy()
        (logits * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    values = {n: p.numpy().astype(np.float64) for n, p in cpu_model.named_parameters()}
    monkeypatch.setattr(CpuTensor, "default_dtype", np.float64)
    ref_model = build_tiny()
    ref_model.load_parameters(values)
    assert all(p.dtype == np.float64 for p in ref_model.parameters())
    logits = ref_model(CpuTensor.from_numpy(g["ids"], requires_grad=False))
    assert logits.dtype == np.float64
    # the FORWARD against the same yardstick: the whole (2, 128, 30522) logits, relative Frobenius (the fixture recorded from the
    # reference is float32 itself; test_forward_matches_reference_fixture compares samples of it element by element)
    e_hip, e_cpu = rel_frobenius(fwd[hip], logits.numpy()), rel_frobenius(fwd[CpuTensor], logits.numpy())
    assert e_hip <= 1e-5, "logits: HIP vs float64 %.2e (float32 CPU backend vs float64 %.2e)" % (e_hip, e_cpu)
    (logits * CpuTensor.from_numpy(w.astype(np.float64), requires_grad=False)).backward(allow_fill=True)
    monkeypatch.undo()
    report = []
    for (n, p), (_, q), (_, r) in zip(cpu_model.named_parameters(), hip_model.named_parameters(), ref_model.named_parameters()):
        assert r.grad.dtype == np.float64
        ref, cpu, got = r.grad.numpy(), p.grad.numpy().astype(np.float64), q.grad.numpy().astype(np.float64)
        if ".key.bias" in n:
            # mathematically zero (softmax is invariant to a per-query constant): all three are rounding noise
            assert np.abs(got).max() < 1e-6 and np.abs(cpu).max() < 1e-6 and np.abs(ref).max() < 1e-12, (n, np.abs(got).max(), np.abs(ref).max())
            continue
        scale = np.linalg.norm(ref) + 1e-300
        e_cpu, e_hip = np.linalg.norm(cpu - ref) / scale, np.linalg.norm(got - ref) / scale
        report.append((n, scale, e_cpu, e_hip))
        assert e_hip <= 1e-5, (n, "HIP vs float64: %.2e (fp32 CPU backend vs float64: %.2e)" % (e_hip, e_cpu))
        assert e_cpu <= 5e-5, (n, "fp32 CPU backend vs float64: %.2e" % e_cpu)
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "bert_grad_errors.txt"), "w") as f:
            f.write("parameter | ||grad|| (float64) | fp32 CPU backend vs float64 | HIP vs float64 (relative Frobenius)\n")
            for n, sc, ec, eh in report:
                f.write("%-70s %.3e  %.2e  %.2e\n" % (n, sc, ec, eh))


def test_small_model_gradcheck(hip):
    model = small_model().map_parameters(lambda p: p.hip())
    ids = hip.from_numpy(np.array([[1, 4, 4, 7]], dtype=np.int32), requires_grad=False)
    dense = model.bert.encoder.layer[0].output.dense

    def f(w):
        dense.weight = w
        return model(ids)[0, :, ::3]
    from lightgrad_amd.autograd.utils.gradcheck import assert_gradcheck
    assert_gradcheck(f, hip.from_numpy(dense.weight.numpy()), eps=1e-2, atol=3e-3, rtol=3e-2)


def composite_softmax(t, axis=-1):
    e = (t - t.max(axis=axis, keepdims=True)).exp()
    return e / e.sum(axis=axis, keepdims=True)


@pytest.mark.parametrize("shape,axis", [((7, 128), -1), ((2, 2, 128, 128), -1), ((5, 33), -1), ((6, 10, 12), 1), ((3, 700), -1), ((2, 2500), -1)])
def test_fused_softmax(hip, shape, axis):
    rng = np.random.RandomState(1)
    x, w = rng.uniform(-3, 3, shape).astype(np.float32), rng.uniform(-1, 1, shape).astype(np.float32)
    tx, ux = hip.from_numpy(x), hip.from_numpy(x)
    y = tx.softmax(axis=axis)
    (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    y2 = composite_softmax(ux, axis)
    (y2 * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(y.numpy(), y2.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(y.numpy().sum(axis=axis), 1.0, rtol=1e-5)
    cy = CpuTensor.from_numpy(x).softmax(axis=axis).numpy()
    np.testing.assert_allclose(y.numpy(), cy, rtol=1e-5, atol=1e-7)
    # the gradient against a float64 run of the composite tape (north star: 1e-5 relative); the fused kernel must be at least
    # as close to exact arithmetic as the composite's own float32 kernels
    with float64_tape():
        rx = CpuTensor.from_numpy(x.astype(np.float64))
        (composite_softmax(rx, axis) * CpuTensor.from_numpy(w.astype(np.float64), requires_grad=False)).backward(allow_fill=True)
    e_fused, e_composite = rel_frobenius(tx.grad.numpy(), rx.grad.numpy()), rel_frobenius(ux.grad.numpy(), rx.grad.numpy())
    assert e_fused <= 1e-5, (e_fused, e_composite)
    assert e_fused <= 2 * e_composite + 1e-7, (e_fused, e_composite)


def composite_attention(q, k, v, heads, scale):
    """the model's own lines (examples/bert.py): head split by stride permutation, scores, scaled softmax, context"""
    b, s, width = q.shape
    d = width // heads
    q4 = q.reshape(b, s, heads, d).transpose(0, 2, 1, 3)
    k4 = k.reshape(b, s, heads, d).transpose(0, 2, 3, 1)
    v4 = v.reshape(b, s, heads, d).transpose(0, 2, 1, 3)
    probs = composite_softmax((q4 @ k4) * scale)
    return (probs @ v4).transpose(0, 2, 1, 3).reshape(b, s, width), probs


@pytest.mark.parametrize("b,s,heads,d", [(8, 128, 2, 64), (2, 32, 1, 64), (3, 96, 4, 32), (1, 64, 3, 32), (2, 64, 2, 64), (1, 128, 1, 32)])
def test_fused_attention(hip, b, s, heads, d):
    """one launch each way (csrc/attention.hip) against the composite tape: context, probabilities and the three input
    gradients within 1e-5 (relative Frobenius) of a float64 run of the composite, and no further from it than twice the
    composite's own float32 kernels; probabilities are rows of a softmax"""
    rng = np.random.RandomState(7)
    width = heads * d
    scale = float(np.sqrt(d)) ** -1
    q, k, v = (rng.uniform(-1.5, 1.5, (b, s, width)).astype(np.float32) for _ in range(3))
    w = rng.uniform(-1, 1, (b, s, width)).astype(np.float32)
    fused = [hip.from_numpy(x) for x in (q, k, v)]
    plain = [hip.from_numpy(x) for x in (q, k, v)]
    assert fused[0].attention_supported(heads)
    out = fused[0].attention(fused[1], fused[2], heads=heads, scale=scale)
    (out * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    out2, probs2 = composite_attention(*plain, heads, scale)
    (out2 * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    with float64_tape():
        ref = [CpuTensor.from_numpy(x.astype(np.float64)) for x in (q, k, v)]
        out_r, probs_r = composite_attention(*ref, heads, scale)
        (out_r * CpuTensor.from_numpy(w.astype(np.float64), requires_grad=False)).backward(allow_fill=True)
    probs = out.attention_probs
    assert probs.shape == (b, heads, s, s) and not probs.requires_grad
    np.testing.assert_allclose(probs.numpy().sum(axis=-1), 1.0, rtol=1e-5)
    for name, got, comp, want in [("context", out, out2, out_r), ("probs", probs, probs2, probs_r)] + \
            [("d" + n, f.grad, c.grad, r.grad) for n, f, c, r in zip("qkv", fused, plain, ref)]:
        e_fused, e_comp = rel_frobenius(got.numpy(), want.numpy()), rel_frobenius(comp.numpy(), want.numpy())
        assert e_fused <= 1e-5, (name, e_fused, e_comp)
        assert e_fused <= 2 * e_comp + 2e-7, (name, e_fused, e_comp)


def test_fused_attention_takes_projection_outputs_as_they_lie(hip):
    """q, k, v as column blocks of ONE (b, s, 3 * width) buffer (row pitch 3 * width): addressed in place, same values"""
    rng = np.random.RandomState(8)
    b, s, heads, d = 2, 64, 2, 32
    width = heads * d
    packed = rng.uniform(-1, 1, (b, s, 3 * width)).astype(np.float32)
    w = rng.uniform(-1, 1, (b, s, width)).astype(np.float32)
    t = hip.from_numpy(packed, requires_grad=False)
    views = [hip(t.data, (b, s, width), t.strides, t.offset + i * width, t.dtype, requires_grad=True) for i in range(3)]
    dense = [hip.from_numpy(np.ascontiguousarray(packed[:, :, i * width:(i + 1) * width])) for i in range(3)]
    res = []
    for qkv in (views, dense):
        out = qkv[0].attention(qkv[1], qkv[2], heads=heads, scale=0.25)
        (out * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        res.append([out.numpy(), out.attention_probs.numpy()] + [x.grad.numpy() for x in qkv])
    for x, y in zip(*res):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(AssertionError, match="unsupported"):
        hip.from_numpy(np.zeros((1, 48, 64), np.float32)).attention(hip.from_numpy(np.zeros((1, 48, 64), np.float32)),
                                                                   hip.from_numpy(np.zeros((1, 48, 64), np.float32)), heads=1)


@pytest.mark.parametrize("shape", [(4, 128), (2, 128, 128), (3, 5, 40), (9, 1000)])
def test_fused_layernorm(hip, shape):
    import lightgrad_amd.nn as nn
    rng = np.random.RandomState(2)
    x, w = rng.uniform(-2, 2, shape).astype(np.float32), rng.uniform(-1, 1, shape).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, shape[-1:]).astype(np.float32), rng.uniform(-1, 1, shape[-1:]).astype(np.float32)
    ln_h, ln_c = nn.LayerNorm(shape[-1]), nn.LayerNorm(shape[-1])
    ln_h.load_parameters({"weight": gamma, "bias": beta})
    ln_c.load_parameters({"weight": gamma, "bias": beta})
    ln_h.map_parameters(lambda p: p.hip())
    tx, cx = hip.from_numpy(x), CpuTensor.from_numpy(x)
    yh, yc = ln_h(tx), ln_c(cx)                                  # fused kernel vs the composite on the CPU backend
    (yh * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    (yc * CpuTensor.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(yh.numpy(), yc.numpy(), rtol=1e-5, atol=2e-6)
    # output and all three gradients against a float64 run of the composite (nn.py:109-124): 1e-5 relative Frobenius each,
    # next to what the float32 CPU backend itself achieves
    with float64_tape():
        ln_r = nn.LayerNorm(shape[-1])
        ln_r.load_parameters({"weight": gamma.astype(np.float64), "bias": beta.astype(np.float64)})
        rx = CpuTensor.from_numpy(x.astype(np.float64))
        yr = ln_r(rx)
        assert yr.dtype == np.float64
        (yr * CpuTensor.from_numpy(w.astype(np.float64), requires_grad=False)).backward(allow_fill=True)
    for what, got, cpu, ref in (("y", yh.numpy(), yc.numpy(), yr.numpy()), ("dx", tx.grad.numpy(), cx.grad.numpy(), rx.grad.numpy()),
                                ("dgamma", ln_h.weight.grad.numpy(), ln_c.weight.grad.numpy(), ln_r.weight.grad.numpy()),
                                ("dbeta", ln_h.bias.grad.numpy(), ln_c.bias.grad.numpy(), ln_r.bias.grad.numpy())):
        e_hip, e_cpu = rel_frobenius(got, ref), rel_frobenius(cpu, ref)
        assert e_hip <= 1e-5, (what, shape, "HIP vs float64 %.2e, float32 CPU backend vs float64 %.2e" % (e_hip, e_cpu))


def test_fused_gelu_and_gradcheck(hip):
    rng = np.random.RandomState(3)
    x = rng.uniform(-4, 4, (64, 130)).astype(np.float32)
    comp = lambda t: 0.5 * t * (1.0 + (t * 0.7978845608 * (1.0 + 0.044715 * t * t)).tanh())   # noqa: E731
    np.testing.assert_allclose(hip.from_numpy(x).gelu().numpy(), comp(CpuTensor.from_numpy(x)).numpy(), rtol=1e-5, atol=1e-6)
    np.random.seed(4)
    check_gradients(hip, "gelu", shapes=[(6, 6)], lowhigh=(-3, 3), tol=2e-3)
    check_gradients(hip, lambda t: t.softmax(axis=-1), shapes=[(4, 7)], tol=2e-3)


def test_embedding_gather_and_scatter_add(hip):
    rng = np.random.RandomState(5)
    table = rng.uniform(-1, 1, (50, 12)).astype(np.float32)
    for dtype in (np.int32, np.int64):
        ids = rng.randint(0, 50, (3, 9)).astype(dtype)
        ids[0, :4] = 7                                             # repeated id: gradient must accumulate
        ids[2, 0] = -1                                             # numpy-style negative index
        w = rng.uniform(-1, 1, (3, 9, 12)).astype(np.float32)
        tt, ct = hip.from_numpy(table), CpuTensor.from_numpy(table)
        out = tt[hip.from_numpy(ids, requires_grad=False)]
        np.testing.assert_array_equal(out.numpy(), table[ids])
        (out * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        (ct[CpuTensor.from_numpy(ids, requires_grad=False)] * CpuTensor.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        np.testing.assert_allclose(tt.grad.numpy(), ct.grad.numpy(), rtol=1e-5, atol=1e-6)
    # the scatter-add for one batch of ids (<= 4096): a table row that receives at most 32 ids is summed in position order like
    # np.add.at - numpy's BITS; hotter rows (a padding id: chunks of 32, combined atomically) and larger batches are not ordered
    from lightgrad_amd.autograd.hip import ops as H
    for n_ids, hot in [(1500, False), (4096, False), (1500, True), (5000, False)]:
        ids = rng.permutation(20000)[:n_ids].astype(np.int32) % 4000          # a few repeats
        ids[10:16] = ids[3]                                                    # one id seven times: still ordered
        if hot:
            ids[100::3] = 17                                                   # hundreds of times: the atomic path
        g = rng.uniform(-1, 1, (n_ids, 12)).astype(np.float32)
        start = rng.uniform(-1, 1, (4000, 12)).astype(np.float32)
        want = start.copy()
        np.add.at(want, ids, g)
        into = hip.from_numpy(start, requires_grad=False)
        H._scatter_add_rows((4000, 12), hip.from_numpy(ids, requires_grad=False), hip.from_numpy(g, requires_grad=False), into=into)
        got = into.numpy()
        if n_ids <= 4096:
            cool = np.ones(4000, bool)
            cool[17] = not hot
            np.testing.assert_array_equal(got[cool], want[cool], err_msg=str((n_ids, hot)))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=str((n_ids, hot)))


def test_attention_c_abi_argument_checks(hip):
    """lg_attention_*: unsupported sizes, pitches below heads * d, misaligned operands and NULL pointers are refused with a text"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    b, s, heads, d = 1, 64, 2, 32
    w = heads * d
    q, k, v, o = (hip.from_numpy(np.zeros((b, s, w), np.float32), requires_grad=False) for _ in range(4))
    p = hip.from_numpy(np.zeros((b, heads, s, s), np.float32), requires_grad=False)
    ok = lambda **kw: lib.lg_attention_fwd_f32(kw.get("q", q.ptr), kw.get("ld", w), s * w, k.ptr, w, s * w, v.ptr, w, s * w, o.ptr, w, s * w,     # noqa: E731
                                               kw.get("p", p.ptr), b, heads, kw.get("s", s), kw.get("d", d), 0.5)
    assert lib.lg_attention_supported(64, 32) == 1 and lib.lg_attention_supported(48, 32) == 0 and lib.lg_attention_supported(256, 64) == 0
    assert lib.lg_attention_supported(128, 48) == 0
    assert ok() == 0
    assert ok(s=48) == -1 and b"unsupported" in lib.lg_last_error()
    assert ok(d=16) == -1
    assert ok(ld=w - 4) == -1 and b"row pitch" in lib.lg_last_error()
    assert ok(q=q.ptr + 4) == -1 and b"aligned" in lib.lg_last_error()
    assert ok(p=None) == -1
    assert lib.lg_attention_bwd_f32(q.ptr, w, s * w, k.ptr, w, s * w, v.ptr, w, s * w, o.ptr, w, s * w, p.ptr, q.ptr, w, s * w, k.ptr, w, s * w,
                                    None, w, s * w, b, heads, s, d, 0.5) == -1
    ids = hip.from_numpy(np.zeros((6,), np.int32), requires_grad=False)
    t = hip.from_numpy(np.zeros((4, 8), np.float32), requires_grad=False)
    out = hip.from_numpy(np.zeros((6, 8), np.float32), requires_grad=False)
    assert lib.lg_gather_sum3_rows_f32(t.ptr, ids.ptr, 6, 4, t.ptr, ids.ptr, 4, 4, t.ptr, ids.ptr, 6, 4, 4, out.ptr, 6, 8) == -1
    assert b"divide" in lib.lg_last_error()
    assert lib.lg_gather_sum3_rows_f32(t.ptr, ids.ptr, 6, 4, t.ptr, ids.ptr, 3, 4, t.ptr, ids.ptr, 6, 4, 2, out.ptr, 6, 8) == -1


@pytest.mark.parametrize("b,s,hidden,heads,d", [(8, 128, 128, 2, 64), (2, 64, 96, 2, 32), (1, 32, 40, 3, 64)])
def test_self_attention_node(hip, b, s, hidden, heads, d):
    """projections + attention as one node (one launch for q / k / v, one for the attention; backward: one attention launch, ONE
    input-gradient product through the three weights): context, probabilities, the input gradient - also on top of a
    contribution the input already holds (a residual branch) - and the six parameter gradients against a float64 run of the
    composite, no further from it than twice the three-Linear + composite-attention form on the same backend"""
    rng = np.random.RandomState(10)
    width = heads * d
    scale = float(np.sqrt(d)) ** -1
    x = rng.uniform(-1, 1, (b, s, hidden)).astype(np.float32)
    params = []
    for _ in range(3):
        params += [rng.uniform(-0.2, 0.2, (width, hidden)).astype(np.float32), rng.uniform(-0.2, 0.2, (width,)).astype(np.float32)]
    w = rng.uniform(-1, 1, (b, s, width)).astype(np.float32)
    r = rng.uniform(-1, 1, (b, s, hidden)).astype(np.float32)

    def run(T, f64, fused):
        cast = (lambda a: a.astype(np.float64)) if f64 else (lambda a: a)
        leaf = T.from_numpy(cast(x))
        tx = leaf * 1.0                                            # an intermediate: its gradient may be built in an epilogue
        ps = [T.from_numpy(cast(p)) for p in params]
        if fused:
            assert tx.self_attention_supported(ps[0], heads)
            out = tx.self_attention(*ps, heads=heads, scale=scale)
            probs = out.attention_probs
        else:
            q, k, v = (tx @ ps[2 * i].transpose(1, 0) + ps[2 * i + 1] for i in range(3))
            out, probs = composite_attention(q, k, v, heads, scale)
        # a residual branch first: the input holds a gradient when the node's backward runs
        ((tx * T.from_numpy(cast(r), requires_grad=False)).sum() + (out * T.from_numpy(cast(w), requires_grad=False)).sum()).backward()
        return [out.numpy(), probs.numpy(), leaf.grad.numpy()] + [p.grad.numpy() for p in ps]
    fused, plain = run(hip, False, True), run(hip, False, False)
    with float64_tape():
        ref = run(CpuTensor, True, False)
    names = ["context", "probs", "dx", "dwq", "dbq", "dwk", "dbk", "dwv", "dbv"]
    for name, f, p_, want in zip(names, fused, plain, ref):
        if name == "dbk":                                          # mathematically zero: rounding noise on every path
            assert np.abs(f).max() < 1e-4 * np.abs(ref[4]).max() + 1e-6
            continue
        e_f, e_p = rel_frobenius(f, want), rel_frobenius(p_, want)
        assert e_f <= 1e-5, (name, e_f, e_p)
        assert e_f <= 2 * e_p + 3e-7, (name, e_f, e_p)


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_embedding_sum_is_the_three_lookups_and_two_adds(hip, dtype):
    """word + position + token-type embeddings in one pass: the bits of the composite line of examples/bert.py, values and
    the three table gradients (position ids shared by the batch: their rows' gradients are summed over it first)"""
    rng = np.random.RandomState(6)
    b, s, width = 3, 10, 12
    tables = [rng.uniform(-1, 1, (n, width)).astype(np.float32) for n in (40, 16, 2)]
    ids = rng.randint(0, 40, (b, s)).astype(dtype)
    ids[1, :3] = 5
    ids[2, 0] = -1
    pos = np.arange(s, dtype=dtype)
    kind = rng.randint(0, 2, (b, s)).astype(dtype)
    w = rng.uniform(-1, 1, (b, s, width)).astype(np.float32)
    fused, plain = [hip.from_numpy(t) for t in tables], [hip.from_numpy(t) for t in tables]
    i0, i1, i2 = (hip.from_numpy(x, requires_grad=False) for x in (ids, pos, kind))
    out = fused[0].embedding_sum(fused[1], fused[2], ids0=i0, ids1=i1, ids2=i2)
    out2 = plain[0][i0] + plain[1][i1] + plain[2][i2]
    np.testing.assert_array_equal(out.numpy(), out2.numpy())
    np.testing.assert_array_equal(out.numpy(), (tables[0][ids] + tables[1][pos]) + tables[2][kind])
    (out * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    (out2 * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    for f, q in zip(fused, plain):
        np.testing.assert_allclose(f.grad.numpy(), q.grad.numpy(), rtol=1e-6, atol=1e-6)
    bad = hip.from_numpy(np.full((b, s), 40, dtype), requires_grad=False)
    out = fused[0].embedding_sum(fused[1], fused[2], ids0=bad, ids1=i1, ids2=i2)
    with pytest.raises(IndexError):
        out.numpy()


def _embedding_sum_grads(T, tables, ids, pos, kind, w, fused):
    ts = [T.from_numpy(t) for t in tables]
    i0, i1, i2 = (T.from_numpy(x, requires_grad=False) for x in (ids, pos, kind))
    out = ts[0].embedding_sum(ts[1], ts[2], ids0=i0, ids1=i1, ids2=i2) if fused else (ts[0][i0] + ts[1][i1]) + ts[2][i2]
    (out * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    return [t.grad.numpy() for t in ts], (ts, (i0, i1, i2))


@pytest.mark.parametrize("batch", [4, 64], ids=["repeats_within_one_ordered_chunk", "batch_64_more_repeats_than_a_chunk"])
def test_embedding_sum_gradients_of_shared_ids_follow_the_tape_at_any_batch(hip, batch):
    """position / token-type ids shared by the batch: with more repeats than the scatter kernel sums in order (32) the
    gradient takes the tape's form - sum over the batch, then one ordered scatter - and stays bit-reproducible; either way it is the
    numpy CPU backend's gradient of the composite line"""
    from lightgrad_amd import CpuTensor
    rng = np.random.RandomState(16)
    s, width = 12, 8
    tables = [rng.uniform(-1, 1, (n, width)).astype(np.float32) for n in (30, 12, 2)]
    ids = rng.randint(0, 30, (batch, s)).astype(np.int32)
    pos = np.arange(s, dtype=np.int32)
    kind = (np.arange(s) >= 5).astype(np.int32)                        # token types: shared by the batch too, only two rows
    w = rng.uniform(-1, 1, (batch, s, width)).astype(np.float32)
    want, _ = _embedding_sum_grads(CpuTensor, tables, ids, pos, kind, w, fused=False)
    got, _ = _embedding_sum_grads(hip, tables, ids, pos, kind, w, fused=True)
    again, _ = _embedding_sum_grads(hip, tables, ids, pos, kind, w, fused=True)
    for g, a, c, name in zip(got, again, want, ("word", "position", "token type")):
        np.testing.assert_array_equal(g, a, err_msg=name)                # run to run: the same bits
        np.testing.assert_allclose(g, c, rtol=1e-5, atol=1e-5 * np.abs(c).max(), err_msg=name)
    if batch > 32:       # the tape's FORM (sum over the batch, then one ordered scatter): one rounding per addend of a 64-term sum
        np.testing.assert_allclose(got[1], want[1], rtol=0, atol=64 * 2.0 ** -24 * np.abs(w).sum(0).max())


def test_shared_ids_refreshed_in_place_are_tiled_again(hip):
    """the tiled copy of a shared id tensor is kept with the tensor's storage - an in-place refresh of the ids (upload_, setitem)
    between two backward passes must not leave the old copy in use"""
    rng = np.random.RandomState(17)
    batch, s, width = 4, 6, 8
    tables = [rng.uniform(-1, 1, (n, width)).astype(np.float32) for n in (20, 10, 3)]
    ids = rng.randint(0, 20, (batch, s)).astype(np.int32)
    w = rng.uniform(-1, 1, (batch, s, width)).astype(np.float32)
    pos_a, pos_b = np.arange(s, dtype=np.int32), np.arange(s, dtype=np.int32)[::-1].copy() + 3
    kind = np.zeros(s, np.int32)
    ts = [hip.from_numpy(t) for t in tables]
    i0, i1, i2 = (hip.from_numpy(x, requires_grad=False) for x in (ids, pos_a, kind))
    tw = hip.from_numpy(w, requires_grad=False)

    def position_gradient():
        for t in ts:
            t.zero_grad()
        (ts[0].embedding_sum(ts[1], ts[2], ids0=i0, ids1=i1, ids2=i2) * tw).backward(allow_fill=True)
        return ts[1].grad.numpy().copy()
    first = position_gradient()
    want_a = np.zeros_like(tables[1])
    np.add.at(want_a, pos_a, w.sum(0))
    np.testing.assert_allclose(first, want_a, rtol=1e-5, atol=1e-6)
    for refresh in ("upload", "setitem", "fill"):
        if refresh == "upload":
            i1.upload_(pos_b)
            now = pos_b
        elif refresh == "setitem":
            i1[...] = hip.from_numpy(pos_a, requires_grad=False)
            now = pos_a
        else:
            i1.fill(7)
            now = np.full(s, 7, np.int32)
        want = np.zeros_like(tables[1])
        np.add.at(want, now, w.sum(0))
        np.testing.assert_allclose(position_gradient(), want, rtol=1e-5, atol=1e-6, err_msg=refresh)


def test_frozen_tensors_refuse_in_place_writes(hip):
    t = hip.from_numpy(np.arange(6, dtype=np.int32), requires_grad=False).freeze()
    for write in (lambda: t.fill(1), lambda: t.upload_(np.zeros(6, np.int32)), lambda: t.__setitem__(slice(0, 2), 7)):
        with pytest.raises(RuntimeError, match="frozen"):
            write()
    np.testing.assert_array_equal(t.numpy(), np.arange(6, dtype=np.int32))


@pytest.mark.parametrize("rows,width,inner", [((8, 128), 128, 512), ((3, 10), 36, 52), ((70,), 64, 64)])
def test_feed_forward_block_against_the_separate_ops(hip, rows, width, inner):
    """dense2(gelu(dense1(x))) + x as one node (gelu and its derivative in GEMM epilogues) against Linear, gelu, Linear, add
    on the same backend: output and all five gradients (the same elementwise expressions; the products may split K differently,
    so to rounding), and against a float64 run of the composite"""
    import lightgrad_amd.nn as nn
    rng = np.random.RandomState(9)
    x = rng.uniform(-1, 1, rows + (width,)).astype(np.float32)
    w = rng.uniform(-1, 1, rows + (width,)).astype(np.float32)
    params = [rng.uniform(-0.3, 0.3, s).astype(np.float32) for s in ((inner, width), (inner,), (width, inner), (width,))]
    res = []
    for fused in (True, False):
        tx = hip.from_numpy(x)
        w1, b1, w2, b2 = (hip.from_numpy(p) for p in params)
        if fused:
            y = tx.feed_forward(w1, b1, w2, b2, tx)
        else:
            up, down = nn.Linear(width, inner), nn.Linear(inner, width)
            up.weight, up.bias, down.weight, down.bias = w1, b1, w2, b2
            y = down(up(tx).gelu(), residual=tx)
        (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        res.append([y.numpy()] + [t.grad.numpy() for t in (tx, w1, b1, w2, b2)])
    for name, a, b in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), *res):
        assert rel_frobenius(a, b) <= 2e-6, (name, rel_frobenius(a, b))
    with float64_tape():
        cx = CpuTensor.from_numpy(x.astype(np.float64))
        cw1, cb1, cw2, cb2 = (CpuTensor.from_numpy(p.astype(np.float64)) for p in params)
        h = cx @ cw1.transpose(1, 0) + cb1
        h = 0.5 * h * (1.0 + (h * 0.7978845608 * (1.0 + 0.044715 * h * h)).tanh())
        y = h @ cw2.transpose(1, 0) + cb2 + cx
        (y * CpuTensor.from_numpy(w.astype(np.float64), requires_grad=False)).backward(allow_fill=True)
    for name, got, want in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), res[0], [y.numpy()] + [t.grad.numpy() for t in (cx, cw1, cb1, cw2, cb2)]):
        assert rel_frobenius(got, want) <= 1e-5, (name, rel_frobenius(got, want))


def test_parameter_gradients_accumulate_in_place(hip):
    """second backward pass into existing gradient buffers (what a training step does after zero_grad): LayerNorm's
    fused dw/db launch and the embedding scatter-add write into the parameters' buffers and must ADD - after an eager
    zero_grad, after two passes without one, and on odd / split-over-workgroups shapes"""
    import lightgrad_amd.nn as nn
    rng = np.random.RandomState(21)
    for rows, cols in [(1024, 128), (7, 5), (300, 257), (64, 1000)]:
        x = rng.uniform(-2, 2, (rows, cols)).astype(np.float32)
        w = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
        grads = {}
        for cls in (CpuTensor, hip):
            np.random.seed(3)
            ln = nn.LayerNorm(cols)
            ln.load_parameters({"weight": rng.uniform(0.5, 1.5, (cols,)).astype(np.float32) * 0 + 1.25, "bias": np.full((cols,), 0.5, np.float32)})
            if cls is hip:
                ln.map_parameters(lambda p: p.hip())
            for _ in range(2):                                   # two passes, no zero_grad in between: grads add up
                (ln(cls.from_numpy(x)) * cls.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
            twice = [ln.weight.grad.numpy().copy(), ln.bias.grad.numpy().copy()]
            for p in ln.parameters():
                p.zero_grad()
            (ln(cls.from_numpy(x)) * cls.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
            grads[cls] = twice + [ln.weight.grad.numpy(), ln.bias.grad.numpy()]
        scale = rows ** 0.5
        for got, ref in zip(grads[hip], grads[CpuTensor]):
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(grads[hip][0], 2 * grads[hip][2], rtol=1e-5, atol=1e-5 * scale)
    table = rng.uniform(-1, 1, (40, 8)).astype(np.float32)
    ids = rng.randint(0, 40, (5, 6)).astype(np.int64)
    w = rng.uniform(-1, 1, (5, 6, 8)).astype(np.float32)
    tt, ct = hip.from_numpy(table), CpuTensor.from_numpy(table)
    for cls, t in ((hip, tt), (CpuTensor, ct)):
        for _ in range(2):
            (t[cls.from_numpy(ids, requires_grad=False)] * cls.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(tt.grad.numpy(), ct.grad.numpy(), rtol=1e-5, atol=1e-6)
    tt.zero_grad()
    (tt[hip.from_numpy(ids, requires_grad=False)] * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(2 * tt.grad.numpy(), ct.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_parameter_gradients_off_the_critical_path(hip):
    """a deep tape queues dW / db and LayerNorm's parameter gradients and launches them together at the end of the pass
    (autograd/hip/tensor.py GradGroup): same gradients as the plain pass - eager, accumulated over two passes, and replayed
    from a hipGraph - and no memory stays parked afterwards"""
    import gc
    from lightgrad_amd.autograd.hip import HipGraph, HipDevice
    from lightgrad_amd.autograd.hip.tensor import GradGroup
    from lightgrad_amd.dist import DataParallel, SingleProcess
    rng = np.random.RandomState(5)
    ids_np = rng.randint(0, 300, (4, 32)).astype(np.int32)
    labels_np = rng.randint(0, 300, (4 * 32,)).astype(np.int64)

    def build():
        np.random.seed(9)
        model = bert.BertForMaskedLM(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                                     vocab_size=300, max_position_embeddings=32, type_vocab_size=2).map_parameters(lambda p: p.hip())
        return model, DataParallel(model.parameters(), SingleProcess(), flatten=True)       # gradients = views of one bucket

    ids, labels = hip.from_numpy(ids_np, requires_grad=False), hip.from_numpy(labels_np, requires_grad=False)

    def one_pass(model, dp, zero=True):
        loss = light.loss.cross_entropy(model(ids).reshape(-1, 300), labels)
        if zero:
            dp.bucket.fill(0)
        loss.backward()
        return loss

    def configure(on):
        GradGroup.enabled = on

    default = GradGroup.enabled
    try:
        results = {}
        for mode in (False, True):
            configure(mode)
            model, dp = build()
            one_pass(model, dp)
            once = dp.bucket.numpy().copy()
            one_pass(model, dp, zero=False)                      # second pass accumulates on top
            results[mode] = (once, dp.bucket.numpy().copy())
        assert np.abs(results[True][0]).max() > 0
        for a, b in zip(results[True], results[False]):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6 * np.abs(b).max())
        # captured: every replay gives the eager gradients again
        configure(True)
        model, dp = build()
        one_pass(model, dp)
        eager = dp.bucket.numpy().copy()
        graph = HipGraph()
        with graph.capture():
            loss = one_pass(model, dp)
        for _ in range(3):
            graph.replay()
            np.testing.assert_allclose(dp.bucket.numpy(), eager, rtol=1e-5, atol=1e-6 * np.abs(eager).max())
        assert np.isfinite(loss.item())
        graph.destroy()
        del loss
        in_use = []
        for _ in range(3):
            one_pass(model, dp)
            HipDevice.synchronize()
            gc.collect()
            in_use.append(HipDevice.pool_stats()["in_use_bytes"])
        assert in_use[2] == in_use[1], in_use                     # nothing stays parked or leaks per pass
    finally:
        GradGroup.enabled = default


def test_bert_training_steps_replayed_from_a_graph_match_the_eager_tape(hip):
    """forward + masked-LM loss + backward (gradient group) + fused AdaBelief on the flat bucket, five steps: run eagerly, and with
    the step captured once and replayed - the same parameters afterwards (the queued weight gradients are launched before the
    optimizer reads them, the optimizer's device step counter advances per replay)"""
    from lightgrad_amd.autograd.hip import HipGraph
    from lightgrad_amd.dist import DataParallel, SingleProcess
    rng = np.random.RandomState(15)
    ids = hip.from_numpy(rng.randint(0, 200, (4, 24)).astype(np.int32), requires_grad=False)
    labels = hip.from_numpy(rng.randint(0, 200, (4 * 24,)).astype(np.int64), requires_grad=False)

    def build():
        np.random.seed(21)
        model = bert.BertForMaskedLM(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                     vocab_size=200, max_position_embeddings=24, type_vocab_size=2).map_parameters(lambda p: p.hip())
        dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-2, fused=True, device_step=True)
        dp.attach(opt)

        def step():
            loss = light.loss.cross_entropy(model(ids).reshape(-1, 200), labels)
            opt.zero_grad()
            loss.backward()
            dp.sync_gradients()
            opt.step()
            return loss
        return model, opt, step

    model_e, _, step_e = build()
    losses_e = [step_e().item() for _ in range(5)]
    model_g, opt_g, step_g = build()
    losses_g = [step_g().item() for _ in range(2)]                 # two eager steps: optimizer state, pool, kernels
    n_params = len(opt_g.parameters)
    graph = HipGraph()
    with graph.capture():
        loss = step_g()
    opt_g.t -= n_params                                            # the capture pass ran the python bookkeeping, not the kernels
    for _ in range(3):
        graph.replay()
        opt_g.on_graph_replay()
        losses_g.append(loss.item())
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-5)
    assert losses_e[-1] < losses_e[0]
    for (n, p), (_, q) in zip(model_e.named_parameters(), model_g.named_parameters()):
        # (an Adam-type update normalises the gradient: where a gradient is rounding noise - the key projections, see the float64
        # test above - a last-bit difference from the atomically summed token-type embedding row moves a weight by a fraction of lr)
        np.testing.assert_allclose(q.numpy(), p.numpy(), rtol=2e-4, atol=5e-5, err_msg=n)


def test_tapes_die_by_reference_counting(hip):
    """no reference cycle through a tape node that keeps its own output (softmax, exp, tanh, pow, max with keepdims): with
    python's cycle collector switched off, eager training steps must not accumulate device memory"""
    import gc
    from lightgrad_amd.autograd.hip import HipDevice
    rng = np.random.RandomState(3)
    model = small_model().map_parameters(lambda p: p.hip())
    ids = hip.from_numpy(np.array([[1, 4, 4, 7], [2, 2, 9, 0]], dtype=np.int32), requires_grad=False)
    x = hip.from_numpy(rng.uniform(0.5, 2, (4, 6)).astype(np.float32))

    def step():
        logits = model(ids)
        loss = (logits * logits).mean() + (x.exp().tanh().sigmoid() ** 2.0).max(axis=1, keepdims=True).sum() + (x ** x).mean()
        for p in model.parameters():
            p.zero_grad()
        x.zero_grad()
        loss.backward()
        return loss.item()

    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        for _ in range(3):
            step()
        HipDevice.synchronize()
        before = HipDevice.pool_stats()["in_use_bytes"]
        for _ in range(20):
            step()
        HipDevice.synchronize()
        after = HipDevice.pool_stats()["in_use_bytes"]
    finally:
        if was_enabled:
            gc.enable()
    assert after == before, (before, after)
