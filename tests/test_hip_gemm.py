"""SGEMM parity: every operand layout (row-/column-major views), ragged and tiny shapes, batches,
accumulate mode through the raw C ABI, and BASELINE's 4096^2 forward+backward through size-independent
properties.  Error metric: relative Frobenius error against float64 numpy <= 1e-5."""
import ctypes
import numpy as np
import lightgrad_amd as light
import pytest
import np_oracle as O

pytestmark = pytest.mark.gpu


def rel_err(got, ref64):
    return np.linalg.norm(got.astype(np.float64) - ref64) / max(np.linalg.norm(ref64), 1e-30)


SHAPES = [(16, 3000, 24), (8, 640, 700), (1, 1, 1), (1, 7, 1), (3, 5, 2), (32, 32, 32), (33, 31, 35), (64, 64, 64), (65, 129, 67), (128, 128, 128),
          (127, 255, 96), (256, 100, 256), (1024, 784, 512), (1024, 512, 10), (1024, 10, 512), (784, 1024, 512), (10, 1024, 512),
          (200, 36, 300), (130, 8, 70), (512, 4, 512), (257, 130, 513)]


@pytest.mark.parametrize("mkn", SHAPES)
def test_all_layouts(hip, mkn):
    M, K, N = mkn
    rng = np.random.RandomState(M * 7 + K * 3 + N)
    a, b = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    ta_t = hip.from_numpy(np.ascontiguousarray(a.T)).transpose(1, 0)       # column-major view of the same values
    tb_t = hip.from_numpy(np.ascontiguousarray(b.T)).transpose(1, 0)
    for x, y, tag in [(ta, tb, "NN"), (ta, tb_t, "NT"), (ta_t, tb, "TN"), (ta_t, tb_t, "TT")]:
        out = (x @ y).numpy()
        assert out.shape == (M, N)
        assert rel_err(out, ref) <= 1e-5, (tag, mkn, rel_err(out, ref))
        np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5 * K ** 0.5, err_msg=tag)


def test_identity_is_exact(hip):
    """A @ I == A bit for bit, in every layout: catches a swapped C/D row<->col map (asymmetric A)"""
    rng = np.random.RandomState(5)
    a = rng.uniform(-1, 1, (192, 160)).astype(np.float32)
    eye = np.eye(160, dtype=np.float32)
    ta, te = hip.from_numpy(a), hip.from_numpy(eye)
    np.testing.assert_array_equal((ta @ te).numpy(), a)
    np.testing.assert_array_equal((ta @ te.transpose(1, 0)).numpy(), a)
    np.testing.assert_array_equal((hip.from_numpy(np.eye(192, dtype=np.float32)) @ ta).numpy(), a)
    at = hip.from_numpy(np.ascontiguousarray(a.T)).transpose(1, 0)
    np.testing.assert_array_equal((at @ te).numpy(), a)


def test_backward_layouts_match_oracle(hip):
    rng = np.random.RandomState(6)
    for (M, K, N) in [(10, 15, 10), (64, 48, 20), (130, 70, 200)]:
        a, b = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
        w = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        ga, gb = O.dot_backward(w.astype(np.float64), a.astype(np.float64), b.astype(np.float64))
        for ta_view in (False, True):
            for tb_view in (False, True):
                ta = hip.from_numpy(np.ascontiguousarray(a.T)).transpose(1, 0) if ta_view else hip.from_numpy(a)
                tb = hip.from_numpy(np.ascontiguousarray(b.T)).transpose(1, 0) if tb_view else hip.from_numpy(b)
                y = ta @ tb
                (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
                assert rel_err(ta.grad.numpy(), ga) <= 1e-5 and rel_err(tb.grad.numpy(), gb) <= 1e-5


def test_linear_weight_gradient_is_dense(hip):
    """x @ W.T(1,0): dW comes back through transpose.backward as a dense (out,in) tensor - no strided accumulate"""
    rng = np.random.RandomState(7)
    x, w = rng.uniform(-1, 1, (32, 48)).astype(np.float32), rng.uniform(-1, 1, (20, 48)).astype(np.float32)
    tx, tw = hip.from_numpy(x), hip.from_numpy(w)
    y = tx @ tw.T(1, 0)
    y.backward(allow_fill=True)
    assert tw.grad.shape == (20, 48) and tw.grad.is_contiguous()
    np.testing.assert_allclose(tw.grad.numpy(), np.ones((32, 20)).T @ x.astype(np.float64), rtol=1e-5, atol=1e-5)


def test_batched(hip):
    rng = np.random.RandomState(8)
    a = rng.uniform(-1, 1, (3, 4, 33, 20)).astype(np.float32)
    b = rng.uniform(-1, 1, (3, 4, 20, 17)).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    assert rel_err((ta @ tb).numpy(), a.astype(np.float64) @ b) <= 1e-5
    b2 = rng.uniform(-1, 1, (20, 17)).astype(np.float32)
    assert rel_err((ta @ hip.from_numpy(b2)).numpy(), a.astype(np.float64) @ b2) <= 1e-5          # one tall GEMM
    b3 = rng.uniform(-1, 1, (4, 20, 17)).astype(np.float32)
    assert rel_err((ta @ hip.from_numpy(b3)).numpy(), a.astype(np.float64) @ b3) <= 1e-5          # broadcast batch
    # attention-style head split: (b, s, h, d) -> transpose(0, 2, 1, 3): batch dims do not collapse
    q = rng.uniform(-1, 1, (2, 16, 4, 8)).astype(np.float32)
    k = rng.uniform(-1, 1, (2, 16, 4, 8)).astype(np.float32)
    tq, tk = hip.from_numpy(q).transpose(0, 2, 1, 3), hip.from_numpy(k).transpose(0, 2, 3, 1)
    ref = q.transpose(0, 2, 1, 3).astype(np.float64) @ k.transpose(0, 2, 3, 1)
    assert rel_err((tq @ tk).numpy(), ref) <= 1e-5
    # vectors
    v = rng.uniform(-1, 1, (20,)).astype(np.float32)
    np.testing.assert_allclose((hip.from_numpy(a[0, 0]) @ hip.from_numpy(v)).numpy(), a[0, 0].astype(np.float64) @ v, rtol=1e-5, atol=1e-5)


def test_c_abi_accumulate_and_errors(hip):
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(9)
    M, K, N = 70, 40, 50
    a, b, c = (rng.uniform(-1, 1, s).astype(np.float32) for s in [(M, K), (K, N), (M, N)])
    ta, tb, tc = hip.from_numpy(a), hip.from_numpy(b), hip.from_numpy(c)
    assert lib.lg_gemm_f32(0, 0, M, N, K, ta.ptr, K, 0, tb.ptr, N, 0, tc.ptr, N, 0, 1, 1) == 0
    np.testing.assert_allclose(tc.numpy(), c + a.astype(np.float64) @ b, rtol=1e-5, atol=1e-5)
    # split-K path (few tiles, long K) with accumulate and a padded ldc
    M2, K2, N2, ldc = 96, 4096, 80, 96
    a2, b2 = rng.uniform(-1, 1, (M2, K2)).astype(np.float32), rng.uniform(-1, 1, (K2, N2)).astype(np.float32)
    c2 = rng.uniform(-1, 1, (M2, ldc)).astype(np.float32)
    ta2, tb2, tc2 = hip.from_numpy(a2), hip.from_numpy(b2), hip.from_numpy(c2)
    assert lib.lg_gemm_f32(0, 0, M2, N2, K2, ta2.ptr, K2, 0, tb2.ptr, N2, 0, tc2.ptr, ldc, 0, 1, 1) == 0
    expect = c2.astype(np.float64)
    expect[:, :N2] += a2.astype(np.float64) @ b2
    assert rel_err(tc2.numpy(), expect) <= 1e-5
    np.testing.assert_array_equal(tc2.numpy()[:, N2:], c2[:, N2:])          # padding columns untouched
    again = hip.from_numpy(c2)
    assert lib.lg_gemm_f32(0, 0, M2, N2, K2, ta2.ptr, K2, 0, tb2.ptr, N2, 0, again.ptr, ldc, 0, 1, 1) == 0
    np.testing.assert_array_equal(again.numpy(), tc2.numpy())                # deterministic slice order
    assert lib.lg_gemm_f32(0, 0, M, N, K, ta.ptr, K - 1, 0, tb.ptr, N, 0, tc.ptr, N, 0, 1, 0) == -1       # lda < K
    assert b"leading dimension" in lib.lg_last_error()
    assert lib.lg_gemm_f32(0, 0, M, N, K, None, K, 0, tb.ptr, N, 0, tc.ptr, N, 0, 1, 0) == -1
    assert lib.lg_ew(12345, 1, L.i64((4,)), tc.ptr, L.i64((1,)), None, None, ta.ptr, L.i64((1,)), None, None, None, None,
                     None, None, 0.0) == -1
    assert b"unknown op" in lib.lg_last_error()
    assert lib.lg_free(ctypes.c_void_p(12345)) == -1


def test_c_abi_three_products_segmented_k_and_activation(hip):
    """lg_gemm_multi3_f32 (three x @ W_i^T + b_i on separately allocated weights, results as column blocks of one buffer),
    lg_gemm_kseg3_f32 (one product whose K runs through three operands; plain, accumulate, addend) and lg_gemm_act_f32 (gelu /
    gelu' in the epilogue) through the raw C ABI against float64, with their argument checks"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(21)
    ptr3 = lambda *ts: (ctypes.c_void_p * 3)(*[t if isinstance(t, int) else t.ptr for t in ts])      # noqa: E731
    M, K, N = 200, 96, 128
    x = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    ws = [rng.uniform(-1, 1, (N, K)).astype(np.float32) for _ in range(3)]
    bs = [rng.uniform(-1, 1, (N,)).astype(np.float32) for _ in range(3)]
    tx, tws, tbs = hip.from_numpy(x), [hip.from_numpy(w) for w in ws], [hip.from_numpy(b) for b in bs]
    packed = hip.from_numpy(np.full((M, 3 * N + 4), 7.0, np.float32))               # row pitch 3N + 4: the last 4 columns stay
    base = packed.ptr
    assert lib.lg_gemm_multi3_f32(0, 1, M, N, K, tx.ptr, K, ptr3(*tws), K, ptr3(base, base + 4 * N, base + 8 * N), 3 * N + 4, ptr3(*tbs)) == 0
    got = packed.numpy()
    for i in range(3):
        ref = x.astype(np.float64) @ ws[i].astype(np.float64).T + bs[i]
        assert rel_err(got[:, i * N:(i + 1) * N], ref) <= 2e-6, i
    np.testing.assert_array_equal(got[:, 3 * N:], 7.0)
    assert lib.lg_gemm_multi3_f32(0, 1, M, N, K, tx.ptr, K, ptr3(*tws), K, ptr3(base, base + 4 * N, base + 8 * N), 3 * N + 4, None) == 0
    assert rel_err(packed.numpy()[:, N:2 * N], x.astype(np.float64) @ ws[1].astype(np.float64).T) <= 2e-6
    assert lib.lg_gemm_multi3_f32(1, 1, M, N, K, tx.ptr, K, ptr3(*tws), K, ptr3(base, base + 4 * N, base + 8 * N), 3 * N + 4, None) == -1
    assert b"transA = 0" in lib.lg_last_error()
    # K through three operands: g (M, 3 * seg) @ [v0; v1; v2], v_i (seg, N2) allocated apart
    seg, N2 = 128, 72
    g = rng.uniform(-1, 1, (M, 3 * seg)).astype(np.float32)
    vs = [rng.uniform(-1, 1, (seg, N2)).astype(np.float32) for _ in range(3)]
    old = rng.uniform(-1, 1, (M, N2)).astype(np.float32)
    tg, tvs = hip.from_numpy(g), [hip.from_numpy(v) for v in vs]
    ref = g.astype(np.float64) @ np.concatenate(vs, axis=0).astype(np.float64)
    out = hip.from_numpy(old.copy())
    assert lib.lg_gemm_kseg3_f32(0, 0, M, N2, seg, tg.ptr, 3 * seg, ptr3(*tvs), N2, out.ptr, N2, 0, None, 0) == 0
    assert rel_err(out.numpy(), ref) <= 2e-6
    assert lib.lg_gemm_kseg3_f32(0, 0, M, N2, seg, tg.ptr, 3 * seg, ptr3(*tvs), N2, out.ptr, N2, 1, None, 0) == 0
    assert rel_err(out.numpy(), 2 * ref) <= 2e-6
    out2, told = hip.from_numpy(np.zeros((M, N2), np.float32)), hip.from_numpy(old)
    assert lib.lg_gemm_kseg3_f32(0, 0, M, N2, seg, tg.ptr, 3 * seg, ptr3(*tvs), N2, out2.ptr, N2, 0, told.ptr, N2) == 0
    assert rel_err(out2.numpy(), ref + old) <= 2e-6
    assert lib.lg_gemm_kseg3_f32(0, 0, M, N2, 96, tg.ptr, 3 * seg, ptr3(*tvs), N2, out.ptr, N2, 0, None, 0) == -1
    assert b"multiple of 64" in lib.lg_last_error()
    # activation in the epilogue
    gelu = lambda t: 0.5 * t * (1.0 + np.tanh(t * 0.7978845608 * (1.0 + 0.044715 * t * t)))             # noqa: E731
    pre, act = hip.from_numpy(np.zeros((M, N), np.float32)), hip.from_numpy(np.zeros((M, N), np.float32))
    assert lib.lg_gemm_act_f32(0, 1, M, N, K, tx.ptr, K, tws[0].ptr, K, pre.ptr, N, tbs[0].ptr, L.ACT_GELU, act.ptr, N) == 0
    ref_pre = x.astype(np.float64) @ ws[0].astype(np.float64).T + bs[0]
    assert rel_err(pre.numpy(), ref_pre) <= 2e-6 and rel_err(act.numpy(), gelu(ref_pre)) <= 2e-6
    np.testing.assert_allclose(act.numpy(), gelu(pre.numpy().astype(np.float64)), rtol=2e-6, atol=1e-7)     # gelu of the stored pre-activation
    up = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    dpre, tup = hip.from_numpy(np.zeros((M, K), np.float32)), hip.from_numpy(up)
    aux = hip.from_numpy(rng.uniform(-2, 2, (M, K)).astype(np.float32))
    assert lib.lg_gemm_act_f32(0, 0, M, K, N, tup.ptr, N, tws[0].ptr, K, dpre.ptr, K, None, L.ACT_GELU_BWD, aux.ptr, K) == 0
    t = aux.numpy().astype(np.float64)
    th = np.tanh(t * 0.7978845608 * (1.0 + 0.044715 * t * t))
    dgelu = 0.5 * (1.0 + th) + 0.5 * t * (1.0 - th * th) * 0.7978845608 * (1.0 + 0.134145 * t * t)
    assert rel_err(dpre.numpy(), (up.astype(np.float64) @ ws[0].astype(np.float64)) * dgelu) <= 2e-6
    assert lib.lg_gemm_act_f32(0, 0, M, K, N, tup.ptr, N, tws[0].ptr, K, dpre.ptr, K, tbs[0].ptr, L.ACT_GELU_BWD, aux.ptr, K) == -1
    assert lib.lg_gemm_act_f32(0, 0, M, K, N, tup.ptr, N, tws[0].ptr, K, dpre.ptr, K, None, 7, aux.ptr, K) == -1


def test_4096_forward_backward_properties(hip):
    """BASELINE config #2 at full size: y = A @ B; y.backward(allow_fill=True)."""
    n = 4096
    np.random.seed(0)
    a = np.random.uniform(-1, 1, (n, n)).astype(np.float32)
    b = np.random.uniform(-1, 1, (n, n)).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    y = ta @ tb
    y.backward(allow_fill=True)
    yn, ga, gb = y.numpy(), ta.grad.numpy(), tb.grad.numpy()
    # (1) sampled blocks against float64 numpy
    rng = np.random.RandomState(1)
    for _ in range(6):
        i, j = rng.randint(0, n - 128, 2)
        ref = a[i:i + 128].astype(np.float64) @ b[:, j:j + 128].astype(np.float64)
        assert rel_err(yn[i:i + 128, j:j + 128], ref) <= 1e-5
    # (2) checksum of checksums: (A @ B) @ 1 == A @ (B @ 1), 1^T (A @ B) == (1^T A) @ B
    ones = np.ones(n)
    np.testing.assert_allclose(yn.astype(np.float64) @ ones, a.astype(np.float64) @ (b.astype(np.float64) @ ones), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(ones @ yn.astype(np.float64), (ones @ a.astype(np.float64)) @ b.astype(np.float64), rtol=1e-4, atol=2e-2)
    # (3) gradients with an all-ones upstream: dA[i, k] = sum_j B[k, j] (every row equal), dB[k, j] = sum_i A[i, k]
    ra, rb = b.astype(np.float64).sum(axis=1), a.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(ga, np.broadcast_to(ra, (n, n)), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(gb, np.broadcast_to(rb[:, None], (n, n)), rtol=1e-4, atol=2e-3)
    assert rel_err(ga, np.broadcast_to(ra, (n, n))) <= 1e-5 and rel_err(gb, np.broadcast_to(rb[:, None], (n, n))) <= 1e-5
    # (4) linearity: (2A) @ B == 2 (A @ B) exactly (power-of-two scaling commutes with rounding)
    y2 = ((ta * 2.0) @ tb).numpy()
    np.testing.assert_array_equal(y2, 2 * yn)


def test_4096_forward_backward_whole_matrices_vs_float64(hip):
    """BASELINE config #2 at full size, no sampling: the WHOLE of y = A @ B, dA = W @ B^T and dB = A^T @ W (random upstream
    gradient W) against float64 numpy - relative Frobenius error <= 1e-5 (north-star tolerance), and element-wise within
    the fp32 forward-error bound"""
    n = 4096
    rng = np.random.RandomState(5)
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    y = ta @ tb
    (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
    a64, b64, w64 = a.astype(np.float64), b.astype(np.float64), w.astype(np.float64)
    for got, ref in ((y.numpy(), a64 @ b64), (ta.grad.numpy(), w64 @ b64.T), (tb.grad.numpy(), a64.T @ w64)):
        assert rel_err(got, ref) <= 1e-5
        # |error| of an fp32 dot product of length 4096 with |terms| <= 1: far below 4096 * 2^-24 * sqrt-ish growth; a wrong
        # tile, a dropped K-slice or a misplaced row would be O(1)
        assert np.abs(got - ref).max() <= 2e-3
    del a64, b64, w64


@pytest.mark.parametrize("mkn", [(512, 1024, 784), (10, 1024, 512), (64, 64, 64), (64, 100, 128), (33, 31, 35), (130, 700, 63),
                                 (1, 5, 1), (200, 36, 300), (128, 2048, 128), (7, 3, 2), (256, 512, 255)])
def test_rowsum_column_of_the_product(hip, mkn):
    """lg_gemm_rowsum_f32: C = A @ B and the row sums of A from one launch (virtual ones-column), every layout,
    N on and off the tile boundary, split-K sizes, write and accumulate modes; plus the nn.Linear (dW, db) use"""
    from lightgrad_amd.autograd.hip import ops as H
    M, K, N = mkn
    rng = np.random.RandomState(M + 13 * K + 101 * N)
    a, b = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
    ref, ref_rs = a.astype(np.float64) @ b.astype(np.float64), a.astype(np.float64).sum(1)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    ta_t = hip.from_numpy(np.ascontiguousarray(a.T)).transpose(1, 0)
    tb_t = hip.from_numpy(np.ascontiguousarray(b.T)).transpose(1, 0)
    atol_rs = 1e-6 * K ** 0.5 * 4
    for x, y, tag in [(ta, tb, "NN"), (ta, tb_t, "NT"), (ta_t, tb, "TN"), (ta_t, tb_t, "TT")]:
        out, rs = H._gemm_rowsum(x, y)
        assert rel_err(out.numpy(), ref) <= 1e-5, tag
        np.testing.assert_allclose(rs.numpy(), ref_rs, rtol=1e-5, atol=atol_rs, err_msg=tag)
    c0, r0 = rng.uniform(-1, 1, (M, N)).astype(np.float32), rng.uniform(-1, 1, (M,)).astype(np.float32)
    tc, tr = hip.from_numpy(c0), hip.from_numpy(r0)
    H._gemm_rowsum(ta_t, tb, accumulate_into=tc, rowsum_into=tr)                       # both accumulate
    assert rel_err(tc.numpy(), ref + c0) <= 1e-5
    np.testing.assert_allclose(tr.numpy(), ref_rs + r0, rtol=1e-5, atol=atol_rs)
    H._gemm_rowsum(ta_t, tb, accumulate_into=tc, overwrite=True, rowsum_into=tr)       # C overwritten, sums accumulate again
    assert rel_err(tc.numpy(), ref) <= 1e-5
    np.testing.assert_allclose(tr.numpy(), 2 * ref_rs + r0, rtol=1e-5, atol=2 * atol_rs)
    H._gemm_rowsum(ta_t, tb, accumulate_into=tc, rowsum_into=tr, rowsum_overwrite=True)
    assert rel_err(tc.numpy(), 2 * ref) <= 1e-5
    np.testing.assert_allclose(tr.numpy(), ref_rs, rtol=1e-5, atol=atol_rs)


def test_long_vectors_and_wide_rows(hip):
    """inner products of multi-million-element vectors and few-row operands with a very long leading dimension
    (tile offsets are 32-bit: the limit depends on rows x ld, not on ld alone)"""
    rng = np.random.RandomState(9)
    n = 3_000_001
    a, b = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    ref = float(a.astype(np.float64) @ b.astype(np.float64))
    assert abs((ta @ tb).item() - ref) <= 1e-5 * float(np.abs(a.astype(np.float64) * b).sum())
    m = rng.uniform(-1, 1, (3, n)).astype(np.float32)
    got = (hip.from_numpy(m) @ tb).numpy()
    ref3 = m.astype(np.float64) @ b.astype(np.float64)
    np.testing.assert_allclose(got, ref3, atol=1e-5 * n ** 0.5 * 4)
    outer = (ta.reshape(n, 1)[:70000] @ tb.reshape(1, n)[:, :5]).numpy()          # K = 1
    np.testing.assert_array_equal(outer, a[:70000, None] * b[None, :5])


def test_attention_shaped_products_forward_backward(hip):
    """(b, s, h, d) head split -> (b, h, s, d) views: scores = q @ k^T, context = p @ v and all four gradients go through
    the two-level batched launch (lg_gemm_batched2_f32), including the column-major gradient of the k^T view"""
    from lightgrad_amd import CpuTensor
    rng = np.random.RandomState(12)
    b, s, h, d = 3, 128, 2, 64
    qn, kn, vn = (rng.uniform(-1, 1, (b, s, h * d)).astype(np.float32) for _ in range(3))
    w = rng.uniform(-1, 1, (b, h, s, d)).astype(np.float32)
    grads = {}
    for cls in (CpuTensor, hip):
        q, k, v = (cls.from_numpy(x) for x in (qn, kn, vn))
        split = lambda t: t.reshape(b, s, h, d).transpose(0, 2, 1, 3)              # noqa: E731
        scores = split(q) @ split(k).transpose(0, 1, 3, 2)
        ctx = (scores * 0.125) @ split(v)
        (ctx * cls.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        grads[cls] = [ctx.numpy()] + [t.grad.numpy() for t in (q, k, v)]
    for got, ref, name in zip(grads[hip], grads[CpuTensor], ["ctx", "dq", "dk", "dv"]):
        assert rel_err(got, ref.astype(np.float64)) <= 2e-5, name


def test_group_queue_edge_cases_through_the_c_abi(hip):
    """lg_gemm_group_*: queued weight-gradient products give the values of immediate launches - more products than the queue holds
    (a flush in the middle), two products onto the SAME output (the second flushes the first, then accumulates), a product with the
    row-sum column, one too large to be queued, LayerNorm parameter gradients riding along, a product of another layout passing
    through a bracket untouched, and a flush forced by lg_sync"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(77)

    def dev(a):
        return hip.from_numpy(np.ascontiguousarray(a, np.float32), requires_grad=False)

    def wgrad(g, x, out, accumulate=0, rowsum=None):
        m, n, k = g.shape[1], x.shape[1], g.shape[0]
        if rowsum is None:
            L.check(lib.lg_gemm_f32(1, 0, m, n, k, g.ptr, m, 0, x.ptr, n, 0, out.ptr, n, 0, 1, accumulate))
        else:
            L.check(lib.lg_gemm_rowsum_f32(1, 0, m, n, k, g.ptr, m, x.ptr, n, out.ptr, n, accumulate, rowsum.ptr, 0))

    shapes = [(256, 96, 80), (512, 128, 128), (100, 33, 50), (1024, 64, 200)] * 5                      # 20 products: > 14
    gs = [dev(rng.uniform(-1, 1, (k, m))) for k, m, n in shapes]
    xs = [dev(rng.uniform(-1, 1, (k, n))) for k, m, n in shapes]
    want = [hip.empty((m, n), requires_grad=False) for k, m, n in shapes]
    for g, x, o in zip(gs, xs, want):
        wgrad(g, x, o)
    got = [hip.empty((m, n), requires_grad=False) for k, m, n in shapes]
    L.check(lib.lg_gemm_group_begin())
    for g, x, o in zip(gs, xs, got):
        wgrad(g, x, o)
    L.check(lib.lg_gemm_group_end())
    L.check(lib.lg_gemm_group_flush())
    for a, b in zip(got, want):            # (a queued product always takes the 64x64 tile, an immediate one may split K differently)
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-5 * np.abs(b.numpy()).max())
    # same output twice (+ accumulate), a row-sum product, an oversized product, a LayerNorm entry, a foreign layout - one bracket
    g1, x1, g2, x2 = gs[1], xs[1], dev(rng.uniform(-1, 1, (512, 128))), dev(rng.uniform(-1, 1, (512, 128)))
    big_g, big_x = dev(rng.uniform(-1, 1, (64, 70000))), dev(rng.uniform(-1, 1, (64, 64)))          # 1094 x 1 tiles > 1024: immediate
    ln_g, ln_xhat = dev(rng.uniform(-1, 1, (300, 96))), dev(rng.uniform(-1, 1, (300, 96)))
    a_nn, b_nn = dev(rng.uniform(-1, 1, (70, 40))), dev(rng.uniform(-1, 1, (40, 90)))
    cs_in = dev(rng.uniform(-1, 1, (333, 1001)))
    sc_ids_np = rng.randint(0, 300, 700).astype(np.int32)
    sc_ids_np[::5] = 11                                                       # a hot row: more than 32 positions
    sc_ids = hip.from_numpy(sc_ids_np, requires_grad=False)
    sc_g = dev(rng.uniform(-1, 1, (700, 12)))

    def run(queued):
        out = hip.empty((128, 128), requires_grad=False)
        rs_out, rs = hip.empty((128, 128), requires_grad=False), hip.empty((128,), requires_grad=False)
        big_out = hip.empty((70000, 64), requires_grad=False)
        dw, db = hip.empty((96,), requires_grad=False), hip.empty((96,), requires_grad=False)
        nn_out = hip.empty((70, 90), requires_grad=False)
        cs_out, cs_acc = hip.empty((1001,), requires_grad=False), dev(np.ones(1001))
        sc_table = dev(np.zeros((300, 12)))
        if queued:
            L.check(lib.lg_gemm_group_begin())
        wgrad(g1, x1, out)                                       # out = g1^T x1
        wgrad(g2, x2, out, accumulate=1)                         # out += g2^T x2: must not share a launch with the first
        wgrad(g1, x1, rs_out, rowsum=rs)
        wgrad(big_g, big_x, big_out)
        L.check(lib.lg_layernorm_param_grads_f32(ln_g.ptr, ln_xhat.ptr, dw.ptr, db.ptr, 300, 96, 0, 0))
        L.check(lib.lg_gemm_f32(0, 0, 70, 90, 40, a_nn.ptr, 40, 0, b_nn.ptr, 90, 0, nn_out.ptr, 90, 0, 1, 0))
        L.check(lib.lg_scatter_add_rows_f32(sc_g.ptr, sc_ids.ptr, 4, sc_table.ptr, 700, 12, 300))          # queued with the LayerNorm entry
        L.check(lib.lg_scatter_add_rows_f32(sc_g.ptr, sc_ids.ptr, 4, sc_table.ptr, 700, 12, 300))          # same table again: flush, then queue
        L.check(lib.lg_gemm_group_colsum_f32(cs_in.ptr, 1001, 333, 1001, cs_out.ptr, 0))      # rides in the group's launch
        L.check(lib.lg_gemm_group_colsum_f32(cs_in.ptr, 1001, 333, 1001, cs_acc.ptr, 1))      # the slot is taken: computed at once
        if queued:
            L.check(lib.lg_gemm_group_end())
            L.check(lib.lg_sync())                               # flushes what is still queued
        return [t.numpy() for t in (out, rs_out, rs, big_out, dw, db, nn_out, cs_out, cs_acc, sc_table)]

    for a, b in zip(run(True), run(False)):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5 * np.abs(b).max())
    want_table = np.zeros((300, 12), np.float64)
    np.add.at(want_table, sc_ids_np, 2 * sc_g.numpy().astype(np.float64))
    np.testing.assert_allclose(run(True)[9], want_table, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(run(True)[7], cs_in.numpy().astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(run(True)[8], 1 + cs_in.numpy().astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    ref = g1.numpy().astype(np.float64).T @ x1.numpy() + g2.numpy().astype(np.float64).T @ x2.numpy()
    np.testing.assert_allclose(run(True)[0], ref, rtol=1e-5, atol=1e-4)


def test_three_products_and_the_loss_in_one_launch_through_the_c_abi(hip):
    """lg_gemm_pair_*: [skinny dW2 (+ db2) with relu on B, dW1 (+ db1), dx] and the loss of a head forward leave as ONE launch and give
    the values of single launches; a held bracket (lg_gemm_pair_hold) lets other products pass untouched; lg_sync launches what is
    held; a bracket that ends without a third product still finishes the loss"""
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    rng = np.random.RandomState(5)
    rows, d_in, hid, outs = 1024, 784, 512, 10

    def dev(a):
        return hip.from_numpy(np.ascontiguousarray(a, np.float32), requires_grad=False)

    err, pre, g1, x, w1 = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((rows, outs), (rows, hid), (rows, hid), (rows, d_in), (hid, d_in)))
    row_loss = rng.uniform(0, 1, (rows,)).astype(np.float32)
    terr, tpre, tg1, tx, tw1, trl = (dev(a) for a in (err, pre, g1, x, w1, row_loss))

    def products(dw2, db2, dw1, db1, dx, between=None):
        # dW2 (+ db2) = err^T @ relu(pre)
        L.check(lib.lg_gemm_fused_f32(1, 0, outs, hid, rows, terr.ptr, outs, tpre.ptr, hid, dw2.ptr, hid, 0, None, db2.ptr, 0, 0, 1))
        if between is not None:
            between()
        L.check(lib.lg_gemm_rowsum_f32(1, 0, hid, d_in, rows, tg1.ptr, hid, tx.ptr, d_in, dw1.ptr, d_in, 0, db1.ptr, 0))
        L.check(lib.lg_gemm_f32(0, 0, rows, d_in, hid, tg1.ptr, hid, 0, tw1.ptr, d_in, 0, dx.ptr, d_in, 0, 1, 0))

    def fresh():
        return [hip.empty(s) for s in ((outs, hid), (outs,), (hid, d_in), (hid,), (rows, d_in))]

    single = fresh()
    products(*single)
    loss_ref = hip.empty(())
    L.check(lib.lg_mse_finalize_f32(trl.ptr, rows, rows * outs, loss_ref.ptr))
    e64, p64, g64 = err.astype(np.float64), np.maximum(pre, 0).astype(np.float64), g1.astype(np.float64)
    refs = [e64.T @ p64, e64.sum(0), g64.T @ x.astype(np.float64), g64.sum(0), g64 @ w1.astype(np.float64)]
    for got, ref in zip(single, refs):
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-5, atol=2e-4)

    def same(a, b):
        for u, v in zip(a, b):
            np.testing.assert_allclose(u.numpy(), v.numpy(), rtol=2e-6, atol=2e-5)      # (K-slice counts may differ: the sums' order)

    # all three in one bracket, the loss riding
    one = fresh()
    loss = hip.empty(())
    before = L.kernel_launches() if hasattr(L, "kernel_launches") else None
    L.check(lib.lg_gemm_pair_begin())
    L.check(lib.lg_gemm_pair_mse_loss(trl.ptr, rows, rows * outs, loss.ptr))
    products(*one)
    L.check(lib.lg_gemm_pair_end())
    same(one, single)
    np.testing.assert_array_equal(loss.numpy(), loss_ref.numpy())
    assert lib.lg_gemm_pair_end() != 0                                        # no bracket open any more

    # held across another product (which must see no bracket: it is complete when its call returns)
    held = fresh()
    loss2 = hip.empty(())
    other = hip.empty((rows, d_in))
    state = {}

    def between():
        L.check(lib.lg_gemm_pair_mse_loss(trl.ptr, rows, rows * outs, loss2.ptr))
        L.check(lib.lg_gemm_pair_hold())
        L.check(lib.lg_gemm_f32(0, 0, rows, d_in, hid, tg1.ptr, hid, 0, tw1.ptr, d_in, 0, other.ptr, d_in, 0, 1, 0))
        state["other"] = other.numpy().copy()                                  # (a device-to-host copy: launches what is held, too)
        L.check(lib.lg_gemm_pair_resume())
    L.check(lib.lg_gemm_pair_begin())
    products(*held, between=between)
    L.check(lib.lg_gemm_pair_end())
    same(held, single)
    np.testing.assert_allclose(state["other"], single[4].numpy(), rtol=2e-6, atol=2e-5)
    np.testing.assert_array_equal(loss2.numpy(), loss_ref.numpy())

    # a bracket with the skinny product alone: a launch of its own at the end, and the loss with it
    alone = fresh()
    loss3 = hip.empty(())
    L.check(lib.lg_gemm_pair_begin())
    L.check(lib.lg_gemm_fused_f32(1, 0, outs, hid, rows, terr.ptr, outs, tpre.ptr, hid, alone[0].ptr, hid, 0, None, alone[1].ptr, 0, 0, 1))
    L.check(lib.lg_gemm_pair_mse_loss(trl.ptr, rows, rows * outs, loss3.ptr))
    L.check(lib.lg_gemm_pair_hold())
    L.check(lib.lg_gemm_pair_end())
    same(alone[:2], single[:2])
    np.testing.assert_array_equal(loss3.numpy(), loss_ref.numpy())
    # hold / resume / loss without a bracket: refused
    assert lib.lg_gemm_pair_hold() != 0 and lib.lg_gemm_pair_resume() != 0
    assert lib.lg_gemm_pair_mse_loss(trl.ptr, rows, rows * outs, loss3.ptr) != 0
