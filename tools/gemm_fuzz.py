"""Randomised parity sweep of the SGEMM entry points against float64 numpy: shapes 1..700 (plus a few large ones), all
four operand layouts, misaligned views (row / column offsets inside a bigger buffer), accumulate, bias, row sums, relu on
either operand, two-level batches, the addend epilogue, paired and queued (group) launches, stride-permuted output layouts.  Run on the GPU box:  python tools/gemm_fuzz.py [seconds=60] [seed=0]"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import ops as H              # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def view(arr2d, transposed):
    """HipTensor viewing values `arr2d` (rows x cols) out of a larger buffer with random offsets; optionally stored
    transposed (so the view is column-major)"""
    r, c = arr2d.shape
    pr, pc = rng.randint(0, 4), rng.randint(0, 6)
    if transposed:
        big = np.zeros((c + pr + rng.randint(0, 3), r + pc + rng.randint(0, 5)), np.float32)
        big[pr:pr + c, pc:pc + r] = arr2d.T
        return HipTensor.from_numpy(big, requires_grad=False)[pr:pr + c, pc:pc + r].transpose(1, 0)
    big = np.zeros((r + pr + rng.randint(0, 3), c + pc + rng.randint(0, 5)), np.float32)
    big[pr:pr + r, pc:pc + c] = arr2d
    return HipTensor.from_numpy(big, requires_grad=False)[pr:pr + r, pc:pc + c]


def rel(got, ref):
    return np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)


def rs_err(got, a_eff):
    """row sums cancel: scale the error by the sum of magnitudes, not by the (possibly tiny) sum"""
    ref = a_eff.sum(1)
    return float(np.max(np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(a_eff).sum(1), 1e-30))) if ref.size else 0.0


t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    big_case = rng.rand() < 0.05
    hi = 2500 if big_case else 700
    M, N, K = (int(rng.randint(1, hi)) for _ in range(3))
    if rng.rand() < 0.3:
        M, N, K = [int(v) if rng.rand() < 0.5 else int(rng.choice([1, 2, 3, 4, 5, 8, 10, 16, 31, 32, 33, 63, 64, 65])) for v in (M, N, K)]
    a, b = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
    ta, tb = view(a, rng.rand() < 0.5), view(b, rng.rand() < 0.5)
    mode = rng.randint(0, 10)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    tag = "M=%d N=%d K=%d mode=%d" % (M, N, K, mode)
    if mode == 0:
        e = rel(H._gemm(ta, tb).numpy(), a64 @ b64)
    elif mode == 1:                                   # accumulate into an existing buffer
        c0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        tc = HipTensor.from_numpy(c0, requires_grad=False)
        H._gemm(ta, tb, accumulate_into=tc)
        e = rel(tc.numpy(), a64 @ b64 + c0)
    elif mode == 2:                                   # bias epilogue
        bias = rng.uniform(-1, 1, (N,)).astype(np.float32)
        e = rel(H._gemm(ta, tb, bias=HipTensor.from_numpy(bias, requires_grad=False)).numpy(), a64 @ b64 + bias)
    elif mode == 3:                                   # product + row sums (dW, db)
        out, rs = H._gemm_rowsum(ta, tb)
        e = max(rel(out.numpy(), a64 @ b64), rs_err(rs.numpy(), a64))
    elif mode == 4:                                   # relu on either operand + row sums / bias
        ra, rb = rng.rand() < 0.5, rng.rand() < 0.5
        a_eff, b_eff = (np.maximum(a64, 0) if ra else a64), (np.maximum(b64, 0) if rb else b64)
        if rng.rand() < 0.5:
            out, rs = H._gemm_fused(ta, tb, relu_a=ra, relu_b=rb, want_rowsum=True)
            e = max(rel(out.numpy(), a_eff @ b_eff), rs_err(rs.numpy(), a_eff))
        else:
            bias = rng.uniform(-1, 1, (N,)).astype(np.float32)
            out, _ = H._gemm_fused(ta, tb, relu_a=ra, relu_b=rb, bias=HipTensor.from_numpy(bias, requires_grad=False))
            e = rel(out.numpy(), a_eff @ b_eff + bias)
    elif mode == 6:                                   # bias + addend epilogue: (a @ b + bias) + r
        bias = rng.uniform(-1, 1, (N,)).astype(np.float32) if rng.rand() < 0.5 else None
        r = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        out = H._gemm(ta, tb, bias=None if bias is None else HipTensor.from_numpy(bias, requires_grad=False),
                      addend=HipTensor.from_numpy(r, requires_grad=False))
        e = rel(out.numpy(), a64 @ b64 + (0 if bias is None else bias) + r)
    elif mode == 7:                                   # two products in one launch (lg_gemm_pair_*): whatever qualifies is paired
        g_ = rng.uniform(-1, 1, (K, M)).astype(np.float32)                       # dW = g^T @ x  and  dx = g @ w
        x_, w_ = rng.uniform(-1, 1, (K, N)).astype(np.float32), rng.uniform(-1, 1, (M, N)).astype(np.float32)
        tg, tx, tw = (HipTensor.from_numpy(v, requires_grad=False) for v in (g_, x_, w_))
        L.check(L.lib().lg_gemm_pair_begin())
        dw = H._gemm(H._swap_last(tg), tx)
        dx = H._gemm(tg, tw)
        L.check(L.lib().lg_gemm_pair_end())
        e = max(rel(dw.numpy(), g_.astype(np.float64).T @ x_), rel(dx.numpy(), g_.astype(np.float64) @ w_))
        tag += " (pair)"
    elif mode == 8:                                   # weight-gradient products queued (lg_gemm_group_*) and flushed together
        cases = []
        L.check(L.lib().lg_gemm_group_begin())
        for _ in range(int(rng.randint(1, 6))):
            kk, mm, nn = int(rng.randint(1, 400)), int(rng.randint(1, 300)), int(rng.randint(1, 300))
            g_, x_ = rng.uniform(-1, 1, (kk, mm)).astype(np.float32), rng.uniform(-1, 1, (kk, nn)).astype(np.float32)
            tg, tx = HipTensor.from_numpy(g_, requires_grad=False), HipTensor.from_numpy(x_, requires_grad=False)
            if rng.rand() < 0.5:
                out, rs = H._gemm_rowsum(H._swap_last(tg), tx)
            else:
                out, rs = H._gemm(H._swap_last(tg), tx), None
            cases.append((g_, x_, out, rs, tg, tx))
        L.check(L.lib().lg_gemm_group_end())
        L.check(L.lib().lg_gemm_group_flush())
        e = 0.0
        for g_, x_, out, rs, _, _ in cases:
            e = max(e, rel(out.numpy(), g_.astype(np.float64).T @ x_))
            if rs is not None:
                e = max(e, rs_err(rs.numpy(), g_.astype(np.float64).T))
        tag = "group of %d" % len(cases)
    elif mode == 9:                                   # batched product stored in a stride-permuted dense layout
        bo, bi, s, d, s2 = (int(rng.randint(1, 4)), int(rng.randint(1, 4)), int(rng.randint(1, 70)), int(rng.randint(1, 40)), int(rng.randint(1, 70)))
        pa, pb = rng.uniform(-1, 1, (bo, bi, s, d)).astype(np.float32), rng.uniform(-1, 1, (bo, bi, d, s2)).astype(np.float32)
        shape = (bo, bi, s, s2)
        perm = list(rng.permutation(4))                                           # memory order, outermost first
        strides, run_ = [0] * 4, 1
        for dim in reversed(perm):
            strides[dim] = run_
            run_ *= shape[dim]
        if 1 not in (strides[-1], strides[-2]) and not (shape[-1] == 1 or shape[-2] == 1):
            strides = None
        out = H._gemm(HipTensor.from_numpy(pa, requires_grad=False), HipTensor.from_numpy(pb, requires_grad=False),
                      out_strides=None if strides is None else tuple(strides))
        e = rel(out.numpy(), pa.astype(np.float64) @ pb)
        tag = "layout %s strides %s" % (shape, strides)
    else:                                             # two-level batch (attention layout), smaller extents
        bo, bi, s, d = int(rng.randint(1, 4)), int(rng.randint(1, 4)), int(rng.randint(1, 70)), int(rng.randint(1, 40))
        q, k = rng.uniform(-1, 1, (bo, s, bi, d)).astype(np.float32), rng.uniform(-1, 1, (bo, s, bi, d)).astype(np.float32)
        tq = HipTensor.from_numpy(q, requires_grad=False).transpose(0, 2, 1, 3)
        tk = HipTensor.from_numpy(k, requires_grad=False).transpose(0, 2, 3, 1)
        e = rel((tq @ tk).numpy(), q.transpose(0, 2, 1, 3).astype(np.float64) @ k.transpose(0, 2, 3, 1))
        tag = "batched2 %s" % ((bo, bi, s, d),)
    worst = max(worst, e)
    if not (e <= 2e-5):
        print("MISMATCH", tag, "rel err %.3e" % e)
        sys.exit(1)
    n += 1
print("gemm_fuzz: %d cases in %.0f s, worst relative Frobenius error %.2e" % (n, time.time() - t0, worst))
