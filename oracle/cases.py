"""Shared table of op cases - TEST INFRASTRUCTURE (see oracle/np_oracle.py header).

`op_cases(T, rng)` yields (name, fn, inputs[, upstream]) where `fn` is written against the
lightgrad tensor API only, so the same table drives
  * oracle/gen_golden.py with T = the REFERENCE's CpuTensor (produces tests/golden/ops.npz), and
  * the parity tests with T = this repo's CpuTensor / HipTensor.
`upstream` False marks forward-only cases (no backward exists in the reference's CPU backend).
The rng call sequence is part of the fixture: do not reorder cases without regenerating.
"""
import numpy as np


def f32(rng, low, high, shape):
    return rng.uniform(low, high, size=shape).astype(np.float32)


def op_cases(T, rng):
    x = f32(rng, -1, 1, (13, 17))
    xp = f32(rng, 0.1, 3, (13, 17))
    for name in ["neg", "exp", "sigmoid", "tanh", "relu", "sin", "cos"]:
        yield ("unary_" + name, lambda t, n=name: getattr(t, n)(), [x])
    yield ("unary_log", lambda t: t.log(), [xp])
    xr = x.copy()
    xr[0, :5] = 0.0      # relu / relu.backward exactly at 0 (gradient passes: cpu/ops.py:229)
    yield ("unary_relu_zero", lambda t: t.relu(), [xr])
    # transposed-view input (reference common.py:32-38)
    yield ("unary_exp_T", lambda t: t.transpose(1, 0).exp(), [x])

    # ---------------------------------------------------------------- binary with broadcast variants
    a = f32(rng, -1, 1, (10, 15))
    b = f32(rng, -1, 1, (10, 15))
    bp = f32(rng, 1, 2, (10, 15))
    ap = f32(rng, 1, 2, (10, 15))
    variants = {"full": (slice(None), slice(None)), "row": (slice(0, 1), slice(None)), "col": (slice(None), slice(0, 1))}
    for vname, sl in variants.items():
        yield ("add_" + vname, lambda p, q: p + q, [a, b[sl].copy()])
        yield ("sub_" + vname, lambda p, q: p - q, [a, b[sl].copy()])
        yield ("subfn_" + vname, lambda p, q: T.sub(p, q), [a, b[sl].copy()])
        yield ("mul_" + vname, lambda p, q: p * q, [a, b[sl].copy()])
        yield ("div_" + vname, lambda p, q: p / q, [a, bp[sl].copy()])
        yield ("divfn_" + vname, lambda p, q: T.div(p, q), [a, bp[sl].copy()])
        yield ("pow_" + vname, lambda p, q: p ** q, [ap, bp[sl].copy()])
    yield ("add_vec", lambda p, q: p + q, [a, b[0].copy()])                   # (10,15) + (15,)  bias-add pattern
    yield ("add_scalar", lambda p: p + 1.5, [a])
    yield ("radd_scalar", lambda p: 1.5 + p, [a])
    yield ("mul_scalar", lambda p: 0.1 * p, [a])
    yield ("rsub_scalar", lambda p: 1.0 - p, [a])
    yield ("rdiv_scalar", lambda p: 1 / p, [ap])
    yield ("pow_scalar2", lambda p: p ** 2, [a])
    yield ("pow_scalar_half", lambda p: p ** 0.5, [ap])
    yield ("pow_scalar_m1", lambda p: p ** -1, [ap])
    yield ("pow_scalar_1p5", lambda p: p ** 1.5, [ap])
    yield ("add_T", lambda p, q: p + q.transpose(1, 0), [a, f32(rng, -1, 1, (15, 10))])

    # ---------------------------------------------------------------- dot
    yield ("dot_10x15x10", lambda p, q: p @ q, [a, f32(rng, -1, 1, (15, 10))])
    yield ("dot_13x54x76", lambda p, q: p @ q, [f32(rng, -1, 1, (13, 54)), f32(rng, -1, 1, (54, 76))])
    yield ("dot_64_TT", lambda p, q: p.transpose(1, 0) @ q.transpose(1, 0), [f32(rng, -1, 1, (64, 64)), f32(rng, -1, 1, (64, 64))])
    yield ("dot_linear", lambda xx, ww: xx @ ww.T(1, 0), [f32(rng, -1, 1, (32, 48)), f32(rng, -1, 1, (20, 48))])
    yield ("dot_130x70x200", lambda p, q: p @ q, [f32(rng, -1, 1, (130, 70)), f32(rng, -1, 1, (70, 200))])
    # batched forward only: the reference CPU dot.backward is 2-D only (cpu/ops.py:116)
    yield ("dot_batched_fwd", lambda p, q: p @ q, [f32(rng, -1, 1, (3, 5, 7)), f32(rng, -1, 1, (3, 7, 4))], False)
    yield ("dot_bcast_fwd", lambda p, q: p @ q, [f32(rng, -1, 1, (3, 5, 7)), f32(rng, -1, 1, (7, 4))], False)

    # ---------------------------------------------------------------- reductions
    r = f32(rng, -1, 1, (12, 9))
    r3 = f32(rng, -1, 1, (4, 6, 5))
    for axis in [None, 0, 1]:
        for keep in [False, True]:
            tag = "%s_%s" % ("all" if axis is None else axis, "keep" if keep else "drop")
            # the reference CPU sum has no backward (cpu/ops.py:293): forward only
            yield ("sum_" + tag, lambda t, ax=axis, k=keep: t.sum(axis=ax, keepdims=k), [r], False)
            yield ("mean_" + tag, lambda t, ax=axis, k=keep: t.mean(axis=ax, keepdims=k), [r], False)
            yield ("max_" + tag, lambda t, ax=axis, k=keep: t.max(axis=ax, keepdims=k), [r])
            yield ("min_" + tag, lambda t, ax=axis, k=keep: t.min(axis=ax, keepdims=k), [r])
    yield ("sum_3d_02", lambda t: t.sum(axis=(0, 2)), [r3], False)
    yield ("sum_3d_last_keep", lambda t: t.sum(axis=-1, keepdims=True), [r3], False)
    yield ("max_3d_last_keep", lambda t: t.max(axis=-1, keepdims=True), [r3])
    ties = np.round(r * 2).astype(np.float32) / 2       # many exact ties: every tied max gets the gradient
    yield ("max_ties_1", lambda t: t.max(axis=1), [ties])
    yield ("softmax_fwd", lambda t: t.softmax(axis=-1), [r], False)

    # ---------------------------------------------------------------- shape / index ops (bit-exact)
    yield ("transpose", lambda t: t.transpose(1, 0), [x])
    yield ("transpose3", lambda t: t.transpose(2, 0, 1), [r3])
    yield ("reshape", lambda t: t.reshape(-1), [x])
    yield ("reshape_of_T", lambda t: t.transpose(1, 0).reshape(17, 13), [x])
    yield ("getitem_row", lambda t: t[3], [x])
    yield ("getitem_slice", lambda t: t[2:9, 1:16:3], [x])
    yield ("getitem_3d", lambda t: t[1, :, 2:4], [r3])

