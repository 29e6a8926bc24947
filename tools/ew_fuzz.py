"""Randomised parity sweep of the elementwise / reduction entry points against numpy: 1-4-D shapes, broadcasting, transposed
and sliced (misaligned) views, in-place forms, two-output backward forms, sum / max / min over random axis sets.
+ - x / relu max min are compared EXACTLY, sums with a K-scaled tolerance.   python tools/ew_fuzz.py [seconds=60] [seed=0]"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightgrad_amd as light                                # noqa: E402
from lightgrad_amd import HipTensor                          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rand_shape():
    nd = rng.randint(1, 5)
    dims = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 17, 31, 32, 33, 64, 65, 100, 128, 130, 257, 512, 1000])) for _ in range(nd)]
    while np.prod(dims) > 3_000_000:
        dims[int(np.argmax(dims))] //= 2
    return tuple(max(1, d) for d in dims)


def rand_view(shape):
    """(numpy array, HipTensor) with the same values; the tensor may be a transposed and / or sliced view of a bigger buffer"""
    nd = len(shape)
    perm = list(rng.permutation(nd)) if rng.rand() < 0.4 else list(range(nd))
    stored = tuple(shape[perm.index(i)] for i in range(nd))             # shape in memory order before the transpose back
    pads = [(int(rng.randint(0, 3)), int(rng.randint(0, 3))) if rng.rand() < 0.4 else (0, 0) for _ in range(nd)]
    big = rng.uniform(-2, 2, tuple(s + p[0] + p[1] for s, p in zip(stored, pads))).astype(np.float32)
    sl = tuple(slice(p[0], p[0] + s) for s, p in zip(stored, pads))
    t = HipTensor.from_numpy(big, requires_grad=False)[sl]
    a = big[sl]
    inv = [perm.index(i) for i in range(nd)]                             # view with `shape`: axis i is stored axis inv... apply perm
    a, t = a.transpose(perm), t.transpose(*perm)
    assert a.shape == tuple(stored[p] for p in perm)
    return a, t


def broadcastable(shape):
    s = list(shape)
    for i in range(len(s)):
        if rng.rand() < 0.3:
            s[i] = 1
    if rng.rand() < 0.3 and len(s) > 1:
        s = s[int(rng.randint(1, len(s))):]
    return tuple(s)


t0, n = time.time(), 0
while time.time() - t0 < budget:
    a, ta = rand_view(rand_shape())
    shape = a.shape
    kind = rng.randint(0, 7)
    tag = "shape=%s kind=%d" % (shape, kind)
    try:
        if kind == 0:                                 # binary with broadcasting
            bshape = broadcastable(shape)
            b, tb = rand_view(bshape)
            for name, f in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y)):
                np.testing.assert_array_equal(f(ta, tb).numpy(), f(a, b), err_msg=tag + " " + name)
            np.testing.assert_allclose((ta / (tb * tb + 1.0)).numpy(), a / (b * b + np.float32(1.0)), rtol=2e-6, err_msg=tag)
        elif kind == 1:                               # scalar operand and unary
            np.testing.assert_array_equal((ta * 1.5 + 0.25).numpy(), a * np.float32(1.5) + np.float32(0.25), err_msg=tag)
            np.testing.assert_array_equal(ta.relu().numpy(), np.maximum(a, 0), err_msg=tag)
            np.testing.assert_allclose(ta.exp().numpy(), np.exp(a), rtol=2e-6, err_msg=tag)
        elif kind == 2:                               # in place into a dense tensor
            bshape = broadcastable(shape)
            b, tb = rand_view(bshape)
            if np.broadcast_shapes(shape, b.shape) == shape:
                acc = HipTensor.from_numpy(np.ascontiguousarray(a), requires_grad=False)
                with light.no_grad():
                    acc += tb
                    acc *= tb
                np.testing.assert_array_equal(acc.numpy(), (a + b) * b, err_msg=tag)
        elif kind == 3:                               # contiguous() / reshape of a view
            np.testing.assert_array_equal(ta.contiguous().numpy(), a, err_msg=tag)
            np.testing.assert_array_equal(ta.reshape(-1).numpy(), a.reshape(-1), err_msg=tag)
        elif kind == 4:                               # reductions over a random axis set
            axes = tuple(sorted(set(int(x) for x in rng.randint(0, len(shape), rng.randint(1, len(shape) + 1)))))
            keep = bool(rng.rand() < 0.5)
            a64 = a.astype(np.float64)
            red = 1
            for ax in axes:
                red *= shape[ax]
            got, ref = ta.sum(axis=axes, keepdims=keep).numpy(), a64.sum(axis=axes, keepdims=keep)
            scale = np.abs(a64).sum(axis=axes, keepdims=keep)             # sums cancel: error relative to the sum of magnitudes
            assert np.all(np.abs(got - ref) <= 1e-6 * scale + 1e-6), tag + " sum %s: worst %.2e of sum|x|" % (
                axes, float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-30))))
            np.testing.assert_array_equal(ta.max(axis=axes, keepdims=keep).numpy(), a.max(axis=axes, keepdims=keep), err_msg=tag)
            np.testing.assert_array_equal(ta.min(axis=axes, keepdims=keep).numpy(), a.min(axis=axes, keepdims=keep), err_msg=tag)
        elif kind == 5:                               # two-output backward form through the tape
            b, tb = rand_view(shape)
            x, y = HipTensor.from_numpy(np.ascontiguousarray(a)), HipTensor.from_numpy(np.ascontiguousarray(b))
            (x * y).backward(allow_fill=True)
            np.testing.assert_array_equal(x.grad.numpy(), b, err_msg=tag)
            np.testing.assert_array_equal(y.grad.numpy(), a, err_msg=tag)
        else:                                         # full reductions and un-broadcast gradient
            assert abs(ta.sum().item() - a.astype(np.float64).sum()) <= 1e-6 * np.abs(a.astype(np.float64)).sum() + 1e-6, tag
            assert ta.max().item() == a.max() and ta.min().item() == a.min(), tag
    except AssertionError as e:
        print("MISMATCH", tag, str(e)[:600])
        sys.exit(1)
    n += 1
print("ew_fuzz: %d cases in %.0f s, all within tolerance" % (n, time.time() - t0))
