"""Tensors that are not float32 on the device (csrc/typed.hip; VERDICT r3 "Missing 3"): int16 / int32 / int64 / float64 through
neg, add, sub, mul (+ scalar operands, broadcasting, a transposed view), the in-place forms, sum / max / min and astype - against
the fixture recorded from the reference's CPU backend.  Integers and the float64 ring operations bit for bit; float64 sums in
another order than numpy's pairwise blocks (1e-13), pow through another libm (1e-13)."""
import numpy as np
import pytest
import lightgrad_amd as light
from conftest import load_golden


def run_typed_cases(T, g, name, exact_everywhere=False):
    dt = np.dtype(name)
    a, b, row = g[name + "/a"], g[name + "/b"], g[name + "/row"]
    assert a.dtype == dt

    def same(got, key, loose=False):
        want = g[name + "/" + key]
        got = np.asarray(got)
        assert got.dtype == want.dtype, "%s/%s: dtype %s, the reference gives %s" % (name, key, got.dtype, want.dtype)
        assert got.shape == want.shape, "%s/%s: shape %s vs %s" % (name, key, got.shape, want.shape)
        if loose and not exact_everywhere:
            np.testing.assert_allclose(got, want, rtol=1e-13, atol=0, err_msg=key)
        else:
            np.testing.assert_array_equal(got, want, err_msg=key)
    A, B, R = (T.from_numpy(x, requires_grad=False) for x in (a, b, row))
    assert A.dtype == dt
    with light.no_grad():
        same((-A).numpy(), "neg")
        same((A + B).numpy(), "add")
        same((A - B).numpy(), "sub")
        same((A * B).numpy(), "mul")
        same((A + R).numpy(), "add_row")
        same((A.transpose(1, 0) * B.transpose(1, 0)).numpy(), "mul_transposed")
        same((A + 3).numpy(), "add_scalar")
        same((5 * A).numpy(), "rmul_scalar")
        same((A + 0.5).numpy(), "add_float_scalar")
        acc = T.from_numpy(a.copy(), requires_grad=False)
        acc += B
        acc *= R
        acc -= 7
        same(acc.numpy(), "inplace")
        for ax, tag in ((None, "all"), (0, "0"), (1, "1")):
            same(A.sum(axis=ax).numpy(), "sum_" + tag, loose=dt.kind == "f")
            same(A.max(axis=ax).numpy(), "max_" + tag)
            same(A.min(axis=ax).numpy(), "min_" + tag)
        same(A.sum(axis=1, keepdims=True).numpy(), "sum_keepdims", loose=dt.kind == "f")
        if dt.kind == "f":
            same((A / B).numpy(), "div")
            same((B ** A).numpy(), "pow", loose=True)
    if dt.kind == "f":
        A, B, R = T.from_numpy(a), T.from_numpy(b), T.from_numpy(row)
        y = ((A * B + R) * A - B)
        (y * T.from_numpy(g[name + "/w"].reshape(7, 1), requires_grad=False)).backward(allow_fill=True)
        same(y.numpy(), "y")
        for t, key in ((A, "grad_a"), (B, "grad_b"), (R, "grad_row")):
            # (the reference hands the gradient of a float64 tensor out as FLOAT32: add_grad copies the first contribution, and
            #  its copy() goes back to the default dtype - tensor.py:116, cpu/tensor.py:39-40; the fixture records that)
            if exact_everywhere:
                same(t.grad.numpy(), key)
            else:
                np.testing.assert_allclose(t.grad.numpy(), g[name + "/" + key], rtol=2e-6, atol=1e-6, err_msg=key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["int16", "int32", "int64", "float64"])
def test_typed_ops_match_the_reference_fixture(hip, name):
    run_typed_cases(hip, load_golden("typed_ops.npz"), name)


@pytest.mark.gpu
def test_mixed_dtypes_ask_for_a_cast_and_other_ops_stay_float32(hip):
    i = hip.from_numpy(np.arange(6, dtype=np.int32), requires_grad=False)
    f = hip.from_numpy(np.arange(6, dtype=np.float32), requires_grad=False)
    with pytest.raises(TypeError, match="astype"):
        i + f
    np.testing.assert_array_equal((i.astype(np.float32) + f).numpy(), 2 * np.arange(6, dtype=np.float32))
    with pytest.raises(TypeError):
        i.exp()
    with pytest.raises(TypeError):
        i / i                                        # numpy would give float64: ask for the cast


@pytest.mark.gpu
def test_astype_all_pairs_and_its_gradient(hip):
    rng = np.random.RandomState(3)
    src = {np.int16: rng.randint(-300, 300, (5, 7)), np.int32: rng.randint(-70000, 70000, (5, 7)),
           np.int64: rng.randint(-2 ** 40, 2 ** 40, (5, 7)), np.float32: rng.uniform(-300, 300, (5, 7)), np.float64: rng.uniform(-300, 300, (5, 7))}
    for s, arr in src.items():
        arr = arr.astype(s)
        t = hip.from_numpy(arr, requires_grad=False)
        for d in src:
            if np.dtype(d).kind == "i" and np.abs(arr).max() > np.iinfo(d).max:
                continue                              # out-of-range conversions are undefined in numpy too
            got = t.astype(d).numpy()
            want = arr.astype(d)
            assert got.dtype == want.dtype
            np.testing.assert_array_equal(got, want, err_msg="%s -> %s" % (np.dtype(s), np.dtype(d)))
        view = t.transpose(1, 0)[1:4]
        np.testing.assert_array_equal(view.astype(np.float64).numpy(), arr.T[1:4].astype(np.float64))
    a = hip.from_numpy(np.asarray([[-1.7, 2.2], [3.9, -0.1]], np.float32))
    (a.astype(np.float64) * 2.0).sum().backward()
    assert a.grad.dtype == np.float32
    np.testing.assert_array_equal(a.grad.numpy(), np.full((2, 2), 2.0, np.float32))


@pytest.mark.gpu
def test_large_typed_reductions_and_empty_operands(hip):
    rng = np.random.RandomState(4)
    x = rng.randint(-30000, 30000, (300, 1001)).astype(np.int16)
    t = hip.from_numpy(x, requires_grad=False)
    assert t.sum().numpy() == x.sum() and t.sum().dtype == np.int64
    np.testing.assert_array_equal(t.sum(axis=0).numpy(), x.sum(axis=0))
    np.testing.assert_array_equal(t.max(axis=1).numpy(), x.max(axis=1))
    np.testing.assert_array_equal(t.transpose(1, 0).min(axis=1).numpy(), x.T.min(axis=1))
    d = rng.uniform(-1, 1, (257, 513))
    d[3, 5] = np.nan
    td = hip.from_numpy(d, requires_grad=False)
    assert np.isnan(td.max().numpy()) and np.isnan(td.min(axis=0).numpy()[5])
    np.testing.assert_allclose(np.nansum(td.sum(axis=1).numpy()), np.nansum(d.sum(axis=1)), rtol=1e-12)
    e = hip.from_numpy(np.zeros((0, 4), np.int32), requires_grad=False)
    assert (e + e).numpy().shape == (0, 4)
    np.testing.assert_array_equal(e.sum(axis=0).numpy(), np.zeros(4, np.int64))
