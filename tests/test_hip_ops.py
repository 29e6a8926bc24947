"""Parity of the HIP path (HipTensor -> C ABI -> gfx950 kernels) with the reference:
golden fixtures produced by the real reference, the numpy oracle on seeded inputs, the repo's CPU
backend, and numerical gradient checks.  Shapes/recipes follow the reference's device test-suite
(test/test_opencl_tensor.py:27-147).  Tolerance: bit-exact for shape/index ops, max/min, relu, neg and the
correctly-rounded arithmetic (+ - * /); fp32 relative 1e-5 for transcendental ops (north star)."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from common import compare_with_numpy, compare_with_cpu, check_gradients, replay_op_cases
import np_oracle as O

pytestmark = pytest.mark.gpu

EXACT_PREFIXES = ("unary_neg", "unary_relu", "add_", "sub_", "subfn_", "mul_", "radd_", "rsub_", "divfn_", "max_", "min_",
                  "transpose", "reshape", "getitem", "pow_scalar2", "pow_scalar_half", "pow_scalar_m1", "rdiv_")


def test_golden_fixtures(hip, golden_ops):
    stats = {"exact": 0, "close": 0}

    def check(name, kind, got, expected):
        reduced_grad = kind != "out" and any(tag in name for tag in ("_row", "_col", "_vec"))   # un-broadcast = a sum
        pow_grad = kind != "out" and name.startswith(("pow_", "rdiv_"))                        # a**(b-1) goes through powf
        if name.startswith(EXACT_PREFIXES) and not reduced_grad and not pow_grad:
            np.testing.assert_array_equal(got, expected, err_msg="%s/%s" % (name, kind))
            stats["exact"] += 1
        else:
            tol = dict(rtol=1e-5, atol=2e-6)
            if name.startswith(("dot_", "sum_", "mean_")):
                tol = dict(rtol=1e-5, atol=1e-5)        # accumulation order differs from BLAS / pairwise numpy
            np.testing.assert_allclose(got, expected, err_msg="%s/%s" % (name, kind), **tol)
            stats["close"] += 1
    assert replay_op_cases(hip, golden_ops, check) >= 85
    assert stats["exact"] > 60 and stats["close"] > 60


# ---- values: the reference's Test_OpenCLTensor recipes (test_opencl_tensor.py:27-87) ----
VALUE_CASES = {
    "transpose": lambda T: compare_with_numpy(T, lambda t: t.transpose(1, 0), shapes=[(64, 64)], rtol=0, atol=0),
    "reshape": lambda T: compare_with_numpy(T, lambda t: t.reshape(-1), shapes=[(64, 64)], rtol=0, atol=0),
    "neg": lambda T: compare_with_numpy(T, lambda x: -x, shapes=[(64, 64)], rtol=0, atol=0),
    "sin": lambda T: compare_with_numpy(T, "sin", shapes=[(64, 64)]),
    "cos": lambda T: compare_with_numpy(T, "cos", shapes=[(64, 64)]),
    "exp": lambda T: compare_with_numpy(T, "exp", shapes=[(64, 64)]),
    "log": lambda T: compare_with_numpy(T, "log", shapes=[(64, 64)], lowhigh=(1e-3, 1)),
    "tanh": lambda T: compare_with_numpy(T, "tanh", shapes=[(64, 64)]),
    "sigmoid": lambda T: compare_with_cpu(T, "sigmoid", shapes=[(64, 64)]),
    "relu": lambda T: compare_with_cpu(T, "relu", shapes=[(64, 64)], rtol=0, atol=0),
    "add": lambda T: compare_with_numpy(T, lambda a, b: a + b, shapes=[(64, 64), (64, 64)], broadcast=True, rtol=0, atol=0),
    "sub": lambda T: compare_with_numpy(T, lambda a, b: a - b, shapes=[(64, 64), (64, 64)], broadcast=True, rtol=0, atol=0),
    "mul": lambda T: compare_with_numpy(T, lambda a, b: a * b, shapes=[(64, 64), (64, 64)], broadcast=True, rtol=0, atol=0),
    "pow": lambda T: compare_with_numpy(T, lambda a, b: a ** b, shapes=[(64, 64), (64, 64)], broadcast=True, lowhigh=(0.01, 1)),
    "div+": lambda T: compare_with_numpy(T, lambda a, b: a / b, shapes=[(64, 64), (64, 64)], broadcast=True, lowhigh=(0.1, 10)),
    "div-": lambda T: compare_with_numpy(T, lambda a, b: a / b, shapes=[(64, 64), (64, 64)], broadcast=True, lowhigh=(-10, -0.1)),
    "dot_T": lambda T: compare_with_numpy(T, lambda a, b: a @ b, shapes=[(64, 64), (64, 64)], transpose=True),
    "dot_rect": lambda T: compare_with_numpy(T, lambda a, b: a @ b, shapes=[(32, 64), (64, 128)]),
    "dot_odd": lambda T: compare_with_numpy(T, lambda a, b: a @ b, shapes=[(13, 54), (54, 76)]),
    "sum": lambda T: [compare_with_numpy(T, "sum", shapes=[(64, 64)], axis=ax, atol=2e-5) for ax in (None, 0, 1)],
    "mean": lambda T: [compare_with_numpy(T, "mean", shapes=[(64, 64)], axis=ax) for ax in (None, 0, 1)],
    "min": lambda T: [compare_with_numpy(T, "min", shapes=[(64, 64)], axis=ax, rtol=0, atol=0) for ax in (None, 0, 1)],
    "max": lambda T: [compare_with_numpy(T, "max", shapes=[(64, 64)], axis=ax, rtol=0, atol=0) for ax in (None, 0, 1)],
    "max_T": lambda T: compare_with_numpy(T, "max", shapes=[(48, 80)], transpose=True, axis=1, rtol=0, atol=0),
    "sum_T": lambda T: compare_with_numpy(T, "sum", shapes=[(48, 80)], transpose=True, axis=0, atol=2e-5),
    "sum_3d": lambda T: [compare_with_numpy(T, "sum", shapes=[(6, 10, 12)], axis=ax, atol=2e-5) for ax in ((0, 2), (1,), (0, 1), -1)],
}


@pytest.mark.parametrize("name", sorted(VALUE_CASES))
def test_values(hip, name):
    np.random.seed(1337 + sum(map(ord, name)))
    VALUE_CASES[name](hip)


# ---- gradients: Test_OpenCL_GradCheck recipes (test_opencl_tensor.py:90-147) ----
def _g(T, *a, **k):
    return check_gradients(T, *a, **k)


GRAD_CASES = {
    "transpose": lambda T: _g(T, lambda x: x.transpose(1, 0), shapes=[(15, 15)]),
    "reshape": lambda T: _g(T, lambda x: x.reshape(-1), shapes=[(15, 15)]),
    "neg": lambda T: _g(T, "neg", shapes=[(15, 15)], broadcast=True, transpose=True),
    "sin": lambda T: _g(T, "sin", shapes=[(15, 15)], broadcast=True, transpose=True),
    "cos": lambda T: _g(T, "cos", shapes=[(15, 15)], broadcast=True, transpose=True),
    "exp": lambda T: _g(T, "exp", shapes=[(15, 15)], broadcast=True, transpose=True),
    "log": lambda T: _g(T, "log", shapes=[(15, 15)], broadcast=True, transpose=True, lowhigh=(0.1, 10), tol=2e-3),
    "sigmoid": lambda T: _g(T, "sigmoid", shapes=[(15, 15)], broadcast=True, transpose=True),
    "tanh": lambda T: _g(T, "tanh", shapes=[(15, 15)], broadcast=True, transpose=True),
    "relu": lambda T: _g(T, "relu", shapes=[(15, 15)], broadcast=True, transpose=True, eps=1e-5, tol=0.002),
    "max": lambda T: [_g(T, "max", shapes=[(2, 2)]), _g(T, "max", shapes=[(2, 2)], axis=0), _g(T, "max", shapes=[(3, 4)], axis=1)],
    "min": lambda T: [_g(T, "min", shapes=[(2, 2)]), _g(T, "min", shapes=[(2, 2)], axis=0), _g(T, "min", shapes=[(3, 4)], axis=1)],
    "sum": lambda T: [_g(T, "sum", shapes=[(2, 2)], transpose=True), _g(T, "sum", shapes=[(2, 2)], axis=0, transpose=True),
                      _g(T, "sum", shapes=[(2, 2)], axis=1, transpose=True)],
    "add": lambda T: _g(T, "add", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True),
    "sub": lambda T: _g(T, "sub", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True),
    "mul": lambda T: _g(T, "mul", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True),
    "pow": lambda T: _g(T, "pow", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True, lowhigh=(0.5, 1), eps=1e-4, tol=0.01),
    "div+": lambda T: _g(T, "div", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True, lowhigh=(0.5, 5), tol=0.005),
    "div-": lambda T: _g(T, "div", shapes=[(5, 5), (5, 5)], broadcast=True, transpose=True, lowhigh=(-5, -0.5), tol=0.005),
    "dot": lambda T: [_g(T, "dot", shapes=[(5, 5), (5, 5)], transpose=True), _g(T, "dot", shapes=[(9, 4), (4, 14)])],
    "dot_batched": lambda T: [_g(T, "dot", shapes=[(2, 4, 5), (2, 5, 3)], tol=2e-3), _g(T, "dot", shapes=[(2, 4, 5), (5, 3)], tol=2e-3)],
    "mean": lambda T: _g(T, lambda x: x.mean(axis=1), shapes=[(6, 7)]),
    "softmax": lambda T: _g(T, lambda x: x.softmax(axis=-1), shapes=[(5, 6)], tol=2e-3),
}


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_gradcheck(hip, name):
    np.random.seed(1337 + sum(map(ord, name)))
    GRAD_CASES[name](hip)


def test_linear_model_gradcheck_and_cpu_parity(hip):
    """test_opencl_tensor.py:134-178: same weights on both backends, compare forward and all gradients"""
    import lightgrad_amd.nn as nn
    np.random.seed(11)

    class Model(nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.l1 = nn.Linear(8, 16)
            self.l2 = nn.Linear(16, 4)

        def forward(self, x):
            return self.l2(self.l1(x).tanh())
    cpu_model, hip_model = Model(), Model()
    hip_model.load_parameters(cpu_model.named_parameters())
    hip_model.map_parameters(lambda p: p.hip())
    assert all(isinstance(p, hip) for p in hip_model.parameters())
    x = CpuTensor.uniform(-1, 1, (4, 8))
    cy, hy = cpu_model(x), hip_model(x.hip())
    np.testing.assert_allclose(hy.numpy(), cy.numpy(), rtol=1e-5, atol=1e-6)
    cy.backward(True)
    hy.backward(True)
    for (n, p), (_, q) in zip(cpu_model.named_parameters(), hip_model.named_parameters()):
        np.testing.assert_allclose(q.grad.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-6, err_msg=n)
    check_gradients(hip, hip_model, shapes=[(16, 8)])


# ---- views, indexing, in-place and dtype-generic layout ops (bit-exact) ----
def test_views_and_indexing(hip):
    rng = np.random.RandomState(0)
    a = rng.uniform(-1, 1, (6, 7, 8)).astype(np.float32)
    t = hip.from_numpy(a)
    for idx in [2, (1, 3), (slice(1, 5), 2), (slice(None), slice(0, 7, 2), slice(3, None)), (Ellipsis, 1), (-1, slice(None, None, -1))]:
        np.testing.assert_array_equal(t[idx].numpy(), a[idx])
    v = t.transpose(2, 0, 1)
    assert not v.is_contiguous() and v.data is t.data               # a view: same storage
    np.testing.assert_array_equal(v.numpy(), a.transpose(2, 0, 1))
    np.testing.assert_array_equal(v.reshape(8, -1).numpy(), a.transpose(2, 0, 1).reshape(8, -1))
    np.testing.assert_array_equal(v[3, 1:4].numpy(), a.transpose(2, 0, 1)[3, 1:4])
    b = a.copy()
    t[1:3, ::2, 4] = 7.5
    b[1:3, ::2, 4] = 7.5
    t[0] = hip.from_numpy(a[5])
    b[0] = a[5]
    t[:, 2, :] = hip.from_numpy(a[0, 0])                           # broadcast row
    b[:, 2, :] = a[0, 0]
    np.testing.assert_array_equal(t.numpy(), b)
    np.testing.assert_array_equal(t[[0, 1]].numpy(), b[[0, 1]])      # integer-array index: device gather (tests/test_hip_index.py)
    np.testing.assert_array_equal(t[[0, 1], [1, 2], [0, 0]].numpy(), b[[0, 1], [1, 2], [0, 0]])   # three index arrays: one flat index
    np.testing.assert_array_equal(t[[0, 1], :, [0, 0]].numpy(), b[[0, 1], :, [0, 0]])   # index arrays apart: their dimension goes first
    with pytest.raises(IndexError):
        t[6]


def test_inplace_ops_alias_storage(hip):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    t = hip.from_numpy(a)
    with light.no_grad():
        u = t
        u += 1.0
        u *= hip.from_numpy(np.full((4,), 2, np.float32))
        u -= hip.from_numpy(a).transpose(1, 0).reshape(4, 3).transpose(1, 0)     # strided operand
        u /= 4.0
    expect = ((a + 1) * 2 - a.T.reshape(4, 3).T) / 4
    np.testing.assert_array_equal(t.numpy(), expect)
    assert u.data is t.data
    g = hip.zeros((4, 3))
    g += hip.from_numpy(a).transpose(1, 0)                                        # grad += transposed view
    np.testing.assert_array_equal(g.numpy(), a.T)


@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.int64, np.float64, np.int16, np.uint8])
def test_layout_ops_any_dtype(hip, dtype):
    rng = np.random.RandomState(1)
    a = (rng.uniform(-100, 100, (5, 6, 7))).astype(dtype)
    t = hip.from_numpy(a)
    assert t.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(t.numpy(), a)
    np.testing.assert_array_equal(t.transpose(1, 2, 0).numpy(), a.transpose(1, 2, 0))
    np.testing.assert_array_equal(t.transpose(1, 2, 0).reshape(-1).numpy(), a.transpose(1, 2, 0).reshape(-1))
    np.testing.assert_array_equal(t[1:4, 2].copy().numpy(), a[1:4, 2])
    z = hip.zeros((3, 5), dtype=dtype)
    z[1] = 3
    e = np.zeros((3, 5), dtype)
    e[1] = 3
    np.testing.assert_array_equal(z.numpy(), e)
    if np.dtype(dtype) == np.uint8:
        with pytest.raises(TypeError, match="not defined for dtype uint8"):
            t + t                        # arithmetic: float32, float64, int16, int32, int64 (tests/test_hip_typed.py)
    elif np.dtype(dtype) != np.float32:
        np.testing.assert_array_equal((t + t).numpy(), a + a)
        with pytest.raises(TypeError):
            t.exp()


def test_scalar_and_empty_shapes(hip):
    s = hip.from_numpy(np.float32(3.0).reshape(()))
    assert s.shape == () and s.numel() == 1 and s.item() == 3.0
    r = hip.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3)).sum()
    assert r.shape == ()                                   # like the CPU backend, not (1,) as the reference's OpenCL tensor
    assert r.item() == 15.0
    e = hip.from_numpy(np.zeros((0, 4), np.float32))
    assert (e + 1.0).numpy().shape == (0, 4) and e.sum(axis=0).numpy().tolist() == [0, 0, 0, 0]
    assert (e.transpose(1, 0) @ e).numpy().shape == (4, 4) and not (e.transpose(1, 0) @ e).numpy().any()
    nine = hip.from_numpy(np.ones((1, 2, 1, 2, 1, 2, 1, 2), np.float32))     # LG_MAX_DIMS
    np.testing.assert_array_equal((nine * 2).sum(axis=(1, 3)).numpy(), np.full((1, 1, 1, 2, 1, 2), 8.0, np.float32))


def test_nan_inf_propagation(hip):
    a = np.array([[np.nan, 1, -np.inf], [2, np.inf, -0.0]], np.float32)
    t = hip.from_numpy(a)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(t.relu().numpy(), np.maximum(a, 0.0))
        np.testing.assert_array_equal(t.max(axis=1).numpy(), np.max(a, axis=1))
        np.testing.assert_array_equal(t.min(axis=0).numpy(), np.min(a, axis=0))
        np.testing.assert_allclose(t.exp().numpy(), np.exp(a), rtol=1e-6)


# ---- sizes of BASELINE's microbench configs: 4096^2 contiguous, broadcast and transposed variants ----
def test_elementwise_full_size(hip):
    rng = np.random.RandomState(0)
    n = 4096
    a, b = rng.uniform(-1, 1, (n, n)).astype(np.float32), rng.uniform(-1, 1, (n, n)).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    np.testing.assert_array_equal((ta + tb).numpy(), a + b)
    np.testing.assert_array_equal((ta * tb).numpy(), a * b)
    np.testing.assert_array_equal(ta.relu().numpy(), np.maximum(a, 0))
    np.testing.assert_allclose(ta.exp().numpy(), np.exp(a), rtol=1e-5)
    np.testing.assert_array_equal((ta + tb.transpose(1, 0)).numpy(), a + b.T)          # LDS-transposing path
    np.testing.assert_array_equal((ta.transpose(1, 0) * tb.transpose(1, 0)).numpy(), a.T * b.T)
    bias = rng.uniform(-1, 1, (n,)).astype(np.float32)
    np.testing.assert_array_equal((ta + hip.from_numpy(bias)).numpy(), a + bias)       # rows path
    col = rng.uniform(-1, 1, (n, 1)).astype(np.float32)
    np.testing.assert_array_equal((ta * hip.from_numpy(col)).numpy(), a * col)
    # odd sizes / unaligned views take the scalar tails
    c = rng.uniform(-1, 1, (1001, 1003)).astype(np.float32)
    tc = hip.from_numpy(c)
    np.testing.assert_array_equal((tc[1:, 1:] + tc[:-1, :-1]).numpy(), c[1:, 1:] + c[:-1, :-1])
    np.testing.assert_array_equal((tc.reshape(-1)[3:] * 2.0).numpy(), c.reshape(-1)[3:] * 2.0)


@pytest.mark.parametrize("shape", [(128, 192), (64, 64), (68, 132), (130, 67), (17, 16), (200, 8192), (4100, 36)])
def test_transposed_operand_tiles(hip, shape):
    """both forms of the LDS-transposing tile (float4 when every extent/stride is a multiple of 4, scalar otherwise):
    one or two transposed inputs, a row-contiguous and a row-broadcast input next to them, two outputs, in place,
    and views that break the 16-byte alignment; shape/index work, so exact"""
    rng = np.random.RandomState(3)
    r, c = shape
    a = rng.uniform(-1, 1, (r, c)).astype(np.float32)
    bt = rng.uniform(-1, 1, (c, r)).astype(np.float32)
    ct = rng.uniform(-1, 1, (c, r)).astype(np.float32)
    col = rng.uniform(-1, 1, (r, 1)).astype(np.float32)
    ta, tbt, tct, tcol = (hip.from_numpy(x) for x in (a, bt, ct, col))
    np.testing.assert_array_equal(tbt.transpose(1, 0).contiguous().numpy(), bt.T)
    np.testing.assert_array_equal((ta + tbt.transpose(1, 0)).numpy(), a + bt.T)
    np.testing.assert_array_equal((tbt.transpose(1, 0) * tct.transpose(1, 0)).numpy(), bt.T * ct.T)
    np.testing.assert_array_equal((tbt.transpose(1, 0) * tcol).numpy(), bt.T * col)
    y = ta * tbt.transpose(1, 0)                                    # mul backward writes two outputs from (a, b^T, g^T)
    (y * tct.transpose(1, 0)).backward(allow_fill=True)
    np.testing.assert_array_equal(ta.grad.numpy(), ct.T * bt.T)
    np.testing.assert_array_equal(tbt.grad.numpy(), (a * ct.T).T)
    with light.no_grad():
        acc = hip.from_numpy(a.copy())
        acc += tbt.transpose(1, 0)
    np.testing.assert_array_equal(acc.numpy(), a + bt.T)
    if r > 20 and c > 20:                                           # offset views: no operand is 16-byte aligned any more
        np.testing.assert_array_equal((ta[1:, 1:] + tbt.transpose(1, 0)[1:, 1:]).numpy(), a[1:, 1:] + bt.T[1:, 1:])
        np.testing.assert_array_equal((ta[:-4, 4:] + tbt[4:, :-4].transpose(1, 0)).numpy(), a[:-4, 4:] + bt[4:, :-4].T)


@pytest.mark.parametrize("shape", [(5796, 5804), (8192, 4096), (4100, 8200)])
def test_transposed_operand_large_tiles(hip, shape):
    """from 2^25 elements on, one transposed input goes through 128 x 128 tiles moved by 1024 threads: whole tiles and
    ragged edges (extents that are multiples of 4 but not of 128), next to a dense and a column operand, in place, and the
    gradient of a transposed view; exact"""
    rng = np.random.RandomState(5)
    r, c = shape
    assert r * c >= 1 << 25
    a = rng.uniform(-1, 1, (r, c)).astype(np.float32)
    bt = rng.uniform(-1, 1, (c, r)).astype(np.float32)
    col = rng.uniform(-1, 1, (r, 1)).astype(np.float32)
    ta, tbt, tcol = hip.from_numpy(a), hip.from_numpy(bt), hip.from_numpy(col)
    np.testing.assert_array_equal(tbt.transpose(1, 0).contiguous().numpy(), bt.T)
    np.testing.assert_array_equal((ta + tbt.transpose(1, 0)).numpy(), a + bt.T)
    np.testing.assert_array_equal((tbt.transpose(1, 0) * tcol).numpy(), bt.T * col)
    with light.no_grad():
        ta += tbt.transpose(1, 0)
    np.testing.assert_array_equal(ta.numpy(), a + bt.T)
    (tbt.transpose(1, 0) * ta).backward(allow_fill=True)                       # two outputs from (b^T, a, ones)
    np.testing.assert_array_equal(tbt.grad.numpy(), (a + bt.T).T)


@pytest.mark.parametrize("shape", [(1024, 30522), (36, 10), (4, 7), (128, 513), (2, 3, 5, 26)])
def test_dense_outputs_with_odd_inner_width(hip, shape):
    """row / column broadcasts and two-output backward forms when the inner width is not a multiple of 4 (the float4
    flat-2D path): bias add of the BERT decoder (1024, 30522) + (30522,), the MLP's (1024, 10) + (10,) shape class"""
    rng = np.random.RandomState(4)
    a = rng.uniform(-1, 1, shape).astype(np.float32)
    row = rng.uniform(-1, 1, shape[-1:]).astype(np.float32)
    col = rng.uniform(-1, 1, shape[:-1] + (1,)).astype(np.float32)
    ta, trow, tcol = hip.from_numpy(a), hip.from_numpy(row), hip.from_numpy(col)
    np.testing.assert_array_equal((ta + trow).numpy(), a + row)
    np.testing.assert_array_equal((tcol * ta).numpy(), col * a)
    np.testing.assert_array_equal((trow - tcol).numpy(), row - col)               # both operands broadcast
    y = ta * trow
    (y * tcol).backward(allow_fill=True)
    np.testing.assert_array_equal(ta.grad.numpy(), col * row * np.ones_like(a))
    np.testing.assert_allclose(trow.grad.numpy(), (col * a).reshape(-1, shape[-1]).astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    with light.no_grad():
        acc = hip.from_numpy(a.copy())
        acc += trow
        acc *= tcol
    np.testing.assert_array_equal(acc.numpy(), (a + row) * col)


def test_reductions_full_size(hip):
    rng = np.random.RandomState(1)
    n = 4096
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    t = hip.from_numpy(a)
    a64 = a.astype(np.float64)
    scale = np.abs(a64).sum()
    assert abs(t.sum().item() - a64.sum()) <= 1e-7 * scale
    np.testing.assert_allclose(t.sum(axis=0).numpy(), a64.sum(axis=0), atol=1e-5 * n ** 0.5 * 2)
    np.testing.assert_allclose(t.sum(axis=1).numpy(), a64.sum(axis=1), atol=1e-5 * n ** 0.5 * 2)
    np.testing.assert_allclose(t.transpose(1, 0).sum(axis=1).numpy(), a64.sum(axis=0), atol=1e-5 * n ** 0.5 * 2)
    assert t.max().item() == a.max() and t.min().item() == a.min()
    np.testing.assert_array_equal(t.max(axis=0).numpy(), a.max(axis=0))
    np.testing.assert_array_equal(t.max(axis=1).numpy(), a.max(axis=1))
    p = np.abs(a)
    np.testing.assert_allclose(hip.from_numpy(p).sum().item(), p.astype(np.float64).sum(), rtol=1e-6)
    # size-independent properties: sum is linear, max of the row maxima is the global max
    np.testing.assert_allclose((t * 2.0).sum(axis=1).numpy(), 2 * t.sum(axis=1).numpy(), rtol=0, atol=0)
    assert t.max(axis=1).max().item() == t.max().item()
    # un-broadcast pattern of func.py:50-56 on the MLP shapes
    g = rng.uniform(-1, 1, (1024, 512)).astype(np.float32)
    np.testing.assert_allclose(hip.from_numpy(g).sum(axis=(0,), keepdims=True).reshape(512).numpy(), g.astype(np.float64).sum(0),
                               atol=2e-4)


def test_max_backward_ties_full_gradient(hip):
    a = np.array([[1, 3, 3], [2, 2, 2]], np.float32)
    t = hip.from_numpy(a)
    t.max(axis=1).backward(allow_fill=True)
    np.testing.assert_array_equal(t.grad.numpy(), [[0, 1, 1], [1, 1, 1]])


def test_oracle_matches_on_seeded_random_ops(hip):
    """HIP path vs the numpy oracle (forward + backward with a random upstream gradient)"""
    rng = np.random.RandomState(42)
    for name, (fwd, bwd) in sorted(O.UNARY.items()):
        x = rng.uniform(0.1, 2, (33, 65)).astype(np.float32)
        w = rng.uniform(-1, 1, x.shape).astype(np.float32)
        t = hip.from_numpy(x)
        y = getattr(t, name)()
        (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        yo = fwd(x)
        np.testing.assert_allclose(y.numpy(), yo, rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(t.grad.numpy(), bwd(w, yo, x)[0], rtol=1e-5, atol=1e-6, err_msg=name)
    for name, (fwd, bwd) in sorted(O.BINARY.items()):
        a = rng.uniform(0.5, 2, (33, 65)).astype(np.float32)
        b = rng.uniform(0.5, 2, (1, 65)).astype(np.float32)
        w = rng.uniform(-1, 1, a.shape).astype(np.float32)
        ta, tb = hip.from_numpy(a), hip.from_numpy(b)
        op = {"add": lambda p, q: p + q, "sub": lambda p, q: p - q, "mul": lambda p, q: p * q, "div": lambda p, q: p / q,
              "divfn": lambda p, q: hip.div(p, q), "pow": lambda p, q: p ** q}[name]
        y = op(ta, tb)
        (y * hip.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        yo = fwd(a, b)
        ga, gb = bwd(w, yo, a, b)
        np.testing.assert_allclose(y.numpy(), yo, rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(ta.grad.numpy(), O.unbroadcast(ga, a.shape), rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(tb.grad.numpy(), O.unbroadcast(gb, b.shape), rtol=2e-5, atol=2e-5, err_msg=name)
