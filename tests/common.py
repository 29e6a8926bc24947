"""Test helpers restating the reference's test/common.py (input-pair generator with broadcast and
transposed-view variants, value comparison, gradient check) for this repo's backends.

Deviation: the reference's broadcast variants are float64 arrays (common.py:25 has no astype); the
HipTensor backend computes in float32 only, so the variants are cast to `dtype` here."""
import numpy as np
from lightgrad_amd.autograd.tensor import AbstractTensor
from lightgrad_amd.autograd.utils.gradcheck import assert_gradcheck


def yield_input_pairs(cls, shapes, lowhigh=(-1, 1), dtype=np.float32, broadcast=False, transpose=False):
    assert len(lowhigh) == 2 and issubclass(cls, AbstractTensor)
    np_arrays = [np.random.uniform(*lowhigh, size=shape).astype(dtype) for shape in shapes]
    cls_arrays = [cls.from_numpy(arr) for arr in np_arrays]
    yield np_arrays, cls_arrays
    if broadcast:
        for i, shape in enumerate(shapes):
            for j in range(len(shape)):
                collapsed = shape[:j] + (1,) + shape[j + 1:]
                arr = np.random.uniform(*lowhigh, size=collapsed).astype(dtype)
                yield (np_arrays[:i] + [arr] + np_arrays[i + 1:],
                       cls_arrays[:i] + [cls.from_numpy(arr)] + cls_arrays[i + 1:])
    if transpose:
        for i, (np_array, cls_array, shape) in enumerate(zip(np_arrays, cls_arrays, shapes)):
            perm = list(reversed(range(len(shape))))
            yield (np_arrays[:i] + [np_array.transpose(*perm)] + np_arrays[i + 1:],
                   cls_arrays[:i] + [cls_array.transpose(*perm)] + cls_arrays[i + 1:])


def compare_with_numpy(cls, fn_or_name, shapes, lowhigh=(-1, 1), dtype=np.float32, broadcast=False, transpose=False,
                       rtol=1e-5, atol=1e-5, **kwargs):
    if isinstance(fn_or_name, str):
        np_fn, cls_fn = getattr(np, fn_or_name), getattr(cls, fn_or_name)
    else:
        np_fn, cls_fn = fn_or_name, fn_or_name
    for np_arrays, cls_arrays in yield_input_pairs(cls, shapes, lowhigh, dtype, broadcast, transpose):
        np_out = np_fn(*np_arrays, **kwargs)
        cls_out = cls_fn(*cls_arrays, **kwargs).numpy()
        assert np.asarray(np_out).shape == cls_out.shape, (np.asarray(np_out).shape, cls_out.shape)
        np.testing.assert_allclose(cls_out, np_out, rtol=rtol, atol=atol)


def compare_with_cpu(cls, fn_or_name, shapes, lowhigh=(-1, 1), dtype=np.float32, broadcast=False, transpose=False,
                     rtol=1e-5, atol=1e-6, **kwargs):
    from lightgrad_amd import CpuTensor
    if isinstance(fn_or_name, str):
        cpu_fn, cls_fn = getattr(CpuTensor, fn_or_name), getattr(cls, fn_or_name)
    else:
        cpu_fn, cls_fn = fn_or_name, fn_or_name
    for np_arrays, cls_arrays in yield_input_pairs(cls, shapes, lowhigh, dtype, broadcast, transpose):
        cpu_arrays = [CpuTensor.from_numpy(arr) for arr in np_arrays]
        cpu_out = cpu_fn(*cpu_arrays, **kwargs).numpy()
        cls_out = cls_fn(*cls_arrays, **kwargs).numpy()
        assert cpu_out.shape == cls_out.shape
        np.testing.assert_allclose(cls_out, cpu_out, rtol=rtol, atol=atol)


def check_gradients(cls, fn_or_name, shapes, lowhigh=(-1, 1), dtype=np.float32, broadcast=False, transpose=False,
                    eps=1e-3, tol=5e-4, **kwargs):
    fn = getattr(cls, fn_or_name) if isinstance(fn_or_name, str) else fn_or_name
    for _, cls_arrays in yield_input_pairs(cls, shapes, lowhigh, dtype, broadcast, transpose):
        for i, arr in enumerate(cls_arrays):
            f = lambda x: fn(*cls_arrays[:i], x, *cls_arrays[i + 1:], **kwargs)   # noqa: E731
            assert_gradcheck(f=f, x=arr, eps=eps, atol=tol, rtol=tol)


def replay_op_cases(T, golden, check):
    """Re-run every case of oracle/cases.py on tensor class T and hand (name, kind, got, expected) to `check`.
    The rng stream is re-drawn exactly as oracle/gen_golden.py did, which also verifies that the stored
    inputs are the ones the case table produces."""
    from cases import op_cases, f32
    rng = np.random.RandomState(20261003)
    n = 0
    for case in op_cases(T, rng):
        name, fn, inputs = case[0], case[1], case[2]
        upstream = case[3] if len(case) > 3 else True
        for i, a in enumerate(inputs):
            np.testing.assert_array_equal(a, golden["%s/in%d" % (name, i)], err_msg="case table / fixture drifted: " + name)
        ts = [T.from_numpy(a.copy()) for a in inputs]
        y = fn(*ts)
        expected = golden[name + "/out"]
        assert tuple(y.shape) == expected.shape, (name, y.shape, expected.shape)
        check(name, "out", y.numpy(), expected)
        if upstream:
            w = f32(rng, -1, 1, expected.shape)
            np.testing.assert_array_equal(w, golden[name + "/w"])
            (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
            for i, t in enumerate(ts):
                key = "%s/grad%d" % (name, i)
                if key in golden.files:
                    assert t.grad is not None, name
                    check(name, "grad%d" % i, t.grad.numpy(), golden[key])
        n += 1
    return n


class float64_tape(object):
    """`with float64_tape():` - CpuTensor builds float64 tensors, so the SAME tape code gives a yardstick in (nearly) exact
    arithmetic: a float32 result is then judged by its distance to that, not to another float32 result"""

    def __enter__(self):
        from lightgrad_amd import CpuTensor
        self._cls, self._saved = CpuTensor, CpuTensor.default_dtype
        CpuTensor.default_dtype = np.float64
        return self

    def __exit__(self, *exc):
        self._cls.default_dtype = self._saved
        return False


def rel_frobenius(got, ref):
    """||got - ref|| / ||ref|| in float64"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-300))


# one encoder layer of BERT at sizes the fused attention kernels take (d = 32, 32 positions): tests/dist_rank_worker.py `bert` mode
DIST_BERT_CFG = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=60,
                     max_position_embeddings=32, type_vocab_size=2)


# ---- the float64 yardstick for multi-step results (Weak 1) ------------------------------------------------------
# A k-step trajectory amplifies float32 rounding noise (AdaBelief divides by the square root of a second moment that starts
# near zero), so two correct float32 runs end 1e-4 apart and say little about each other.  The SAME tape code run on CpuTensor
# in float64 is (nearly) exact: a float32 result is judged by its distance to that - it must be within the north star's 1e-5
# (relative Frobenius) or no further away than twice what the float32 CPU backend itself is.
def mlp_trajectory_on_cpu(w0, x, onehot, steps, make_optimizer, dtype):
    """(losses, {name: final weights}) of `steps` training steps of the MLP on CpuTensor in `dtype`; make_optimizer(parameters)"""
    import lightgrad_amd as light
    from lightgrad_amd import CpuTensor
    from test_cpu_backend import MLP
    d_in, d_hid, d_out = w0["l1.weight"].shape[1], w0["l1.weight"].shape[0], w0["l2.weight"].shape[0]

    def run():
        model = MLP(d_in, d_hid, d_out)
        model.load_parameters({n: a.astype(dtype) for n, a in w0.items()})
        assert all(p.dtype == dtype for p in model.parameters())
        opt = make_optimizer(model.parameters())
        tx, tt = CpuTensor.from_numpy(x.astype(dtype)), CpuTensor.from_numpy(onehot.astype(dtype))
        losses = []
        for _ in range(steps):
            l = light.loss.mse(model(tx), tt)
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(l.item())
        return losses, {n: p.numpy().copy() for n, p in model.named_parameters()}
    if np.dtype(dtype) == np.float64:
        with float64_tape():
            return run()
    return run()


def assert_as_close_to_float64_as_the_cpu_backend(got: dict, cpu32: dict, ref64: dict, floor=1e-5, what=""):
    """every array of `got` within `floor` (relative Frobenius) of the float64 result, or no further from it than twice the float32
    CPU backend's own distance"""
    for n in ref64:
        e_got, e_cpu = rel_frobenius(got[n], ref64[n]), rel_frobenius(cpu32[n], ref64[n])
        assert e_got <= max(floor, 2 * e_cpu), "%s %s: %.3g from the float64 run, the float32 CPU backend %.3g" % (what, n, e_got, e_cpu)
