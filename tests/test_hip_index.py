"""Integer-array indexing on the device (csrc/index.hip) against the reference's fixtures - the same checks as
tests/test_fancy_index_cpu.py, run on HipTensor - plus the UNFUSED tape form of loss.cross_entropy
(loss.py:14-24: softmax, `y[range(n), labels]`, `-=`, `/=`) on the device, any dtype, views, and how an index out
of range surfaces (IndexError at the next synchronising call; at once for host-side indices)."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from conftest import load_golden
from test_fancy_index_cpu import APART_CASES, check_apart, check_apart_inplace, TAKE_CASES, PAIR_CASES, MULTI_CASES, check_take, check_inplace, check_dataset, check_multi, check_multi_inplace

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(TAKE_CASES))
def test_take_one_axis(hip, name):
    check_take(hip, load_golden("fancy_index.npz"), name, TAKE_CASES[name], "idx")


@pytest.mark.parametrize("name", PAIR_CASES)
def test_pair_rows_labels(hip, name):
    g = load_golden("fancy_index.npz")
    n = g[name + "/in"].shape[0]
    check_take(hip, g, name, lambda t, lab: t[range(n), lab], "labels")


def test_inplace_forms(hip):
    check_inplace(hip, load_golden("fancy_index.npz"))


@pytest.mark.parametrize("name", sorted(MULTI_CASES))
def test_several_index_arrays_and_masks(hip, name):
    """two / three index arrays on neighbouring axes (broadcast together) and boolean masks, as the reference's CPU path takes them
    through numpy (cpu/ops.py:234-255): values, gradients and in-place forms bit for bit against fixtures recorded from it"""
    check_multi(hip, load_golden("fancy_index_multi.npz"), name)


def test_several_index_arrays_and_masks_in_place(hip):
    check_multi_inplace(hip, load_golden("fancy_index_multi.npz"))


def test_index_forms_that_are_errors(hip):
    t = hip.from_numpy(np.zeros((4, 5, 6), np.float32))
    with pytest.raises(IndexError, match="out of bounds"):
        t[np.asarray([0, 1]), np.asarray([1, 5])]
    with pytest.raises(IndexError, match="broadcast"):
        t[np.asarray([0, 1, 2]), np.asarray([1, 2])]
    with pytest.raises(IndexError, match="boolean index did not match"):
        t[np.zeros((4, 4), bool)]
    assert t[np.asarray([0, 1]), :, np.asarray([1, 2])].shape == (2, 5)         # index arrays apart: their dimension goes first
    with pytest.raises(IndexError, match="newaxis"):
        t[np.asarray([0, 1]), None, np.asarray([1, 2])]


@pytest.mark.parametrize("name", sorted(APART_CASES))
def test_index_arrays_apart(hip, name):
    """index arrays on non-neighbouring axes (numpy moves the index dimensions to the front), plain integers among index arrays, a
    mask next to an array - values, shapes and gradients against fixtures recorded from the reference's CPU path; the index arrays
    as device tensors (folded on the device, no read-back) and as host arrays"""
    check_apart(hip, load_golden("fancy_index_apart.npz"), name)


def test_index_arrays_apart_in_place(hip):
    check_apart_inplace(hip, load_golden("fancy_index_apart.npz"))


def test_device_indices_are_not_read_back(hip, monkeypatch):
    """several device-resident index arrays are folded by a kernel: nothing but a boolean mask's COUNT may come back to the host"""
    rng = np.random.RandomState(8)
    a = rng.uniform(-1, 1, (6, 5, 7)).astype(np.float32)
    t = hip.from_numpy(a)
    i0, i1 = rng.randint(0, 6, (4, 3)), rng.randint(-7, 7, (3,))
    d0, d1 = hip.from_numpy(i0.astype(np.int32), requires_grad=False), hip.from_numpy(i1.astype(np.int64), requires_grad=False)
    mask = rng.uniform(0, 1, (6, 5)) > 0.5
    dm = hip.from_numpy(mask, requires_grad=False)
    reads = []
    real = hip.numpy
    monkeypatch.setattr(hip, "numpy", lambda self: (reads.append(self.shape), real(self))[1])
    y = t[d0, :, d1]
    z = t[dm]
    monkeypatch.setattr(hip, "numpy", real)
    assert reads == [(1,)], reads                     # the mask's count, nothing else
    np.testing.assert_array_equal(y.numpy(), a[i0, :, i1])
    np.testing.assert_array_equal(z.numpy(), a[mask])
    bad = hip.from_numpy(np.asarray([0, 6]), requires_grad=False)          # out of range on the device: reported at the next sync
    out = t[bad, :, hip.from_numpy(np.asarray([1, 2]), requires_grad=False)]
    with pytest.raises(IndexError):
        out.numpy()
    big = rng.uniform(0, 1, (300, 41)) > 0.7                              # a mask over several scan blocks
    src = rng.uniform(-1, 1, (300, 41, 2)).astype(np.float32)
    np.testing.assert_array_equal(hip.from_numpy(src)[hip.from_numpy(big, requires_grad=False)].numpy(), src[big])
    none = np.zeros((300, 41), bool)
    assert hip.from_numpy(src)[hip.from_numpy(none, requires_grad=False)].shape == (0, 2)


def test_dataset_epoch_matches_reference(hip):
    check_dataset(hip, load_golden("fancy_index.npz"))


@pytest.mark.parametrize("name", ["n8_c10_i64", "n5_c3_i32", "n33_c130_i16", "n1_c1000_i64"])
def test_unfused_cross_entropy_tape_on_device(hip, name, monkeypatch):
    """the generic expression of loss.cross_entropy, with the fused hook switched off: device softmax, the (range, labels)
    gather, log, mean; backward with the in-place `p[range(n), labels] -= 1; p /= n` - against the reference's fixture"""
    monkeypatch.setattr(hip, "_fused_cross_entropy", None)
    g = load_golden("cross_entropy.npz")
    y = hip.from_numpy(g[name + "/logits"].copy())
    labels = hip.from_numpy(g[name + "/labels"], requires_grad=False)
    loss = light.loss.cross_entropy(y, labels)
    (loss * hip.from_numpy(g[name + "/w"], requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_allclose(loss.numpy(), g[name + "/loss"], rtol=1e-5)
    np.testing.assert_allclose(y.grad.numpy(), g[name + "/grad"], rtol=1e-5, atol=1e-7)


def test_any_dtype_views_and_mixed_basic_indices(hip):
    rng = np.random.RandomState(3)
    for dtype in (np.uint8, np.int16, np.int32, np.float32, np.int64, np.float64):
        a = (rng.uniform(-100, 100, (5, 6, 7))).astype(dtype)
        t = hip.from_numpy(a, requires_grad=False)
        i = np.asarray([4, 0, 0, -1, 2])
        np.testing.assert_array_equal(t[i].numpy(), a[i])
        np.testing.assert_array_equal(t[:, i].numpy(), a[:, i])
        np.testing.assert_array_equal(t[..., i].numpy(), a[..., i])
        np.testing.assert_array_equal(t[1:4, :, hip.from_numpy(i.astype(np.int32), requires_grad=False)].numpy(), a[1:4, :, i])
        np.testing.assert_array_equal(t[2, i].numpy(), a[2, i])                       # an int in front drops a dimension
        np.testing.assert_array_equal(t[None, :, [1, 3]].numpy(), a[None, :, [1, 3]])
        np.testing.assert_array_equal(t.transpose(2, 0, 1)[i].numpy(), a.transpose(2, 0, 1)[i])      # gather from a strided view
        np.testing.assert_array_equal(t[range(5), [5, 0, 1, 1, 3]].numpy(), a[range(5), [5, 0, 1, 1, 3]])
        idx2 = rng.randint(-5, 5, (2, 3, 2))
        np.testing.assert_array_equal(t[idx2].numpy(), a[idx2])                       # an index of any shape
    # assignment through a view that is not dense, scalar and tensor values
    a = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    t = hip.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[1:5, [0, 7, 3]] = 9.0
        t[[5, 0]] = hip.from_numpy(np.arange(16, dtype=np.float32).reshape(2, 8), requires_grad=False)
        t.transpose(1, 0)[[2, 4], 1:3] = np.asarray([[1.0, 2.0], [3.0, 4.0]], np.float32)
    a[1:5, [0, 7, 3]] = 9.0
    a[[5, 0]] = np.arange(16, dtype=np.float32).reshape(2, 8)
    a.T[[2, 4], 1:3] = [[1.0, 2.0], [3.0, 4.0]]
    np.testing.assert_array_equal(t.numpy(), a)


def test_gradient_through_index_on_a_view_and_repeated_indices(hip):
    rng = np.random.RandomState(9)
    an = rng.uniform(-1, 1, (5, 4, 3)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, 3, 3)).astype(np.float32)
    res = {}
    for cls in (CpuTensor, hip):
        t = cls.from_numpy(an.copy())
        y = t[1:3, np.asarray([0, 0, 3])]                                             # basic slice + repeated index
        (y * cls.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        res[cls] = (y.numpy(), t.grad.numpy())
    np.testing.assert_array_equal(res[hip][0], res[CpuTensor][0])
    np.testing.assert_allclose(res[hip][1], res[CpuTensor][1], rtol=1e-6)


def test_index_out_of_range_is_reported(hip):
    from lightgrad_amd.autograd.hip import HipDevice
    t = hip.from_numpy(np.arange(12, dtype=np.float32).reshape(4, 3), requires_grad=False)
    with pytest.raises(IndexError):
        t[[0, 4]]                                                                     # host-side index: at once, like numpy
    np.testing.assert_array_equal(t[[0, 1], [0, 1]].numpy(), [0., 4.])                # pairs of index arrays: one flat index, host-built
    bad = hip.from_numpy(np.asarray([1, 7], np.int64), requires_grad=False)
    out = t[bad]                                                                      # device-side index: the kernel cannot raise ...
    with pytest.raises(IndexError):
        out.numpy()                                                                   # ... the next synchronising call does
    HipDevice.synchronize()                                                           # reported once
    np.testing.assert_array_equal(out.numpy()[0], [3, 4, 5])
    assert np.isnan(out.numpy()[1]).all()
    # labels out of range in the fused cross-entropy: NaN loss + the same report (the reference raises IndexError, loss.py:19)
    logits = hip.from_numpy(np.zeros((3, 5), np.float32))
    loss = light.loss.cross_entropy(logits, hip.from_numpy(np.asarray([0, 5, 1], np.int64), requires_grad=False))
    with pytest.raises(IndexError):
        loss.item()
    assert np.isnan(loss.item())
