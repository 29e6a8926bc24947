"""Integer-array indexing and Dataset batching on the CPU backend against fixtures recorded from the reference
(tests/golden/fancy_index.npz, written by oracle/gen_golden.py): one index array on any axis, the (range, labels)
pair of loss.cross_entropy, in-place forms, a seeded Dataset epoch."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from conftest import load_golden

TAKE_CASES = {
    "take_axis0": lambda t, i: t[i], "take_axis1": lambda t, i: t[:, i], "take_axis0_2d_index": lambda t, i: t[i],
    "take_middle_axis": lambda t, i: t[:, i, :], "take_negative": lambda t, i: t[i],
}
PAIR_CASES = ["pair_rows_labels", "pair_rows_labels_i16"]


def check_take(cls, g, name, index_fn, key):
    t = cls.from_numpy(g[name + "/in"].copy())
    idx = cls.from_numpy(g[name + "/" + key], requires_grad=False)
    y = index_fn(t, idx)
    np.testing.assert_array_equal(y.numpy(), g[name + "/out"])                       # index ops are bit-exact
    (y * cls.from_numpy(g[name + "/w"], requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_array_equal(t.grad.numpy(), g[name + "/grad"])
    # the same index given as a host array / list
    y2 = index_fn(cls.from_numpy(g[name + "/in"].copy()), g[name + "/" + key])
    np.testing.assert_array_equal(y2.numpy(), g[name + "/out"])


@pytest.mark.parametrize("name", sorted(TAKE_CASES))
def test_take_one_axis(name):
    check_take(CpuTensor, load_golden("fancy_index.npz"), name, TAKE_CASES[name], "idx")


@pytest.mark.parametrize("name", PAIR_CASES)
def test_pair_rows_labels(name):
    g = load_golden("fancy_index.npz")
    n = g[name + "/in"].shape[0]
    check_take(CpuTensor, g, name, lambda t, lab: t[range(n), lab], "labels")


def check_inplace(cls, g):
    t = cls.from_numpy(g["pair_isub/in"].copy(), requires_grad=False)
    with light.no_grad():
        t[range(8), cls.from_numpy(g["pair_isub/labels"], requires_grad=False)] -= 1
    np.testing.assert_array_equal(t.numpy(), g["pair_isub/out"])
    t = cls.from_numpy(g["put_axis0/in"].copy(), requires_grad=False)
    with light.no_grad():
        t[cls.from_numpy(g["put_axis0/idx"], requires_grad=False)] = cls.from_numpy(g["put_axis0/val"], requires_grad=False)
    np.testing.assert_array_equal(t.numpy(), g["put_axis0/out"])


def test_inplace_forms():
    check_inplace(CpuTensor, load_golden("fancy_index.npz"))


def check_dataset(cls, g):
    np.random.seed(int(g["dataset/seed"]))
    ds = light.data.Dataset((cls.from_numpy(g["dataset/X"], requires_grad=False), cls.from_numpy(g["dataset/Y"], requires_grad=False)),
                            shuffle=True, batchsize=8)
    assert ds.n == 20 and len(ds) == int(g["dataset/n_batches"]) == 3
    batches = list(ds)
    assert len(batches) == 3
    for k, (bx, by) in enumerate(batches):
        assert isinstance(bx, cls) and isinstance(by, cls) and by.dtype == np.int16
        np.testing.assert_array_equal(bx.numpy(), g["dataset/x%d" % k])
        np.testing.assert_array_equal(by.numpy(), g["dataset/y%d" % k])
    assert batches[-1][0].shape == (4, 3, 2)                                          # ragged last batch
    # no shuffling: batches are plain slices
    plain = light.data.Dataset((cls.from_numpy(g["dataset/X"], requires_grad=False),), shuffle=False, batchsize=7)
    np.testing.assert_array_equal(np.concatenate([b[0].numpy() for b in plain]), g["dataset/X"])


def test_dataset_epoch_matches_reference():
    check_dataset(CpuTensor, load_golden("fancy_index.npz"))


def test_repeated_indices_accumulate_in_backward():
    """documented divergence: the reference's `grad[idx] = out_grad` (cpu/ops.py:245) keeps the LAST of repeated rows;
    the gradient of a gather sums them (embedding semantics)"""
    t = CpuTensor.from_numpy(np.zeros((4, 2), np.float32))
    y = t[np.asarray([1, 1, 3])]
    (y * CpuTensor.from_numpy(np.asarray([[1, 2], [10, 20], [5, 5]], np.float32), requires_grad=False)).backward(allow_fill=True)
    np.testing.assert_array_equal(t.grad.numpy(), [[0, 0], [11, 22], [0, 0], [5, 5]])


# ---- several index arrays on neighbouring axes, boolean masks (tests/golden/fancy_index_multi.npz, recorded from the reference) ----
MULTI_CASES = {
    "two_arrays": lambda D, g, n: (D(g[n + "/i0"]), D(g[n + "/i1"])),
    "three_arrays": lambda D, g, n: (D(g[n + "/i0"]), D(g[n + "/i1"]), D(g[n + "/i2"])),
    "two_arrays_broadcast": lambda D, g, n: (D(g[n + "/i0"]), D(g[n + "/i1"])),
    "two_arrays_middle": lambda D, g, n: (slice(None), D(g[n + "/i0"]), D(g[n + "/i1"])),
    "mask_axis0": lambda D, g, n: D(g[n + "/mask"]),
    "mask_full": lambda D, g, n: D(g[n + "/mask"]),
    "mask_two_axes_of_three": lambda D, g, n: D(g[n + "/mask"]),
}


def check_multi(cls, g, name):
    for as_tensor in (True, False):                 # index arrays as backend tensors, and as host arrays
        D = (lambda v: cls.from_numpy(v, requires_grad=False)) if as_tensor else (lambda v: v)
        t = cls.from_numpy(g[name + "/in"].copy())
        y = t[MULTI_CASES[name](D, g, name)]
        np.testing.assert_array_equal(y.numpy(), g[name + "/out"])
        (y * cls.from_numpy(g[name + "/w"], requires_grad=False)).backward(allow_fill=True)
        np.testing.assert_array_equal(t.grad.numpy(), g[name + "/grad"])


def check_multi_inplace(cls, g):
    D = lambda v: cls.from_numpy(v, requires_grad=False)          # noqa: E731
    t = cls.from_numpy(g["put_two_arrays/in"].copy(), requires_grad=False)
    with light.no_grad():
        t[D(g["put_two_arrays/i0"]), D(g["put_two_arrays/i1"])] = D(g["put_two_arrays/val"])
    np.testing.assert_array_equal(t.numpy(), g["put_two_arrays/out"])
    t = cls.from_numpy(g["put_mask_scalar/in"].copy(), requires_grad=False)
    with light.no_grad():
        t[D(g["put_mask_scalar/mask"])] = 0.25
    np.testing.assert_array_equal(t.numpy(), g["put_mask_scalar/out"])


@pytest.mark.parametrize("name", sorted(MULTI_CASES))
def test_several_index_arrays_and_masks(name):
    check_multi(CpuTensor, load_golden("fancy_index_multi.npz"), name)


def test_several_index_arrays_and_masks_in_place():
    check_multi_inplace(CpuTensor, load_golden("fancy_index_multi.npz"))


# ---- index arrays on NON-neighbouring axes, ints among them, a mask next to an array (tests/golden/fancy_index_apart.npz) ----
APART_CASES = {
    "arrays_apart": lambda D, g, n: (D(g[n + "/i0"]), slice(None), D(g[n + "/i1"])),
    "arrays_apart_4d": lambda D, g, n: (D(g[n + "/i0"]), slice(1, 3), D(g[n + "/i1"])),
    "int_and_array_apart": lambda D, g, n: (2, slice(None), D(g[n + "/i0"])),
    "int_next_to_array": lambda D, g, n: (slice(None), 3, D(g[n + "/i0"])),
    "array_int_array": lambda D, g, n: (D(g[n + "/i0"]), -2, D(g[n + "/i1"])),
    "arrays_apart_broadcast": lambda D, g, n: (D(g[n + "/i0"]), slice(None, None, 2), D(g[n + "/i1"])),
    "mask_and_array": lambda D, g, n: (slice(None), D(g[n + "/mask"]), D(g[n + "/i0"])),
    "mask_apart_from_array": lambda D, g, n: (D(g[n + "/mask"]), slice(None), D(g[n + "/i0"])),
}


def check_apart(cls, g, name):
    for as_tensor in (True, False):
        D = (lambda v: cls.from_numpy(v, requires_grad=False)) if as_tensor else (lambda v: v)
        t = cls.from_numpy(g[name + "/in"].copy())
        y = t[APART_CASES[name](D, g, name)]
        assert y.shape == g[name + "/out"].shape, (name, y.shape, g[name + "/out"].shape)
        np.testing.assert_array_equal(y.numpy(), g[name + "/out"])
        (y * cls.from_numpy(g[name + "/w"], requires_grad=False)).backward(allow_fill=True)
        np.testing.assert_array_equal(t.grad.numpy(), g[name + "/grad"])


def check_apart_inplace(cls, g):
    D = lambda v: cls.from_numpy(v, requires_grad=False)          # noqa: E731
    t = cls.from_numpy(g["put_arrays_apart/in"].copy(), requires_grad=False)
    with light.no_grad():
        t[D(g["put_arrays_apart/i0"]), :, D(g["put_arrays_apart/i1"])] = D(g["put_arrays_apart/val"])
    np.testing.assert_array_equal(t.numpy(), g["put_arrays_apart/out"])


@pytest.mark.parametrize("name", sorted(APART_CASES))
def test_index_arrays_apart(name):
    check_apart(CpuTensor, load_golden("fancy_index_apart.npz"), name)


def test_index_arrays_apart_in_place():
    check_apart_inplace(CpuTensor, load_golden("fancy_index_apart.npz"))
