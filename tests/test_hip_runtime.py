"""Runtime behaviour of the HIP backend: the caching pool, 64-bit indexing beyond 2^31 elements, unaligned /
offset views through every kernel family, initialisers, the profiler hook and C-ABI argument checking."""
import ctypes
import gc
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor

pytestmark = pytest.mark.gpu


def test_pool_reuses_and_trims(hip):
    from lightgrad_amd.autograd.hip import HipDevice
    gc.collect()
    HipDevice.trim_pool()
    s0 = HipDevice.pool_stats()
    a = hip.zeros((1024, 1024))
    s1 = HipDevice.pool_stats()
    assert s1["in_use_bytes"] - s0["in_use_bytes"] >= 4 << 20
    ptr = a.ptr
    del a
    gc.collect()
    s2 = HipDevice.pool_stats()
    assert s2["in_use_bytes"] == s0["in_use_bytes"] and s2["reserved_bytes"] == s1["reserved_bytes"]   # cached, not freed
    b = hip.zeros((1024, 1024))
    assert b.ptr == ptr and HipDevice.pool_stats()["hip_malloc_calls"] == s1["hip_malloc_calls"]        # reused: no hipMalloc
    del b
    gc.collect()
    HipDevice.trim_pool()
    assert HipDevice.pool_stats()["reserved_bytes"] <= s0["reserved_bytes"]
    info = HipDevice.info()
    assert info["compute_units"] == 256 and info["wavefront_size"] == 64 and info["hbm_bytes"] > 200e9


def test_more_than_2_31_elements(hip):
    """the 64-bit index paths of the elementwise, fill, copy and reduction kernels (8.6 GB per tensor)"""
    n = (1 << 31) + 4096
    t = hip.empty((n,), requires_grad=False)
    t.fill(0.5)
    u = t * 4.0                                                    # flat path
    assert u.max().item() == 2.0 and u.min().item() == 2.0
    np.testing.assert_allclose(u.sum().item(), 2.0 * n, rtol=1e-6)
    tail = u[n - 5:].numpy()
    np.testing.assert_array_equal(tail, np.full(5, 2.0, np.float32))
    v = u.reshape(2, n // 2)
    w = (v.transpose(1, 0)[::2] + 1.0)                             # gather path with 64-bit offsets
    assert w.shape == (n // 4, 2)
    np.testing.assert_array_equal(w[n // 4 - 3:].numpy(), np.full((3, 2), 3.0, np.float32))
    np.testing.assert_array_equal(v.sum(axis=0)[-4:].numpy(), np.full(4, 4.0, np.float32))
    del t, u, v, w
    gc.collect()
    from lightgrad_amd.autograd.hip import HipDevice
    HipDevice.trim_pool()


def test_offset_and_unaligned_views_everywhere(hip):
    rng = np.random.RandomState(0)
    a = rng.uniform(-1, 1, (131, 67)).astype(np.float32)
    b = rng.uniform(-1, 1, (67, 45)).astype(np.float32)
    ta, tb = hip.from_numpy(a), hip.from_numpy(b)
    va, vb = ta[1:, 3:], tb[3:, 1:]                                 # 4-byte aligned only, odd leading dimensions
    sa, sb = a[1:, 3:], b[3:, 1:]
    np.testing.assert_array_equal((va * 2.0 + va).numpy(), sa * 2.0 + sa)
    np.testing.assert_allclose(va.sum(axis=0).numpy(), sa.astype(np.float64).sum(0), atol=1e-4)
    np.testing.assert_array_equal(va.max(axis=1).numpy(), sa.max(1))
    np.testing.assert_allclose((va @ vb).numpy(), sa.astype(np.float64) @ sb, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose((va.transpose(1, 0)[:, ::2] @ va[::2]).numpy(), sa.T[:, ::2].astype(np.float64) @ sa[::2], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(va.softmax(axis=-1).numpy(), CpuTensor.from_numpy(np.ascontiguousarray(sa)).softmax(axis=-1).numpy(), rtol=1e-5, atol=1e-7)
    z = hip.zeros((131, 67))
    z[1:, 3:] = va
    e = np.zeros((131, 67), np.float32)
    e[1:, 3:] = sa
    np.testing.assert_array_equal(z.numpy(), e)


def test_initialisers_match_cpu_backend_for_the_same_seed(hip):
    np.random.seed(7)
    c = CpuTensor.xavier((33, 17)).numpy()
    cu = CpuTensor.uniform(-2, 3, (5, 6)).numpy()
    np.random.seed(7)
    h = hip.xavier((33, 17))
    hu = hip.uniform(-2, 3, (5, 6))
    assert h.ctx is None and h.requires_grad
    np.testing.assert_array_equal(h.numpy(), c)                    # same RNG stream, exact division by sqrt(numel)
    np.testing.assert_array_equal(hu.numpy(), cu)
    np.testing.assert_array_equal(hip.ones((2, 3)).numpy(), np.ones((2, 3), np.float32))
    np.testing.assert_array_equal(CpuTensor.from_numpy(c).hip().cpu().numpy(), c)


def test_profiler_sees_hip_ops(hip):
    from lightgrad_amd.autograd.utils.profiler import Profiler
    a = hip.uniform(-1, 1, (64, 64))
    with Profiler() as p:
        (a @ a).relu().sum().backward()
    t = p.table()
    assert t["dot"][1] == 1 and t["dot"][3] == 1 and t["relu"][3] == 1 and t["sum"][3] == 1


def test_device_profiler_reports_kernel_time(hip):
    from lightgrad_amd.autograd.hip import HipProfiler
    a = hip.uniform(-1, 1, (2048, 2048))
    with HipProfiler() as p:
        for _ in range(3):
            (a @ a).relu().sum().backward()
    assert p.table()["dot"][1] == 3
    assert p.device_ms[False]["dot"] > 0.05 and p.device_ms[True]["dot"] > p.device_ms[False]["dot"]   # 2 GEMMs vs 1
    assert p.device_ms[False]["dot"] > 10 * p.device_ms[False]["relu"] * 0 + 0.0
    p.print()


def test_c_abi_argument_checks(hip):
    from lightgrad_amd.autograd.hip import lib as L
    lib = L.lib()
    t = hip.zeros((4, 4))
    nine = L.i64((1,) * 9)
    assert lib.lg_ew(L.EW_NEG, 9, nine, t.ptr, nine, None, None, t.ptr, nine, None, None, None, None, None, None, 0.0) == -1
    assert b"ndim" in lib.lg_last_error()
    sh, st, z = L.i64((4, 4)), L.i64((4, 1)), L.i64((0, 1))
    assert lib.lg_ew(L.EW_NEG, 2, sh, t.ptr, z, None, None, t.ptr, st, None, None, None, None, None, None, 0.0) == -1   # broadcast output
    assert b"zero stride" in lib.lg_last_error()
    assert lib.lg_ew(L.EW_ADD, 2, sh, t.ptr, st, None, None, None, None, None, None, None, None, None, None, 0.0) == -1  # no tensor operand
    assert lib.lg_reduce(7, 2, sh, t.ptr, st, 1, t.ptr) == -1
    assert lib.lg_reduce(0, 2, sh, t.ptr, st, 4, t.ptr) == -1                                                     # axis >= ndim
    assert lib.lg_copy_strided(3, 2, sh, t.ptr, st, t.ptr, st) == -1                                              # itemsize 3
    assert lib.lg_softmax_f32(t.ptr, t.ptr, 4, 0) == -1
    assert lib.lg_gather_rows_f32(t.ptr, t.ptr, 2, t.ptr, 1, 4, 4) == -1                                          # int16 ids
    assert lib.lg_adam_multi_dev_f32(t.ptr, t.ptr, t.ptr, t.ptr, 0, L.i64((0,)), 1e-3, .9, .999, 1e-8, t.ptr, 0, 1.0, 1) == -1        # no segments
    assert lib.lg_adam_multi_dev_f32(t.ptr, t.ptr, t.ptr, t.ptr, 65, L.i64((0,) * 66), 1e-3, .9, .999, 1e-8, t.ptr, 0, 1.0, 1) == 0    # 65 EMPTY segments: two groups, nothing to do
    assert lib.lg_adam_multi_dev_f32(t.ptr, t.ptr, t.ptr, t.ptr, 2, L.i64((0, 8, 4)), 1e-3, .9, .999, 1e-8, t.ptr, 0, 1.0, 1) == -1    # decreasing offsets
    ev = ctypes.c_void_p()
    assert lib.lg_event_create(ctypes.byref(ev)) == 0 and lib.lg_event_record(ev) == 0 and lib.lg_event_destroy(ev) == 0
    np.testing.assert_array_equal(t.numpy(), np.zeros((4, 4), np.float32))                                        # nothing was written


def test_upload_between_graph_replays(hip):
    """HipTensor.upload_ (lg_memcpy_h2d_async): new batches fed into the static inputs of a captured step between replays; the
    source array may be reused at once"""
    from lightgrad_amd.autograd.hip import HipGraph
    rng = np.random.RandomState(0)
    x, t = hip.zeros((64, 32), requires_grad=False), hip.zeros((64, 1), requires_grad=False)
    g = HipGraph()
    (x * t).sum()                                                  # warm-up
    with g.capture():
        res = (x * t).sum()
    scratch_x, scratch_t = np.empty((64, 32), np.float32), np.empty((64, 1), np.float32)
    for _ in range(6):
        bx, bt = rng.uniform(-1, 1, (64, 32)).astype(np.float32), rng.uniform(-1, 1, (64, 1)).astype(np.float32)
        scratch_x[...], scratch_t[...] = bx, bt
        x.upload_(scratch_x)
        t.upload_(scratch_t)
        scratch_x[...] = 7.0                                       # staged: the upload no longer reads the source
        g.replay()
        np.testing.assert_allclose(res.item(), float((bx.astype(np.float64) * bt).sum()), rtol=1e-4, atol=1e-4)
    g.destroy()
