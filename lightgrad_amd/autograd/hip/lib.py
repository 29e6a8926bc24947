"""ctypes binding of liblghip.so (C ABI: include/lghip.h).

This is the whole host<->device boundary of the HipTensor backend - the place
where the reference's OpenCL backend calls pyopencl (opencl/tensor.py:64-92,
opencl/kernels.py:186-194, :330-334, :489-499).  There is NO fallback: if the
library has not been built, or no MI355X is visible, using a HipTensor raises.
"""
import ctypes
import os
from ctypes import c_int, c_int64, c_uint32, c_uint64, c_size_t, c_float, c_double, c_void_p, c_char_p, POINTER

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get("LIGHTGRAD_HIP_LIB") or os.path.join(_PKG_DIR, "liblghip.so")   # override: kernel experiments
COMM_LIB_PATH = os.path.join(_PKG_DIR, "liblghip_comm.so")

# lg_ew op ids (lghip.h: lg_ew_op_t)
ACT_GELU, ACT_GELU_BWD = 1, 2       # lg_gemm_act_f32
EW_COPY, EW_NEG, EW_EXP, EW_LOG, EW_RELU, EW_SIGMOID, EW_TANH, EW_SIN, EW_COS, EW_SQRT, EW_GELU = range(11)
(EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_POW, EW_RELU_BWD, EW_SIGMOID_BWD, EW_TANH_BWD, EW_LOG_BWD,
 EW_SIN_BWD, EW_COS_BWD, EW_EQ, EW_GE, EW_BIAS_RELU, EW_GELU_BWD) = range(32, 47)
EW_MAX_BWD, EW_FMA = 64, 65
EW_MUL_BWD, EW_DIV_BWD, EW_POW_BWD = 96, 97, 98
RED_SUM, RED_MAX, RED_MIN = 0, 1, 2
# lg_dtype_t (lghip.h): the dtypes lg_ew_typed / lg_reduce_typed / lg_cast know
DT_I16, DT_I32, DT_I64, DT_F64, DT_F32 = 1, 2, 3, 4, 5


class DeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("arch", ctypes.c_char * 32),
                ("compute_units", ctypes.c_int32), ("clock_mhz", ctypes.c_int32),
                ("wavefront_size", ctypes.c_int32), ("lds_bytes_per_cu", ctypes.c_int32),
                ("hbm_bytes", ctypes.c_uint64), ("l2_bytes", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_I64P = POINTER(c_int64)

# name -> (restype, argtypes); must list every symbol include/lghip.h declares (tests check this)
PROTOTYPES = {
    "lg_last_error": (c_char_p, []),
    "lg_version": (c_char_p, []),
    "lg_device_count": (c_int, [POINTER(c_int)]),
    "lg_init": (c_int, [c_int]),
    "lg_device": (c_int, [POINTER(c_int)]),
    "lg_device_info": (c_int, [POINTER(DeviceInfo)]),
    "lg_peer_info": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "lg_stream": (c_void_p, []),
    "lg_sync": (c_int, []),
    "lg_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "lg_free": (c_int, [c_void_p]),
    "lg_pool_trim": (c_int, []),
    "lg_pool_stats": (c_int, [POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    "lg_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lg_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lg_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lg_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lg_event_create": (c_int, [POINTER(c_void_p)]),
    "lg_event_record": (c_int, [c_void_p]),
    "lg_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "lg_event_destroy": (c_int, [c_void_p]),
    "lg_graph_begin": (c_int, []),
    "lg_graph_end": (c_int, [POINTER(c_void_p)]),
    "lg_graph_launch": (c_int, [c_void_p]),
    "lg_graph_destroy": (c_int, [c_void_p]),
    "lg_graph_kernel_count": (c_int, [c_void_p, POINTER(c_int)]),
    "lg_copy_strided": (c_int, [c_int, c_int, _I64P, c_void_p, _I64P, c_void_p, _I64P]),
    "lg_fill_strided": (c_int, [c_int, c_int, _I64P, c_void_p, _I64P, c_uint64]),
    "lg_ew": (c_int, [c_int, c_int, _I64P, c_void_p, _I64P, c_void_p, _I64P,
                      c_void_p, _I64P, c_void_p, _I64P, c_void_p, _I64P, c_void_p, _I64P, c_float]),
    "lg_reduce": (c_int, [c_int, c_int, _I64P, c_void_p, _I64P, c_uint32, c_void_p]),
    "lg_ew_typed": (c_int, [c_int, c_int, c_int, _I64P, c_void_p, _I64P, c_void_p, _I64P, c_void_p, _I64P, c_double, c_int64]),
    "lg_reduce_typed": (c_int, [c_int, c_int, c_int, _I64P, c_void_p, _I64P, c_uint32, c_void_p]),
    "lg_cast": (c_int, [c_int, c_int, c_int, _I64P, c_void_p, _I64P, c_void_p, _I64P]),
    "lg_reduce_acc": (c_int, [c_int, c_int, _I64P, c_void_p, _I64P, c_uint32, c_void_p, c_int]),
    "lg_gemm_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                            c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int]),
    "lg_gemm_pair_begin": (c_int, []),
    "lg_gemm_pair_end": (c_int, []),
    "lg_gemm_pair_mse_loss": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "lg_gemm_pair_hold": (c_int, []),
    "lg_gemm_pair_resume": (c_int, []),
    "lg_gemm_group_begin": (c_int, []),
    "lg_gemm_group_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int]),
    "lg_gemm_group_flush": (c_int, []),
    "lg_gemm_group_end": (c_int, []),
    "lg_adam_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double,
                                 c_double, c_double, c_double, c_double, c_int]),
    "lg_adam_step_dev_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double,
                                     c_double, c_void_p, c_int64, c_int64, c_double, c_int]),
    "lg_counter_add_i64": (c_int, [c_void_p, c_int64]),
    "lg_adam_multi_dev_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, _I64P, c_double, c_double, c_double,
                                      c_double, c_void_p, c_int64, c_double, c_int]),
    "lg_adam_plan_create": (c_int, [POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                    c_double, c_double, c_double, c_double, c_double, c_int]),
    "lg_adam_plan_destroy": (c_int, [c_void_p]),
    "lg_adam_epilogue_arm": (c_int, [c_void_p, c_int64, c_void_p]),
    "lg_adam_epilogue_finish": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "lg_adam_epilogue_disarm": (c_int, []),
    "lg_gemm_bias_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                 c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "lg_gemm_addend_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_void_p, c_int64]),
    "lg_gemm_act_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                               c_void_p, c_int, c_void_p, c_int64]),
    "lg_gemm_multi3_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "lg_gemm_kseg3_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                 c_void_p, c_int64]),
    "lg_gemm_fused_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                  c_int, c_void_p, c_void_p, c_int, c_int, c_int]),
    "lg_gemm_batched2_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                     c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int]),
    "lg_gemm_rowsum_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_int64, c_int, c_void_p, c_int]),
    "lg_mse_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "lg_head_fwd_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int64, c_int64, c_int64]),
    "lg_gemm_bias_head_fwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_int, POINTER(c_int)]),
    "lg_head_fwd_grad_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_int64, c_int64]),
    "lg_head_bwd_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "lg_mse_finalize_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "lg_softmax_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64]),
    "lg_softmax_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64]),
    "lg_softmax_scaled_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float]),
    "lg_softmax_scaled_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float]),
    "lg_attention_supported": (c_int, [c_int64, c_int64]),
    "lg_attention_fwd_f32": (c_int, [c_void_p, c_int64, c_int64] * 4 + [c_void_p] + [c_int64] * 4 + [c_float]),
    "lg_attention_bwd_f32": (c_int, [c_void_p, c_int64, c_int64] * 4 + [c_void_p] + [c_void_p, c_int64, c_int64] * 3 + [c_int64] * 4 + [c_float]),
    "lg_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double]),
    "lg_layernorm_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64]),
    "lg_cross_entropy_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64]),
    "lg_cross_entropy_mean_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64]),
    "lg_layernorm_param_grads_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int]),
    "lg_take_axis": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "lg_put_axis": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_void_p, c_uint64]),
    "lg_scatter_add_axis_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "lg_index_fold": (c_int, [c_int, _I64P, POINTER(c_void_p), POINTER(c_int), _I64P, c_int, _I64P, POINTER(_I64P), c_void_p]),
    "lg_mask_nonzero": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "lg_gather_rows_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64]),
    "lg_gather_sum3_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64] * 3 + [c_int, c_void_p, c_int64, c_int64]),
    "lg_scatter_add_rows_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64]),
}

# include/lghip_p2p.h: the peer-window gradient exchange, exported by liblghip.so itself (no RCCL)
P2P_HANDLE_BYTES, P2P_MAX_RANKS = 64, 8
P2P_SUM, P2P_MAX = 0, 1
P2P_PROTOTYPES = {
    "lg_p2p_export": (c_int, [c_int, c_int, c_int64, c_void_p]),          # char handle[64]
    "lg_p2p_connect": (c_int, [c_void_p]),
    "lg_p2p_rank": (c_int, [POINTER(c_int), POINTER(c_int), _I64P]),
    "lg_p2p_state": (c_int, [POINTER(c_int), c_void_p]),                   # int* failed, char memory_kind[32]
    "lg_p2p_debug_seed_epochs": (c_int, [c_int]),
    "lg_p2p_allreduce_f32": (c_int, [c_void_p, c_int64, c_int]),
    "lg_p2p_adam_multi_dev_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, _I64P, c_double, c_double, c_double,
                                          c_double, c_void_p, c_int64, c_double, c_int]),
    "lg_p2p_disconnect": (c_int, []),
    "lg_p2p_free": (c_int, []),
}

COMM_PROTOTYPES = {
    "lg_comm_last_error": (c_char_p, []),
    "lg_comm_get_unique_id": (c_int, [c_void_p]),          # char id[128]
    "lg_comm_init": (c_int, [c_int, c_int, c_void_p]),
    "lg_comm_rank": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "lg_comm_selftest": (c_int, [c_double]),
    "lg_comm_allreduce_f32": (c_int, [c_void_p, c_int64, c_int]),
    "lg_comm_fork": (c_int, []),
    "lg_comm_allreduce_forked_f32": (c_int, [c_void_p, c_int64, c_int]),
    "lg_comm_join": (c_int, []),
    "lg_comm_broadcast_f32": (c_int, [c_void_p, c_int64, c_int]),
    "lg_comm_destroy": (c_int, []),
}


class HipError(RuntimeError):
    pass


def load_library(path=LIB_PATH, prototypes=PROTOTYPES, mode=ctypes.DEFAULT_MODE):
    """dlopen + prototype every entry point.  Does not touch the GPU."""
    if not os.path.exists(path):
        raise HipError(
            "%s is missing: the HIP library has not been built. Build it with "
            "`make -C lightgrad_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "The HipTensor backend has no CPU fallback." % path)
    lib = ctypes.CDLL(path, mode=mode)
    if prototypes is PROTOTYPES:
        prototypes = dict(PROTOTYPES, **P2P_PROTOTYPES)          # both headers are implemented by liblghip.so
    for name, (restype, argtypes) in prototypes.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = restype, argtypes
    return lib


_lib = None


def lib():
    """The initialised library: loaded, bound to the GPU of this process (LOCAL_RANK, default 0)."""
    global _lib
    if _lib is None:
        handle = load_library(mode=ctypes.RTLD_GLOBAL)
        n = c_int(0)
        handle.lg_device_count(ctypes.byref(n))
        if n.value < 1:
            raise HipError("no HIP device visible: the HipTensor backend needs an MI355X (gfx950); "
                           "there is no CPU fallback - use CpuTensor explicitly for host execution")
        device = int(os.environ.get("LIGHTGRAD_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if not 0 <= device < n.value:
            # never wrap around: two ranks on one GPU make RCCL report a duplicate device or hang
            raise HipError("this process is bound to HIP device %d (LIGHTGRAD_HIP_DEVICE / LOCAL_RANK) but only %d device(s) "
                           "are visible: start at most one rank per GPU" % (device, n.value))
        rc = handle.lg_init(device)
        if rc != 0:
            raise HipError("lg_init(%d) failed: %s" % (device, handle.lg_last_error().decode()))
        _lib = handle
    return _lib


LG_EINDEX = -6


def check(rc):
    if rc != 0:
        if rc == LG_EINDEX:          # what numpy reports at once, a kernel can only report at the next synchronisation
            raise IndexError(_lib.lg_last_error().decode())
        raise HipError("liblghip error %d: %s" % (rc, _lib.lg_last_error().decode()))


_comm = None


def comm_lib():
    global _comm
    if _comm is None:
        lib()   # the core library must be loaded (RTLD_GLOBAL) and initialised first
        _comm = load_library(COMM_LIB_PATH, COMM_PROTOTYPES)
    return _comm


def comm_check(rc):
    if rc != 0:
        raise HipError("liblghip_comm error %d: %s" % (rc, _comm.lg_comm_last_error().decode()))


# ---- cached int64 arrays for shapes / strides ------------------------------------
_arr_cache = {}


def i64(values):
    """ctypes int64 array for a tuple of ints (cached: the same few shapes recur every step)."""
    arr = _arr_cache.get(values)
    if arr is None:
        if len(_arr_cache) > 65536:
            _arr_cache.clear()
        arr = (c_int64 * max(1, len(values)))(*values)
        _arr_cache[values] = arr
    return arr
