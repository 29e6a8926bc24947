"""The C-ABI libraries load on a CPU-only machine and export exactly what include/*.h declares.
No compute calls here (no GPU): only dlopen, symbol lookup and error paths that need no device."""
import ctypes
import os
import re
import pytest
from conftest import ROOT
from lightgrad_amd.autograd.hip import lib as hiplib

DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void\s*\*|char\s*\*|const char\s*\*)\s*(lg_[a-z0-9_]+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


def test_core_library_exports_every_declared_symbol():
    names = declared("lghip.h")
    assert len(names) >= 28, names
    assert sorted(hiplib.PROTOTYPES) == names, "python prototypes and include/lghip.h disagree"
    handle = hiplib.load_library()          # raises if the .so is missing or lacks a symbol
    for n in names:
        assert getattr(handle, n) is not None


def test_core_library_exports_the_peer_window_exchange():
    names = declared("lghip_p2p.h")
    assert len(names) == 9, names
    assert sorted(hiplib.P2P_PROTOTYPES) == names, "python prototypes and include/lghip_p2p.h disagree"
    handle = hiplib.load_library()
    for n in names:
        assert getattr(handle, n) is not None
    text = open(os.path.join(ROOT, "include", "lghip_p2p.h")).read()
    assert "#define LG_P2P_HANDLE_BYTES %d" % hiplib.P2P_HANDLE_BYTES in text
    assert re.search(r"#define LG_P2P_MAX_RANKS\s+%d" % hiplib.P2P_MAX_RANKS, text)
    # without lg_init nothing is allocated or exported
    buf = ctypes.create_string_buffer(hiplib.P2P_HANDLE_BYTES)
    n = ctypes.c_int(0)
    handle.lg_device_count(ctypes.byref(n))
    if n.value == 0:
        assert handle.lg_p2p_export(0, 2, 1024, buf) == -4          # LG_ENOTINIT
        assert handle.lg_p2p_allreduce_f32(None, 0, 0) == -4


def test_comm_library_exports_every_declared_symbol():
    names = declared("lghip_comm.h")
    assert sorted(hiplib.COMM_PROTOTYPES) == names
    ctypes.CDLL(hiplib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    handle = hiplib.load_library(hiplib.COMM_LIB_PATH, hiplib.COMM_PROTOTYPES)
    for n in names:
        assert getattr(handle, n) is not None


def test_uninitialised_library_fails_loudly():
    handle = hiplib.load_library()
    assert b"liblghip" in handle.lg_version()
    n = ctypes.c_int(-1)
    assert handle.lg_device_count(ctypes.byref(n)) == 0
    if n.value == 0:
        p = ctypes.c_void_p()
        assert handle.lg_malloc(ctypes.byref(p), 16) == -4          # LG_ENOTINIT
        assert b"lg_init" in handle.lg_last_error()
        assert handle.lg_sync() == -4


def test_enum_ids_match_header():
    text = open(os.path.join(ROOT, "include", "lghip.h")).read()
    body = re.search(r"typedef enum \{(.*?)\} lg_ew_op_t;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    value, ids = -1, {}
    for tok in body.split(","):
        tok = tok.strip()
        if not tok:
            continue
        if "=" in tok:
            name, v = (s.strip() for s in tok.split("="))
            value = int(v)
        else:
            name, value = tok, value + 1
        ids[name] = value
    for name, v in ids.items():
        assert getattr(hiplib, name[3:]) == v, name      # LG_EW_ADD -> hiplib.EW_ADD


def test_hip_tensor_has_no_cpu_fallback():
    """without a GPU every HipTensor constructor raises HipError instead of computing on the host"""
    handle = hiplib.load_library()
    n = ctypes.c_int(0)
    handle.lg_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    import numpy as np
    from lightgrad_amd import HipTensor, CpuTensor
    with pytest.raises(hiplib.HipError, match="no HIP device"):
        HipTensor.zeros((2, 2))
    with pytest.raises(hiplib.HipError):
        CpuTensor.from_numpy(np.ones(3, np.float32)).hip()
