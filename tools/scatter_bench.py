"""Embedding-gradient scatter-add of one batch (1024 ids x 128 floats into a 30522 x 128 table): back-to-back (table lines hot
in the caches) and with 512 MiB of other traffic between two launches (cold), per LG_SCATTER_OWNER setting."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L, HipDevice   # noqa: E402

lib = L.lib()
rng = np.random.RandomState(0)
table = HipTensor.from_numpy(np.zeros((30522, 128), np.float32), requires_grad=False)
g = HipTensor.from_numpy(rng.uniform(-1, 1, (1024, 128)).astype(np.float32), requires_grad=False)
big = HipTensor.from_numpy(np.zeros(128 << 20, np.float32), requires_grad=False)


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def scatter(ids):
    L.check(lib.lg_scatter_add_rows_f32(g.ptr, ids.ptr, 4, table.ptr, 1024, 128, 30522))


def timed(ids, cold):
    best = 1e9
    for _ in range(5):
        if cold:
            big.fill(1.0)                     # 512 MiB written: nothing of the table is left in L2 / Infinity Cache
        e0, e1 = event(), event()
        L.check(lib.lg_event_record(e0))
        scatter(ids)
        L.check(lib.lg_event_record(e1))
        HipDevice.synchronize()
        ms = ctypes.c_float()
        L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        best = min(best, 1e3 * ms.value)
    return best


for name, ids_np in [("random ids (few repeats)", rng.randint(0, 30522, 1024)), ("positions 0..127 x 8", np.tile(np.arange(128), 8)),
                     ("one id 1024 times", np.zeros(1024)), ("ids 0..1023 (neighbouring rows)", np.arange(1024))]:
    ids = HipTensor.from_numpy(ids_np.astype(np.int32), requires_grad=False)
    scatter(ids)
    print("%-34s hot %6.2f us   cold %6.2f us   [LG_SCATTER_OWNER=%s]" % (name, timed(ids, False), timed(ids, True), os.environ.get("LG_SCATTER_OWNER", "1")))
