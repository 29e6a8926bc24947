"""Where does the time of one mid-size GEMM launch go?  Needs the experiments build (make -C lightgrad_amd/csrc timeline):
every workgroup of the LAST launch stamps the 100 MHz wall clock at entry / first K-tile in LDS / K loop done / slab
drained / ticket drawn / fold done / epilogue issued.  Prints, per shape, the spread of workgroup entry times and the median
duration of every phase (microseconds).

    LIGHTGRAD_HIP_LIB=lightgrad_amd/liblghip_timeline.so python tools/gemm_timeline.py
"""
import ctypes
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LIGHTGRAD_HIP_LIB", os.path.join(ROOT, "lightgrad_amd", "liblghip_timeline.so"))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

lib = L.lib()
lib.lg_debug_gemm_timeline.restype = ctypes.c_int
lib.lg_debug_gemm_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
rng = np.random.RandomState(0)
t = lambda *s: HipTensor.from_numpy(rng.uniform(-1, 1, s).astype(np.float32))   # noqa: E731
x, w1, g1 = t(1024, 784), t(512, 784), t(1024, 512)
out = HipTensor.empty((1024 * 1024,), requires_grad=False)
db = HipTensor.empty((1024,), requires_grad=False)


def gemm(ta, tb, M, N, K, A, lda, B, ldb):
    return lambda: L.check(lib.lg_gemm_f32(ta, tb, M, N, K, A.ptr, lda, 0, B.ptr, ldb, 0, out.ptr, N, 0, 1, 0))


cases = {
    "fwd1  x @ W1^T   (1024x512, K=784) NT": gemm(0, 1, 1024, 512, 784, x, 784, w1, 784),
    "dx    g1 @ W1    (1024x784, K=512) NN": gemm(0, 0, 1024, 784, 512, g1, 512, w1, 784),
    "dW1   g1^T @ x   (512x784, K=1024) TN": gemm(1, 0, 512, 784, 1024, g1, 512, x, 784),
    "dW1+db1 rowsum   (512x785, K=1024) TN": lambda: L.check(lib.lg_gemm_rowsum_f32(1, 0, 512, 784, 1024, g1.ptr, 512, x.ptr, 784, out.ptr, 784, 0, db.ptr, 0)),
}


def pair():
    L.check(lib.lg_gemm_pair_begin())
    L.check(lib.lg_gemm_rowsum_f32(1, 0, 512, 784, 1024, g1.ptr, 512, x.ptr, 784, out.ptr, 784, 0, db.ptr, 0))
    L.check(lib.lg_gemm_f32(0, 0, 1024, 784, 512, g1.ptr, 512, 0, w1.ptr, 784, 0, out2.ptr, 784, 0, 1, 0))
    L.check(lib.lg_gemm_pair_end())


out2 = HipTensor.empty((1024 * 1024,), requires_grad=False)
cases["PAIR dW1+db1 | dx in one launch ('tiles' = workgroups of the first product)"] = pair
names = ["entry -> first K-tile in LDS", "K loop", "slab write + drain", "ticket", "fold (last arriver)", "epilogue"]
for name, fn in cases.items():
    runs = []
    for rep in range(6):
        fn()
        buf = np.zeros((4096, 8), np.uint64)
        nwg, sl, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.check(lib.lg_debug_gemm_timeline(buf.ctypes.data, 4096, ctypes.byref(nwg), ctypes.byref(sl), ctypes.byref(tiles)))
        runs.append(buf[:nwg.value].astype(np.int64))
    tl = runs[-1]
    n = tl.shape[0]
    t0 = tl[:, 0].min()
    rel = (tl - t0) / 100.0                                         # microseconds since the first workgroup entered
    rel[tl == 0] = np.nan
    print("%s   %d workgroups (%d tiles x %d K-slices)" % (name, n, tiles.value, sl.value))
    if name.startswith("PAIR"):
        n1 = tiles.value
        for tag, part in (("first product ", rel[:n1]), ("second product", rel[n1:])):
            print("   %s: entry median +%.2f us (last +%.2f), K loop median %.2f us (p90 %.2f), finished median +%.2f us, last +%.2f us"
                  % (tag, np.nanmedian(part[:, 0]), np.nanmax(part[:, 0]), np.nanmedian(part[:, 2] - part[:, 1]),
                     np.nanpercentile(part[:, 2] - part[:, 1], 90), np.nanmedian(np.nanmax(part, axis=1)), np.nanmax(part)))
    print("   workgroup entry: median +%.2f us, p90 +%.2f, last +%.2f" % (np.nanmedian(rel[:, 0]), np.nanpercentile(rel[:, 0], 90), np.nanmax(rel[:, 0])))
    for i, ph in enumerate(names):
        d = rel[:, i + 1] - rel[:, i]
        if np.all(np.isnan(d)):
            continue
        print("   %-30s median %.2f us   p90 %.2f   max %.2f   (%d workgroups)" % (ph, np.nanmedian(d), np.nanpercentile(d, 90), np.nanmax(d), np.sum(~np.isnan(d))))
    print("   last stamp of the launch: +%.2f us after the first entry" % np.nanmax(rel))
