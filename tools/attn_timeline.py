"""Where does the time of a fused attention launch go?  Needs the experiments build (make -C lightgrad_amd/csrc timeline):
every workgroup of the LAST launch stamps the 100 MHz wall clock between its phases.  Prints the median duration of every
phase (microseconds) for the forward and for both roles of the backward at tiny-BERT's size.

    python tools/attn_timeline.py [batch seq heads d]
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
lib.lg_debug_attn_timeline.restype = ctypes.c_int
lib.lg_debug_attn_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
b, s, heads, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 128, 2, 64)
rng = np.random.RandomState(0)
q, k, v, w = (HipTensor.from_numpy(rng.uniform(-1, 1, (b, s, heads * d)).astype(np.float32)) for _ in range(4))


def stamps():
    buf = np.zeros((4096, 16), np.uint64)
    n = lib.lg_debug_attn_timeline(buf.ctypes.data, 4096)
    assert n > 0
    return buf[:n].astype(np.float64) / 100.0          # microseconds


def report(title, t, names):
    print("%s: %d workgroups, entries spread over %.2f us, first entry to last exit %.2f us" %
          (title, len(t), t[:, 0].max() - t[:, 0].min(), t[:, len(names)].max() - t[:, 0].min()))
    for i, n in enumerate(names):
        print("    %-46s median %6.2f us   max %6.2f us" % (n, np.median(t[:, i + 1] - t[:, i]), (t[:, i + 1] - t[:, i]).max()))


for rep in range(3):
    out = q.attention(k, v, heads=heads, scale=d ** -0.5)
    fwd = stamps()
    (out * w).backward(allow_fill=True)
    bwd = stamps()
    for t in (q, k, v):
        t.zero_grad()
report("forward", fwd, ["loads -> LDS", "scores (MFMA) -> LDS", "softmax, P -> HBM", "context MFMAs", "partial sums -> LDS", "fold, O -> HBM"])
half = len(bwd) // 2                       # the query-role workgroups fill the first half of the grid
report("backward, query role", bwd[:half], ["loads -> LDS", "dP (MFMA) -> LDS", "dS rows, shifts published", "dQ MFMAs", "partial sums -> LDS", "fold, dQ -> HBM"])
report("backward, key role", bwd[half:], ["loads -> LDS", "dP of the block's keys (MFMA)", "wait for the shifts", "dS, P of the block -> LDS", "dV / dK MFMAs", "dV / dK -> HBM"])
print("key role: entry %.2f us after the first query-role entry (median)" % float(np.median(bwd[half:, 0]) - bwd[:half, 0].min()))
