"""SURVEY.md §8(d) microbenchmark of the HBM-bound kernels: forward and backward forms of add / mul / relu / exp,
sum / max reductions, the broadcast, leading-axis and transposed-operand variants, at 64 MiB per tensor (fits the
256 MB Infinity Cache) and 512 MiB per tensor (true HBM).  GB/s = algorithmic bytes / HIP-event time.

    python tools/hbm_bench.py [--iters 20] > gpurun_out/hbm_bench.txt
"""
import argparse
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                    # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice, lib as L             # noqa: E402
from lightgrad_amd.autograd.hip import ops as H                        # noqa: E402


def timed(fn, iters):
    lib = L.lib()
    for _ in range(3):
        fn()
    e0, e1, ms = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_float()
    L.check(lib.lg_event_create(ctypes.byref(e0)))
    L.check(lib.lg_event_create(ctypes.byref(e1)))
    L.check(lib.lg_event_record(e0))
    for _ in range(iters):
        fn()
    L.check(lib.lg_event_record(e1))
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    lib.lg_event_destroy(e0)
    lib.lg_event_destroy(e1)
    return ms.value / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="run the cases whose name contains this text")
    args = ap.parse_args()
    L.lib()
    print("%-34s %-16s %10s %10s %8s" % ("kernel", "shape", "ms", "GB/s", "of 8TB/s"))
    for shape in [(4096, 4096), (16384, 8192)]:
        n = shape[0] * shape[1]
        a, b, g = (HipTensor.empty(shape, requires_grad=False) for _ in range(3))
        a.fill(0.5)
        b.fill(-0.25)
        g.fill(1.5)
        bias = HipTensor.empty((shape[1],), requires_grad=False)
        bias.fill(0.125)
        rowmax = a.max(axis=1, keepdims=True)
        sq = (8192, 8192) if n > (1 << 25) else (4096, 4096)
        ta = HipTensor.empty(sq, requires_grad=False)
        ta.fill(1.0)
        tb = HipTensor.empty(sq, requires_grad=False)
        tb.fill(2.0)
        nsq = sq[0] * sq[1]
        out = HipTensor.empty(shape, requires_grad=False)
        cases = [
            ("add fwd", 12 * n, lambda: H._binary(L.EW_ADD, a, b, out=out)),
            ("mul fwd", 12 * n, lambda: H._binary(L.EW_MUL, a, b, out=out)),
            ("relu fwd", 8 * n, lambda: H._ew(L.EW_RELU, shape, [a], out=out)),
            ("exp fwd", 8 * n, lambda: H._ew(L.EW_EXP, shape, [a], out=out)),
            ("relu bwd  g*(t>=0)", 12 * n, lambda: H._ew(L.EW_RELU_BWD, shape, [a, g], out=out)),
            ("exp bwd   y*g", 12 * n, lambda: H._binary(L.EW_MUL, a, g, out=out)),
            ("mul bwd   (g*b, a*g)", 20 * n, lambda: H._ew(L.EW_MUL_BWD, shape, [a, b, g], n_out=2)),
            ("max bwd   g*(x==val)", 8 * n, lambda: H._ew(L.EW_MAX_BWD, shape, [a, rowmax, rowmax], out=out)),
            ("iadd      a += b", 12 * n, lambda: a.__iadd__(b)),
            ("add bias  (N,C)+(C,)", 8 * n, lambda: H._binary(L.EW_ADD, a, bias, out=out)),
            ("sum -> ()", 4 * n, lambda: a.sum()),
            ("max -> ()", 4 * n, lambda: a.max()),
            ("sum axis=0 (un-broadcast)", 4 * n, lambda: a.sum(axis=0)),
            ("sum axis=1", 4 * n, lambda: a.sum(axis=1)),
            ("max axis=1 keepdims", 4 * n, lambda: a.max(axis=1, keepdims=True)),
            ("add a + b.T (transposed view)", 12 * nsq, lambda: ta + tb.transpose(1, 0)),
            ("contiguous(b.T)", 8 * nsq, lambda: tb.transpose(1, 0).contiguous()),
            ("iadd a += b.T", 12 * nsq, lambda: ta.__iadd__(tb.transpose(1, 0))),
            ("fill", 4 * n, lambda: out.fill(0.0)),
        ]
        for name, nbytes, fn in cases:
            if args.only not in name:
                continue
            ms = timed(fn, args.iters)
            gbs = nbytes / ms / 1e6
            shp = "%dx%d" % (sq if ".T" in name else shape)
            print("%-34s %-16s %10.4f %10.1f %7.1f%%" % (name, shp, ms, gbs, gbs / 80.0))
        del a, b, g, out, ta, tb
        HipDevice.synchronize()
        L.check(L.lib().lg_pool_trim())


if __name__ == "__main__":
    main()
