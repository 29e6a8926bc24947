"""Probe for the rocprofv3 fault analysed in profiles/README.md (r2): replay a SMALL captured graph often enough that the
packets it submits cross the end of the 16384-packet AQL ring several times.  Run it plain and under
`rocprofv3 --kernel-trace`: if only the profiled run dies near 16384 submitted packets, the fault is the profiler's queue
interceptor and not tied to the size of the graph.

    python tools/graph_wrap_probe.py [kernels_per_graph] [replays]
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                   # noqa: E402
from lightgrad_amd.autograd.hip import HipGraph, HipDevice            # noqa: E402
import lightgrad_amd as light                                         # noqa: E402

kernels = int(sys.argv[1]) if len(sys.argv) > 1 else 10
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
a = HipTensor.from_numpy(np.ones((64, 64), np.float32), requires_grad=False)
b = HipTensor.from_numpy(np.full((64, 64), 0.5, np.float32), requires_grad=False)


def body():
    with light.no_grad():
        t = a
        for _ in range(kernels):
            t = t * b
        return t


body()
g = HipGraph()
with g.capture():
    out = body()
for i in range(replays):
    g.replay()
    if (i + 1) % 100 == 0 or kernels >= 100:
        HipDevice.synchronize()
        print("replay %d  (~%d kernel packets submitted)" % (i + 1, (i + 1) * kernels), flush=True)
HipDevice.synchronize()
print("done: %d replays x %d kernels, out[0,0] = %g" % (replays, kernels, out.numpy()[0, 0]), flush=True)
