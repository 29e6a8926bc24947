"""What does a parallel branch cost in a replayed hipGraph?  A chain of 40 tiny kernels on the compute stream is captured with
B side brackets (lg_side_begin / one tiny kernel on another tensor / lg_side_end) spread over it, joined at the end of the
capture; the graph is replayed 200 times.  Prints microseconds per replay for B = 0, 1, 2, 4, 8 and the same work with the
bracketed kernels left on the compute stream.

    python tools/graph_branch_probe.py
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                    # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice, HipGraph, lib as L   # noqa: E402

lib = L.lib()
CHAIN, REPLAYS = 40, 200
main = HipTensor.from_numpy(np.zeros(256, np.float32), requires_grad=False)
side = [HipTensor.from_numpy(np.zeros(256, np.float32), requires_grad=False) for _ in range(8)]


def record(branches, on_side_stream, join_after=None):
    """join_after: kernels of the chain between a bracket and the lg_side_join that follows it (None: one join, at the end)"""
    every = CHAIN // branches if branches else CHAIN + 1
    graph = HipGraph()
    with graph.capture():
        b, join_at = 0, -1
        for k in range(CHAIN):
            main.__iadd__(1.0)
            if k == join_at:
                L.check(lib.lg_side_join())
            if branches and k % every == every - 1 and b < branches:
                if on_side_stream:
                    L.check(lib.lg_side_begin())
                side[b].__iadd__(1.0)
                if on_side_stream:
                    L.check(lib.lg_side_end())
                    if join_after is not None:
                        join_at = k + join_after
                b += 1
    return graph


def timed(graph):
    for _ in range(5):
        graph.replay()
    HipDevice.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPLAYS):
        graph.replay()
    HipDevice.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / REPLAYS
    graph.destroy()
    return us


print("%d dependent tiny kernels per graph, %d replays" % (CHAIN, REPLAYS))
for branches in (0, 1, 2, 4, 8):
    inline = timed(record(branches, False))
    forked = timed(record(branches, True)) if branches else inline
    print("  %d extra kernels:  all on the compute stream %7.1f us per replay   each in a side bracket (a branch of the graph) %7.1f us"
          "   -> %.1f us per branch" % (branches, inline, forked, (forked - inline) / branches if branches else 0.0))
print("the same with every branch joined back 2 chain kernels after its fork (the shape of a gradient exchange inside a training step):")
for branches in (1, 2, 4, 8):
    inline = timed(record(branches, False))
    forked = timed(record(branches, True, join_after=2))
    print("  %d fork + join pairs:  inline %7.1f us per replay   forked %7.1f us   -> %.1f us per pair" % (branches, inline, forked, (forked - inline) / branches))
