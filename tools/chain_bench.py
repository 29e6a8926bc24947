"""The hidden layer's product followed by the skinny head: two launches against lg_gemm_bias_head_fwd_f32's one (MNIST MLP shapes),
back-to-back launches timed with HIP events.

    python tools/chain_bench.py > gpurun_out/chain_bench.txt
"""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                    # noqa: E402
from lightgrad_amd.autograd.hip import lib as L                        # noqa: E402
from hbm_bench import timed                                            # noqa: E402


def main():
    lib = L.lib()
    rows, d_in, hidden, outs = 1024, 784, 512, 10
    rng = np.random.RandomState(0)
    mk = lambda *s: HipTensor.from_numpy(rng.uniform(-1, 1, s).astype(np.float32), requires_grad=False)       # noqa: E731
    x, w1, b1, w2, b2, t = mk(rows, d_in), mk(hidden, d_in), mk(hidden), mk(outs, hidden), mk(outs), mk(rows, outs)
    pre, y, e, r, dx, gp = (HipTensor.empty(s, requires_grad=False) for s in ((rows, hidden), (rows, outs), (rows, outs), (rows,), (rows, hidden), (rows, hidden)))
    n = ctypes.c_int(0)

    def two():
        L.check(lib.lg_gemm_bias_f32(0, 1, rows, hidden, d_in, x.ptr, d_in, 0, w1.ptr, d_in, 0, pre.ptr, hidden, 0, 1, b1.ptr))
        L.check(lib.lg_head_fwd_grad_f32(pre.ptr, hidden, 1, w2.ptr, b2.ptr, t.ptr, y.ptr, e.ptr, r.ptr, dx.ptr, gp.ptr, rows, hidden, outs))

    def one():
        L.check(lib.lg_gemm_bias_head_fwd_f32(x.ptr, d_in, w1.ptr, d_in, b1.ptr, pre.ptr, rows, hidden, d_in, 1, w2.ptr, b2.ptr, t.ptr,
                                              y.ptr, e.ptr, r.ptr, dx.ptr, gp.ptr, outs, 1, ctypes.byref(n)))
    for rep in range(3):
        a = timed(two, 50) * 1e3
        b = timed(one, 50) * 1e3
        print("product, then head: %.2f us    chained (%d launch): %.2f us" % (a, n.value, b))
    if os.environ.get("LG_CHAIN_NO_ROWS") == "1":
        return
    # inside hipGraphs of 8 repetitions (how the training step runs them)
    from lightgrad_amd.autograd.hip.graph import HipGraph
    graphs = []
    for fn in (two, one):
        g = HipGraph()
        with g.capture():
            for _ in range(8):
                fn()
        graphs.append(g)
    for rep in range(3):
        a = timed(graphs[0].replay, 20) * 1e3 / 8
        b = timed(graphs[1].replay, 20) * 1e3 / 8
        print("replayed graphs of 8:  product, then head: %.2f us    chained: %.2f us" % (a, b))


if __name__ == "__main__":
    main()
