"""The forward product of the MNIST MLP's first layer (1024 x 512 x 784, NT, bias) replayed from a hipGraph of 20 launches, for the tile
LG_GEMM_TILE forces (7 = 64x32 with two K-groups, the cost model's choice; experiments: 5, 6):   LG_GEMM_TILE=7 python tools/fwd_tile_bench.py"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                    # noqa: E402
from lightgrad_amd.autograd.hip import lib as L                        # noqa: E402
from lightgrad_amd.autograd.hip.graph import HipGraph                  # noqa: E402
from hbm_bench import timed                                            # noqa: E402

lib = L.lib()
rows, d_in, hidden = 1024, 784, 512
rng = np.random.RandomState(0)
mk = lambda *s: HipTensor.from_numpy(rng.uniform(-1, 1, s).astype(np.float32), requires_grad=False)       # noqa: E731
x, w1, b1 = mk(rows, d_in), mk(hidden, d_in), mk(hidden)
pre = HipTensor.empty((rows, hidden), requires_grad=False)
fn = lambda: L.check(lib.lg_gemm_bias_f32(0, 1, rows, hidden, d_in, x.ptr, d_in, 0, w1.ptr, d_in, 0, pre.ptr, hidden, 0, 1, b1.ptr))   # noqa: E731
fn()
g = HipGraph()
with g.capture():
    for _ in range(20):
        fn()
print("LG_GEMM_TILE=%s: %s" % (os.environ.get("LG_GEMM_TILE", "auto"), "  ".join("%.2f us" % (timed(g.replay, 20) * 1e3 / 20) for _ in range(4))))
