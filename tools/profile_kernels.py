"""Launch each hot kernel a few times through the C ABI - the workload for rocprofv3 PMC passes
(FETCH_SIZE / WRITE_SIZE per dispatch; see profiles/README.md for the commands and corrections)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                       # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice, lib as L  # noqa: E402

lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = 3
rng = np.random.RandomState(0)
a = HipTensor.from_numpy(rng.uniform(-1, 1, (n, n)).astype(np.float32))
b = HipTensor.from_numpy(rng.uniform(-1, 1, (n, n)).astype(np.float32))
c = HipTensor.empty((n, n), requires_grad=False)
for ta, tb in [(0, 0), (0, 1), (1, 0)]:
    for _ in range(reps):
        L.check(lib.lg_gemm_f32(ta, tb, n, n, n, a.ptr, n, 0, b.ptr, n, 0, c.ptr, n, 0, 1, 0))
big = (16384, 8192)
p, q, r = (HipTensor.empty(big, requires_grad=False) for _ in range(3))
p.fill(0.5)
q.fill(0.25)
for _ in range(reps):
    _ = p + q
    _ = p.relu()
    _ = p.exp()
    _ = p.sum()
    _ = p.sum(axis=0)
    _ = p.max(axis=1)
HipDevice.synchronize()
print("done")
