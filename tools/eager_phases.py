"""Host time of the eager (python tape every step) MLP training step, phase by phase - perf_counter around each phase, no
synchronisation inside the loop (the GPU runs behind): where the 110 us go.   python tools/eager_phases.py [steps]"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightgrad_amd as light                                   # noqa: E402
from lightgrad_amd import HipTensor                              # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice                 # noqa: E402
from lightgrad_amd.dist import SingleProcess, DataParallel       # noqa: E402


class MLP(light.nn.Module):
    def __init__(self):
        light.nn.Module.__init__(self)
        self.l1, self.l2 = light.nn.Linear(784, 512), light.nn.Linear(512, 10)

    def forward(self, x):
        return self.l2(self.l1(x.reshape(-1, 784)).relu())


np.random.seed(0)
model = MLP().map_parameters(lambda p: p.hip())
dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True)
dp.attach(opt)
x = HipTensor.from_numpy(np.random.uniform(0, 1, (1024, 784)).astype(np.float32))
t = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 1024)])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
names = ("model(x)", "loss.mse", "zero_grad", "backward", "sync_gradients", "opt.step", "drop the tape")
acc = [0.0] * len(names)
pc = time.perf_counter
for it in range(n + 100):
    if it == 100:
        HipDevice.synchronize()
        acc = [0.0] * len(names)
        t_all = pc()
    t0 = pc(); y = model(x)
    t1 = pc(); loss = light.loss.mse(y, t)
    t2 = pc(); opt.zero_grad()
    t3 = pc(); loss.backward()
    t4 = pc(); dp.sync_gradients()
    t5 = pc(); opt.step()
    t6 = pc(); del y, loss
    t7 = pc()
    for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7))):
        acc[k] += b - a
HipDevice.synchronize()
total = pc() - t_all
print("%.1f us per eager step (%d steps, timers included)" % (1e6 * total / n, n))
for name, v in zip(names, acc):
    print("  %-16s %6.1f us" % (name, 1e6 * v / n))
