"""Upper bound for folding the optimizer update into the backward kernels: the MNIST-MLP step replayed from a hipGraph with and
without its update launch (round 3: 59.7 vs 53.5 us).  Not pursued: in the paired launch dx = g @ W reads the W that the dW tiles' 
epilogues would overwrite.      python tools/no_opt_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import lightgrad_amd as light
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import HipGraph, HipDevice
from lightgrad_amd.dist import DataParallel, SingleProcess
class MLP(light.nn.Module):
    def __init__(self):
        light.nn.Module.__init__(self); self.l1 = light.nn.Linear(784, 512); self.l2 = light.nn.Linear(512, 10)
    def forward(self, x): return self.l2(self.l1(x.reshape(-1, 784)).relu())
np.random.seed(0)
model = MLP().map_parameters(lambda p: p.hip())
dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True); dp.attach(opt)
x = HipTensor.from_numpy(np.random.uniform(0, 1, (1024, 784)).astype(np.float32))
t = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 1024)])
def fb():
    l = light.loss.mse(model(x), t); opt.zero_grad(); l.backward(); return l
def full():
    l = fb(); opt.step(); return l
for name, fn in (("fwd+bwd+update", full), ("fwd+bwd only", fb)):
    for _ in range(3): fn()
    g = HipGraph()
    with g.capture():
        for _ in range(25): fn()
    for _ in range(5): g.replay()
    HipDevice.synchronize(); t0 = time.perf_counter()
    for _ in range(80): g.replay()
    HipDevice.synchronize(); dt = time.perf_counter() - t0
    print("%-16s %.2f us per step, %d kernels per step" % (name, 1e6 * dt / 2000, g.kernel_count() // 25))
