"""cProfile of the eager MLP training step on the GPU box: where does the host time go?"""
import cProfile, pstats, os, sys, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightgrad_amd as light
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import HipDevice

class MLP(light.nn.Module):
    def __init__(self):
        light.nn.Module.__init__(self)
        self.l1 = light.nn.Linear(784, 512); self.l2 = light.nn.Linear(512, 10)
    def forward(self, x):
        return self.l2(self.l1(x.reshape(-1, 784)).relu())
np.random.seed(0)
model = MLP().map_parameters(lambda p: p.hip())
opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True)
x = HipTensor.from_numpy(np.random.uniform(0, 1, (1024, 784)).astype(np.float32))
t = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 1024)])
def step():
    l = light.loss.mse(model(x), t); opt.zero_grad(); l.backward(); opt.step(); return l
for _ in range(20): step()
HipDevice.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
HipDevice.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
