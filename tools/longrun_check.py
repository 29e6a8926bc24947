"""long-horizon check: 3000 training steps of the MLP on the GPU (eager tape, fused optimizer) against the numpy oracle - losses\nat steps 200 / 1000 / 3000 (run on the GPU box: python tools/longrun_check.py)"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import lightgrad_amd as light
from lightgrad_amd import HipTensor
import np_oracle as O
w0, xc, tc, _ = O.synthetic_mlp_problem(0)
# GPU eager (fused optimizer, python-side t)
class MLP(light.nn.Module):
    def __init__(self):
        light.nn.Module.__init__(self); self.l1 = light.nn.Linear(784, 512); self.l2 = light.nn.Linear(512, 10)
    def forward(self, x): return self.l2(self.l1(x.reshape(-1, 784)).relu())
m = MLP(); m.load_parameters({"l1.weight": w0["l1.weight"], "l1.bias": w0["l1.bias"], "l2.weight": w0["l2.weight"], "l2.bias": w0["l2.bias"]})
m.map_parameters(lambda p: p.hip())
opt = light.optim.AdaBelief(m.parameters(), lr=1e-3, fused=True)
x, t = HipTensor.from_numpy(xc), HipTensor.from_numpy(tc)
marks = [200, 1000, 3000]
gl = {}
for s in range(1, 3001):
    l = light.loss.mse(m(x), t); opt.zero_grad(); l.backward(); opt.step()
    if s in marks: gl[s] = l.item()
print("gpu eager", gl)
w = {k: v.copy() for k, v in w0.items()}
o = O.make_optimizer("adabelief")
cl = {}
t0 = time.time()
for s in range(1, 3001):
    loss, grads, _ = O.mlp_loss_and_grads(w, xc, tc)
    for name in O.PARAM_ORDER: w[name] += o.delta(name, grads[name])
    if s in marks: cl[s] = float(loss)
print("cpu oracle", cl, "%.0f s" % (time.time() - t0))
