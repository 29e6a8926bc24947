"""cProfile of the eager (python tape every step) MLP training step: where the host time goes.
    python tools/profile_eager.py [steps]"""
import cProfile
import os
import pstats
import sys
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


def step():
    loss = light.loss.mse(model(x), t)
    opt.zero_grad()
    loss.backward()
    dp.sync_gradients()
    opt.step()


if "--bert" in sys.argv:
    # tiny-BERT forward + backward as bench.py runs it (batch 8 x 128, masked-LM cross-entropy)
    import importlib.util
    sys.argv.remove("--bert")
    spec = importlib.util.spec_from_file_location("bert_example", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "bert.py"))
    bert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bert)
    bmodel = bert.BertForMaskedLM(**bert.TINY).map_parameters(lambda p: p.hip())
    ids = HipTensor.from_numpy(np.random.randint(0, 30522, (8, 128)).astype(np.int32), requires_grad=False)
    labels = HipTensor.from_numpy(np.random.randint(0, 30522, (8 * 128,)).astype(np.int64), requires_grad=False)
    bdp = DataParallel(bmodel.parameters(), SingleProcess(), flatten=True)

    def step():                                                   # noqa: F811
        loss = light.loss.cross_entropy(bmodel(ids).reshape(-1, 30522), labels)
        bdp.bucket.fill(0)
        loss.backward()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for _ in range(50):
    step()
HipDevice.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n):
    step()
HipDevice.synchronize()
print("%.1f us per eager step (un-profiled)" % (1e6 * (time.perf_counter() - t0) / n))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
HipDevice.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(45)
