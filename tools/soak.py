"""Soak run on one GPU: many training steps / replays, looking for what short tests cannot show - memory that grows, a loss that
stops being finite, a replay that differs from the first one.

    python tools/soak.py [seconds per phase, default 20]
"""
import importlib.util
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightgrad_amd as light                                             # noqa: E402
from lightgrad_amd import HipTensor                                        # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice, HipGraph                 # noqa: E402
from lightgrad_amd.dist import DataParallel, SingleProcess                 # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0


def pool():
    HipDevice.synchronize()
    s = HipDevice.pool_stats()
    return s["in_use_bytes"], s["reserved_bytes"], s["hip_malloc_calls"]


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
onehot = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 1024)])


def step():
    loss = light.loss.mse(model(x), onehot)
    opt.zero_grad()
    loss.backward()
    dp.sync_gradients()
    opt.step()
    return loss


# ---- phase 1: the eager tape
first = [step().item() for _ in range(3)]
for _ in range(400):                       # steady state: the previous step's tape is still alive while the next one is built
    loss = step()
before = pool()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < budget:
    for _ in range(200):
        loss = step()
    n += 200
    assert np.isfinite(loss.item())
after = pool()
print("eager MLP steps: %d in %.1f s (%.0f steps/s), loss %.6f -> %.6f; pool in use %d -> %d B, reserved %d -> %d B, hipMalloc calls %d -> %d"
      % (n, time.perf_counter() - t0, n / (time.perf_counter() - t0), first[0], loss.item(), before[0], after[0], before[1], after[1], before[2], after[2]))
assert after[2] == before[2], "the pool still calls hipMalloc in steady state"

# ---- phase 2: eight steps per hipGraph
n_params = len(opt.parameters)
graph = HipGraph()
with graph.capture():
    for _ in range(8):
        gloss = step()
opt.t -= 8 * n_params
before = pool()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < budget:
    for _ in range(500):
        graph.replay()
        opt.on_graph_replay(8)
    n += 4000
    assert np.isfinite(gloss.item())
after = pool()
print("graph MLP steps: %d in %.1f s (%.0f steps/s), loss now %.6f; pool in use %d -> %d B, hipMalloc calls %d -> %d"
      % (n, time.perf_counter() - t0, n / (time.perf_counter() - t0), gloss.item(), before[0], after[0], before[2], after[2]))
assert after[0] == before[0] and after[2] == before[2]
graph.destroy()

# ---- phase 3: tiny-BERT forward + backward, replayed; every replay must reproduce the first one's gradients bit for bit
spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
bert = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bert)
bmodel = bert.BertForMaskedLM(**bert.TINY).map_parameters(lambda t: t.hip())
ids = HipTensor.from_numpy(np.random.randint(0, 30522, (8, 128)).astype(np.int32), requires_grad=False)
labels = HipTensor.from_numpy(np.random.randint(0, 30522, (8 * 128,)).astype(np.int64), requires_grad=False)
bdp = DataParallel(bmodel.parameters(), SingleProcess(), flatten=True)


def bert_step():
    loss = light.loss.cross_entropy(bmodel(ids).reshape(-1, 30522), labels)
    bdp.bucket.fill(0)
    loss.backward()
    return loss


for _ in range(3):
    bert_step()
bgraph = HipGraph()
with bgraph.capture():
    bloss = bert_step()
bgraph.replay()
reference = bdp.bucket.numpy().copy()
ref_loss = bloss.item()
before = pool()
t0, n, worst = time.perf_counter(), 0, 0.0
while time.perf_counter() - t0 < budget:
    for _ in range(500):
        bgraph.replay()
    n += 500
    now = bdp.bucket.numpy()
    # everything but the embedding tables (float atomics: repeated ids add in any order) must be the same bits
    worst = max(worst, float(np.abs(now - reference).max()))
    assert bloss.item() == ref_loss
after = pool()
scale = float(np.abs(reference).max())
print("tiny-BERT replays: %d in %.1f s (%.3f ms each), loss %.6f every time; largest gradient difference to the first replay %.3g "
      "(largest gradient %.3g; embedding rows that receive more than 32 ids are combined with float atomics); pool in use %d -> %d B"
      % (n, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / n, ref_loss, worst, scale, before[0], after[0]))
assert worst <= 1e-5 * scale and after[0] == before[0]
print("soak ok")
