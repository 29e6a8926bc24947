"""Eager tiny-BERT TRAINING steps (forward, masked-LM loss, backward, fused AdaBelief) for a while: the memory pool must be in steady
state - no hipMalloc, no growth - with python's cycle collector on, off (`nogc`: tapes must die by reference counting alone) or
forced (`collect`).      python tools/bert_train_soak.py [default|nogc|collect]"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import lightgrad_amd as light
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import HipDevice
from lightgrad_amd.dist import DataParallel, SingleProcess
spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
bert = importlib.util.module_from_spec(spec); spec.loader.exec_module(bert)
np.random.seed(0)
m = bert.BertForMaskedLM(**bert.TINY).map_parameters(lambda t: t.hip())
ids = HipTensor.from_numpy(np.random.randint(0, 30522, (8, 128)).astype(np.int32), requires_grad=False)
labels = HipTensor.from_numpy(np.random.randint(0, 30522, (1024,)).astype(np.int64), requires_grad=False)
dp = DataParallel(m.parameters(), SingleProcess(), flatten=True)
opt = light.optim.AdaBelief(m.parameters(), lr=1e-4, fused=True, device_step=True); dp.attach(opt)
def step():
    loss = light.loss.cross_entropy(m(ids).reshape(-1, 30522), labels)
    opt.zero_grad(); loss.backward(); dp.sync_gradients(); opt.step()
    return loss
for _ in range(50): l = step()
HipDevice.synchronize(); s0 = HipDevice.pool_stats(); t0 = time.time(); first = l.item()
import gc
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "nogc":
    gc.disable()
for i in range(3000):
    l = step()
    if mode == "collect" and i % 50 == 0:
        gc.collect()
    if i % 500 == 0:
        HipDevice.synchronize(); s = HipDevice.pool_stats()
        print("step %4d: reserved %.1f MB  in use %.1f MB  hipMalloc calls %d  gc counts %s" % (i, s["reserved_bytes"] / 1e6, s["in_use_bytes"] / 1e6, s["hip_malloc_calls"], gc.get_count()))
HipDevice.synchronize(); s1 = HipDevice.pool_stats()
print("eager BERT training steps: 3000 in %.1f s, loss %.4f -> %.4f; pool in use %d -> %d, hipMalloc calls %d -> %d" % (time.time()-t0, first, l.item(), s0["in_use_bytes"], s1["in_use_bytes"], s0["hip_malloc_calls"], s1["hip_malloc_calls"]))
print("reserved %.1f -> %.1f MB" % (s0["reserved_bytes"] / 1e6, s1["reserved_bytes"] / 1e6))
assert s0["hip_malloc_calls"] == s1["hip_malloc_calls"] and s0["reserved_bytes"] == s1["reserved_bytes"], "the pool grows in steady state"
