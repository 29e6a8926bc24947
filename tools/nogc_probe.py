"""The CNN and MLP examples trained for a few steps with python's cycle collector OFF: device memory must not grow and the
collector must find nothing afterwards - tapes are acyclic and die by reference counting.   python tools/nogc_probe.py"""
import gc, importlib.util, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import lightgrad_amd as light
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import HipDevice
spec = importlib.util.spec_from_file_location("mnist_example", os.path.join(ROOT, "examples", "mnist.py"))
mn = importlib.util.module_from_spec(spec); spec.loader.exec_module(mn)
np.random.seed(0)
for name, Model, loss_fn in [("CNN + cross-entropy", mn.CNN, "ce"), ("MLP + mse", mn.NN, "mse")]:
    model = Model().map_parameters(lambda p: p.hip())
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3)
    rng = np.random.RandomState(1)
    x = HipTensor.from_numpy(rng.uniform(0, 1, (64, 1, 28, 28)).astype(np.float32))
    labels = rng.randint(0, 10, 64)
    onehot = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[labels])
    lab = HipTensor.from_numpy(labels.astype(np.int64), requires_grad=False)
    def step():
        y = model(x)
        loss = light.loss.cross_entropy(y, lab) if loss_fn == "ce" else light.loss.mse(y, onehot)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss.item()
    gc.collect(); gc.disable()
    for _ in range(5): step()
    HipDevice.synchronize(); a = HipDevice.pool_stats()
    for _ in range(40): l = step()
    HipDevice.synchronize(); b = HipDevice.pool_stats()
    n = gc.collect(); gc.enable()
    print("%s: in use %d -> %d B, hipMalloc %d -> %d, unreachable objects found afterwards: %d, loss %.4f" % (name, a["in_use_bytes"], b["in_use_bytes"], a["hip_malloc_calls"], b["hip_malloc_calls"], n, l))
