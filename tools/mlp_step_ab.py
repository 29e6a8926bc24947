"""A/B of the MLP training step (784 -> 512 -> 10, batch 1024, AdaBelief) replayed from hipGraphs of 32 steps:
optimizer as a launch of its own vs applied by the backward kernels (optim.Adam.fuse_update_into_backward).
    python tools/mlp_step_ab.py [rounds]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightgrad_amd as light  # noqa: E402
from lightgrad_amd import HipTensor  # noqa: E402
from lightgrad_amd.autograd.hip import HipDevice, HipGraph  # noqa: E402
from lightgrad_amd.dist import DataParallel, SingleProcess  # noqa: E402


class MLP(light.nn.Module):
    def __init__(self):
        light.nn.Module.__init__(self)
        self.l1, self.l2 = light.nn.Linear(784, 512), light.nn.Linear(512, 10)

    def forward(self, x):
        return self.l2(self.l1(x.reshape(-1, 784)).relu())


def build(in_backward, unroll=32, data_input=False):
    np.random.seed(0)
    model = MLP().map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, device_step=True)
    dp.attach(opt)
    if in_backward:
        opt.fuse_update_into_backward()
    rng = np.random.RandomState(1000)
    x = HipTensor.from_numpy(rng.uniform(0, 1, (1024, 784)).astype(np.float32), requires_grad=not data_input)
    onehot = HipTensor.from_numpy(np.eye(10, dtype=np.float32)[rng.randint(0, 10, 1024)])

    def step():
        loss = light.loss.mse(model(x), onehot)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    for _ in range(4):
        step()
    g = HipGraph()
    with g.capture():
        for _ in range(unroll):
            loss = step()
    return g, loss, unroll, (model, dp, opt, x, onehot)


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    variants = {"optimizer launch": build(False), "update in backward kernels": build(True)}
    for name, (g, loss, unroll, _) in variants.items():
        for _ in range(20):
            g.replay()
        print("%-28s kernels per step: %.1f" % (name, g.kernel_count() / unroll))
    HipDevice.synchronize()
    for r in range(rounds):
        line = []
        for name, (g, loss, unroll, _) in variants.items():
            HipDevice.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                g.replay()
            HipDevice.synchronize()
            line.append("%s %.2f us/step" % (name, 1e6 * (time.perf_counter() - t0) / (100 * unroll)))
        print("   ".join(line))
