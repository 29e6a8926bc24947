"""MNIST-MLP training loop on synthetic data (counterpart of the reference's examples/mnist.py:24-67 with
the BASELINE shapes 784 -> 512 -> 10, batch 1024; the real dataset needs network access).

    python examples/mnist.py [--cpu] [--steps 200] [--graph]
"""
import argparse
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lightgrad_amd as light  # noqa: E402
import lightgrad_amd.nn as nn  # noqa: E402
from lightgrad_amd.autograd.utils.profiler import Profiler  # noqa: E402


class NN(nn.Module):
    def __init__(self, hidden=512):
        nn.Module.__init__(self)
        self.l1 = nn.Linear(28 * 28, hidden)
        self.l2 = nn.Linear(hidden, 10)

    def forward(self, x):
        return self.l2(self.l1(x.reshape(-1, 28 * 28)).relu())


def synthetic_batch(rng, batch):
    x = rng.uniform(0, 1, (batch, 1, 28, 28)).astype(np.float32)
    labels = rng.randint(0, 10, batch)
    one_hot = light.zeros((batch, 10))
    one_hot[range(batch), labels] = 1          # fancy setitem on the CPU tensor, then moved to the device
    return light.from_numpy(x), one_hot


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    to_device = (lambda t: t) if args.cpu else (lambda t: t.hip())
    np.random.seed(0)
    model = NN().map_parameters(to_device)
    optim = light.optim.AdaBelief(model.parameters(), lr=0.001, fused=not args.cpu)
    rng = np.random.RandomState(1)
    with Profiler() as prof:
        losses, t0 = [], time.perf_counter()
        for i in range(args.steps):
            x, one_hot = synthetic_batch(rng, args.batch)
            y = model(to_device(x))
            l = light.loss.mse(y, to_device(one_hot))
            optim.zero_grad()
            l.backward()
            optim.step()
            losses.append(l.item())
        dt = time.perf_counter() - t0
    print("loss %.5f -> %.5f   %.1f steps/s (incl. host batch generation and upload)" % (losses[0], losses[-1], args.steps / dt))
    prof.print(topn=12)
