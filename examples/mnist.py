"""MNIST-MLP training loop on synthetic data (counterpart of the reference's examples/mnist.py:24-67 with
the BASELINE shapes 784 -> 512 -> 10, batch 1024; the real dataset needs network access).  A NEW batch is
generated on the host and uploaded EVERY step, so the printed rate is the PCIe-inclusive one.

    python examples/mnist.py [--cpu] [--steps 200] [--graph] [--cnn]

--graph: the step is captured once into a hipGraph; every iteration uploads the batch into the graph's static
input tensors (pinned staging, asynchronous) and replays the graph.
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


class CNN(nn.Module):
    """ the reference's examples/mnist.py:14-22 """
    def __init__(self):
        nn.Module.__init__(self)
        self.c1 = nn.Conv2d(1, 8, kernelsize=3, bias=False, pad=0)
        self.c2 = nn.Conv2d(8, 16, kernelsize=3, bias=False, pad=0)
        self.l1 = nn.Linear(5 * 5 * 16, 10)

    def forward(self, x):
        y = self.c1(x).max_pool().relu()
        y = self.c2(y).max_pool().relu()
        return self.l1(y.reshape(-1, 5 * 5 * 16))


def synthetic_batch(rng, batch):
    x = rng.uniform(0, 1, (batch, 1, 28, 28)).astype(np.float32)
    labels = rng.randint(0, 10, batch)
    one_hot = np.zeros((batch, 10), np.float32)
    one_hot[np.arange(batch), labels] = 1
    return x, one_hot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--cnn", action="store_true", help="the convolutional model instead of the MLP")
    ap.add_argument("--loss", choices=["mse", "ce"], default="mse", help="ce: the alternative the reference keeps commented out (mnist.py:57)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    to_device = (lambda t: t) if args.cpu else (lambda t: t.hip())
    np.random.seed(0)
    model = (CNN() if args.cnn else NN()).map_parameters(to_device)
    rng = np.random.RandomState(1)
    batches = [synthetic_batch(rng, args.batch) for _ in range(8)]       # host-side data, re-uploaded every step
    losses = []

    if args.graph and not args.cpu:
        from lightgrad_amd import HipTensor
        from lightgrad_amd.autograd.hip import HipGraph, HipDevice
        optim = light.optim.AdaBelief(model.parameters(), lr=0.001, fused=True, device_step=True)
        x_static = HipTensor.from_numpy(batches[0][0])
        t_static = HipTensor.from_numpy(batches[0][1])

        def step():
            l = light.loss.mse(model(x_static), t_static)
            optim.zero_grad()
            l.backward()
            optim.step()
            return l
        for _ in range(3):
            step()                                                       # eager warm-up
        graph = HipGraph()
        with graph.capture():
            loss = step()
        optim.t -= len(optim.parameters)
        HipDevice.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            x, one_hot = batches[i % len(batches)]
            x_static.upload_(x)                                          # asynchronous, but on the compute stream
            t_static.upload_(one_hot)
            graph.replay()
            optim.on_graph_replay()
        final = loss.item()                                              # the only synchronisation
        dt = time.perf_counter() - t0
        print("graph replay + per-step upload: final loss %.5f   %.1f steps/s (PCIe-inclusive, upload and compute in one stream)" % (final, args.steps / dt))

        return

    optim = light.optim.AdaBelief(model.parameters(), lr=0.001, fused=not args.cpu)
    with Profiler() as prof:
        t0 = time.perf_counter()
        for i in range(args.steps):
            x, one_hot = batches[i % len(batches)]
            y = model(to_device(light.from_numpy(x)))
            if args.loss == "ce":
                labels = light.from_numpy(one_hot.argmax(-1).astype(np.int64), requires_grad=False)
                l = light.loss.cross_entropy(y, to_device(labels))
            else:
                l = light.loss.mse(y, to_device(light.from_numpy(one_hot)))
            optim.zero_grad()
            l.backward()
            optim.step()
            losses.append(l.item())                                      # synchronises every step, like the reference loop
        dt = time.perf_counter() - t0
    print("loss %.5f -> %.5f   %.1f steps/s (eager tape, per-step upload and loss.item())" % (losses[0], losses[-1], args.steps / dt))
    prof.print(topn=12)


if __name__ == "__main__":
    main()
