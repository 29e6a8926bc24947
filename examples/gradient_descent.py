"""BASELINE config #1: plain gradient descent on a small expression (counterpart of the reference's
examples/gradient_descent.py).  Runs on the CPU backend by default, `--hip` moves it to the GPU."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lightgrad_amd as light  # noqa: E402


def run(to_device=lambda t: t, steps=100, lr=0.1, seed=1234):
    np.random.seed(seed)
    a, b, c = (to_device(light.uniform(-1, 1, shape=(10, 10))) for _ in range(3))
    ys = []
    for _ in range(steps):
        y = (a.tanh() + b.sigmoid()) @ (c.relu() - a.sigmoid())
        y.backward(allow_fill=True)          # start back-propagation from a non-item tensor
        with light.no_grad():
            a -= lr * a.grad
            b -= lr * b.grad
            c -= lr * c.grad
        y.zero_grad(traverse_graph=True)
        ys.append(y.sum().item())
    return ys


if __name__ == "__main__":
    ys = run((lambda t: t.hip()) if "--hip" in sys.argv else (lambda t: t))
    print("objective: first %.5f  last %.5f" % (ys[0], ys[-1]))
