"""Which op carries the error of the tiny-BERT query / key gradients?  Runs the same forward + backward on the fp32 CPU
backend, on HipTensor and on the CPU backend in float64 (the yardstick), keeps the intermediates of every self-attention
block, and prints the relative Frobenius error of each intermediate's value and gradient against float64.
    python tools/bert_grad_probe.py"""
import math
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lightgrad_amd as light                                        # noqa: E402
from lightgrad_amd import CpuTensor, HipTensor                        # noqa: E402
from test_bert_cpu import bert, build_tiny                            # noqa: E402

stash = []


def forward(self, hidden, attention_mask=None):
    b, s, _ = hidden.shape
    q = self.query(hidden).reshape(b, s, self.h, self.d).transpose(0, 2, 1, 3)
    k = self.key(hidden).reshape(b, s, self.h, self.d).transpose(0, 2, 3, 1)
    v = self.value(hidden).reshape(b, s, self.h, self.d).transpose(0, 2, 1, 3)
    raw = q @ k
    scores = raw / math.sqrt(self.d)
    probs = scores.softmax(axis=-1)
    ctx = probs @ v
    context = ctx.transpose(0, 2, 1, 3).reshape(b, s, self.h * self.d)
    stash.append({"hidden": hidden, "q": q, "k": k, "v": v, "raw": raw, "scores": scores, "probs": probs, "ctx": ctx})
    return context, probs


bert.BertSelfAttention.forward = forward
g = np.load(os.path.join(ROOT, "tests", "golden", "bert_tiny_forward.npz"))
rng = np.random.RandomState(0)
w = rng.uniform(-1, 1, (2, 128, 30522)).astype(np.float32)
runs = {}
cpu_model = build_tiny()
values = {n: p.numpy().astype(np.float64) for n, p in cpu_model.named_parameters()}
for tag in ("cpu32", "hip", "f64"):
    stash.clear()
    if tag == "cpu32":
        model, T, ww = cpu_model, CpuTensor, w
    elif tag == "hip":
        model, T, ww = build_tiny().map_parameters(lambda p: p.hip()), HipTensor, w
    else:
        CpuTensor.default_dtype = np.float64
        model, T, ww = build_tiny(), CpuTensor, w.astype(np.float64)
        model.load_parameters(values)
    logits = model(T.from_numpy(g["ids"], requires_grad=False))
    (logits * T.from_numpy(ww, requires_grad=False)).backward(allow_fill=True)
    runs[tag] = [{k: (t.numpy().astype(np.float64), None if t.grad is None else t.grad.numpy().astype(np.float64)) for k, t in layer.items()} for layer in stash]
    runs[tag + "_params"] = {n: p.grad.numpy().astype(np.float64) for n, p in model.named_parameters()}
CpuTensor.default_dtype = np.float32


def rel(a, b):
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


for li in range(len(runs["f64"])):
    print("self-attention of layer %d: relative Frobenius error against float64 (fp32 CPU backend | HIP), ||float64 value||" % li)
    for k in ("hidden", "q", "k", "v", "raw", "scores", "probs", "ctx"):
        ref_v, ref_g = runs["f64"][li][k]
        line = "  %-7s value %.1e | %.1e  (%.2e)" % (k, rel(runs["cpu32"][li][k][0], ref_v), rel(runs["hip"][li][k][0], ref_v), np.linalg.norm(ref_v))
        if ref_g is not None and runs["hip"][li][k][1] is not None and runs["cpu32"][li][k][1] is not None:
            line += "    grad %.1e | %.1e  (%.2e)" % (rel(runs["cpu32"][li][k][1], ref_g), rel(runs["hip"][li][k][1], ref_g), np.linalg.norm(ref_g))
        print(line)
for n in sorted(runs["f64_params"]):
    if ".query." in n or ".key." in n or ".value." in n:
        r = runs["f64_params"][n]
        print("%-60s %.1e | %.1e  (%.2e)" % (n, rel(runs["cpu32_params"][n], r), rel(runs["hip_params"][n], r), np.linalg.norm(r)))

# isolate the GEMM that makes dq: same inputs (HIP's own d(raw) and k), three evaluations
print("\ndq = d(raw) @ k^T recomputed from HIP's own operands (relative Frobenius error against float64 of the SAME operands):")
for li in range(len(runs["f64"])):
    draw = runs["hip"][li]["raw"][1]                 # (b, h, s, s) float64 copies of HIP's fp32 values
    k = runs["hip"][li]["k"][0]                      # (b, h, d, s)
    exact = draw @ np.swapaxes(k, -1, -2)
    np32 = (draw.astype(np.float32) @ np.swapaxes(k, -1, -2).astype(np.float32)).astype(np.float64)
    td, tk = HipTensor.from_numpy(draw.astype(np.float32), requires_grad=False), HipTensor.from_numpy(np.ascontiguousarray(np.swapaxes(k, -1, -2)).astype(np.float32), requires_grad=False)
    dev = (td @ tk).numpy().astype(np.float64)
    cond = np.linalg.norm(np.abs(draw) @ np.abs(np.swapaxes(k, -1, -2))) / np.linalg.norm(exact)
    print("  layer %d: numpy fp32 %.1e | HIP GEMM (dense operands) %.1e | tape's q.grad %.1e   (condition |A||B|/|AB| = %.0f -> fp32 bound ~%.0e)"
          % (li, rel(np32, exact), rel(dev, exact), rel(runs["hip"][li]["q"][1], exact), cond, 6e-8 * cond))
