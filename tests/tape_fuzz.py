"""Differential fuzzing of the tape: the same randomly drawn training program on CpuTensor (numpy, pinned to the reference) and on
another backend, every observable compared.

Why: the HIP backend answers many op sequences with something other than one kernel per op - a relu that never runs (folded into
the next Linear), a skinny Linear computed together with its loss, weight and input gradients sharing a launch, gradients added by
GEMM epilogues, a lazily zeroed gradient bucket, an optimizer update inside a hipGraph or inside the backward kernels.  Each of
these peepholes is tested on the sequence it was made for; what nobody writes by hand are the sequences in between (VERDICT r3,
Weak 10).  A program here is a random walk over exactly those ingredients: Linear layers of random sizes (some skinny enough for the
head kernels, some with no bias), relu / tanh / sigmoid, residual adds, scalar multiplies, reshapes and transposed views, reads of
intermediates in the middle of a forward pass (which make lazy tensors real), in-place writes into inputs that lazy tensors still
want to read, several losses, repeated backward passes with and without zero_grad, and one of the optimizers in one of its forms.

    run_program(T, seed, ...) -> {label: ndarray}          # everything a user could look at, in program order
    compare(cpu, other)                                     # raises AssertionError naming the first label that differs

Used by tests/test_hip_tape_fuzz.py (GPU) and tests/test_tape_fuzz_cpu.py (the generator itself: CpuTensor float32 against a
float64 run)."""
import numpy as np
import lightgrad_amd as light


class Net(light.nn.Module):
    def __init__(self, dims, biases, norms=None):
        light.nn.Module.__init__(self)
        self.layers = light.nn.ModuleList(*[light.nn.Linear(a, b, bias=bias) for a, b, bias in zip(dims[:-1], dims[1:], biases)])
        norms = norms or [False] * (len(dims) - 1)
        self.norms = light.nn.ModuleList(*[light.nn.LayerNorm(b) if on else light.nn.Module() for b, on in zip(dims[1:], norms)])


def draw_program(seed):
    """the random choices of one program, independent of the backend"""
    rng = np.random.RandomState(seed)
    deep = seed % 4 == 3                             # every fourth program is deep enough for the queued weight gradients (GradGroup)
    depth = int(rng.randint(5, 8)) if deep else int(rng.randint(1, 4))
    batch = int(rng.choice([1, 3, 8, 32, 33, 64, 200]))
    dims = [int(rng.choice([4, 8, 12, 20, 48, 64, 100]))]
    for k in range(depth):
        last = k == depth - 1
        dims.append(int(rng.choice([1, 2, 5, 10, 16]) if (last and rng.rand() < 0.6) else rng.choice([4, 8, 24, 32, 48, 96])))
    prog = {
        "batch": batch, "dims": dims, "biases": [bool(rng.rand() < 0.8) for _ in range(depth)],
        "acts": [str(rng.choice(["relu", "relu", "tanh", "sigmoid", "none"])) for _ in range(depth)],
        "residual": [bool(rng.rand() < 0.3) for _ in range(depth)],
        "scale": [float(rng.choice([1.0, 0.5, -2.0])) if rng.rand() < 0.3 else None for _ in range(depth)],
        "peek": [bool(rng.rand() < 0.25) for _ in range(depth)],           # read an intermediate mid-forward
        "poke_input": [bool(rng.rand() < 0.15) for _ in range(depth)],     # write into the input in place mid-forward
        "view_in": str(rng.choice(["plain", "plain", "reshape", "transposed"])),
        "norm": [bool(deep and rng.rand() < 0.5 and dims[k + 1] >= 4) for k in range(depth)],      # nn.LayerNorm behind the layer
        "loss": str(rng.choice(["mse", "mse", "sum", "weighted", "ce"])) if dims[-1] >= 2 else "mse",
        "optimizer": str(rng.choice(["sgd", "adam", "adabelief", "adabelief"])),
        "fused": bool(rng.rand() < 0.6), "device_step": bool(rng.rand() < 0.5),
        "steps": int(rng.randint(1, 4)),
        "zero_grad": str(rng.choice(["before_backward", "before_backward", "after_step", "never"])),
        "second_backward": bool(rng.rand() < 0.25),                        # two backward passes over two forward passes, gradients add up
        "x_requires_grad": bool(rng.rand() < 0.7),
        "seed": int(seed),
    }
    if prog["norm"][-1] and prog["loss"] == "sum":
        # the sum over a freshly normalised row is 0 in exact arithmetic: the loss and every gradient behind it would be rounding
        # noise (which AdaBelief then normalises into steps of size lr) - nothing two correct backends have to agree on
        prog["loss"] = "weighted"
    return prog


def run_program(T, prog, dtype=np.float32, prepare=None):
    """run `prog` on tensor class T; returns the ordered list of (label, array) of everything observable.
    prepare(model, optimizer_factory) -> optimizer lets a backend test choose an optimizer form (flat buckets, ...)"""
    rng = np.random.RandomState(1000 + prog["seed"])
    batch, dims = prog["batch"], prog["dims"]
    np.random.seed(prog["seed"])                     # nn.Linear draws its weights from numpy's global stream
    model = Net(dims, prog["biases"], prog["norm"])
    w0 = [(n, p.numpy().astype(dtype)) for n, p in model.named_parameters()]
    model.load_parameters(w0)
    if T is not light.CpuTensor:
        model.map_parameters(lambda p: getattr(p, T.__module__.split(".")[-2])())        # p.hip()
    x_np = rng.uniform(-1, 1, (batch, dims[0])).astype(dtype)
    target_np = rng.uniform(-1, 1, (batch, dims[-1])).astype(dtype)
    w_np = rng.uniform(-1, 1, (batch, dims[-1])).astype(dtype)
    poke_np = rng.uniform(-1, 1, (batch, dims[0])).astype(dtype)
    labels_np = rng.randint(0, dims[-1], batch).astype(np.int64)

    def make_opt(params):
        kind = prog["optimizer"]
        if kind == "sgd":
            return light.optim.SGD(params, lr=1e-2, momentum=0.9)
        cls = light.optim.Adam if kind == "adam" else light.optim.AdaBelief
        kw = {}
        if T is not light.CpuTensor:
            kw = {"fused": prog["fused"], "device_step": prog["fused"] and prog["device_step"]}
        return cls(params, lr=1e-2, eps=1e-3, **kw)
    opt = prepare(model, make_opt) if prepare is not None else make_opt(model.parameters())
    out = []

    def see(label, t):
        out.append((label, np.array(t.numpy(), dtype=np.float64)))

    def forward(tag):
        if prog["view_in"] == "transposed":
            x = T.from_numpy(np.ascontiguousarray(x_np.T), requires_grad=prog["x_requires_grad"])
            h = x.transpose(1, 0)
        else:
            x = T.from_numpy(x_np.copy(), requires_grad=prog["x_requires_grad"])
            h = x.reshape(-1, dims[0]) if prog["view_in"] == "reshape" else x
        h_in = h
        for k, layer in enumerate(model.layers):
            prev = h
            h = layer(h)
            act = prog["acts"][k]
            if act != "none":
                h = getattr(h, act)()
            if prog["norm"][k]:
                h = model.norms[k](h)
            if prog["residual"][k] and prev.shape == h.shape:
                h = h + prev
            if prog["scale"][k] is not None:
                h = h * prog["scale"][k]
            if prog["peek"][k]:
                see("%s/h%d" % (tag, k), h)
            if prog["poke_input"][k] and prog["view_in"] == "plain":
                with light.no_grad():                # a later write into the input must not show in anything computed before it
                    x[...] = T.from_numpy(poke_np, requires_grad=False)
        return x, h_in, h

    def loss_of(h):
        if prog["loss"] == "mse":
            return light.loss.mse(h, T.from_numpy(target_np, requires_grad=False))
        if prog["loss"] == "sum":
            return h.sum()
        if prog["loss"] == "ce":
            return light.loss.cross_entropy(h, T.from_numpy(labels_np, requires_grad=False))
        return (h * T.from_numpy(w_np, requires_grad=False)).sum()

    for step in range(prog["steps"]):
        tag = "s%d" % step
        x, h_in, h = forward(tag)
        loss = loss_of(h)
        if prog["zero_grad"] == "before_backward":
            opt.zero_grad()
        loss.backward()
        if prog["second_backward"]:
            x2, _, h2 = forward(tag + "b")
            loss_of(h2).backward()                   # parameter gradients accumulate over the two passes
        see(tag + "/loss", loss)
        see(tag + "/out", h)
        for n, p in model.named_parameters():
            if p.grad is not None:
                see("%s/grad/%s" % (tag, n), p.grad)
        if prog["x_requires_grad"] and x.grad is not None:
            see(tag + "/grad/x", x.grad)
        opt.step()
        if prog["zero_grad"] == "after_step":
            opt.zero_grad()
        for n, p in model.named_parameters():
            see("%s/param/%s" % (tag, n), p)
    return out


def compare(ref, got, rtol=2e-4, atol=2e-5, what=""):
    assert [l for l, _ in ref] == [l for l, _ in got], "%s: the two runs looked at different things:\n%s\n%s" % (what, [l for l, _ in ref], [l for l, _ in got])
    for (label, a), (_, b) in zip(ref, got):
        assert a.shape == b.shape, "%s %s: shape %s vs %s" % (what, label, a.shape, b.shape)
        scale = max(1.0, float(np.abs(a).max()) if a.size else 1.0)
        if not np.allclose(a, b, rtol=rtol, atol=atol * scale, equal_nan=True):
            bad = np.abs(a - b) > atol * scale + rtol * np.abs(a)
            raise AssertionError("%s %s: %d of %d values differ, max |diff| %.3g at scale %.3g" % (what, label, int(bad.sum()), a.size, float(np.abs(a - b).max()), scale))
