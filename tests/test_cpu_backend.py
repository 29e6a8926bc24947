"""CpuTensor + autograd host against the reference: golden fixtures (values and gradients) and the
reference's own CPU test suite restated (test/test_cpu_tensor.py:12-56: gradchecks).  CPU only."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from lightgrad_amd.autograd import Function, AbstractTensor, Gradients
from lightgrad_amd.autograd.utils.gradcheck import gradcheck
from common import check_gradients, replay_op_cases
from conftest import load_golden
import np_oracle as O


def test_golden_ops_bitwise(golden_ops):
    """same numpy, same expressions as the reference => identical bits"""
    def check(name, kind, got, expected):
        np.testing.assert_array_equal(got, expected, err_msg="%s/%s" % (name, kind))
    assert replay_op_cases(CpuTensor, golden_ops, check) >= 85


def test_gradient_descent_example():
    """BASELINE config #1: examples/gradient_descent.py on the CPU backend"""
    g = load_golden("gradient_descent.npz")
    np.random.seed(1234)
    a, b, c = (light.uniform(-1, 1, shape=(10, 10)) for _ in range(3))
    np.testing.assert_array_equal(a.numpy(), g["a0"])
    ys = []
    for _ in range(100):
        y = (a.tanh() + b.sigmoid()) @ (c.relu() - a.sigmoid())
        y.backward(allow_fill=True)
        with light.no_grad():
            a -= 0.1 * a.grad
            b -= 0.1 * b.grad
            c -= 0.1 * c.grad
        y.zero_grad(traverse_graph=True)
        ys.append(y.sum().item())
    np.testing.assert_array_equal(np.asarray(ys), g["ys"])
    np.testing.assert_array_equal(a.numpy(), g["a_final"])


class MLP(light.nn.Module):
    def __init__(self, d_in, d_hid, d_out):
        light.nn.Module.__init__(self)
        self.d_in = d_in
        self.l1 = light.nn.Linear(d_in, d_hid)
        self.l2 = light.nn.Linear(d_hid, d_out)

    def forward(self, x):
        return self.l2(self.l1(x.reshape(-1, self.d_in)).relu())


def train(model, to_dev, x, onehot, steps, opt):
    losses, g0 = [], None
    for s in range(steps):
        l = light.loss.mse(model(to_dev(x)), to_dev(onehot))
        opt.zero_grad()
        l.backward()
        if s == 0:
            g0 = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
        opt.step()
        losses.append(l.item())
    return losses, g0


@pytest.mark.parametrize("opt_name", ["adabelief", "adam", "sgd"])
def test_mlp_small_trajectory_bitwise(opt_name):
    g = load_golden("mlp_small_%s.npz" % opt_name)
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    model = MLP(d_in, d_hid, d_out)
    model.load_parameters({n: g["w0/" + n] for n in O.PARAM_ORDER})
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), g["labels"]] = 1
    opt = {"adabelief": lambda p: light.optim.AdaBelief(p, lr=1e-3), "adam": lambda p: light.optim.Adam(p, lr=1e-3),
           "sgd": lambda p: light.optim.SGD(p, lr=1e-4, momentum=0.9)}[opt_name](model.parameters())
    losses, g0 = train(model, CpuTensor.from_numpy, g["x"], onehot, steps, opt)
    np.testing.assert_array_equal(np.asarray(losses), g["losses"])
    for n, p in model.named_parameters():
        np.testing.assert_array_equal(g0[n], g["g0/" + n])
        np.testing.assert_array_equal(p.numpy(), g["wf/" + n])


def test_mlp_full_size_first_steps():
    g = load_golden("mlp_full_adabelief.npz")
    d_in, d_hid, d_out, batch, steps, seed = (int(v) for v in g["config"])
    np.random.seed(seed)
    model = MLP(d_in, d_hid, d_out)                       # xavier init consumes the global RNG like the reference
    x = np.random.uniform(0, 1, size=(batch, d_in)).astype(np.float32)
    labels = np.random.randint(0, d_out, size=batch)
    np.testing.assert_array_equal(labels, g["labels"])
    onehot = np.zeros((batch, d_out), np.float32)
    onehot[np.arange(batch), labels] = 1
    losses, _ = train(model, CpuTensor.from_numpy, x, onehot, 2, light.optim.AdaBelief(model.parameters(), lr=1e-3))
    np.testing.assert_allclose(losses, g["losses"][:2], rtol=1e-6)


# ---- the reference's CPU gradcheck suite (test/test_cpu_tensor.py:15-56), same shapes / tolerances ----
np.random.seed(1234)
cpu_check = lambda *a, **k: check_gradients(CpuTensor, *a, **k)   # noqa: E731

GRADCHECKS = {
    "transpose": lambda: cpu_check(CpuTensor.transpose, shapes=[(45, 65)]),
    "reshape": lambda: cpu_check(lambda x: CpuTensor.reshape(x, -1), shapes=[(45, 65)]),
    "neg": lambda: cpu_check(CpuTensor.neg, shapes=[(10, 15)]),
    "sin": lambda: cpu_check(CpuTensor.sin, shapes=[(10, 15)]),
    "cos": lambda: cpu_check(CpuTensor.cos, shapes=[(10, 15)]),
    "exp": lambda: cpu_check(CpuTensor.exp, shapes=[(10, 15)]),
    "log": lambda: cpu_check(CpuTensor.log, shapes=[(10, 15)], lowhigh=(0.1, 10)),
    "sigmoid": lambda: cpu_check(CpuTensor.sigmoid, shapes=[(10, 15)]),
    "tanh": lambda: cpu_check(CpuTensor.tanh, shapes=[(10, 15)]),
    "relu": lambda: cpu_check(CpuTensor.relu, shapes=[(10, 15)], eps=1e-5, tol=0.002),
    "max": lambda: cpu_check(CpuTensor.max, shapes=[(10, 15)]),
    "min": lambda: cpu_check(CpuTensor.min, shapes=[(10, 15)]),
    "add": lambda: cpu_check(CpuTensor.add, shapes=[(10, 15), (10, 15)], broadcast=True),
    "sub": lambda: cpu_check(CpuTensor.sub, shapes=[(10, 15), (10, 15)], broadcast=True),
    "mul": lambda: cpu_check(CpuTensor.mul, shapes=[(10, 15), (10, 15)], broadcast=True),
    "pow": lambda: cpu_check(CpuTensor.pow, shapes=[(10, 15), (10, 15)], broadcast=True, lowhigh=(1, 2), eps=1e-5, tol=0.01),
    "dot": lambda: cpu_check(CpuTensor.dot, shapes=[(10, 15), (15, 10)]),
    "div+": lambda: cpu_check(CpuTensor.div, shapes=[(10, 15), (10, 15)], broadcast=True, lowhigh=(0.1, 10), tol=5e-3),
    "div-": lambda: cpu_check(CpuTensor.div, shapes=[(10, 15), (10, 15)], broadcast=True, lowhigh=(-10, -0.1), tol=5e-3),
    # not in the reference (its CPU sum has no backward): the gap this repo closes
    "sum": lambda: cpu_check(CpuTensor.sum, shapes=[(6, 7)]),
    "sum_axis0": lambda: cpu_check(lambda x: x.sum(axis=0), shapes=[(6, 7)]),
    "sum_axis1_keep": lambda: cpu_check(lambda x: x.sum(axis=1, keepdims=True), shapes=[(6, 7)]),
    "mean": lambda: cpu_check(lambda x: x.mean(axis=1), shapes=[(6, 7)]),
    "softmax": lambda: cpu_check(lambda x: x.softmax(axis=-1), shapes=[(5, 6)], tol=2e-3),
    "dot_batched": lambda: cpu_check(CpuTensor.dot, shapes=[(2, 4, 5), (2, 5, 3)], tol=2e-3),
    "dot_batch_bcast": lambda: cpu_check(CpuTensor.dot, shapes=[(2, 4, 5), (5, 3)], tol=2e-3),
}


@pytest.mark.parametrize("name", sorted(GRADCHECKS))
def test_gradcheck(name):
    np.random.seed(1234 + sum(map(ord, name)))
    GRADCHECKS[name]()


def test_gradcheck_linear_model():
    """test/test_cpu_tensor.py:43-55"""
    np.random.seed(1234)

    class Model(light.nn.Module):
        def __init__(self):
            light.nn.Module.__init__(self)
            self.l1 = light.nn.Linear(8, 16)
            self.l2 = light.nn.Linear(16, 4)

        def forward(self, x):
            return self.l2(self.l1(x).tanh())
    cpu_check(Model(), shapes=[(16, 8)])


def test_layernorm_gradcheck():
    np.random.seed(5)
    ln = light.nn.LayerNorm(6)
    cpu_check(lambda x: ln(x), shapes=[(4, 6)], tol=5e-3)


# ---- behaviours of the op-registration surface (tensor.py:136-161, func.py, grads.py) ----

def test_register_op_rules():
    class T1(CpuTensor):
        pass

    class myop(Function):
        def forward(ctx, a):
            return a
    T1.register_op()(myop)
    assert hasattr(T1, "myop")
    with pytest.raises(RuntimeError, match="already registered"):
        T1.register_op()(myop)
    T1.register_op(overwrite=True)(myop)
    with pytest.raises(TypeError, match="must inherit from Function"):
        T1.register_op("bad", int)


def test_backend_converters_exist():
    assert callable(AbstractTensor.cpu) and callable(AbstractTensor.hip)
    t = CpuTensor.from_numpy(np.arange(4, dtype=np.float32))
    np.testing.assert_array_equal(t.cpu().numpy(), t.numpy())


def test_backward_rules():
    t = CpuTensor.from_numpy(np.ones((2, 2), np.float32))
    y = t * 2
    with pytest.raises(RuntimeError, match="Can only backpropagate from item tensors"):
        y.backward()
    y.backward(allow_fill=True)
    np.testing.assert_array_equal(t.grad.numpy(), np.full((2, 2), 2, np.float32))
    z = t.__iadd__(1.0)
    with pytest.raises(RuntimeError, match="Cannot Backward through iadd"):
        z.backward(allow_fill=True)
    with light.no_grad():
        assert (t * 2).ctx is None
    assert Gradients._is_enabled()


def test_mixed_backends_rejected():
    class Other(CpuTensor):
        pass
    a = CpuTensor.from_numpy(np.ones(2, np.float32))
    b = Other(np.ones(2, np.float32))
    with pytest.raises(AssertionError, match="same type"):
        Other.add(b, a)


def test_diamond_graph_is_differentiated_correctly():
    """deliberate divergence from the reference's LIFO walk (grads.py:36):
    d/dx [sin(exp x) + exp x] = exp(x) * (cos(exp x) + 1)"""
    x = CpuTensor.from_numpy(np.array([0.1, 2.3], np.float32))
    h = x.exp()
    (h.sin() + h).backward(allow_fill=True)
    e = np.exp(np.array([0.1, 2.3]))
    np.testing.assert_allclose(x.grad.numpy(), e * (np.cos(e) + 1), rtol=1e-5)
    x2 = CpuTensor.from_numpy(np.array([0.1, 2.3], np.float32))
    h2 = x2.exp()
    (h2 + h2.sin()).backward(allow_fill=True)
    np.testing.assert_allclose(x2.grad.numpy(), x.grad.numpy(), rtol=1e-6)


def test_profiler_counts_ops():
    from lightgrad_amd.autograd.utils.profiler import Profiler
    a = CpuTensor.from_numpy(np.ones((3, 3), np.float32))
    with Profiler() as p:
        (a @ a).relu().backward(allow_fill=True)
    table = p.table()
    assert table["dot"][1] == 1 and table["dot"][3] == 1 and table["relu"][1] == 1


def test_module_parameter_plumbing():
    m, m2 = MLP(4, 3, 2), MLP(4, 3, 2)
    names = [n for n, _ in m.named_parameters()]
    assert names == list(O.PARAM_ORDER)
    m2.load_parameters(m.named_parameters())
    for (_, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        np.testing.assert_array_equal(p.numpy(), q.numpy())
    m2.map_parameters(lambda p: CpuTensor.from_numpy(p.numpy() * 0))
    assert all(np.all(p.numpy() == 0) for p in m2.parameters())
    assert gradcheck(lambda x: m(x), CpuTensor.uniform(-1, 1, (2, 4)))


def test_constants_of_the_tape_do_not_grow_nodes():
    """extension: an op whose inputs all have requires_grad=False returns a constant (no node, requires_grad False),
    so data tensors cost no backward work; every gradient that IS wanted is unchanged"""
    rng = np.random.RandomState(0)
    x_np = rng.uniform(0, 1, (6, 2, 4)).astype(np.float32)
    target = rng.uniform(0, 1, (6, 3)).astype(np.float32)
    grads = []
    for needs in (True, False):
        np.random.seed(3)
        model = light.nn.Linear(8, 3)
        x = CpuTensor.from_numpy(x_np, requires_grad=needs)
        flat = x.reshape(-1, 8)
        assert flat.requires_grad == needs and (flat.ctx is not None) == needs
        loss = light.loss.mse(model(flat), CpuTensor.from_numpy(target, requires_grad=False))
        loss.backward()
        assert (x.grad is not None) == needs
        grads.append([p.grad.numpy().copy() for p in model.parameters()])
    for a, b in zip(*grads):
        np.testing.assert_array_equal(a, b)
    c = CpuTensor.from_numpy(x_np, requires_grad=False) * 2.0 + 1.0
    assert c.ctx is None and not c.requires_grad
    c.backward(allow_fill=True)                                    # nothing to do, like backward on a leaf
    assert c.grad is None


def test_linear_residual_argument_is_the_plain_sum():
    """nn.Linear.forward(x, residual=r) (extension used by examples/bert.py) == Linear(x) + r, values and gradients, bit for bit"""
    rng = np.random.RandomState(12)
    x, r, g = (rng.uniform(-1, 1, (3, 5, 8)).astype(np.float32) for _ in range(3))
    lin = light.nn.Linear(8, 8)
    results = []
    for fused in (True, False):
        for p in lin.parameters():
            p.zero_grad()
        tx, tr = CpuTensor.from_numpy(x), CpuTensor.from_numpy(r)
        y = lin(tx, residual=tr) if fused else lin(tx) + tr
        (y * CpuTensor.from_numpy(g, requires_grad=False)).backward(allow_fill=True)
        results.append([y.numpy().copy(), tx.grad.numpy().copy(), tr.grad.numpy().copy()] + [p.grad.numpy().copy() for p in lin.parameters()])
    for a, b in zip(*results):
        np.testing.assert_array_equal(a, b)
