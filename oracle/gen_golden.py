"""Generate tests/golden/*.npz from the REAL reference (ndoll1998/lightgrad @ /root/reference).

TEST INFRASTRUCTURE - runs only in the build container (the reference never travels to the
GPU box).  It imports the reference's CPU backend with the pyopencl stub described in
SURVEY.md §8c (the reference's own `import pyopencl` fails with an ordinary ModuleNotFoundError
here), evaluates the hot-path ops forward + backward on seeded inputs and stores inputs and
outputs as small fixtures.  The fixtures pin both the numpy oracle (oracle/np_oracle.py) and
the product backends (CpuTensor on CPU, HipTensor on the GPU).

    python oracle/gen_golden.py            # rewrites tests/golden/

Only DATA is written (inputs / expected outputs); no reference source text is copied.
"""
import os
import sys
from unittest import mock

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import f32  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    cl = mock.MagicMock(name="pyopencl")
    cl.__path__ = []

    def _none():
        raise RuntimeError("no opencl")      # swallowed by the reference's bare except (opencl/device.py:16-22)
    cl.get_platforms = _none
    tools = mock.MagicMock(name="pyopencl.tools")
    cl.tools = tools
    sys.modules["pyopencl"], sys.modules["pyopencl.tools"] = cl, tools
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import lightgrad as light               # noqa: E402
    return light


def main():
    light = import_reference()
    T = light.CpuTensor
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(20261003)

    # ---------------------------------------------------------------- elementwise / unary
    cases = {}

    def run(name, fn, inputs, upstream=True):
        """forward + backward of fn(*tensors) with upstream gradient w (via `(y * w).backward(allow_fill)`)"""
        ts = [T.from_numpy(a.copy()) for a in inputs]
        y = fn(*ts)
        rec = {"in%d" % i: a for i, a in enumerate(inputs)}
        rec["out"] = np.array(y.numpy())
        if upstream:
            w = f32(rng, -1, 1, y.shape)
            rec["w"] = w
            (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
            for i, t in enumerate(ts):
                if t.grad is not None:
                    rec["grad%d" % i] = np.array(t.grad.numpy())
        for k, v in rec.items():
            cases["%s/%s" % (name, k)] = v

    from cases import op_cases
    for case in op_cases(T, rng):
        run(*case)

    np.savez_compressed(os.path.join(OUT, "ops.npz"), **cases)

    # ---------------------------------------------------------------- examples/gradient_descent.py trajectory
    # the script body, seeded as in SURVEY.md §8a row a13 (np.random.seed(1234) right before it)
    np.random.seed(1234)
    A = light.uniform(-1, 1, shape=(10, 10))
    Bt = light.uniform(-1, 1, shape=(10, 10))
    Ct = light.uniform(-1, 1, shape=(10, 10))
    init = [A.numpy().copy(), Bt.numpy().copy(), Ct.numpy().copy()]
    ys = []
    for _ in range(100):
        y = (A.tanh() + Bt.sigmoid()) @ (Ct.relu() - A.sigmoid())
        y.backward(allow_fill=True)
        with light.no_grad():
            A -= 0.1 * A.grad
            Bt -= 0.1 * Bt.grad
            Ct -= 0.1 * Ct.grad
        y.zero_grad(traverse_graph=True)
        ys.append(y.sum().item())
    np.savez_compressed(os.path.join(OUT, "gradient_descent.npz"), ys=np.asarray(ys, dtype=np.float64),
                        a0=init[0], b0=init[1], c0=init[2],
                        a_final=A.numpy(), b_final=Bt.numpy(), c_final=Ct.numpy())

    # ---------------------------------------------------------------- MLP training trajectories
    import lightgrad.nn as nn

    def mlp_traj(d_in, d_hid, d_out, batch, steps, opt_name, seed, store_inputs):
        class MLP(nn.Module):
            def __init__(self):
                nn.Module.__init__(self)
                self.l1 = nn.Linear(d_in, d_hid)
                self.l2 = nn.Linear(d_hid, d_out)

            def forward(self, xx):
                return self.l2(self.l1(xx.reshape(-1, d_in)).relu())
        np.random.seed(seed)
        model = MLP()
        w0 = {n: p.numpy().copy() for n, p in model.named_parameters()}
        xb = np.random.uniform(0, 1, size=(batch, d_in)).astype(np.float32)
        labels = np.random.randint(0, d_out, size=batch)
        onehot = np.zeros((batch, d_out), dtype=np.float32)
        onehot[np.arange(batch), labels] = 1
        if opt_name == "adabelief":
            opt = light.optim.AdaBelief(model.parameters(), lr=1e-3)
        elif opt_name == "adam":
            opt = light.optim.Adam(model.parameters(), lr=1e-3)
        else:
            opt = light.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
        losses, grads0 = [], None
        for s in range(steps):
            yy = model(T.from_numpy(xb))
            l = light.loss.mse(yy, T.from_numpy(onehot))
            opt.zero_grad()
            l.backward()
            if s == 0:
                grads0 = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
            opt.step()
            losses.append(l.item())
        rec = {"losses": np.asarray(losses, dtype=np.float64), "labels": labels.astype(np.int64),
               "config": np.asarray([d_in, d_hid, d_out, batch, steps, seed], dtype=np.int64)}
        wf = {n: p.numpy() for n, p in model.named_parameters()}
        if store_inputs:
            rec["x"] = xb
            for n in w0:
                rec["w0/" + n] = w0[n]
                rec["wf/" + n] = wf[n]
                rec["g0/" + n] = grads0[n]
        else:
            # full-size config: inputs are regenerated from the seed; keep digests + a strided sample of outputs
            for n in w0:
                rec["w0sum/" + n] = np.asarray([w0[n].astype(np.float64).sum(), np.abs(w0[n]).astype(np.float64).sum()])
                rec["wfsum/" + n] = np.asarray([wf[n].astype(np.float64).sum(), np.abs(wf[n]).astype(np.float64).sum()])
                rec["wfsample/" + n] = wf[n].reshape(-1)[::max(1, wf[n].size // 64)][:64].copy()
                rec["g0sample/" + n] = grads0[n].reshape(-1)[::max(1, grads0[n].size // 64)][:64].copy()
        return rec

    np.savez_compressed(os.path.join(OUT, "mlp_small_adabelief.npz"), **mlp_traj(20, 16, 10, 8, 12, "adabelief", 7, True))
    np.savez_compressed(os.path.join(OUT, "mlp_small_adam.npz"), **mlp_traj(20, 16, 10, 8, 12, "adam", 8, True))
    np.savez_compressed(os.path.join(OUT, "mlp_small_sgd.npz"), **mlp_traj(20, 16, 10, 8, 12, "sgd", 9, True))
    np.savez_compressed(os.path.join(OUT, "mlp_full_adabelief.npz"), **mlp_traj(784, 512, 10, 1024, 5, "adabelief", 0, False))
    # ---------------------------------------------------------------- CNN ops: conv (all dims / strides), pad, pooling
    cnn = {}
    rng2 = np.random.RandomState(77)

    def run_cnn(name, fn, inputs):
        ts = [T.from_numpy(a.copy()) for a in inputs]
        y = fn(*ts)
        w = f32(rng2, -1, 1, y.shape)
        (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        cnn[name + "/out"], cnn[name + "/w"] = np.array(y.numpy()), w
        for i, (a, t) in enumerate(zip(inputs, ts)):
            cnn["%s/in%d" % (name, i)] = a
            cnn["%s/grad%d" % (name, i)] = np.array(t.grad.numpy())
    for dim in (1, 2, 3):
        for size, k, stride, cin, cout in [(6, 3, 1, 2, 3), (9, 3, 2, 1, 2), (9, 5, 3, 3, 1), (7, 7, 1, 2, 2)]:
            if dim == 3 and size == 9:
                size = 7 if k <= 3 else 9
            if k > size:
                continue
            xin = f32(rng2, -1, 1, (2, cin) + (size,) * dim)
            kin = f32(rng2, -1, 1, (cout, cin) + (k,) * dim)
            run_cnn("conv%dd_s%d_k%d_st%d_c%d_%d" % (dim, size, k, stride, cin, cout), lambda a, b, st=stride: a.conv(b, strides=st), [xin, kin])
    img = f32(rng2, -1, 1, (2, 3, 7, 6))
    run_cnn("pad2", lambda a: a.pad(2), [img])
    run_cnn("pad_1_3", lambda a: a.pad((1, 3), value=0.5), [img])
    run_cnn("max_pool", lambda a: a.max_pool(), [img])
    run_cnn("min_pool_3x2", lambda a: a.min_pool(kernel=(3, 2)), [img])
    run_cnn("max_pool_2x3", lambda a: a.max_pool(kernel=(2, 3)), [img])
    np.savez_compressed(os.path.join(OUT, "cnn_ops.npz"), **cnn)

    # ---------------------------------------------------------------- loss.cross_entropy (loss.py:14-24), labels int64/int32/int16
    from lightgrad.loss import cross_entropy as ref_cross_entropy
    ce = {}
    rng3 = np.random.RandomState(99)
    for name, (n, c), ldt in [("n8_c10_i64", (8, 10), np.int64), ("n5_c3_i32", (5, 3), np.int32),
                              ("n33_c130_i16", (33, 130), np.int16), ("n1_c1000_i64", (1, 1000), np.int64)]:
        logits = f32(rng3, -4, 4, (n, c))
        labels = rng3.randint(0, c, size=n).astype(ldt)
        up = f32(rng3, 0.5, 2, ())
        y = T.from_numpy(logits.copy())
        loss = ref_cross_entropy(y, T.from_numpy(labels, requires_grad=False))
        (loss * T.from_numpy(up, requires_grad=False)).backward(allow_fill=True)
        ce[name + "/logits"], ce[name + "/labels"], ce[name + "/w"] = logits, labels, up
        ce[name + "/loss"], ce[name + "/grad"] = np.array(loss.numpy()), np.array(y.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "cross_entropy.npz"), **ce)

    # ---------------------------------------------------------------- fancy indexing: integer index on one
    # axis, the (arange, labels) pair of loss.cross_entropy (loss.py:19, :22), Dataset batching (data.py:15-32)
    fi = {}
    rng4 = np.random.RandomState(4242)

    def run_take(name, shape, index_fn, idx_arrays):
        a = f32(rng4, -1, 1, shape)
        t = T.from_numpy(a.copy())
        y = index_fn(t)
        w = f32(rng4, -1, 1, y.shape)
        (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        fi[name + "/in"], fi[name + "/out"], fi[name + "/w"], fi[name + "/grad"] = a, np.array(y.numpy()), w, np.array(t.grad.numpy())
        for k, v in idx_arrays.items():
            fi["%s/%s" % (name, k)] = v
    i0 = rng4.permutation(9)[:6].astype(np.int64)
    run_take("take_axis0", (9, 5), lambda t: t[T.from_numpy(i0, requires_grad=False)], {"idx": i0})
    i1 = rng4.permutation(7)[:4].astype(np.int32)
    run_take("take_axis1", (4, 7), lambda t: t[:, T.from_numpy(i1, requires_grad=False)], {"idx": i1})
    i2 = rng4.permutation(6).reshape(2, 3).astype(np.int64)
    run_take("take_axis0_2d_index", (6, 3, 4), lambda t: t[T.from_numpy(i2, requires_grad=False)], {"idx": i2})
    i3 = rng4.permutation(8)[:5].astype(np.int16)
    run_take("take_middle_axis", (3, 8, 2), lambda t: t[:, T.from_numpy(i3, requires_grad=False), :], {"idx": i3})
    i4 = np.asarray([-1, 0, -3], np.int64)
    run_take("take_negative", (5, 4), lambda t: t[T.from_numpy(i4, requires_grad=False)], {"idx": i4})
    lab = rng4.randint(0, 10, size=8).astype(np.int64)
    run_take("pair_rows_labels", (8, 10), lambda t: t[range(8), T.from_numpy(lab, requires_grad=False)], {"labels": lab})
    lab16 = rng4.randint(0, 3, size=5).astype(np.int16)
    run_take("pair_rows_labels_i16", (5, 3), lambda t: t[range(5), T.from_numpy(lab16, requires_grad=False)], {"labels": lab16})
    # in-place forms (no gradient): y[range, labels] -= 1 (loss.py:22) and a[idx] = v on one axis
    a = f32(rng4, -1, 1, (8, 10))
    t = T.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[range(8), T.from_numpy(lab, requires_grad=False)] -= 1
    fi["pair_isub/in"], fi["pair_isub/labels"], fi["pair_isub/out"] = a, lab, np.array(t.numpy())
    a = f32(rng4, -1, 1, (9, 5))
    v = f32(rng4, -1, 1, (6, 5))
    t = T.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[T.from_numpy(i0, requires_grad=False)] = T.from_numpy(v, requires_grad=False)
    fi["put_axis0/in"], fi["put_axis0/idx"], fi["put_axis0/val"], fi["put_axis0/out"] = a, i0, v, np.array(t.numpy())
    # Dataset: shuffle (np.random.permutation -> t[perm]) and batch slices t[i*bs:(i+1)*bs, ...]
    from lightgrad.data import Dataset as RefDataset
    X = f32(rng4, 0, 1, (20, 3, 2))
    Y = rng4.randint(0, 10, size=20).astype(np.int16)
    np.random.seed(31)
    ds = RefDataset((T.from_numpy(X, requires_grad=False), T.from_numpy(Y, requires_grad=False)), shuffle=True, batchsize=8)
    batches = [(np.array(bx.numpy()), np.array(by.numpy())) for bx, by in ds]
    fi["dataset/X"], fi["dataset/Y"], fi["dataset/seed"], fi["dataset/n_batches"] = X, Y, np.asarray(31), np.asarray(len(batches))
    for k, (bx, by) in enumerate(batches):
        fi["dataset/x%d" % k], fi["dataset/y%d" % k] = bx, by
    np.savez_compressed(os.path.join(OUT, "fancy_index.npz"), **fi)

    # ---------------------------------------------------------------- more index forms the reference's CPU path accepts through numpy
    # (cpu/ops.py:234-255): several index arrays on neighbouring axes (broadcast together), boolean masks.  Unique index tuples:
    # the reference ASSIGNS in getitem.backward (cpu/ops.py:245), which equals accumulation only without repeats.
    fi2 = {}
    rng5 = np.random.RandomState(777)

    def run_multi(name, shape, make_index, arrays):
        a = f32(rng5, -1, 1, shape)
        t = T.from_numpy(a.copy())
        y = t[make_index(lambda v: T.from_numpy(v, requires_grad=False))]
        w = f32(rng5, -1, 1, y.shape)
        (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        fi2[name + "/in"], fi2[name + "/out"], fi2[name + "/w"], fi2[name + "/grad"] = a, np.array(y.numpy()), w, np.array(t.grad.numpy())
        for k, v in arrays.items():
            fi2["%s/%s" % (name, k)] = v
    flat = rng5.permutation(6 * 5)[:7]
    r2, c2 = (flat // 5).astype(np.int64), (flat % 5).astype(np.int64)
    run_multi("two_arrays", (6, 5, 3), lambda D: (D(r2), D(c2)), {"i0": r2, "i1": c2})
    flat = rng5.permutation(4 * 3 * 5)[:9]
    a3, b3, c3 = (flat // 15).astype(np.int64), ((flat // 5) % 3).astype(np.int32), (flat % 5).astype(np.int64)
    run_multi("three_arrays", (4, 3, 5), lambda D: (D(a3), D(b3), D(c3)), {"i0": a3, "i1": b3, "i2": c3})
    rb, cb = rng5.permutation(6)[:3].reshape(3, 1).astype(np.int64), rng5.permutation(7)[:4].astype(np.int64)
    run_multi("two_arrays_broadcast", (6, 7), lambda D: (D(rb), D(cb)), {"i0": rb, "i1": cb})
    rm, cm = rng5.permutation(5)[:4].astype(np.int64), rng5.permutation(6)[:4].astype(np.int64)
    run_multi("two_arrays_middle", (2, 5, 6, 3), lambda D: (slice(None), D(rm), D(cm)), {"i0": rm, "i1": cm})
    m1 = rng5.uniform(0, 1, 8) > 0.5
    m1[0] = True
    run_multi("mask_axis0", (8, 4), lambda D: D(m1), {"mask": m1})
    m2 = rng5.uniform(0, 1, (5, 6)) > 0.6
    m2[2, 3] = True
    run_multi("mask_full", (5, 6), lambda D: D(m2), {"mask": m2})
    run_multi("mask_two_axes_of_three", (5, 6, 2), lambda D: D(m2), {"mask": m2})
    # in-place forms
    a = f32(rng5, -1, 1, (6, 5, 3))
    v = f32(rng5, -1, 1, (7, 3))
    t = T.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[T.from_numpy(r2, requires_grad=False), T.from_numpy(c2, requires_grad=False)] = T.from_numpy(v, requires_grad=False)
    fi2["put_two_arrays/in"], fi2["put_two_arrays/i0"], fi2["put_two_arrays/i1"], fi2["put_two_arrays/val"], fi2["put_two_arrays/out"] = a, r2, c2, v, np.array(t.numpy())
    a = f32(rng5, -1, 1, (5, 6))
    t = T.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[T.from_numpy(m2, requires_grad=False)] = 0.25
    fi2["put_mask_scalar/in"], fi2["put_mask_scalar/mask"], fi2["put_mask_scalar/out"] = a, m2, np.array(t.numpy())
    np.savez_compressed(os.path.join(OUT, "fancy_index_multi.npz"), **fi2)

    # ---------------------------------------------------------------- round 4: index arrays on NON-neighbouring axes (numpy moves
    # the index dimensions to the front), plain integers among index arrays (numpy counts them as index arrays for that rule), a
    # mask next to an index array - all through the reference's CPU path (cpu/ops.py:234-255).  Unique index tuples (see above).
    fi3 = {}
    rng6 = np.random.RandomState(4242)

    def run_apart(name, shape, make_index, arrays):
        a = f32(rng6, -1, 1, shape)
        t = T.from_numpy(a.copy())
        y = t[make_index(lambda v: T.from_numpy(v, requires_grad=False))]
        w = f32(rng6, -1, 1, y.shape)
        (y * T.from_numpy(w, requires_grad=False)).backward(allow_fill=True)
        fi3[name + "/in"], fi3[name + "/out"], fi3[name + "/w"], fi3[name + "/grad"] = a, np.array(y.numpy()), w, np.array(t.grad.numpy())
        for k, v in arrays.items():
            fi3["%s/%s" % (name, k)] = v
    flat = rng6.permutation(4 * 6)[:5]
    ia, ib = (flat // 6).astype(np.int64), (flat % 6).astype(np.int32)
    run_apart("arrays_apart", (4, 5, 6), lambda D: (D(ia), slice(None), D(ib)), {"i0": ia, "i1": ib})
    run_apart("arrays_apart_4d", (4, 3, 6, 2), lambda D: (D(ia), slice(1, 3), D(ib)), {"i0": ia, "i1": ib})
    ic = rng6.permutation(6)[:4].astype(np.int64)
    run_apart("int_and_array_apart", (4, 5, 6), lambda D: (2, slice(None), D(ic)), {"i0": ic})
    run_apart("int_next_to_array", (4, 5, 6), lambda D: (slice(None), 3, D(ic)), {"i0": ic})
    run_apart("array_int_array", (4, 5, 6), lambda D: (D(ia), -2, D(ib)), {"i0": ia, "i1": ib})
    icol = rng6.permutation(4)[:3].reshape(3, 1).astype(np.int64)
    irow = rng6.permutation(6)[:2].astype(np.int64)
    run_apart("arrays_apart_broadcast", (4, 5, 6), lambda D: (D(icol), slice(None, None, 2), D(irow)), {"i0": icol, "i1": irow})
    mk = rng6.uniform(0, 1, 5) > 0.4
    mk[1] = True
    ik = rng6.permutation(6)[:int(mk.sum())].astype(np.int64)
    run_apart("mask_and_array", (4, 5, 6), lambda D: (slice(None), D(mk), D(ik)), {"mask": mk, "i0": ik})
    run_apart("mask_apart_from_array", (5, 3, 6), lambda D: (D(mk), slice(None), D(ik)), {"mask": mk, "i0": ik})
    a = f32(rng6, -1, 1, (4, 5, 6))
    v = f32(rng6, -1, 1, (5, 5))
    t = T.from_numpy(a.copy(), requires_grad=False)
    with light.no_grad():
        t[T.from_numpy(ia, requires_grad=False), :, T.from_numpy(ib, requires_grad=False)] = T.from_numpy(v, requires_grad=False)
    fi3["put_arrays_apart/in"], fi3["put_arrays_apart/i0"], fi3["put_arrays_apart/i1"], fi3["put_arrays_apart/val"], fi3["put_arrays_apart/out"] = a, ia, ib, v, np.array(t.numpy())
    np.savez_compressed(os.path.join(OUT, "fancy_index_apart.npz"), **fi3)

    # ---------------------------------------------------------------- tensors that are not float32 (round 4): what the reference's
    # CPU backend - numpy - computes for int16 / int32 / int64 / float64 operands (cpu/tensor.py:45-46 keeps the dtype;
    # cpu/ops.py:52-84, :120-139, :260-293): neg, add, sub, mul (+ scalar operands, broadcasting, a transposed view), the
    # in-place forms, sum / max / min over None / 0 / 1, and for float64 div, pow and the gradients of a small expression
    ty = {}
    rng = np.random.RandomState(2026)
    for dt in (np.int16, np.int32, np.int64, np.float64):
        name = np.dtype(dt).name
        if dt is np.float64:
            a, b, row = rng.uniform(-3, 3, (7, 9)), rng.uniform(0.5, 3, (7, 9)), rng.uniform(-2, 2, (9,))
        else:
            info = np.iinfo(dt)
            a = rng.randint(info.min // 2, info.max // 2, (7, 9)).astype(dt)
            b = rng.randint(info.min // 2, info.max // 2, (7, 9)).astype(dt)      # sums / products wrap around: numpy's bits
            row = rng.randint(-100, 100, (9,)).astype(dt)
        a[2, 3] = a[5, 1] = a.max()                                         # ties for max
        ty[name + "/a"], ty[name + "/b"], ty[name + "/row"] = a, b, row
        A, Bt, R = T.from_numpy(a), T.from_numpy(b), T.from_numpy(row)
        with light.no_grad():
            ty[name + "/neg"] = (-A).numpy()
            ty[name + "/add"] = (A + Bt).numpy()
            ty[name + "/sub"] = (A - Bt).numpy()
            ty[name + "/mul"] = (A * Bt).numpy()
            ty[name + "/add_row"] = (A + R).numpy()
            ty[name + "/mul_transposed"] = (A.transpose(1, 0) * Bt.transpose(1, 0)).numpy()
            ty[name + "/add_scalar"] = (A + 3).numpy()
            ty[name + "/rmul_scalar"] = (5 * A).numpy()
            ty[name + "/add_float_scalar"] = (A + 0.5).numpy()               # integers: float64 result
            acc = T.from_numpy(a.copy())
            acc += Bt
            acc *= R
            acc -= 7
            ty[name + "/inplace"] = acc.numpy()
            for ax, tag in ((None, "all"), (0, "0"), (1, "1")):
                ty["%s/sum_%s" % (name, tag)] = np.asarray(A.sum(axis=ax).numpy())
                ty["%s/max_%s" % (name, tag)] = np.asarray(A.max(axis=ax).numpy())
                ty["%s/min_%s" % (name, tag)] = np.asarray(A.min(axis=ax).numpy())
            ty[name + "/sum_keepdims"] = A.sum(axis=1, keepdims=True).numpy()
            if dt is np.float64:
                ty[name + "/div"] = (A / Bt).numpy()
                ty[name + "/pow"] = (Bt ** A).numpy()
        if dt is np.float64:
            # gradients through the tape in float64: d/da, d/db of ((a * b + row) * a - b).max(axis=1) weighted by w
            w = rng.uniform(-1, 1, (7,))
            ty[name + "/w"] = w
            A, Bt, R = T.from_numpy(a), T.from_numpy(b), T.from_numpy(row)
            y = ((A * Bt + R) * A - Bt)
            (y * T.from_numpy(w.reshape(7, 1))).backward(allow_fill=True)
            ty[name + "/y"] = y.numpy()
            ty[name + "/grad_a"], ty[name + "/grad_b"], ty[name + "/grad_row"] = A.grad.numpy(), Bt.grad.numpy(), R.grad.numpy()
    for k, v in ty.items():
        assert isinstance(v, np.ndarray), k
    np.savez_compressed(os.path.join(OUT, "typed_ops.npz"), **ty)

    # ---------------------------------------------------------------- tiny-BERT forward (BASELINE config #5)
    # model classes loaded from the reference's examples/bert.py by file path; its Embedding.forward hard-codes
    # `.opencl()` (bert.py:19-21), replaced here by the same CPU lookup without the device hop.
    # Only the forward exists in the reference (its backward fails on three counts).
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_bert", os.path.join(REF, "examples", "bert.py"))
    ref_bert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_bert)
    ref_bert.Embedding.forward = lambda self, ids: self.weight.cpu()[ids.cpu()]
    tiny = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=30522,
                max_position_embeddings=512, type_vocab_size=2, attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0)
    np.random.seed(42)
    model = ref_bert.BertForMaskedLM(**tiny)
    ids = np.random.randint(0, tiny["vocab_size"], size=(2, 128)).astype(np.int32)
    # the reference reshapes the mask to (1, 1, *mask.shape) (bert.py:82), which only broadcasts for batch 1
    mask = np.ones((1, 128), dtype=np.float32)
    mask[0, 100:] = 0
    with light.no_grad():
        logits = model(T.from_numpy(ids)).numpy()
        logits_masked = model(T.from_numpy(ids[:1]), attention_mask=T.from_numpy(mask)).numpy()
    wsum = {n: float(np.abs(p.numpy().astype(np.float64)).sum()) for n, p in model.named_parameters()}
    np.savez_compressed(os.path.join(OUT, "bert_tiny_forward.npz"), ids=ids, mask=mask,
                        logits_sample=logits[:, :, ::509].copy(), logits_masked_sample=logits_masked[:, ::8, ::509].copy(),
                        logits_digest=np.asarray([logits.astype(np.float64).sum(), np.abs(logits).astype(np.float64).sum()]),
                        argmax=logits.argmax(-1).astype(np.int64),
                        param_names=np.asarray(sorted(wsum)), param_abs_sums=np.asarray([wsum[k] for k in sorted(wsum)]))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
