"""Differential fuzzing of the HIP tape against the CPU backend (tests/tape_fuzz.py): random training programs over the ingredients
the backend's peepholes are made of - lazy relu, the fused head, paired gradient launches, epilogue accumulation, lazily zeroed
gradients, the optimizer in its forms (tape, fused per parameter, flat buckets, update inside the backward kernels, inside a
hipGraph) - every observable value compared with what numpy computes for the same program."""
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from tape_fuzz import draw_program, run_program, compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(8))
def test_random_programs_match_the_cpu_backend(hip, block):
    for seed in range(block * 25, block * 25 + 25):
        prog = draw_program(seed)
        ref = run_program(CpuTensor, prog)
        got = run_program(hip, prog)
        compare(ref, got, what="seed %d %r" % (seed, prog))


def _flat(in_backward):
    def prepare(model, make_opt):
        from lightgrad_amd.dist import DataParallel, SingleProcess
        dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
        opt = make_opt(model.parameters())
        dp.attach(opt)
        if in_backward:
            opt.fuse_update_into_backward()
        return opt
    return prepare


@pytest.mark.parametrize("in_backward", [False, True], ids=["flat_buckets", "update_in_backward"])
def test_random_programs_with_flat_bucket_optimizers(hip, in_backward):
    """the same programs with the optimizer over flat buckets (lazy zero_grad, one update launch) and with the update applied by
    the backward kernels - the forms bench.py runs.  Programs that the second form refuses by contract (two backward passes per
    step, no zero_grad before backward) are skipped for it."""
    ran = 0
    for seed in range(300, 380):
        prog = draw_program(seed)
        if prog["optimizer"] == "sgd":
            continue
        prog = dict(prog, fused=True, device_step=True)
        if in_backward and (prog["second_backward"] or prog["zero_grad"] != "before_backward"):
            continue
        ref = run_program(CpuTensor, prog)
        got = run_program(hip, prog, prepare=_flat(in_backward))
        compare(ref, got, what="seed %d %r" % (seed, prog))
        ran += 1
    assert ran >= 15


def test_random_programs_replayed_from_a_graph(hip):
    """a program's step recorded in a hipGraph after one eager step and replayed: the values of the eager HIP run, bit for bit"""
    from lightgrad_amd.autograd.hip import HipGraph
    from tape_fuzz import Net
    ran = 0
    for seed in range(500, 620):
        prog = draw_program(seed)
        if prog["optimizer"] == "sgd" or any(prog["peek"]) or any(prog["poke_input"]) or prog["second_backward"]:
            continue                                   # host reads / writes inside the step cannot be recorded
        rng = np.random.RandomState(seed)
        x_np = rng.uniform(-1, 1, (prog["batch"], prog["dims"][0])).astype(np.float32)
        t_np = rng.uniform(-1, 1, (prog["batch"], prog["dims"][-1])).astype(np.float32)

        def build():
            np.random.seed(seed)
            model = Net(prog["dims"], prog["biases"], prog["norm"]).map_parameters(lambda p: p.hip())
            cls = light.optim.Adam if prog["optimizer"] == "adam" else light.optim.AdaBelief
            opt = cls(model.parameters(), lr=1e-2, eps=1e-3, fused=True, device_step=True)
            x, t = hip.from_numpy(x_np, requires_grad=prog["x_requires_grad"]), hip.from_numpy(t_np, requires_grad=False)

            def step():
                h = x
                for k, layer in enumerate(model.layers):
                    h = layer(h)
                    if prog["acts"][k] != "none":
                        h = getattr(h, prog["acts"][k])()
                    if prog["norm"][k]:
                        h = model.norms[k](h)
                loss = light.loss.mse(h, t)
                opt.zero_grad()
                loss.backward()
                opt.step()
                return loss
            return model, opt, step
        model_e, _, step_e = build()
        eager = [step_e().item() for _ in range(5)]
        model_g, opt_g, step_g = build()
        losses = [step_g().item()]
        graph = HipGraph()
        with graph.capture():
            loss = step_g()
        opt_g.t -= len(opt_g.parameters)
        for _ in range(4):
            graph.replay()
            opt_g.on_graph_replay()
            losses.append(loss.item())
        np.testing.assert_array_equal(losses, eager, err_msg="seed %d" % seed)
        for (n, p), (_, q) in zip(model_g.named_parameters(), model_e.named_parameters()):
            np.testing.assert_array_equal(p.numpy(), q.numpy(), err_msg="seed %d %s" % (seed, n))
        ran += 1
    assert ran >= 12
