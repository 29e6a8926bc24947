"""Data parallel on HipTensor replicas with world_size 2 - on ONE GPU (BASELINE config #4).

The two ranks are fresh interpreters started by the helper process of tests/conftest.py (`spawn_ranks`), both bound to HIP
device 0.  RCCL refuses a second rank on a device, but hipIpc memory handles are per PROCESS: the hand-written peer-window
exchange (csrc/p2p.hip, dist.PeerWindowCommunicator) runs between the two processes exactly as it runs between two GPUs of
a node - remote stores into the peer's window, system-scope flags, the reduce inside the optimizer launch - only the wire is
the GPU's own HBM instead of xGMI.  The host-staged communicator (D2H, gloo, H2D) stays as the plain cross-check.

Assertions = tests/test_dist_cpu.py: SUM of per-rank gradients == gradient of the concatenated batch (oracle), the update
uses SUM x 1/world (and clearly not SUM), replicas bit-identical; for the device exchange also bit-reproducible runs."""
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join("tests", "dist_rank_worker.py")
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _run(spawn_ranks, tmp_path, *flags, env=None, wait_for_all=False, timeout=300, world=2):
    res = spawn_ranks(world, [WORKER, "--out", str(tmp_path)] + [str(f) for f in flags], env=env, timeout=timeout, wait_for_all=wait_for_all)
    assert res["rc"] == 0, "ranks failed (rc %s):\n%s" % (res["rc"], "\n---- next rank ----\n".join(res["outputs"]))
    return res


def _check_reduced_gradient_against_float64(r0, r1, w0, g_cat32, floor=1e-5):
    """the bucket after the exchange against the gradient of the concatenated batch formed in float64 (the oracle's code on float64
    copies of the same float32 inputs): relative Frobenius <= 1e-5, or no further from it than twice the float32 oracle is"""
    import np_oracle as O
    from common import rel_frobenius
    x = np.concatenate([r0["x"], r1["x"]]).astype(np.float64)
    t = np.concatenate([r0["onehot"], r1["onehot"]]).astype(np.float64)
    _, g64, _ = O.mlp_loss_and_grads({n: a.astype(np.float64) for n, a in w0.items()}, x, t)
    for n in O.PARAM_ORDER:
        assert g64[n].dtype == np.float64
        e, e32 = rel_frobenius(r0["g/" + n], g64[n]), rel_frobenius(g_cat32[n], g64[n])
        assert e <= max(floor, 2 * e32), "reduced gradient %s: %.3g from float64 (float32 oracle: %.3g)" % (n, e, e32)
    return g64


def _check_first_update_against_float64(r0, w0, g_cat32, g64):
    """w1 - w0 after the first step == AdaBelief's update for SUM x 1/world.  The difference of two float32 weights carries their
    rounding (half an ulp of the WEIGHT each, not of the update), so the yardstick for "as good as float32 gets" is the float32
    oracle's update pushed through float32 weights the same way"""
    import np_oracle as O
    from common import rel_frobenius
    opt64, opt32 = O.AdamState(1e-3, belief=True, eps=0.05), O.AdamState(1e-3, belief=True, eps=0.05)
    d_mean = {}
    for n in O.PARAM_ORDER:
        d64 = opt64.delta(n, g64[n] * 0.5)
        d32 = opt32.delta(n, g_cat32[n] * np.float32(0.5))
        d_mean[n] = d32
        assert d64.dtype == np.float64
        got = r0["w1/" + n].astype(np.float64) - w0[n]
        e, e32 = rel_frobenius(got, d64), rel_frobenius((w0[n] + d32).astype(np.float64) - w0[n], d64)
        assert e <= max(1e-5, 2 * e32), "first update %s: %.3g from float64 (float32 oracle through float32 weights: %.3g)" % (n, e, e32)
    return d_mean


def _check_training(tmp_path, reduced_tolerance=1e-5):
    import np_oracle as O
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for n in O.PARAM_ORDER:
        np.testing.assert_array_equal(r0["w0/" + n], r1["w0/" + n])      # broadcast from rank 0
        np.testing.assert_array_equal(r0["g/" + n], r1["g/" + n])        # same reduced gradient everywhere
        np.testing.assert_array_equal(r0["wf/" + n], r1["wf/" + n])      # replicas stay bit-identical
    assert r0["digest"] == r1["digest"]
    w0 = {n: r0["w0/" + n] for n in O.PARAM_ORDER}
    x = np.concatenate([r0["x"], r1["x"]])
    t = np.concatenate([r0["onehot"], r1["onehot"]])
    _, g_cat, _ = O.mlp_loss_and_grads(w0, x, t)
    for n in O.PARAM_ORDER:
        np.testing.assert_allclose(r0["g/" + n], g_cat[n], rtol=reduced_tolerance, atol=1e-6, err_msg=n)
    g64 = _check_reduced_gradient_against_float64(r0, r1, w0, g_cat, floor=max(1e-5, reduced_tolerance))
    d_mean = _check_first_update_against_float64(r0, w0, g_cat, g64)
    opt_sum = O.AdamState(1e-3, belief=True, eps=0.05)
    d_sum = {n: opt_sum.delta(n, g_cat[n]) for n in O.PARAM_ORDER}
    for n in O.PARAM_ORDER:                                             # ... and clearly not SUM
        got = r0["w1/" + n].astype(np.float64) - w0[n]
        assert np.abs(got - d_sum[n]).max() > 20 * np.abs(got - d_mean[n]).max(), n
    assert not np.array_equal(r0["losses"], r1["losses"])               # different batches per rank
    return r0, r1


@pytest.mark.parametrize("overlap,fused", [(0, 0), (1, 0), (0, 1), (1, 1)],
                         ids=["tape_optimizer", "tape_optimizer_overlap_hooks", "flat_bucket_fused_optimizer", "flat_bucket_fused_overlap_hooks"])
def test_two_ranks_on_one_device_host_staged_collectives(spawn_ranks, tmp_path, overlap, fused):
    _run(spawn_ranks, tmp_path, "--comm", "host", "--overlap", overlap, "--fused", fused)
    r0, _ = _check_training(tmp_path)
    assert not r0["in_optimizer"]


@pytest.mark.parametrize("fused,graph", [(0, 0), (1, 0), (1, 1)],
                         ids=["allreduce_kernel_tape_optimizer", "exchange_inside_the_optimizer_launch", "exchange_inside_a_replayed_hipgraph"])
def test_two_ranks_on_one_device_peer_window_exchange(spawn_ranks, tmp_path, fused, graph):
    """the device-side exchange, nothing staged through the host"""
    _run(spawn_ranks, tmp_path, "--comm", "p2p", "--fused", fused, "--graph", graph, "--steps", 6)
    r0, r1 = _check_training(tmp_path)
    assert bool(r0["in_optimizer"]) == bool(fused)
    first = {k: r0[k].copy() for k in r0.files}
    # the same job again: bit-reproducible (the owner sums in rank order)
    again = tmp_path / "again"
    again.mkdir()
    _run(spawn_ranks, again, "--comm", "p2p", "--fused", fused, "--graph", graph, "--steps", 6)
    s0 = np.load(again / "rank0.npz")
    for k in first:
        np.testing.assert_array_equal(first[k], s0[k], err_msg=k)


def test_exchange_in_graph_equals_eager(spawn_ranks, tmp_path):
    """six steps replayed from a hipGraph give the bits of six eager steps"""
    eager, graph = tmp_path / "eager", tmp_path / "graph"
    eager.mkdir()
    graph.mkdir()
    _run(spawn_ranks, eager, "--comm", "p2p", "--fused", 1, "--graph", 0, "--steps", 6)
    _run(spawn_ranks, graph, "--comm", "p2p", "--fused", 1, "--graph", 1, "--steps", 6)
    for r in (0, 1):
        a, b = np.load(eager / ("rank%d.npz" % r)), np.load(graph / ("rank%d.npz" % r))
        np.testing.assert_array_equal(a["losses"], b["losses"])
        for k in a.files:
            if k.startswith("wf/"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_exchange_at_mnist_mlp_size_in_a_replayed_graph(spawn_ranks, tmp_path):
    """BASELINE config #4 at its real size (784 -> 512 -> 10, batch 1024 per rank: 399 exchange workgroups per launch), 200 steps
    replayed from a hipGraph with both ranks on one GPU.  This is the case that needs each rank on CUs of its own
    (dist.shared_gpu_environment): without the masks the waiting workgroups of one rank keep the other's kernels off the device."""
    _run(spawn_ranks, tmp_path, "--comm", "p2p", "--fused", 1, "--graph", 1, "--steps", 200, "--dims", "784,512,10", "--batch", 1024,
         env={"LIGHTGRAD_TEST_WINDOW_FLOATS": 1 << 22})
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        if k.startswith(("w0/", "g/", "w1/", "wf/")):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    assert np.all(np.isfinite(r0["losses"])) and r0["losses"][-1] < r0["losses"][0]
    import np_oracle as O
    w0 = {n: r0["w0/" + n] for n in O.PARAM_ORDER}
    _, g_cat, _ = O.mlp_loss_and_grads(w0, np.concatenate([r0["x"], r1["x"]]), np.concatenate([r0["onehot"], r1["onehot"]]))
    # SUM over the ranks == gradient of the concatenated 2048-sample batch: against the oracle evaluated in FLOAT64 on the same
    # float32 inputs, whole arrays, at the north star's 1e-5 (relative Frobenius); two float32 results can only be compared loosely
    _check_reduced_gradient_against_float64(r0, r1, w0, g_cat)


def test_exchange_of_a_bucket_beyond_448_chunks(spawn_ranks, tmp_path):
    """a 1.06 M-parameter model (1040 chunks of 1024 floats: three pieces per exchange workgroup, parameters that straddle pieces),
    exchange inside the optimizer launch, replayed from a hipGraph"""
    import np_oracle as O
    _run(spawn_ranks, tmp_path, "--comm", "p2p", "--fused", 1, "--graph", 1, "--steps", 4, "--dims", "1000,1050,10", "--batch", 16,
         env={"LIGHTGRAD_TEST_WINDOW_FLOATS": 1 << 21})
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        if k.startswith(("w0/", "g/", "w1/", "wf/")):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    w0 = {n: r0["w0/" + n] for n in O.PARAM_ORDER}
    _, g_cat, _ = O.mlp_loss_and_grads(w0, np.concatenate([r0["x"], r1["x"]]), np.concatenate([r0["onehot"], r1["onehot"]]))
    g64 = _check_reduced_gradient_against_float64(r0, r1, w0, g_cat)
    _check_first_update_against_float64(r0, w0, g_cat, g64)


@pytest.mark.parametrize("world,window", [(2, 1 << 16), (4, 1 << 16), (2, 1 << 21)], ids=["two_ranks", "four_ranks", "two_ranks_several_pieces_per_workgroup"])
def test_peer_window_collectives(spawn_ranks, tmp_path, world, window):
    res = _run(spawn_ranks, tmp_path, "--mode", "collectives", world=world, env={"LIGHTGRAD_TEST_WINDOW_FLOATS": window})
    assert len(res["outputs"]) == world and all("collectives ok" in o for o in res["outputs"]), res["outputs"]


def test_four_ranks_on_one_device(spawn_ranks, tmp_path):
    """world_size 4 (each rank on a quarter of the CUs): every chunk now has THREE pushers and the owner sums four contributions in
    rank order; exchange inside the optimizer launch, steps 2.. replayed from a hipGraph.  SUM == gradient of the concatenated
    four batches, the update uses SUM / 4, four bit-identical replicas."""
    import np_oracle as O
    _run(spawn_ranks, tmp_path, "--comm", "p2p", "--fused", 1, "--graph", 1, "--steps", 6, world=4)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(4)]
    for other in r[1:]:
        for k in r[0].files:
            if k.startswith(("w0/", "g/", "w1/", "wf/")):
                np.testing.assert_array_equal(r[0][k], other[k], err_msg=k)
    w0 = {n: r[0]["w0/" + n] for n in O.PARAM_ORDER}
    _, g_cat, _ = O.mlp_loss_and_grads(w0, np.concatenate([q["x"] for q in r]), np.concatenate([q["onehot"] for q in r]))
    opt = O.AdamState(1e-3, belief=True, eps=0.05)
    for n in O.PARAM_ORDER:
        np.testing.assert_allclose(r[0]["g/" + n], g_cat[n], rtol=1e-5, atol=1e-6, err_msg=n)
        got = r[0]["w1/" + n].astype(np.float64) - w0[n]
        np.testing.assert_allclose(got, opt.delta(n, g_cat[n] * np.float32(0.25)), rtol=2e-3, atol=1e-8, err_msg=n)
    assert len({float(q["losses"][0]) for q in r}) == 4               # four different batches


def test_a_lost_peer_is_an_error_not_a_hang(spawn_ranks, tmp_path):
    res = _run(spawn_ranks, tmp_path, "--mode", "lost_peer", env={"LG_P2P_TIMEOUT_MS": "300"}, wait_for_all=True, timeout=120)
    assert "lost peer reported" in res["outputs"][0], res["outputs"][0]


def test_launches_in_flight_on_a_dead_communicator_are_reported_too(spawn_ranks, tmp_path):
    res = _run(spawn_ranks, tmp_path, "--mode", "lost_peer_in_flight", env={"LG_P2P_TIMEOUT_MS": "300"}, wait_for_all=True, timeout=120)
    assert "in-flight launch on a dead communicator reported" in res["outputs"][0], res["outputs"][0]


def test_exchange_counts_wrap_around(spawn_ranks, tmp_path):
    res = _run(spawn_ranks, tmp_path, "--mode", "epoch_wrap", timeout=180)
    assert all("epochs wrapped" in o for o in res["outputs"]), res["outputs"]


# ---- bench.py itself with two ranks on the one GPU (the multi-rank bench under the driver's pytest) ---------------
def _bench_line(output):
    import json
    lines = [l for l in output.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "expected one JSON line from rank 0, got %d:\n%s" % (len(lines), output[-4000:])
    return json.loads(lines[0])


BENCH = ["bench.py", "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "40", "--warmup", "5", "--no-extras"]


def test_bench_two_ranks_end_to_end(spawn_ranks):
    """`bench.py --gpus 2 --rehearse-on-one-gpu`: communicators, forms of the exchange, calibration, the JSON line"""
    res = spawn_ranks(2, BENCH, timeout=420)
    assert res["rc"] == 0, "\n---- next rank ----\n".join(res["outputs"])
    out = _bench_line(res["outputs"][0])
    assert out["n_gpus"] == 2 and out["steps"] == 40 and out["metric"].startswith("REHEARSAL_")
    ranks = out["ranks"]
    assert ranks["communicator_ranks"] == 2 and ranks["world_size"] == 2
    assert ranks["exchange_form"] in ("p2p", "graph-inline", "eager")
    assert ranks["communicator"] == "PeerWindowCommunicator" and ranks["communicator_fallback"] is None
    assert ranks["peer_window_memory"] in ("uncached", "fine-grained", "hipMalloc")
    assert np.isfinite(out["final_loss"]) and out["value"] > 0
    assert len(ranks["per_rank_steps_per_sec"]) == 2
    pre = ranks["preflight"]
    assert pre["devices_visible"] >= 1 and pre["can_access_peer"][0][0] == 1
    cost = ranks["exchange_us_per_step"]
    assert cost["form"] == ranks["exchange_form"] and cost["with_exchange_us"] > 0 and cost["without_exchange_us"] > 0
    cal = ranks["exchange_calibration_steps_per_sec"]
    assert isinstance(cal["eager"], float) and isinstance(cal["p2p"], float), cal


def test_bench_falls_back_when_rccl_does_not_come_up(spawn_ranks):
    """RCCL initialisation fails on rank 1 and never returns on rank 0 (what ncclCommInitRank does when a peer falls out):
    every rank drops RCCL after --comm-open-timeout and the job runs on the peer windows; the line says so"""
    res = spawn_ranks(2, BENCH + ["--comm-open-timeout", "4"], env={"LG_BENCH_FAIL_RCCL": "init:1"}, timeout=420)
    assert res["rc"] == 0, "\n---- next rank ----\n".join(res["outputs"])
    out = _bench_line(res["outputs"][0])
    ranks = out["ranks"]
    assert ranks["communicator"] == "PeerWindowCommunicator" and ranks["communicator_ranks"] == 2
    assert "RCCL not usable" in ranks["communicator_fallback"] and "simulated RCCL initialisation failure" in ranks["communicator_fallback"]
    assert "rank 0: no answer within 4 s" in ranks["communicators_not_usable"]["rccl"]
    assert ranks["exchange_form"] in ("p2p", "graph-inline", "eager") and np.isfinite(out["final_loss"]) and out["value"] > 0


def test_bench_watchdog_ends_a_stuck_form_with_the_line_in_hand(spawn_ranks):
    res = spawn_ranks(2, BENCH + ["--exchange-timeout", "15"], env={"LG_BENCH_SIMULATE_HANG": "p2p"}, timeout=420, wait_for_all=True)
    assert res["codes"] == [3, 3], (res["codes"], "\n---- next rank ----\n".join(res["outputs"]))
    out = _bench_line(res["outputs"][0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["ranks"]["exchange_form"] == "eager"
    assert "no answer within 15 s" in out["ranks"]["exchange_calibration_steps_per_sec"]["p2p"]
    assert out["ranks"]["in_graph_exchange_fallback"] == "p2p: watchdog"


def test_two_ranks_bert_layer_with_fused_blocks(spawn_ranks, tmp_path):
    """BASELINE configs 4 + 5 together: an encoder layer of BERT (self-attention node, feed-forward node, embedding sum, LayerNorms,
    masked-LM cross-entropy) data parallel over two rank processes, the gradient exchange inside the multi-tensor optimizer
    launch.  Replicas start from rank 0's parameters and stay bit-identical; the reduced gradient is the sum of what the numpy
    CPU backend gets for the two batches from the same parameters."""
    import importlib.util
    import lightgrad_amd as light
    from lightgrad_amd import CpuTensor
    _run(spawn_ranks, tmp_path, "--mode", "bert", "--steps", 2, env={"LIGHTGRAD_TEST_WINDOW_FLOATS": str(1 << 17)})
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    names = sorted(k[3:] for k in r0.files if k.startswith("w0/"))
    assert len(names) > 20
    for n in names:
        np.testing.assert_array_equal(r0["w0/" + n], r1["w0/" + n], err_msg=n)
        np.testing.assert_array_equal(r0["g/" + n], r1["g/" + n], err_msg=n)
        np.testing.assert_array_equal(r0["wf/" + n], r1["wf/" + n], err_msg=n)
        assert not np.array_equal(r0["wf/" + n], r0["w0/" + n]) or ".key.bias" in n, n          # two updates happened
    assert r0["digest"] == r1["digest"]
    assert {"self_attention", "feed_forward", "embedding_sum", "layer_norm"} <= set(r0["nodes"].tolist())     # the fused blocks ran
    assert not np.array_equal(r0["losses"], r1["losses"])
    spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
    bert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bert)
    from common import DIST_BERT_CFG as BERT_CFG
    total = {n: 0.0 for n in names}
    for r in (r0, r1):
        model = bert.BertForMaskedLM(**BERT_CFG)
        model.load_parameters({n: r0["w0/" + n] for n in names})
        logits = model(CpuTensor.from_numpy(r["ids"], requires_grad=False))
        loss = light.loss.cross_entropy(logits.reshape(-1, BERT_CFG["vocab_size"]), CpuTensor.from_numpy(r["labels"], requires_grad=False))
        loss.backward()
        np.testing.assert_allclose(loss.item(), r["losses"][0], rtol=1e-5)
        for n, p in model.named_parameters():
            total[n] = total[n] + p.grad.numpy().astype(np.float64)
    for n in names:
        ref, got = total[n], r0["g/" + n].astype(np.float64)
        if ".key.bias" in n:                                  # mathematically zero
            assert np.abs(got).max() < 1e-6
            continue
        assert np.linalg.norm(got - ref) <= 2e-5 * np.linalg.norm(ref) + 1e-9, (n, np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
