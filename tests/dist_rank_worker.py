"""One rank of a multi-process GPU test (started by tests/rank_spawner.py through the `spawn_ranks` fixture; RANK / WORLD_SIZE /
MASTER_* / LIGHTGRAD_RCCL_ID_FILE come from the environment).  Every rank binds HIP device 0: the ranks SHARE the one GPU of
the test box.  Modes:

  train        the data-parallel MLP training loop of tests/test_dist_cpu.py on HipTensor replicas; writes rank<r>.npz
               --comm host   collectives staged through the host (dist.HostStagedCommunicator: D2H, gloo, H2D)
               --comm p2p    collectives through peer-mapped device memory (dist.PeerWindowCommunicator, csrc/p2p.hip);
                             with --fused 1 the exchange rides inside the optimizer launch, with --graph 1 the steps after
                             the first are replayed from a hipGraph
  bert         one encoder layer of BERT with its fused blocks (attention node, feed-forward node, embedding sum) under DataParallel with
               the peer-window exchange inside the optimizer launch: two steps on per-rank batches; writes rank<r>.npz
  collectives  all-reduce SUM / MAX, broadcast, barrier of the peer-window communicator on awkward sizes, checked in place
  lost_peer    rank 1 never joins a collective: rank 0 must get HipError (LG_ECOMM) at its next synchronisation, not hang;
               the communicator stays failed (later collectives refused), close() does not wait for the lost peer
  lost_peer_in_flight   a launch enqueued behind the one that gave up, before any report, is reported too
  epoch_wrap   exchange counts seeded just below 2^31 and 2^32: the collectives keep working across the wrap
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
os.environ["LIGHTGRAD_HIP_DEVICE"] = "0"          # every rank on the one GPU ...
if os.environ.get("LIGHTGRAD_WORKER_MASK", "1") == "1":  # ... each on its own share of the CUs (dist.shared_gpu_environment)
    os.environ["LG_CU_MASK"] = "%s/%s" % (os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"))

import faulthandler  # noqa: E402
import numpy as np  # noqa: E402

faulthandler.dump_traceback_later(int(os.environ.get("LIGHTGRAD_WORKER_DUMP_AFTER", "100")), exit=False)   # a stuck rank says where


_T0 = time.time()


def log(msg):
    if os.environ.get("LIGHTGRAD_WORKER_VERBOSE"):
        sys.stderr.write("[rank %s %7.3f s] %s\n" % (os.environ.get("RANK"), time.time() - _T0, msg))
        sys.stderr.flush()


def make_comm(kind, rank, world):
    if kind == "host":
        import torch.distributed as dist
        from lightgrad_amd.dist import HostStagedCommunicator
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
        return HostStagedCommunicator()
    from lightgrad_amd.dist import PeerWindowCommunicator
    # a small window by default: large buffers then take several launches
    return PeerWindowCommunicator(rank, world, capacity_floats=int(os.environ.get("LIGHTGRAD_TEST_WINDOW_FLOATS", 1 << 16)))


def train(args, rank, world):
    import lightgrad_amd as light
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipGraph
    from lightgrad_amd.dist import DataParallel
    from test_cpu_backend import MLP
    import np_oracle as O
    comm = make_comm(args.comm, rank, world)
    log("communicator up")
    np.random.seed(100 + rank)                     # deliberately different init per rank: broadcast must fix it
    d_in, d_hidden, d_out = args.dims
    model = MLP(d_in, d_hidden, d_out).map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), comm, flatten=args.fused, overlap=args.overlap)
    w_start = {n: p.numpy().copy() for n, p in model.named_parameters()}
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, eps=0.05, grad_scale=dp.grad_scale, fused=args.fused, device_step=args.fused)
    if args.fused:
        dp.attach(opt)                             # flat buckets; with the peer-window communicator: exchange inside the optimizer launch
    in_optimizer = dp._exchange_in_optimizer
    log("model, bucket, optimizer ready")
    _, x, onehot, _ = O.synthetic_mlp_problem(500 + rank, d_in, d_hidden, d_out, args.batch)     # own batch per rank
    xt, tt = HipTensor.from_numpy(x), HipTensor.from_numpy(onehot)

    def step():
        l = light.loss.mse(model(xt), tt)
        opt.zero_grad()
        l.backward()
        dp.sync_gradients()
        log("backward launched")
        g = None if in_optimizer else {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
        opt.step()
        log("optimizer launched")
        if in_optimizer:                           # the bucket holds the summed gradient once the optimizer launch has run
            g = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
        return l, g
    losses = []
    l, g_sum = step()
    log("first step launched and read back")
    losses.append(l.item())
    w_after_first = {n: p.numpy().copy() for n, p in model.named_parameters()}
    if args.graph:
        assert args.fused
        graph = HipGraph()
        with graph.capture():
            gl = light.loss.mse(model(xt), tt)
            opt.zero_grad()
            gl.backward()
            dp.sync_gradients()
            opt.step()
        opt.t -= len(opt.parameters)
        for _ in range(args.steps - 1):
            graph.replay()
            opt.on_graph_replay()
            losses.append(gl.item())
    else:
        for _ in range(args.steps - 1):
            l, _ = step()
            losses.append(l.item())
    np.savez(os.path.join(args.out, "rank%d.npz" % rank), x=x, onehot=onehot, losses=np.asarray(losses),
             digest=np.asarray(dp.parameter_digest()), in_optimizer=np.asarray(in_optimizer),
             **{"w0/" + n: v for n, v in w_start.items()}, **{"g/" + n: v for n, v in g_sum.items()},
             **{"w1/" + n: v for n, v in w_after_first.items()},
             **{"wf/" + n: p.numpy() for n, p in model.named_parameters()})
    comm.close()
    if args.comm == "host":
        import torch.distributed as dist
        dist.destroy_process_group()


from common import DIST_BERT_CFG as BERT_CFG  # noqa: E402


def bert_batch(rank):
    rng = np.random.RandomState(900 + rank)
    return rng.randint(0, BERT_CFG["vocab_size"], (2, 32)).astype(np.int32), rng.randint(0, BERT_CFG["vocab_size"], (64,)).astype(np.int64)


def bert_train(args, rank, world):
    import importlib.util
    import lightgrad_amd as light
    from lightgrad_amd import HipTensor
    from lightgrad_amd.dist import DataParallel
    spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
    bert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bert)
    comm = make_comm("p2p", rank, world)
    np.random.seed(300 + rank)                     # different init per rank: the broadcast must fix it
    model = bert.BertForMaskedLM(**BERT_CFG).map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), comm, flatten=True)
    w_start = {n: p.numpy().copy() for n, p in model.named_parameters()}
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, grad_scale=dp.grad_scale, fused=True, device_step=True)
    dp.attach(opt)
    assert dp._exchange_in_optimizer
    ids, labels = bert_batch(rank)
    tids, tlabels = HipTensor.from_numpy(ids, requires_grad=False), HipTensor.from_numpy(labels, requires_grad=False)
    fused_nodes = set()
    losses, g_sum = [], None
    for it in range(args.steps):
        logits = model(tids)
        loss = light.loss.cross_entropy(logits.reshape(-1, BERT_CFG["vocab_size"]), tlabels)
        node, seen = logits.ctx, set()
        stack = [loss.ctx]
        while stack:                               # which tape nodes did this model run on?
            f = stack.pop()
            if f is None or id(f) in seen:
                continue
            seen.add(id(f))
            fused_nodes.add(f.__class__.__name__)
            stack.extend(getattr(t, "ctx", None) for t in f._parents if hasattr(t, "ctx"))
        opt.zero_grad()
        loss.backward()
        dp.sync_gradients()
        opt.step()
        if it == 0:
            g_sum = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}     # the bucket holds the summed gradient
        losses.append(loss.item())
    np.savez(os.path.join(args.out, "rank%d.npz" % rank), ids=ids, labels=labels, losses=np.asarray(losses),
             digest=np.asarray(dp.parameter_digest()), nodes=np.asarray(sorted(fused_nodes)),
             **{"w0/" + n: v for n, v in w_start.items()}, **{"g/" + n: v for n, v in g_sum.items()},
             **{"wf/" + n: p.numpy() for n, p in model.named_parameters()})
    comm.close()


def collectives(args, rank, world):
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipDevice
    comm = make_comm("p2p", rank, world)
    assert comm.ranks_seen() == world
    rngs = [np.random.RandomState(7000 + r) for r in range(world)]
    # sizes: one value, not a multiple of 4, exactly one piece, several pieces + tail, beyond the window (65536 floats)
    for n in (1, 3, 1024, 5000, 65536, 200001):
        parts = [g.uniform(-1, 1, n).astype(np.float32) for g in rngs]
        want_sum = parts[0].copy()
        for q in parts[1:]:
            want_sum = want_sum + q                            # rank order, fp32: the kernel's bits
        want_max = np.maximum.reduce(parts)
        t = HipTensor.from_numpy(parts[rank], requires_grad=False)
        comm.allreduce_sum_(t)
        np.testing.assert_array_equal(t.numpy(), want_sum, err_msg="sum n=%d" % n)
        t = HipTensor.from_numpy(parts[rank], requires_grad=False)
        comm.allreduce_max_(t)
        np.testing.assert_array_equal(t.numpy(), want_max, err_msg="max n=%d" % n)
        for root in range(world):
            t = HipTensor.from_numpy(parts[rank], requires_grad=False)
            comm.broadcast_(t, root)
            np.testing.assert_array_equal(t.numpy(), parts[root], err_msg="broadcast n=%d root=%d" % (n, root))
    # several 1024-float pieces per workgroup: a launch never has more than 448 workgroups (LIGHTGRAD_TEST_WINDOW_FLOATS >= 2 Mi)
    big = int(os.environ.get("LIGHTGRAD_TEST_WINDOW_FLOATS", 1 << 16))
    if big >= (1 << 21):
        for n in (1 << 20, (1 << 21) - 3):                     # 1024 / 2048 pieces -> 3 / 5 per workgroup, the second with a tail
            parts = [g.uniform(-1, 1, n).astype(np.float32) for g in rngs]
            want = parts[0].copy()
            for q in parts[1:]:
                want = want + q
            t = HipTensor.from_numpy(parts[rank], requires_grad=False)
            comm.allreduce_sum_(t)
            np.testing.assert_array_equal(t.numpy(), want, err_msg="sum n=%d (multi-piece)" % n)
    # an unaligned view of a bucket (element offset 1): the dword path
    base = np.zeros(4099, np.float32)
    base[1:] = rngs[0].uniform(-1, 1, 4098).astype(np.float32) * (rank + 1)
    t = HipTensor.from_numpy(base, requires_grad=False)
    view = HipTensor(t.data, (4098,), None, t.offset + 1, t.dtype, requires_grad=False)      # a view (getitem copies)
    assert view.is_contiguous() and view.ptr % 16 == 4
    comm.allreduce_sum_(view)
    want = np.zeros(4098, np.float32)
    for r in range(world):
        c = (base[1:] / np.float32(rank + 1)) * np.float32(r + 1)
        want = c if r == 0 else want + c
    np.testing.assert_allclose(t.numpy()[1:], want, rtol=1e-6)
    assert t.numpy()[0] == 0
    # many collectives back to back without a host synchronisation in between (epochs, no flag is ever reset)
    t = HipTensor.from_numpy(np.full(2048, float(rank + 1), np.float32), requires_grad=False)
    for _ in range(200):
        comm.allreduce_max_(t)
    comm.allreduce_sum_(t)
    HipDevice.synchronize()
    np.testing.assert_array_equal(t.numpy(), np.full(2048, float(world * world), np.float32))
    comm.close()
    print("rank %d: collectives ok" % rank)


def lost_peer(args, rank, world):
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipDevice, lib as L
    comm = make_comm("p2p", rank, world)           # includes one barrier: both ranks are connected
    if rank != 0:
        time.sleep(4.0)                            # never joins the collective below; exits without close()
        return
    t = HipTensor.from_numpy(np.ones(4096, np.float32), requires_grad=False)
    t0 = time.time()
    comm.allreduce_sum_(t)
    try:
        HipDevice.synchronize()
    except L.HipError as e:
        took = time.time() - t0
        assert "peer-window exchange" in str(e) and "liblghip error -5" in str(e), str(e)
        assert took < 3.0, took
        print("rank 0: lost peer reported after %.2f s: %s" % (took, e))
    else:
        raise AssertionError("a collective without its peer returned normally")
    # the communicator stays failed: a second collective is REFUSED (never launched with waits that fall through) ...
    assert comm.failed()
    before = t.numpy().copy()
    try:
        comm.allreduce_sum_(t)
    except L.HipError as e:
        assert "liblghip error -5" in str(e) and "lost a peer earlier" in str(e), str(e)
        print("rank 0: second collective refused: %s" % e)
    else:
        raise AssertionError("a collective on a failed communicator was launched")
    HipDevice.synchronize()                        # ... nothing pending, nothing reported twice for the same launch
    np.testing.assert_array_equal(t.numpy(), before)
    # ... and close() does not wait for the peer that is gone (no barrier, no "bye")
    t0 = time.time()
    comm.close()
    assert time.time() - t0 < 2.0, time.time() - t0
    print("rank 0: closed a failed communicator in %.2f s" % (time.time() - t0))


def lost_peer_in_flight(args, rank, world):
    """an exchange recorded in a hipGraph and replayed AFTER the communicator died runs its (short) waits on the dead
    communicator and is reported again: an exchange that did not happen is never taken for one that did"""
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipDevice, HipGraph, lib as L
    comm = make_comm("p2p", rank, world)
    b = HipTensor.from_numpy(np.ones(4096, np.float32), requires_grad=False)
    graph = HipGraph()
    with graph.capture():
        comm.allreduce_sum_(b)
    graph.replay()                                 # both ranks: works
    HipDevice.synchronize()
    np.testing.assert_array_equal(b.numpy(), np.full(4096, float(world), np.float32))
    if rank != 0:
        time.sleep(5.0)                            # gone from here on; exits without close()
        return
    errors = []
    for _ in range(2):                             # first replay: the wait gives up; second: dead communicator, short waits
        t0 = time.time()
        graph.replay()
        try:
            HipDevice.synchronize()
        except L.HipError as e:
            assert "liblghip error -5" in str(e), str(e)
            errors.append(time.time() - t0)
    assert len(errors) == 2, errors
    assert errors[1] < 0.25, errors                # no second LG_P2P_TIMEOUT_MS wait
    assert comm.failed()
    print("rank 0: in-flight launch on a dead communicator reported (%s s)" % errors)
    comm.close()


def epoch_wrap(args, rank, world):
    """exchange counts live modulo 2^32: collectives keep working across 2^31 and across 2^32"""
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipDevice, lib as L
    from lightgrad_amd.dist import _exchange_blobs
    comm = make_comm("p2p", rank, world)
    lib = L.lib()
    prefix = os.environ["LIGHTGRAD_RCCL_ID_FILE"] + ".wrap"
    for k, seed in enumerate((0x7FFFFFFF - 3, -3)):
        HipDevice.synchronize()
        _exchange_blobs(rank, world, b"idle", "%s.a%d" % (prefix, k), timeout=60)      # nobody has a launch in flight
        L.check(lib.lg_p2p_debug_seed_epochs(seed))
        _exchange_blobs(rank, world, b"seeded", "%s.b%d" % (prefix, k), timeout=60)    # nobody launches before everyone has seeded
        for it in range(8):                                                             # crosses the wrap at the 4th / 3rd exchange
            parts = [np.random.RandomState(31 * it + r).uniform(-1, 1, 5000).astype(np.float32) for r in range(world)]
            want = parts[0].copy()
            for q in parts[1:]:
                want = want + q
            t = HipTensor.from_numpy(parts[rank], requires_grad=False)
            comm.allreduce_sum_(t)
            np.testing.assert_array_equal(t.numpy(), want, err_msg="seed %d exchange %d" % (seed, it))
    comm.close()
    print("rank %d: epochs wrapped" % rank)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train")
    ap.add_argument("--comm", default="p2p")
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--fused", type=int, default=0)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=".")
    ap.add_argument("--dims", default="20,16,10", help="MLP sizes: inputs,hidden,outputs")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    a.overlap, a.fused, a.graph = bool(a.overlap), bool(a.fused), bool(a.graph)
    a.dims = tuple(int(v) for v in a.dims.split(","))
    try:
        {"train": train, "bert": bert_train, "collectives": collectives, "lost_peer": lost_peer, "lost_peer_in_flight": lost_peer_in_flight,
         "epoch_wrap": epoch_wrap}[a.mode](a, int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        time.sleep(float(os.environ.get("LIGHTGRAD_WORKER_LINGER", "0")))      # debugging: let the other rank report too before the job is stopped
        sys.exit(1)
