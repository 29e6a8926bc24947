"""Data-parallel logic on CPU: world_size 2 over gloo (torch.distributed), CpuTensor replicas.
Pins SURVEY.md §8e: all-reduce SUM of per-rank gradients == gradient of the concatenated batch
(mse.backward has no 1/N), replicas stay bit-identical, gradients live in one flat bucket."""
import os
import socket
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, overlap):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import lightgrad_amd as light
    from lightgrad_amd import CpuTensor
    from lightgrad_amd.dist import GlooCommunicator, DataParallel
    from test_cpu_backend import MLP
    import np_oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    comm = GlooCommunicator()
    np.random.seed(100 + rank)                     # deliberately different init per rank: broadcast must fix it
    model = MLP(20, 16, 10)
    dp = DataParallel(model.parameters(), comm, overlap=overlap)
    w_start = {n: p.numpy().copy() for n, p in model.named_parameters()}
    # eps large enough to matter: Adam-type updates are scale-invariant up to eps, so only with a visible eps does the
    # first update tell "SUM x 1/world" from "SUM"
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, eps=0.05, grad_scale=dp.grad_scale)
    _, x, onehot, _ = O.synthetic_mlp_problem(500 + rank, 20, 16, 10, 8)     # own batch per rank
    losses, g_sum, w_after_first = [], None, None
    for step in range(3):
        l = light.loss.mse(model(CpuTensor.from_numpy(x)), CpuTensor.from_numpy(onehot))
        opt.zero_grad()
        l.backward()
        if overlap:       # step 0 learns the number of gradient writes per backward pass, later steps start the exchange at the last one
            assert dp._exchange_started == (step > 0) and dp._writes_seen == 4
        dp.sync_gradients()
        if overlap:
            assert dp._writes_expected == 4 and dp._writes_seen == 0 and not dp._exchange_started
        if step == 0:
            g_sum = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
            assert all(p.grad.data.base is not None for p in model.parameters())      # still views into the bucket
        opt.step()
        if step == 0:
            w_after_first = {n: p.numpy().copy() for n, p in model.named_parameters()}
        losses.append(l.item())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, onehot=onehot, losses=np.asarray(losses),
             digest=np.asarray(dp.parameter_digest()),
             **{"w0/" + n: v for n, v in w_start.items()}, **{"g/" + n: v for n, v in g_sum.items()},
             **{"w1/" + n: v for n, v in w_after_first.items()},
             **{"wf/" + n: p.numpy() for n, p in model.named_parameters()})
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [False, True], ids=["exchange_after_backward", "exchange_overlapped"])
def test_two_rank_data_parallel_matches_concatenated_batch(tmp_path, overlap):
    import torch.multiprocessing as mp
    import np_oracle as O
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for n in O.PARAM_ORDER:
        np.testing.assert_array_equal(r0["w0/" + n], r1["w0/" + n])      # broadcast from rank 0
        np.testing.assert_array_equal(r0["g/" + n], r1["g/" + n])        # same reduced gradient everywhere
        np.testing.assert_array_equal(r0["wf/" + n], r1["wf/" + n])      # replicas stay bit-identical
    assert r0["digest"] == r1["digest"]
    # all-reduce SUM == gradient of the concatenated batch (single process, oracle)
    w0 = {n: r0["w0/" + n] for n in O.PARAM_ORDER}
    x = np.concatenate([r0["x"], r1["x"]])
    t = np.concatenate([r0["onehot"], r1["onehot"]])
    _, g_cat, _ = O.mlp_loss_and_grads(w0, x, t)
    for n in O.PARAM_ORDER:
        np.testing.assert_allclose(r0["g/" + n], g_cat[n], rtol=1e-5, atol=1e-6, err_msg=n)
    # the update uses the MEAN gradient: SUM x grad_scale, grad_scale = 1/world.  One AdaBelief step from
    # w0 with half the concatenated-batch gradient must land on the weights both ranks hold after their first step -
    # and one with the un-scaled SUM must land somewhere else (eps = 0.05 makes the update depend on the scale)
    opt = O.AdamState(1e-3, belief=True, eps=0.05)
    d_mean = {n: opt.delta(n, g_cat[n] * np.float32(0.5)) for n in O.PARAM_ORDER}
    opt_sum = O.AdamState(1e-3, belief=True, eps=0.05)
    d_sum = {n: opt_sum.delta(n, g_cat[n]) for n in O.PARAM_ORDER}
    for n in O.PARAM_ORDER:
        np.testing.assert_array_equal(r0["w1/" + n], r1["w1/" + n])
        got = r0["w1/" + n].astype(np.float64) - w0[n]
        np.testing.assert_allclose(got, d_mean[n], rtol=1e-3, atol=1e-9, err_msg=n)           # (w0 + d) - w0 loses bits of d
        assert np.abs(got - d_sum[n]).max() > 20 * np.abs(got - d_mean[n]).max(), n           # ... and it is not the SUM update
    assert not np.array_equal(r0["losses"], r1["losses"])               # different batches per rank


def test_single_process_communicator_is_identity():
    import lightgrad_amd as light
    from lightgrad_amd import CpuTensor
    from lightgrad_amd.dist import SingleProcess, DataParallel
    from test_cpu_backend import MLP
    np.random.seed(3)
    model = MLP(6, 5, 4)
    dp = DataParallel(model.parameters(), SingleProcess())
    assert dp.grad_scale == 1.0 and dp.bucket.numel() == 6 * 5 + 5 + 5 * 4 + 4
    l = light.loss.mse(model(CpuTensor.uniform(0, 1, (3, 6))), CpuTensor.zeros((3, 4)))
    l.backward()
    dp.sync_gradients()
    flat = np.concatenate([p.grad.numpy().reshape(-1) for p in model.parameters()])
    np.testing.assert_array_equal(dp.bucket.numpy(), flat)             # gradients were accumulated INTO the bucket
    assert np.abs(flat).sum() > 0


def test_overlap_rejects_a_gradient_written_after_the_exchange_started():
    """the exchange starts at the learnt number of gradient writes; one more write in the same step must raise"""
    import lightgrad_amd as light
    from lightgrad_amd import CpuTensor
    from lightgrad_amd.dist import SingleProcess, DataParallel
    from test_cpu_backend import MLP
    np.random.seed(3)
    model = MLP(6, 5, 4)
    dp = DataParallel(model.parameters(), SingleProcess(), overlap=True)
    dp.always_sync = True
    x, t = CpuTensor.uniform(0, 1, (3, 6)), CpuTensor.zeros((3, 4))

    def backward_once():
        l = light.loss.mse(model(x), t)
        for p in model.parameters():
            p.zero_grad()
        l.backward()
    backward_once()
    assert not dp._exchange_started
    dp.sync_gradients()
    assert dp._writes_expected == 4
    backward_once()
    assert dp._exchange_started
    with pytest.raises(RuntimeError, match="after the gradient exchange"):
        light.loss.mse(model(x), t).backward()                    # a second backward before sync_gradients()
    dp.reset_overlap()
    backward_once()
    dp.sync_gradients()
    assert dp._writes_expected == 4
