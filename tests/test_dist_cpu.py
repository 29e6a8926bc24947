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


def _worker(rank, world, port, out_dir):
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
    dp = DataParallel(model.parameters(), comm)
    w_start = {n: p.numpy().copy() for n, p in model.named_parameters()}
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, grad_scale=dp.grad_scale)
    _, x, onehot, _ = O.synthetic_mlp_problem(500 + rank, 20, 16, 10, 8)     # own batch per rank
    losses, g_sum = [], None
    for step in range(3):
        l = light.loss.mse(model(CpuTensor.from_numpy(x)), CpuTensor.from_numpy(onehot))
        opt.zero_grad()
        l.backward()
        dp.sync_gradients()
        if step == 0:
            g_sum = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
            assert all(p.grad.data.base is not None for p in model.parameters())      # still views into the bucket
        opt.step()
        losses.append(l.item())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, onehot=onehot, losses=np.asarray(losses),
             digest=np.asarray(dp.parameter_digest()),
             **{"w0/" + n: v for n, v in w_start.items()}, **{"g/" + n: v for n, v in g_sum.items()},
             **{"wf/" + n: p.numpy() for n, p in model.named_parameters()})
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_matches_concatenated_batch(tmp_path):
    import torch.multiprocessing as mp
    import np_oracle as O
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
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
    # the update uses the MEAN gradient (grad_scale = 1/world)
    opt = O.make_optimizer("adabelief")
    w1 = {n: w0[n] + opt.delta(n, g_cat[n] * np.float32(0.5)) for n in O.PARAM_ORDER}
    assert not np.array_equal(r0["losses"], r1["losses"])               # different batches per rank
    _ = w1


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
