"""Data parallel on HipTensor replicas with world_size 2 - on ONE GPU.  RCCL refuses a second rank on the same device, so the
collective itself travels through the host (dist.HostStagedCommunicator: D2H, gloo, H2D); everything above it is the code
that runs at N > 1: the flat HipTensor gradient bucket, the gradient-written hooks, zero-pending gradients, the fused
multi-tensor AdaBelief with grad_scale = 1 / world on device step counters, replica identity.  Same assertions as
tests/test_dist_cpu.py: SUM of per-rank gradients == gradient of the concatenated batch (oracle), the update
uses SUM x 1/world, replicas stay bit-identical."""
import os
import socket
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Runs only on request and in a pytest process of its own (LIGHTGRAD_MULTIPROC_GPU_TESTS=1 python -m pytest tests/test_hip_dist.py
# -m gpu): the ranks are child processes, and the process that starts them must not have touched the GPU itself - which the
# one process of a whole `pytest -m gpu` run has, long before it gets here.  Result of the last run: profiles/README.md.
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("LIGHTGRAD_MULTIPROC_GPU_TESTS") != "1",
                                 reason="multi-process GPU test: run it alone with LIGHTGRAD_MULTIPROC_GPU_TESTS=1")]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, overlap, fused):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import lightgrad_amd as light
    from lightgrad_amd import HipTensor
    from lightgrad_amd.dist import HostStagedCommunicator, DataParallel
    from test_cpu_backend import MLP
    import np_oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    comm = HostStagedCommunicator()
    np.random.seed(100 + rank)                     # deliberately different init per rank: broadcast must fix it
    model = MLP(20, 16, 10).map_parameters(lambda p: p.hip())
    dp = DataParallel(model.parameters(), comm, flatten=fused, overlap=overlap)
    w_start = {n: p.numpy().copy() for n, p in model.named_parameters()}
    opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, eps=0.05, grad_scale=dp.grad_scale, fused=fused, device_step=fused)
    if fused:
        dp.attach(opt)                             # flat buckets: zero_grad = one flag, update = one launch
    _, x, onehot, _ = O.synthetic_mlp_problem(500 + rank, 20, 16, 10, 8)     # own batch per rank
    xt, tt = HipTensor.from_numpy(x), HipTensor.from_numpy(onehot)
    losses, g_sum, w_after_first = [], None, None
    for step in range(3):
        l = light.loss.mse(model(xt), tt)
        opt.zero_grad()
        l.backward()
        dp.sync_gradients()
        if step == 0:
            g_sum = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap,fused", [(False, False), (True, False), (False, True), (True, True)],
                         ids=["tape_optimizer", "tape_optimizer_overlap_hooks", "flat_bucket_fused_optimizer", "flat_bucket_fused_overlap_hooks"])
def test_two_gpu_ranks_on_one_device_match_the_concatenated_batch(tmp_path, overlap, fused):
    import torch.multiprocessing as mp
    import np_oracle as O
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), overlap, fused), nprocs=2, join=True, start_method="spawn")
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
        np.testing.assert_allclose(r0["g/" + n], g_cat[n], rtol=1e-5, atol=1e-6, err_msg=n)
    opt = O.AdamState(1e-3, belief=True, eps=0.05)
    d_mean = {n: opt.delta(n, g_cat[n] * np.float32(0.5)) for n in O.PARAM_ORDER}
    opt_sum = O.AdamState(1e-3, belief=True, eps=0.05)
    d_sum = {n: opt_sum.delta(n, g_cat[n]) for n in O.PARAM_ORDER}
    for n in O.PARAM_ORDER:
        got = r0["w1/" + n].astype(np.float64) - w0[n]
        np.testing.assert_allclose(got, d_mean[n], rtol=2e-3, atol=1e-8, err_msg=n)
        assert np.abs(got - d_sum[n]).max() > 20 * np.abs(got - d_mean[n]).max(), n
    assert not np.array_equal(r0["losses"], r1["losses"])               # different batches per rank
