"""liblghip_comm.so on the GPU box: RCCL loads, a communicator initialises and collectives run on the
library's stream.  One GPU is all the test box has, so world_size is 1 here (RCCL refuses two ranks on one
device); the multi-rank logic is covered on CPU over gloo (tests/test_dist_cpu.py)."""
import numpy as np
import pytest
import lightgrad_amd as light

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_collectives_and_data_parallel_wrapper(hip, tmp_path):
    from lightgrad_amd.dist import RcclCommunicator, DataParallel
    from test_cpu_backend import MLP
    comm = RcclCommunicator(rank=0, world_size=1, id_path=str(tmp_path / "rccl.id"))
    try:
        t = hip.from_numpy(np.arange(1000, dtype=np.float32), requires_grad=False)
        comm.allreduce_sum_(t)
        comm.allreduce_max_(t)
        comm.broadcast_(t, 0)
        comm.barrier()
        np.testing.assert_array_equal(t.numpy(), np.arange(1000, dtype=np.float32))
        np.random.seed(2)
        model = MLP(12, 8, 4).map_parameters(lambda p: p.hip())
        dp = DataParallel(model.parameters(), comm)
        opt = light.optim.AdaBelief(model.parameters(), lr=1e-3, fused=True, grad_scale=dp.grad_scale)
        x, y = hip.uniform(0, 1, (5, 12)), hip.zeros((5, 4))
        before = [p.numpy().copy() for p in model.parameters()]
        l = light.loss.mse(model(x), y)
        opt.zero_grad()
        l.backward()
        dp.sync_gradients()
        flat = np.concatenate([p.grad.numpy().reshape(-1) for p in model.parameters()])
        np.testing.assert_array_equal(dp.bucket.numpy(), flat)          # gradients live in the bucket
        opt.step()
        assert all(not np.array_equal(b, p.numpy()) for b, p in zip(before, model.parameters()))
    finally:
        comm.close()
