"""What the peer-window gradient exchange costs per optimizer launch, with TWO rank processes on ONE GPU (each on its own
half of the CUs: dist.shared_gpu_environment).  Times back-to-back launches of

    plain   lg_adam_multi_dev_f32       (no exchange: what a single-GPU step pays)
    fused   lg_p2p_adam_multi_dev_f32   (exchange inside the optimizer launch, csrc/p2p.hip)
    split   lg_p2p_allreduce_f32 + lg_adam_multi_dev_f32   (the exchange as a launch of its own)

(P2P_BENCH_MASK=0: without the CU masks - a CU-masked stream alone costs ~15 us per launch on this stack) over the MNIST-MLP bucket (407 050 floats in 4 segments) with HIP events on the compute stream, eager and replayed from a
hipGraph of 8 launches.  The wire here is the GPU's own HBM, not xGMI: the numbers show the protocol's fixed costs (flag round
trips, drains, the extra pass over the bucket), not link bandwidth.

    python tools/p2p_bench.py            # starts its two ranks itself (lightgrad_amd.launch)
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from lightgrad_amd.dist import shared_gpu_environment, PeerWindowCommunicator
    masked = os.environ.get("P2P_BENCH_MASK", "1") == "1"
    env = shared_gpu_environment(rank, world)
    if not masked:
        env.pop("LG_CU_MASK")           # only exchange and update launches here: their workgroups all fit on the chip together
    os.environ.update(env)
    import numpy as np
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipDevice, HipGraph, lib as L
    lib = L.lib()
    comm = PeerWindowCommunicator(rank, world)
    offsets = (0, 401408, 401920, 407040, 407050)
    n = offsets[-1]
    rng = np.random.RandomState(rank)
    p, m, v = (HipTensor.from_numpy(rng.uniform(-1, 1, n).astype(np.float32), requires_grad=False) for _ in range(3))
    v = HipTensor.from_numpy(np.abs(v.numpy()), requires_grad=False)
    g = HipTensor.from_numpy(rng.uniform(-1e-3, 1e-3, n).astype(np.float32), requires_grad=False)
    chunks = sum(-(-(b - a) // 1024) for a, b in zip(offsets[:-1], offsets[1:]))
    step_plain = HipTensor._new_step_counter(0, slots=4 * 392)
    step_fused = HipTensor._new_step_counter(0, slots=chunks)
    off = L.i64(offsets)

    def plain():
        L.check(lib.lg_adam_multi_dev_f32(p.ptr, g.ptr, m.ptr, v.ptr, 4, off, 1e-3, 0.9, 0.999, 1e-8, step_plain.ptr, 4 * 392, 0.5, 1))

    def fused():
        L.check(lib.lg_p2p_adam_multi_dev_f32(p.ptr, g.ptr, m.ptr, v.ptr, 4, off, 1e-3, 0.9, 0.999, 1e-8, step_fused.ptr, chunks, 0.5, 1))

    def split():
        L.check(lib.lg_p2p_allreduce_f32(g.ptr, n, 0))
        plain()

    def event():
        e = ctypes.c_void_p()
        L.check(lib.lg_event_create(ctypes.byref(e)))
        return e

    def timed(fn, reps):
        comm.barrier()
        e0, e1 = event(), event()
        L.check(lib.lg_event_record(e0))
        for _ in range(reps):
            fn()
        L.check(lib.lg_event_record(e1))
        HipDevice.synchronize()
        ms = ctypes.c_float()
        L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return 1e3 * ms.value / reps

    out = {}
    for name, fn in (("plain", plain), ("fused", fused), ("split", split)):
        for _ in range(20):
            fn()
        out[name + "_eager_us"] = min(timed(fn, 400) for _ in range(3))
        graph = HipGraph()
        with graph.capture():
            for _ in range(8):
                fn()
        for _ in range(5):
            graph.replay()
        out[name + "_graph_us"] = min(timed(graph.replay, 100) for _ in range(3)) / 8
    if rank == 0:
        print("MNIST-MLP bucket (407 050 floats, 4 segments), world %d on ONE GPU, %s; us per optimizer launch:"
              % (world, "each rank on %d of 256 CUs (CU-masked stream)" % (256 // world) if masked else "no CU masks"))
        for k in sorted(out):
            print("    %-18s %8.2f" % (k, out[k]))
        print("    exchange inside the optimizer launch costs %.2f us over the plain update (graph replay), as a launch of its own %.2f us"
              % (out["fused_graph_us"] - out["plain_graph_us"], out["split_graph_us"] - out["plain_graph_us"]))
    comm.close()         # (the bucket is summed in place launch after launch: its VALUES mean nothing here)


if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:
        rank_main()
    else:
        import importlib.util
        spec = importlib.util.spec_from_file_location("lightgrad_launch", os.path.join(ROOT, "lightgrad_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        sys.exit(launch.spawn_ranks(nproc, [os.path.abspath(__file__)], timeout=200))
