"""tiny-BERT forward + backward (batch 8 x 128, masked-LM cross-entropy on every position) as bench.py runs it: eager ms, hipGraph
replay ms (HIP events around all replays), and - with `--trace <kernel_trace.csv>` of a rocprofv3 run of THIS script - the kernels
of one replay in order with their durations.

    python tools/bert_bench.py [--replays 50]
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/bert_bench.py --replays 6; python tools/bert_bench.py --trace out/*/*_kernel_trace.csv
"""
import collections
import csv
import importlib.util
import os
import re
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def show_trace(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cross_entropy" in r["Kernel_Name"]]
    assert len(marks) >= 3, "need at least three forward+backward passes in the trace"
    seg = rows[marks[-2]:marks[-1]]                     # one pass, starting at its loss kernel (backward first, then the next forward)
    busy, by_name = 0.0, collections.OrderedDict()
    for r in seg:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += us
        name = re.sub(r"lg::|void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:84]
        print("%8.2f us  %-84s %s workgroups x %s" % (us, name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Workgroup_Size_X"]))
        n, t = by_name.get(name, (0, 0.0))
        by_name[name] = (n + 1, t + us)
    print("\n%d kernels, %.1f us of kernel time in one forward+backward; by kernel:" % (len(seg), busy))
    for name, (n, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
        print("%8.1f us  %3d x  %s" % (t, n, name))


def main():
    if "--trace" in sys.argv:
        return show_trace(sys.argv[sys.argv.index("--trace") + 1])
    import lightgrad_amd as light
    from lightgrad_amd import HipTensor
    from lightgrad_amd.autograd.hip import HipGraph, HipDevice
    from lightgrad_amd.dist import DataParallel, SingleProcess
    replays = int(sys.argv[sys.argv.index("--replays") + 1]) if "--replays" in sys.argv else 50
    spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
    bert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bert)
    np.random.seed(0)
    model = bert.BertForMaskedLM(**bert.TINY).map_parameters(lambda t: t.hip())
    ids = HipTensor.from_numpy(np.random.randint(0, bert.TINY["vocab_size"], (8, 128)).astype(np.int32), requires_grad=False)
    labels = HipTensor.from_numpy(np.random.randint(0, bert.TINY["vocab_size"], (8 * 128,)).astype(np.int64), requires_grad=False)
    dp = DataParallel(model.parameters(), SingleProcess(), flatten=True)
    state = {}

    def step():
        logits = model(ids)
        state["loss"] = light.loss.cross_entropy(logits.reshape(-1, bert.TINY["vocab_size"]), labels)
        dp.bucket.fill(0)
        state["loss"].backward()
    for _ in range(3):
        step()
    HipDevice.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    HipDevice.synchronize()
    print("eager tape: %.3f ms per forward+backward, loss %.6f" % (1e3 * (time.perf_counter() - t0) / 5, state["loss"].item()))
    graph = HipGraph()
    with graph.capture():
        step()
    for _ in range(3):
        graph.replay()
    HipDevice.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    HipDevice.synchronize()
    print("hipGraph:   %.3f ms per forward+backward over %d replays, loss %.6f" % (1e3 * (time.perf_counter() - t0) / replays, replays, state["loss"].item()))


if __name__ == "__main__":
    main()
