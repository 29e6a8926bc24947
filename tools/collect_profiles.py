"""Run HERE after tools/refresh_profiles.sh ran on the GPU box: copies the judged artefacts from gpurun_out/final into
profiles/<round>/ and rewrites pmc_traffic.json.   python tools/collect_profiles.py r1 v3"""
import collections, csv, glob, json, os, shutil, sys
rnd, tag = sys.argv[1], sys.argv[2]
src, dst = "gpurun_out/final", os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_%s.json" % tag))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof_%s.json" % tag))
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)   # gpurun merges into gpurun_out/: older runs may linger
shutil.copy(newest(src + "/stats/*/*_kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats_%s.csv" % tag))
shutil.copy(os.path.join(src, "mlp_step_trace.txt"), os.path.join(dst, "mlp_step_trace_%s.txt" % tag))
shutil.copy(os.path.join(src, "gemm_sweep.txt"), os.path.join(dst, "gemm_sweep_%s.txt" % tag))
for extra in ("hbm_bench", "mlp_gemm_bench", "head_bench", "gemm_timeline", "pmc_sq_mlp_gemm", "step_gap_one_step_per_graph",
              "step_gap_eight_steps_per_graph", "wrap_summary", "bert_step_trace", "bert_bench", "bert_gemm_bench", "ce_bench", "graph_branch_probe", "soak", "dist_two_ranks_one_gpu",
              "p2p_bench_cu_masked", "p2p_bench_no_mask", "ipc_probe", "gemm_ring_lab_run", "attn_timeline", "mlp_step_ab",
              "step_trace_update_in_backward", "step_trace_optimizer_launch", "chain_bench"):
    if os.path.exists(os.path.join(src, extra + ".txt")):
        shutil.copy(os.path.join(src, extra + ".txt"), os.path.join(dst, "%s_%s.txt" % (extra, tag)))
for name in ("bench_rehearsal_two_ranks_one_gpu", "bench_rehearsal_rccl_fallback", "bench_rehearsal_under_torchrun", "bench_update_in_backward", "bench_optimizer_launch"):
    if os.path.exists(os.path.join(src, name + ".json")):
        shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, "%s_%s.json" % (name, tag)))
pmc = {}
for kind in ("fetch", "write"):
    f = newest("%s/%s/*/*_counter_collection.csv" % (src, kind))
    shutil.copy(f, os.path.join(dst, "pmc_%s_size_%s.csv" % (kind, tag)))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[kind] = sum(v) / len(v)
for k, v in sorted(pmc.items()):
    print("%-110s FETCH %10.1f KB  WRITE %10.1f KB" % (k[:110], v.get("fetch", 0), v.get("write", 0)))
nn = [v for k, v in pmc.items() if "sgemm_mfma<256, 256, 32, 4, 4, true, false" in k][0]
json.dump({"sgemm_mfma_256x256_NN_4096": {
    "fetch_size_kb_raw": nn["fetch"], "write_size_kb": nn["write"],
    "hbm_bytes_per_launch": int((2 * nn["fetch"] + nn["write"]) * 1024), "algorithmic_bytes_per_launch": 3 * 4096 * 4096 * 4,
    "correction": "FETCH_SIZE doubled (gfx950 reports half of a 16 B/lane read: MI355X_MICROARCH.md HBM section), WRITE_SIZE exact; KB -> bytes x1024",
    "source": "profiles/%s/pmc_fetch_size_%s.csv, pmc_write_size_%s.csv (rocprofv3 --pmc, one counter per pass, tools/profile_kernels.py)" % (rnd, tag, tag)}},
    open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
d = json.load(open(os.path.join(src, "bench.json")))
for f in glob.glob(src + "/wrap_rocprof_*.log"):           # the profiler-side fault: keep the head of every log (banner, progress, backtrace)
    with open(f, errors="replace") as fh:
        lines = fh.read().splitlines()
    keep = [l for l in lines if "replay" in l or "done" in l or "SIGSEGV" in l or "PC:" in l or l.strip().startswith("@")]
    with open(os.path.join(dst, os.path.basename(f).replace(".log", "_%s.txt" % tag)), "w") as fh:
        fh.write("\n".join(keep[-40:]) + "\n")
if os.path.exists("gpurun_out/bert_grad_errors.txt"):
    shutil.copy("gpurun_out/bert_grad_errors.txt", os.path.join(dst, "bert_grad_errors_%s.txt" % tag))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "secondary", "roofline", "tiny_bert_fwd_bwd", "cpu_baseline")}, indent=0)[:2500])
print({k: v["GB/s"] for k, v in d["roofline_hbm"].items() if isinstance(v, dict)})
