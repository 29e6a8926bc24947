#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: produces everything profiles/README.md cites under gpurun_out/final/.
# Every step has its own limit; a step killed at its limit stops the script (never start another GPU step after a hang).
export TMPDIR=/tmp
out=gpurun_out/final
rm -rf $out; mkdir -p $out
step() {   # step <name> <seconds> <command...>   (stdout -> $out/<name>.out unless the command redirects)
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" bash -c "$*" ; local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 3 ]; then echo "step $name was killed at its limit or reported a hung exchange (rc $rc): stopping"; exit $rc; fi
}
step bench 400 "python bench.py > $out/bench.json 2> $out/bench.err"
step stats 400 "rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err"
step fetch 200 "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 tools/profile_kernels.py > $out/fetch.log 2>&1"
step write 200 "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 tools/profile_kernels.py > $out/write.log 2>&1"
step trace1 200 "rocprofv3 --kernel-trace --output-format csv -d $out/trace1 -- python3 bench.py --steps 200 --warmup 10 --no-extras --graph-steps 1 > $out/bench_one_step_per_graph.json 2> $out/trace1.err"
python tools/step_trace.py $out/stats/*/*_kernel_trace.csv > $out/mlp_step_trace.txt; tail -1 $out/mlp_step_trace.txt
python tools/step_gap.py $out/trace1/*/*_kernel_trace.csv > $out/step_gap_one_step_per_graph.txt; cat $out/step_gap_one_step_per_graph.txt
python tools/step_gap.py $out/stats/*/*_kernel_trace.csv > $out/step_gap_eight_steps_per_graph.txt; cat $out/step_gap_eight_steps_per_graph.txt
step sweep 300 "for n in 1024 2048 3072 4096 8192; do python tools/gemm_bench.py \$n 3; done > $out/gemm_sweep.txt 2>&1"; cat $out/gemm_sweep.txt
step hbm 200 "python tools/hbm_bench.py > $out/hbm_bench.txt 2>&1"; tail -3 $out/hbm_bench.txt
step mlpgemm 100 "python tools/mlp_gemm_bench.py > $out/mlp_gemm_bench.txt 2>&1"; cat $out/mlp_gemm_bench.txt
step headbench 100 "python tools/head_bench.py > $out/head_bench.txt 2>&1"; cat $out/head_bench.txt
step timeline 100 "python tools/gemm_timeline.py > $out/gemm_timeline.txt 2>&1"; tail -3 $out/gemm_timeline.txt
# tiny-BERT: one forward+backward kernel by kernel, the variants that were measured against it, its GEMM shapes, its loss kernel
step berttrace 300 "rocprofv3 --kernel-trace --output-format csv -d $out/berttrace -- python3 tools/bert_bench.py --replays 6 > $out/berttrace.log 2>&1"
python tools/bert_bench.py --trace $out/berttrace/*/*_kernel_trace.csv > $out/bert_step_trace.txt; tail -30 $out/bert_step_trace.txt
step bertbench 200 "python tools/bert_bench.py > $out/bert_bench.txt 2>&1; echo '--- LIGHTGRAD_GRAD_GROUP=0 (weight gradients launched where the tape makes them, dW and dx paired)' >> $out/bert_bench.txt; LIGHTGRAD_GRAD_GROUP=0 python tools/bert_bench.py >> $out/bert_bench.txt 2>&1"; cat $out/bert_bench.txt
step attntl 100 "python tools/attn_timeline.py > $out/attn_timeline.txt 2>&1"; cat $out/attn_timeline.txt
step bertgemm 100 "python tools/bert_gemm_bench.py > $out/bert_gemm_bench.txt 2>&1"; cat $out/bert_gemm_bench.txt
step cebench 100 "python tools/ce_bench.py > $out/ce_bench.txt 2>&1; LG_CE_HELD=0 python tools/ce_bench.py >> $out/ce_bench.txt 2>&1"; cat $out/ce_bench.txt
step rehearse 400 "python bench.py --gpus 2 --rehearse-on-one-gpu --steps 80 --warmup 10 --no-cpu-baseline > $out/bench_rehearsal_two_ranks_one_gpu.json 2> $out/bench_rehearsal.err"; tail -c 300 $out/bench_rehearsal_two_ranks_one_gpu.json
step rehearsefb 400 "LG_BENCH_FAIL_RCCL=init:1 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 80 --warmup 10 --no-extras --comm-open-timeout 5 > $out/bench_rehearsal_rccl_fallback.json 2> $out/bench_rehearsal_rccl_fallback.err"; tail -c 300 $out/bench_rehearsal_rccl_fallback.json
step torchrun 400 "python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse-on-one-gpu --steps 80 --warmup 10 --no-extras > $out/bench_rehearsal_under_torchrun.json 2> $out/bench_rehearsal_under_torchrun.err"; tail -c 300 $out/bench_rehearsal_under_torchrun.json
step ab 100 "python tools/mlp_step_ab.py 3 > $out/mlp_step_ab.txt 2>&1"; cat $out/mlp_step_ab.txt
step traceopt 200 "w=\$(mktemp -d /tmp/tr_XXXX); rocprofv3 --kernel-trace --output-format csv -d \$w -- python3 bench.py --steps 400 --warmup 40 --graph-steps 8 --no-extras --no-update-in-backward > $out/bench_optimizer_launch.json 2> $out/traceopt.err; python tools/kernel_window.py \$(find \$w -name '*kernel_trace.csv' | head -1) 8 head_fwd > $out/step_trace_optimizer_launch.txt; rm -rf \$w"; cat $out/step_trace_optimizer_launch.txt
step chain 100 "python tools/chain_bench.py > $out/chain_bench.txt 2>&1"; cat $out/chain_bench.txt
step dist2 300 "python -m pytest tests/test_hip_dist.py -m gpu -q > $out/dist_two_ranks_one_gpu.txt 2>&1"; tail -2 $out/dist_two_ranks_one_gpu.txt
step p2pbench 200 "python tools/p2p_bench.py 2 > $out/p2p_bench_cu_masked.txt 2>&1; P2P_BENCH_MASK=0 python tools/p2p_bench.py 2 > $out/p2p_bench_no_mask.txt 2>&1"; cat $out/p2p_bench_cu_masked.txt
step ipcprobe 100 "tools/ipc_probe.bin > $out/ipc_probe.txt 2>&1"; tail -3 $out/ipc_probe.txt
step ringlab 100 "tools/gemm_ring_lab.bin > $out/gemm_ring_lab_run.txt 2>&1"; cat $out/gemm_ring_lab_run.txt
step soak 200 "python tools/soak.py 20 > $out/soak.txt 2>&1"; cat $out/soak.txt
# SQ counters of the six MLP GEMM shapes + the head kernels (one counter pair per pass)
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  step sq$i 200 "rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/sq$i -- python3 tools/mlp_gemm_bench.py > $out/sq$i.log 2>&1"
done
python3 - <<'PY' > gpurun_out/final/pmc_sq_mlp_gemm.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/final/sq*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sgemm" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:64], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(agg.items()):
    print("%s  grid=%s" % (k, grid))
    for c, v in sorted(cs.items()):
        print("    %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $out/pmc_sq_mlp_gemm.txt | head -60
# (the rocprofv3 fault at the first wrap of the 16384-packet AQL ring is recorded in profiles/r2/wrap_*: tools/graph_wrap_probe.py
#  stays as a manual one-off check and is NOT part of a refresh - reproducing a profiler crash on the GPU box proves nothing new)
