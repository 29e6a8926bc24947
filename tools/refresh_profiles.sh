#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: produces everything profiles/README.md cites under gpurun_out/final/.
export TMPDIR=/tmp
out=gpurun_out/final
rm -rf $out; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 tools/profile_kernels.py > $out/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 tools/profile_kernels.py > $out/write.log 2>&1; echo "write rc=$?"
python tools/step_trace.py $out/stats/*/*_kernel_trace.csv > $out/mlp_step_trace.txt; tail -1 $out/mlp_step_trace.txt
for n in 1024 2048 3072 4096 8192; do timeout -k 5 100 python tools/gemm_bench.py $n 3; done > $out/gemm_sweep.txt 2>&1; cat $out/gemm_sweep.txt
timeout -k 10 200 python tools/hbm_bench.py > $out/hbm_bench.txt 2>&1; tail -3 $out/hbm_bench.txt
timeout -k 10 100 python tools/mlp_gemm_bench.py > $out/mlp_gemm_bench.txt 2>&1; cat $out/mlp_gemm_bench.txt
