#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the 4096^3 GEMM kernels (one counter group per pass), summarised per kernel.
export TMPDIR=/tmp
out=gpurun_out/pmc_gemm
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 tools/gemm_bench.py 4096 1 > $out/p$i.log 2>&1; echo "pass $i ($grp) rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sgemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
