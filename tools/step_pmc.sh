#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the kernels of the replayed MLP training step itself (one counter pair per pass), summarised per kernel.
export TMPDIR=/tmp
out=gpurun_out/step_pmc
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --steps 64 --warmup 8 --graph-steps 8 --no-extras --no-cpu-baseline > $out/p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done
python3 - <<'PY' > gpurun_out/step_pmc/step_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/step_pmc/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("sgemm_triple", "sgemm_mfma<64, 32", "sgemm_mfma<32, 32", "head_fwd", "adam_multi")):
            agg[(k[:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(agg.items()):
    print("%s  grid=%s" % (k, grid))
    for c, v in sorted(cs.items()):
        print("    %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        mf, bz = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(cs["SQ_BUSY_CYCLES"]) / len(cs["SQ_BUSY_CYCLES"])
        print("    -> MFMA busy / (SQ busy x 4 SIMDs per CU... see profiles/README) raw ratio %.3f" % (mf / bz if bz else 0))
PY
cat gpurun_out/step_pmc/step_pmc.txt
