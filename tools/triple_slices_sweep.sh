# K-slice counts of the three products of the backward launch (dW2, dW1, dx), forced by position: LG_GEMM_PAIR_SLICES="a,b,c"; "" = the cost models
for v in "" "8,5,1" "8,5,2" "8,5,3" "4,5,2" "16,5,2" "8,4,2" "8,6,2" "8,5,4"; do
  echo "LG_GEMM_PAIR_SLICES=$v"; LG_GEMM_PAIR_SLICES=$v timeout -k 10 100 python bench.py --steps 2000 --warmup 200 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"])"
done
