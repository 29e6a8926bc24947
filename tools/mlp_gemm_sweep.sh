python tools/hbm_bench.py 2>&1 | grep -E "sum|max|kernel"
