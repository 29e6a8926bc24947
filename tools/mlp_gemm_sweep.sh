LG_GEMM_TILE=9 LG_GEMM_SLICES=1 python tools/gemm_probe.py 1024,512,64,0,1 1024,512,1024,0,1 2048,1024,1024,0,1 4096,1024,1024,0,1 1024,784,512,0,0 512,784,1024,1,0
python tools/mlp_gemm_bench.py
for n in 1024 2048 4096; do python tools/gemm_bench.py $n 3; done
