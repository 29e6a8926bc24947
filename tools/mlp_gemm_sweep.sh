python tools/mlp_gemm_bench.py | cut -c1-60
for n in 512 1024 2048; do python tools/gemm_bench.py $n 3; done
LG_GEMM_TILE=9 python tools/gemm_probe.py 1024,128,128,0,1 1024,512,128,0,1 1024,128,512,0,1 1024,30522,128,0,1 128,30522,1024,1,0 1024,128,30522,0,0 
