// Micro-benchmark: which streaming shape gets closest to the HBM roof for c = a + b on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o gpurun_out/stream_bench && ./gpurun_out/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ inline float4 nt_load(const float4* p) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
__device__ inline void nt_store(float4 r, float4* p) { v4f t = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); }

template <int U, bool NT>
__global__ void __launch_bounds__(256) add_k(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long nvec) {
    const long stride = long(gridDim.x) * blockDim.x;
    long v = long(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; v + (U - 1) * stride < nvec; v += U * stride) {
        float4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) { x[u] = nt_load(a + v + u * stride); y[u] = nt_load(b + v + u * stride); }
            else { x[u] = a[v + u * stride]; y[u] = b[v + u * stride]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 r = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w);
            if (NT) nt_store(r, c + v + u * stride); else c[v + u * stride] = r;
        }
    }
    for (; v < nvec; v += stride) {
        float4 x = a[v], y = b[v];
        c[v] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

// contiguous chunk per block instead of grid-stride
template <int U>
__global__ void __launch_bounds__(256) add_chunk(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long nvec) {
    long base = (long(blockIdx.x) * 256 * U) + threadIdx.x;
    float4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < nvec) { x[u] = a[base + u * 256]; y[u] = b[base + u * 256]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < nvec) c[base + u * 256] = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w);
}

template <class F>
double time_ms(F launch, int reps = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const long n = 16384L * 8192, nvec = n / 4;
    float *a, *b, *c;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    const double gb = 3.0 * n * 4 / 1e9;
#define RUN(NAME, GRID, ...) { double ms = time_ms([&] { hipLaunchKernelGGL(__VA_ARGS__, dim3(GRID), dim3(256), 0, 0, (const float4*)a, (const float4*)b, (float4*)c, nvec); }); printf("%-34s grid %-8ld %.3f ms  %.0f GB/s\n", NAME, long(GRID), ms, gb / ms * 1e3); }
    for (long grid : {1024L, 2048L, 4096L, 8192L, 16384L}) {
        RUN("grid-stride U1", grid, (add_k<1, false>));
        RUN("grid-stride U2", grid, (add_k<2, false>));
        RUN("grid-stride U4", grid, (add_k<4, false>));
        RUN("grid-stride U2 nontemporal", grid, (add_k<2, true>));
        RUN("grid-stride U4 nontemporal", grid, (add_k<4, true>));
    }
    RUN("one float4 per thread", (nvec + 255) / 256, (add_chunk<1>));
    RUN("chunk U2", (nvec + 511) / 512, (add_chunk<2>));
    RUN("chunk U4", (nvec + 1023) / 1024, (add_chunk<4>));
    RUN("chunk U8", (nvec + 2047) / 2048, (add_chunk<8>));
    hipDeviceSynchronize();
    return 0;
}
