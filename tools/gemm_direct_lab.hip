// Lab (experiments only): a BARRIER-FREE K loop for the mid-size NT GEMM of the MLP step (1024 x 512 x 784).
// Every wave owns a 32x32 accumulator and fetches its own A / B fragments straight from global memory into registers
// (float4 along K per lane: lane (r, h) takes k = kb + 4h .. 4h+3 of row r, the same k order for A and B), double
// buffered in registers by chunks of 64 k - no LDS, no __syncthreads.  Measures the launch against the LDS-staged kernel
// of the library (15.5 us; K loop 0.68 us per 32 k against 0.43 us of MFMAs).
//   hipcc -O3 --offload-arch=gfx950 -o tools/gemm_direct_lab.bin tools/gemm_direct_lab.hip && tools/gemm_direct_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SLICES>
__global__ void __launch_bounds__(256) direct_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                 int M, int N, int K, int lda, int ldb, int ldc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int tiles_n = N / 64;
    const int t = blockIdx.x / SLICES, slice = blockIdx.x % SLICES;
    const int tm = t / tiles_n, tn = t % tiles_n;
    const int kper = ((K / SLICES + 7) / 8) * 8;
    const int k0 = slice * kper, k1 = (k0 + kper < K) ? k0 + kper : K;
    const float* a = A + size_t(tm * 64 + wm * 32 + r) * lda + 4 * h;
    const float* b = B + size_t(tn * 64 + wn * 32 + r) * ldb + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int CH = 8;                       // 8-k blocks per chunk
    float4 ca[CH], cb[CH], na[CH], nb[CH];
    auto load = [&](float4* xa, float4* xb, int kb0) {
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int kb = kb0 + q * 8;
            if (kb + 8 <= k1) { xa[q] = *reinterpret_cast<const float4*>(a + kb); xb[q] = *reinterpret_cast<const float4*>(b + kb); }
            else { xa[q] = make_float4(0, 0, 0, 0); xb[q] = make_float4(0, 0, 0, 0); }
        }
    };
    auto mul = [&](const float4* xa, const float4* xb) {
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].x, xb[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].y, xb[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].z, xb[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].w, xb[q].w, acc, 0, 0, 0);
        }
    };
    load(ca, cb, k0);
    for (int kb = k0; kb < k1; kb += 2 * CH * 8) {
        load(na, nb, kb + CH * 8);
        mul(ca, cb);
        load(ca, cb, kb + 2 * CH * 8);
        mul(na, nb);
    }
    float* c = C + size_t(slice) * M * ldc;
    const int col = tn * 64 + wn * 32 + r;
    const int row0 = tm * 64 + wm * 32 + 4 * h;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[size_t(row0 + (e & 3) + 8 * (e >> 2)) * ldc + col] = acc[e];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int M = 1024, N = 512, K = 784;
    std::vector<float> ha(size_t(M) * K), hb(size_t(N) * K);
    srand(1);
    for (auto& v : ha) v = rand() / float(RAND_MAX) * 2 - 1;
    for (auto& v : hb) v = rand() / float(RAND_MAX) * 2 - 1;
    float *A, *B, *C;
    CK(hipMalloc(&A, ha.size() * 4)); CK(hipMalloc(&B, hb.size() * 4)); CK(hipMalloc(&C, size_t(4) * M * N * 4));
    CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int slices) {
        const int grid = (M / 64) * (N / 64) * slices;
        if (slices == 1) hipLaunchKernelGGL(direct_nt<1>, dim3(grid), dim3(256), 0, s, A, B, C, M, N, K, K, K, N);
        else if (slices == 2) hipLaunchKernelGGL(direct_nt<2>, dim3(grid), dim3(256), 0, s, A, B, C, M, N, K, K, K, N);
        else hipLaunchKernelGGL(direct_nt<4>, dim3(grid), dim3(256), 0, s, A, B, C, M, N, K, K, K, N);
    };
    for (int slices : {1, 2, 4}) {
        for (int i = 0; i < 20; ++i) run(slices);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i) run(slices);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> hc(size_t(slices) * M * N);
        CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 200; ++t) {
            const int i = rand() % M, j = rand() % N;
            double ref = 0, got = 0;
            for (int k = 0; k < K; ++k) ref += double(ha[size_t(i) * K + k]) * hb[size_t(j) * K + k];
            for (int sl = 0; sl < slices; ++sl) got += hc[size_t(sl) * M * N + size_t(i) * N + j];
            worst = fmax(worst, fabs(got - ref));
        }
        printf("direct_nt  %d K-slice(s), %3d workgroups: %.2f us per launch   (max |err| on 200 samples %.2e)\n",
               slices, (M / 64) * (N / 64) * slices, 1e3 * ms / 200, worst);
    }
    return 0;
}
