// Fused mean-squared-error forward: the reference's loss.mse (loss.py:4-12) runs
//   err = y + (-y_hat);  (err ** 2).mean() / 2  =  (sum(err*err) * (1/N)) * 0.5
// as seven tape ops; here one pass writes `err` (saved for backward) and block partial sums, and the last
// step applies the same two scalings with the same fp32 roundings.  HBM-bound: 8 B read + 4 B written per element.
#include "common.h"

namespace lg {

// sum over the block (blockDim.x = 64 * NW); every thread returns the same value
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += lds[w];
    return s;
}

__global__ void __launch_bounds__(256) mse_partial(const float* __restrict__ y, const float* __restrict__ t, float* __restrict__ err,
                                                   float* __restrict__ partial, int64_t n) {
    __shared__ float lds[4];
    float acc = 0.f;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float e = y[i] + (-t[i]);
        err[i] = e;
        acc += e * e;
    }
    const float s = block_sum<4>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one block: sums `count` partials in index order and applies (sum * inv_n) * 0.5
__global__ void __launch_bounds__(256) mse_final(const float* __restrict__ partial, int count, float* __restrict__ loss, float inv_n) {
    __shared__ float lds[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    const float s = block_sum<4>(acc, lds);
    if (threadIdx.x == 0) loss[0] = (s * inv_n) * 0.5f;
}

// small inputs: a single block does both steps
__global__ void __launch_bounds__(1024) mse_single(const float* __restrict__ y, const float* __restrict__ t, float* __restrict__ err,
                                                   float* __restrict__ loss, int64_t n, float inv_n, int vec) {
    __shared__ float lds[16];
    float acc = 0.f;
    int64_t done = 0;
    if (vec) {      // all three pointers 16-byte aligned
        const int64_t nv = n / 4;
        for (int64_t i = threadIdx.x; i < nv; i += 1024) {
            const float4 a = reinterpret_cast<const float4*>(y)[i], b = reinterpret_cast<const float4*>(t)[i];
            float4 e;
            e.x = a.x + (-b.x); e.y = a.y + (-b.y); e.z = a.z + (-b.z); e.w = a.w + (-b.w);
            reinterpret_cast<float4*>(err)[i] = e;
            acc += (e.x * e.x + e.y * e.y) + (e.z * e.z + e.w * e.w);
        }
        done = nv * 4;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += 1024) {
        const float e = y[i] + (-t[i]);
        err[i] = e;
        acc += e * e;
    }
    const float s = block_sum<16>(acc, lds);
    if (threadIdx.x == 0) loss[0] = (s * inv_n) * 0.5f;
}

}  // namespace lg

using namespace lg;

extern "C" int lg_mse_f32(const float* y, const float* t, float* err, float* loss, int64_t n) {
    LG_REQUIRE_INIT();
    LG_ARG(n > 0, "lg_mse_f32: empty input");
    LG_ARG(y && t && err && loss, "lg_mse_f32: NULL pointer");
    hipStream_t s = rt().stream;
    const float inv_n = float(1.0 / double(n));      // python's `s.numel() / t.numel()` rounded once to fp32
    if (n <= 32768) {
        hipLaunchKernelGGL(mse_single, dim3(1), dim3(1024), 0, s, y, t, err, loss, n, inv_n,
                           int(aligned16(y) && aligned16(t) && aligned16(err)));
    } else {
        const unsigned blocks = unsigned(((n + 1023) / 1024) < 4096 ? ((n + 1023) / 1024) : 4096);   // partial sums to combine
        float* partial = nullptr;
        int rc = lg_malloc(reinterpret_cast<void**>(&partial), blocks * sizeof(float));
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL(mse_partial, dim3(blocks), dim3(256), 0, s, y, t, err, partial, n);
        hipLaunchKernelGGL(mse_final, dim3(1), dim3(256), 0, s, partial, int(blocks), loss, inv_n);
        rc = lg_free(partial);
        if (rc != LG_OK) return rc;
    }
    LG_CHECK_LAUNCH();
    return LG_OK;
}
