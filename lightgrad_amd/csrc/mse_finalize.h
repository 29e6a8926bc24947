// The scalar of loss.mse from the per-row sums of squared errors that head_fwd leaves behind (head.hip): shared by the kernels
// that may finish it - head_bwd's spare workgroup, lg_mse_finalize_f32, and the spare workgroup of the three-product launch of
// gemm.hip - so the loss has the same bits whoever does.
#pragma once
#include "common.h"

namespace lg {

// (sum_r row_loss[r] * inv_n) * 0.5 by ONE workgroup of 256 threads, always in the same order: thread t takes rows
// t, t+256, ..., then a butterfly over the wavefront, then the four wavefronts.  Shared by head_bwd's loss workgroup and
// by mse_finalize, so the loss has the same bits whoever finishes it.
__device__ __forceinline__ void finalize_loss(const float* __restrict__ row_loss, int64_t rows, float inv_n, float* __restrict__ loss,
                                              float* lds4) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) {                                   // 256 summing threads whatever the size of the workgroup: one order
        float v = 0.f;
        for (int64_t i = tid; i < rows; i += 256) v += row_loss[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) lds4[wave] = v;
    }
    __syncthreads();
    if (tid == 0) loss[0] = (((lds4[0] + lds4[1]) + (lds4[2] + lds4[3])) * inv_n) * 0.5f;
}

}  // namespace lg
