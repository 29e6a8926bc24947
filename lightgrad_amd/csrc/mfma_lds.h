// Wave-level MFMA building blocks over LDS operands (gfx950, fp32), shared by attention.hip and gemm_shortk.hip.
//
// v_mfma_f32_32x32x2f32 takes one float per lane: lane (r, h) = (lane & 31, lane >> 5) holds A[r][k + h] and B[k + h][r].  Any
// order of the k values will do as long as A and B agree, so a wave walks K in groups of 8 with lane half h on
// k = 8j + 4h .. 8j + 4h + 3: an operand that is contiguous along K is then ONE ds_read_b128 per four MFMAs, the others four
// ds_read_b32 of 32 consecutive floats.  Row pitches: K-contiguous rows + 4 floats (16 lanes of a b128 read hit 16 different
// 16-byte slots), M/N-contiguous rows = 8 mod 16 floats (the two lane halves, 4 rows apart, land on different halves of the
// 64 banks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lg {

typedef float af32x16 __attribute__((ext_vector_type(16)));
typedef float af32x4 __attribute__((ext_vector_type(4)));

// acc (32 x 32) += A (32 x K) @ B (K x 32) by one wave; K a multiple of 8.
//   AKC: A[m][k] at As[m * pa + k], else at As[k * pa + m];   BKC: B[k][n] at Bs[n * pb + k], else at Bs[k * pb + n]
template <bool KC>
__device__ __forceinline__ void fetch4(float (&x)[4], const float* base, int pitch, int k, int r) {
    if constexpr (KC) {
        const af32x4 t = *reinterpret_cast<const af32x4*>(base + r * pitch + k);
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = base[(k + e) * pitch + r];
    }
}

// G groups of 8 k-values from k0 on: every operand fetch is issued before the first MFMA, so all but the first LDS round
// trip hide behind the matrix cores
template <bool AKC, bool BKC, int G>
__device__ __forceinline__ void mma_groups(af32x16& acc, const float* As, int pa, const float* Bs, int pb, int k0, int r, int h) {
    float a[G][4], b[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        fetch4<AKC>(a[g], As, pa, k0 + 8 * g + 4 * h, r);
        fetch4<BKC>(b[g], Bs, pb, k0 + 8 * g + 4 * h, r);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][e], b[g][e], acc, 0, 0, 0);
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void wave_mma(af32x16& acc, const float* As, int pa, const float* Bs, int pb, int K, int r, int h) {
    int k = 0;
    for (; k + 32 <= K; k += 32) mma_groups<AKC, BKC, 4>(acc, As, pa, Bs, pb, k, r, h);
    for (; k < K; k += 8) mma_groups<AKC, BKC, 1>(acc, As, pa, Bs, pb, k, r, h);
}

// row of accumulator element e in lane half h (column = lane & 31)
__device__ __forceinline__ int acc_row(int e, int h) { return 4 * h + (e & 3) + 8 * (e >> 2); }

__device__ __forceinline__ af32x16 zero16() {
    af32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// rows x D floats from global (row pitch ld) into registers (N float4 per thread, all loads in flight together), and from
// there to LDS (row pitch pitch)
template <int D, int N>
__device__ __forceinline__ void load_rows(af32x4 (&v)[N], const float* src, int64_t ld, int rows) {
    constexpr int Q = D / 4;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f0 = threadIdx.x + i * 256, f = f0 < rows * Q ? f0 : 0;      // no branch: a branch costs the loads their overlap
        v[i] = *reinterpret_cast<const af32x4*>(src + int64_t(f / Q) * ld + (f % Q) * 4);
    }
}
template <int D, int N>
__device__ __forceinline__ void store_rows(const af32x4 (&v)[N], float* dst, int pitch, int rows) {
    constexpr int Q = D / 4;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < rows * Q) *reinterpret_cast<af32x4*>(dst + (f / Q) * pitch + (f % Q) * 4) = v[i];
    }
}

}  // namespace lg
