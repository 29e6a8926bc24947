// The forward pass of a skinny output layer and its mse loss for ONE row, by one wavefront (head.hip: head_fwd; gemm.hip: the
// same rows computed by waiting workgroups at the end of the hidden layer's GEMM launch).  One body, so the bits are the same
// whichever launch runs it.
#pragma once
#include "common.h"

namespace lg {

__device__ __forceinline__ float relu_keep_nan(float x) { return (x != x) ? x : (x > 0.0f ? x : 0.0f); }   // np.maximum(x, 0)

struct HeadFwd {
    const float* x;        // [rows, hidden], row pitch ldx
    const float* w;        // [outs, hidden] dense
    const float* bias;     // [outs] or NULL
    const float* target;   // [rows, outs] dense
    float*       y;        // [rows, outs]
    float*       err;      // [rows, outs]
    float*       row_loss; // [rows]
    float*       gpre;     // [rows, hidden] dense or NULL (relu != 0 only): (err @ W) * (x >= 0) - what head_bwd's tile workgroups would
                           // write for the gradient `err` itself, i.e. relu.backward's result when the loss is the root of backward()
    float*       dx;       // [rows, hidden] dense, with gpre: err @ W, the layer's input gradient before the mask
    int64_t      rows, ldx;
    int          hidden, outs, relu;
};

template <int OMAX>
__device__ __forceinline__ void head_dot_chunk(float (&acc)[OMAX], float4 h, const float* w_lds, int k, const HeadFwd& a) {
    if (a.relu) { h.x = relu_keep_nan(h.x); h.y = relu_keep_nan(h.y); h.z = relu_keep_nan(h.z); h.w = relu_keep_nan(h.w); }
#pragma unroll
    for (int j = 0; j < OMAX; ++j) {
        if (j < a.outs) {
            const float4 w = *reinterpret_cast<const float4*>(w_lds + j * a.hidden + k);
            acc[j] = __builtin_fmaf(h.x, w.x, __builtin_fmaf(h.y, w.y, __builtin_fmaf(h.z, w.z, __builtin_fmaf(h.w, w.w, acc[j]))));
        }
    }
}

// the row of head_bwd's dx / g_pre tiles for g = err: the same chain of fused multiply-adds over j (ascending, from 0) and the
// same mask - head_bwd's bits
template <int OMAX>
__device__ __forceinline__ void head_grad_chunk(const float (&ev)[OMAX], const float4 h, const float* w_lds, int k, const HeadFwd& a,
                                                float* q, float* qd) {
    float4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < OMAX; ++j) {
        if (j < a.outs) {
            const float4 w = *reinterpret_cast<const float4*>(w_lds + j * a.hidden + k);
            d.x = __builtin_fmaf(ev[j], w.x, d.x); d.y = __builtin_fmaf(ev[j], w.y, d.y);
            d.z = __builtin_fmaf(ev[j], w.z, d.z); d.w = __builtin_fmaf(ev[j], w.w, d.w);
        }
    }
    *reinterpret_cast<float4*>(qd + k) = d;
    d.x *= (h.x >= 0.0f ? 1.0f : 0.0f); d.y *= (h.y >= 0.0f ? 1.0f : 0.0f);
    d.z *= (h.z >= 0.0f ? 1.0f : 0.0f); d.w *= (h.w >= 0.0f ? 1.0f : 0.0f);
    *reinterpret_cast<float4*>(q + k) = d;
}

// COHERENT: the row of x was written by OTHER workgroups of the same launch (write-through stores, then a ticket this wavefront
// has seen): it is read with sc1 loads, never served from a stale cache line, once - the four 16-byte pieces a lane may own
// (hidden <= 1024) stay in registers for the second use.
template <int OMAX, bool COHERENT>
__device__ __forceinline__ void head_fwd_row(const HeadFwd& a, const float* w_lds, int64_t row, int lane) {
    float acc[OMAX];
#pragma unroll
    for (int j = 0; j < OMAX; ++j) acc[j] = 0.f;
    const float* p = a.x + row * a.ldx;
    [[maybe_unused]] float4 kept[4];
    if constexpr (COHERENT) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, a.hidden * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane * 4 + 256 * c) * 4, 0, 16);      // (beyond hidden: zeros)
            kept[c] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = lane * 4 + 256 * c;
            if (k < a.hidden) head_dot_chunk<OMAX>(acc, kept[c], w_lds, k, a);
        }
    } else {
        for (int k = lane * 4; k < a.hidden; k += 256) head_dot_chunk<OMAX>(acc, *reinterpret_cast<const float4*>(p + k), w_lds, k, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < OMAX; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    float s = 0.f;                                                      // lane j < outs keeps output j
#pragma unroll
    for (int j = 0; j < OMAX; ++j) s = (lane == j) ? acc[j] : s;
    float e2 = 0.f, e = 0.f;
    if (lane < a.outs) {
        const float yv = a.bias ? s + a.bias[lane] : s;
        e = yv + (-a.target[row * a.outs + lane]);
        a.y[row * a.outs + lane] = yv;
        a.err[row * a.outs + lane] = e;
        e2 = e * e;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);          // OMAX <= 16: lanes 0..15
    if (lane == 0) a.row_loss[row] = e2;
    if (a.gpre) {
        // while the row's pre-activations are in the cache (or in registers) and W is in LDS
        float ev[OMAX];
#pragma unroll
        for (int j = 0; j < OMAX; ++j) ev[j] = __shfl(e, j, 64);
        float* q = a.gpre + row * a.hidden;
        float* qd = a.dx + row * a.hidden;
        if constexpr (COHERENT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = lane * 4 + 256 * c;
                if (k < a.hidden) head_grad_chunk<OMAX>(ev, kept[c], w_lds, k, a, q, qd);
            }
        } else {
            for (int k = lane * 4; k < a.hidden; k += 256) head_grad_chunk<OMAX>(ev, *reinterpret_cast<const float4*>(p + k), w_lds, k, a, q, qd);
        }
    }
}

}  // namespace lg
