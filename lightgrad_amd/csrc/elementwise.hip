// Elementwise fp32 kernels for gfx950 (HBM-bound: the goal is coalesced 16-byte
// accesses and enough waves in flight; roofline = HBM bandwidth).
//
// Three access paths, chosen on the host from the collapsed stride pattern:
//   flat  : every operand is contiguous or a broadcast scalar  -> float4 grid-stride loop
//   rows  : innermost dimension has stride 1 (or 0) everywhere -> float4 along the inner dim,
//           outer index decomposed once per vector (bias add `(N,512)+(512,)`, keepdims
//           broadcasts of softmax / LayerNorm)
//   gather: arbitrary strides (transposed views)               -> one element per thread
// The arithmetic of each op is a functor; results are identical on all three paths.
//
// Reference semantics: the op strings of opencl/ops.py:40-400 and the numpy expressions
// of cpu/ops.py (cited per functor).  The code generator of opencl/kernels.py:24-195 is
// not reproduced: ops are compiled ahead of time for gfx950.
#include "common.h"
#include "gelu_common.h"

namespace lg {

// ---- op functors: in[NIN] -> out[NOUT] ----------------------------------------------
#define LG_OP(NAME, NIN_, NOUT_, BODY)                                                   \
    struct NAME {                                                                        \
        static constexpr int NIN = NIN_, NOUT = NOUT_;                                   \
        __device__ __forceinline__ static void apply(const float* in, float* out) { BODY } \
    };

LG_OP(OpCopy, 1, 1, out[0] = in[0];)
LG_OP(OpNeg, 1, 1, out[0] = -in[0];)
LG_OP(OpExp, 1, 1, out[0] = expf(in[0]);)
LG_OP(OpLog, 1, 1, out[0] = logf(in[0]);)
// np.maximum(t, 0.0) (cpu/ops.py:226): NaN propagates, -0.0 vs 0.0 -> 0.0
LG_OP(OpRelu, 1, 1, float t = in[0]; out[0] = (t != t) ? t : (t > 0.0f ? t : 0.0f);)
LG_OP(OpSigmoid, 1, 1, out[0] = 1.0f / (1.0f + expf(-in[0]));)
LG_OP(OpTanh, 1, 1, out[0] = tanhf(in[0]);)
LG_OP(OpSin, 1, 1, out[0] = sinf(in[0]);)
LG_OP(OpCos, 1, 1, out[0] = cosf(in[0]);)
LG_OP(OpSqrt, 1, 1, out[0] = sqrtf(in[0]);)
// tanh-approximated gelu (gelu_common.h: shared with the GEMM epilogues)
LG_OP(OpGelu, 1, 1, out[0] = gelu_value(in[0]);)

LG_OP(OpAdd, 2, 1, out[0] = in[0] + in[1];)
LG_OP(OpSub, 2, 1, out[0] = in[0] - in[1];)
LG_OP(OpMul, 2, 1, out[0] = in[0] * in[1];)
LG_OP(OpDiv, 2, 1, out[0] = in[0] / in[1];)
LG_OP(OpPow, 2, 1, out[0] = powf(in[0], in[1]);)
LG_OP(OpReluBwd, 2, 1, out[0] = in[1] * (in[0] >= 0.0f ? 1.0f : 0.0f);)
LG_OP(OpSigmoidBwd, 2, 1, out[0] = in[0] * (1.0f - in[0]) * in[1];)
LG_OP(OpTanhBwd, 2, 1, out[0] = (1.0f - in[0] * in[0]) * in[1];)
LG_OP(OpLogBwd, 2, 1, out[0] = (1.0f / in[0]) * in[1];)
LG_OP(OpSinBwd, 2, 1, out[0] = cosf(in[0]) * in[1];)
LG_OP(OpCosBwd, 2, 1, out[0] = -sinf(in[0]) * in[1];)
// a = x, b = g
LG_OP(OpGeluBwd, 2, 1, out[0] = gelu_grad(in[0], in[1]);)
LG_OP(OpEq, 2, 1, out[0] = in[0] == in[1] ? 1.0f : 0.0f;)
LG_OP(OpGe, 2, 1, out[0] = in[0] >= in[1] ? 1.0f : 0.0f;)

LG_OP(OpMaxBwd, 3, 1, out[0] = in[2] * (in[0] == in[1] ? 1.0f : 0.0f);)
// plain multiply then add (two roundings) - the tape's `a * b + c`, not a fused fma
LG_OP(OpFma, 3, 1, out[0] = __fadd_rn(__fmul_rn(in[0], in[1]), in[2]);)
LG_OP(OpBiasRelu, 2, 1, float t = in[0] + in[1]; out[0] = (t != t) ? t : (t > 0.0f ? t : 0.0f);)

LG_OP(OpMulBwd, 3, 2, out[0] = in[2] * in[1]; out[1] = in[0] * in[2];)
LG_OP(OpDivBwd, 3, 2, out[0] = in[2] / in[1]; out[1] = -in[0] / (in[1] * in[1]) * in[2];)
LG_OP(OpPowBwd, 4, 2, out[0] = in[1] * powf(in[0], in[1] - 1.0f) * in[2]; out[1] = in[2] * in[3] * logf(in[0]);)

// ---- kernel arguments ---------------------------------------------------------------
struct EwArgs {
    const float* in[4];     // nullptr -> operand is `scalar`
    float*       out[2];
    float        scalar;
};

// operand slots inside IterDesc::stride: 0,1 = outputs, 2..5 = inputs
constexpr int kOutSlot = 0, kInSlot = 2;

// ---- flat path -----------------------------------------------------------------------
// step[i] in {0,1}: broadcast scalar-in-memory or contiguous.
template <class Op>
__global__ void __launch_bounds__(256) ew_flat_vec4(EwArgs a, int64_t nvec, int in_step_mask) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float x[4][4];
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i) {
            if (a.in[i] == nullptr) {
                x[i][0] = x[i][1] = x[i][2] = x[i][3] = a.scalar;
            } else if ((in_step_mask >> i) & 1) {
                float4 t = reinterpret_cast<const float4*>(a.in[i])[v];
                x[i][0] = t.x; x[i][1] = t.y; x[i][2] = t.z; x[i][3] = t.w;
            } else {
                float t = a.in[i][0];
                x[i][0] = x[i][1] = x[i][2] = x[i][3] = t;
            }
        }
        float y[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float in[4], out[2];
#pragma unroll
            for (int i = 0; i < Op::NIN; ++i) in[i] = x[i][k];
            Op::apply(in, out);
#pragma unroll
            for (int o = 0; o < Op::NOUT; ++o) y[o][k] = out[o];
        }
#pragma unroll
        for (int o = 0; o < Op::NOUT; ++o)
            reinterpret_cast<float4*>(a.out[o])[v] = make_float4(y[o][0], y[o][1], y[o][2], y[o][3]);
    }
}

template <class Op>
__global__ void __launch_bounds__(256) ew_flat_scalar(EwArgs a, int64_t begin, int64_t n, int in_step_mask) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = begin + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        float in[4], out[2];
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i)
            in[i] = a.in[i] == nullptr ? a.scalar : a.in[i][((in_step_mask >> i) & 1) ? e : 0];
        Op::apply(in, out);
#pragma unroll
        for (int o = 0; o < Op::NOUT; ++o) a.out[o][e] = out[o];
    }
}

// ---- rows path -------------------------------------------------------------------------
// inner dimension (last collapsed dim) has stride 1 for outputs and 1 or 0 for inputs;
// inner % 4 == 0 and every stride-1 operand is 16-byte aligned in every row.
template <class Op, typename IdxT>
__global__ void __launch_bounds__(256) ew_rows_vec4(EwArgs a, IterDesc d, int64_t nvec) {
    const int nd = d.ndim;
    const IdxT inner_v = IdxT(d.shape[nd - 1] / 4);
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        IdxT rem = IdxT(v);
        IdxT col = (rem % inner_v) * 4;
        rem /= inner_v;
        int64_t off[kMaxOps];
#pragma unroll
        for (int s = 0; s < kMaxOps; ++s) off[s] = int64_t(col) * d.stride[s][nd - 1];
        for (int k = nd - 2; k >= 0; --k) {
            IdxT sz = IdxT(d.shape[k]);
            IdxT idx = rem % sz;
            rem /= sz;
#pragma unroll
            for (int s = 0; s < kMaxOps; ++s) off[s] += int64_t(idx) * d.stride[s][k];
        }
        float x[4][4];
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i) {
            if (a.in[i] == nullptr) {
                x[i][0] = x[i][1] = x[i][2] = x[i][3] = a.scalar;
            } else if (d.stride[kInSlot + i][nd - 1] != 0) {
                float4 t = *reinterpret_cast<const float4*>(a.in[i] + off[kInSlot + i]);
                x[i][0] = t.x; x[i][1] = t.y; x[i][2] = t.z; x[i][3] = t.w;
            } else {
                float t = a.in[i][off[kInSlot + i]];
                x[i][0] = x[i][1] = x[i][2] = x[i][3] = t;
            }
        }
        float y[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float in[4], out[2];
#pragma unroll
            for (int i = 0; i < Op::NIN; ++i) in[i] = x[i][k];
            Op::apply(in, out);
#pragma unroll
            for (int o = 0; o < Op::NOUT; ++o) y[o][k] = out[o];
        }
#pragma unroll
        for (int o = 0; o < Op::NOUT; ++o)
            *reinterpret_cast<float4*>(a.out[o] + off[kOutSlot + o]) = make_float4(y[o][0], y[o][1], y[o][2], y[o][3]);
    }
}

// ---- gather path ---------------------------------------------------------------------------
template <class Op, typename IdxT>
__global__ void __launch_bounds__(256) ew_gather(EwArgs a, IterDesc d) {
    const int nd = d.ndim;
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.numel; e += stride) {
        IdxT rem = IdxT(e);
        int64_t off[kMaxOps];
#pragma unroll
        for (int s = 0; s < kMaxOps; ++s) off[s] = 0;
        for (int k = nd - 1; k >= 0; --k) {
            IdxT sz = IdxT(d.shape[k]);
            IdxT idx = rem % sz;
            rem /= sz;
#pragma unroll
            for (int s = 0; s < kMaxOps; ++s) off[s] += int64_t(idx) * d.stride[s][k];
        }
        float in[4], out[2];
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i) in[i] = a.in[i] == nullptr ? a.scalar : a.in[i][off[kInSlot + i]];
        Op::apply(in, out);
#pragma unroll
        for (int o = 0; o < Op::NOUT; ++o) a.out[o][off[kOutSlot + o]] = out[o];
    }
}

// ---- 2-D transposed-operand path --------------------------------------------------------
// out[r][c] = f(...) where outputs and some inputs are row-contiguous (inner stride 1) and the
// other inputs are column-contiguous (outer stride 1): the pattern of `grad += g.T-view`
// (tensor.py:118 fed by transpose.backward) and of contiguous() on a transposed view.
// A 64x64 tile is staged through LDS so that both sides move in full 256-byte rows.
constexpr int kTT = 64;

// Tiles are walked along diagonals: consecutive workgroups differ in BOTH tile coordinates.  With a row-major walk
// the workgroups in flight touch, on the transposed side, addresses that are equal modulo the leading dimension
// (a power of two for the usual sizes), i.e. a handful of HBM channels (measured 2.0 TB/s at 8192^2).
__device__ __forceinline__ void diagonal_tile(unsigned b, int tiles_r, int tiles_c, int64_t& r0, int64_t& c0) {
    const unsigned tc = b % unsigned(tiles_c), j = b / unsigned(tiles_c);
    const unsigned tr = (j + tc) % unsigned(tiles_r);
    r0 = int64_t(tr) * kTT;
    c0 = int64_t(tc) * kTT;
}

template <class Op>
__global__ void __launch_bounds__(256) ew_transposed_tile(EwArgs a, IterDesc d, int tr_mask, int tiles_r, int tiles_c) {
    __shared__ float tile[2][kTT][kTT + 1];   // at most two transposed inputs (host guarantees it)
    const int64_t R = d.shape[0], C = d.shape[1];
    int64_t r0, c0;
    diagonal_tile(blockIdx.x, tiles_r, tiles_c, r0, c0);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    // stage column-contiguous inputs: walk them along their fast axis (rows of the logical matrix)
#pragma unroll
    for (int i = 0; i < Op::NIN; ++i) {
        if (a.in[i] != nullptr && ((tr_mask >> i) & 1)) {
            for (int cc = ty; cc < kTT; cc += 4) {
                int64_t r = r0 + tx, c = c0 + cc;
                if (r < R && c < C)
                    tile[__popc(tr_mask & ((1 << i) - 1))][cc][tx] =
                        a.in[i][r * d.stride[kInSlot + i][0] + c * d.stride[kInSlot + i][1]];
            }
        }
    }
    __syncthreads();
    for (int rr = ty; rr < kTT; rr += 4) {
        int64_t r = r0 + rr, c = c0 + tx;
        if (r < R && c < C) {
            float in[4], out[2];
#pragma unroll
            for (int i = 0; i < Op::NIN; ++i) {
                if (a.in[i] == nullptr) in[i] = a.scalar;
                else if ((tr_mask >> i) & 1) in[i] = tile[__popc(tr_mask & ((1 << i) - 1))][tx][rr];
                else in[i] = a.in[i][r * d.stride[kInSlot + i][0] + c * d.stride[kInSlot + i][1]];
            }
            Op::apply(in, out);
#pragma unroll
            for (int o = 0; o < Op::NOUT; ++o) a.out[o][r * d.stride[kOutSlot + o][0] + c * d.stride[kOutSlot + o][1]] = out[o];
        }
    }
}

// float4 form of the same tile: R % 4 == 0, C % 4 == 0, every operand 16-byte aligned with leading strides % 4 == 0.
// Transposed inputs are read as float4 along their fast axis (16 lanes = one 256-byte run, 16 runs per pass) and
// scattered into tile[r][c]; the compute pass reads 4 consecutive c per lane and moves float4 on the row-major side.
template <class Op, int NT>
__global__ void __launch_bounds__(256) ew_transposed_tile_v4(EwArgs a, IterDesc d, int tr_mask, int tiles_r, int tiles_c) {
    __shared__ float tile[NT][kTT][kTT + 1];
    const int64_t R = d.shape[0], C = d.shape[1];
    int64_t r0, c0;
    diagonal_tile(blockIdx.x, tiles_r, tiles_c, r0, c0);
    const int q = threadIdx.x & 15, line = threadIdx.x >> 4;   // 16 float4 per 64-wide run, 16 runs per pass
#pragma unroll
    for (int i = 0; i < Op::NIN; ++i) {
        if (a.in[i] != nullptr && ((tr_mask >> i) & 1)) {
            const int t = __popc(tr_mask & ((1 << i) - 1));
            const int64_t ld = d.stride[kInSlot + i][1];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int cc = p * 16 + line;
                const int64_t r = r0 + 4 * q, c = c0 + cc;
                if (r < R && c < C) {
                    const float4 v = *reinterpret_cast<const float4*>(a.in[i] + c * ld + r);
                    tile[t][4 * q + 0][cc] = v.x;
                    tile[t][4 * q + 1][cc] = v.y;
                    tile[t][4 * q + 2][cc] = v.z;
                    tile[t][4 * q + 3][cc] = v.w;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int rr = p * 16 + line;
        const int64_t r = r0 + rr, c = c0 + 4 * q;
        if (r < R && c < C) {
            float x[4][4];
#pragma unroll
            for (int i = 0; i < Op::NIN; ++i) {
                if (a.in[i] == nullptr) {
                    x[i][0] = x[i][1] = x[i][2] = x[i][3] = a.scalar;
                } else if ((tr_mask >> i) & 1) {
                    const int t = __popc(tr_mask & ((1 << i) - 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[i][k] = tile[t][rr][4 * q + k];
                } else if (d.stride[kInSlot + i][1] != 0) {
                    const float4 v = *reinterpret_cast<const float4*>(a.in[i] + r * d.stride[kInSlot + i][0] + c);
                    x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
                } else {
                    const float v = a.in[i][r * d.stride[kInSlot + i][0]];
                    x[i][0] = x[i][1] = x[i][2] = x[i][3] = v;
                }
            }
            float y[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float in[4], out[2];
#pragma unroll
                for (int i = 0; i < Op::NIN; ++i) in[i] = x[i][k];
                Op::apply(in, out);
#pragma unroll
                for (int o = 0; o < Op::NOUT; ++o) y[o][k] = out[o];
            }
#pragma unroll
            for (int o = 0; o < Op::NOUT; ++o)
                *reinterpret_cast<float4*>(a.out[o] + r * d.stride[kOutSlot + o][0] + c) = make_float4(y[o][0], y[o][1], y[o][2], y[o][3]);
        }
    }
}

// Matrices the Infinity Cache does not hold, ONE transposed input: a 128 x 128 tile moved by 1024 threads, so that every
// visit to a DRAM page moves 512 bytes on both sides instead of 256 while the CU still holds 32 waves (two workgroups of
// 66 KiB of LDS from the dynamic pool).  Measured at 8192^2 (`a + b.T` / `contiguous(b.T)` / `a += b.T`, TB/s): 64 x 64 by
// 256 threads 5.28 / 5.16 / 5.20; this form 5.90 / 5.72 / 5.87; the same tile by 256 threads 4.30 / 4.03 / 4.28 (8 waves
// per CU); 256 x 128 and 128 x 256 by 1024 threads 5.68 / 5.22 and 5.61 / 5.17; walking 2 x 2 or 4 x 4 groups of 64 x 64
// tiles together: no change; a lane map without LDS bank conflicts: 5.73 / 5.64 (`profiles/r3/transposed_tiles.txt`).
template <class Op, int TR, int TC, int NTH>
__global__ void __launch_bounds__(NTH) ew_transposed_big_v4(EwArgs a, IterDesc d, int ti, int tiles_r, int tiles_c) {
    extern __shared__ float big_tile[];
    constexpr int PITCH = TC + 1;
    constexpr int QR = TR / 4, QC = TC / 4;            // float4 per run on the transposed side / per output row
    constexpr int RUNS = NTH / QR, ROWS = NTH / QC;    // runs (output rows) one pass of the workgroup covers
    const int64_t R = d.shape[0], C = d.shape[1];
    const unsigned b = blockIdx.x;
    const unsigned tc = b % unsigned(tiles_c), j = b / unsigned(tiles_c);
    const unsigned tr = (j + tc) % unsigned(tiles_r);
    const int64_t r0 = int64_t(tr) * TR, c0 = int64_t(tc) * TC;
    {
        const int q = threadIdx.x % QR, line = threadIdx.x / QR;
        const int64_t ld = d.stride[kInSlot + ti][1];
        const float* src = a.in[ti];
#pragma unroll
        for (int p = 0; p < TC / RUNS; ++p) {
            const int cc = p * RUNS + line;
            const int64_t r = r0 + 4 * q, c = c0 + cc;
            if (r < R && c < C) {
                const float4 v = *reinterpret_cast<const float4*>(src + c * ld + r);
                big_tile[(4 * q + 0) * PITCH + cc] = v.x;
                big_tile[(4 * q + 1) * PITCH + cc] = v.y;
                big_tile[(4 * q + 2) * PITCH + cc] = v.z;
                big_tile[(4 * q + 3) * PITCH + cc] = v.w;
            }
        }
    }
    __syncthreads();
    const int q = threadIdx.x % QC, line = threadIdx.x / QC;
#pragma unroll
    for (int p = 0; p < TR / ROWS; ++p) {
        const int rr = p * ROWS + line;
        const int64_t r = r0 + rr, c = c0 + 4 * q;
        if (r < R && c < C) {
            float x[4][4];
#pragma unroll
            for (int i = 0; i < Op::NIN; ++i) {
                if (a.in[i] == nullptr) {
                    x[i][0] = x[i][1] = x[i][2] = x[i][3] = a.scalar;
                } else if (i == ti) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[i][k] = big_tile[rr * PITCH + 4 * q + k];
                } else if (d.stride[kInSlot + i][1] != 0) {
                    const float4 v = *reinterpret_cast<const float4*>(a.in[i] + r * d.stride[kInSlot + i][0] + c);
                    x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
                } else {
                    const float v = a.in[i][r * d.stride[kInSlot + i][0]];
                    x[i][0] = x[i][1] = x[i][2] = x[i][3] = v;
                }
            }
            float y[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float in[4], out[2];
#pragma unroll
                for (int i = 0; i < Op::NIN; ++i) in[i] = x[i][k];
                Op::apply(in, out);
#pragma unroll
                for (int o = 0; o < Op::NOUT; ++o) y[o][k] = out[o];
            }
#pragma unroll
            for (int o = 0; o < Op::NOUT; ++o)
                *reinterpret_cast<float4*>(a.out[o] + r * d.stride[kOutSlot + o][0] + c) = make_float4(y[o][0], y[o][1], y[o][2], y[o][3]);
        }
    }
}

template <class Op, int TR, int TC, int NTH>
static void launch_transposed_big(hipStream_t s, const EwArgs& args, const IterDesc& d, int ti) {
    constexpr size_t lds = size_t(TR) * (TC + 1) * sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ew_transposed_big_v4<Op, TR, TC, NTH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        once = true;
    }
    const int64_t tiles_r = (d.shape[0] + TR - 1) / TR, tiles_c = (d.shape[1] + TC - 1) / TC;
    hipLaunchKernelGGL((ew_transposed_big_v4<Op, TR, TC, NTH>), dim3(unsigned(tiles_r * tiles_c)), dim3(NTH), lds, s, args, d, ti,
                       int(tiles_r), int(tiles_c));
}

// ---- dense 2-D outputs, any 2-D inputs ---------------------------------------------------
// outputs (and the inputs flagged in dense_mask) are dense rows x cols and move as float4 over the FLAT index, whatever
// cols is; the other inputs (row / column broadcasts, odd views) are fetched per element at row*s0 + col*s1.  This is
// the bias add of a layer whose width is not a multiple of 4 ((1024, 30522) + (30522,): 2.6 -> 5 TB/s over the
// per-element gather).
template <class Op>
__global__ void __launch_bounds__(256) ew_flat2d_vec4(EwArgs a, IterDesc d, int dense_mask, int64_t nvec) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= nvec) return;
    const int64_t cols = d.shape[1], e0 = v * 4;
    int64_t row = e0 / cols, col = e0 - row * cols;
    float x[4][4];
#pragma unroll
    for (int i = 0; i < Op::NIN; ++i) {
        if (a.in[i] == nullptr) {
            x[i][0] = x[i][1] = x[i][2] = x[i][3] = a.scalar;
        } else if ((dense_mask >> i) & 1) {
            const float4 t = reinterpret_cast<const float4*>(a.in[i])[v];
            x[i][0] = t.x; x[i][1] = t.y; x[i][2] = t.z; x[i][3] = t.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i)
            if (a.in[i] != nullptr && !((dense_mask >> i) & 1))
                x[i][k] = a.in[i][row * d.stride[kInSlot + i][0] + col * d.stride[kInSlot + i][1]];
        if (++col == cols) { col = 0; ++row; }
    }
    float y[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float in[4], out[2];
#pragma unroll
        for (int i = 0; i < Op::NIN; ++i) in[i] = x[i][k];
        Op::apply(in, out);
#pragma unroll
        for (int o = 0; o < Op::NOUT; ++o) y[o][k] = out[o];
    }
#pragma unroll
    for (int o = 0; o < Op::NOUT; ++o)
        reinterpret_cast<float4*>(a.out[o])[v] = make_float4(y[o][0], y[o][1], y[o][2], y[o][3]);
}

// ---- host dispatch ----------------------------------------------------------------------------
template <class Op>
static int launch_ew(const EwArgs& args, const IterDesc& d) {
    hipStream_t s = rt().stream;
    if (d.numel == 0) return LG_OK;
    const int nd = d.ndim;
    const int nslots_in = Op::NIN, nslots_out = Op::NOUT;

    // classify
    bool flat = (nd == 1);
    int step_mask = 0;
    if (flat) {
        for (int o = 0; o < nslots_out; ++o) flat = flat && (d.stride[kOutSlot + o][0] == 1 || d.numel == 1);
        for (int i = 0; i < nslots_in; ++i) {
            if (!args.in[i]) continue;
            int64_t st = d.stride[kInSlot + i][0];
            if (st == 1 || d.numel == 1) step_mask |= (st == 1) << i;
            else if (st != 0) flat = false;
        }
    }
    if (flat) {
        bool vec_ok = true;
        for (int o = 0; o < nslots_out; ++o) vec_ok = vec_ok && aligned16(args.out[o]);
        for (int i = 0; i < nslots_in; ++i)
            if (args.in[i] && ((step_mask >> i) & 1)) vec_ok = vec_ok && aligned16(args.in[i]);
        int64_t nvec = vec_ok ? d.numel / 4 : 0;
        if (nvec > 0) {
            hipLaunchKernelGGL((ew_flat_vec4<Op>), dim3(stream_grid(nvec)), dim3(256), 0, s, args, nvec, step_mask);
        }
        int64_t tail_begin = nvec * 4;
        if (tail_begin < d.numel) {
            // numel == 1 with stride 0 outputs is handled here as well (e reads index 0)
            int mask = step_mask;
            hipLaunchKernelGGL((ew_flat_scalar<Op>), dim3(stream_grid(d.numel - tail_begin)), dim3(256), 0, s, args,
                               tail_begin, d.numel, mask);
        }
        return LG_OK;
    }

    const bool small_idx = d.numel < (int64_t(1) << 31);

    // rows path?
    bool rows = (d.shape[nd - 1] % 4 == 0);
    for (int o = 0; o < nslots_out && rows; ++o) {
        rows = rows && d.stride[kOutSlot + o][nd - 1] == 1 && aligned16(args.out[o]);
        for (int k = 0; k < nd - 1 && rows; ++k) rows = rows && (d.stride[kOutSlot + o][k] % 4 == 0);
    }
    for (int i = 0; i < nslots_in && rows; ++i) {
        if (!args.in[i]) continue;
        int64_t st = d.stride[kInSlot + i][nd - 1];
        if (st == 1) {
            rows = rows && aligned16(args.in[i]);
            for (int k = 0; k < nd - 1 && rows; ++k) rows = rows && (d.stride[kInSlot + i][k] % 4 == 0);
        } else if (st != 0) {
            rows = false;
        }
    }
    if (rows) {
        int64_t nvec = d.numel / 4;
        if (small_idx)
            hipLaunchKernelGGL((ew_rows_vec4<Op, uint32_t>), dim3(stream_grid(nvec)), dim3(256), 0, s, args, d, nvec);
        else
            hipLaunchKernelGGL((ew_rows_vec4<Op, uint64_t>), dim3(stream_grid(nvec)), dim3(256), 0, s, args, d, nvec);
        return LG_OK;
    }

    // 2-D with transposed inputs?
    if (nd == 2 && d.shape[0] >= 16 && d.shape[1] >= 16) {
        bool ok = true;
        int tr_mask = 0;
        for (int o = 0; o < nslots_out; ++o) ok = ok && d.stride[kOutSlot + o][1] == 1;
        for (int i = 0; i < nslots_in && ok; ++i) {
            if (!args.in[i]) continue;
            int64_t s0 = d.stride[kInSlot + i][0], s1 = d.stride[kInSlot + i][1];
            if (s1 == 1 || s1 == 0) continue;      // row-contiguous or broadcast along the row
            if (s0 == 1) tr_mask |= 1 << i;        // column-contiguous
            else ok = false;
        }
        if (ok && tr_mask != 0 && __builtin_popcount(tr_mask) <= 2) {
            int64_t tiles_r = (d.shape[0] + kTT - 1) / kTT, tiles_c = (d.shape[1] + kTT - 1) / kTT;
            if (tiles_r * tiles_c < (int64_t(1) << 31)) {
                bool v4 = d.shape[0] % 4 == 0 && d.shape[1] % 4 == 0;
                for (int o = 0; o < nslots_out && v4; ++o) v4 = aligned16(args.out[o]) && d.stride[kOutSlot + o][0] % 4 == 0;
                for (int i = 0; i < nslots_in && v4; ++i) {
                    if (!args.in[i]) continue;
                    const int64_t s0 = d.stride[kInSlot + i][0], s1 = d.stride[kInSlot + i][1];
                    if ((tr_mask >> i) & 1) v4 = aligned16(args.in[i]) && s1 % 4 == 0;
                    else if (s1 == 1) v4 = aligned16(args.in[i]) && s0 % 4 == 0;
                }
                const bool one = __builtin_popcount(tr_mask) == 1;
                if (v4 && one && d.numel >= (int64_t(1) << 25))                   // 128 MiB and more per operand
                    launch_transposed_big<Op, 128, 128, 1024>(s, args, d, __builtin_ctz(tr_mask));
                else if (v4 && one)
                    hipLaunchKernelGGL((ew_transposed_tile_v4<Op, 1>), dim3(unsigned(tiles_r * tiles_c)), dim3(256), 0, s, args, d,
                                       tr_mask, int(tiles_r), int(tiles_c));
                else if (v4)
                    hipLaunchKernelGGL((ew_transposed_tile_v4<Op, 2>), dim3(unsigned(tiles_r * tiles_c)), dim3(256), 0, s, args, d,
                                       tr_mask, int(tiles_r), int(tiles_c));
                else
                    hipLaunchKernelGGL((ew_transposed_tile<Op>), dim3(unsigned(tiles_r * tiles_c)), dim3(256), 0, s, args, d,
                                       tr_mask, int(tiles_r), int(tiles_c));
                return LG_OK;
            }
        }
    }

    // dense 2-D outputs with broadcast / strided 2-D inputs
    if (nd == 2 && d.numel % 4 == 0 && d.numel / 4 < (int64_t(1) << 31) * 256) {
        bool ok = true;
        int dense_mask = 0;
        for (int o = 0; o < nslots_out; ++o)
            ok = ok && d.stride[kOutSlot + o][1] == 1 && d.stride[kOutSlot + o][0] == d.shape[1] && aligned16(args.out[o]);
        for (int i = 0; i < nslots_in && ok; ++i) {
            if (!args.in[i]) continue;
            if (d.stride[kInSlot + i][1] == 1 && d.stride[kInSlot + i][0] == d.shape[1] && aligned16(args.in[i])) dense_mask |= 1 << i;
        }
        if (ok) {
            const int64_t nvec = d.numel / 4;
            hipLaunchKernelGGL((ew_flat2d_vec4<Op>), dim3(unsigned((nvec + 255) / 256)), dim3(256), 0, s, args, d, dense_mask, nvec);
            return LG_OK;
        }
    }

    if (small_idx)
        hipLaunchKernelGGL((ew_gather<Op, uint32_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, args, d);
    else
        hipLaunchKernelGGL((ew_gather<Op, uint64_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, args, d);
    return LG_OK;
}

}  // namespace lg

using namespace lg;

extern "C" int lg_ew(int op, int ndim, const int64_t* shape,
                     void* out0, const int64_t* out0_strides,
                     void* out1, const int64_t* out1_strides,
                     const void* a, const int64_t* a_strides,
                     const void* b, const int64_t* b_strides,
                     const void* c, const int64_t* c_strides,
                     const void* d, const int64_t* d_strides,
                     float scalar) {
    LG_REQUIRE_INIT();
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_ew: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(ndim == 0 || shape != nullptr, "lg_ew: shape is NULL");
    LG_ARG(out0 != nullptr, "lg_ew: out0 is NULL");

    int nin, nout;
    if (op >= LG_EW_COPY && op <= LG_EW_GELU) { nin = 1; nout = 1; }
    else if (op >= LG_EW_ADD && op <= LG_EW_GELU_BWD) { nin = 2; nout = 1; }
    else if (op >= LG_EW_MAX_BWD && op <= LG_EW_FMA) { nin = 3; nout = 1; }
    else if (op == LG_EW_MUL_BWD || op == LG_EW_DIV_BWD) { nin = 3; nout = 2; }
    else if (op == LG_EW_POW_BWD) { nin = 4; nout = 2; }
    else { set_error("lg_ew: unknown op id %d", op); return LG_EINVAL; }

    const void* ins[4] = {a, b, c, d};
    const int64_t* in_st[4] = {a_strides, b_strides, c_strides, d_strides};
    int nscalar = 0;
    for (int i = 0; i < nin; ++i) {
        if (ins[i] == nullptr) ++nscalar;
        else LG_ARG(ndim == 0 || in_st[i] != nullptr, "lg_ew: operand %d has no strides", i);
    }
    LG_ARG(nscalar <= 1, "lg_ew: at most one operand may be the scalar");
    LG_ARG(nscalar < nin, "lg_ew: at least one operand must be a tensor");
    if (nout == 2) LG_ARG(out1 != nullptr, "lg_ew: op %d writes two outputs, out1 is NULL", op);
    LG_ARG(ndim == 0 || out0_strides != nullptr, "lg_ew: out0 has no strides");
    LG_ARG(nout == 1 || ndim == 0 || out1_strides != nullptr, "lg_ew: out1 has no strides");

    const int64_t* strides[kMaxOps] = {out0_strides, nout == 2 ? out1_strides : nullptr,
                                       ins[0] ? in_st[0] : nullptr, (nin > 1 && ins[1]) ? in_st[1] : nullptr,
                                       (nin > 2 && ins[2]) ? in_st[2] : nullptr, (nin > 3 && ins[3]) ? in_st[3] : nullptr};
    IterDesc desc;
    LG_ARG(build_iter(ndim, shape, strides, kMaxOps, desc), "lg_ew: bad shape");
    // outputs must not broadcast (every element written exactly once)
    for (int o = 0; o < nout; ++o)
        for (int k = 0; k < desc.ndim; ++k)
            LG_ARG(desc.shape[k] == 1 || desc.stride[o][k] != 0, "lg_ew: output %d has a zero stride over an extent > 1", o);

    EwArgs args;
    for (int i = 0; i < 4; ++i) args.in[i] = i < nin ? static_cast<const float*>(ins[i]) : nullptr;
    args.out[0] = static_cast<float*>(out0);
    args.out[1] = nout == 2 ? static_cast<float*>(out1) : nullptr;
    args.scalar = scalar;

    int rc;
    switch (op) {
#define LG_CASE(ID, OP) case ID: rc = launch_ew<OP>(args, desc); break;
        LG_CASE(LG_EW_COPY, OpCopy) LG_CASE(LG_EW_NEG, OpNeg) LG_CASE(LG_EW_EXP, OpExp) LG_CASE(LG_EW_LOG, OpLog)
        LG_CASE(LG_EW_RELU, OpRelu) LG_CASE(LG_EW_SIGMOID, OpSigmoid) LG_CASE(LG_EW_TANH, OpTanh)
        LG_CASE(LG_EW_SIN, OpSin) LG_CASE(LG_EW_COS, OpCos) LG_CASE(LG_EW_SQRT, OpSqrt) LG_CASE(LG_EW_GELU, OpGelu)
        LG_CASE(LG_EW_GELU_BWD, OpGeluBwd)
        LG_CASE(LG_EW_ADD, OpAdd) LG_CASE(LG_EW_SUB, OpSub) LG_CASE(LG_EW_MUL, OpMul) LG_CASE(LG_EW_DIV, OpDiv)
        LG_CASE(LG_EW_POW, OpPow) LG_CASE(LG_EW_RELU_BWD, OpReluBwd) LG_CASE(LG_EW_SIGMOID_BWD, OpSigmoidBwd)
        LG_CASE(LG_EW_TANH_BWD, OpTanhBwd) LG_CASE(LG_EW_LOG_BWD, OpLogBwd) LG_CASE(LG_EW_SIN_BWD, OpSinBwd)
        LG_CASE(LG_EW_COS_BWD, OpCosBwd) LG_CASE(LG_EW_EQ, OpEq) LG_CASE(LG_EW_GE, OpGe)
        LG_CASE(LG_EW_MAX_BWD, OpMaxBwd) LG_CASE(LG_EW_FMA, OpFma) LG_CASE(LG_EW_BIAS_RELU, OpBiasRelu)
        LG_CASE(LG_EW_MUL_BWD, OpMulBwd) LG_CASE(LG_EW_DIV_BWD, OpDivBwd) LG_CASE(LG_EW_POW_BWD, OpPowBwd)
#undef LG_CASE
        default: set_error("lg_ew: unknown op id %d", op); return LG_EINVAL;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
