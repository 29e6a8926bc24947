// The skinny output layer of a classifier and its loss as two launches instead of five
// (VERDICT r1 item 3).  nn.Linear with <= 16 output features ("the N = 10 head" of the
// MNIST MLP: y = relu(pre) @ W^T + b, 1024 x 10 x 512) has no use for matrix cores: it moves 2 MB and
// does 10 MFLOP, so as MFMA tiles it was pure launch latency (forward 9.8, loss 4.9, dW+db 11.5, dh 5.9,
// relu-backward 5.2 us).  Here:
//
//   head_fwd   y = act(x) @ W^T + b,  err = y + (-target),  row_loss[r] = sum_j err[r][j]^2
//              one wavefront per row (lanes split K, W staged once per workgroup in LDS, butterfly sum)
//              - nn.py:96 + loss.py:4-8 of the reference.  The scalar loss (sum row_loss * (1/N)) * 0.5 (loss.py:9-10) is
//              NOT folded here: a cross-workgroup hand-off inside the launch costs 4 us (measured: 9.0 us with it, 5.0
//              without; an empty launch is 3.0) - it is finished by one extra workgroup of head_bwd, or by
//              lg_mse_finalize_f32 when somebody reads the loss first.
//   head_bwd   dx = g @ W,  g_pre = dx * (pre >= 0),  dW = g^T @ act(x),  db = column sums of g
//              one launch, three kinds of workgroups, no hand-off between them: "slab" workgroups own 8 columns of dW
//              over ALL rows (g staged in LDS, sums over thread rows through LDS in a fixed order), "tile" workgroups
//              write 256 x 32 tiles of dx / g_pre (every element needs only its row of g and its column of W), and one
//              workgroup finishes the loss of the forward pass
//              - what linear.backward + relu.backward of the tape compute (cpu/ops.py:114-116, :229, func.py:50-56).
//
// act = relu when the tape's relu is still lazy (autograd/hip/ops.py), identity otherwise.  The dot products are chains
// of fused multiply-adds in a fixed order (like the BLAS / MFMA GEMMs they replace, they are not expression-by-expression
// restatements of numpy ufuncs): results agree with the GEMM form to rounding (<= 1e-6 relative) and are
// bit-reproducible from run to run.  Everything elementwise (bias add, err, the relu mask) rounds once per operation
// like the tape's ops.
#include "common.h"
#include "adam_common.h"
#include "mse_finalize.h"
#include "head_fwd_body.h"
#include <cstdlib>

namespace lg {


__global__ void __launch_bounds__(256) mse_finalize(const float* __restrict__ row_loss, int64_t rows, float inv_n, float* __restrict__ loss) {
    __shared__ float lds4[4];
    finalize_loss(row_loss, rows, inv_n, loss, lds4);
}

template <int OMAX>
__global__ void __launch_bounds__(256) head_fwd(HeadFwd a) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];          // [outs][hidden]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = a.outs * a.hidden;
    for (int i = tid * 4; i < wn; i += 1024) *reinterpret_cast<float4*>(w_lds + i) = *reinterpret_cast<const float4*>(a.w + i);
    __syncthreads();
    for (int64_t row = int64_t(blockIdx.x) * 4 + wave; row < a.rows; row += int64_t(gridDim.x) * 4)
        head_fwd_row<OMAX, false>(a, w_lds, row, lane);
}

// The same rows with W in REGISTERS instead of LDS: a lane owns the columns k = 4 lane + 256 c (c < CH) of every row it meets, so it
// needs exactly those columns of W - OMAX x CH float4, loaded once, next to the first row's loads (one memory latency instead of
// stage + barrier + row).  No LDS traffic per row (the staged form reads 20 KB of W per row and wavefront).  Same chains of fused
// multiply-adds in the same order: the staged kernel's bits.
template <int OMAX, int CH>
__global__ void __launch_bounds__(256) head_fwd_regs(HeadFwd a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t row = int64_t(blockIdx.x) * 4 + wave;
    if (row >= a.rows) return;
    float4 wr[OMAX][CH];
#pragma unroll
    for (int j = 0; j < OMAX; ++j)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = lane * 4 + 256 * c;
            wr[j][c] = (j < a.outs && k < a.hidden) ? *reinterpret_cast<const float4*>(a.w + j * a.hidden + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    for (; row < a.rows; row += int64_t(gridDim.x) * 4) {
        const float* p = a.x + row * a.ldx;
        float4 h[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = lane * 4 + 256 * c;
            h[c] = k < a.hidden ? *reinterpret_cast<const float4*>(p + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float acc[OMAX];
#pragma unroll
        for (int j = 0; j < OMAX; ++j) acc[j] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (lane * 4 + 256 * c < a.hidden) {
                float4 v = h[c];
                if (a.relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); v.z = relu_keep_nan(v.z); v.w = relu_keep_nan(v.w); }
#pragma unroll
                for (int j = 0; j < OMAX; ++j)
                    if (j < a.outs)
                        acc[j] = __builtin_fmaf(v.x, wr[j][c].x, __builtin_fmaf(v.y, wr[j][c].y, __builtin_fmaf(v.z, wr[j][c].z, __builtin_fmaf(v.w, wr[j][c].w, acc[j]))));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < OMAX; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        float s = 0.f;
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
        for (int off = 8; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);
        if (lane == 0) a.row_loss[row] = e2;
        if (a.gpre) {
            float ev[OMAX];
#pragma unroll
            for (int j = 0; j < OMAX; ++j) ev[j] = __shfl(e, j, 64);
            float* q = a.gpre + row * a.hidden;
            float* qd = a.dx + row * a.hidden;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = lane * 4 + 256 * c;
                if (k < a.hidden) {
                    float4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < OMAX; ++j)
                        if (j < a.outs) {
                            d.x = __builtin_fmaf(ev[j], wr[j][c].x, d.x); d.y = __builtin_fmaf(ev[j], wr[j][c].y, d.y);
                            d.z = __builtin_fmaf(ev[j], wr[j][c].z, d.z); d.w = __builtin_fmaf(ev[j], wr[j][c].w, d.w);
                        }
                    *reinterpret_cast<float4*>(qd + k) = d;
                    d.x *= (h[c].x >= 0.0f ? 1.0f : 0.0f); d.y *= (h[c].y >= 0.0f ? 1.0f : 0.0f);
                    d.z *= (h[c].z >= 0.0f ? 1.0f : 0.0f); d.w *= (h[c].w >= 0.0f ? 1.0f : 0.0f);
                    *reinterpret_cast<float4*>(q + k) = d;
                }
            }
        }
    }
}

template <int OMAX>
static bool launch_head_fwd_regs(const HeadFwd& a, dim3 grid, hipStream_t s) {
    const int ch = (a.hidden + 255) / 256;
    if (ch == 1)                    { hipLaunchKernelGGL((head_fwd_regs<OMAX, 1>), grid, dim3(256), 0, s, a); return true; }
    if (ch == 2)                    { hipLaunchKernelGGL((head_fwd_regs<OMAX, 2>), grid, dim3(256), 0, s, a); return true; }
    if constexpr (OMAX * 4 <= 40) { if (ch <= 4) { hipLaunchKernelGGL((head_fwd_regs<OMAX, 4>), grid, dim3(256), 0, s, a); return true; } }
    return false;                   // W does not fit the registers: the staged kernel
}

struct HeadBwd {
    const float* x;        // [rows, hidden], row pitch ldx: the layer input, or the pre-activation when relu != 0
    const float* g;        // [rows, outs] dense: gradient of the layer output
    const float* w;        // [outs, hidden] dense
    float*       dx;       // [rows, hidden] dense or NULL
    float*       gpre;     // [rows, hidden] dense or NULL: dx * (x >= 0), relu.backward's result (relu != 0 only)
    float*       dw;       // [outs, hidden] dense or NULL
    float*       db;       // [outs] or NULL
    const float* row_loss; // [rows] from head_fwd, or NULL
    float*       loss;     // [1]: finished from row_loss by the launch's last workgroup
    float        inv_n;
    int64_t      rows, ldx;
    int          hidden, outs, relu, dw_accumulate, db_accumulate;
    int          n_slabs;      // workgroups 0 .. n_slabs-1 reduce dW / db, then n_tiles write dx / gpre tiles, then the loss
    int          n_tiles;
    int          col_blocks;   // dx tiles per row of tiles
    int          g_dense;      // outs == OMAX and g 16-byte aligned: g is staged with float4 copies
    int          tile_rows;    // rows of a dx / gpre tile: kHeadRows next to slab workgroups, 64 when the launch has tiles only (four
                               // times the workgroups: 64 tiles of 256 rows leave three quarters of the chip idle, 7.4 us at 1024 x 512)
    // the optimizer's update of W / b applied by the slab workgroups to the gradient values they are about to store
    // (lg_adam_epilogue_arm, optim.hip); the new W goes to the plan's second buffer - the dx tiles of this launch read the old one
    const AdamPlan* adam_w;
    const AdamPlan* adam_b;
};

constexpr int kHeadThreads = 1024;                 // 16 wavefronts: a slab workgroup has a CU to itself
constexpr int kHeadRows = 256, kHeadCols = 32;     // dx / gpre tile
constexpr int kSlabCols = 8;                       // columns of dW one reducing workgroup owns (over ALL rows)

// Kinds of workgroups in one launch, none of which waits for another:
//   slab workgroups  own 8 columns of dW for every row: 128 thread rows x 8 columns, g staged in LDS in chunks of up to
//                    1024 rows, the activations of a chunk requested from memory before the chunk's g is staged (both
//                    latencies overlap), sums over the thread rows by shuffles inside a wavefront and through LDS across the
//                    16 wavefronts, always in the same order.  No cross-workgroup reduction: no tickets, no partial slabs.
//                    (Measured on the way here, 1024 x 512 x 10, back-to-back launches, an empty launch = 3.0 us: a version
//                    that reduced 64-row tiles across workgroups inside the launch 14.3 us; slabs with 256 threads and g
//                    staged element by element 13.3 us - 6.2 us of it that staging loop, 2.4 us the multiply-adds; float4
//                    staging + FMA 8.5 us; 1024 threads per workgroup - a slab is latency-bound and has its CU to itself -
//                    this version.)
//   tile workgroups  write 256 x 32 tiles of dx (and gpre): each element needs its row of g and its column of W only.
//   loss workgroup   the last one: finishes the scalar loss of head_fwd from its row sums.
template <int OMAX>
__global__ void __launch_bounds__(kHeadThreads) head_bwd(HeadBwd a) {
    constexpr int CHUNK = OMAX <= 10 ? 1024 : 512;             // rows of g in LDS at a time (40 KiB / 32 KiB)
    constexpr int TR = kHeadThreads / kSlabCols;               // thread rows of a slab workgroup
    constexpr int RPT = CHUNK / TR;                            // rows per thread and chunk
    constexpr int NW = kHeadThreads / 64;
    __shared__ __attribute__((aligned(16))) float g_lds[CHUNK * OMAX];
    __shared__ float red[NW * OMAX * (kSlabCols + 1)];
    const int tid = threadIdx.x;
    if (int(blockIdx.x) < a.n_slabs) {
        const int tc = tid & 7, tr = tid >> 3, wave = tid >> 6;
        const int slab = blockIdx.x;
        const int k = slab * kSlabCols + tc;
        const bool kin = k < a.hidden;
        const bool does_db = slab == 0 && tc == 0;
        float acc[OMAX], dbacc[OMAX];
#pragma unroll
        for (int j = 0; j < OMAX; ++j) { acc[j] = 0.f; dbacc[j] = 0.f; }
        for (int64_t c0 = 0; c0 < a.rows; c0 += CHUNK) {
            float xv[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int64_t r = c0 + tr + TR * i;
                xv[i] = (r < a.rows && kin) ? a.x[r * a.ldx + k] : 0.f;
            }
            __syncthreads();                                   // the previous chunk's g has been consumed
            const int64_t live = (a.rows - c0 < CHUNK ? a.rows - c0 : CHUNK) * OMAX;      // floats of g in this chunk
            if (a.g_dense) {
                const float* src = a.g + c0 * OMAX;            // 16-byte aligned: c0 is a multiple of CHUNK
                for (int i = tid * 4; i < CHUNK * OMAX; i += kHeadThreads * 4) {
                    float4 v = {0.f, 0.f, 0.f, 0.f};
                    if (i + 3 < live) v = *reinterpret_cast<const float4*>(src + i);
                    else if (i < live) { v.x = src[i]; if (i + 1 < live) v.y = src[i + 1]; if (i + 2 < live) v.z = src[i + 2]; }
                    *reinterpret_cast<float4*>(g_lds + i) = v;
                }
            } else {
                for (int i = tid; i < CHUNK * OMAX; i += kHeadThreads) {
                    const int rr = i / OMAX, j = i % OMAX;
                    g_lds[i] = (j < a.outs && c0 + rr < a.rows) ? a.g[(c0 + rr) * a.outs + j] : 0.f;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int rr = tr + TR * i;
                const float h = a.relu ? relu_keep_nan(xv[i]) : xv[i];
#pragma unroll
                for (int j = 0; j < OMAX; ++j) acc[j] = __builtin_fmaf(g_lds[rr * OMAX + j], h, acc[j]);
            }
            if (does_db) {
#pragma unroll
                for (int i = 0; i < RPT; ++i)
#pragma unroll
                    for (int j = 0; j < OMAX; ++j) dbacc[j] += g_lds[(tr + TR * i) * OMAX + j];
            }
        }
        // the 8 thread rows of a wavefront (lanes c, c+8, ..., c+56 share column c), then the 16 wavefronts through LDS
#pragma unroll
        for (int j = 0; j < OMAX; ++j) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                acc[j] += __shfl_xor(acc[j], off, 64);
                dbacc[j] += __shfl_xor(dbacc[j], off, 64);     // only column 0 of slab 0 carries values; the others sum zeros
            }
        }
        if ((tid & 63) < kSlabCols) {
#pragma unroll
            for (int j = 0; j < OMAX; ++j) {
                red[(wave * OMAX + j) * (kSlabCols + 1) + tc] = acc[j];
                if (tc == 0) red[(wave * OMAX + j) * (kSlabCols + 1) + kSlabCols] = dbacc[j];
            }
        }
        __syncthreads();
        static_assert(OMAX * (kSlabCols + 1) <= 192, "the writers below are the first three wavefronts");
        if (tid < 192) {
            // (whole wavefronts enter: the step's scalars are formed by lane 0 and handed to the others)
            AdamScalars cw, cb;
            int64_t done_w = 0, done_b = 0;
            if (a.adam_w) cw = adam_plan_scalars(a.adam_w, done_w);
            if (a.adam_b && slab == 0) cb = adam_plan_scalars(a.adam_b, done_b);
            if (tid < OMAX * (kSlabCols + 1)) {
                const int j = tid / (kSlabCols + 1), c = tid % (kSlabCols + 1);
                if (j < a.outs && (c < kSlabCols || slab == 0)) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += red[(w * OMAX + j) * (kSlabCols + 1) + c];
                    if (c < kSlabCols) {
                        const int kk = slab * kSlabCols + c;
                        if (a.dw && kk < a.hidden) {
                            const int64_t idx = int64_t(j) * a.hidden + kk;
                            const float gv = a.dw_accumulate ? a.dw[idx] + s : s;
                            a.dw[idx] = gv;
                            if (a.adam_w) {
                                float P = a.adam_w->p_in[idx], M = a.adam_w->m[idx], V = a.adam_w->v[idx];
                                adam_elem(P, gv, M, V, cw);
                                a.adam_w->p_out[idx] = P; a.adam_w->m[idx] = M; a.adam_w->v[idx] = V;
                            }
                        }
                    } else if (a.db) {
                        const float gv = a.db_accumulate ? a.db[j] + s : s;
                        a.db[j] = gv;
                        if (a.adam_b) {
                            float P = a.adam_b->p_in[j], M = a.adam_b->m[j], V = a.adam_b->v[j];
                            adam_elem(P, gv, M, V, cb);
                            a.adam_b->p_out[j] = P; a.adam_b->m[j] = M; a.adam_b->v[j] = V;
                        }
                    }
                }
            }
            if (tid == 0 && slab == 0) {              // the one workgroup per step that advances the optimizer's step number
                if (a.adam_w && a.adam_w->step_out) __hip_atomic_store(a.adam_w->step_out, done_w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.adam_b && a.adam_b->step_out) __hip_atomic_store(a.adam_b->step_out, done_b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (int(blockIdx.x) >= a.n_slabs + a.n_tiles) {            // the loss workgroup
        finalize_loss(a.row_loss, a.rows, a.inv_n, a.loss, red);
        return;
    }
    // ---- dx / gpre tile ----
    constexpr int TRT = kHeadThreads / kHeadCols;              // 32 thread rows, 8 rows each
    float* w_lds = red;                                         // [OMAX][kHeadCols] (fits: 16 * 10 * 9 >= 10 * 32)
    static_assert(NW * (kSlabCols + 1) >= kHeadCols && kHeadRows * 16 <= CHUNK * OMAX, "LDS reuse");
    const int t = int(blockIdx.x) - a.n_slabs;
    const int cb = t % a.col_blocks;
    const int tile_rows = a.tile_rows;                          // kHeadRows or less (a multiple of TRT)
    const int64_t r0 = int64_t(t / a.col_blocks) * tile_rows;
    const int tc = tid & 31, tr = tid >> 5;
    const int k = cb * kHeadCols + tc;
    const bool kin = k < a.hidden;
    float xv[kHeadRows / TRT];
#pragma unroll
    for (int i = 0; i < kHeadRows / TRT; ++i) {
        const int64_t r = r0 + tr + TRT * i;
        xv[i] = (a.gpre && TRT * i < tile_rows && r < a.rows && kin) ? a.x[r * a.ldx + k] : 0.f;
    }
    for (int i = tid; i < OMAX * kHeadCols; i += kHeadThreads) {
        const int j = i / kHeadCols, c = i % kHeadCols;
        const int kk = cb * kHeadCols + c;
        w_lds[i] = (j < a.outs && kk < a.hidden) ? a.w[int64_t(j) * a.hidden + kk] : 0.f;
    }
    for (int i = tid; i < tile_rows * OMAX; i += kHeadThreads) {
        const int rr = i / OMAX, j = i % OMAX;
        g_lds[i] = (j < a.outs && r0 + rr < a.rows) ? a.g[(r0 + rr) * a.outs + j] : 0.f;
    }
    __syncthreads();
    float w[OMAX];
#pragma unroll
    for (int j = 0; j < OMAX; ++j) w[j] = w_lds[j * kHeadCols + tc];
#pragma unroll
    for (int i = 0; i < kHeadRows / TRT; ++i) {
        const int rr = tr + TRT * i;
        const int64_t r = r0 + rr;
        if (rr < tile_rows && r < a.rows && kin) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < OMAX; ++j) s = __builtin_fmaf(g_lds[rr * OMAX + j], w[j], s);
            if (a.dx) a.dx[r * a.hidden + k] = s;
            if (a.gpre) a.gpre[r * a.hidden + k] = s * (xv[i] >= 0.0f ? 1.0f : 0.0f);
        }
    }
}

}  // namespace lg

using namespace lg;

extern "C" int lg_head_fwd_grad_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                                    float* y, float* err, float* row_loss, float* dx, float* gpre, int64_t rows, int64_t hidden, int64_t outs);

extern "C" int lg_head_fwd_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                               float* y, float* err, float* row_loss, int64_t rows, int64_t hidden, int64_t outs) {
    return lg_head_fwd_grad_f32(x, ldx, relu, w, bias, target, y, err, row_loss, nullptr, nullptr, rows, hidden, outs);
}

extern "C" int lg_head_fwd_grad_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                                    float* y, float* err, float* row_loss, float* dx, float* gpre, int64_t rows, int64_t hidden, int64_t outs) {
    LG_REQUIRE_INIT();
    LG_ARG((dx == nullptr) == (gpre == nullptr), "lg_head_fwd_grad_f32: dx and gpre go together");
    LG_ARG(gpre == nullptr || (relu && aligned16(gpre) && aligned16(dx)), "lg_head_fwd_grad_f32: gpre is relu.backward's result: it needs relu != 0; dx and gpre 16-byte aligned");
    LG_ARG(rows > 0 && hidden > 0 && outs > 0 && outs <= 16, "lg_head_fwd_f32: need rows > 0, hidden > 0, 1 <= outs <= 16 (got %lld, %lld, %lld)",
           (long long)rows, (long long)hidden, (long long)outs);
    LG_ARG(x && w && target && y && err && row_loss, "lg_head_fwd_f32: NULL pointer");
    LG_ARG(hidden % 4 == 0 && ldx % 4 == 0 && ldx >= hidden && aligned16(x) && aligned16(w),
           "lg_head_fwd_f32: hidden and ldx must be multiples of 4 and x, w 16-byte aligned");
    LG_ARG(outs * hidden * 4 <= 64 * 1024, "lg_head_fwd_f32: W (%lld x %lld) does not fit the 64 KiB LDS stage", (long long)outs, (long long)hidden);
    HeadFwd a{};
    a.x = x; a.w = w; a.bias = bias; a.target = target; a.y = y; a.err = err; a.row_loss = row_loss; a.gpre = gpre; a.dx = dx;
    a.rows = rows; a.ldx = ldx; a.hidden = int(hidden); a.outs = int(outs); a.relu = relu;
    int64_t grid = (rows + 3) / 4;
    if (grid > 1024) grid = 1024;
    const size_t lds = size_t(outs * hidden) * sizeof(float);
    hipStream_t s = rt().stream;
    static const char* form_env = getenv("LG_HEAD_FWD");            // experiments: "lds" = the staged kernel always
    if (!(form_env && form_env[0] == 'l')) {
        const dim3 g(static_cast<unsigned>(grid));
        const bool done = outs <= 4 ? launch_head_fwd_regs<4>(a, g, s) : outs <= 8 ? launch_head_fwd_regs<8>(a, g, s)
                        : outs <= 10 ? launch_head_fwd_regs<10>(a, g, s) : launch_head_fwd_regs<16>(a, g, s);
        if (done) { LG_CHECK_LAUNCH(); return LG_OK; }
    }
    if (outs <= 4)        hipLaunchKernelGGL(head_fwd<4>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else if (outs <= 8)   hipLaunchKernelGGL(head_fwd<8>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else if (outs <= 10)  hipLaunchKernelGGL(head_fwd<10>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else                  hipLaunchKernelGGL(head_fwd<16>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

namespace lg {
// (gemm.hip: the loss a pair bracket carried and found no three-product launch for)
int lg_mse_finalize_job(const float* row_loss, int64_t rows, float inv_n, float* loss) {
    hipLaunchKernelGGL(mse_finalize, dim3(1), dim3(256), 0, rt().stream, row_loss, rows, inv_n, loss);
    LG_CHECK_LAUNCH();
    return LG_OK;
}
}  // namespace lg

extern "C" int lg_mse_finalize_f32(const float* row_loss, int64_t rows, int64_t n, float* loss) {
    LG_REQUIRE_INIT();
    LG_ARG(row_loss && loss && rows > 0 && n > 0, "lg_mse_finalize_f32: bad arguments");
    hipLaunchKernelGGL(mse_finalize, dim3(1), dim3(256), 0, rt().stream, row_loss, rows, float(1.0 / double(n)), loss);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_head_bwd_f32(const float* x, int64_t ldx, int relu, const float* g, const float* w,
                               float* dx, float* gpre, float* dw, int dw_accumulate, float* db, int db_accumulate,
                               int64_t rows, int64_t hidden, int64_t outs, const float* row_loss, float* loss) {
    LG_REQUIRE_INIT();
    LG_ARG(rows > 0 && hidden > 0 && outs > 0 && outs <= 16, "lg_head_bwd_f32: need rows > 0, hidden > 0, 1 <= outs <= 16 (got %lld, %lld, %lld)",
           (long long)rows, (long long)hidden, (long long)outs);
    LG_ARG(x && g && w, "lg_head_bwd_f32: NULL pointer");
    LG_ARG(ldx >= hidden, "lg_head_bwd_f32: ldx < hidden");
    LG_ARG(gpre == nullptr || relu, "lg_head_bwd_f32: gpre is relu.backward's result and needs relu != 0");
    LG_ARG(hidden < (int64_t(1) << 24), "lg_head_bwd_f32: hidden too large");
    LG_ARG((row_loss == nullptr) == (loss == nullptr), "lg_head_bwd_f32: row_loss and loss go together");
    HeadBwd a{};
    a.x = x; a.g = g; a.w = w; a.dx = dx; a.gpre = gpre; a.dw = dw; a.db = db;
    a.rows = rows; a.ldx = ldx; a.hidden = int(hidden); a.outs = int(outs); a.relu = relu;
    a.dw_accumulate = dw_accumulate; a.db_accumulate = db_accumulate;
    a.row_loss = row_loss; a.loss = loss;
    a.inv_n = float(1.0 / double(rows * outs));       // python's `1 / numel` rounded once to fp32 (as lg_mse_f32)
    a.n_slabs = dw ? int((hidden + kSlabCols - 1) / kSlabCols) : (db ? 1 : 0);       // the bias gradient alone: slab 0 does it
    a.col_blocks = int((hidden + kHeadCols - 1) / kHeadCols);
    a.tile_rows = a.n_slabs > 0 ? kHeadRows : 64;
    const int64_t tiles = (dx || gpre) ? int64_t(a.col_blocks) * ((rows + a.tile_rows - 1) / a.tile_rows) : 0;
    const int64_t grid = a.n_slabs + tiles + (loss ? 1 : 0);
    if (grid == 0) return LG_OK;
    LG_ARG(grid < (int64_t(1) << 30), "lg_head_bwd_f32: problem too large for one launch");
    a.n_tiles = int(tiles);
    const int omax = outs <= 4 ? 4 : (outs <= 8 ? 8 : (outs <= 10 ? 10 : 16));
    a.g_dense = (omax == outs && aligned16(g)) ? 1 : 0;
    {
        int arc = LG_OK;
        if (dw) { a.adam_w = adam_epilogue_take(dw, outs * hidden, dw_accumulate, &arc); if (arc != LG_OK) return arc; }
        if (db) { a.adam_b = adam_epilogue_take(db, outs, db_accumulate, &arc); if (arc != LG_OK) return arc; }
    }
    hipStream_t s = rt().stream;
    if (omax == 4)        hipLaunchKernelGGL(head_bwd<4>, dim3(unsigned(grid)), dim3(kHeadThreads), 0, s, a);
    else if (omax == 8)   hipLaunchKernelGGL(head_bwd<8>, dim3(unsigned(grid)), dim3(kHeadThreads), 0, s, a);
    else if (omax == 10)  hipLaunchKernelGGL(head_bwd<10>, dim3(unsigned(grid)), dim3(kHeadThreads), 0, s, a);
    else                  hipLaunchKernelGGL(head_bwd<16>, dim3(unsigned(grid)), dim3(kHeadThreads), 0, s, a);
    LG_CHECK_LAUNCH();
    return LG_OK;
}
