// The skinny output layer of a classifier and its loss as two launches instead of five
// (VERDICT r1 item 3).  nn.Linear with <= 16 output features ("the N = 10 head" of the
// MNIST MLP: y = relu(pre) @ W^T + b, 1024 x 10 x 512) has no use for matrix cores: it moves 2 MB and
// does 10 MFLOP, so as MFMA tiles it was pure launch latency (forward 9.8, loss 4.9, dW+db 11.5, dh 5.9,
// relu-backward 5.2 us).  Here:
//
//   head_fwd   y = act(x) @ W^T + b,  err = y + (-target),  loss = (sum err^2 * (1/N)) * 0.5
//              one wavefront per row (lanes split K, W staged once per workgroup in LDS, butterfly sum),
//              the loss folded inside the launch (per-workgroup partial, ticket, last arriver sums in index order)
//              - nn.py:96 + loss.py:4-10 of the reference.  Optionally advances a device step counter (the
//              optimizer's, see optim.hip) - a kernel that runs once per step carries that increment for free.
//   head_bwd   dx = g @ W,  g_pre = dx * (pre >= 0),  dW = g^T @ act(x),  db = column sums of g
//              64 x 32 tiles of (rows x hidden): every element of dx needs only its row of g and its column of W;
//              dW / db are reduced over the tile's rows in registers + LDS, then over the row blocks inside the
//              launch (write-through partial slabs, one ticket per column block, fixed summation order)
//              - what linear.backward + relu.backward of the tape compute (cpu/ops.py:114-116, :229, func.py:50-56).
//
// act = relu when the tape's relu is still lazy (autograd/hip/ops.py), identity otherwise.  Products and sums are
// plain fp32 operations in a fixed order (no contraction, -ffp-contract=off): results agree with the GEMM form to
// rounding (<= 1e-6 relative), and are bit-reproducible from run to run.
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
    float*       loss;     // [1]
    float*       partial;  // [gridDim.x]
    int*         ticket;   // zero on entry and on exit
    int64_t*     bump;     // optional: bump[0] += 1 (once per launch)
    int64_t      rows, ldx;
    int          hidden, outs, relu;
    float        inv_n;
};

template <int OMAX>
__global__ void __launch_bounds__(256) head_fwd(HeadFwd a) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];          // [outs][hidden]
    __shared__ float wave_part[4];
    __shared__ int arrived_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = a.outs * a.hidden;
    for (int i = tid * 4; i < wn; i += 1024) *reinterpret_cast<float4*>(w_lds + i) = *reinterpret_cast<const float4*>(a.w + i);
    __syncthreads();
    float wave_acc = 0.f;                                                   // sum of err^2 over this wave's rows
    for (int64_t row = int64_t(blockIdx.x) * 4 + wave; row < a.rows; row += int64_t(gridDim.x) * 4) {
        float acc[OMAX];
#pragma unroll
        for (int j = 0; j < OMAX; ++j) acc[j] = 0.f;
        const float* p = a.x + row * a.ldx;
        for (int k = lane * 4; k < a.hidden; k += 256) {
            float4 h = *reinterpret_cast<const float4*>(p + k);
            if (a.relu) { h.x = relu_keep_nan(h.x); h.y = relu_keep_nan(h.y); h.z = relu_keep_nan(h.z); h.w = relu_keep_nan(h.w); }
#pragma unroll
            for (int j = 0; j < OMAX; ++j) {
                if (j < a.outs) {
                    const float4 w = *reinterpret_cast<const float4*>(w_lds + j * a.hidden + k);
                    acc[j] += (h.x * w.x + h.y * w.y) + (h.z * w.z + h.w * w.w);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < OMAX; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        float s = 0.f;                                                      // lane j < outs keeps output j
#pragma unroll
        for (int j = 0; j < OMAX; ++j) s = (lane == j) ? acc[j] : s;
        float e2 = 0.f;
        if (lane < a.outs) {
            const float yv = a.bias ? s + a.bias[lane] : s;
            const float e = yv + (-a.target[row * a.outs + lane]);
            a.y[row * a.outs + lane] = yv;
            a.err[row * a.outs + lane] = e;
            e2 = e * e;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);          // OMAX <= 16: lanes 0..15
        wave_acc += e2;
    }
    if (lane == 0) wave_part[wave] = wave_acc;
    __syncthreads();
    if (tid == 0) {
        const float v = (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]);
        __hip_atomic_store(a.partial + blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int order = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = order == int(gridDim.x) - 1;
        if (last) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived_last = last;
    }
    __syncthreads();
    if (!arrived_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the last workgroup to arrive: sum the partials, thread t taking t, t+256, ... then a fixed tree
    float v = 0.f;
    for (int i = tid; i < int(gridDim.x); i += 256) v += __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) wave_part[wave] = v;
    __syncthreads();
    if (tid == 0) {
        const float total = (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]);
        a.loss[0] = (total * a.inv_n) * 0.5f;
        if (a.bump) a.bump[0] += 1;
    }
}

struct HeadBwd {
    const float* x;        // [rows, hidden], row pitch ldx: the layer input, or the pre-activation when relu != 0
    const float* g;        // [rows, outs] dense: gradient of the layer output
    const float* w;        // [outs, hidden] dense
    float*       dx;       // [rows, hidden] dense or NULL
    float*       gpre;     // [rows, hidden] dense or NULL: dx * (x >= 0), relu.backward's result (relu != 0 only)
    float*       dw;       // [outs, hidden] dense or NULL
    float*       db;       // [outs] or NULL
    float*       slabs;    // [col_blocks][row_blocks][OMAX][33]
    int*         tickets;  // one per column block, zero on entry and on exit
    int64_t      rows, ldx;
    int          hidden, outs, relu, dw_accumulate, db_accumulate, row_blocks;
};

constexpr int kHeadRows = 64, kHeadCols = 32;

template <int OMAX>
__global__ void __launch_bounds__(256) head_bwd(HeadBwd a) {
    __shared__ float g_lds[kHeadRows][OMAX];
    __shared__ float w_lds[OMAX][kHeadCols];
    __shared__ float red[8][OMAX][kHeadCols + 1];
    __shared__ int arrived_last;
    const int tid = threadIdx.x, tc = tid & 31, tr = tid >> 5;
    const int cb = blockIdx.x, rb = blockIdx.y;
    const int k = cb * kHeadCols + tc;
    const bool kin = k < a.hidden;
    for (int i = tid; i < OMAX * kHeadCols; i += 256) {
        const int j = i / kHeadCols, c = i % kHeadCols;
        const int kk = cb * kHeadCols + c;
        w_lds[j][c] = (j < a.outs && kk < a.hidden) ? a.w[int64_t(j) * a.hidden + kk] : 0.f;
    }
    float acc[OMAX], w[OMAX], dbacc = 0.f;
#pragma unroll
    for (int j = 0; j < OMAX; ++j) acc[j] = 0.f;
    const int64_t tiles = (a.rows + kHeadRows - 1) / kHeadRows;
    for (int64_t tile = rb; tile < tiles; tile += a.row_blocks) {
        const int64_t r0 = tile * kHeadRows;
        __syncthreads();                                                   // w_lds staged / previous tile's g_lds consumed
        for (int i = tid; i < kHeadRows * OMAX; i += 256) {
            const int rr = i / OMAX, j = i % OMAX;
            g_lds[rr][j] = (j < a.outs && r0 + rr < a.rows) ? a.g[(r0 + rr) * a.outs + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OMAX; ++j) w[j] = w_lds[j][tc];
        if (cb == 0 && tid < OMAX) {                                       // bias gradient: column sums of g
            float s = 0.f;
            for (int rr = 0; rr < kHeadRows; ++rr) s += g_lds[rr][tid];
            dbacc += s;
        }
#pragma unroll
        for (int i = 0; i < kHeadRows / 8; ++i) {
            const int rr = tr + 8 * i;
            const int64_t r = r0 + rr;
            if (r < a.rows && kin) {
                const float xv = a.x[r * a.ldx + k];
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < OMAX; ++j) s += g_lds[rr][j] * w[j];
                if (a.dx) a.dx[r * a.hidden + k] = s;
                if (a.gpre) a.gpre[r * a.hidden + k] = s * (xv >= 0.0f ? 1.0f : 0.0f);
                const float h = a.relu ? relu_keep_nan(xv) : xv;
#pragma unroll
                for (int j = 0; j < OMAX; ++j) acc[j] += g_lds[rr][j] * h;
            }
        }
    }
    if (a.dw == nullptr && a.db == nullptr) return;
    // rows of this workgroup: 8 thread rows -> one value per (output j, column)
#pragma unroll
    for (int j = 0; j < OMAX; ++j) red[tr][j][tc] = acc[j];
    if (cb == 0 && tid < OMAX) red[0][tid][kHeadCols] = dbacc;
    __syncthreads();
    constexpr int SLAB = OMAX * (kHeadCols + 1);
    float* mine = a.slabs + (int64_t(cb) * a.row_blocks + rb) * SLAB;
    for (int i = tid; i < SLAB; i += 256) {
        const int j = i / (kHeadCols + 1), c = i % (kHeadCols + 1);
        float s;
        if (c < kHeadCols) {
            s = red[0][j][c];
#pragma unroll
            for (int t = 1; t < 8; ++t) s += red[t][j][c];
        } else {
            s = cb == 0 ? red[0][j][kHeadCols] : 0.f;
        }
        if (a.row_blocks == 1) red[0][j][c] = s;                          // nothing to fold: hand over through LDS
        else __hip_atomic_store(mine + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.row_blocks > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* ticket = a.tickets + cb;
            const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = order == a.row_blocks - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived_last = last;
        }
        __syncthreads();
        if (!arrived_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
    // fold the row blocks in index order and write this column block of dW (and db from column block 0)
    const float* base = a.slabs + int64_t(cb) * a.row_blocks * SLAB;
    for (int i = tid; i < SLAB; i += 256) {
        const int j = i / (kHeadCols + 1), c = i % (kHeadCols + 1);
        if (j >= a.outs) continue;
        float s;
        if (a.row_blocks == 1) {
            s = red[0][j][c];
        } else {
            s = 0.f;
            int b = 0;
            for (; b + 3 < a.row_blocks; b += 4) {                        // four loads in flight, summed in order
                const float v0 = __hip_atomic_load(base + int64_t(b) * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float v1 = __hip_atomic_load(base + int64_t(b + 1) * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float v2 = __hip_atomic_load(base + int64_t(b + 2) * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float v3 = __hip_atomic_load(base + int64_t(b + 3) * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s = (((s + v0) + v1) + v2) + v3;
            }
            for (; b < a.row_blocks; ++b) s += __hip_atomic_load(base + int64_t(b) * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (c < kHeadCols) {
            const int kk = cb * kHeadCols + c;
            if (a.dw && kk < a.hidden) {
                float* d = a.dw + int64_t(j) * a.hidden + kk;
                *d = a.dw_accumulate ? *d + s : s;
            }
        } else if (cb == 0 && a.db) {
            a.db[j] = a.db_accumulate ? a.db[j] + s : s;
        }
    }
}

}  // namespace lg

using namespace lg;

extern "C" int lg_head_fwd_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                               float* y, float* err, float* loss, int64_t rows, int64_t hidden, int64_t outs,
                               int64_t* step_counter) {
    LG_REQUIRE_INIT();
    LG_ARG(rows > 0 && hidden > 0 && outs > 0 && outs <= 16, "lg_head_fwd_f32: need rows > 0, hidden > 0, 1 <= outs <= 16 (got %lld, %lld, %lld)",
           (long long)rows, (long long)hidden, (long long)outs);
    LG_ARG(x && w && target && y && err && loss, "lg_head_fwd_f32: NULL pointer");
    LG_ARG(hidden % 4 == 0 && ldx % 4 == 0 && ldx >= hidden && aligned16(x) && aligned16(w),
           "lg_head_fwd_f32: hidden and ldx must be multiples of 4 and x, w 16-byte aligned");
    LG_ARG(outs * hidden * 4 <= 64 * 1024, "lg_head_fwd_f32: W (%lld x %lld) does not fit the 64 KiB LDS stage", (long long)outs, (long long)hidden);
    HeadFwd a{};
    a.x = x; a.w = w; a.bias = bias; a.target = target; a.y = y; a.err = err; a.loss = loss;
    a.rows = rows; a.ldx = ldx; a.hidden = int(hidden); a.outs = int(outs); a.relu = relu;
    a.inv_n = float(1.0 / double(rows * outs));       // python's `1 / numel` rounded once to fp32 (as lg_mse_f32)
    a.bump = step_counter;
    int64_t grid = (rows + 3) / 4;
    if (grid > 1024) grid = 1024;
    int rc = lg_malloc(reinterpret_cast<void**>(&a.partial), size_t(grid) * sizeof(float));
    if (rc != LG_OK) return rc;
    a.ticket = rt().gemm_tickets;
    const size_t lds = size_t(outs * hidden) * sizeof(float);
    hipStream_t s = rt().stream;
    if (outs <= 4)        hipLaunchKernelGGL(head_fwd<4>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else if (outs <= 8)   hipLaunchKernelGGL(head_fwd<8>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else if (outs <= 10)  hipLaunchKernelGGL(head_fwd<10>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    else                  hipLaunchKernelGGL(head_fwd<16>, dim3(unsigned(grid)), dim3(256), lds, s, a);
    rc = lg_free(a.partial);
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_head_bwd_f32(const float* x, int64_t ldx, int relu, const float* g, const float* w,
                               float* dx, float* gpre, float* dw, int dw_accumulate, float* db, int db_accumulate,
                               int64_t rows, int64_t hidden, int64_t outs) {
    LG_REQUIRE_INIT();
    LG_ARG(rows > 0 && hidden > 0 && outs > 0 && outs <= 16, "lg_head_bwd_f32: need rows > 0, hidden > 0, 1 <= outs <= 16 (got %lld, %lld, %lld)",
           (long long)rows, (long long)hidden, (long long)outs);
    LG_ARG(x && g && w, "lg_head_bwd_f32: NULL pointer");
    LG_ARG(ldx >= hidden, "lg_head_bwd_f32: ldx < hidden");
    LG_ARG(gpre == nullptr || relu, "lg_head_bwd_f32: gpre is relu.backward's result and needs relu != 0");
    HeadBwd a{};
    a.x = x; a.g = g; a.w = w; a.dx = dx; a.gpre = gpre; a.dw = dw; a.db = db;
    a.rows = rows; a.ldx = ldx; a.hidden = int(hidden); a.outs = int(outs); a.relu = relu;
    a.dw_accumulate = dw_accumulate; a.db_accumulate = db_accumulate;
    const int64_t col_blocks = (hidden + kHeadCols - 1) / kHeadCols;
    const int64_t tiles = (rows + kHeadRows - 1) / kHeadRows;
    LG_ARG(col_blocks <= rt().n_gemm_tickets, "lg_head_bwd_f32: hidden too large");
    // enough workgroups for 256 CUs, but never more row blocks than the last arriver can fold cheaply
    int64_t row_blocks = (2 * 256 + col_blocks - 1) / col_blocks;
    if (row_blocks > tiles) row_blocks = tiles;
    if (row_blocks > 64) row_blocks = 64;
    if (row_blocks < 1) row_blocks = 1;
    a.row_blocks = int(row_blocks);
    const int omax = outs <= 4 ? 4 : (outs <= 8 ? 8 : (outs <= 10 ? 10 : 16));
    int rc = lg_malloc(reinterpret_cast<void**>(&a.slabs), size_t(col_blocks * row_blocks) * omax * (kHeadCols + 1) * sizeof(float));
    if (rc != LG_OK) return rc;
    a.tickets = rt().gemm_tickets;
    hipStream_t s = rt().stream;
    const dim3 grid{unsigned(col_blocks), unsigned(row_blocks), 1u};
    if (omax == 4)        hipLaunchKernelGGL(head_bwd<4>, grid, dim3(256), 0, s, a);
    else if (omax == 8)   hipLaunchKernelGGL(head_bwd<8>, grid, dim3(256), 0, s, a);
    else if (omax == 10)  hipLaunchKernelGGL(head_bwd<10>, grid, dim3(256), 0, s, a);
    else                  hipLaunchKernelGGL(head_bwd<16>, grid, dim3(256), 0, s, a);
    rc = lg_free(a.slabs);
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
