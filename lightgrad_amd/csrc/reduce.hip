// fp32 reductions (sum / max / min over an arbitrary set of axes) for gfx950.
// HBM-bound: algorithmic traffic is 4 B per input element; the kernels keep loads
// coalesced (16 B per lane where the layout allows) and combine with wave64 shuffles.
//
// Two kernels cover every stride pattern:
//   red_rows : the reduced axes form ONE contiguous run (trailing-axis sums of softmax /
//              LayerNorm, full reductions).  One wave per output when there are many short
//              rows; few long rows are split over many workgroups into partials that a
//              second launch of the same kernel combines.
//   red_cols : everything else - one thread per OUTPUT element walking the reduced index
//              space (leading-axis sums such as the bias un-broadcast `sum((1024,512)->(1,512))`
//              of func.py:50-56: lanes run along the contiguous kept axis, so loads coalesce);
//              long reductions are split over blockIdx.y into partials that the last workgroup to
//              arrive (per-column-block ticket) folds inside the same launch.
// Semantics: opencl/ops.py:344-400 with cpu/ops.py:260-293 as numeric ground truth
// (max/min exact incl. NaN propagation like np.max; sum is a tree sum in fp32).
// The multi-pass LDS tree of opencl/kernels.py:344-501 is not reproduced.
#include "common.h"
#include <cmath>
#include <cstdlib>

namespace lg {

struct RedDesc {
    int     nk, nr;                       // kept / reduced dims after collapsing (each >= 1)
    int     accumulate;                   // final pass only: out += result instead of out = result
    int64_t n_out, rlen;
    int64_t kshape[LG_MAX_DIMS], kstride[LG_MAX_DIMS];
    int64_t rshape[LG_MAX_DIMS], rstride[LG_MAX_DIMS];
};

template <int OP> struct Red;
template <> struct Red<LG_RED_SUM> {
    __device__ static float identity() { return 0.0f; }
    __device__ static float comb(float a, float b) { return a + b; }
};
template <> struct Red<LG_RED_MAX> {
    __device__ static float identity() { return -INFINITY; }
    __device__ static float comb(float a, float b) { return (a > b || a != a) ? a : b; }
};
template <> struct Red<LG_RED_MIN> {
    __device__ static float identity() { return INFINITY; }
    __device__ static float comb(float a, float b) { return (a < b || a != a) ? a : b; }
};

template <int OP>
__device__ __forceinline__ float wave_reduce(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = Red<OP>::comb(v, __shfl_down(v, off, 64));
    return v;   // valid in lane 0
}

__device__ __forceinline__ int64_t kept_offset(const RedDesc& d, int64_t out_idx) {
    int64_t off = 0;
    for (int k = d.nk - 1; k >= 0; --k) {
        int64_t sz = d.kshape[k];
        off += (out_idx % sz) * d.kstride[k];
        out_idx /= sz;
    }
    return off;
}

// ---- rows: reduced run is contiguous (rstride[0] == 1, nr == 1) ---------------------------
// grid.x = ceil(n_out / ROWS_PER_BLOCK) when splits == 1 (one wave per row, 4 rows per block),
// else grid = (splits, n_out) with the whole block on one row segment.
template <int OP>
__global__ void __launch_bounds__(256) red_rows_wave(const float* __restrict__ in, float* __restrict__ out, RedDesc d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = int64_t(blockIdx.x) * 4 + wave;
    if (row >= d.n_out) return;
    const float* p = in + kept_offset(d, row);
    float acc = Red<OP>::identity();
    const int64_t n = d.rlen;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const int64_t nv = n / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int64_t i = lane; i < nv; i += 64) {
            float4 t = p4[i];
            acc = Red<OP>::comb(acc, Red<OP>::comb(Red<OP>::comb(t.x, t.y), Red<OP>::comb(t.z, t.w)));
        }
        for (int64_t i = nv * 4 + lane; i < n; i += 64) acc = Red<OP>::comb(acc, p[i]);
    } else {
        for (int64_t i = lane; i < n; i += 64) acc = Red<OP>::comb(acc, p[i]);
    }
    acc = wave_reduce<OP>(acc);
    if (lane == 0) out[row] = d.accumulate ? out[row] + acc : acc;
}

constexpr int kRowsFoldGroup = 32;     // segments per first-level ticket (<= 64: folded by one wave)
template <int OP>
__global__ void __launch_bounds__(256) red_rows_split(const float* __restrict__ in, float* out, float* partial, int* tickets,
                                                      RedDesc d, int64_t seg, int64_t tstride) {
    __shared__ float wsum[4];
    __shared__ int arrived_last, arrived_last2;      // (one word per level: the first is still being read by slower wavefronts when the first wavefront moves on)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.y, split = blockIdx.x, splits = gridDim.x;
    const int64_t begin = split * seg;
    int64_t end = begin + seg;
    if (end > d.rlen) end = d.rlen;
    const float* p = in + kept_offset(d, row);
    float acc0 = Red<OP>::identity(), acc1 = Red<OP>::identity();
    // seg is a multiple of 4, so alignment of p decides alignment of the segment
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const int64_t v0 = begin / 4, v1 = end / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        int64_t i = v0 + threadIdx.x;
        for (; i + 256 < v1; i += 512) {
            float4 s = p4[i], t = p4[i + 256];
            acc0 = Red<OP>::comb(acc0, Red<OP>::comb(Red<OP>::comb(s.x, s.y), Red<OP>::comb(s.z, s.w)));
            acc1 = Red<OP>::comb(acc1, Red<OP>::comb(Red<OP>::comb(t.x, t.y), Red<OP>::comb(t.z, t.w)));
        }
        for (; i < v1; i += 256) {
            float4 s = p4[i];
            acc0 = Red<OP>::comb(acc0, Red<OP>::comb(Red<OP>::comb(s.x, s.y), Red<OP>::comb(s.z, s.w)));
        }
        for (int64_t e = v1 * 4 + threadIdx.x; e < end; e += 256) acc1 = Red<OP>::comb(acc1, p[e]);
    } else {
        for (int64_t e = begin + threadIdx.x; e < end; e += 256) acc0 = Red<OP>::comb(acc0, p[e]);
    }
    float acc = wave_reduce<OP>(Red<OP>::comb(acc0, acc1));
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (splits == 1) {
        if (threadIdx.x == 0) {
            const float v = Red<OP>::comb(Red<OP>::comb(wsum[0], wsum[1]), Red<OP>::comb(wsum[2], wsum[3]));
            out[row] = d.accumulate ? out[row] + v : v;
        }
        return;
    }
    // several workgroups per row: the second pass of a two-launch reduction, inside this launch.  A workgroup publishes its
    // partial (write-through, drained) and takes a ticket.  ONE ticket for all 768 workgroups of a full sum costs more than the
    // launch it saves - atomics on one address are served one after the other, 13 ns each (64 MiB: 14.2 -> 23.9 us,
    // profiles/r4/hbm_sum_single_ticket.txt) - so the fold has two levels: groups of kRowsFoldGroup consecutive segments with a
    // ticket each (on cache lines of their own); a group's last arriver folds the group and takes the row's ticket; the last
    // of those folds the groups.  Fixed trees at both levels: bit-reproducible.
    const int64_t groups = (splits + kRowsFoldGroup - 1) / kRowsFoldGroup, grp = split / kRowsFoldGroup;
    const int64_t in_group = (grp == groups - 1) ? splits - grp * kRowsFoldGroup : kRowsFoldGroup;
    float* const partial2 = partial + int64_t(gridDim.y) * splits;              // [row][group]
    int* const t1 = tickets + (row * (groups + 1) + grp) * tstride;
    int* const t2 = tickets + (row * (groups + 1) + groups) * tstride;
    if (threadIdx.x == 0) {
        const float v = Red<OP>::comb(Red<OP>::comb(wsum[0], wsum[1]), Red<OP>::comb(wsum[2], wsum[3]));
        __hip_atomic_store(partial + row * splits + split, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int last = __hip_atomic_fetch_add(t1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == int(in_group) - 1;
        if (last) __hip_atomic_store(t1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived_last = last;
    }
    __syncthreads();
    if (!arrived_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wave == 0) {                                          // in_group <= 64 partials: one per lane of the first wave
        const float x = lane < in_group ? __hip_atomic_load(partial + row * splits + grp * kRowsFoldGroup + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : Red<OP>::identity();
        const float v = wave_reduce<OP>(x);
        if (lane == 0) {
            __hip_atomic_store(partial2 + row * groups + grp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int last = __hip_atomic_fetch_add(t2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == int(groups) - 1;
            if (last) __hip_atomic_store(t2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived_last2 = last;
        }
    }
    __syncthreads();
    if (!arrived_last2 || wave != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float f = Red<OP>::identity();
    for (int64_t r = lane; r < groups; r += 64)
        f = Red<OP>::comb(f, __hip_atomic_load(partial2 + row * groups + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    f = wave_reduce<OP>(f);
    if (lane == 0) out[row] = d.accumulate ? out[row] + f : f;
}

// ---- cols / general: one thread per output element --------------------------------------------
// gridDim.y > 1: the reduced range is split over blockIdx.y.  Each split publishes its partial (write-through store,
// drained), takes a ticket for its block of 256 outputs, and the workgroup that arrives last folds the partials in
// split order - the in-launch form of a second pass (cdna_hip_programming.md, in-launch split-K recipe).
template <int OP>
__global__ void __launch_bounds__(256) red_cols(const float* __restrict__ in, float* out, float* partial, int* tickets,
                                                RedDesc d, int64_t chunk) {
    const int64_t o_raw = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const bool live = o_raw < d.n_out;
    const int64_t o = live ? o_raw : d.n_out - 1;            // idle lanes of the last block recompute a valid output
    const int64_t split = blockIdx.y;
    const int64_t begin = split * chunk;
    int64_t end = begin + chunk;
    if (end > d.rlen) end = d.rlen;
    const float* p = in + kept_offset(d, o);
    float a0 = Red<OP>::identity(), a1 = Red<OP>::identity(), a2 = Red<OP>::identity(), a3 = Red<OP>::identity();
    if (d.nr == 1) {
        const int64_t rs = d.rstride[0];
        int64_t r = begin;
        for (; r + 3 < end; r += 4) {
            float x0 = p[r * rs], x1 = p[(r + 1) * rs], x2 = p[(r + 2) * rs], x3 = p[(r + 3) * rs];
            a0 = Red<OP>::comb(a0, x0); a1 = Red<OP>::comb(a1, x1); a2 = Red<OP>::comb(a2, x2); a3 = Red<OP>::comb(a3, x3);
        }
        for (; r < end; ++r) a0 = Red<OP>::comb(a0, p[r * rs]);
    } else {
        // odometer over the reduced dims, innermost fastest
        int64_t idx[LG_MAX_DIMS];
        int64_t rem = begin, off = 0;
        for (int k = d.nr - 1; k >= 0; --k) {
            idx[k] = rem % d.rshape[k];
            rem /= d.rshape[k];
            off += idx[k] * d.rstride[k];
        }
        for (int64_t r = begin; r < end; ++r) {
            a0 = Red<OP>::comb(a0, p[off]);
            int k = d.nr - 1;
            idx[k] += 1;
            off += d.rstride[k];
            while (k > 0 && idx[k] == d.rshape[k]) {
                off -= d.rshape[k] * d.rstride[k];
                idx[k] = 0;
                --k;
                idx[k] += 1;
                off += d.rstride[k];
            }
        }
    }
    float v = Red<OP>::comb(Red<OP>::comb(a0, a1), Red<OP>::comb(a2, a3));
    const int splits = gridDim.y;
    if (splits > 1) {
        __hip_atomic_store(partial + split * d.n_out + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int arrived_last;
        if (threadIdx.x == 0) {
            int* ticket = tickets + blockIdx.x;
            const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = order == splits - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived_last = last;
        }
        __syncthreads();
        if (!arrived_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 8 partials in flight per thread (their loads cross XCDs: ~1 us each), combined in a fixed order
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = Red<OP>::identity();
        const float* q = partial + o;
        int r = 0;
        for (; r + 7 < splits; r += 8) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = __hip_atomic_load(q + int64_t(r + e) * d.n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = Red<OP>::comb(f[e], x[e]);
        }
        for (; r < splits; ++r)
            f[0] = Red<OP>::comb(f[0], __hip_atomic_load(q + int64_t(r) * d.n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        v = Red<OP>::comb(Red<OP>::comb(Red<OP>::comb(f[0], f[1]), Red<OP>::comb(f[2], f[3])),
                          Red<OP>::comb(Red<OP>::comb(f[4], f[5]), Red<OP>::comb(f[6], f[7])));
    }
    if (live) out[o] = d.accumulate ? out[o] + v : v;
}

// ---- columns of a row-major matrix: 64-column x many-row tiles -------------------------------------
// The common leading-axis reduction (kept axis contiguous, one reduced axis: `sum((N, C) -> (C,))`, the bias un-broadcast
// of func.py:50-56).  red_cols gives every thread ONE column and at most four loads in flight: at 64 MiB that was 2.8 TB/s
// (35 % of the 8 TB/s spec), bound by memory latency and by a fold over 32 partials.  Here a workgroup owns 64 columns as
// 16 float4 lanes x 16 thread rows, each thread keeps four 16-byte loads in flight, the thread rows combine through LDS, and
// only a few row chunks (<= 16) are left to fold across workgroups - all their partials are fetched at once.
template <int OP>
__global__ void __launch_bounds__(256) red_cols_tile(const float* __restrict__ in, float* out, float* partial, int* tickets,
                                                     int64_t n_out, int64_t rlen, int64_t rstride, int64_t chunk, int accumulate) {
    __shared__ float lds[16][64 + 4];
    __shared__ int arrived_last;
    const int tid = threadIdx.x, tc = tid & 15, tr = tid >> 4;
    const int64_t col = int64_t(blockIdx.x) * 64 + tc * 4;
    const bool live = col < n_out;                                   // n_out is a multiple of 4
    const int64_t begin = int64_t(blockIdx.y) * chunk;
    int64_t end = begin + chunk;
    if (end > rlen) end = rlen;
    const float id = Red<OP>::identity();
    constexpr int U = 8;                                             // 16-byte loads in flight per thread
    float4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = make_float4(id, id, id, id);
    if (live) {
        const float* p = in + col;
        int64_t r = begin + tr;
        for (; r + 16 * (U - 1) < end; r += 16 * U) {
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const float4*>(p + (r + 16 * u) * rstride);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a[u].x = Red<OP>::comb(a[u].x, x[u].x); a[u].y = Red<OP>::comb(a[u].y, x[u].y);
                a[u].z = Red<OP>::comb(a[u].z, x[u].z); a[u].w = Red<OP>::comb(a[u].w, x[u].w);
            }
        }
        for (; r < end; r += 16) {
            const float4 x = *reinterpret_cast<const float4*>(p + r * rstride);
            a[0].x = Red<OP>::comb(a[0].x, x.x); a[0].y = Red<OP>::comb(a[0].y, x.y);
            a[0].z = Red<OP>::comb(a[0].z, x.z); a[0].w = Red<OP>::comb(a[0].w, x.w);
        }
    }
#pragma unroll
    for (int w = U / 2; w >= 1; w /= 2)
#pragma unroll
        for (int u = 0; u < w; ++u) {
            a[u].x = Red<OP>::comb(a[u].x, a[u + w].x); a[u].y = Red<OP>::comb(a[u].y, a[u + w].y);
            a[u].z = Red<OP>::comb(a[u].z, a[u + w].z); a[u].w = Red<OP>::comb(a[u].w, a[u + w].w);
        }
    const float4 t = a[0];
    lds[tr][tc * 4 + 0] = t.x; lds[tr][tc * 4 + 1] = t.y; lds[tr][tc * 4 + 2] = t.z; lds[tr][tc * 4 + 3] = t.w;
    __syncthreads();
    const int64_t o = int64_t(blockIdx.x) * 64 + tid;                // threads 0..63: one output column each
    float v = id;
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v = Red<OP>::comb(v, lds[k][tid]);
    }
    const int splits = gridDim.y;
    if (splits > 1) {
        if (tid < 64 && o < n_out) __hip_atomic_store(partial + int64_t(blockIdx.y) * n_out + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* ticket = tickets + blockIdx.x;
            const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = order == splits - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived_last = last;
        }
        __syncthreads();
        if (!arrived_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < 64 && o < n_out) {
            float x[16];                                             // splits <= 16: every partial of this column in flight at once
#pragma unroll
            for (int k = 0; k < 16; ++k)
                x[k] = k < splits ? __hip_atomic_load(partial + int64_t(k) * n_out + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : id;
            v = id;
#pragma unroll
            for (int k = 0; k < 16; ++k) v = Red<OP>::comb(v, x[k]);
        }
    }
    if (tid < 64 && o < n_out) out[o] = accumulate ? out[o] + v : v;
}

// ---- host ---------------------------------------------------------------------------------------

// collapse a list of (shape, stride) pairs in place; returns the new count (>= 1)
static int collapse(int n, int64_t* shp, int64_t* st) {
    int m = 0;
    for (int d = 0; d < n; ++d) {
        if (shp[d] == 1) continue;
        if (m > 0 && st[m - 1] == st[d] * shp[d]) {
            shp[m - 1] *= shp[d];
            st[m - 1] = st[d];
        } else {
            shp[m] = shp[d];
            st[m] = st[d];
            ++m;
        }
    }
    if (m == 0) { shp[0] = 1; st[0] = 0; m = 1; }
    return m;
}

template <int OP>
static int run_reduce(const float* in, float* out, RedDesc& d) {
    hipStream_t s = rt().stream;
    const bool rows = (d.nr == 1) && (d.rstride[0] == 1 || d.rlen == 1);
    if (rows) {
        // many rows, or short rows: a wave per row
        if (d.n_out >= 256 || d.rlen <= 8192) {
            hipLaunchKernelGGL((red_rows_wave<OP>), dim3(unsigned((d.n_out + 3) / 4)), dim3(256), 0, s, in, out, d);
            return LG_OK;
        }
        // few long rows: split each row over enough blocks to fill the chip
        static const char* rb_env = getenv("LG_RED_BLOCKS");
        int64_t want_blocks = rb_env ? atoi(rb_env) : 768;       // measured best for a 512 MiB full sum: 5.77 TB/s (tools/reduce_bench.py)
        int64_t splits = (want_blocks + d.n_out - 1) / d.n_out;
        int64_t min_seg = 4096;                                   // at least 16 KiB per block
        if (splits * min_seg > d.rlen) splits = (d.rlen + min_seg - 1) / min_seg;
        if (splits < 1) splits = 1;
        int64_t seg = ((d.rlen + splits - 1) / splits + 3) & ~int64_t(3);
        splits = (d.rlen + seg - 1) / seg;
        if (d.n_out > 65535) {   // grid.y limit: fall back to a wave per row
            hipLaunchKernelGGL((red_rows_wave<OP>), dim3(unsigned((d.n_out + 3) / 4)), dim3(256), 0, s, in, out, d);
            return LG_OK;
        }
        if (splits == 1) {
            hipLaunchKernelGGL((red_rows_split<OP>), dim3(1, unsigned(d.n_out)), dim3(256), 0, s, in, out, nullptr, nullptr, d, seg, int64_t(1));
            return LG_OK;
        }
        const int64_t groups = (splits + kRowsFoldGroup - 1) / kRowsFoldGroup, n_tickets = d.n_out * (groups + 1);
        if (n_tickets > rt().n_gemm_tickets) {   // (n_out < 256 on this path and a few dozen groups: the pool is far larger)
            hipLaunchKernelGGL((red_rows_wave<OP>), dim3(unsigned((d.n_out + 3) / 4)), dim3(256), 0, s, in, out, d);
            return LG_OK;
        }
        int64_t tstride = rt().n_gemm_tickets / n_tickets;          // tickets on cache lines of their own while the pool allows
        if (tstride > 32) tstride = 32;
        float* partial = nullptr;
        int rc = lg_malloc(reinterpret_cast<void**>(&partial), size_t(d.n_out * (splits + groups)) * sizeof(float));
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL((red_rows_split<OP>), dim3(unsigned(splits), unsigned(d.n_out)), dim3(256), 0, s, in, out, partial,
                           rt().gemm_tickets, d, seg, tstride);
        return lg_free(partial);   // stream-ordered: the block is only reused by later launches
    }

    // columns of a row-major matrix (kept axis contiguous, one strided reduced axis): the tiled kernel
    if (d.nk == 1 && d.nr == 1 && d.kstride[0] == 1 && d.n_out % 4 == 0 && d.rstride[0] % 4 == 0 && aligned16(in) && d.rlen >= 64 &&
        d.n_out >= 64) {
        const int64_t bx = (d.n_out + 63) / 64;
        static const char* tw_env = getenv("LG_RED_TILE_WGS");      // experiments only
        // workgroups in total: one per CU while the matrix fits the 256 MiB Infinity Cache, two beyond (measured, sum(axis=0):
        // 4096^2: 256 -> 13.0 us, 512 -> 14.1, 1024 -> 16.7 (red_cols: 23.7); 16384 x 8192: 256 -> 108 us, 384-512 -> 86, 1024 -> 97
        // (red_cols: 91)); few enough row chunks that the fold fetches every partial at once
        const int64_t target = tw_env ? atoi(tw_env) : (d.n_out * d.rlen * 4 <= (int64_t(128) << 20) ? 256 : 512);
        int64_t splits = (target + bx - 1) / bx;
        if (splits > 16) splits = 16;
        if (splits * 64 > d.rlen) splits = d.rlen / 64;            // at least 64 rows per workgroup
        if (splits < 1) splits = 1;
        int64_t chunk = ((d.rlen + splits - 1) / splits + 15) & ~int64_t(15);
        splits = (d.rlen + chunk - 1) / chunk;
        if (bx <= rt().n_gemm_tickets && bx < (int64_t(1) << 31)) {
            float* partial = nullptr;
            if (splits > 1) {
                int rc = lg_malloc(reinterpret_cast<void**>(&partial), size_t(d.n_out * splits) * sizeof(float));
                if (rc != LG_OK) return rc;
            }
            hipLaunchKernelGGL((red_cols_tile<OP>), dim3(unsigned(bx), unsigned(splits)), dim3(256), 0, s, in, out, partial,
                               rt().gemm_tickets, d.n_out, d.rlen, d.rstride[0], chunk, d.accumulate);
            return splits > 1 ? lg_free(partial) : LG_OK;
        }
    }
    // general / column reduce
    int64_t blocks_x = (d.n_out + 255) / 256;
    int64_t splits = 1;
    static const char* cb_env = getenv("LG_RED_BLOCKS");
    const int64_t col_blocks = cb_env ? atoi(cb_env) : 1536;     // measured best for sum(axis=0) of 16384 x 8192: 5.81 TB/s
    if (blocks_x < col_blocks && d.rlen >= 64) {
        splits = col_blocks / blocks_x;
        if (splits * 16 > d.rlen) splits = d.rlen / 16;           // at least 16 elements per thread
        // the workgroup that folds reads one partial per split and thread, 8 at a time across XCDs: beyond ~32 splits
        // that serial tail outweighs the extra parallelism (4096^2: 96 splits 34 us, 32 splits 21 us)
        const int64_t cap = d.rlen >= 8192 ? 64 : 32;
        if (splits > cap) splits = cap;
        if (splits < 1) splits = 1;
        if (splits > 65535) splits = 65535;
    }
    int64_t chunk = (d.rlen + splits - 1) / splits;
    splits = (d.rlen + chunk - 1) / chunk;
    if (blocks_x >= (int64_t(1) << 31)) { set_error("lg_reduce: output too large"); return LG_EINVAL; }
    if (splits > 1 && blocks_x > rt().n_gemm_tickets) { splits = 1; chunk = d.rlen; }     // more output blocks than tickets
    if (splits == 1) {
        hipLaunchKernelGGL((red_cols<OP>), dim3(unsigned(blocks_x), 1), dim3(256), 0, s, in, out, nullptr, nullptr, d, chunk);
        return LG_OK;
    }
    float* partial = nullptr;
    int rc = lg_malloc(reinterpret_cast<void**>(&partial), size_t(d.n_out * splits) * sizeof(float));
    if (rc != LG_OK) return rc;
    hipLaunchKernelGGL((red_cols<OP>), dim3(unsigned(blocks_x), unsigned(splits)), dim3(256), 0, s, in, out, partial,
                       rt().gemm_tickets, d, chunk);
    return lg_free(partial);
}

}  // namespace lg

using namespace lg;

extern "C" int lg_reduce_acc(int op, int ndim, const int64_t* shape, const void* in, const int64_t* in_strides,
                             uint32_t axis_mask, void* out, int accumulate);

extern "C" int lg_reduce(int op, int ndim, const int64_t* shape, const void* in, const int64_t* in_strides,
                         uint32_t axis_mask, void* out) {
    return lg_reduce_acc(op, ndim, shape, in, in_strides, axis_mask, out, 0);
}

extern "C" int lg_reduce_acc(int op, int ndim, const int64_t* shape, const void* in, const int64_t* in_strides,
                             uint32_t axis_mask, void* out, int accumulate) {
    LG_REQUIRE_INIT();
    LG_ARG(!accumulate || op == LG_RED_SUM, "lg_reduce_acc: accumulate is defined for sums only");
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_reduce: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(in != nullptr && out != nullptr, "lg_reduce: NULL pointer");
    LG_ARG(ndim == 0 || (shape != nullptr && in_strides != nullptr), "lg_reduce: NULL shape/strides");
    LG_ARG(op == LG_RED_SUM || op == LG_RED_MAX || op == LG_RED_MIN, "lg_reduce: unknown op id %d", op);
    LG_ARG((axis_mask >> ndim) == 0, "lg_reduce: axis_mask 0x%x names a dimension >= ndim %d", axis_mask, ndim);

    RedDesc d{};
    d.accumulate = accumulate ? 1 : 0;
    int nk = 0, nr = 0;
    d.n_out = 1;
    d.rlen = 1;
    for (int k = 0; k < ndim; ++k) {
        LG_ARG(shape[k] >= 0, "lg_reduce: negative extent");
        if ((axis_mask >> k) & 1u) {
            d.rshape[nr] = shape[k]; d.rstride[nr] = in_strides[k]; ++nr;
            d.rlen *= shape[k];
        } else {
            d.kshape[nk] = shape[k]; d.kstride[nk] = in_strides[k]; ++nk;
            d.n_out *= shape[k];
        }
    }
    if (d.n_out == 0) return LG_OK;
    if (d.rlen == 0) {
        LG_ARG(op == LG_RED_SUM, "lg_reduce: zero-size reduction has no identity for max/min");
        if (accumulate) return LG_OK;
        uint64_t zero = 0;
        int64_t n = d.n_out, one = 1;
        return lg_fill_strided(4, 1, &n, out, &one, zero);
    }
    d.nk = collapse(nk, d.kshape, d.kstride);
    d.nr = collapse(nr, d.rshape, d.rstride);

    int rc;
    const float* src = static_cast<const float*>(in);
    float* dst = static_cast<float*>(out);
    switch (op) {
        case LG_RED_SUM: rc = run_reduce<LG_RED_SUM>(src, dst, d); break;
        case LG_RED_MAX: rc = run_reduce<LG_RED_MAX>(src, dst, d); break;
        default:         rc = run_reduce<LG_RED_MIN>(src, dst, d); break;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
