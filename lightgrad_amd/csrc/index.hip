// Integer-array ("fancy") indexing along ONE axis, for any item size: gather, assignment and fp32 scatter-add
// (SURVEY.md 8f row 3).  The reference does these on the CPU backend with numpy (cpu/ops.py:234-255: `a[idx]`,
// `grad[idx] = out_grad`, `a[idx] = val`) and its OpenCL backend has no counterpart (examples/bert.py:19-21 round-trips
// embeddings through the host); its callers are Dataset shuffling / batching (data.py:15-32: `t[perm]`), embedding
// lookups and the label pick `y[range(n), labels]` of loss.cross_entropy (loss.py:19, :22).
//
// The tensor is seen as [outer][axis_len][inner] (dense); element e of the result [outer][n_idx][inner] reads
// src[o][idx[j]][i].  `pair_period` = n > 0 selects the paired form instead: the index array has n entries, entry
// (o mod n) belongs to outer position o, and the result is [outer][inner] - `y[range(n), labels]` with outer = n.
// Negative indices count from the end like numpy's.  An index outside [-axis_len, axis_len) cannot raise from a
// kernel: the gather writes all-ones bytes (NaN for floats, -1 for integers), the writes are skipped, and the launch
// raises the device status flag, which the next synchronising call (lg_sync, lg_memcpy_d2h) turns into LG_EINDEX -
// numpy's IndexError, one synchronisation late.  HBM-bound: itemsize bytes read + written per element.
#include "common.h"

namespace lg {

struct AxisIndex {
    int64_t outer, axis_len, inner, n_idx, total;
    int64_t pair_period;      // 0: plain take; n: paired
    int*    status;
};

template <typename IdT>
__device__ __forceinline__ bool locate(const AxisIndex& d, const IdT* __restrict__ idx, int64_t e, int64_t& src_off) {
    int64_t o, j, i;
    if (d.pair_period) {
        o = e / d.inner;
        i = e - o * d.inner;
        j = o % d.pair_period;
    } else {
        const int64_t per_outer = d.n_idx * d.inner;
        o = e / per_outer;
        const int64_t rem = e - o * per_outer;
        j = rem / d.inner;
        i = rem - j * d.inner;
    }
    int64_t r = int64_t(idx[j]);
    if (r < 0) r += d.axis_len;
    if (r < 0 || r >= d.axis_len) {
        __hip_atomic_fetch_or(d.status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    src_off = (o * d.axis_len + r) * d.inner + i;
    return true;
}

template <typename T, typename IdT>
__global__ void __launch_bounds__(256) take_axis(const T* __restrict__ src, const IdT* __restrict__ idx, T* __restrict__ dst, AxisIndex d) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.total; e += stride) {
        int64_t off;
        dst[e] = locate(d, idx, e, off) ? src[off] : T(~T(0));
    }
}

template <typename T, typename IdT>
__global__ void __launch_bounds__(256) put_axis(T* __restrict__ dst, const IdT* __restrict__ idx, const T* __restrict__ val, T scalar, AxisIndex d) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.total; e += stride) {
        int64_t off;
        if (locate(d, idx, e, off)) dst[off] = val ? val[e] : scalar;
    }
}

template <typename IdT>
__global__ void __launch_bounds__(256) scatter_add_axis(float* __restrict__ dst, const IdT* __restrict__ idx, const float* __restrict__ src, AxisIndex d) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.total; e += stride) {
        int64_t off;
        if (locate(d, idx, e, off)) atomicAdd(dst + off, src[e]);          // repeated indices accumulate (global_atomic_add_f32)
    }
}

static bool describe(AxisIndex& d, int64_t outer, int64_t axis_len, int64_t inner, int64_t n_idx, int64_t pair_period) {
    if (outer < 0 || axis_len < 0 || inner < 0 || n_idx < 0 || pair_period < 0) return false;
    if (pair_period && (pair_period != n_idx || outer % pair_period)) return false;
    d.outer = outer; d.axis_len = axis_len; d.inner = inner; d.n_idx = n_idx; d.pair_period = pair_period;
    d.total = pair_period ? outer * inner : outer * n_idx * inner;
    d.status = rt().status_dev;
    return true;
}

template <typename IdT>
static void launch_take(int itemsize, const void* src, const IdT* idx, void* dst, const AxisIndex& d) {
    const dim3 grid(stream_grid(d.total)), block(256);
    hipStream_t s = rt().stream;
    switch (itemsize) {
        case 1: hipLaunchKernelGGL((take_axis<uint8_t, IdT>), grid, block, 0, s, static_cast<const uint8_t*>(src), idx, static_cast<uint8_t*>(dst), d); break;
        case 2: hipLaunchKernelGGL((take_axis<uint16_t, IdT>), grid, block, 0, s, static_cast<const uint16_t*>(src), idx, static_cast<uint16_t*>(dst), d); break;
        case 4: hipLaunchKernelGGL((take_axis<uint32_t, IdT>), grid, block, 0, s, static_cast<const uint32_t*>(src), idx, static_cast<uint32_t*>(dst), d); break;
        default: hipLaunchKernelGGL((take_axis<uint64_t, IdT>), grid, block, 0, s, static_cast<const uint64_t*>(src), idx, static_cast<uint64_t*>(dst), d); break;
    }
}

template <typename T, typename IdT>
static void launch_put_typed(void* dst, const IdT* idx, const void* val, uint64_t bits, const AxisIndex& d) {
    T scalar;
    memcpy(&scalar, &bits, sizeof(T));
    hipLaunchKernelGGL((put_axis<T, IdT>), dim3(stream_grid(d.total)), dim3(256), 0, rt().stream, static_cast<T*>(dst), idx,
                       static_cast<const T*>(val), scalar, d);
}

template <typename IdT>
static void launch_put(int itemsize, void* dst, const IdT* idx, const void* val, uint64_t bits, const AxisIndex& d) {
    switch (itemsize) {
        case 1: launch_put_typed<uint8_t, IdT>(dst, idx, val, bits, d); break;
        case 2: launch_put_typed<uint16_t, IdT>(dst, idx, val, bits, d); break;
        case 4: launch_put_typed<uint32_t, IdT>(dst, idx, val, bits, d); break;
        default: launch_put_typed<uint64_t, IdT>(dst, idx, val, bits, d); break;
    }
}

}  // namespace lg

using namespace lg;

#define LG_INDEX_ARGS(fn)                                                                                               \
    LG_REQUIRE_INIT();                                                                                                  \
    LG_ARG(idx_itemsize == 2 || idx_itemsize == 4 || idx_itemsize == 8, fn ": indices must be int16, int32 or int64");  \
    AxisIndex d;                                                                                                        \
    LG_ARG(describe(d, outer, axis_len, inner, n_idx, pair_period), fn ": bad extents");                                \
    if (d.total == 0) return LG_OK;                                                                                     \
    LG_ARG(idx != nullptr, fn ": NULL index array")

extern "C" int lg_take_axis(int itemsize, const void* src, int64_t outer, int64_t axis_len, int64_t inner,
                            const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, void* dst) {
    LG_INDEX_ARGS("lg_take_axis");
    LG_ARG(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8, "lg_take_axis: itemsize %d not in {1,2,4,8}", itemsize);
    LG_ARG(src && dst, "lg_take_axis: NULL pointer");
    if (idx_itemsize == 2)      launch_take(itemsize, src, static_cast<const int16_t*>(idx), dst, d);
    else if (idx_itemsize == 4) launch_take(itemsize, src, static_cast<const int32_t*>(idx), dst, d);
    else                        launch_take(itemsize, src, static_cast<const int64_t*>(idx), dst, d);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_put_axis(int itemsize, void* dst, int64_t outer, int64_t axis_len, int64_t inner,
                           const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, const void* val, uint64_t scalar_bits) {
    LG_INDEX_ARGS("lg_put_axis");
    LG_ARG(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8, "lg_put_axis: itemsize %d not in {1,2,4,8}", itemsize);
    LG_ARG(dst != nullptr, "lg_put_axis: NULL pointer");
    if (idx_itemsize == 2)      launch_put(itemsize, dst, static_cast<const int16_t*>(idx), val, scalar_bits, d);
    else if (idx_itemsize == 4) launch_put(itemsize, dst, static_cast<const int32_t*>(idx), val, scalar_bits, d);
    else                        launch_put(itemsize, dst, static_cast<const int64_t*>(idx), val, scalar_bits, d);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_scatter_add_axis_f32(float* dst, int64_t outer, int64_t axis_len, int64_t inner,
                                       const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, const float* src) {
    LG_INDEX_ARGS("lg_scatter_add_axis_f32");
    LG_ARG(dst && src, "lg_scatter_add_axis_f32: NULL pointer");
    const dim3 grid(stream_grid(d.total)), block(256);
    hipStream_t s = rt().stream;
    if (idx_itemsize == 2)      hipLaunchKernelGGL(scatter_add_axis<int16_t>, grid, block, 0, s, dst, static_cast<const int16_t*>(idx), src, d);
    else if (idx_itemsize == 4) hipLaunchKernelGGL(scatter_add_axis<int32_t>, grid, block, 0, s, dst, static_cast<const int32_t*>(idx), src, d);
    else                        hipLaunchKernelGGL(scatter_add_axis<int64_t>, grid, block, 0, s, dst, static_cast<const int64_t*>(idx), src, d);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

// ---- several index arrays folded into one, boolean masks compacted: on the device (round 4) ---------------------------------
// numpy's `a[i, j]` with index arrays i, j (cpu/ops.py:234-255 hands the tuple to numpy) reads a[i[e], j[e]] for every element
// e of the arrays' common BROADCAST shape.  lg_index_fold forms the row-major flat index i[e] * len_j + j[e] (any number of
// arrays up to 8, integers of 2 / 4 / 8 bytes, constants for plain integers in the tuple) so that the one-array gather /
// assignment / scatter-add kernels above do the rest - without reading device-resident index arrays back to the host.
// A boolean mask selects where it is True: lg_mask_nonzero writes the flat positions of the True elements, in order (numpy's
// mask.nonzero() folded over the mask's axes), and their number - which the host needs for the result's SHAPE: one 8-byte
// read-back instead of the whole mask.
namespace lg {

constexpr int kFoldMax = 8;
struct FoldArgs {
    int         k, ndim;
    int64_t     numel;
    int64_t     shape[kFoldMax];              // the broadcast shape (ndim entries)
    int64_t     len[kFoldMax];                // extent each index runs over
    const void* idx[kFoldMax];                // NULL: the constant
    int64_t     constant[kFoldMax];
    int         itemsize[kFoldMax];
    int64_t     stride[kFoldMax][kFoldMax];   // [array][dim], elements, 0 where the array is broadcast
    int*        status;
};

__global__ void __launch_bounds__(256) index_fold(FoldArgs a, int64_t* __restrict__ out) {
    const int64_t step = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < a.numel; e += step) {
        int64_t pos[kFoldMax];
        int64_t rem = e;
        for (int d = a.ndim - 1; d >= 0; --d) { pos[d] = rem % a.shape[d]; rem /= a.shape[d]; }
        int64_t flat = 0;
        bool bad = false;
        for (int j = 0; j < a.k; ++j) {
            int64_t v = a.constant[j];
            if (a.idx[j]) {
                int64_t off = 0;
                for (int d = 0; d < a.ndim; ++d) off += pos[d] * a.stride[j][d];
                v = a.itemsize[j] == 2 ? int64_t(static_cast<const int16_t*>(a.idx[j])[off])
                  : a.itemsize[j] == 4 ? int64_t(static_cast<const int32_t*>(a.idx[j])[off]) : static_cast<const int64_t*>(a.idx[j])[off];
            }
            if (v < 0) v += a.len[j];
            bad = bad || v < 0 || v >= a.len[j];
            flat = flat * a.len[j] + v;
        }
        if (bad) {
            __hip_atomic_fetch_or(a.status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            flat = 0;                          // (a valid position: the consumer kernels read / write something harmless; the flag reports)
        }
        out[e] = flat;
    }
}

constexpr int kMaskPerThread = 16, kMaskPerBlock = 256 * kMaskPerThread;

__global__ void __launch_bounds__(256) mask_count(const uint8_t* __restrict__ mask, int64_t n, int64_t* __restrict__ block_counts) {
    __shared__ int red[256];
    const int64_t base = int64_t(blockIdx.x) * kMaskPerBlock + int64_t(threadIdx.x) * kMaskPerThread;
    int c = 0;
    for (int i = 0; i < kMaskPerThread; ++i) c += (base + i < n && mask[base + i]) ? 1 : 0;
    red[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (int(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0];
}

// exclusive scan of the block counts by ONE workgroup (chunks of 256 with a running carry); total -> *count
__global__ void __launch_bounds__(256) mask_scan(int64_t* __restrict__ block_counts, int64_t nblocks, int64_t* __restrict__ count) {
    __shared__ int64_t buf[256];
    int64_t carry = 0;
    for (int64_t c0 = 0; c0 < nblocks; c0 += 256) {
        const int64_t i = c0 + threadIdx.x;
        const int64_t v = i < nblocks ? block_counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int w = 1; w < 256; w <<= 1) {
            const int64_t add = int(threadIdx.x) >= w ? buf[threadIdx.x - w] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks) block_counts[i] = carry + buf[threadIdx.x] - v;        // exclusive
        carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ void __launch_bounds__(256) mask_positions(const uint8_t* __restrict__ mask, int64_t n, const int64_t* __restrict__ block_offsets,
                                                      int64_t* __restrict__ out) {
    __shared__ int scan[256];
    const int64_t base = int64_t(blockIdx.x) * kMaskPerBlock + int64_t(threadIdx.x) * kMaskPerThread;
    int c = 0;
    for (int i = 0; i < kMaskPerThread; ++i) c += (base + i < n && mask[base + i]) ? 1 : 0;
    scan[threadIdx.x] = c;
    __syncthreads();
    for (int w = 1; w < 256; w <<= 1) {
        const int add = int(threadIdx.x) >= w ? scan[threadIdx.x - w] : 0;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t at = block_offsets[blockIdx.x] + scan[threadIdx.x] - c;
    for (int i = 0; i < kMaskPerThread; ++i)
        if (base + i < n && mask[base + i]) out[at++] = base + i;
}

}  // namespace lg

extern "C" int lg_index_fold(int k, const int64_t* lens, const void* const* idx, const int* idx_itemsize, const int64_t* constants,
                             int ndim, const int64_t* shape, const int64_t* const* idx_strides, int64_t* out) {
    LG_REQUIRE_INIT();
    LG_ARG(k >= 1 && k <= kFoldMax && ndim >= 0 && ndim <= kFoldMax, "lg_index_fold: %d index arrays / %d dimensions (at most %d)", k, ndim, kFoldMax);
    LG_ARG(lens && idx && idx_itemsize && constants && out && (ndim == 0 || (shape && idx_strides)), "lg_index_fold: NULL pointer");
    FoldArgs a{};
    a.k = k; a.ndim = ndim; a.numel = 1; a.status = rt().status_dev;
    for (int d = 0; d < ndim; ++d) { LG_ARG(shape[d] >= 0, "lg_index_fold: negative extent"); a.shape[d] = shape[d]; a.numel *= shape[d]; }
    for (int j = 0; j < k; ++j) {
        LG_ARG(lens[j] >= 0, "lg_index_fold: negative axis length");
        a.len[j] = lens[j]; a.idx[j] = idx[j]; a.constant[j] = constants[j]; a.itemsize[j] = idx_itemsize[j];
        LG_ARG(idx[j] == nullptr || idx_itemsize[j] == 2 || idx_itemsize[j] == 4 || idx_itemsize[j] == 8, "lg_index_fold: indices must be int16, int32 or int64");
        if (idx[j]) for (int d = 0; d < ndim; ++d) a.stride[j][d] = idx_strides[j][d];
    }
    if (a.numel == 0) return LG_OK;
    hipLaunchKernelGGL(index_fold, dim3(stream_grid(a.numel)), dim3(256), 0, rt().stream, a, out);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_mask_nonzero(const void* mask, int64_t n, int64_t* out_positions, int64_t* out_count) {
    LG_REQUIRE_INIT();
    LG_ARG(n >= 0 && out_count != nullptr && (n == 0 || (mask && out_positions)), "lg_mask_nonzero: bad arguments");
    if (n == 0) { LG_HIP(hipMemsetAsync(out_count, 0, sizeof(int64_t), rt().stream)); return LG_OK; }
    const int64_t nblocks = (n + kMaskPerBlock - 1) / kMaskPerBlock;
    LG_ARG(nblocks < (int64_t(1) << 31), "lg_mask_nonzero: mask too large");
    int64_t* counts = nullptr;
    { const int rc = lg_malloc(reinterpret_cast<void**>(&counts), size_t(nblocks) * sizeof(int64_t)); if (rc != LG_OK) return rc; }
    hipStream_t s = rt().stream;
    const uint8_t* m = static_cast<const uint8_t*>(mask);
    hipLaunchKernelGGL(mask_count, dim3(unsigned(nblocks)), dim3(256), 0, s, m, n, counts);
    hipLaunchKernelGGL(mask_scan, dim3(1), dim3(256), 0, s, counts, nblocks, out_count);
    hipLaunchKernelGGL(mask_positions, dim3(unsigned(nblocks)), dim3(256), 0, s, m, n, counts, out_positions);
    LG_CHECK_LAUNCH();
    return lg_free(counts);          // stream-ordered
}
