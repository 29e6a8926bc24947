// Row-wise fused kernels for the tiny-BERT path: softmax, LayerNorm (forward and
// backward) and embedding gather / scatter-add.  All HBM-bound: one pass over each row, the row lives in
// registers of ONE wavefront (rows up to 64 * 32 = 2048 floats; the BERT rows are 128 wide) and is reduced with
// wave64 shuffles - no LDS, no second read.  Longer rows take the same code with a strided loop that re-reads.
//
// Reference semantics (composites these replace):
//   softmax   autograd/ops.py:62-66   exps = (t - max).exp(); exps / exps.sum()      ('/' = a * b**-1, ops.py:30-36)
//   LayerNorm nn.py:109-124           D = x - mean; V = mean(D*D); D / (V + eps)**0.5 * weight + bias
//   Embedding examples/bert.py:14-21  weight[ids]  (the reference round-trips through the CPU and drops the gradient)
#include "common.h"
#include "tail_jobs.h"

namespace lg {

constexpr int kRowRegs = 32;     // floats per lane held in registers: rows up to 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = (v > o || v != v) ? v : o;      // NaN propagates like np.max
    }
    return v;
}

// ---- softmax over the last axis --------------------------------------------------------------------
// one wave per row; 4 rows per 256-thread block
// REGS = floats per lane kept in registers (row length <= 64 * REGS); 0 = strided loops that re-read the row
template <int REGS>
__global__ void __launch_bounds__(256) softmax_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    constexpr bool IN_REGS = REGS > 0;
    float v[IN_REGS ? REGS : 1];
    float m = -INFINITY;
    if constexpr (IN_REGS) {
#pragma unroll
        for (int k = 0; k < REGS; ++k) {
            const int64_t c = lane + 64 * k;
            v[k] = c < cols ? xr[c] * scale : -INFINITY;        // softmax(x * scale): the product rounded to fp32 first, like the two-op form
            m = (v[k] > m || v[k] != v[k]) ? v[k] : m;
        }
    } else {
        for (int64_t c = lane; c < cols; c += 64) { const float t = xr[c] * scale; m = (t > m || t != t) ? t : m; }
    }
    m = wave_max(m);
    float s = 0.f;
    if constexpr (IN_REGS) {
#pragma unroll
        for (int k = 0; k < REGS; ++k) {
            const int64_t c = lane + 64 * k;
            v[k] = c < cols ? expf(v[k] + (-m)) : 0.f;
            s += v[k];
        }
    } else {
        for (int64_t c = lane; c < cols; c += 64) s += expf(xr[c] * scale + (-m));
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    if constexpr (IN_REGS) {
#pragma unroll
        for (int k = 0; k < REGS; ++k) {
            const int64_t c = lane + 64 * k;
            if (c < cols) yr[c] = v[k] * inv;
        }
    } else {
        for (int64_t c = lane; c < cols; c += 64) yr[c] = expf(xr[c] * scale + (-m)) * inv;
    }
}

// dx = y * (g - sum(g * y) / sum(y)) [* scale, for softmax(x * scale)].  The shift sum(g*y) is formed, divided by sum(y) (1 up to the rounding of y) and
// subtracted in DOUBLE: every row of dx must sum to zero, and consumers rely on it - the query / key gradients of
// attention are d(scores) @ k and d(scores)^T @ q with k, q nearly constant along the summed axis at initialisation, a
// contraction that cancels everything EXCEPT the row sums of d(scores) (condition ~150 on tiny-BERT).  With an fp32 shift
// every element of a row carries the same error eps*|shift| (|shift| ~ 30 |dx| there): measured against a float64 run of the
// tape the query-weight gradient was 3.3e-4 off, the fp32 CPU backend 9e-6 (tools/bert_grad_probe.py); with the double shift
// the error left is eps*|dx| per element.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(256) softmax_bwd(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ dx,
                                                   int64_t rows, int64_t cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* yr = y + row * cols;
    const float* gr = g + row * cols;
    float* dr = dx + row * cols;
    double dot = 0.0, norm = 0.0;
    for (int64_t c = lane; c < cols; c += 64) { const double yc = double(yr[c]); dot += double(gr[c]) * yc; norm += yc; }
    dot = wave_sum_f64(dot);
    norm = wave_sum_f64(norm);
    const double shift = dot / norm;
    for (int64_t c = lane; c < cols; c += 64) dr[c] = float(double(yr[c]) * (double(gr[c]) - shift)) * scale;
}

// ---- LayerNorm over the last axis ---------------------------------------------------------------------
// y = ((x - mean) * rstd) * w + b ; saves xhat = (x - mean) * rstd and rstd for the backward
__global__ void __launch_bounds__(256) layernorm_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd,
                                                     int64_t rows, int64_t cols, float eps, float inv_n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    float s = 0.f;
    for (int64_t c = lane; c < cols; c += 64) s += xr[c];
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
    for (int64_t c = lane; c < cols; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float var = wave_sum(q) * inv_n;
    const float r = 1.0f / sqrtf(var + eps);
    if (lane == 0) rstd[row] = r;
    for (int64_t c = lane; c < cols; c += 64) {
        const float h = (xr[c] - mean) * r;
        xhat[row * cols + c] = h;
        y[row * cols + c] = h * w[c] + b[c];
    }
}

// dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)),  gh = g * w
__global__ void __launch_bounds__(256) layernorm_bwd(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, float* __restrict__ dx, int64_t rows, int64_t cols,
                                                     float inv_n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* gr = g + row * cols;
    const float* hr = xhat + row * cols;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t c = lane; c < cols; c += 64) {
        const float gh = gr[c] * w[c];
        s1 += gh;
        s2 += gh * hr[c];
    }
    s1 = wave_sum(s1) * inv_n;
    s2 = wave_sum(s2) * inv_n;
    const float r = rstd[row];
    for (int64_t c = lane; c < cols; c += 64) dx[row * cols + c] = r * (gr[c] * w[c] - s1 - hr[c] * s2);
}

__global__ void __launch_bounds__(256) layernorm_param_grads(LnParamGrads a) {
    layernorm_param_grads_body(a, int(blockIdx.x), blockIdx.y);
}

__global__ void __launch_bounds__(256) layernorm_param_grads_group(LnParamGradsGroup grp) {
    const LnParamGrads& a = grp.e[blockIdx.z];
    if (int(blockIdx.x) >= a.blocks_x || int(blockIdx.y) >= a.splits) return;      // this entry's own grid is smaller
    layernorm_param_grads_body(a, int(blockIdx.x), blockIdx.y);
}

// ---- cross entropy of softmax(logits) against integer labels (reference loss.py:14-24) ------------------------
// one wave per row: p = softmax(row); nll[row] = -log(p[label]); d[row][c] = (p[c] - [c == label]) / rows
template <typename LabelT>
__global__ void __launch_bounds__(256) cross_entropy_rows(const float* __restrict__ x, const LabelT* __restrict__ labels,
                                                          float* __restrict__ dlogits, float* __restrict__ nll, int64_t rows,
                                                          int64_t cols, float inv_rows, int* status) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    float m = -INFINITY;
    for (int64_t c = lane; c < cols; c += 64) { const float t = xr[c]; m = (t > m || t != t) ? t : m; }
    m = wave_max(m);
    float s = 0.f;
    for (int64_t c = lane; c < cols; c += 64) s += expf(xr[c] + (-m));
    s = wave_sum(s);
    const float inv = 1.0f / s;
    int64_t label = int64_t(labels[row]);
    if (label < 0) label += cols;
    if (label < 0 || label >= cols) {                 // the reference raises IndexError (loss.py:19); a kernel cannot: NaN + status flag
        if (lane == 0) { nll[row] = __builtin_nanf(""); __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    for (int64_t c = lane; c < cols; c += 64) {
        const float p = expf(xr[c] + (-m)) * inv;
        dlogits[row * cols + c] = (c == label ? p - 1.0f : p) * inv_rows;
        if (c == label) nll[row] = -logf(p);
    }
}

// wide rows (a vocabulary): one WORKGROUP per row.  Pass 1 keeps a running (max, sum of exp) per thread over coalesced
// loads, combined across the workgroup; pass 2 re-reads the row (L2 / Infinity Cache) and writes the gradient.
template <typename LabelT>
__global__ void __launch_bounds__(256) cross_entropy_wide(const float* __restrict__ x, const LabelT* __restrict__ labels,
                                                          float* __restrict__ dlogits, float* __restrict__ nll, int64_t cols,
                                                          float inv_rows, int* status) {
    __shared__ float red_m[4], red_s[4];
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xr = x + row * cols;
    float m = -INFINITY, sum = 0.f;
    auto absorb = [&](float t) {
        if (t > m || t != t) {                       // new maximum (NaN propagates like np.max): rescale the running sum
            sum = sum * expf(m - t) + 1.0f;
            m = t;
        } else if (t != -INFINITY) {                 // (-inf) - (-inf) would be NaN; exp(-inf - m) is 0 anyway
            sum += expf(t - m);
        }
    };
    int64_t c = threadIdx.x;
    for (; c + 768 < cols; c += 1024) {
        const float t0 = xr[c], t1 = xr[c + 256], t2 = xr[c + 512], t3 = xr[c + 768];
        absorb(t0); absorb(t1); absorb(t2); absorb(t3);
    }
    for (; c < cols; c += 256) absorb(xr[c]);
    // combine (m, sum) pairs: wave, then the four waves
    const float wm = wave_max(m);
    sum = (m == -INFINITY && wm == -INFINITY) ? 0.f : sum * expf(m - wm);
    sum = wave_sum(sum);
    if (lane == 0) { red_m[wave] = wm; red_s[wave] = sum; }
    __syncthreads();
    float M = red_m[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = (red_m[w] > M || red_m[w] != red_m[w]) ? red_m[w] : M;
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += (red_m[w] == -INFINITY && M == -INFINITY) ? 0.f : red_s[w] * expf(red_m[w] - M);
    const float inv = 1.0f / S;
    int64_t label = int64_t(labels[row]);
    if (label < 0) label += cols;
    if (label < 0 || label >= cols) {
        if (threadIdx.x == 0) { nll[row] = __builtin_nanf(""); __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    for (c = threadIdx.x; c < cols; c += 256) {
        const float p = expf(xr[c] + (-M)) * inv;
        dlogits[row * cols + c] = (c == label ? p - 1.0f : p) * inv_rows;
        if (c == label) nll[row] = -logf(p);
    }
}

// wide rows that one workgroup can HOLD in registers (cols <= THREADS * PER): the row is read from memory once - maximum, exp
// and sum work on the held values (numpy's own order of operations: max, exp(x - max), sum, divide) and the gradient is
// written straight from them.  Two shapes: 1024 threads x 8 / 16 values, and for a vocabulary 512 threads x 60 values - at 128
// VGPRs a CU then holds TWO independent workgroups whose load, exp and store phases overlap.  BERT's (1024, 30522) logits,
// back-to-back launches (tools/ce_bench.py): two-pass kernel 72.7 us, this one 58.6 us = 4.3 TB/s of logits read + gradient
// written (61.4 without the line-aligned walk below).  Measured and not kept: a 2^x-based exp for non-positive arguments, 8
// instructions instead of ~25 - 70.8 us, SLOWER: the kernel is bound by its memory phases, not by the exps.
template <typename LabelT, int PER, int THREADS>
__global__ void __launch_bounds__(THREADS, 4) cross_entropy_held(const float* __restrict__ x, const LabelT* __restrict__ labels,
                                                                 float* __restrict__ dlogits, float* nll, int64_t cols, float inv_rows,
                                                                 int* status, float* mean_out, int* ticket) {
    constexpr int WAVES = THREADS / 64;
    __shared__ float red_m[WAVES], red_s[WAVES];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // A row starts wherever the previous one ended (30522 floats: 8-byte aligned at best).  The threads walk the row from
    // the 128-byte line its first element lies in: every wave access then covers whole lines - unaligned, each 256-byte
    // access touches three lines instead of two, for the reads and for the gradient writes alike.
    const int off = int((row * cols) & 31);      // elements between that line's start and the row's first element
    const float* xr = x + row * cols - off;
    const int ncols = int(cols) + off;           // <= THREADS * PER; positions [off, ncols) are the row
    float v[PER];
    float m = -INFINITY;
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 16) {
#pragma unroll
        for (int i = i0; i < i0 + 16 && i < PER; ++i) {
            const int c = i * THREADS + tid;
            v[i] = xr[c < ncols ? (c >= off ? c : off) : ncols - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = i * THREADS + tid;
        v[i] = (c >= off && c < ncols) ? v[i] : -INFINITY;
        m = (v[i] > m || v[i] != v[i]) ? v[i] : m;
    }
    m = wave_max(m);
    if (lane == 0) red_m[wave] = m;
    __syncthreads();
    float M = red_m[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) M = (red_m[w] > M || red_m[w] != red_m[w]) ? red_m[w] : M;
    float s = 0.f;
    static_assert(PER % 4 == 0, "four values at a time");
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 4) {
#pragma unroll
        for (int i = i0; i < i0 + 4; ++i) {
            const int c = i * THREADS + tid;
            v[i] = (c >= off && c < ncols) ? expf(v[i] + (-M)) : 0.f;
            s += v[i];
        }
        __builtin_amdgcn_sched_barrier(0);       // four exps in flight, not all of them: their temporaries would not fit next to v[]
    }
    s = wave_sum(s);
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) S += red_s[w];
    const float inv = 1.0f / S;
    int64_t label = int64_t(labels[row]);
    if (label < 0) label += cols;
    if (label < 0 || label >= cols) {
        if (tid == 0) { __hip_atomic_store(nll + row, __builtin_nanf(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    const int ilabel = (label < 0 || label >= cols) ? -1 : int(label) + off;
    float* dr = dlogits + row * cols - off;
    float p_label = -1.0f;                         // the probability at the label, for the one thread that holds it
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 8) {
#pragma unroll
        for (int i = i0; i < i0 + 8 && i < PER; ++i) {
            const int c = i * THREADS + tid;
            const float p = v[i] * inv;
            p_label = c == ilabel ? p : p_label;
            if (c >= off && c < ncols) dr[c] = (c == ilabel ? p - 1.0f : p) * inv_rows;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // per-row losses are published write-through: the workgroup that arrives LAST reads all of them for the mean (below)
    if (p_label >= 0.0f || p_label != p_label) __hip_atomic_store(nll + row, -logf(p_label), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mean_out == nullptr) return;
    // mean over rows inside this launch (saves the reduction + scaling launches behind it): drain, take a ticket, the last
    // workgroup sums the row losses in a fixed order - thread t takes rows t, t + THREADS, ...; waves, then the workgroup, in
    // index order - and multiplies by 1/rows like the two-kernel form
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int arrived_last;
    if (tid == 0) {
        const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = order == int(gridDim.x) - 1;
        if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived_last = last;
    }
    __syncthreads();
    if (!arrived_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float total = 0.f;
    for (int64_t i = tid; i < int64_t(gridDim.x); i += THREADS)
        total += __hip_atomic_load(nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    total = wave_sum(total);
    __syncthreads();                             // red_s is free again
    if (lane == 0) red_s[wave] = total;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red_s[w];
        mean_out[0] = t * inv_rows;
    }
}

template <typename LabelT>
static void launch_cross_entropy_held(const float* logits, const LabelT* labels, float* dlogits, float* nll, int64_t rows, int64_t cols,
                                      float inv_rows, float* mean_out) {
    const dim3 grid{unsigned(rows)};
    hipStream_t s = rt().stream;
    int* ticket = rt().gemm_tickets + rt().n_gemm_tickets - 1;          // the last slot: nobody else counts that far
    if (cols + 31 <= 1024 * 8)       hipLaunchKernelGGL((cross_entropy_held<LabelT, 8, 1024>), grid, dim3(1024), 0, s, logits, labels, dlogits, nll, cols, inv_rows, rt().status_dev, mean_out, ticket);
    else if (cols + 31 <= 1024 * 16) hipLaunchKernelGGL((cross_entropy_held<LabelT, 16, 1024>), grid, dim3(1024), 0, s, logits, labels, dlogits, nll, cols, inv_rows, rt().status_dev, mean_out, ticket);
    else                             hipLaunchKernelGGL((cross_entropy_held<LabelT, 60, 512>), grid, dim3(512), 0, s, logits, labels, dlogits, nll, cols, inv_rows, rt().status_dev, mean_out, ticket);
}

// ---- embedding: out[i, :] = table[ids[i], :] ; grad_table[ids[i], :] += grad_out[i, :] -------------------------
template <typename IdT>
__global__ void __launch_bounds__(256) gather_rows(const float* __restrict__ table, const IdT* __restrict__ ids, float* __restrict__ out,
                                                   int64_t n_ids, int64_t row_len, int64_t table_rows, int* status) {
    const int64_t total = n_ids * row_len;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / row_len, c = e - i * row_len;
        int64_t r = int64_t(ids[i]);
        if (r < 0) r += table_rows;                               // numpy-style negative index
        const bool ok = r >= 0 && r < table_rows;
        if (!ok) __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // NaN + LG_EINDEX at the next sync
        out[e] = ok ? table[r * row_len + c] : __builtin_nanf("");
    }
}

// out[i, :] = (t0[ids0[i % n0], :] + t1[ids1[i % n1], :]) + t2[ids2[i % n2], :]: the three lookups of a BERT embedding layer and
// their two additions (examples/bert.py:36-40: word + position + token type, five kernels) as one pass; the sums in that order,
// each rounded to fp32 - the values of the separate kernels.  `i % n`: an id tensor that is broadcast over leading axes.
struct GatherSum3 {
    const float* table[3];
    const void*  ids[3];
    int64_t      n[3], rows[3];
};
template <typename IdT>
__global__ void __launch_bounds__(256) gather_sum3_rows(GatherSum3 a, float* __restrict__ out, int64_t n_out, int64_t row_len, int* status) {
    const int64_t total = n_out * row_len;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / row_len, c = e - i * row_len;
        float v[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t r = int64_t(static_cast<const IdT*>(a.ids[k])[i % a.n[k]]);
            if (r < 0) r += a.rows[k];
            const bool in = r >= 0 && r < a.rows[k];
            v[k] = in ? a.table[k][r * row_len + c] : __builtin_nanf("");
            ok = ok && in;
        }
        if (!ok) __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[e] = (v[0] + v[1]) + v[2];
    }
}

template <typename IdT>
__global__ void __launch_bounds__(256) scatter_add_rows(const float* __restrict__ grad_out, const IdT* __restrict__ ids,
                                                        float* __restrict__ grad_table, int64_t n_ids, int64_t row_len, int64_t table_rows,
                                                        int* status) {
    const int64_t total = n_ids * row_len;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / row_len, c = e - i * row_len;
        int64_t r = int64_t(ids[i]);
        if (r < 0) r += table_rows;
        if (r >= 0 && r < table_rows) atomicAdd(grad_table + r * row_len + c, grad_out[e]);   // global_atomic_add_f32, agent scope
        else __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <typename IdT>
__global__ void __launch_bounds__(256) scatter_add_rows_chunked(const float* __restrict__ grad_out, const IdT* __restrict__ ids,
                                                                float* __restrict__ grad_table, int64_t n_ids, int64_t row_len,
                                                                int64_t table_rows, int* status) {
    scatter_add_rows_chunked_body<IdT>(grad_out, ids, grad_table, n_ids, row_len, table_rows, status, blockIdx.x);
}

__global__ void __launch_bounds__(256) param_grads_tail_group(TailGroup grp) {
    tail_group_body(grp, int(blockIdx.x));
}

}  // namespace lg

using namespace lg;

namespace lg {
struct LnGroupState {
    int        count = 0;          // LayerNorm entries queued
    int        n_scatter = 0;      // embedding scatter-adds queued
    TailGroup  grp;
    int64_t    tickets = 0;
};
static LnGroupState& ln_group() { static LnGroupState s; return s; }

// does a queued LayerNorm parameter gradient or embedding scatter write (into) `ptr`?  A GEMM product that arrives later for the same
// buffer must not be queued in front of it (the GEMM queue leaves first): gemm.hip flushes everything, in order, before it queues
bool ln_group_writes(const void* ptr) {
    LnGroupState& S = ln_group();
    if (ptr == nullptr) return false;
    for (int i = 0; i < S.count; ++i)
        if (S.grp.ln.e[i].dw == ptr || S.grp.ln.e[i].db == ptr) return true;
    for (int i = 0; i < S.n_scatter; ++i)
        if (S.grp.sc[i].table == ptr) return true;
    return false;
}

// the queued jobs as one argument block with the workgroup ranges of its entries; false: nothing is queued
static bool tail_prepare(LnGroupState& S) {
    if (S.count == 0 && S.n_scatter == 0) return false;
    S.grp.ln.count = S.count;
    S.grp.n_scatter = S.n_scatter;
    S.grp.status = rt().status_dev;
    int64_t at = 0;
    for (int i = 0; i < S.count; ++i) { S.grp.first[i] = int(at); at += int64_t(S.grp.ln.e[i].blocks_x) * S.grp.ln.e[i].splits; }
    for (int i = 0; i < S.n_scatter; ++i) { S.grp.first[S.count + i] = int(at); at += S.grp.sc[i].n_ids; }
    S.grp.first[S.count + S.n_scatter] = int(at);
    return true;
}

static int tail_done(LnGroupState& S) {
    int rc = LG_OK;
    for (int i = 0; i < S.count; ++i)
        if (S.grp.ln.e[i].partial) { const int r = lg_free(S.grp.ln.e[i].partial); if (r != LG_OK) rc = r; }       // stream-ordered
    S.count = 0;
    S.n_scatter = 0;
    S.tickets = 0;
    return rc;
}

// gemm.hip, when it launches its queue of products: the queued jobs ride in that launch.  tail_take hands out the argument
// block (false: nothing queued); the caller launches it and then calls tail_taken.
bool tail_take(TailGroup* out) {
    LnGroupState& S = ln_group();
    if (!tail_prepare(S)) return false;
    *out = S.grp;
    return true;
}
int tail_taken() { return tail_done(ln_group()); }

int ln_group_flush_pending() {
    LnGroupState& S = ln_group();
    if (!tail_prepare(S)) return LG_OK;
    if (S.count == 1 && S.n_scatter == 0) {
        const LnParamGrads& a = S.grp.ln.e[0];
        hipLaunchKernelGGL(layernorm_param_grads, dim3(unsigned(a.blocks_x), unsigned(a.splits)), dim3(256), 0, rt().stream, a);
    } else {
        hipLaunchKernelGGL(param_grads_tail_group, dim3(unsigned(S.grp.first[S.count + S.n_scatter])), dim3(256), 0, rt().stream, S.grp);
    }
    return tail_done(S);
}
}  // namespace lg

extern "C" int lg_softmax_scaled_f32(const float* x, float* y, int64_t rows, int64_t cols, float scale) {
    LG_REQUIRE_INIT();
    LG_ARG(rows >= 0 && cols >= 1, "lg_softmax_f32: bad shape (%lld, %lld)", (long long)rows, (long long)cols);
    if (rows == 0) return LG_OK;
    LG_ARG(x && y, "lg_softmax_f32: NULL pointer");
    const unsigned blocks = unsigned((rows + 3) / 4);
    hipStream_t s = rt().stream;
    if (cols <= 128)       hipLaunchKernelGGL(softmax_fwd<2>, dim3(blocks), dim3(256), 0, s, x, y, rows, cols, scale);
    else if (cols <= 512)  hipLaunchKernelGGL(softmax_fwd<8>, dim3(blocks), dim3(256), 0, s, x, y, rows, cols, scale);
    else if (cols <= 2048) hipLaunchKernelGGL(softmax_fwd<kRowRegs>, dim3(blocks), dim3(256), 0, s, x, y, rows, cols, scale);
    else                   hipLaunchKernelGGL(softmax_fwd<0>, dim3(blocks), dim3(256), 0, s, x, y, rows, cols, scale);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_softmax_f32(const float* x, float* y, int64_t rows, int64_t cols) {
    return lg_softmax_scaled_f32(x, y, rows, cols, 1.0f);          // x * 1.0f is x, bit for bit
}

extern "C" int lg_softmax_scaled_bwd_f32(const float* y, const float* g, float* dx, int64_t rows, int64_t cols, float scale) {
    LG_REQUIRE_INIT();
    LG_ARG(rows >= 0 && cols >= 1, "lg_softmax_bwd_f32: bad shape");
    if (rows == 0) return LG_OK;
    LG_ARG(y && g && dx, "lg_softmax_bwd_f32: NULL pointer");
    hipLaunchKernelGGL(softmax_bwd, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, rt().stream, y, g, dx, rows, cols, scale);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_softmax_bwd_f32(const float* y, const float* g, float* dx, int64_t rows, int64_t cols) {
    return lg_softmax_scaled_bwd_f32(y, g, dx, rows, cols, 1.0f);
}

extern "C" int lg_layernorm_f32(const float* x, const float* w, const float* b, float* y, float* xhat, float* rstd,
                                int64_t rows, int64_t cols, double eps) {
    LG_REQUIRE_INIT();
    LG_ARG(rows >= 0 && cols >= 1, "lg_layernorm_f32: bad shape");
    if (rows == 0) return LG_OK;
    LG_ARG(x && w && b && y && xhat && rstd, "lg_layernorm_f32: NULL pointer");
    hipLaunchKernelGGL(layernorm_fwd, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, rt().stream, x, w, b, y, xhat, rstd, rows, cols,
                       float(eps), float(1.0 / double(cols)));
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_layernorm_bwd_f32(const float* g, const float* w, const float* xhat, const float* rstd, float* dx,
                                    int64_t rows, int64_t cols) {
    LG_REQUIRE_INIT();
    LG_ARG(rows >= 0 && cols >= 1, "lg_layernorm_bwd_f32: bad shape");
    if (rows == 0) return LG_OK;
    LG_ARG(g && w && xhat && rstd && dx, "lg_layernorm_bwd_f32: NULL pointer");
    hipLaunchKernelGGL(layernorm_bwd, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, rt().stream, g, w, xhat, rstd, dx, rows, cols,
                       float(1.0 / double(cols)));
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_gather_rows_f32(const float* table, const void* ids, int id_itemsize, float* out, int64_t n_ids, int64_t row_len,
                                  int64_t table_rows) {
    LG_REQUIRE_INIT();
    LG_ARG(id_itemsize == 4 || id_itemsize == 8, "lg_gather_rows_f32: ids must be int32 or int64");
    LG_ARG(n_ids >= 0 && row_len >= 0 && table_rows >= 0, "lg_gather_rows_f32: bad shape");
    if (n_ids == 0 || row_len == 0) return LG_OK;
    LG_ARG(table && ids && out, "lg_gather_rows_f32: NULL pointer");
    const unsigned grid = stream_grid(n_ids * row_len);
    if (id_itemsize == 4)
        hipLaunchKernelGGL(gather_rows<int32_t>, dim3(grid), dim3(256), 0, rt().stream, table, static_cast<const int32_t*>(ids), out, n_ids, row_len, table_rows, rt().status_dev);
    else
        hipLaunchKernelGGL(gather_rows<int64_t>, dim3(grid), dim3(256), 0, rt().stream, table, static_cast<const int64_t*>(ids), out, n_ids, row_len, table_rows, rt().status_dev);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_gather_sum3_rows_f32(const float* t0, const void* ids0, int64_t n0, int64_t rows0,
                                       const float* t1, const void* ids1, int64_t n1, int64_t rows1,
                                       const float* t2, const void* ids2, int64_t n2, int64_t rows2,
                                       int id_itemsize, float* out, int64_t n_out, int64_t row_len) {
    LG_REQUIRE_INIT();
    LG_ARG(id_itemsize == 4 || id_itemsize == 8, "lg_gather_sum3_rows_f32: ids must be int32 or int64");
    LG_ARG(n_out >= 0 && row_len >= 0 && rows0 >= 0 && rows1 >= 0 && rows2 >= 0, "lg_gather_sum3_rows_f32: bad shape");
    if (n_out == 0 || row_len == 0) return LG_OK;
    LG_ARG(t0 && t1 && t2 && ids0 && ids1 && ids2 && out, "lg_gather_sum3_rows_f32: NULL pointer");
    LG_ARG(n0 >= 1 && n1 >= 1 && n2 >= 1 && n_out % n0 == 0 && n_out % n1 == 0 && n_out % n2 == 0,
           "lg_gather_sum3_rows_f32: every id count must divide the %lld output rows", (long long)n_out);
    GatherSum3 a{{t0, t1, t2}, {ids0, ids1, ids2}, {n0, n1, n2}, {rows0, rows1, rows2}};
    const unsigned grid = stream_grid(n_out * row_len);
    if (id_itemsize == 4) hipLaunchKernelGGL(gather_sum3_rows<int32_t>, dim3(grid), dim3(256), 0, rt().stream, a, out, n_out, row_len, rt().status_dev);
    else                  hipLaunchKernelGGL(gather_sum3_rows<int64_t>, dim3(grid), dim3(256), 0, rt().stream, a, out, n_out, row_len, rt().status_dev);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_scatter_add_rows_f32(const float* grad_out, const void* ids, int id_itemsize, float* grad_table, int64_t n_ids,
                                       int64_t row_len, int64_t table_rows) {
    LG_REQUIRE_INIT();
    LG_ARG(id_itemsize == 4 || id_itemsize == 8, "lg_scatter_add_rows_f32: ids must be int32 or int64");
    LG_ARG(n_ids >= 0 && row_len >= 0 && table_rows >= 0, "lg_scatter_add_rows_f32: bad shape");
    if (n_ids == 0 || row_len == 0) return LG_OK;
    LG_ARG(grad_out && ids && grad_table, "lg_scatter_add_rows_f32: NULL pointer");
    static const char* owner_env = getenv("LG_SCATTER_OWNER");        // experiments only: 0 = atomics for every size
    if (n_ids <= 4096 && !(owner_env && atoi(owner_env) == 0) && gemm_group_is_open()) {
        // inside a gradient group bracket (lg_gemm_group_begin): queued, and launched with the LayerNorm parameter gradients -
        // the embedding tables' gradients are the last kernels of a backward pass, three launches of ~5 us of work each
        LnGroupState& S = ln_group();
        bool clash = false;
        for (int i = 0; i < S.n_scatter; ++i) clash = clash || S.grp.sc[i].table == grad_table;        // tied tables: in call order
        if (clash || S.n_scatter == kScatterGroupMax) {
            // an early flush sends the queued GEMM products FIRST: one of them may still have to OVERWRITE the very buffer a queued
            // scatter adds into (a tied embedding / decoder weight after zero_grad: dW comes with beta = 0) - call order is kept
            const int rc = gemm_group_flush_pending();
            if (rc != LG_OK) return rc;
        }
        ScatterJob& j = S.grp.sc[S.n_scatter++];
        j.grad_out = grad_out; j.ids = ids; j.table = grad_table;
        j.n_ids = n_ids; j.row_len = row_len; j.table_rows = table_rows; j.id_itemsize = id_itemsize;
        return LG_OK;
    }
    if (n_ids <= 4096 && !(owner_env && atoi(owner_env) == 0)) {
        // the ids of one batch: chunks of 32 positions per id are summed in position order, few or no atomics (see the kernel)
        if (id_itemsize == 4)
            hipLaunchKernelGGL(scatter_add_rows_chunked<int32_t>, dim3(unsigned(n_ids)), dim3(256), 0, rt().stream, grad_out, static_cast<const int32_t*>(ids), grad_table, n_ids, row_len, table_rows, rt().status_dev);
        else
            hipLaunchKernelGGL(scatter_add_rows_chunked<int64_t>, dim3(unsigned(n_ids)), dim3(256), 0, rt().stream, grad_out, static_cast<const int64_t*>(ids), grad_table, n_ids, row_len, table_rows, rt().status_dev);
        LG_CHECK_LAUNCH();
        return LG_OK;
    }
    const unsigned grid = stream_grid(n_ids * row_len);
    if (id_itemsize == 4)
        hipLaunchKernelGGL(scatter_add_rows<int32_t>, dim3(grid), dim3(256), 0, rt().stream, grad_out, static_cast<const int32_t*>(ids), grad_table, n_ids, row_len, table_rows, rt().status_dev);
    else
        hipLaunchKernelGGL(scatter_add_rows<int64_t>, dim3(grid), dim3(256), 0, rt().stream, grad_out, static_cast<const int64_t*>(ids), grad_table, n_ids, row_len, table_rows, rt().status_dev);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

static int cross_entropy_impl(const float* logits, const void* labels, int label_itemsize, float* dlogits, float* nll,
                              int64_t rows, int64_t cols, float* mean_out, bool* mean_done) {
    *mean_done = false;
    LG_REQUIRE_INIT();
    LG_ARG(label_itemsize == 2 || label_itemsize == 4 || label_itemsize == 8, "lg_cross_entropy_f32: labels must be int16/int32/int64");
    LG_ARG(rows >= 0 && cols >= 1, "lg_cross_entropy_f32: bad shape");
    if (rows == 0) return LG_OK;
    LG_ARG(logits && labels && dlogits && nll, "lg_cross_entropy_f32: NULL pointer");
    const dim3 grid(unsigned((rows + 3) / 4)), block(256);
    const float inv_rows = float(1.0 / double(rows));
    hipStream_t s = rt().stream;
    static const char* ce_env = getenv("LG_CE_HELD");        // experiments only: 0 = the two-pass kernel for every width
    if (cols >= 4096 && cols + 31 <= 512 * 60 && rows < (int64_t(1) << 31) && !(ce_env && atoi(ce_env) == 0)) {
        // a vocabulary per row that fits one workgroup's registers: a single pass over memory
        if (label_itemsize == 2)      launch_cross_entropy_held(logits, static_cast<const int16_t*>(labels), dlogits, nll, rows, cols, inv_rows, mean_out);
        else if (label_itemsize == 4) launch_cross_entropy_held(logits, static_cast<const int32_t*>(labels), dlogits, nll, rows, cols, inv_rows, mean_out);
        else                          launch_cross_entropy_held(logits, static_cast<const int64_t*>(labels), dlogits, nll, rows, cols, inv_rows, mean_out);
        *mean_done = mean_out != nullptr;
        LG_CHECK_LAUNCH();
        return LG_OK;
    }
    if (cols >= 4096 && rows < (int64_t(1) << 31)) {       // wider still: one workgroup per row, two passes
        const dim3 wgrid{unsigned(rows)};
        if (label_itemsize == 2)
            hipLaunchKernelGGL(cross_entropy_wide<int16_t>, wgrid, block, 0, s, logits, static_cast<const int16_t*>(labels), dlogits, nll, cols, inv_rows, rt().status_dev);
        else if (label_itemsize == 4)
            hipLaunchKernelGGL(cross_entropy_wide<int32_t>, wgrid, block, 0, s, logits, static_cast<const int32_t*>(labels), dlogits, nll, cols, inv_rows, rt().status_dev);
        else
            hipLaunchKernelGGL(cross_entropy_wide<int64_t>, wgrid, block, 0, s, logits, static_cast<const int64_t*>(labels), dlogits, nll, cols, inv_rows, rt().status_dev);
        LG_CHECK_LAUNCH();
        return LG_OK;
    }
    if (label_itemsize == 2)
        hipLaunchKernelGGL(cross_entropy_rows<int16_t>, grid, block, 0, s, logits, static_cast<const int16_t*>(labels), dlogits, nll, rows, cols, inv_rows, rt().status_dev);
    else if (label_itemsize == 4)
        hipLaunchKernelGGL(cross_entropy_rows<int32_t>, grid, block, 0, s, logits, static_cast<const int32_t*>(labels), dlogits, nll, rows, cols, inv_rows, rt().status_dev);
    else
        hipLaunchKernelGGL(cross_entropy_rows<int64_t>, grid, block, 0, s, logits, static_cast<const int64_t*>(labels), dlogits, nll, rows, cols, inv_rows, rt().status_dev);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_cross_entropy_f32(const float* logits, const void* labels, int label_itemsize, float* dlogits, float* nll,
                                    int64_t rows, int64_t cols) {
    bool done = false;
    return cross_entropy_impl(logits, labels, label_itemsize, dlogits, nll, rows, cols, nullptr, &done);
}

extern "C" int lg_cross_entropy_mean_f32(const float* logits, const void* labels, int label_itemsize, float* dlogits, float* nll,
                                         float* mean, int64_t rows, int64_t cols) {
    LG_ARG(mean != nullptr && rows >= 1, "lg_cross_entropy_mean_f32: needs rows >= 1 and a place for the mean");
    bool done = false;
    int rc = cross_entropy_impl(logits, labels, label_itemsize, dlogits, nll, rows, cols, mean, &done);
    if (rc != LG_OK || done) return rc;
    // the kernels for narrow / very wide rows leave the mean to the generic reduction: sum, then * (1 / rows)
    const int64_t shape[1] = {rows}, strides[1] = {1};
    rc = lg_reduce(LG_RED_SUM, 1, shape, nll, strides, 1u, mean);
    if (rc != LG_OK) return rc;
    const int64_t none[1] = {1};
    return lg_ew(LG_EW_MUL, 0, none, mean, none, nullptr, nullptr, mean, none, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                 float(1.0 / double(rows)));                       // mean = mean * (1 / rows): the scalar is operand b
}

extern "C" int lg_layernorm_param_grads_f32(const float* g, const float* xhat, float* dw, float* db, int64_t rows, int64_t cols,
                                            int dw_accumulate, int db_accumulate) {
    LG_REQUIRE_INIT();
    LG_ARG(rows >= 0 && cols >= 1, "lg_layernorm_param_grads_f32: bad shape");
    LG_ARG(g && xhat && dw && db, "lg_layernorm_param_grads_f32: NULL pointer");
    const int64_t blocks_x = (cols + 255) / 256;
    LG_ARG(blocks_x < (int64_t(1) << 31), "lg_layernorm_param_grads_f32: too many columns");
    int64_t splits = 1;
    if (rows >= 64) {
        splits = 512 / blocks_x;                               // enough workgroups to fill the chip ...
        if (splits * 16 > rows) splits = rows / 16;            // ... with at least 16 rows each
        if (splits > 32) splits = 32;                          // ... and a short fold
        if (splits < 1) splits = 1;
        if (blocks_x > rt().n_gemm_tickets / 2) splits = 1;
    }
    const int64_t chunk = rows > 0 ? (rows + splits - 1) / splits : 1;
    splits = rows > 0 ? (rows + chunk - 1) / chunk : 1;
    LnParamGrads a{};
    a.g = g; a.xhat = xhat; a.dw = dw; a.db = db;
    a.rows = rows; a.cols = cols; a.chunk = chunk;
    a.blocks_x = int(blocks_x); a.splits = int(splits);
    a.acc_w = dw_accumulate; a.acc_b = db_accumulate;
    if (splits > 1) {
        int rc = lg_malloc(reinterpret_cast<void**>(&a.partial), size_t(splits * 2 * cols) * sizeof(float));
        if (rc != LG_OK) return rc;
    }
    // While a gradient group is open (lg_gemm_group_begin) the launch is queued with the group's other work: the parameter
    // gradients of all LayerNorms of a backward pass go out as ONE launch (7.5 us each alone, most of it launch + hand-off).
    // Tickets: the upper half of the array, the GEMM group counts from the bottom.
    LnGroupState& S = ln_group();
    const bool queue = gemm_group_is_open() && blocks_x <= 64 && rows > 0;
    if (queue) {
        bool clash = false;
        for (int i = 0; i < S.count; ++i) clash = clash || S.grp.ln.e[i].dw == dw || S.grp.ln.e[i].db == db;
        if (clash || S.count == kLnGroupMax || S.tickets + blocks_x > rt().n_gemm_tickets / 2 - 1) {      // (the very last slot: the loss kernel's)
            const int rc = gemm_group_flush_pending();      // queued GEMM products first, then this group: call order (see lg_scatter_add_rows_f32)
            if (rc != LG_OK) return rc;
        }
        a.tickets = rt().gemm_tickets + rt().n_gemm_tickets / 2 + S.tickets;
        S.tickets += blocks_x;
        S.grp.ln.e[S.count++] = a;
        return LG_OK;
    }
    a.tickets = rt().gemm_tickets + rt().n_gemm_tickets / 2;
    hipLaunchKernelGGL(layernorm_param_grads, dim3(unsigned(blocks_x), unsigned(splits)), dim3(256), 0, rt().stream, a);
    LG_CHECK_LAUNCH();
    return splits > 1 ? lg_free(a.partial) : LG_OK;
}
