// Jobs that leave at the END of a deep backward pass next to the queued weight-gradient products (lg_gemm_group_*): the
// parameter gradients of LayerNorms and the scatter-adds of embedding tables.  Device code and argument blocks, shared by
// rowwise.hip (their own launch) and gemm.hip (as extra workgroups of the group's GEMM launch: memory- and latency-bound work
// beside MFMA-bound work, 34 us of tiny-BERT's backward that no longer wait for the 102 us in front of them).
#pragma once
#include "common.h"

namespace lg {

// ---- LayerNorm parameter gradients: dw[c] = sum_r g[r][c] * xhat[r][c], db[c] = sum_r g[r][c] -------------------
// one thread per column (coalesced along the row), the rows split over blockIdx.y; with more than one split the
// partial sums are published write-through, a per-column-block ticket elects the last workgroup, which folds them in
// split order (cdna_hip_programming.md, in-launch split-K recipe).  Replaces mul + two column sums + two adds.
struct LnParamGrads {
    const float* g;
    const float* xhat;
    float*       dw;
    float*       db;
    float*       partial;      // [splits][2][cols] when splits > 1
    int*         tickets;      // one per column block
    int64_t      rows, cols, chunk;
    int          blocks_x, splits;
    int          acc_w, acc_b;
};

__device__ __forceinline__ void layernorm_param_grads_body(const LnParamGrads& a, int block_x, int64_t split) {
    const float* __restrict__ g = a.g;
    const float* __restrict__ xhat = a.xhat;
    float* dw = a.dw;
    float* db = a.db;
    float* partial = a.partial;
    const int64_t rows = a.rows, cols = a.cols, chunk = a.chunk, splits = a.splits;
    const int acc_w = a.acc_w, acc_b = a.acc_b;
    const int64_t c_raw = int64_t(block_x) * 256 + threadIdx.x;
    const bool live = c_raw < cols;
    const int64_t c = live ? c_raw : cols - 1;
    const int64_t r0 = split * chunk;
    const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float sw[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gv = g[(r + e) * cols + c], hv = xhat[(r + e) * cols + c];
            sw[e] += gv * hv;
            sb[e] += gv;
        }
    }
    for (; r < r1; ++r) {
        const float gv = g[r * cols + c];
        sw[0] += gv * xhat[r * cols + c];
        sb[0] += gv;
    }
    float vw = (sw[0] + sw[1]) + (sw[2] + sw[3]), vb = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    if (splits > 1) {
        __hip_atomic_store(partial + (split * 2 + 0) * cols + c, vw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(partial + (split * 2 + 1) * cols + c, vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int arrived_last;
        if (threadIdx.x == 0) {
            int* ticket = a.tickets + block_x;
            const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = order == int(splits) - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived_last = last;
        }
        __syncthreads();
        if (!arrived_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        vw = vb = 0.f;
        int64_t s0 = 0;
        for (; s0 + 3 < splits; s0 += 4) {
            float xw[4], xb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xw[e] = __hip_atomic_load(partial + ((s0 + e) * 2 + 0) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xb[e] = __hip_atomic_load(partial + ((s0 + e) * 2 + 1) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { vw += xw[e]; vb += xb[e]; }
        }
        for (; s0 < splits; ++s0) {
            vw += __hip_atomic_load(partial + (s0 * 2 + 0) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vb += __hip_atomic_load(partial + (s0 * 2 + 1) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (live) {
        dw[c] = acc_w ? dw[c] + vw : vw;
        db[c] = acc_b ? db[c] + vb : vb;
    }
}

// the parameter gradients of several LayerNorms in one launch (queued while lg_gemm_group_* is open, see lg_layernorm_param_grads_f32)
constexpr int kLnGroupMax = 8;
struct LnParamGradsGroup {
    LnParamGrads e[kLnGroupMax];
    int          count;
};
// The same for FEW ids (the lookup of one batch): one workgroup per id position i.  Every workgroup looks through all ids and
// learns its RANK among the positions that hold the same id, and how many there are.  The positions of an id are cut into
// chunks of kChunk consecutive ranks; the first position of a chunk (rank % kChunk == 0) is its leader: it adds, in position
// order, its own row of grad_out and those of the up to kChunk - 1 positions that follow it; the other positions have nothing
// to do.  An id that occurs at most kChunk times has ONE leader, which adds straight onto the table row - no atomics, and the
// order of np.add.at (reference cpu/ops.py:242-246): numpy's bits.  A hotter id (a padding token, a token-type id: a thousand
// times the same row) has several leaders, each adding its chunk's partial sum atomically - a thousand contributions to one
// row cost 32 atomics per element instead of 1024 (tools/scatter_bench.py: 28.4 -> 14.5 us; ids without repeats 6 us either way).
constexpr int kChunk = 32;
template <typename IdT>
__device__ __forceinline__ void scatter_add_rows_chunked_body(const float* __restrict__ grad_out, const IdT* __restrict__ ids,
                                                              float* __restrict__ grad_table, int64_t n_ids, int64_t row_len,
                                                              int64_t table_rows, int* status, int64_t i) {
    __shared__ int before_total, same_total, found;
    __shared__ int wave_matches[4];
    __shared__ int later[kChunk];                    // the positions of this chunk behind the leader, ascending
    const int tid = threadIdx.x;
    int64_t mine = int64_t(ids[i]);
    if (mine < 0) mine += table_rows;
    if (mine < 0 || mine >= table_rows) {
        if (tid == 0) __hip_atomic_fetch_or(status, LG_STATUS_BAD_INDEX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (tid == 0) { before_total = 0; same_total = 0; found = 0; }
    __syncthreads();
    int before = 0, same = 0;
    for (int64_t j = tid; j < n_ids; j += 256) {
        int64_t r = int64_t(ids[j]);
        if (r < 0) r += table_rows;
        if (r == mine) { ++same; before += (j < i); }
    }
    if (same) { atomicAdd(&same_total, same); if (before) atomicAdd(&before_total, before); }       // LDS counters
    __syncthreads();
    const int rank = before_total, count = same_total;
    if (rank % kChunk != 0) return;                                   // not a chunk leader
    const int want = (count - rank - 1 < kChunk - 1) ? count - rank - 1 : kChunk - 1;      // followers in this chunk
    // the next `want` positions with the same id, in ascending order: 256 positions at a time, ranked by ballots
    for (int64_t base = i + 1; base < n_ids; base += 256) {
        if (found >= want) break;                                     // uniform: read after the barrier below / the one above
        const int64_t j = base + tid;
        bool hit = false;
        if (j < n_ids) {
            int64_t r = int64_t(ids[j]);
            if (r < 0) r += table_rows;
            hit = (r == mine);
        }
        const unsigned long long ballot = __ballot(hit);
        if ((tid & 63) == 0) wave_matches[tid >> 6] = __popcll(ballot);
        __syncthreads();
        int slot = found + __popcll(ballot & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) slot += wave_matches[w];
        if (hit && slot < want) later[slot] = int(j - i);             // offsets fit an int: n_ids <= 4096
        __syncthreads();
        if (tid == 0) found += wave_matches[0] + wave_matches[1] + wave_matches[2] + wave_matches[3];
        __syncthreads();
    }
    float* row = grad_table + mine * row_len;
    const bool alone = count <= kChunk;                               // the only leader of this id: plain, ordered update
    for (int64_t c = tid; c < row_len; c += 256) {
        float acc = alone ? row[c] + grad_out[i * row_len + c] : grad_out[i * row_len + c];
        for (int k = 0; k < want; ++k) acc += grad_out[(i + later[k]) * row_len + c];
        if (alone) row[c] = acc;
        else atomicAdd(row + c, acc);
    }
}

// embedding gradients queued next to the LayerNorm parameter gradients (lg_gemm_group_*): they leave in the same launch
constexpr int kScatterGroupMax = 4;
struct ScatterJob {
    const float* grad_out;
    const void*  ids;
    float*       table;
    int64_t      n_ids, row_len, table_rows;
    int          id_itemsize;
};
struct TailGroup {
    LnParamGradsGroup ln;
    ScatterJob        sc[kScatterGroupMax];
    int               n_scatter;
    int*              status;
    int               first[kLnGroupMax + kScatterGroupMax + 1];    // first workgroup of each entry (LayerNorm entries, then scatter jobs); the last: all
};

// workgroup t of the jobs: entry e owns [first[e], first[e + 1]) - blocks_x * splits workgroups for a LayerNorm entry (x fastest),
// one per id position for a scatter job.  (A padded 3-D grid - every entry as large as the largest - spent 20 of the launch's
// 34 us on tiny-BERT dispatching 7 000 workgroups that return at once.)
__device__ __forceinline__ void tail_group_body(const TailGroup& grp, int t) {
    const int entries = grp.ln.count + grp.n_scatter;
    int e = 0;
    while (e + 1 < entries && t >= grp.first[e + 1]) ++e;              // uniform: scalar loads
    const int local = t - grp.first[e];
    if (e < grp.ln.count) {
        const LnParamGrads& a = grp.ln.e[e];
        layernorm_param_grads_body(a, local % a.blocks_x, local / a.blocks_x);
        return;
    }
    const ScatterJob& j = grp.sc[e - grp.ln.count];
    const int64_t i = local;
    if (i >= j.n_ids) return;
    if (j.id_itemsize == 4)
        scatter_add_rows_chunked_body<int32_t>(j.grad_out, static_cast<const int32_t*>(j.ids), j.table, j.n_ids, j.row_len, j.table_rows, grp.status, i);
    else
        scatter_add_rows_chunked_body<int64_t>(j.grad_out, static_cast<const int64_t*>(j.ids), j.table, j.n_ids, j.row_len, j.table_rows, grp.status, i);
}

}  // namespace lg
