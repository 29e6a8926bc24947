// Self-attention of a short sequence in one launch each way (gfx950, fp32 MFMA).
//
//   forward    P = softmax((Q K^T) * scale)   O = P V          per (batch, head), reference examples/bert.py:78-88
//   backward   dV = P^T dO   dP = dO V^T   dS = P o (dP - shift) * scale   dQ = dS K   dK = dS^T Q
//
// The composite form is five launches forward (three Linear products apart: scores GEMM, scaled softmax, context GEMM) and
// three backward (two two-product launches and the softmax backward), each a few microseconds of work behind a launch
// floor of its own; at tiny-BERT's size (S = 128, D = 64, 16 (batch, head) pairs) everything one (batch, head) pair needs
// fits the LDS of one CU.  The scores never reach HBM; the probabilities do, once: the model returns them
// (bert.py:88) and the backward reads them instead of recomputing the softmax.
//
// Work split.  Forward: one workgroup per (batch, head, block of 32 queries).  Backward: two roles, one workgroup each per
// (batch, head, block of 32): the QUERY role makes dQ of its 32 queries, the KEY role dK and dV of its 32 keys.  The key role
// needs dS[:, block] for ALL queries: dP only for its own keys (32 MFMAs per wave) but the softmax shift of every row, a sum
// over the whole row of dP.  The query-role workgroups form those sums anyway and hand them over inside the launch: each
// publishes its 32 shifts (write-through doubles) and raises a counter of its (batch, head) pair; a key-role workgroup waits
// for the counter to reach S / 32.  The query-role workgroups fill the front of the grid, so whoever waits, waits for
// workgroups that are already running and wait for nothing; the wait is bounded (2 s, then the device status flag).  The
// first version recomputed dP for all queries in the key role instead (128 MFMAs per wave and a row pass over S rows:
// 14.8 us for the role against 11.6 us with the hand-off, tools/attn_timeline.py).  Both roles run the same MFMA sequence on the
// same operands, so their dP - and dS - agree bit for bit.  The shift is formed in double exactly like the separate softmax
// backward (rowwise.hip: softmax_bwd explains why).
//
// MFMA operands come from LDS.  v_mfma_f32_32x32x2f32 takes one float per lane: lane (r, h) = (lane & 31, lane >> 5) holds
// A[r][k + h] and B[k + h][r].  Any order of the k values will do as long as A and B agree, so a wave walks K in groups of
// 8 with lane half h on k = 8j + 4h .. 8j + 4h + 3: an operand that is contiguous along K is then ONE ds_read_b128 per
// four MFMAs, the others four ds_read_b32 of 32 consecutive floats.  Row pitches: K-contiguous rows + 4 floats (16 lanes of a
// b128 read hit 16 different 16-byte slots), M/N-contiguous rows = 8 mod 16 floats (the two lane halves, 4 rows apart, land
// on different halves of the 64 banks).
#include "common.h"
#include "mfma_lds.h"
#include <cmath>

namespace lg {

#ifdef LG_GEMM_TIMELINE
// experiments build only (make timeline; tools/attn_timeline.py): 16 timestamps of the 100 MHz wall clock per workgroup
#define LG_ATL(slot) do { if (a.tl && threadIdx.x == 0) a.tl[size_t((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (slot)] = wall_clock64(); } while (0)
#define LG_ATL_FIELD unsigned long long* tl;
#else
#define LG_ATL(slot) do { } while (0)
#define LG_ATL_FIELD
#endif

struct AttnArgs {
    LG_ATL_FIELD
    const float *q, *k, *v;          // element (b, s, head, d) of X at X + b * sbX + s * ldX + head * D + d
    int64_t ldq, sbq, ldk, sbk, ldv, sbv;
    float* o;                        // context, same addressing
    int64_t ldo, sbo;
    float* p;                        // probabilities (batch, heads, S, S), dense
    int S, heads;
    float scale;
};

template <int D>
constexpr int attn_fwd_lds_floats(int S) { return 32 * (D + 4) + S * (D + 4) + S * (D + 8) + 32 * (S + 4) + 3 * 1024; }

template <int D>
__global__ void __launch_bounds__(256) attn_fwd(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S;
    constexpr int PQ = D + 4, PV = D + 8;
    const int PP = S + 4;
    float* Qs = lds;
    float* Ks = Qs + 32 * PQ;
    float* Vs = Ks + S * PQ;
    float* Ps = Vs + S * PV;
    float* Red = Ps + 32 * PP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
    LG_ATL(0);

    {
        constexpr int NB = 32 * (D / 4) / 256 > 0 ? 32 * (D / 4) / 256 : 1, NA = 128 * (D / 4) / 256;   // float4 per thread: 32 rows / all rows
        af32x4 rq[NB], rk[NA], rv[NA];
        load_rows<D, NB>(rq, a.q + int64_t(b) * a.sbq + int64_t(q0) * a.ldq + head * D, a.ldq, 32);
        load_rows<D, NA>(rk, a.k + int64_t(b) * a.sbk + head * D, a.ldk, S);
        load_rows<D, NA>(rv, a.v + int64_t(b) * a.sbv + head * D, a.ldv, S);
        store_rows<D, NB>(rq, Qs, PQ, 32);
        store_rows<D, NA>(rk, Ks, PQ, S);
        store_rows<D, NA>(rv, Vs, PV, S);
    }
    __syncthreads();
    LG_ATL(1);

    // scores of 32 queries against keys [32 wave, 32 wave + 32), scaled (the product rounded to fp32 first, like `scores * c`)
    if (32 * wave < S) {
        af32x16 acc = zero16();
        wave_mma<true, true>(acc, Qs, PQ, Ks + 32 * wave * PQ, PQ, D, r, h);
#pragma unroll
        for (int e = 0; e < 16; ++e) Ps[acc_row(e, h) * PP + 32 * wave + r] = acc[e] * a.scale;
    }
    __syncthreads();
    LG_ATL(2);

    // softmax of each row: 8 threads per row, float4 columns sub, sub + 8, ... held in registers between the three passes;
    // exp(x - max) * (1 / sum) (autograd/ops.py:62-66)
    {
        const int row = tid >> 3, sub = tid & 7;
        float* pr = Ps + row * PP;
        af32x4 t[4];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = sub * 4 + 32 * i;
            if (c < S) {
                t[i] = *reinterpret_cast<const af32x4*>(pr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) m = (t[i][e] > m || t[i][e] != t[i][e]) ? t[i][e] : m;
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) { const float o = __shfl_xor(m, off, 64); m = (o > m || o != o) ? o : m; }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (sub * 4 + 32 * i < S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[i][e] = expf(t[i][e] + (-m)); s += t[i][e]; }
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) s += __shfl_xor(s, off, 64);
        const float inv = 1.0f / s;
        float* pg = a.p + ((int64_t(b) * a.heads + head) * S + q0 + row) * S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = sub * 4 + 32 * i;
            if (c < S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[i][e] *= inv;
                *reinterpret_cast<af32x4*>(pr + c) = t[i];
                *reinterpret_cast<af32x4*>(pg + c) = t[i];
            }
        }
    }
    __syncthreads();
    LG_ATL(3);

    // context = P @ V: D / 32 column tiles, the keys split over the remaining waves, partial sums folded in wave order
    constexpr int NT = D / 32, KP = 4 / NT;
    const int n = wave % NT, kp = wave / NT;
    const int kspan = S / KP;
    af32x16 acc = zero16();
    wave_mma<true, false>(acc, Ps + kp * kspan, PP, Vs + kp * kspan * PV + 32 * n, PV, kspan, r, h);
    LG_ATL(4);
    if (kp > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Red[((kp - 1) * NT + n) * 1024 + e * 64 + lane] = acc[e];
    }
    __syncthreads();
    LG_ATL(5);
    if (kp == 0) {
        for (int q = 1; q < KP; ++q) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += Red[((q - 1) * NT + n) * 1024 + e * 64 + lane];
        }
        float* og = a.o + int64_t(b) * a.sbo + int64_t(q0) * a.ldo + head * D + 32 * n + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) og[int64_t(acc_row(e, h)) * a.ldo] = acc[e];
    }
    LG_ATL(6);
}

struct AttnBwdArgs {
    LG_ATL_FIELD
    const float *q, *k, *v, *g;      // g = gradient of the context; addressing as in AttnArgs
    int64_t ldq, sbq, ldk, sbk, ldv, sbv, ldg, sbg;
    const float* p;                  // probabilities saved by the forward
    float *dq, *dk, *dv;
    int64_t lddq, sbdq, lddk, sbdk, lddv, sbdv;
    int S, heads, batch;
    float scale;
    double* shift;                   // [batch, heads, S]: the softmax shift of every query row, query role -> key role
    int*    flags;                   // [batch * heads][2]: rows published / key-role workgroups served; zero between launches
    int*    status;                  // device status flag (a wait that gives up raises it)
};

// The probabilities a thread needs for its rows: row = (tid >> 3) + 32 * pass, float4 columns (tid & 7) + 8 * i.  Fetched into
// registers ahead of the MFMAs whose result they meet, so the row pass below never waits for HBM.
template <int PASSES>
struct ProbRows { af32x4 y[PASSES][4]; };

template <int PASSES>
__device__ __forceinline__ void load_probs(ProbRows<PASSES>& pr, const float* y, int S, int rows) {
    const int sub = threadIdx.x & 7;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row0 = (threadIdx.x >> 3) + 32 * p, row = row0 < rows ? row0 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c0 = sub * 4 + 32 * i, c = c0 < S ? c0 : 0;
            pr.y[p][i] = *reinterpret_cast<const af32x4*>(y + int64_t(row) * S + c);
        }
    }
}

// dS = float(double(y) * (double(g) - shift)) * scale, shift = sum(g * y) / sum(y) over the row in double, in place, for 32 rows of
// dP held in LDS (pitch pp) against their probabilities y (load_probs); 8 threads per row.  The shift of every row is also
// stored to `shift_out` (write-through: workgroups on other XCDs read it).
__device__ __forceinline__ void softmax_bwd_rows(float* dp, int pp, const ProbRows<1>& pr, int S, float scale, double* shift_out) {
    const int sub = threadIdx.x & 7, row = threadIdx.x >> 3;
    float* gr = dp + row * pp;
    af32x4 g4[4];
    double dot = 0.0, norm = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub * 4 + 32 * i;
        if (c < S) {
            g4[i] = *reinterpret_cast<const af32x4*>(gr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double yc = double(pr.y[0][i][e]); dot += double(g4[i][e]) * yc; norm += yc; }
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) { dot += __shfl_xor(dot, off, 64); norm += __shfl_xor(norm, off, 64); }
    const double shift = dot / norm;
    if (sub == 0) __hip_atomic_store(shift_out + row, shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub * 4 + 32 * i;
        if (c < S) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[i][e] = float(double(pr.y[0][i][e]) * (double(g4[i][e]) - shift)) * scale;
            *reinterpret_cast<af32x4*>(gr + c) = g4[i];
        }
    }
}

// floats of LDS: the larger of the two roles
template <int D>
constexpr int attn_bwd_lds_floats(int S) {
    const int query = 32 * (D + 4) + S * (D + 4) + S * (D + 8) + 32 * (S + 4) + 3 * 1024;
    const int key = 2 * S * (D + 8) + 32 * (D + 4) + 2 * S * 40 + 2048;
    return query > key ? query : key;
}

template <int D>
__global__ void __launch_bounds__(256) attn_bwd(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // the query-role workgroups of the whole grid are dispatched before any key-role one (role in the slowest grid index): a
    // key-role workgroup that waits, waits for workgroups that are already running and wait for nothing
    const int role = int(blockIdx.z) >= a.batch ? 1 : 0;
    const int blk = blockIdx.x, head = blockIdx.y, b = int(blockIdx.z) - role * a.batch;
    const int j0 = blk * 32, nblk = S / 32;
    const int bh = b * a.heads + head;
    const float* pg = a.p + int64_t(bh) * S * S;
    double* shifts = a.shift + int64_t(bh) * S;
    int* flags = a.flags + 2 * bh;
    constexpr int NT = D / 32;
    constexpr int NB = 32 * (D / 4) / 256 > 0 ? 32 * (D / 4) / 256 : 1, NA = 128 * (D / 4) / 256;       // float4 per thread: 32 rows / all rows

    LG_ATL(0);
    if (role == 0) {
        // ---- query role: dQ of queries [j0, j0 + 32), and the shift of their rows for the key role ------------------
        constexpr int PG = D + 4, PVK = D + 4, PK = D + 8;
        const int PP = S + 4;
        float* Gs = lds;                 // dO rows of the block          32 x PG
        float* Vs = Gs + 32 * PG;        // V, K-contiguous B of dP       S x PVK
        float* Ks = Vs + S * PVK;        // K, N-contiguous B of dQ       S x PK
        float* Ss = Ks + S * PK;         // dP, then dS                   32 x PP
        float* Red = Ss + 32 * PP;
        {
            af32x4 rg[NB], rv[NA], rk[NA];
            load_rows<D, NB>(rg, a.g + int64_t(b) * a.sbg + int64_t(j0) * a.ldg + head * D, a.ldg, 32);
            load_rows<D, NA>(rv, a.v + int64_t(b) * a.sbv + head * D, a.ldv, S);
            load_rows<D, NA>(rk, a.k + int64_t(b) * a.sbk + head * D, a.ldk, S);
            store_rows<D, NB>(rg, Gs, PG, 32);
            store_rows<D, NA>(rv, Vs, PVK, S);
            store_rows<D, NA>(rk, Ks, PK, S);
        }
        ProbRows<1> probs;
        load_probs<1>(probs, pg + int64_t(j0) * S, S, 32);
        __syncthreads();
        LG_ATL(1);
        if (32 * wave < S) {
            af32x16 acc = zero16();
            wave_mma<true, true>(acc, Gs, PG, Vs + 32 * wave * PVK, PVK, D, r, h);
#pragma unroll
            for (int e = 0; e < 16; ++e) Ss[acc_row(e, h) * PP + 32 * wave + r] = acc[e];
        }
        __syncthreads();
        LG_ATL(2);
        softmax_bwd_rows(Ss, PP, probs, S, a.scale, shifts + j0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the shifts have left this CU
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // 32 more rows are published
        LG_ATL(3);
        constexpr int KP = 4 / NT;
        const int n = wave % NT, kp = wave / NT;
        const int kspan = S / KP;
        af32x16 acc = zero16();
        wave_mma<true, false>(acc, Ss + kp * kspan, PP, Ks + kp * kspan * PK + 32 * n, PK, kspan, r, h);
        LG_ATL(4);
        if (kp > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Red[((kp - 1) * NT + n) * 1024 + e * 64 + lane] = acc[e];
        }
        __syncthreads();
        LG_ATL(5);
        if (kp == 0) {
            for (int q = 1; q < KP; ++q) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] += Red[((q - 1) * NT + n) * 1024 + e * 64 + lane];
            }
            float* dst = a.dq + int64_t(b) * a.sbdq + int64_t(j0) * a.lddq + head * D + 32 * n + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[int64_t(acc_row(e, h)) * a.lddq] = acc[e];
        }
        LG_ATL(6);
        return;
    }

    // ---- key role: dK and dV of keys [j0, j0 + 32) --------------------------------------------------------------
    // dS[:, block] = P[:, block] o (dP[:, block] - shift) * scale needs dP only for the block's own keys (32 MFMAs per wave, query
    // rows [32 w, 32 w + 32) each) - and the shift of EVERY row, a sum over the whole row of dP that the query-role workgroups of
    // this (batch, head) have formed: they publish it, a counter says how many of them have.  The values of dP agree bit for bit
    // between the roles (the same MFMA sequence over the same operands), so dS does too.
    constexpr int PG = D + 8;            // dO: K-contiguous A of dP (two-way conflicts there) and N-contiguous B of dV
    constexpr int PVK = D + 4, PQN = D + 8, PC = 40;
    float* Gs = lds;                     // dO, all queries               S x PG
    float* Qs = Gs + S * PG;             // Q, N-contiguous B of dK       S x PQN
    float* Vj = Qs + S * PQN;            // V rows of the block           32 x PVK
    float* Pc = Vj + 32 * PVK;           // P[:, block]                   S x PC
    float* Dc = Pc + S * PC;             // dS[:, block]                  S x PC
    float* Red = Dc + S * PC;
    {
        af32x4 rg[NA], rq[NA], rv[NB];
        load_rows<D, NA>(rg, a.g + int64_t(b) * a.sbg + head * D, a.ldg, S);
        load_rows<D, NA>(rq, a.q + int64_t(b) * a.sbq + head * D, a.ldq, S);
        load_rows<D, NB>(rv, a.v + int64_t(b) * a.sbv + int64_t(j0) * a.ldv + head * D, a.ldv, 32);
        store_rows<D, NA>(rg, Gs, PG, S);
        store_rows<D, NA>(rq, Qs, PQN, S);
        store_rows<D, NB>(rv, Vj, PVK, 32);
    }
    const bool active = 32 * wave < S;
    // the probabilities that meet this wave's block of dP: element e of the accumulator is (row 32 w + acc_row(e, h), key j0 + r)
    float y[16];
    if (active) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = pg[int64_t(32 * wave + acc_row(e, h)) * S + j0 + r];
    }
    __syncthreads();
    LG_ATL(1);
    af32x16 dp = zero16();
    if (active) wave_mma<true, true>(dp, Gs + 32 * wave * PG, PG, Vj, PVK, D, r, h);
    LG_ATL(2);
    // wait for the shifts of all S rows
    if (tid == 0) {
        // (bounded: 2 s of the 100 MHz wall clock, then the device status flag is raised and the launch runs to its end)
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) {
                __hip_atomic_fetch_or(a.status, LG_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    LG_ATL(3);
    if (active) {
        double sh[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) sh[e] = __hip_atomic_load(shifts + 32 * wave + acc_row(e, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * wave + acc_row(e, h);
            Dc[row * PC + r] = float(double(y[e]) * (double(dp[e]) - sh[e])) * a.scale;
            Pc[row * PC + r] = y[e];
        }
    }
    __syncthreads();
    // every wave has read its shifts: this workgroup is served; the last one of the (batch, head) pair clears the counters
    if (tid == 0) {
        const int served = __hip_atomic_fetch_add(flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (served == nblk - 1) {
            __hip_atomic_store(flags, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flags + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    LG_ATL(4);
    // dV = P[:, block]^T @ dO and dK = dS[:, block]^T @ Q: 2 * NT output tiles of 32 x 32 over four waves
    constexpr int TILES = 2 * NT, KP = 4 / TILES > 0 ? 4 / TILES : 1;
    const int tile = wave % TILES, kp = wave / TILES;
    const bool is_dk = tile >= NT;
    const int n = tile % NT;
    const int kspan = S / KP;
    af32x16 acc = zero16();
    if (is_dk) wave_mma<false, false>(acc, Dc + kp * kspan * PC, PC, Qs + kp * kspan * PQN + 32 * n, PQN, kspan, r, h);
    else       wave_mma<false, false>(acc, Pc + kp * kspan * PC, PC, Gs + kp * kspan * PG + 32 * n, PG, kspan, r, h);
    LG_ATL(5);
    if constexpr (KP > 1) {
        if (kp > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Red[(kp - 1) * TILES * 1024 + tile * 1024 + e * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (kp > 0) return;
        for (int q = 1; q < KP; ++q) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += Red[(q - 1) * TILES * 1024 + tile * 1024 + e * 64 + lane];
        }
    }
    float* dst = is_dk ? a.dk + int64_t(b) * a.sbdk + int64_t(j0) * a.lddk + head * D + 32 * n + r
                       : a.dv + int64_t(b) * a.sbdv + int64_t(j0) * a.lddv + head * D + 32 * n + r;
    const int64_t ldd = is_dk ? a.lddk : a.lddv;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[int64_t(acc_row(e, h)) * ldd] = acc[e];
    LG_ATL(6);
}

template <class K>
static int allow_lds(K kernel, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e != hipSuccess) { set_error("attention: %zu bytes of LDS refused: %s", bytes, hipGetErrorString(e)); return LG_EHIP; }
    return LG_OK;
}

#ifdef LG_GEMM_TIMELINE
static unsigned long long* g_atl = nullptr;
static int g_atl_wgs = 0;
static unsigned long long* timeline_buffer(int wgs) {
    constexpr int kMax = 4096;
    if (!g_atl && hipMalloc(reinterpret_cast<void**>(&g_atl), size_t(kMax) * 16 * 8) != hipSuccess) return nullptr;
    if (wgs > kMax) return nullptr;
    (void)hipMemsetAsync(g_atl, 0, size_t(wgs) * 16 * 8, rt().stream);
    g_atl_wgs = wgs;
    return g_atl;
}
#endif

static bool ok_operand(const void* p, int64_t ld, int64_t sb) { return p && aligned16(p) && ld % 4 == 0 && sb % 4 == 0; }

}  // namespace lg

using namespace lg;

extern "C" int lg_attention_supported(int64_t S, int64_t D) {
    return (D == 64 || D == 32) && S >= 32 && S <= 128 && S % 32 == 0;
}

extern "C" int lg_attention_fwd_f32(const float* q, int64_t ldq, int64_t sbq, const float* k, int64_t ldk, int64_t sbk,
                                    const float* v, int64_t ldv, int64_t sbv, float* o, int64_t ldo, int64_t sbo, float* p,
                                    int64_t batch, int64_t heads, int64_t S, int64_t D, float scale) {
    LG_REQUIRE_INIT();
    LG_ARG(lg_attention_supported(S, D), "lg_attention_fwd_f32: S = %lld (32..128, multiple of 32), D = %lld (32 or 64) unsupported",
           (long long)S, (long long)D);
    LG_ARG(batch >= 0 && heads >= 1 && batch <= 65535 && heads <= 65535, "lg_attention_fwd_f32: bad batch / heads");
    if (batch == 0) return LG_OK;
    LG_ARG(ok_operand(q, ldq, sbq) && ok_operand(k, ldk, sbk) && ok_operand(v, ldv, sbv) && ok_operand(o, ldo, sbo) && p && aligned16(p),
           "lg_attention_fwd_f32: operands must be 16-byte aligned with pitches that are multiples of 4");
    LG_ARG(ldq >= heads * D && ldk >= heads * D && ldv >= heads * D && ldo >= heads * D, "lg_attention_fwd_f32: row pitch below heads * D");
    AttnArgs a{
#ifdef LG_GEMM_TIMELINE
        timeline_buffer(int(S / 32 * heads * batch)),
#endif
        q, k, v, ldq, sbq, ldk, sbk, ldv, sbv, o, ldo, sbo, p, int(S), int(heads), scale};
    const dim3 grid(unsigned(S / 32), unsigned(heads), unsigned(batch));
    if (D == 64) {
        const size_t bytes = size_t(attn_fwd_lds_floats<64>(int(S))) * 4;
        int rc = allow_lds(&attn_fwd<64>, bytes);
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL(attn_fwd<64>, grid, dim3(256), bytes, rt().stream, a);
    } else {
        const size_t bytes = size_t(attn_fwd_lds_floats<32>(int(S))) * 4;
        int rc = allow_lds(&attn_fwd<32>, bytes);
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL(attn_fwd<32>, grid, dim3(256), bytes, rt().stream, a);
    }
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_attention_bwd_f32(const float* q, int64_t ldq, int64_t sbq, const float* k, int64_t ldk, int64_t sbk,
                                    const float* v, int64_t ldv, int64_t sbv, const float* g, int64_t ldg, int64_t sbg,
                                    const float* p, float* dq, int64_t lddq, int64_t sbdq, float* dk, int64_t lddk, int64_t sbdk,
                                    float* dv, int64_t lddv, int64_t sbdv, int64_t batch, int64_t heads, int64_t S, int64_t D,
                                    float scale) {
    LG_REQUIRE_INIT();
    LG_ARG(lg_attention_supported(S, D), "lg_attention_bwd_f32: S = %lld (32..128, multiple of 32), D = %lld (32 or 64) unsupported",
           (long long)S, (long long)D);
    LG_ARG(batch >= 0 && heads >= 1 && batch <= 65535 && heads <= 65535, "lg_attention_bwd_f32: bad batch / heads");
    if (batch == 0) return LG_OK;
    LG_ARG(ok_operand(q, ldq, sbq) && ok_operand(k, ldk, sbk) && ok_operand(v, ldv, sbv) && ok_operand(g, ldg, sbg) &&
               ok_operand(dq, lddq, sbdq) && ok_operand(dk, lddk, sbdk) && ok_operand(dv, lddv, sbdv) && p && aligned16(p),
           "lg_attention_bwd_f32: operands must be 16-byte aligned with pitches that are multiples of 4");
    const int64_t w = heads * D;
    LG_ARG(ldq >= w && ldk >= w && ldv >= w && ldg >= w && lddq >= w && lddk >= w && lddv >= w, "lg_attention_bwd_f32: row pitch below heads * D");
    LG_ARG(batch * heads <= rt().n_attn_pairs && 2 * batch <= 65535, "lg_attention_bwd_f32: more than %d (batch, head) pairs in one launch",
           rt().n_attn_pairs);
    int* flags = rt().attn_flags;
    double* shift = nullptr;
    {
        const int mrc = lg_malloc(reinterpret_cast<void**>(&shift), size_t(batch * heads * S) * sizeof(double));
        if (mrc != LG_OK) return mrc;
    }
    AttnBwdArgs a{
#ifdef LG_GEMM_TIMELINE
        timeline_buffer(int(2 * (S / 32) * heads * batch)),
#endif
        q, k, v, g, ldq, sbq, ldk, sbk, ldv, sbv, ldg, sbg, p, dq, dk, dv, lddq, sbdq, lddk, sbdk, lddv, sbdv, int(S), int(heads), int(batch), scale,
        shift, flags, rt().status_dev};
    const dim3 grid(unsigned(S / 32), unsigned(heads), unsigned(2 * batch));
    if (D == 64) {
        const size_t bytes = size_t(attn_bwd_lds_floats<64>(int(S))) * 4;
        int rc = allow_lds(&attn_bwd<64>, bytes);
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL(attn_bwd<64>, grid, dim3(256), bytes, rt().stream, a);
    } else {
        const size_t bytes = size_t(attn_bwd_lds_floats<32>(int(S))) * 4;
        int rc = allow_lds(&attn_bwd<32>, bytes);
        if (rc != LG_OK) return rc;
        hipLaunchKernelGGL(attn_bwd<32>, grid, dim3(256), bytes, rt().stream, a);
    }
    LG_CHECK_LAUNCH();
    return lg_free(shift);          // stream-ordered: the block is only reused by later launches
}

#ifdef LG_GEMM_TIMELINE
// experiments build only: the 16 timestamps per workgroup of the LAST attention launch; returns the workgroup count
extern "C" int lg_debug_attn_timeline(unsigned long long* host_out, int max_wgs) {
    if (!g_atl || g_atl_wgs > max_wgs) return -1;
    if (hipStreamSynchronize(rt().stream) != hipSuccess) return -1;
    if (hipMemcpy(host_out, g_atl, size_t(g_atl_wgs) * 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return g_atl_wgs;
}
#endif
