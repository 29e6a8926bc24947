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
// needs dS[:, block] for ALL queries and therefore the softmax shift of every row, a sum over the whole row of dP: it
// recomputes dP for all queries (S x S x D, 128 MFMAs per wave at S = 128) rather than wait for the query-role workgroups -
// no hand-off inside the launch, and both roles run the same code on the same values, so their dS agree bit for bit.
// The shift is formed in double exactly like the separate softmax backward (rowwise.hip: softmax_bwd explains why).
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
    int S, heads;
    float scale;
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

// dS = float(double(y) * (double(g) - shift)) * scale, shift = sum(g * y) / sum(y) over the row in double, for `rows` rows of
// dP held in LDS (pitch pp) against their probabilities y (load_probs).  8 threads per row.
//   WHOLE: dS replaces dP in place.   !WHOLE: only columns [c0, c0 + 32) are kept - dS into dsc, y into pc (pitch pc_pitch).
template <bool WHOLE, int PASSES>
__device__ __forceinline__ void softmax_bwd_rows(float* dp, int pp, const ProbRows<PASSES>& pr, int S, int rows, float scale,
                                                 int c0, float* dsc, float* pc, int pc_pitch) {
    const int sub = threadIdx.x & 7;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = (threadIdx.x >> 3) + 32 * p;
        if (row < rows) {                                        // (all 64 lanes of a wave agree: rows is a multiple of 32)
            float* gr = dp + row * pp;
            af32x4 g4[4];
            double dot = 0.0, norm = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = sub * 4 + 32 * i;
                if (c < S) {
                    g4[i] = *reinterpret_cast<const af32x4*>(gr + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double yc = double(pr.y[p][i][e]); dot += double(g4[i][e]) * yc; norm += yc; }
                }
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) { dot += __shfl_xor(dot, off, 64); norm += __shfl_xor(norm, off, 64); }
            const double shift = dot / norm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = sub * 4 + 32 * i;
                if (c < S) {
                    if constexpr (WHOLE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) g4[i][e] = float(double(pr.y[p][i][e]) * (double(g4[i][e]) - shift)) * scale;
                        *reinterpret_cast<af32x4*>(gr + c) = g4[i];
                    } else if (c >= c0 && c < c0 + 32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dsc[row * pc_pitch + (c - c0) + e] = float(double(pr.y[p][i][e]) * (double(g4[i][e]) - shift)) * scale;
                            pc[row * pc_pitch + (c - c0) + e] = pr.y[p][i][e];
                        }
                    }
                }
            }
        }
    }
}

// floats of LDS: the key role is the larger one
template <int D>
constexpr int attn_bwd_lds_floats(int S) {
    const int query = 32 * (D + 4) + S * (D + 4) + S * (D + 8) + 32 * (S + 4) + 3 * 1024;
    const int big = S * (S + 4) > S * (D + 8) ? S * (S + 4) : S * (D + 8);
    const int key = S * (D + 8) + big + 2 * S * 40 + 2048;
    return query > key ? query : key;
}

template <int D>
__global__ void __launch_bounds__(256) attn_bwd(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int blk = blockIdx.x >> 1, role = blockIdx.x & 1, head = blockIdx.y, b = blockIdx.z;
    const int j0 = blk * 32;
    const float* pg = a.p + (int64_t(b) * a.heads + head) * S * S;
    constexpr int NT = D / 32;
    constexpr int NB = 32 * (D / 4) / 256 > 0 ? 32 * (D / 4) / 256 : 1, NA = 128 * (D / 4) / 256;       // float4 per thread: 32 rows / all rows

    LG_ATL(0);
    if (role == 0) {
        // ---- query role: dQ of queries [j0, j0 + 32) ----------------------------------------------------------
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
        softmax_bwd_rows<true, 1>(Ss, PP, probs, S, 32, a.scale, 0, nullptr, nullptr, 0);
        __syncthreads();
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
    constexpr int PG = D + 8;            // dO: K-contiguous A of dP (two-way conflicts there) and N-contiguous B of dV
    constexpr int PVK = D + 4, PQN = D + 8, PC = 40;
    const int PP = S + 4;
    const int big = S * PP > S * PQN ? S * PP : S * PQN;
    float* Gs = lds;                     // dO, all queries               S x PG
    float* R1 = Gs + S * PG;             // V (S x PVK), then dP (S x PP), then Q (S x PQN)
    float* Pc = R1 + big;                // P[:, block]                   S x PC
    float* Dc = Pc + S * PC;             // dS[:, block]                  S x PC
    float* Red = Dc + S * PC;
    af32x4 rq[NA];                       // Q waits in registers until dP has left R1
    {
        af32x4 rg[NA], rv[NA];
        load_rows<D, NA>(rg, a.g + int64_t(b) * a.sbg + head * D, a.ldg, S);
        load_rows<D, NA>(rv, a.v + int64_t(b) * a.sbv + head * D, a.ldv, S);
        load_rows<D, NA>(rq, a.q + int64_t(b) * a.sbq + head * D, a.ldq, S);
        store_rows<D, NA>(rg, Gs, PG, S);
        store_rows<D, NA>(rv, R1, PVK, S);
    }
    ProbRows<4> probs;
    load_probs<4>(probs, pg, S, S);
    __syncthreads();
    LG_ATL(1);
    // dP for ALL queries: wave w takes query rows [32 w, 32 w + 32) against every key block
    af32x16 dp[4];
    const bool active = 32 * wave < S;
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dp[t] = zero16();
            if (32 * t < S) wave_mma<true, true>(dp[t], Gs + 32 * wave * PG, PG, R1 + 32 * t * PVK, PVK, D, r, h);
        }
    }
    __syncthreads();                     // every wave is done with V
    LG_ATL(2);
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (32 * t < S) {
#pragma unroll
                for (int e = 0; e < 16; ++e) R1[(32 * wave + acc_row(e, h)) * PP + 32 * t + r] = dp[t][e];
            }
        }
    }
    __syncthreads();
    LG_ATL(3);
    softmax_bwd_rows<false, 4>(R1, PP, probs, S, S, a.scale, j0, Dc, Pc, PC);
    __syncthreads();                     // dP is no longer needed: Q takes its place
    LG_ATL(4);
    store_rows<D, NA>(rq, R1, PQN, S);
    __syncthreads();
    LG_ATL(5);
    // dV = P[:, block]^T @ dO and dK = dS[:, block]^T @ Q: 2 * NT output tiles of 32 x 32 over four waves
    constexpr int TILES = 2 * NT, KP = 4 / TILES > 0 ? 4 / TILES : 1;
    const int tile = wave % TILES, kp = wave / TILES;
    const bool is_dk = tile >= NT;
    const int n = tile % NT;
    const int kspan = S / KP;
    af32x16 acc = zero16();
    if (is_dk) wave_mma<false, false>(acc, Dc + kp * kspan * PC, PC, R1 + kp * kspan * PQN + 32 * n, PQN, kspan, r, h);
    else       wave_mma<false, false>(acc, Pc + kp * kspan * PC, PC, Gs + kp * kspan * PG + 32 * n, PG, kspan, r, h);
    LG_ATL(6);
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
    LG_ATL(7);
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
    AttnBwdArgs a{
#ifdef LG_GEMM_TIMELINE
        timeline_buffer(int(2 * (S / 32) * heads * batch)),
#endif
        q, k, v, g, ldq, sbq, ldk, sbk, ldv, sbv, ldg, sbg, p, dq, dk, dv, lddq, sbdq, lddk, sbdk, lddv, sbdv, int(S), int(heads), scale};
    const dim3 grid(unsigned(2 * (S / 32)), unsigned(heads), unsigned(batch));
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
    return LG_OK;
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
