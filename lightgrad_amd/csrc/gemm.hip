// SGEMM on the gfx950 matrix cores: C (+)= op(A) @ op(B), fp32 in / fp32 accumulate,
// built on v_mfma_f32_32x32x2_f32 (exact fp32 products, 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak).
//
// Structure (one workgroup = BM x BN tile of C, WM x WN waves, each wave a TM x TN grid of
// 32x32 MFMA accumulators):
//   global --buffer_load_dwordx4--> registers --ds_write_b128--> LDS (double buffered) --> fragments --> MFMA
//   * global loads are inline-asm buffer loads with counted vmcnt waits (a ring of 1-2 K-tiles in flight): descriptor at
//     the K-tile, loop-invariant 32-bit offsets, every predicate (row / column beyond the matrix, k beyond the slice)
//     folded into an out-of-range offset that the hardware answers with 0 - no branches, no zero-fill in the K loop.
//   * the next K-tile is requested BEFORE the MFMAs of the current tile and written to the other LDS buffer in the
//     MIDDLE of them: one barrier per K-tile, L2 / HBM latency hidden under the MFMAs (64 cycles each).  Waves with a
//     single accumulator also fetch the next half tile's LDS fragments under the current half's MFMAs.
//   * too few tiles for 256 CUs: split-K INSIDE the launch (write-through partial tiles, per-tile ticket, the last
//     workgroup to arrive folds them in slice order) - see the kernel; slice count from a fitted cost model.
//   * epilogues: bias, accumulate (beta = 1), row sums of A as a virtual ones-column of B (dW and db in one launch).
//   * operands are consumed in whatever layout the tensor has (A row- or column-major, B row- or
//     column-major) - no transposition or padding copies.  A "K-contiguous" operand is staged as
//     [row][BK+4] and a lane fetches FOUR k-values with one ds_read_b128; an "M/N-contiguous"
//     operand is staged as [BK][rows] and read with conflict-free ds_read_b32.  Both feed the same
//     MFMA because the k order inside a K-block of 8 is a free choice: MFMA step s (0..3) and
//     lane half h (0..1) use k = kb + 4h + s for BOTH operands.
//   * the (BK+4) row pitch makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots
//     (pitch/4 is odd), i.e. conflict-free without a swizzle.
//   * workgroup ids are remapped so that the blocks sharing an XCD (ids equal mod 8) work on
//     neighbouring tiles and share A/B panels in that XCD's L2.
//   * edges are predicated (zero-filled loads, masked stores): any M, N, K >= 1.
// Reference semantics: `dot` = numpy matmul (cpu/ops.py:107-116); the tiled OpenCL kernel with its
// pad-to-128 and contiguous() copies (opencl/kernels.py:201-337) is not reproduced.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace lg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef LG_SMALL_PD
#define LG_SMALL_PD 2
#endif
constexpr int kSmallTilePrefetch = LG_SMALL_PD;

// n / d for n < 2^31 with a host-made multiplier: (n * m) >> s, m = floor(2^(31+l) / d) + 1, l = ceil(log2 d).  A lone
// small workgroup cannot issue its first load before it knows its tile: five emulated integer divisions (~25
// instructions each) at the top of the kernel are time it cannot hide.
struct FastDiv {
    uint32_t m, s;
    __device__ __forceinline__ int div(int n) const { return int((uint64_t(uint32_t(n)) * m) >> s); }
};
static FastDiv make_fastdiv(int64_t d) {
    if (d < 1) d = 1;
    uint32_t l = 0;
    while ((int64_t(1) << l) < d) ++l;
    FastDiv f;
    f.m = uint32_t((uint64_t(1) << (31 + l)) / uint64_t(d) + 1);
    f.s = 31 + l;
    return f;
}

struct GemmArgs {
    const float* A;
    const float* B;
    float*       C;
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int64_t sA, sB, sC;     // batch strides (elements) of the outer batch index
    int64_t sA2, sB2, sC2;  // ... of the inner batch index (batch = outer * batch_inner + inner; attention: (b, head))
    int     batch_inner;    // >= 1
    int     tiles_m, tiles_n;
    int     nwg;            // tiles_m * tiles_n * batch * k_slices
    int     accumulate;
    const float* bias;      // optional [N]: added to every row of the product (nn.Linear's `+ b`, nn.py:96)
    // split-K: slice s of k_slices handles k in [s*k_per_slice, min(K, (s+1)*k_per_slice)) and writes its partial
    // tile to the workspace W (accumulator layout, see the kernel); the workgroup that arrives LAST at a tile
    // (per-tile ticket) sums the slices in index order - deterministic - and runs the epilogue
    int     group_m;        // tile rows walked before moving to the next tile column
    int     k_slices;
    int64_t k_per_slice;    // multiple of BK
    float*  W;
    int*    tickets;        // one per (batch, tile), zero on entry and on exit
    // optional row sums of op(A) (= the bias gradient when op(A) = g^T, nn.py:96 / func.py:50-56): computed as column N
    // of C against a VIRTUAL column of ones appended to op(B) - the tiling, split-K and fold treat it like any column
    float*  rowsum;         // [M] or NULL
    int     rowsum_accumulate;
    FastDiv div_per_batch, div_slices, div_gspan, div_group, div_last_group, div_batch_inner;
    int     k_tail;         // K % 4 != 0: K-contiguous float4s of the last tile carry elements beyond K, zeroed before LDS
    // relu folded into its consumers (the tape's relu stays lazy, autograd/hip/ops.py): op(A) / op(B) are passed through
    // np.maximum(., 0) on their way to LDS
    int     relu_a, relu_b;
#ifdef LG_GEMM_TIMELINE
    // experiments build only (make timeline; tools/gemm_timeline.py): 8 timestamps of the 100 MHz wall clock per workgroup
    unsigned long long* tl;
#endif
};

#ifdef LG_GEMM_TIMELINE
#define LG_TL(slot) do { if (g.tl && threadIdx.x == 0) g.tl[size_t(blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define LG_TL(slot) do { } while (0)
#endif

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of
// tile ids.  Bijective for any nwg (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, pos = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

// the 256x256 tile never splits K (its launches have >= 256 tiles or lose to smaller tiles in the cost model); leaving
// the fold out of that instantiation keeps its main loop free of spills
template <int BM, int BN> constexpr bool kHasExtras = BM * BN < 256 * 256;      // relu operands, row sums: small tiles only
template <int BM, int BN, int KG = 1> constexpr bool kCanSplitK = kHasExtras<BM, BN> && KG == 1;

// KG = 2: "K-groups" - the split of K happens INSIDE the workgroup.  Twice the waves on the same BM x BN tile; a K-step stages
// BK*KG k-values, wave group g multiplies k in [g*BK, (g+1)*BK) of it, and after the loop group 1 hands its accumulators to
// group 0 through LDS.  For outputs with too few 64x64 tiles to fill the chip this replaces the cross-workgroup split-K (slab
// write + drain + ticket + fold = 2.6 us of a 14.6 us launch, profiles/r2/gemm_timeline_v1.txt) by one LDS exchange: a 64x32
// tile with 2 x 2 waves does per wave exactly the MFMA work of a 64x64 tile with 2 K-slices, one workgroup per CU.
template <int BM, int BN, int BK, bool AKC, bool BKC, int KG>
constexpr int gemm_lds_floats() {
    constexpr int BKS = BK * KG;
    return 2 * ((AKC ? BM * (BKS + 4) : BKS * BM) + (BKC ? BN * (BKS + 4) : BKS * BN));
}

// One output tile (or K-slice of one): the whole kernel body, as a device function of the workgroup index so that ONE launch
// can work on two independent products (sgemm_pair below).
template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, bool VA, bool VB, int PD, int KG = 1>
__device__ __forceinline__ void sgemm_tile(const GemmArgs& g, const int bid, float* __restrict__ lds) {
    constexpr int NT = WM * WN * KG * 64;
    constexpr int BKS = BK * KG;                         // k-values staged per K-step
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1 && BK % 16 == 0 && (KG == 1 || KG == 2), "tile config");
    constexpr int A_PITCH = AKC ? BKS + 4 : BM;          // floats per LDS row
    constexpr int B_PITCH = BKC ? BKS + 4 : BN;
    constexpr int A_TILE = AKC ? BM * A_PITCH : BKS * A_PITCH;
    constexpr int B_TILE = BKC ? BN * B_PITCH : BKS * B_PITCH;
    constexpr int A_ELEMS = BM * BKS / NT, B_ELEMS = BN * BKS / NT;   // floats staged per thread
    static_assert(A_ELEMS % 4 == 0 && B_ELEMS % 4 == 0, "staging must divide into float4");

    static_assert(2 * (A_TILE + B_TILE) == gemm_lds_floats<BM, BN, BK, AKC, BKC, KG>(), "LDS size");
    // buffer b: A tile at lds + b*(A_TILE+B_TILE), B tile right behind it
    constexpr int BUF = A_TILE + B_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kg = wave / (WM * WN), wl = wave % (WM * WN);          // K-group of this wave, wave inside the group
    const int wm = wl / WN, wn = wl % WN;
    const int kofs = kg * BK;                                        // its k offset inside a staged K-step
    const int r = lane & 31, h = lane >> 5;
    LG_TL(0);                                        // workgroup entered

    // tile coordinates
    const int id = xcd_remap(bid, g.nwg);
    const int per_batch = g.tiles_m * g.tiles_n;
    const int bs = g.div_per_batch.div(id);        // (batch, k-slice) pair
    const int batch = g.div_slices.div(bs), slice = bs - batch * g.k_slices;
    const int t = id - bs * per_batch;
    // grouped order: consecutive ids walk GROUP_M tile rows before moving to the next tile column, so the
    // ~64 workgroups resident on one XCD at a time cover a near-square patch of C and share both their
    // A row-panels and their B column-panels in that XCD's 4 MiB L2
    const int GROUP_M = g.group_m;
    const int gspan = GROUP_M * g.tiles_n;
    const int group = g.div_gspan.div(t);
    const int first_m = group * GROUP_M, in_group = t - group * gspan;
    const bool last_group = g.tiles_m - first_m < GROUP_M;
    const int gsize = last_group ? g.tiles_m - first_m : GROUP_M;
    const int tn = last_group ? g.div_last_group.div(in_group) : g.div_group.div(in_group);
    const int tm = first_m + (in_group - tn * gsize);
    const int64_t m0 = int64_t(tm) * BM, n0 = int64_t(tn) * BN;
    const int b_outer = g.div_batch_inner.div(batch), b_inner = batch - b_outer * g.batch_inner;
    const float* __restrict__ A = g.A + int64_t(b_outer) * g.sA + int64_t(b_inner) * g.sA2;
    const float* __restrict__ B = g.B + int64_t(b_outer) * g.sB + int64_t(b_inner) * g.sB2;
    float* __restrict__ C = g.C + int64_t(b_outer) * g.sC + int64_t(b_inner) * g.sC2;
    const int64_t ldc = g.ldc;
    const int accumulate = g.accumulate;
    const float* __restrict__ bias = g.bias;
    const int64_t k_begin = int64_t(slice) * g.k_per_slice;
    const int64_t k_end = (k_begin + g.k_per_slice < g.K) ? k_begin + g.k_per_slice : g.K;
    // the virtual ones-column (row sums of A) lives in column N of this workgroup's tile, if at all
    const bool has_virtual = kHasExtras<BM, BN> && g.rowsum != nullptr && n0 <= g.N && g.N < n0 + BN;

    // staging registers: a ring of PD K-tiles in flight between global memory and LDS (PD = 1: the tile fetched at the
    // top of an iteration is written to LDS in its middle; small tiles have too few MFMAs per K-tile to cover the
    // load latency that way and run PD = 2..3)
    using StageA = std::conditional_t<VA, f32x4, float>;
    using StageB = std::conditional_t<VB, f32x4, float>;
    constexpr int A_CHUNKS = VA ? A_ELEMS / 4 : A_ELEMS, B_CHUNKS = VB ? B_ELEMS / 4 : B_ELEMS;
    constexpr int NL = A_CHUNKS + B_CHUNKS;          // loads per thread and K-tile
    static_assert((PD - 1) * NL + NL <= 63, "vmcnt is a 6-bit counter");
    StageA ra_ring[PD][A_CHUNKS];
    StageB rb_ring[PD][B_CHUNKS];

    // Global -> register staging through buffer loads: the descriptor starts at the K-tile's first element (wave-uniform),
    // the per-thread byte offsets are loop invariant, and every predicate (row / column beyond the matrix, k beyond the
    // slice) turns the offset into one past the descriptor's range, for which the hardware returns 0 - no branches and
    // no zero-fill moves in the K loop.
    constexpr unsigned OOB = 0x80000000u;            // == num_records
    unsigned offA[A_CHUNKS], offB[B_CHUNKS];
    unsigned virt_mask = 0;                          // B chunks of this thread that start at the virtual column
    int krem_ring[PD];                               // k extent of the tile waiting in each ring slot
    int kcA[A_CHUNKS], kcB[B_CHUNKS];                // k coordinate of the chunk inside its K-tile
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int f = tid + i * NT;
        int row, kk;                                  // row: index along M inside the tile, kk: index along K
        if constexpr (VA) { if constexpr (AKC) { row = f / (BKS / 4); kk = (f % (BKS / 4)) * 4; } else { kk = f / (BM / 4); row = (f % (BM / 4)) * 4; } }
        else              { if constexpr (AKC) { row = f / BKS; kk = f % BKS; } else { kk = f / BM; row = f % BM; } }
        kcA[i] = kk;
        // 32-bit on purpose: the host guarantees that a tile's farthest byte offset is below 2^31
        const unsigned bytes = AKC ? (unsigned(row) * unsigned(g.lda) + unsigned(kk)) * 4u : (unsigned(kk) * unsigned(g.lda) + unsigned(row)) * 4u;
        offA[i] = (m0 + row < g.M) ? bytes : OOB;
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
        const int f = tid + i * NT;
        int col, kk;
        if constexpr (VB) { if constexpr (BKC) { col = f / (BKS / 4); kk = (f % (BKS / 4)) * 4; } else { kk = f / (BN / 4); col = (f % (BN / 4)) * 4; } }
        else              { if constexpr (BKC) { col = f / BKS; kk = f % BKS; } else { kk = f / BN; col = f % BN; } }
        kcB[i] = kk;
        const unsigned bytes = BKC ? (unsigned(col) * unsigned(g.ldb) + unsigned(kk)) * 4u : (unsigned(kk) * unsigned(g.ldb) + unsigned(col)) * 4u;
        offB[i] = (n0 + col < g.N) ? bytes : OOB;
        if (has_virtual && n0 + col == g.N) virt_mask |= 1u << i;
    }
    const float* const Atile0 = AKC ? A + m0 * g.lda : A + m0;      // + k0 (AKC) / + k0 * lda per K-tile
    const float* const Btile0 = BKC ? B + n0 * g.ldb : B + n0;

    // The loads are inline asm so that the K loop can keep PD tiles in flight: hipcc's own wait insertion is
    // conservative across the loop back-edge (it drains every outstanding load before the first LDS write).  The
    // destination registers are NOT protected until wait_tile() below has run for that ring slot.
    auto descriptor = [](const float* base) {
        const unsigned long long p = reinterpret_cast<unsigned long long>(base);
        u32x4 d;
        d[0] = unsigned(p); d[1] = unsigned(p >> 32) & 0xffffu; d[2] = OOB; d[3] = 0x00020000u;
        return d;
    };
    // tiles are requested strictly in order: running pointers instead of index arithmetic (every scalar instruction in
    // the K loop of a lone small-tile workgroup shows up in its time)
    const float* nextA = AKC ? Atile0 + k_begin : Atile0 + k_begin * g.lda;
    const float* nextB = BKC ? Btile0 + k_begin : Btile0 + k_begin * g.ldb;
    const int64_t stepA = AKC ? int64_t(BKS) : int64_t(BKS) * g.lda, stepB = BKC ? int64_t(BKS) : int64_t(BKS) * g.ldb;
    const int nkt = int((k_end - k_begin + BKS - 1) / BKS);         // K-steps of this slice (32-bit: scalar compares in the loop)
    const int last_krem = int(k_end - k_begin) - (nkt - 1) * BKS;    // k values of the last one
    int requested = 0;
    auto load_tile = [&](int slot) {
        const int krem = requested == nkt - 1 ? last_krem : BKS;     // k values of this tile inside the slice
        ++requested;
        krem_ring[slot] = krem;
        const u32x4 da = descriptor(nextA);
        const u32x4 db = descriptor(nextB);
        nextA += stepA;
        nextB += stepB;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const unsigned off = kcA[i] < krem ? offA[i] : OOB;
            if constexpr (VA) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra_ring[slot][i]) : "v"(off), "s"(da));
            else              asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(ra_ring[slot][i]) : "v"(off), "s"(da));
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const unsigned off = kcB[i] < krem ? offB[i] : OOB;
            if constexpr (VB) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb_ring[slot][i]) : "v"(off), "s"(db));
            else              asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(rb_ring[slot][i]) : "v"(off), "s"(db));
        }
    };
    // wait until the loads of ring slot `slot` have landed, given that `younger` tiles were requested after it
    auto wait_tile = [&](int slot, int younger) {
        if (PD >= 3 && younger >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PD >= 3 ? 2 * NL : 0) : "memory");
        else if (PD >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PD >= 2 ? NL : 0) : "memory");
        else                              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) asm volatile("" : "+v"(ra_ring[slot][i]));
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) asm volatile("" : "+v"(rb_ring[slot][i]));
    };

    auto store_tile = [&](int buf, int slot) {
        float* a = lds + buf * BUF;
        float* b = lds + buf * BUF + A_TILE;
        // K not a multiple of 4: the last float4 of a K-contiguous row runs into the next row - zero what lies beyond the
        // slice (only the slice's last tile can be short)
        if (g.k_tail && krem_ring[slot] < BKS) {
            asm volatile("; short K tail" ::: "memory");        // keeps this rare fix-up a branch, not selects in every iteration
            if constexpr (VA && AKC) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i)
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (kcA[i] + e >= krem_ring[slot]) ra_ring[slot][i][e] = 0.f;
            }
            if constexpr (VB && BKC) {
#pragma unroll
                for (int i = 0; i < B_CHUNKS; ++i)
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (kcB[i] + e >= krem_ring[slot]) rb_ring[slot][i][e] = 0.f;
            }
        }
        if constexpr (kHasExtras<BM, BN>) if (g.relu_a) {
            asm volatile("; relu(A) on the way to LDS" ::: "memory");
#pragma unroll
            for (int i = 0; i < A_CHUNKS; ++i) {
                if constexpr (VA) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float x = ra_ring[slot][i][e]; ra_ring[slot][i][e] = (x != x) ? x : (x > 0.0f ? x : 0.0f); }
                } else {
                    const float x = ra_ring[slot][i]; ra_ring[slot][i] = (x != x) ? x : (x > 0.0f ? x : 0.0f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int f = tid + i * NT;
            if constexpr (VA) {
                if constexpr (AKC) *reinterpret_cast<f32x4*>(a + (f / (BKS / 4)) * A_PITCH + (f % (BKS / 4)) * 4) = ra_ring[slot][i];
                else               *reinterpret_cast<f32x4*>(a + (f / (BM / 4)) * A_PITCH + (f % (BM / 4)) * 4) = ra_ring[slot][i];
            } else {
                if constexpr (AKC) a[(f / BKS) * A_PITCH + (f % BKS)] = ra_ring[slot][i];
                else               a[(f / BM) * A_PITCH + (f % BM)] = ra_ring[slot][i];
            }
        }
        if constexpr (kHasExtras<BM, BN>) if (g.relu_b) {
            asm volatile("; relu(B) on the way to LDS" ::: "memory");
#pragma unroll
            for (int i = 0; i < B_CHUNKS; ++i) {
                if constexpr (VB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float x = rb_ring[slot][i][e]; rb_ring[slot][i][e] = (x != x) ? x : (x > 0.0f ? x : 0.0f); }
                } else {
                    const float x = rb_ring[slot][i]; rb_ring[slot][i] = (x != x) ? x : (x > 0.0f ? x : 0.0f);
                }
            }
        }
        if constexpr (kHasExtras<BM, BN>) if (has_virtual) {
            // the loads returned zeros for the virtual column (beyond N): put the ones in, for the k values that exist
#pragma unroll
            for (int i = 0; i < B_CHUNKS; ++i) {
                const bool virt = (virt_mask >> i) & 1u, kv = kcB[i] < krem_ring[slot];
                const float one = kv ? 1.0f : 0.0f;
                if constexpr (VB) {
                    if constexpr (BKC) {                                                           // row N of B^T: ones along k
                        if (virt) {
                            const int kr = krem_ring[slot] - kcB[i];
                            rb_ring[slot][i] = f32x4{kr > 0 ? 1.f : 0.f, kr > 1 ? 1.f : 0.f, kr > 2 ? 1.f : 0.f, kr > 3 ? 1.f : 0.f};
                        }
                    }
                    else               { if (virt) rb_ring[slot][i] = f32x4{one, 0.f, 0.f, 0.f}; }     // k-row of B: column N, then beyond
                } else {
                    if (virt) rb_ring[slot][i] = one;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int f = tid + i * NT;
            if constexpr (VB) {
                if constexpr (BKC) *reinterpret_cast<f32x4*>(b + (f / (BKS / 4)) * B_PITCH + (f % (BKS / 4)) * 4) = rb_ring[slot][i];
                else               *reinterpret_cast<f32x4*>(b + (f / (BN / 4)) * B_PITCH + (f % (BN / 4)) * 4) = rb_ring[slot][i];
            } else {
                if constexpr (BKC) b[(f / BKS) * B_PITCH + (f % BKS)] = rb_ring[slot][i];
                else               b[(f / BN) * B_PITCH + (f % BN)] = rb_ring[slot][i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute_tile = [&](int buf, int kb_begin, int kb_end) {
        const float* a = lds + buf * BUF + (AKC ? (wm * TM * 32 + r) * A_PITCH + 4 * h + kofs : (4 * h + kofs) * A_PITCH + wm * TM * 32 + r);
        const float* b = lds + buf * BUF + A_TILE + (BKC ? (wn * TN * 32 + r) * B_PITCH + 4 * h + kofs : (4 * h + kofs) * B_PITCH + wn * TN * 32 + r);
#pragma unroll
        for (int kb = kb_begin; kb < kb_end; kb += 8) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (AKC) {
                    const float4 v = *reinterpret_cast<const float4*>(a + i * 32 * A_PITCH + kb);
                    fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) fa[i][s] = a[(kb + s) * A_PITCH + i * 32];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BKC) {
                    const float4 v = *reinterpret_cast<const float4*>(b + j * 32 * B_PITCH + kb);
                    fb[j][0] = v.x; fb[j][1] = v.y; fb[j][2] = v.z; fb[j][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) fb[j][s] = b[(kb + s) * B_PITCH + j * 32];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
    };

    // tile j waits in ring slot j % PD; tiles 0 .. PD-1 are requested up front
#pragma unroll
    for (int u = 0; u < PD; ++u)
        if (u < nkt) load_tile(u);
    {
        const int younger = nkt - 1 < PD - 1 ? nkt - 1 : PD - 1;
        wait_tile(0, younger);
    }
    store_tile(0, 0);
    __syncthreads();
    LG_TL(1);                                        // first K-tile in LDS
    if constexpr (TM * TN == 1) {
        // One accumulator per wave (small tiles: often ONE such wave per SIMD, nothing else to hide latency behind):
        // the fragments of the next half K-tile are fetched from LDS while the MFMAs of the current half run.  The
        // barrier sits between the halves: before it every wave has read ALL of tile kt and written its share of tile
        // kt+1, after it the first-half fragments of tile kt+1 are fetched under the second half's MFMAs.
        constexpr int HG = BK / 16;                  // k-groups of 8 per half tile
        float fa[2][HG][4], fb[2][HG][4];
        auto read_half = [&](int buf, int half, int slot) {
            const float* a = lds + buf * BUF + (AKC ? (wm * 32 + r) * A_PITCH + 4 * h + kofs : (4 * h + kofs) * A_PITCH + wm * 32 + r);
            const float* b = lds + buf * BUF + A_TILE + (BKC ? (wn * 32 + r) * B_PITCH + 4 * h + kofs : (4 * h + kofs) * B_PITCH + wn * 32 + r);
#pragma unroll
            for (int q = 0; q < HG; ++q) {
                const int kb = half * (BK / 2) + q * 8;
                if constexpr (AKC) {
                    const float4 v = *reinterpret_cast<const float4*>(a + kb);
                    fa[slot][q][0] = v.x; fa[slot][q][1] = v.y; fa[slot][q][2] = v.z; fa[slot][q][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[slot][q][e] = a[(kb + e) * A_PITCH];
                }
                if constexpr (BKC) {
                    const float4 v = *reinterpret_cast<const float4*>(b + kb);
                    fb[slot][q][0] = v.x; fb[slot][q][1] = v.y; fb[slot][q][2] = v.z; fb[slot][q][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[slot][q][e] = b[(kb + e) * B_PITCH];
                }
            }
        };
        auto mfma_part = [&](int slot, int q0, int q1) {
#pragma unroll
            for (int q = q0; q < q1; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][q][e], fb[slot][q][e], acc[0][0], 0, 0, 0);
        };
        auto mfma_half = [&](int slot) { mfma_part(slot, 0, HG); };
        constexpr int HQ = HG > 1 ? HG / 2 : HG;     // MFMA groups issued before the LDS write of the next tile
        read_half(0, 0, 0);
        for (int kt0 = 0; kt0 < nkt; kt0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int kt = kt0 + u;
                if (kt < nkt) {
                    const int cur = kt & 1;
                    if (PD == 1 && kt + 1 < nkt) load_tile(0);
                    read_half(cur, 1, 1);
                    // the LDS write of tile kt+1 goes in the MIDDLE of the dependent MFMA chain: its latency, and the
                    // barrier's, then run under the MFMAs still queued instead of after the last one
                    mfma_part(0, 0, HQ);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < nkt) {
                        int younger = nkt - (kt + 2);
                        if (younger > PD - 2) younger = PD - 2;
                        wait_tile((u + 1) % PD, PD == 1 ? 0 : (younger < 0 ? 0 : younger));
                        store_tile(cur ^ 1, (u + 1) % PD);
                    }
                    if (PD > 1 && kt + PD < nkt) load_tile(u);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_part(0, HQ, HG);
                    __syncthreads();
                    if (kt + 1 < nkt) read_half(cur ^ 1, 0, 0);
                    mfma_half(1);
                }
            }
        }
    } else {
        for (int kt0 = 0; kt0 < nkt; kt0 += PD) {
    #pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int kt = kt0 + u;
                if (kt < nkt) {
                    const int cur = kt & 1;
                    // PD == 1: tile kt+1 is requested here and written to LDS in the middle of this iteration.
                    // PD  > 1: slot u (tile kt, in LDS since the last iteration) is refilled with tile kt + PD below, after
                    //          the LDS write of tile kt+1 - whose loads were requested PD-1 iterations ago.
                    if (PD == 1 && kt + 1 < nkt) load_tile(0);
                    compute_tile(cur, 0, BK / 2);
                    if (kt + 1 < nkt) {
                        int younger = nkt - (kt + 2);                 // tiles kt+2 .. kt+PD-1 requested after tile kt+1
                        if (younger > PD - 2) younger = PD - 2;
                        wait_tile((u + 1) % PD, PD == 1 ? 0 : (younger < 0 ? 0 : younger));
                        store_tile(cur ^ 1, (u + 1) % PD);                // ds_writes issue in the shadow of the second half's MFMAs
                    }
                    if (PD > 1 && kt + PD < nkt) load_tile(u);
                    compute_tile(cur, BK / 2, BK);
                    __syncthreads();
                }
            }
        }
    }

    LG_TL(2);                                        // K loop done
    if constexpr (KG == 2) {
        // K-group 1 hands its accumulators to group 0 through LDS (the staging buffers are free after the loop's last
        // barrier): slot (wave-in-group, accumulator, register) holds one value per lane, so both sides move 256 B per
        // wave instruction without bank conflicts.  Group 1 is done after that.
        float* x = lds + (wl * TM * TN * 16) * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[((i * TN + j) * 16 + e) * 64] = acc[i][j][e];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += x[((i * TN + j) * 16 + e) * 64];
    }
    if constexpr (kCanSplitK<BM, BN, KG>) if (g.k_slices > 1) {
        // split-K, folded inside the launch (cdna_hip_programming.md, in-launch split-K recipe, write-through form).
        // Partial tiles go to the workspace in ACCUMULATOR layout - 16-byte piece ((i*TN + j)*4 + q) of thread tid at
        // byte (((i*TN + j)*4 + q)*NT + tid)*16 of the (batch, slice, tile) slab - so stores and the fold move 1 KiB per
        // wave instruction.  The slices of a tile run on different XCDs whose L2s are not coherent: slabs are stored
        // write-through (sc1) and drained, ONE lane takes an agent-scope ticket, and the workgroup that arrives last
        // reads every slab with sc1 loads (never served from a stale line), sums them in slice order - the same
        // order in every run, so results are bit-reproducible - and runs the epilogue.  The ticket is reset for the
        // next launch by that workgroup.
        constexpr int SC1 = 16;                                   // aux bit of the raw buffer builtins
        constexpr int TILE_BYTES = BM * BN * 4;
        char* slab0 = reinterpret_cast<char*>(g.W) + (int64_t(batch) * g.k_slices * per_batch + t) * int64_t(TILE_BYTES);
        const int64_t slice_stride = int64_t(per_batch) * TILE_BYTES;
        {
            const auto mine = __builtin_amdgcn_make_buffer_rsrc(slab0 + slice * slice_stride, 0, TILE_BYTES, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][4 * q + e]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, mine, (((i * TN + j) * 4 + q) * NT + tid) * 16, 0, SC1);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
        __syncthreads();
        LG_TL(3);                                                 // slab written and drained
        int* arrived_last = reinterpret_cast<int*>(lds);          // the staging buffers are free after the K loop's last barrier
        if (tid == 0) {
            int* ticket = g.tickets + batch * per_batch + t;
            const int order = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = order == g.k_slices - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *arrived_last = last;
        }
        __syncthreads();
        LG_TL(4);                                                 // ticket drawn
        if (!*arrived_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // no instruction: keeps the slab loads below the ticket
        constexpr int CH = TM * TN >= 4 ? 1 : (TM * TN == 2 ? 2 : 4);     // slices whose loads are in flight together
        for (int s0 = 0; s0 < g.k_slices; s0 += CH) {
            u32x4 v[CH][TM * TN * 4];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int sl = s0 + c < g.k_slices ? s0 + c : s0;
                const auto src = __builtin_amdgcn_make_buffer_rsrc(slab0 + sl * slice_stride, 0, TILE_BYTES, 0x00020000);
#pragma unroll
                for (int f = 0; f < TM * TN * 4; ++f) v[c][f] = __builtin_amdgcn_raw_buffer_load_b128(src, (f * NT + tid) * 16, 0, SC1);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const bool live = s0 + c < g.k_slices, first = s0 + c == 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float w = __uint_as_float(v[c][(i * TN + j) * 4 + e / 4][e % 4]);
                            acc[i][j][e] = first ? w : (live ? acc[i][j][e] + w : acc[i][j][e]);
                        }
            }
        }
    }

    LG_TL(5);                                        // slabs folded (or nothing to fold)
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + (wn * TN + j) * 32 + r;
            const int64_t row0 = m0 + (wm * TM + i) * 32 + 4 * h;
            const bool vcol = has_virtual && col == g.N;             // this lane holds row sums, not a column of C
            if (col < g.N || vcol) {
                const float bv = (bias && !vcol) ? bias[col] : 0.f;
                float* const dst = vcol ? g.rowsum : C + col;
                const int64_t dstride = vcol ? 1 : ldc;
                const bool acc_flag = vcol ? g.rowsum_accumulate != 0 : accumulate != 0;
                // The values first, the stores last, and the read-modify-write of `C += ...` in a block of its own:
                // on gfx9-family hardware stores count in vmcnt like loads, so when the old-value loads sat in the same
                // straight-line code as the stores, the wait in front of every add also drained the PREVIOUS store -
                // 16 serialised store round trips, 2.0 of a 15.6 us launch (tools/gemm_timeline.py).
                float val[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] = bias ? acc[i][j][e] + bv : acc[i][j][e];
                if (acc_flag) {
                    float old[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                        old[e] = row < g.M ? dst[row * dstride] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) val[e] = old[e] + val[e];
                }
                float* p = dst + row0 * dstride;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < g.M) p[((e & 3) + 8 * (e >> 2)) * dstride] = val[e];
                }
            }
        }
    }
    LG_TL(6);                                        // epilogue stores issued
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, bool VA, bool VB, int PD, int KG = 1>
__global__ void __launch_bounds__(WM * WN * KG * 64) sgemm_mfma(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, AKC, BKC, KG>()];
    sgemm_tile<BM, BN, BK, WM, WN, AKC, BKC, VA, VB, PD, KG>(g, blockIdx.x, lds);
}

// Two independent products in ONE launch: workgroups [0, first.nwg) work on the first, the rest on the second.  Made for the
// pair every Linear's backward produces - dW (+ db) = g^T @ x and dx = g @ W: both have too few 64x64 tiles to give a CU more
// than one workgroup, and a lone workgroup leaves the matrix cores idle during its waits (0.68 us per K-tile against 0.43 us
// of MFMAs); together their workgroups share the CUs and interleave (0.51 us per K-tile each when two are resident), and one
// launch floor (3 us) disappears.  Layouts: the first product A^T-form (M-contiguous A, N-contiguous B), the second
// K-contiguous A and N-contiguous B - what dense row-major g, x, W give.
template <int PD>
__global__ void __launch_bounds__(256) sgemm_pair_wgrad_xgrad(GemmArgs first, GemmArgs second) {
    constexpr int L1 = gemm_lds_floats<64, 64, 32, false, false, 1>(), L2 = gemm_lds_floats<64, 64, 32, true, false, 1>();
    __shared__ __attribute__((aligned(16))) float lds[L1 > L2 ? L1 : L2];
    if (int(blockIdx.x) < first.nwg) sgemm_tile<64, 64, 32, 2, 2, false, false, true, true, PD, 1>(first, blockIdx.x, lds);
    else                             sgemm_tile<64, 64, 32, 2, 2, true, false, true, true, PD, 1>(second, int(blockIdx.x) - first.nwg, lds);
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, int KG>
static void launch_layout(const GemmArgs& g, bool va, bool vb) {
    dim3 grid(g.nwg), block(WM * WN * KG * 64);
    hipStream_t s = rt().stream;
    constexpr int PD = (BM * BN <= 64 * 64) ? kSmallTilePrefetch : 1;
    if (va && vb)  hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, true, PD, KG>), grid, block, 0, s, g);
    else if (va)   hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, false, 1, KG>), grid, block, 0, s, g);
    else if (vb)   hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, false, true, 1, KG>), grid, block, 0, s, g);
    else           hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, false, false, 1, KG>), grid, block, 0, s, g);
}

// ---- two products in one launch (lg_gemm_pair_begin / _end) --------------------------------------------------------------
// Between begin and end, up to two products that resolve to the 64x64 tile in the layouts of sgemm_pair_wgrad_xgrad are
// prepared but not launched; end launches them together.  Anything else flushes what is pending (single launches, in call
// order) and runs as usual - so the bracket never changes results, only how many launches there are.
#ifdef LG_GEMM_TIMELINE
static void lg_debug_timeline_state(unsigned long long* buf, int nwg, int slices, int tiles);
#endif
struct PairState {
    int      active = 0;       // 0: no bracket; 1: collecting; 2: bracket open but no longer collecting
    int      count = 0;
    GemmArgs args[2];
};
static PairState& pair_state() { static PairState p; return p; }

static void pair_launch_single(const GemmArgs& g, int slot) {
    dim3 grid(g.nwg), block(256);
    constexpr int PD = kSmallTilePrefetch;
    if (slot == 0) hipLaunchKernelGGL((sgemm_mfma<64, 64, 32, 2, 2, false, false, true, true, PD, 1>), grid, block, 0, rt().stream, g);
    else           hipLaunchKernelGGL((sgemm_mfma<64, 64, 32, 2, 2, true, false, true, true, PD, 1>), grid, block, 0, rt().stream, g);
}

static int pair_flush(bool keep_collecting) {
    PairState& P = pair_state();
    int rc = LG_OK;
    if (P.count == 2) {
#ifdef LG_GEMM_TIMELINE
        if (P.args[0].tl) {
            (void)hipMemsetAsync(P.args[0].tl, 0, size_t(P.args[0].nwg + P.args[1].nwg) * 64, rt().stream);
            lg_debug_timeline_state(P.args[0].tl, P.args[0].nwg + P.args[1].nwg, P.args[0].k_slices * 100 + P.args[1].k_slices, P.args[0].nwg);
        }
#endif
        hipLaunchKernelGGL((sgemm_pair_wgrad_xgrad<kSmallTilePrefetch>), dim3(P.args[0].nwg + P.args[1].nwg), dim3(256), 0, rt().stream,
                           P.args[0], P.args[1]);
    } else if (P.count == 1) {
        pair_launch_single(P.args[0], 0);
    }
    for (int i = 0; i < P.count; ++i)
        if (P.args[i].W) { const int r = lg_free(P.args[i].W); if (r != LG_OK) rc = r; }       // stream-ordered: reused by later launches only
    P.count = 0;
    if (!keep_collecting && P.active) P.active = 2;
    return rc;
}

// true: `g` (fully prepared, workspace allocated) has been taken over and will be launched by lg_gemm_pair_end
static bool pair_try_defer(GemmArgs& g, bool akc, bool bkc, bool va, bool vb, int64_t batch) {
    PairState& P = pair_state();
    if (P.active != 1) return false;
    const int slot = P.count;
    const bool fits = slot < 2 && va && vb && batch == 1 && !bkc && (slot == 0 ? !akc : akc);
    const int64_t first_tiles = slot == 1 ? int64_t(P.args[0].tiles_m) * P.args[0].tiles_n : 0;
    if (!fits || first_tiles + int64_t(g.tiles_m) * g.tiles_n > rt().n_gemm_tickets) return false;
    g.tickets = rt().gemm_tickets + first_tiles;           // the two products fold their K-slices with disjoint tickets
    P.args[slot] = g;
    P.count = slot + 1;
    return true;
}

#ifdef LG_GEMM_TIMELINE
static unsigned long long* g_tl_buf = nullptr;
static int g_tl_nwg = 0, g_tl_slices = 0, g_tl_tiles = 0;
static void lg_debug_timeline_state(unsigned long long* buf, int nwg, int slices, int tiles) {
    g_tl_buf = buf; g_tl_nwg = nwg; g_tl_slices = slices; g_tl_tiles = tiles;
}
#endif

template <int BM, int BN, int BK, int WM, int WN, int KG = 1>
static int launch_config(const GemmArgs& base, bool akc, bool bkc, bool va, bool vb, int64_t batch) {
    GemmArgs g = base;
    g.tiles_m = int((g.M + BM - 1) / BM);
    g.tiles_n = int((g.N + (g.rowsum ? 1 : 0) + BN - 1) / BN);
    const int64_t tiles = int64_t(g.tiles_m) * g.tiles_n * batch;
    // split-K when the tile grid alone cannot fill the chip.  The slice count minimises a small model of the launch,
    // fitted to measurements on the MLP / BERT shapes (tools/mlp_gemm_bench.py with LG_GEMM_SLICES):
    //   K-tiles per workgroup x time per K-tile (x workgroups per CU once they exceed the CUs)  +  cost of the fold
    // with 0.65 us per K-tile of a lone 64x64 workgroup (0.45 for the two-wave tiles, 2.4 for 128x128) and
    // 2 + 0.6*slices us for writing, publishing and folding the slabs.
    int64_t slices = 1;
    if (kCanSplitK<BM, BN, KG> && tiles < 256 && g.K >= 4 * BK) {
        const double cus = rt().compute_units > 0 ? rt().compute_units : 256;
        const double t_iter = BM * BN >= 128 * 128 ? 2.4 : (BM * BN >= 64 * 64 ? 0.65 : 0.45);
        const int64_t k_tiles = (g.K + BK - 1) / BK;
        double best = 1e30;
        for (int64_t sl = 1; sl <= 64 && sl * 2 <= k_tiles; ++sl) {
            const double per_cu = double(tiles * sl) / cus;
            const double kt = double((k_tiles + sl - 1) / sl);
            double cost;
            if (BM * BN == 64 * 64) {
                // refitted in round 2 from per-workgroup timestamps (tools/gemm_timeline.py, LG_GEMM_SLICES sweep): a lone
                // 4-wave workgroup needs 0.68 us per K-tile (one wave per SIMD: its waits are idle matrix-core time), c
                // workgroups sharing a CU need 0.43*c + 0.17 us for one K-tile EACH (their waits interleave; the matrix
                // cores become the limit), the launch ends with the CUs that host ceil(per_cu) of them; publishing + folding
                // the slabs costs 1.8 + 0.85*slices us
                const double c = per_cu > 1.0 ? double(int64_t(per_cu + 0.999)) : 1.0;
                cost = kt * (c <= 1.0 ? 0.68 : 0.43 * c + 0.17) + (sl > 1 ? 1.8 + 0.85 * double(sl) : 0.0);
            } else {
                cost = kt * t_iter * (per_cu > 1.0 ? per_cu : 1.0) + (sl > 1 ? 2.0 + 0.6 * double(sl) : 0.0);
            }
            if (cost < best) { best = cost; slices = sl; }
        }
    }
    static const char* slices_env = getenv("LG_GEMM_SLICES");      // experiments only
    if (kCanSplitK<BM, BN, KG> && slices_env && atoi(slices_env) >= 1) slices = atoi(slices_env);
    static const char* pair_slices_env = getenv("LG_GEMM_PAIR_SLICES");      // experiments only: "<first>,<second>"
    if (kCanSplitK<BM, BN, KG> && pair_slices_env && pair_state().active == 1) {
        int s0 = 0, s1 = 0;
        if (sscanf(pair_slices_env, "%d,%d", &s0, &s1) == 2) { const int v = pair_state().count == 0 ? s0 : s1; if (v >= 1) slices = v; }
    }
    g.k_per_slice = ((g.K + slices - 1) / slices + BK - 1) / BK * BK;
    slices = (g.K + g.k_per_slice - 1) / g.k_per_slice;
    g.k_slices = int(slices);
    g.nwg = int(tiles * slices);
    g.div_per_batch = make_fastdiv(int64_t(g.tiles_m) * g.tiles_n);
    g.div_slices = make_fastdiv(slices);
    g.div_gspan = make_fastdiv(int64_t(g.group_m) * g.tiles_n);
    g.div_group = make_fastdiv(g.group_m);
    g.div_last_group = make_fastdiv(g.tiles_m % g.group_m ? g.tiles_m % g.group_m : g.group_m);
    g.div_batch_inner = make_fastdiv(g.batch_inner);
    g.W = nullptr;
    g.tickets = rt().gemm_tickets;
#ifdef LG_GEMM_TIMELINE
    {
        static unsigned long long* tl_buf = nullptr;
        if (!tl_buf) (void)hipMalloc(reinterpret_cast<void**>(&tl_buf), size_t(1) << 22);
        g.tl = (tl_buf && size_t(g.nwg) * 64 <= (size_t(1) << 22)) ? tl_buf : nullptr;
        if (g.tl) (void)hipMemsetAsync(g.tl, 0, size_t(g.nwg) * 64, rt().stream);
        lg_debug_timeline_state(tl_buf, g.nwg, g.k_slices, int(tiles));
    }
#endif
    if (slices > 1 && tiles > rt().n_gemm_tickets) {       // more tiles than tickets: plenty of workgroups anyway
        slices = 1;
        g.k_per_slice = (g.K + BK - 1) / BK * BK;
        g.k_slices = 1;
        g.nwg = int(tiles);
        g.div_slices = make_fastdiv(1);
    }
    if (slices > 1) {
        int rc = lg_malloc(reinterpret_cast<void**>(&g.W), size_t(tiles * slices) * BM * BN * sizeof(float));
        if (rc != LG_OK) return rc;
    }
    // (measured and not kept: two wave groups per single-accumulator tile, each on half of every K-tile - the loop is bound
    // by the workgroup barrier, which more waves of the SAME workgroup do not hide: 25.9 -> 25.4 us at 1024x512x1024)
    if constexpr (BM == 64 && BN == 64 && WM == 2 && WN == 2 && KG == 1) {
        if (pair_try_defer(g, akc, bkc, va, vb, batch)) return LG_OK;
    }
    if (pair_state().count) {                         // something else inside a pair bracket: what is pending goes first
        const int prc = pair_flush(false);
        if (prc != LG_OK) return prc;
    }
    if (akc && bkc) launch_layout<BM, BN, BK, WM, WN, true, true, KG>(g, va, vb);
    else if (akc)   launch_layout<BM, BN, BK, WM, WN, true, false, KG>(g, va, vb);
    else if (bkc)   launch_layout<BM, BN, BK, WM, WN, false, true, KG>(g, va, vb);
    else            launch_layout<BM, BN, BK, WM, WN, false, false, KG>(g, va, vb);
    if (slices > 1) return lg_free(g.W);     // stream-ordered: reused only by later launches
    return LG_OK;
}

}  // namespace lg

using namespace lg;

static int gemm_impl(int transA, int transB, int64_t M, int64_t N, int64_t K,
                     const float* A, int64_t lda, int64_t strideA,
                     const float* B, int64_t ldb, int64_t strideB,
                     float* C, int64_t ldc, int64_t strideC,
                     int64_t batch, int accumulate, const float* bias, float* rowsum = nullptr, int rowsum_accumulate = 0,
                     int64_t batch_inner = 1, int64_t strideA2 = 0, int64_t strideB2 = 0, int64_t strideC2 = 0,
                     int relu_a = 0, int relu_b = 0) {
    LG_REQUIRE_INIT();
    LG_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "lg_gemm_f32: negative extent (M=%lld N=%lld K=%lld batch=%lld)",
           (long long)M, (long long)N, (long long)K, (long long)batch);
    if (M == 0 || N == 0 || batch == 0) return LG_OK;
    LG_ARG(A && B && C, "lg_gemm_f32: NULL operand");
    LG_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
           "lg_gemm_f32: leading dimension too small (lda=%lld ldb=%lld ldc=%lld for M=%lld N=%lld K=%lld tA=%d tB=%d)",
           (long long)lda, (long long)ldb, (long long)ldc, (long long)M, (long long)N, (long long)K, transA, transB);
    LG_ARG(rowsum == nullptr || (batch == 1 && bias == nullptr), "lg_gemm_rowsum_f32: one matrix product, no bias");
    if (K == 0) {
        LG_ARG(bias == nullptr, "lg_gemm_bias_f32: K == 0 with a bias is not supported");
        if (rowsum && !rowsum_accumulate) {
            int64_t shape1[1] = {M}, st1[1] = {1};
            int rc1 = lg_fill_strided(4, 1, shape1, rowsum, st1, 0);
            if (rc1 != LG_OK) return rc1;
        }
        if (accumulate) return LG_OK;
        // empty sum: C = 0
        int64_t shape[4] = {batch / batch_inner, batch_inner, M, N}, st[4] = {strideC, strideC2, ldc, 1};
        return lg_fill_strided(4, 4, shape, C, st, 0);
    }

    // operand tiles are addressed with 32-bit byte offsets from a per-tile descriptor base: the farthest element of a
    // tile (at most 256 rows of a K-contiguous operand, 32 k-rows of the other kind) must stay below 2 GiB
    {
        auto far = [](bool k_contiguous, int64_t rows, int64_t ld) {
            const int64_t lines = k_contiguous ? (rows < 256 ? rows : 256) : 32;
            return (lines - 1) * ld * 4 + 1024;
        };
        LG_ARG(far(!transA, M, lda) < (int64_t(1) << 31) && far(transB != 0, N, ldb) < (int64_t(1) << 31),
               "lg_gemm_f32: leading dimension too large for 32-bit tile offsets (lda=%lld ldb=%lld)", (long long)lda, (long long)ldb);
    }
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.sA2 = strideA2; g.sB2 = strideB2; g.sC2 = strideC2;
    g.batch_inner = int(batch_inner);
    g.accumulate = accumulate;
    g.bias = bias;
    g.rowsum = rowsum;
    g.rowsum_accumulate = rowsum_accumulate;
    g.k_tail = (K % 4 != 0) ? 1 : 0;
    g.relu_a = relu_a; g.relu_b = relu_b;
    const bool fused_extras = rowsum != nullptr || relu_a || relu_b;     // not compiled into the 256x256 tile
    LG_ARG(!(relu_a || relu_b) || batch == 1, "lg_gemm_fused_f32: one matrix product");
    static const char* group_env = getenv("LG_GEMM_GROUP");
    g.group_m = group_env ? atoi(group_env) : 8;
    if (g.group_m < 1) g.group_m = 1;

    const bool akc = !transA;   // A[m*lda + k]: k is the contiguous index
    const bool bkc = transB != 0;   // B[n*ldb + k]
    // float4 staging: buffer loads of 4 dwords need dword alignment only, and a float4 that runs past the contiguous
    // extent is harmless: rows / columns beyond M / N are never stored, k beyond K is zeroed before it reaches LDS, and
    // the up to 12 bytes it may read behind the operand's last row exist (lg_malloc pads every block by 16 bytes; memory
    // from elsewhere must be readable that far - include/lghip.h).  The one exception: the virtual ones-column must start
    // a float4 of its own.
    const bool va = true;
    const bool vb = (rowsum == nullptr) || bkc || N % 4 == 0;
    (void)strideA2; (void)strideB2;

    // tile choice: largest tile that still yields enough workgroups for 256 CUs
    auto nblocks = [&](int64_t bm, int64_t bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
    LG_ARG(nblocks(32, 32) < (int64_t(1) << 30), "lg_gemm_f32: problem too large for one launch");
    // Tile choice.  Skinny outputs take the 64x32 / 32x64 tiles.  Otherwise three tiles compete:
    //   256x256 (16 waves, 1 WG/CU)  140-143 TFLOP/s at 4096^3   128x128 (4 waves)  ~132   64x64 (4 waves, + split-K)  138-139 at
    //   4096^3, 134 at 3072^3, 126 at 2048^3 (efficiencies refitted after the buffer-load / pipelined small-tile loop)
    // and the one with the smallest modelled time wins: (workgroups per CU, rounded up) x tile area / efficiency.  The
    // model reproduces the measured ranking from 512^3 to 8192^3 (tools/gemm_bench.py; profiles/README.md): small and
    // awkward sizes prefer many small tiles (3072^3: 64x64 = 107 vs 88-92 TFLOP/s), 4096^3 and up the 256x256 tile.
    // LG_GEMM_TILE = 0 / 2 / 9 forces 128 / 256 / 64 for experiments.
    int rc;
    static const char* tile_env = getenv("LG_GEMM_TILE");
    int tile = tile_env ? atoi(tile_env) : -1;
    if (N <= 32) {
        rc = launch_config<64, 32, 32, 2, 1>(g, akc, bkc, va, vb, batch);
    } else if (M <= 32) {
        rc = launch_config<32, 64, 32, 1, 2>(g, akc, bkc, va, vb, batch);
    } else {
        if (tile < 0) {
            const int64_t cus = rt().compute_units > 0 ? rt().compute_units : 256;
            auto cost = [&](int64_t bm, int64_t bn, double eff) {
                const int64_t per_cu = (nblocks(bm, bn) + cus - 1) / cus;
                return double(per_cu) * double(bm * bn) / eff;
            };
            const double c256 = fused_extras ? 1e300 : cost(256, 256, 0.88), c128 = cost(128, 128, 0.74), c64 = cost(64, 64, 0.85);
            tile = (c256 <= c128 && c256 <= c64) ? 2 : (c128 <= c64 ? 0 : 9);
            if (tile == 9 && batch == 1) {
                // Too few 64x64 tiles to fill the chip: split K across workgroups (slabs + ticket + fold) or INSIDE a workgroup
                // on a half-size tile (64x32 / 32x64 with two K-groups, one LDS exchange)?  Same model as launch_config (us):
                // K-steps x time per step (+ hand-off), constants from tools/gemm_timeline.py - a K-group step stages 64 k and
                // takes 0.81 us alone on a CU.  Measured at 1024x512x784: 15.9 -> 14.3 us; at 512x785x1024 a wash (kept split).
                const int64_t t64 = nblocks(64, 64);
                const int64_t k32 = (K + 31) / 32, k64 = (K + 63) / 64;
                double best64 = 1e30;
                for (int64_t sl = 1; sl <= 64 && (sl == 1 || sl * 2 <= k32); ++sl) {
                    const double per_cu = double(t64 * sl) / double(cus);
                    const double c = per_cu > 1.0 ? double(int64_t(per_cu + 0.999)) : 1.0;
                    const double v = double((k32 + sl - 1) / sl) * (c <= 1.0 ? 0.68 : 0.43 * c + 0.17) + (sl > 1 ? 1.8 + 0.85 * double(sl) : 0.0);
                    if (v < best64) best64 = v;
                    if (t64 >= 256 || K < 128) break;            // launch_config splits only below 256 tiles
                }
                const int64_t t7 = nblocks(64, 32), t8 = nblocks(32, 64);
                const int64_t tk = t7 <= t8 ? t7 : t8;
                const double ck = double((tk + cus - 1) / cus);
                const double kgroups = double(k64) * (ck <= 1.0 ? 0.81 : 0.5 * ck + 0.3) * (ck > 1.0 ? 1.0 : 1.0) + 0.3;
                if (tk <= cus && kgroups + 0.5 < best64) tile = t7 <= t8 ? 7 : 8;
            }
        }
        switch (tile) {
            case 1:  rc = launch_config<256, 128, 32, 4, 2>(g, akc, bkc, va, vb, batch); break;   // 8 waves (experiments only)
            case 2:
                if (!fused_extras) rc = launch_config<256, 256, 32, 4, 4>(g, akc, bkc, va, vb, batch);
                else               rc = launch_config<64, 64, 32, 2, 2>(g, akc, bkc, va, vb, batch);      // extras are not compiled into the big tile
                break;
            case 7:  rc = launch_config<64, 32, 32, 2, 1, 2>(g, akc, bkc, va, vb, batch); break;   // K split inside the workgroup
            case 8:  rc = launch_config<32, 64, 32, 1, 2, 2>(g, akc, bkc, va, vb, batch); break;
            case 9:  rc = launch_config<64, 64, 32, 2, 2>(g, akc, bkc, va, vb, batch); break;
            default: rc = launch_config<128, 128, 32, 2, 2>(g, akc, bkc, va, vb, batch); break;
        }
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, int64_t strideA,
                           const float* B, int64_t ldb, int64_t strideB,
                           float* C, int64_t ldc, int64_t strideC,
                           int64_t batch, int accumulate) {
    return gemm_impl(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, accumulate, nullptr);
}

extern "C" int lg_gemm_bias_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                const float* A, int64_t lda, int64_t strideA,
                                const float* B, int64_t ldb, int64_t strideB,
                                float* C, int64_t ldc, int64_t strideC,
                                int64_t batch, const float* bias) {
    return gemm_impl(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, 0, bias);
}

extern "C" int lg_gemm_rowsum_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                  const float* A, int64_t lda, const float* B, int64_t ldb,
                                  float* C, int64_t ldc, int accumulate, float* rowsum, int rowsum_accumulate) {
    LG_ARG(rowsum != nullptr, "lg_gemm_rowsum_f32: rowsum is NULL");
    return gemm_impl(transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, 1, accumulate, nullptr, rowsum, rowsum_accumulate);
}

extern "C" int lg_gemm_batched2_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                    const float* A, int64_t lda, int64_t strideA_outer, int64_t strideA_inner,
                                    const float* B, int64_t ldb, int64_t strideB_outer, int64_t strideB_inner,
                                    float* C, int64_t ldc, int64_t strideC_outer, int64_t strideC_inner,
                                    int64_t batch_outer, int64_t batch_inner, int accumulate) {
    LG_ARG(batch_outer >= 0 && batch_inner >= 0 && batch_inner < (int64_t(1) << 30), "lg_gemm_batched2_f32: bad batch extents");
    if (batch_outer == 0 || batch_inner == 0) return LG_OK;
    return gemm_impl(transA, transB, M, N, K, A, lda, strideA_outer, B, ldb, strideB_outer, C, ldc, strideC_outer,
                     batch_outer * batch_inner, accumulate, nullptr, nullptr, 0, batch_inner, strideA_inner, strideB_inner, strideC_inner);
}

extern "C" int lg_gemm_fused_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                 const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                 int accumulate, const float* bias, float* rowsum, int rowsum_accumulate,
                                 int relu_a, int relu_b) {
    LG_ARG(!(bias && rowsum), "lg_gemm_fused_f32: bias and rowsum exclude each other");
    LG_ARG(!(bias && accumulate), "lg_gemm_fused_f32: bias and accumulate exclude each other");
    return gemm_impl(transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, 1, accumulate, bias, rowsum, rowsum_accumulate,
                     1, 0, 0, 0, relu_a, relu_b);
}

#ifdef LG_GEMM_TIMELINE
// experiments build only: timestamps of the LAST GEMM launch (8 x uint64 per workgroup; 100 MHz wall clock)
extern "C" int lg_debug_gemm_timeline(unsigned long long* host, int max_wgs, int* nwg, int* slices, int* tiles) {
    LG_REQUIRE_INIT();
    LG_HIP(hipStreamSynchronize(rt().stream));
    const int n = lg::g_tl_nwg < max_wgs ? lg::g_tl_nwg : max_wgs;
    if (lg::g_tl_buf && n > 0) LG_HIP(hipMemcpy(host, lg::g_tl_buf, size_t(n) * 64, hipMemcpyDeviceToHost));
    *nwg = n; *slices = lg::g_tl_slices; *tiles = lg::g_tl_tiles;
    return LG_OK;
}
#endif

extern "C" int lg_gemm_pair_begin(void) {
    LG_REQUIRE_INIT();
    PairState& P = lg::pair_state();
    LG_ARG(P.active == 0, "lg_gemm_pair_begin: a pair bracket is already open");
    P.active = 1;
    P.count = 0;
    return LG_OK;
}

extern "C" int lg_gemm_pair_end(void) {
    LG_REQUIRE_INIT();
    PairState& P = lg::pair_state();
    LG_ARG(P.active != 0, "lg_gemm_pair_end: no pair bracket is open");
    const int rc = lg::pair_flush(true);
    P.active = 0;
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
