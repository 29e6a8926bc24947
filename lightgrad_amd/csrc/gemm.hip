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
#include "gelu_common.h"
#include "adam_common.h"
#include "tail_jobs.h"
#include "mse_finalize.h"
#include "head_fwd_body.h"
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
    // optional [M, N] with row pitch ldadd, added last: C = (A @ B + bias) + addend - a residual connection
    // (`dense(h) + h_in`, examples/bert.py:101) or a gradient that already exists (`grad + g @ W`) without a pass of its own
    const float* addend;
    int64_t      ldadd;
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
    short   relu_a, relu_b;
    // an activation in the epilogue (small tiles only; lg_gemm_act_f32).  act = 1: C keeps the pre-activation (product + bias)
    // and aux[m][n] receives gelu of it.  act = 2: the product is multiplied by gelu'(aux[m][n]) - the gelu backward of the
    // tape applied to the input gradient where it is made.  aux: [M, N] with row pitch ldadd (never together with an addend;
    // the argument block of the group launch has no room for a pitch of its own).
    int     act;
    float*  aux;
    // three products of one shape that share op(A), in ONE launch (lg_gemm_multi3_f32; batch = 3): product b reads
    // B + boff[b], adds bias + biasoff[b] and writes C + coff[b] - three separately allocated weights, no packing copy
    int     multi;
    int64_t boff[3], coff[3], biasoff[3];
    // one product whose K runs through three separately allocated B operands of seg_k k-values each (lg_gemm_kseg3_f32:
    // dx = [g0 | g1 | g2] @ [W0; W1; W2]): every seg_steps K-steps the running B pointer jumps to the next operand
    int     seg_k, seg_steps;
    int64_t seg_jump[2];
    // the optimizer's update of the parameter whose gradient this product is, applied in the epilogue (lg_adam_epilogue_arm;
    // small tiles, one K-group): adam_c for C (dense, ldc == N), adam_r for the row sums.  Device pointers, NULL almost always.
    const AdamPlan* adam_c;
    const AdamPlan* adam_r;
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

// the 256x256 tile never splits K (its launches have >= 256 tiles or lose to smaller tiles in the cost model); it also leaves
// out everything else optional - with 64 of its 128 registers per lane holding accumulators it has none to spare (its four
// shipped instantiations use exactly 128 and no scratch: tests/test_gemm_registers.py compiles them and checks)
template <int BM, int BN> constexpr bool kHasExtras = BM * BN < 256 * 256;      // relu operands, row sums: small tiles only
template <int BM, int BN, int KG = 1> constexpr bool kCanSplitK = kHasExtras<BM, BN> && KG == 1;
// the optimizer's update in the epilogue (lg_adam_epilogue_arm): tiles up to 128x128 with one K-group
template <int BM, int BN, int KG = 1> constexpr bool kCanAdam = BM * BN <= 128 * 128 && KG == 1;

// KG = 2: "K-groups" - the split of K happens INSIDE the workgroup.  Twice the waves on the same BM x BN tile; a K-step stages
// BK*KG k-values, wave group g multiplies k in [g*BK, (g+1)*BK) of it, and after the loop group 1 hands its accumulators to
// group 0 through LDS.  For outputs with too few 64x64 tiles to fill the chip this replaces the cross-workgroup split-K (slab
// write + drain + ticket + fold = 2.6 us of a 14.6 us launch, profiles/r2/gemm_timeline_v1.txt) by one LDS exchange: a 64x32
// tile with 2 x 2 waves does per wave exactly the MFMA work of a 64x64 tile with 2 K-slices, one workgroup per CU.
// gather staging (gemm_tile_body.inc): an M- / N-contiguous operand of the 256x256 tile is transposed on its way to LDS.
// EXPERIMENT, off in the shipped library (-DLG_GEMM_GATHER_STAGING builds it): the four dwords of a chunk land in four
// separately allocated registers (an inline-asm load cannot target an element of a register tuple) and packing them for the
// 16-byte LDS store costs the 128-register tile 9-20 spilled registers inside its K loop (round 4, tools: /tmp build of the
// 256x256 instantiations only; profiles/r4/README.md).  The lead that remains: loads the compiler can see (no inline asm), at
// the price of its conservative vmcnt waits, or fragments read as ds_read_b64 along m / n with the accumulator rows permuted.
#ifdef LG_GEMM_GATHER_STAGING
template <int BM, int BN, bool XKC> constexpr bool kGatherStaging = !XKC && BM * BN == 256 * 256;
#else
template <int BM, int BN, bool XKC> constexpr bool kGatherStaging = false;
#endif
template <int BM, int BN, int BK, bool AKC, bool BKC, int KG>
constexpr int gemm_lds_floats() {
    constexpr int BKS = BK * KG;
    constexpr bool ALK = AKC || kGatherStaging<BM, BN, AKC>, BLK = BKC || kGatherStaging<BM, BN, BKC>;
    return 2 * ((ALK ? BM * (BKS + 4) : BKS * BM) + (BLK ? BN * (BKS + 4) : BKS * BN));
}

// The kernel body lives in gemm_tile_body.inc and is included textually: here as the whole kernel, and twice - once per
// product - in the two-product launch below.
// XT: 1 = three products on one op(A) (lg_gemm_multi3_f32), 2 = K through three B operands (lg_gemm_kseg3_f32).  Separate
// instantiations: as run-time branches of the common kernel their few instructions and argument loads cost every GEMM launch
// 0.4 - 1 us (the MLP step 59.9 -> 62.3 us), on the path in front of a workgroup's first global load.
template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, bool VA, bool VB, int PD, int KG = 1, int XT = 0>
__global__ void __launch_bounds__(WM * WN * KG * 64) sgemm_mfma(GemmArgs g) {
#define LG_TILE_OWNS_LDS 1
#define LG_TILE_BID blockIdx.x
#include "gemm_tile_body.inc"
#undef LG_TILE_OWNS_LDS
#undef LG_TILE_BID
}

// The hidden layer's product and the skinny output layer + loss behind it in ONE launch (lg_gemm_bias_head_fwd_f32):
// workgroups [0, g.nwg) are the GEMM's tiles (64x32, two K-groups: the forward product of the MNIST MLP's first layer), the
// rest do head_fwd's rows - four rows each, W2 staged in LDS on arrival - but wait for the tiles of their rows first.  The tiles'
// stores are write-through and each storing wavefront takes the ticket of its row of tiles when they are drained; a row
// workgroup polls that ticket (one lane, bounded: 2 s, then the device status flag), reads its rows with sc1 loads and hands
// the ticket on; the last reader of a row of tiles resets it.  No deadlock: tile workgroups never wait, and row workgroups are
// dispatched after all of them (each XCD hands out its share of the grid in order).
// MEASURED AND NOT USED by the tape (profiles/r4/chain_bench.txt, MNIST MLP shapes, replayed graphs): 22.0 us against 20.6 us for
// the two launches.  The tiles alone take 14.7 us in this form (13.7 with plain stores and no ticket), and the rows finish 7.3 us
// after them - as long as head_fwd takes as a kernel of its own, boundary included: a flag that has to travel through memory, rows
// that have to come from memory (sc1) instead of the L2, and nothing of the rows' work overlaps the tiles'.  What a kernel boundary
// costs on this chip (3 us) is less than what it takes to pass data between workgroups on different XCDs by hand.  Kept behind
// `chain = 1` of the entry point, with its parity test.
struct HeadChain {
    HeadFwd h;
    int*    tickets;        // one per row of tiles, `stride` ints apart; zero on entry and on exit
    int     stride;
    int     per_row;        // arrivals of tile wavefronts per row of tiles: tiles_n * WM * WN
    int*    status;
};
static const HeadChain*& pending_chain() { static const HeadChain* p = nullptr; return p; }     // set by lg_gemm_bias_head_fwd_f32 around its product
constexpr int kChainTicketBase = 16384, kChainTicketStride = 32, kChainRowBlocks = 64;      // (pairs / groups count from 0, LayerNorm's queue from the middle)
template <int OMAX, int PD>
__global__ void __launch_bounds__(256) sgemm_bias_head_fwd(GemmArgs g, HeadChain c) {
    constexpr int BM = 64, BN = 32, BK = 32, WM = 2, WN = 1, KG = 2, XT = 0;
    constexpr bool AKC = true, BKC = true, VA = true, VB = true;
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, AKC, BKC, KG>()];
    if (int(blockIdx.x) < g.nwg) {
        int* const signal_tickets = c.tickets;
        const int signal_stride = c.stride, signal_per_row = c.per_row;
#define LG_TILE_OWNS_LDS 0
#define LG_TILE_BID int(blockIdx.x)
#undef LG_TILE_SIGNAL
#define LG_TILE_SIGNAL 1
#include "gemm_tile_body.inc"
#undef LG_TILE_SIGNAL
#undef LG_TILE_BID
#undef LG_TILE_OWNS_LDS
    } else {
        const HeadFwd& a = c.h;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wn = a.outs * a.hidden;
        for (int i = tid * 4; i < wn; i += 1024) *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(a.w + i);
        const int64_t row0 = int64_t(int(blockIdx.x) - g.nwg) * 4;          // rows row0 .. row0 + 3: one row of tiles (4 divides BM)
        int* const ticket = c.tickets + int(row0 / BM) * c.stride;
        const int64_t block_rows = a.rows - (row0 / BM) * BM < BM ? a.rows - (row0 / BM) * BM : BM;
        const int readers = int((block_rows + 3) / 4);                       // row workgroups of this row of tiles
        int* const flag = ticket + 16;                                       // raised by the wavefront that completes the row of tiles
        int* const done = ticket + 17;                                       // row workgroups that have seen it
        __shared__ int seen;
        if (tid == 0) {
            int ok = 1;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 200000000ull) {
                    __hip_atomic_fetch_or(c.status, LG_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = 0;
                    break;
                }
            }
            seen = ok;
        }
        __syncthreads();                                                     // W2 is staged, the rows' tiles are in memory
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");              // no instruction: keeps the row's loads below the wait
        const int64_t row = row0 + wave;
        if (row < a.rows) head_fwd_row<OMAX, true>(a, lds, row, lane);
        // off the path of the rows' loads: the last row workgroup that has seen the flag lowers it for the next launch (after a
        // give-up the host resets the pool: check_device_status)
        if (tid == 0 && seen) {
            if (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == readers - 1) {
                __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Two independent products in ONE launch: workgroups [0, first.nwg) work on the first, the rest on the second.  Made for the
// pair every Linear's backward produces - dW (+ db) = g^T @ x and dx = g @ W: both have too few 64x64 tiles to give a CU more
// than one workgroup, and a lone workgroup leaves the matrix cores idle during its waits (0.68 us per K-tile against 0.43 us
// of MFMAs); together their workgroups share the CUs and interleave (0.51 us per K-tile each when two are resident), and one
// launch floor (3 us) disappears.  Layouts: the first product A^T-form (M-contiguous A, N-contiguous B), the second
// K-contiguous A and N-contiguous B - what dense row-major g, x, W give.
template <int PD, bool SECOND_BKC = false>
__global__ void __launch_bounds__(256) sgemm_pair_wgrad_xgrad(GemmArgs first, GemmArgs second) {
    constexpr int L1 = gemm_lds_floats<64, 64, 32, false, false, 1>(), L2 = gemm_lds_floats<64, 64, 32, true, SECOND_BKC, 1>();
    __shared__ __attribute__((aligned(16))) float lds[L1 > L2 ? L1 : L2];
    constexpr int BM = 64, BN = 64, BK = 32, WM = 2, WN = 2, KG = 1, XT = 0;
    constexpr bool VA = true, VB = true;
#define LG_TILE_OWNS_LDS 0
    if (int(blockIdx.x) < first.nwg) {
        constexpr bool AKC = false, BKC = false;
        const GemmArgs& g = first;
#define LG_TILE_BID int(blockIdx.x)
#include "gemm_tile_body.inc"
#undef LG_TILE_BID
    } else {
        constexpr bool AKC = true, BKC = SECOND_BKC;
        const GemmArgs& g = second;
#define LG_TILE_BID (int(blockIdx.x) - first.nwg)
#include "gemm_tile_body.inc"
#undef LG_TILE_BID
    }
#undef LG_TILE_OWNS_LDS
}

// THREE products and the scalar of the loss in one launch: what the backward pass of `Linear -> relu -> skinny Linear -> mse`
// (the MNIST MLP of the headline) has left to do once the gradient of the hidden layer exists - the head's weight gradient
// dW2 (+ db2) = err^T @ relu(pre) (10 x 512 over 1024 rows: a handful of tiles, on its own pure launch latency; as slab workgroups
// of head_bwd a 6.6 us dependent chain in front of everything else), the hidden layer's dW1 (+ db1) = g^T @ x, dx = g @ W1 - and
// one spare workgroup that finishes the loss of the forward pass (mse_finalize.h).  Workgroups [0, w[0].nwg) the first
// A^T-form product, then the second, then the K-contiguous-A product, then the loss.  A kernel of its own: the pair's
// workgroups do not pay for the third argument block.
int lg_mse_finalize_job(const float* row_loss, int64_t rows, float inv_n, float* loss);     // head.hip
struct PairFirsts { GemmArgs w[2]; };
struct LossJob {
    const float* row_loss;     // [rows] from head_fwd; NULL: no job
    float*       loss;
    int64_t      rows;
    float        inv_n;
};
template <int PD>
__global__ void __launch_bounds__(256) sgemm_triple_wgrad2_xgrad(PairFirsts firsts, GemmArgs second, LossJob job) {
    constexpr int L1 = gemm_lds_floats<64, 64, 32, false, false, 1>(), L2 = gemm_lds_floats<64, 64, 32, true, false, 1>();
    __shared__ __attribute__((aligned(16))) float lds[L1 > L2 ? L1 : L2];
    constexpr int BM = 64, BN = 64, BK = 32, WM = 2, WN = 2, KG = 1, XT = 0;
    constexpr bool VA = true, VB = true;
#define LG_TILE_OWNS_LDS 0
    const int n0w = firsts.w[0].nwg, n1w = firsts.w[1].nwg;
    if (int(blockIdx.x) < n0w + n1w) {
        constexpr bool AKC = false, BKC = false;
        const int which = int(blockIdx.x) >= n0w ? 1 : 0;              // uniform: scalar loads from the chosen block
        const GemmArgs g = firsts.w[which];
#define LG_TILE_BID (int(blockIdx.x) - (which ? n0w : 0))
#include "gemm_tile_body.inc"
#undef LG_TILE_BID
    } else if (int(blockIdx.x) < n0w + n1w + second.nwg) {
        constexpr bool AKC = true, BKC = false;
        const GemmArgs& g = second;
#define LG_TILE_BID (int(blockIdx.x) - n0w - n1w)
#include "gemm_tile_body.inc"
#undef LG_TILE_BID
    } else {
        finalize_loss(job.row_loss, job.rows, job.inv_n, job.loss, lds);
    }
#undef LG_TILE_OWNS_LDS
}

// Up to kGroupMax independent products of ONE layout in one launch: the weight gradients dW (+ db) = g^T @ x of the Linear
// layers of a deep network (M-contiguous A, N-contiguous B).  Each of them is a small output with a long K - 12 us alone, 7 of
// them launch, prologue, split-K hand-off and epilogue, with a handful of workgroups on a 256-CU chip; queued during the
// backward pass and launched together at its end they cost what the largest one costs.
constexpr int kGroupMax = 14;
// what a queued product needs of GemmArgs (one matrix, no bias / addend / activation / relu): the group's argument block has
// to hold 14 of them AND the jobs that ride along, within the 4 KiB a kernel may take
struct GroupProduct {
    const float* A;
    const float* B;
    float*       C;
    float*       W;
    int*         tickets;
    float*       rowsum;
    int64_t      M, N, K, lda, ldb, ldc, k_per_slice;
    int          tiles_m, tiles_n, nwg, accumulate, group_m, k_slices, rowsum_accumulate, k_tail;
    FastDiv      div_per_batch, div_slices, div_gspan, div_group, div_last_group, div_batch_inner;
};
static GroupProduct pack_product(const GemmArgs& g) {
    return GroupProduct{g.A, g.B, g.C, g.W, g.tickets, g.rowsum, g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.k_per_slice,
                        g.tiles_m, g.tiles_n, g.nwg, g.accumulate, g.group_m, g.k_slices, g.rowsum_accumulate, g.k_tail,
                        g.div_per_batch, g.div_slices, g.div_gspan, g.div_group, g.div_last_group, g.div_batch_inner};
}
__device__ __forceinline__ GemmArgs unpack_product(const GroupProduct& p) {
    GemmArgs g{};
    g.A = p.A; g.B = p.B; g.C = p.C; g.W = p.W; g.tickets = p.tickets; g.rowsum = p.rowsum;
    g.M = p.M; g.N = p.N; g.K = p.K; g.lda = p.lda; g.ldb = p.ldb; g.ldc = p.ldc; g.k_per_slice = p.k_per_slice;
    g.tiles_m = p.tiles_m; g.tiles_n = p.tiles_n; g.nwg = p.nwg; g.accumulate = p.accumulate; g.group_m = p.group_m;
    g.k_slices = p.k_slices; g.rowsum_accumulate = p.rowsum_accumulate; g.k_tail = p.k_tail;
    g.div_per_batch = p.div_per_batch; g.div_slices = p.div_slices; g.div_gspan = p.div_gspan; g.div_group = p.div_group;
    g.div_last_group = p.div_last_group; g.div_batch_inner = p.div_batch_inner;
    g.batch_inner = 1;
    return g;
}

struct GemmGroup {
    GroupProduct p[kGroupMax];
    int      first[kGroupMax + 1];      // first workgroup of each product (after the other roles' workgroups); first[count] = all of them
    int      count;
    // one more job may ride along: column sums of a dense [rows, cols] matrix - the bias gradient of a Linear whose row-sum
    // column would cost a whole extra tile column in its weight-gradient product (BERT's decoder: db = column sums of the
    // (1024, 30522) logits gradient, 125 MB to read).  Memory-bound workgroups next to MFMA-bound ones: the 27 us of the
    // separate reduction hide inside the group's launch.  cs_wgs workgroups of 64 columns each, at the FRONT of the grid.
    const float* cs_in;
    float*       cs_out;
    int64_t      cs_rows, cs_cols, cs_ld;
    int          cs_accumulate, cs_wgs;
    // and the LayerNorm parameter gradients / embedding scatter-adds queued beside the products (tail_jobs.h): tail_wgs
    // workgroups at the end of the grid
    int          tail_wgs;
    TailGroup    tail;
};

static_assert(sizeof(GemmGroup) <= 4096, "the group travels as kernel arguments");

template <int PD>
__global__ void __launch_bounds__(256) sgemm_group_wgrad(GemmGroup grp) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<64, 64, 32, false, false, 1>()];
    constexpr int BM = 64, BN = 64, BK = 32, WM = 2, WN = 2, KG = 1, XT = 0;
    constexpr bool VA = true, VB = true, AKC = false, BKC = false;
    if (int(blockIdx.x) < grp.cs_wgs) {
        // column-sum role: 64 columns per workgroup, four row groups of 64 threads (row group q takes rows q, q + 4, ...), eight
        // loads in flight per thread; the four partial sums are combined in a fixed order through LDS
        const int q = threadIdx.x >> 6, cl = threadIdx.x & 63;
        const int64_t col = int64_t(blockIdx.x) * 64 + cl;
        const bool live = col < grp.cs_cols;
        const float* in = grp.cs_in + (live ? col : 0);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t r = q;
        for (; r + 28 < grp.cs_rows; r += 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += in[(r + 4 * e) * grp.cs_ld];
        }
        for (; r < grp.cs_rows; r += 4) acc[0] += in[r * grp.cs_ld];
        lds[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (q == 0 && live) {
            const float v = (lds[cl] + lds[64 + cl]) + (lds[128 + cl] + lds[192 + cl]);
            grp.cs_out[col] = grp.cs_accumulate ? grp.cs_out[col] + v : v;
        }
        return;
    }
    const int bid = int(blockIdx.x) - grp.cs_wgs;
    if (bid >= grp.first[grp.count]) {
        // LayerNorm parameter gradients / embedding scatter-adds, BEHIND the products: they take the slots the last products
        // leave (in front of them they held every CU for their ~15 us first: 118 us for the launch instead of 102 + 34 apart)
        tail_group_body(grp.tail, bid - grp.first[grp.count]);
        return;
    }
    int which = 0;
    while (which + 1 < grp.count && bid >= grp.first[which + 1]) ++which;      // uniform: scalar loads
    const GemmArgs g = unpack_product(grp.p[which]);
#define LG_TILE_OWNS_LDS 0
#define LG_TILE_BID (bid - grp.first[which])
#include "gemm_tile_body.inc"
#undef LG_TILE_BID
#undef LG_TILE_OWNS_LDS
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, int KG>
static void launch_layout(const GemmArgs& g, bool va, bool vb) {
    dim3 grid(g.nwg), block(WM * WN * KG * 64);
    hipStream_t s = rt().stream;
    // K-tiles in flight between global memory and LDS.  The large tiles ran with ONE until the end of round 3 ("enough MFMAs per
    // K-tile to cover the latency"): with two, 4096^3 NN 140.9 -> 143.3, NT 143.0 -> 146.6 TFLOP/s on the 256 x 256 tile (TN 138,
    // unchanged: both its operands come out of LDS as single floats), 132 -> 136-137 on the 128 x 128 tile.
    //
    // NOT three on the 256 x 256 tile, and why (the "memory access fault" of round 3's experiment): the ring
    // registers are written by inline-asm buffer loads that the compiler believes complete at once.  A third slot makes the
    // ring 48 registers next to 64 accumulators - more than the 128 a lane of a 1024-thread workgroup has - so the allocator
    // SPILLS ring registers (the experiment's 112-192 bytes of scratch): it stores such a register right behind the asm
    // statement, i.e. BEFORE the load has landed, hands the physical register to another value - an LDS or global address, a
    // loop bound - and the load then lands on top of that value.  A wild address is the fault; the reloaded "tile" is garbage
    // even when nothing faults.  The ring's slot arithmetic and vmcnt counts are sound for any PD (the wait in front of
    // tile kt+1 always leaves at most PD-2 younger tiles = (PD-2)*NL loads outstanding; gemm_tile_body.inc).  So the rule is:
    // a kernel whose ring is in flight across compiler-scheduled code must not spill - the shipped 256 x 256 instantiations
    // are checked for zero scratch (tests/test_gemm_registers.py), and deeper rings on this tile do not build:
    constexpr int PD = (BM * BN <= 64 * 64) ? kSmallTilePrefetch : 2;
    static_assert(BM * BN < 256 * 256 || PD <= 2, "a third K-tile in flight does not fit the 256x256 tile's 128 registers: ring registers would spill while their loads are in flight");
    if constexpr (kHasExtras<BM, BN>) {
        // (the entry points have checked: multi comes K-contiguous on both sides, seg_k with an N-contiguous B, float4 staging)
        if constexpr (AKC && BKC)  { if (g.multi) { hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, true, PD, KG, 1>), grid, block, 0, s, g); return; } }
        if constexpr (AKC && !BKC) { if (g.seg_k) { hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, true, PD, KG, 2>), grid, block, 0, s, g); return; } }
    }
    if constexpr (!kHasExtras<BM, BN>) {
        // the 256x256 tile exists in its float4 / gather form only (gemm_impl: va is always set, vb unless row sums ride along -
        // and those never take this tile)
        hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, true, PD, KG>), grid, block, 0, s, g);
        return;
    }
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
    int      active = 0;       // 0: no bracket; 1: collecting; 2: bracket open but no longer collecting; 3: HELD (lg_gemm_pair_hold):
                               //    what has been collected waits, other products launch as if there were no bracket
    int      count = 0;
    bool     second_bkc = false;     // layout of the K-contiguous-A product's B: K-contiguous (g @ v^T of attention) or N-contiguous (g @ W)
    // what may be collected: [F], [F, S], [F, F], [F, F, S] - F an A^T-form product (M-contiguous A, N-contiguous B: g^T @ x,
    // probs^T @ dO), S one with a K-contiguous A (g @ W, dS @ k; dO @ v^T).  [F, S] leaves as sgemm_pair_wgrad_xgrad, [F, F, S]
    // as sgemm_triple_wgrad2_xgrad, anything else as single launches in call order.
    GemmArgs args[3];
    bool     is_s[3] = {false, false, false};
    int64_t  tiles = 0;        // tickets handed out so far
    LossJob  job{};            // lg_gemm_pair_mse_loss: finished by a spare workgroup of the three-product launch, else by a launch of its own
};
static PairState& pair_state() { static PairState p; return p; }

static void pair_launch_single(const GemmArgs& g, bool is_s) {
    dim3 grid(g.nwg), block(256);
    constexpr int PD = kSmallTilePrefetch;
    if (!is_s) hipLaunchKernelGGL((sgemm_mfma<64, 64, 32, 2, 2, false, false, true, true, PD, 1>), grid, block, 0, rt().stream, g);
    else if (pair_state().second_bkc) hipLaunchKernelGGL((sgemm_mfma<64, 64, 32, 2, 2, true, true, true, true, PD, 1>), grid, block, 0, rt().stream, g);
    else           hipLaunchKernelGGL((sgemm_mfma<64, 64, 32, 2, 2, true, false, true, true, PD, 1>), grid, block, 0, rt().stream, g);
}

// The K-slice count of an A^T-form product of a shared launch (the weight gradient, the one that splits K) was chosen as if it had
// the chip to itself; its workgroups share the CUs with the other products'.  Once all are known the count is chosen again from the same
// microsecond model with the joint residency: K-tiles per workgroup x (0.43 c + 0.17) us for c = ceil(all workgroups of the launch /
// CUs) resident per CU, + 1.8 + 0.85 slices for publishing and folding the slabs.  MNIST MLP (104 + 208 tiles, K = 1024): 5 slices
// instead of 4 (728 workgroups: still three per CU; six would make four): the step 60.6 -> 59.6 us
// (profiles/r4/pair_slices_sweep2.txt).  Batched products keep what they have.
static int pair_retune(GemmArgs& g, int64_t others_nwg) {
    const int64_t tiles = int64_t(g.tiles_m) * g.tiles_n;
    static const char* pair_slices_env = getenv("LG_GEMM_PAIR_SLICES");      // experiments: the forced counts stay
    if (pair_slices_env || g.nwg != tiles * g.k_slices || g.seg_k || tiles >= 256 || g.K < 128) return LG_OK;
    constexpr int BK = 32, BM = 64, BN = 64;
    const double cus = rt().compute_units > 0 ? rt().compute_units : 256;
    const int64_t k_tiles = (g.K + BK - 1) / BK;
    int64_t best_sl = g.k_slices;
    double best = 1e30;
    for (int64_t sl = 1; sl <= 64 && (sl == 1 || sl * 2 <= k_tiles); ++sl) {
        const double per_cu = double(tiles * sl + others_nwg) / cus;
        const double c = per_cu > 1.0 ? double(int64_t(per_cu + 0.999)) : 1.0;
        const double cost = double((k_tiles + sl - 1) / sl) * (c <= 1.0 ? 0.68 : 0.43 * c + 0.17) + (sl > 1 ? 1.8 + 0.85 * double(sl) : 0.0);
        if (cost < best) { best = cost; best_sl = sl; }
    }
    int64_t k_per_slice = ((g.K + best_sl - 1) / best_sl + BK - 1) / BK * BK;
    best_sl = (g.K + k_per_slice - 1) / k_per_slice;
    if (best_sl == g.k_slices) return LG_OK;
    if (g.W) { const int rc = lg_free(g.W); g.W = nullptr; if (rc != LG_OK) return rc; }
    g.k_per_slice = k_per_slice;
    g.k_slices = int(best_sl);
    g.nwg = int(tiles * best_sl);
    g.div_slices = make_fastdiv(best_sl);
    if (best_sl > 1) {
        const int rc = lg_malloc(reinterpret_cast<void**>(&g.W), size_t(tiles * best_sl) * BM * BN * sizeof(float));
        if (rc != LG_OK) return rc;
    }
    return LG_OK;
}

static int pair_flush(bool keep_collecting) {
    PairState& P = pair_state();
    int rc = LG_OK;
    bool job_done = false;
    if (P.count == 3) {
        // (the first product - the skinny head's dW2 - keeps its own slice count: a few tiles whatever it is)
        rc = pair_retune(P.args[1], int64_t(P.args[0].nwg) + P.args[2].nwg);
        if (rc != LG_OK) { P.count = 0; return rc; }
        PairFirsts firsts;
        firsts.w[0] = P.args[0];
        firsts.w[1] = P.args[1];
        const int spare = P.job.loss ? 1 : 0;
        hipLaunchKernelGGL((sgemm_triple_wgrad2_xgrad<kSmallTilePrefetch>), dim3(P.args[0].nwg + P.args[1].nwg + P.args[2].nwg + spare), dim3(256), 0,
                           rt().stream, firsts, P.args[2], P.job);
        job_done = true;
    } else if (P.count == 2 && P.is_s[1]) {
        rc = pair_retune(P.args[0], P.args[1].nwg);
        if (rc != LG_OK) { P.count = 0; return rc; }
#ifdef LG_GEMM_TIMELINE
        if (P.args[0].tl) {
            (void)hipMemsetAsync(P.args[0].tl, 0, size_t(P.args[0].nwg + P.args[1].nwg) * 64, rt().stream);
            lg_debug_timeline_state(P.args[0].tl, P.args[0].nwg + P.args[1].nwg, P.args[0].k_slices * 100 + P.args[1].k_slices, P.args[0].nwg);
        }
#endif
        if (P.second_bkc)
            hipLaunchKernelGGL((sgemm_pair_wgrad_xgrad<kSmallTilePrefetch, true>), dim3(P.args[0].nwg + P.args[1].nwg), dim3(256), 0, rt().stream,
                               P.args[0], P.args[1]);
        else
            hipLaunchKernelGGL((sgemm_pair_wgrad_xgrad<kSmallTilePrefetch, false>), dim3(P.args[0].nwg + P.args[1].nwg), dim3(256), 0, rt().stream,
                               P.args[0], P.args[1]);
    } else {
        for (int i = 0; i < P.count; ++i) pair_launch_single(P.args[i], P.is_s[i]);
    }
    for (int i = 0; i < P.count; ++i)
        if (P.args[i].W) { const int r = lg_free(P.args[i].W); if (r != LG_OK) rc = r; }       // stream-ordered: reused by later launches only
    P.count = 0;
    P.tiles = 0;
    if (job_done) P.job = LossJob{};
    if (!keep_collecting && P.active) P.active = 2;
    return rc;
}

// true: `g` (fully prepared, workspace allocated) has been taken over and will be launched by lg_gemm_pair_end
static bool pair_try_defer(GemmArgs& g, bool akc, bool bkc, bool va, bool vb, int64_t batch) {
    PairState& P = pair_state();
    if (P.active != 1) return false;
    const int slot = P.count;
    // Batched products (attention: one matrix per (batch, head)) share launches like single ones.
    const bool f_form = !akc && !bkc, s_form = akc;
    bool fits = slot < 3 && va && vb && batch >= 1;
    if (slot == 0)      fits = fits && f_form;
    else if (slot == 1) fits = fits && !P.is_s[0] && (f_form || s_form);
    else                fits = fits && !P.is_s[1] && s_form && !bkc && batch == 1 && !g.relu_a && !g.relu_b;
    // (three share a launch only as single matrices, the third with an N-contiguous B: what the MLP's backward produces)
    if (slot == 1 && f_form) fits = fits && batch == 1 && P.args[0].nwg == int64_t(P.args[0].tiles_m) * P.args[0].tiles_n * P.args[0].k_slices;
    const int64_t tiles = int64_t(g.tiles_m) * g.tiles_n * batch;
    if (!fits || P.tiles + tiles > rt().n_gemm_tickets) return false;
    g.tickets = rt().gemm_tickets + P.tiles;               // the products fold their K-slices with disjoint tickets
    P.tiles += tiles;
    if (s_form) P.second_bkc = bkc;
    P.is_s[slot] = s_form;
    P.args[slot] = g;
    P.count = slot + 1;
    return true;
}

// ---- many weight-gradient products in one launch (lg_gemm_group_begin / _end / _flush) ----------------------------------
// Between begin and end, a product that resolves to the 64x64 tile in the A^T-form layout (M-contiguous A, N-contiguous B,
// one matrix, at most kGroupTiles output tiles) is prepared and QUEUED; the queue outlives the bracket and goes out as one
// launch on lg_gemm_group_flush, when it is full, when a product arrives that writes where a queued one writes (a weight
// shared by two layers), and before anything that lets the host or another graph see results (lg_sync, device-to-host
// copies, graph launch, the end of a capture).  The CALLER keeps the operands alive and unchanged, and does not read the
// results, until then.
constexpr int kGroupTiles = 1024;
struct GroupState {
    int       active = 0;
    int       count = 0;
    int64_t   tiles = 0;          // tickets handed to the queued products
    GemmArgs  queued[kGroupMax];  // the products as prepared; packed into grp when they leave
    GemmGroup grp;
};
static GroupState& group_state() { static GroupState s; return s; }

static int group_flush() {
    GroupState& G = group_state();
    if (G.count == 0) {
        if (G.grp.cs_wgs > 0) {                       // column sums queued with nothing to ride on: the plain reduction
            const int64_t shape[2] = {G.grp.cs_rows, G.grp.cs_cols}, strides[2] = {G.grp.cs_ld, 1};
            G.grp.cs_wgs = 0;
            return lg_reduce_acc(LG_RED_SUM, 2, shape, G.grp.cs_in, strides, 1u, G.grp.cs_out, G.grp.cs_accumulate);
        }
        return LG_OK;
    }
    int rc = LG_OK;
    if (G.count == 1 && G.grp.cs_wgs == 0) {
        pair_launch_single(G.queued[0], 0);
    } else {
        // small products first: theirs are the long dependency chains (few workgroups, split-K hand-off and fold), a product
        // with hundreds of tiles (BERT's decoder) fills the chip behind them
        GemmArgs sorted[kGroupMax];
        int n = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < G.count; ++i)
                if ((G.queued[i].nwg > 256) == (pass == 1)) sorted[n++] = G.queued[i];
        G.grp.first[0] = 0;
        for (int i = 0; i < G.count; ++i) { G.queued[i] = sorted[i]; G.grp.p[i] = pack_product(sorted[i]); G.grp.first[i + 1] = G.grp.first[i] + sorted[i].nwg; }
        G.grp.count = G.count;
        // the LayerNorm parameter gradients and embedding scatter-adds queued beside the products ride in this launch - unless
        // one of them writes where a product writes (a tied table: its scatter must ADD to what the product stores, in call order)
        bool ride = true;
        for (int i = 0; i < G.count; ++i) ride = ride && !ln_group_writes(G.queued[i].C) && !ln_group_writes(G.queued[i].rowsum);
        ride = ride && !ln_group_writes(G.grp.cs_wgs > 0 ? G.grp.cs_out : nullptr);
        G.grp.tail_wgs = 0;
        if (ride && tail_take(&G.grp.tail)) {
            G.grp.tail_wgs = G.grp.tail.first[G.grp.tail.ln.count + G.grp.tail.n_scatter];
        }
        hipLaunchKernelGGL((sgemm_group_wgrad<kSmallTilePrefetch>), dim3(G.grp.cs_wgs + G.grp.tail_wgs + G.grp.first[G.count]), dim3(256), 0, rt().stream, G.grp);
        if (G.grp.tail_wgs > 0) { const int r = tail_taken(); if (r != LG_OK) rc = r; }
    }
    for (int i = 0; i < G.count; ++i)
        if (G.queued[i].W) { const int r = lg_free(G.queued[i].W); if (r != LG_OK) rc = r; }     // stream-ordered
    G.count = 0;
    G.tiles = 0;
    G.grp.cs_wgs = 0;
    return rc;
}

// true: `g` (fully prepared, workspace allocated) has been queued
static bool group_try_defer(GemmArgs& g, bool akc, bool bkc, bool va, bool vb, int64_t batch, int& rc) {
    GroupState& G = group_state();
    rc = LG_OK;
    if (G.active != 1 || pair_state().active) return false;
    const int64_t tiles = int64_t(g.tiles_m) * g.tiles_n;
    if (akc || bkc || !va || !vb || batch != 1 || tiles > kGroupTiles || g.relu_a || g.relu_b || g.multi || g.seg_k) return false;
    if (g.adam_c || g.adam_r) return false;              // (the group's argument block does not carry optimizer plans)
    bool clash = false;
    for (int i = 0; i < G.count; ++i)
        clash = clash || G.queued[i].C == g.C || (g.rowsum && G.queued[i].rowsum == g.rowsum);
    if (ln_group_writes(g.C) || ln_group_writes(g.rowsum)) {
        // an embedding scatter / LayerNorm gradient queued EARLIER writes the same buffer: everything queued leaves now, in call order
        rc = gemm_group_flush_pending();
        if (rc != LG_OK) return false;
    } else if (clash || G.count == kGroupMax || G.tiles + tiles > rt().n_gemm_tickets / 2) {     // (the upper half: LayerNorm's queue)
        rc = group_flush();
        if (rc != LG_OK) return false;
    }
    g.tickets = rt().gemm_tickets + G.tiles;            // disjoint tickets: the products fold their K-slices side by side
    G.tiles += tiles;
    G.queued[G.count] = g;
    G.count += 1;
    return true;
}

bool gemm_group_is_open() { return group_state().active == 1; }

int gemm_group_flush_pending() {
    // (a pair bracket that stays open across tape nodes - the skinny head's weight gradient waiting for the hidden layer's
    // products - holds prepared launches too: whoever is about to look at results gets them launched, the bracket goes on collecting)
    if (pair_state().count) { const int prc = pair_flush(true); if (prc != LG_OK) return prc; }
    const int rc = group_flush();
    const int rc2 = ln_group_flush_pending();
    return rc != LG_OK ? rc : rc2;
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
        int sv[3] = {0, 0, 0};
        const int got = sscanf(pair_slices_env, "%d,%d,%d", &sv[0], &sv[1], &sv[2]);      // "<first>,<second>[,<third>]" by position in the bracket
        if (got >= 2) { const int c = pair_state().count; const int v = c < got ? sv[c] : sv[got - 1]; if (v >= 1) slices = v; }
    }
    if (g.seg_k) { slices = 1; g.seg_steps = g.seg_k / (BK * KG); }      // (seg_k is a multiple of 64 = the widest K-step)
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
        int grc = LG_OK;
        if (group_try_defer(g, akc, bkc, va, vb, batch, grc)) return LG_OK;
        if (grc != LG_OK) return grc;
    }
    if (pair_state().count && pair_state().active != 3) {      // something else inside a pair bracket: what is pending goes first
        const int prc = pair_flush(false);
        if (prc != LG_OK) return prc;
    }
    if constexpr (BM == 64 && BN == 32 && WM == 2 && WN == 1 && KG == 2) {
        if (pending_chain() && akc && bkc && va && vb && batch == 1 && !g.rowsum && !g.addend && !g.act && !g.relu_a && !g.relu_b) {
            // lg_gemm_bias_head_fwd_f32: the output layer's rows wait at the end of THIS launch
            HeadChain c = *pending_chain();
            pending_chain() = nullptr;
            c.per_row = g.tiles_n * WM * WN;
            static const char* no_rows_env = getenv("LG_CHAIN_NO_ROWS");           // experiments: the tiles alone (results incomplete)
            const int row_wgs = (no_rows_env && atoi(no_rows_env) == 1) ? 0 : int((c.h.rows + 3) / 4);
            dim3 grid(g.nwg + row_wgs), block(256);
            constexpr int PD = 2;
            const int omax = c.h.outs <= 4 ? 4 : (c.h.outs <= 8 ? 8 : (c.h.outs <= 10 ? 10 : 16));
            if (omax == 4)       hipLaunchKernelGGL((sgemm_bias_head_fwd<4, PD>), grid, block, 0, rt().stream, g, c);
            else if (omax == 8)  hipLaunchKernelGGL((sgemm_bias_head_fwd<8, PD>), grid, block, 0, rt().stream, g, c);
            else if (omax == 10) hipLaunchKernelGGL((sgemm_bias_head_fwd<10, PD>), grid, block, 0, rt().stream, g, c);
            else                 hipLaunchKernelGGL((sgemm_bias_head_fwd<16, PD>), grid, block, 0, rt().stream, g, c);
            return LG_OK;
        }
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
                     int relu_a = 0, int relu_b = 0, const float* addend = nullptr, int64_t ldadd = 0,
                     int act = 0, float* aux = nullptr, int64_t ldaux = 0, const GemmArgs* extras = nullptr) {
    LG_REQUIRE_INIT();
    LG_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "lg_gemm_f32: negative extent (M=%lld N=%lld K=%lld batch=%lld)",
           (long long)M, (long long)N, (long long)K, (long long)batch);
    if (M == 0 || N == 0 || batch == 0) return LG_OK;
    LG_ARG(A && B && C, "lg_gemm_f32: NULL operand");
    LG_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
           "lg_gemm_f32: leading dimension too small (lda=%lld ldb=%lld ldc=%lld for M=%lld N=%lld K=%lld tA=%d tB=%d)",
           (long long)lda, (long long)ldb, (long long)ldc, (long long)M, (long long)N, (long long)K, transA, transB);
    LG_ARG(rowsum == nullptr || (batch == 1 && bias == nullptr), "lg_gemm_rowsum_f32: one matrix product, no bias");
    LG_ARG(addend == nullptr || (batch == 1 && !accumulate && rowsum == nullptr && ldadd >= N && K > 0),
           "lg_gemm_addend_f32: one matrix product with K > 0, no accumulate / row sums, ldadd >= N");
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
    g.addend = addend;
    g.ldadd = ldadd;
    g.rowsum = rowsum;
    g.rowsum_accumulate = rowsum_accumulate;
    g.k_tail = (K % 4 != 0) ? 1 : 0;
    g.relu_a = relu_a; g.relu_b = relu_b;
    g.act = act; g.aux = aux;
    if (act) g.ldadd = ldaux;
    LG_ARG(act == 0 || (batch == 1 && rowsum == nullptr && addend == nullptr && aux != nullptr && ldaux >= N && K > 0 && (act == 1 || act == 2)),
           "lg_gemm_act_f32: one matrix product with K > 0, act 1 or 2, aux [M, N] with ldaux >= N");
    if (extras) {               // (lg_gemm_multi3_f32 / lg_gemm_kseg3_f32 have checked their own arguments)
        g.multi = extras->multi;
        for (int i = 0; i < 3; ++i) { g.boff[i] = extras->boff[i]; g.coff[i] = extras->coff[i]; g.biasoff[i] = extras->biasoff[i]; }
        g.seg_k = extras->seg_k;
        g.seg_jump[0] = extras->seg_jump[0]; g.seg_jump[1] = extras->seg_jump[1];
    }
    // the optimizer's update in this launch's epilogue: a single product that OVERWRITES an armed gradient (optim.hip)
    {
        int arc = LG_OK;
        const bool plain = batch == 1 && ldc == N && bias == nullptr && addend == nullptr && act == 0 && !g.multi && !g.seg_k;
        g.adam_c = adam_epilogue_take(C, M * N, plain ? accumulate : 1, &arc);
        if (arc != LG_OK) return arc;
        if (rowsum) {
            g.adam_r = adam_epilogue_take(rowsum, M, plain ? rowsum_accumulate : 1, &arc);
            if (arc != LG_OK) return arc;
        }
    }
    const bool has_adam = g.adam_c != nullptr || g.adam_r != nullptr;
    const bool fused_extras = rowsum != nullptr || relu_a || relu_b || act != 0 || g.multi || g.seg_k || has_adam;     // not compiled into the 256x256 tile
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
    if ((M <= 32 || N <= 32) && lg::pair_state().active == 1 && lg::pair_state().count < 3 && !akc && !bkc && batch == 1) {
        // a skinny A^T-form product inside a collecting bracket (the head's dW2 = err^T @ relu(pre): 10 x 512): the 64x64 tile,
        // the only one that shares a launch - on its own tile it would be a launch of its own
        rc = launch_config<64, 64, 32, 2, 2>(g, akc, bkc, va, vb, batch);
    } else if (N <= 32) {
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
            // inside a pair bracket the 64x64 tile stays: only it can share a launch, and for the small products of a
            // transformer layer's Linear backward that is worth more than the K-group tile (tiny-BERT, 12 Linear layers:
            // 0.980 -> 0.901 ms per forward+backward).  LG_GEMM_PAIR_KG=1 (experiments) lets the K-group tile win again.
            static const char* pair_kg_env = getenv("LG_GEMM_PAIR_KG");
            const bool keep_for_pair = (lg::pair_state().active == 1 && !(pair_kg_env && atoi(pair_kg_env) == 1))
                                       || (lg::group_state().active == 1 && !akc && !bkc && nblocks(64, 64) <= lg::kGroupTiles);
            if (tile == 9 && batch == 1 && !keep_for_pair && !has_adam) {
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
                // ... and when even 32x32 tiles give a CU no more than two workgroups: FOUR K-groups of 16 k on 32x32 tiles - twice the
                // workgroups, each wave the same MFMA work; two workgroups per CU fill each other's waits (the forward product of the
                // MNIST MLP: 13.16 -> 12.43 us, profiles/r4/fwd_tile_32x32_kgroups.txt; K-groups of 32 k: 13.0)
                if ((tile == 7 || tile == 8) && nblocks(32, 32) <= 2 * cus && !lg::pending_chain()) tile = 5;      // (the chained experiment is built on tile 7)
            }
        }
        if (has_adam && (tile == 1 || tile == 2 || tile == 5 || tile == 7 || tile == 8)) tile = 9;       // (forced by LG_GEMM_TILE: kCanAdam tiles only)
        switch (tile) {
            case 1:  rc = launch_config<256, 128, 32, 4, 2>(g, akc, bkc, va, vb, batch); break;   // 8 waves (experiments only)
            case 2:
                if (!fused_extras) rc = launch_config<256, 256, 32, 4, 4>(g, akc, bkc, va, vb, batch);
                else               rc = launch_config<64, 64, 32, 2, 2>(g, akc, bkc, va, vb, batch);      // extras are not compiled into the big tile
                break;
            case 5:  rc = launch_config<32, 32, 16, 1, 1, 4>(g, akc, bkc, va, vb, batch); break;   // four K-groups inside the workgroup
            case 7:  rc = launch_config<64, 32, 32, 2, 1, 2>(g, akc, bkc, va, vb, batch); break;   // K split inside the workgroup
            case 8:  rc = launch_config<32, 64, 32, 1, 2, 2>(g, akc, bkc, va, vb, batch); break;
            // (measured and not kept in round 3: <64, 32, 16, 2, 1, 4> - four K-groups of 16 k, two waves per SIMD, the K-group
            //  exchange of gemm_tile_body.inc takes any KG: 42 / 42 parity tests green, fwd1 14.67 us against 14.42)
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

extern "C" int lg_head_fwd_grad_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                                    float* y, float* err, float* row_loss, float* dx, float* gpre, int64_t rows, int64_t hidden, int64_t outs);

extern "C" int lg_gemm_bias_head_fwd_f32(const float* x, int64_t ldx, const float* w1, int64_t ldw1, const float* b1, float* pre,
                                         int64_t rows, int64_t hidden, int64_t d_in, int relu,
                                         const float* w2, const float* b2, const float* target, float* y, float* err, float* row_loss,
                                         float* dx, float* gpre, int64_t outs, int chain, int* launches) {
    LG_REQUIRE_INIT();
    LG_ARG(x && w1 && b1 && pre && w2 && target && y && err && row_loss, "lg_gemm_bias_head_fwd_f32: NULL pointer");
    LG_ARG(rows > 0 && hidden > 0 && d_in > 0 && outs > 0 && outs <= 16, "lg_gemm_bias_head_fwd_f32: need rows, hidden, d_in > 0 and 1 <= outs <= 16");
    LG_ARG((dx == nullptr) == (gpre == nullptr) && (gpre == nullptr || relu), "lg_gemm_bias_head_fwd_f32: dx and gpre go together and need relu != 0");
    LG_ARG(hidden % 4 == 0 && aligned16(pre) && aligned16(w2) && outs * hidden * 4 <= 64 * 1024,
           "lg_gemm_bias_head_fwd_f32: hidden a multiple of 4, pre and w2 16-byte aligned, w2 within 64 KiB (lg_head_fwd_f32's conditions)");
    HeadChain c{};
    c.h.x = pre; c.h.w = w2; c.h.bias = b2; c.h.target = target; c.h.y = y; c.h.err = err; c.h.row_loss = row_loss; c.h.gpre = gpre; c.h.dx = dx;
    c.h.rows = rows; c.h.ldx = hidden; c.h.hidden = int(hidden); c.h.outs = int(outs); c.h.relu = relu;
    c.tickets = rt().gemm_tickets + kChainTicketBase;
    c.stride = kChainTicketStride;
    c.status = rt().status_dev;
    // one launch when the rows fit the row workgroups' registers and tickets, W2 fits the LDS the product's tile owns, and the product
    // resolves to the tile this launch is built on (gemm_impl's cost model: 64x32 with two K-groups); else two launches
    constexpr int kLdsFloats = gemm_lds_floats<64, 32, 32, true, true, 2>();
    const bool can_chain = chain != 0 && hidden <= 1024 && outs * hidden <= kLdsFloats && rows <= int64_t(kChainRowBlocks) * 64
                           && kChainTicketBase + kChainRowBlocks * kChainTicketStride <= rt().n_gemm_tickets / 2 && aligned16(dx) && aligned16(gpre);
    if (can_chain) pending_chain() = &c;
    const int rc = gemm_impl(0, 1, rows, hidden, d_in, x, ldx, 0, w1, ldw1, 0, pre, hidden, 0, 1, 0, b1);
    const bool chained = can_chain && pending_chain() == nullptr;
    pending_chain() = nullptr;
    if (rc != LG_OK) return rc;
    if (launches) *launches = chained ? 1 : 2;
    if (chained) return LG_OK;
    return lg_head_fwd_grad_f32(pre, hidden, relu, w2, b2, target, y, err, row_loss, dx, gpre, rows, hidden, outs);
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
    int rc = lg::pair_flush(true);
    P.active = 0;
    if (P.job.loss) {           // nothing it could ride on: a launch of its own (the loss never waits beyond the bracket)
        const int r = lg::lg_mse_finalize_job(P.job.row_loss, P.job.rows, P.job.inv_n, P.job.loss);
        P.job = lg::LossJob{};
        if (r != LG_OK) rc = r;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_gemm_pair_hold(void) {
    LG_REQUIRE_INIT();
    PairState& P = lg::pair_state();
    LG_ARG(P.active == 1 || P.active == 2, "lg_gemm_pair_hold: no pair bracket is open");
    if (P.active == 1) P.active = 3;
    return LG_OK;
}

extern "C" int lg_gemm_pair_resume(void) {
    LG_REQUIRE_INIT();
    PairState& P = lg::pair_state();
    LG_ARG(P.active != 0, "lg_gemm_pair_resume: no pair bracket is open");
    if (P.active == 3) P.active = 1;
    return LG_OK;
}

extern "C" int lg_gemm_pair_mse_loss(const float* row_loss, int64_t rows, int64_t n, float* loss) {
    LG_REQUIRE_INIT();
    PairState& P = lg::pair_state();
    LG_ARG(P.active != 0, "lg_gemm_pair_mse_loss: no pair bracket is open");
    LG_ARG(row_loss && loss && rows > 0 && n > 0, "lg_gemm_pair_mse_loss: bad arguments");
    LG_ARG(P.job.loss == nullptr, "lg_gemm_pair_mse_loss: the bracket already carries a loss");
    P.job.row_loss = row_loss; P.job.loss = loss; P.job.rows = rows;
    P.job.inv_n = float(1.0 / double(n));                 // python's `1 / numel` rounded once to fp32 (as lg_mse_finalize_f32)
    return LG_OK;
}

extern "C" int lg_gemm_group_begin(void) {
    LG_REQUIRE_INIT();
    GroupState& G = lg::group_state();
    LG_ARG(G.active == 0, "lg_gemm_group_begin: a group bracket is already open");
    G.active = 1;                  // what earlier brackets queued stays queued
    return LG_OK;
}

extern "C" int lg_gemm_group_flush(void) {
    LG_REQUIRE_INIT();
    const int rc = lg::gemm_group_flush_pending();
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_gemm_group_end(void) {
    LG_REQUIRE_INIT();
    GroupState& G = lg::group_state();
    LG_ARG(G.active != 0, "lg_gemm_group_end: no group bracket is open");
    G.active = 0;                  // nothing is launched here: lg_gemm_group_flush does that
    return LG_OK;
}

extern "C" int lg_gemm_addend_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                  const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                  const float* bias, const float* addend, int64_t ldadd) {
    LG_ARG(addend != nullptr, "lg_gemm_addend_f32: addend is NULL");
    return gemm_impl(transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, 1, 0, bias, nullptr, 0, 1, 0, 0, 0, 0, 0, addend, ldadd);
}

extern "C" int lg_gemm_act_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                               const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                               const float* bias, int act, float* aux, int64_t ldaux) {
    LG_ARG(act == LG_ACT_GELU || act == LG_ACT_GELU_BWD, "lg_gemm_act_f32: unknown activation id %d", act);
    LG_ARG(act == LG_ACT_GELU || bias == nullptr, "lg_gemm_act_f32: the backward form takes no bias");
    return gemm_impl(transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, 1, 0, bias, nullptr, 0, 1, 0, 0, 0, 0, 0, nullptr, 0,
                     act, aux, ldaux);
}

extern "C" int lg_gemm_multi3_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                  const float* const* B, int64_t ldb, float* const* C, int64_t ldc, const float* const* bias) {
    LG_ARG(B && C && B[0] && B[1] && B[2] && C[0] && C[1] && C[2], "lg_gemm_multi3_f32: NULL operand");
    LG_ARG(bias == nullptr || (bias[0] && bias[1] && bias[2]), "lg_gemm_multi3_f32: three bias rows or none");
    LG_ARG(!lg::pair_state().active, "lg_gemm_multi3_f32: not inside a pair bracket");
    LG_ARG(transA == 0 && transB == 1, "lg_gemm_multi3_f32: x @ W^T only (transA = 0, transB = 1)");
    GemmArgs x{};
    x.multi = 1;
    for (int i = 0; i < 3; ++i) {
        x.boff[i] = B[i] - B[0];
        x.coff[i] = C[i] - C[0];
        x.biasoff[i] = bias ? bias[i] - bias[0] : 0;
    }
    return gemm_impl(transA, transB, M, N, K, A, lda, 0, B[0], ldb, 0, C[0], ldc, 0, 3, 0, bias ? bias[0] : nullptr, nullptr, 0, 1, 0, 0, 0, 0, 0,
                     nullptr, 0, 0, nullptr, 0, &x);
}

extern "C" int lg_gemm_kseg3_f32(int transA, int transB, int64_t M, int64_t N, int64_t seg_k, const float* A, int64_t lda,
                                 const float* const* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                                 const float* addend, int64_t ldadd) {
    LG_ARG(B && B[0] && B[1] && B[2], "lg_gemm_kseg3_f32: NULL operand");
    LG_ARG(seg_k >= 64 && seg_k % 64 == 0, "lg_gemm_kseg3_f32: seg_k = %lld must be a positive multiple of 64", (long long)seg_k);
    LG_ARG(!(accumulate && addend), "lg_gemm_kseg3_f32: accumulate or addend, not both");
    LG_ARG(!lg::pair_state().active, "lg_gemm_kseg3_f32: not inside a pair bracket");
    LG_ARG(transA == 0 && transB == 0, "lg_gemm_kseg3_f32: g @ W only (transA = 0, transB = 0)");
    GemmArgs x{};
    x.seg_k = int(seg_k);
    // at the end of operand i the running pointer stands at B[i] + seg_k * step, step = 1 along a K-contiguous B, ldb otherwise
    const int64_t span = seg_k * (transB ? 1 : ldb);
    x.seg_jump[0] = (B[1] - B[0]) - span;
    x.seg_jump[1] = (B[2] - B[1]) - span;
    return gemm_impl(transA, transB, M, N, 3 * seg_k, A, lda, 0, B[0], ldb, 0, C, ldc, 0, 1, accumulate, nullptr, nullptr, 0, 1, 0, 0, 0, 0, 0,
                     addend, ldadd, 0, nullptr, 0, &x);
}

extern "C" int lg_gemm_group_colsum_f32(const float* in, int64_t ld, int64_t rows, int64_t cols, float* out, int accumulate) {
    LG_REQUIRE_INIT();
    LG_ARG(in && out && rows >= 0 && cols >= 1 && ld >= cols, "lg_gemm_group_colsum_f32: bad arguments");
    GroupState& G = lg::group_state();
    const int64_t wgs = (cols + 63) / 64;
    if (G.active == 1 && G.grp.cs_wgs == 0 && rows > 0 && wgs <= 65536 && !lg::pair_state().active) {
        for (int i = 0; i < G.count; ++i)                    // never next to a queued product that writes the same buffer
            if (G.queued[i].rowsum == out || G.queued[i].C == out) { const int rc = lg::group_flush(); if (rc != LG_OK) return rc; break; }
        G.grp.cs_in = in; G.grp.cs_out = out;
        G.grp.cs_rows = rows; G.grp.cs_cols = cols; G.grp.cs_ld = ld;
        G.grp.cs_accumulate = accumulate;
        G.grp.cs_wgs = int(wgs);
        return LG_OK;
    }
    const int64_t shape[2] = {rows, cols}, strides[2] = {ld, 1};
    return lg_reduce_acc(LG_RED_SUM, 2, shape, in, strides, 1u, out, accumulate);
}
