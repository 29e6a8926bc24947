// SGEMM on the gfx950 matrix cores: C (+)= op(A) @ op(B), fp32 in / fp32 accumulate,
// built on v_mfma_f32_32x32x2_f32 (exact fp32 products, 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak).
//
// Structure (one workgroup = BM x BN tile of C, WM x WN waves, each wave a TM x TN grid of
// 32x32 MFMA accumulators):
//   global --float4--> registers --ds_write_b128--> LDS (double buffered) --> fragments --> MFMA
//   * the next K-tile's global loads are issued BEFORE the MFMAs of the current tile and written
//     to the other LDS buffer AFTER them: one barrier per K-tile, HBM/L2 latency hidden under
//     BK/2 * TM * TN MFMAs (64 cycles each).
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

namespace lg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float*       C;
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int64_t sA, sB, sC;     // batch strides (elements)
    int     tiles_m, tiles_n;
    int     nwg;            // tiles_m * tiles_n * batch * k_slices
    int     accumulate;
    const float* bias;      // optional [N]: added to every row of the product (nn.Linear's `+ b`, nn.py:96)
    // split-K: slice s of k_slices handles k in [s*k_per_slice, min(K, (s+1)*k_per_slice)) and writes its partial
    // product to W + (batch*k_slices + s)*M*N (dense, ld = N); splitk_combine sums the slices in a fixed order
    int     group_m;        // tile rows walked before moving to the next tile column
    int     k_slices;
    int64_t k_per_slice;    // multiple of BK
    float*  W;
};

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of
// tile ids.  Bijective for any nwg (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, pos = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, bool VA, bool VB>
__global__ void __launch_bounds__(WM * WN * 64) sgemm_mfma(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1 && BK % 16 == 0, "tile config");
    constexpr int A_PITCH = AKC ? BK + 4 : BM;          // floats per LDS row
    constexpr int B_PITCH = BKC ? BK + 4 : BN;
    constexpr int A_TILE = AKC ? BM * A_PITCH : BK * A_PITCH;
    constexpr int B_TILE = BKC ? BN * B_PITCH : BK * B_PITCH;
    constexpr int A_ELEMS = BM * BK / NT, B_ELEMS = BN * BK / NT;   // floats staged per thread
    static_assert(A_ELEMS % 4 == 0 && B_ELEMS % 4 == 0, "staging must divide into float4");

    __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];
    // buffer b: A tile at lds + b*(A_TILE+B_TILE), B tile right behind it
    constexpr int BUF = A_TILE + B_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // tile coordinates
    const int id = xcd_remap(blockIdx.x, g.nwg);
    const int per_batch = g.tiles_m * g.tiles_n;
    const int bs = id / per_batch;                 // (batch, k-slice) pair
    const int batch = bs / g.k_slices, slice = bs - batch * g.k_slices;
    const int t = id - bs * per_batch;
    // grouped order: consecutive ids walk GROUP_M tile rows before moving to the next tile column, so the
    // ~64 workgroups resident on one XCD at a time cover a near-square patch of C and share both their
    // A row-panels and their B column-panels in that XCD's 4 MiB L2
    const int GROUP_M = g.group_m;
    const int gspan = GROUP_M * g.tiles_n;
    const int first_m = (t / gspan) * GROUP_M;
    const int gsize = (g.tiles_m - first_m) < GROUP_M ? (g.tiles_m - first_m) : GROUP_M;
    const int tm = first_m + (t % gspan) % gsize, tn = (t % gspan) / gsize;
    const int64_t m0 = int64_t(tm) * BM, n0 = int64_t(tn) * BN;
    const float* __restrict__ A = g.A + int64_t(batch) * g.sA;
    const float* __restrict__ B = g.B + int64_t(batch) * g.sB;
    float* __restrict__ C = g.k_slices > 1 ? g.W + int64_t(bs) * g.M * g.N : g.C + int64_t(batch) * g.sC;
    const int64_t ldc = g.k_slices > 1 ? g.N : g.ldc;
    const int accumulate = g.k_slices > 1 ? 0 : g.accumulate;
    const float* __restrict__ bias = g.k_slices > 1 ? nullptr : g.bias;   // split-K: the combine pass adds it once
    const int64_t k_begin = int64_t(slice) * g.k_per_slice;
    const int64_t k_end = (k_begin + g.k_per_slice < g.K) ? k_begin + g.k_per_slice : g.K;

    float ra[A_ELEMS], rb[B_ELEMS];   // staging registers

    auto load_tile = [&](int64_t k0) {
        if constexpr (VA) {
#pragma unroll
            for (int i = 0; i < A_ELEMS / 4; ++i) {
                const int f = tid + i * NT;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (AKC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    if (m0 + row < g.M && k0 + kq * 4 < k_end)
                        v = *reinterpret_cast<const float4*>(A + (m0 + row) * g.lda + k0 + kq * 4);
                } else {
                    const int kk = f / (BM / 4), mq = f % (BM / 4);
                    if (k0 + kk < k_end && m0 + mq * 4 < g.M)
                        v = *reinterpret_cast<const float4*>(A + (k0 + kk) * g.lda + m0 + mq * 4);
                }
                ra[4 * i] = v.x; ra[4 * i + 1] = v.y; ra[4 * i + 2] = v.z; ra[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ELEMS; ++i) {
                const int e = tid + i * NT;
                float v = 0.f;
                if constexpr (AKC) {
                    const int row = e / BK, kk = e % BK;
                    if (m0 + row < g.M && k0 + kk < k_end) v = A[(m0 + row) * g.lda + k0 + kk];
                } else {
                    const int kk = e / BM, mm = e % BM;
                    if (k0 + kk < k_end && m0 + mm < g.M) v = A[(k0 + kk) * g.lda + m0 + mm];
                }
                ra[i] = v;
            }
        }
        if constexpr (VB) {
#pragma unroll
            for (int i = 0; i < B_ELEMS / 4; ++i) {
                const int f = tid + i * NT;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (BKC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    if (n0 + row < g.N && k0 + kq * 4 < k_end)
                        v = *reinterpret_cast<const float4*>(B + (n0 + row) * g.ldb + k0 + kq * 4);
                } else {
                    const int kk = f / (BN / 4), nq = f % (BN / 4);
                    if (k0 + kk < k_end && n0 + nq * 4 < g.N)
                        v = *reinterpret_cast<const float4*>(B + (k0 + kk) * g.ldb + n0 + nq * 4);
                }
                rb[4 * i] = v.x; rb[4 * i + 1] = v.y; rb[4 * i + 2] = v.z; rb[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_ELEMS; ++i) {
                const int e = tid + i * NT;
                float v = 0.f;
                if constexpr (BKC) {
                    const int row = e / BK, kk = e % BK;
                    if (n0 + row < g.N && k0 + kk < k_end) v = B[(n0 + row) * g.ldb + k0 + kk];
                } else {
                    const int kk = e / BN, nn = e % BN;
                    if (k0 + kk < k_end && n0 + nn < g.N) v = B[(k0 + kk) * g.ldb + n0 + nn];
                }
                rb[i] = v;
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* a = lds + buf * BUF;
        float* b = lds + buf * BUF + A_TILE;
        if constexpr (VA) {
#pragma unroll
            for (int i = 0; i < A_ELEMS / 4; ++i) {
                const int f = tid + i * NT;
                const float4 v = make_float4(ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]);
                if constexpr (AKC) *reinterpret_cast<float4*>(a + (f / (BK / 4)) * A_PITCH + (f % (BK / 4)) * 4) = v;
                else               *reinterpret_cast<float4*>(a + (f / (BM / 4)) * A_PITCH + (f % (BM / 4)) * 4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ELEMS; ++i) {
                const int e = tid + i * NT;
                if constexpr (AKC) a[(e / BK) * A_PITCH + (e % BK)] = ra[i];
                else               a[(e / BM) * A_PITCH + (e % BM)] = ra[i];
            }
        }
        if constexpr (VB) {
#pragma unroll
            for (int i = 0; i < B_ELEMS / 4; ++i) {
                const int f = tid + i * NT;
                const float4 v = make_float4(rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]);
                if constexpr (BKC) *reinterpret_cast<float4*>(b + (f / (BK / 4)) * B_PITCH + (f % (BK / 4)) * 4) = v;
                else               *reinterpret_cast<float4*>(b + (f / (BN / 4)) * B_PITCH + (f % (BN / 4)) * 4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_ELEMS; ++i) {
                const int e = tid + i * NT;
                if constexpr (BKC) b[(e / BK) * B_PITCH + (e % BK)] = rb[i];
                else               b[(e / BN) * B_PITCH + (e % BN)] = rb[i];
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
        const float* a = lds + buf * BUF + (AKC ? (wm * TM * 32 + r) * A_PITCH + 4 * h : (4 * h) * A_PITCH + wm * TM * 32 + r);
        const float* b = lds + buf * BUF + A_TILE + (BKC ? (wn * TN * 32 + r) * B_PITCH + 4 * h : (4 * h) * B_PITCH + wn * TN * 32 + r);
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

    const int64_t nkt = (k_end - k_begin + BK - 1) / BK;
    load_tile(k_begin);
    store_tile(0);
    __syncthreads();
    for (int64_t kt = 0; kt < nkt; ++kt) {
        const int cur = int(kt & 1);
        const bool more = kt + 1 < nkt;
        if (more) load_tile(k_begin + (kt + 1) * BK);      // in flight during the MFMAs below
        compute_tile(cur, 0, BK / 2);
        if (more) store_tile(cur ^ 1);           // ds_writes issue in the shadow of the second half's MFMAs
        compute_tile(cur, BK / 2, BK);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + (wn * TN + j) * 32 + r;
            const int64_t row0 = m0 + (wm * TM + i) * 32 + 4 * h;
            if (col < g.N) {
                const float bv = bias ? bias[col] : 0.f;
                float old[16];
                if (accumulate) {      // C += ...: fetch the 16 old values first so the loads overlap, then add and store
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                        old[e] = row < g.M ? C[row * ldc + col] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < g.M) {
                        const float val = bias ? acc[i][j][e] + bv : acc[i][j][e];
                        C[row * ldc + col] = accumulate ? old[e] + val : val;
                    }
                }
            }
        }
    }
}

// C[b][m][n] (+)= sum_s W[b][s][m][n], slices added in index order (deterministic)
__global__ void __launch_bounds__(256) splitk_combine(const float* __restrict__ W, float* __restrict__ C, int64_t M, int64_t N,
                                                      int64_t ldc, int64_t sC, int slices, int64_t total, int accumulate,
                                                      const float* __restrict__ bias) {
    const int64_t mn = M * N;
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / mn, r = e - b * mn;
        const int64_t m = r / N, n = r - m * N;
        const float* w = W + b * slices * mn + r;
        float acc = w[0];
        for (int s = 1; s < slices; ++s) acc += w[int64_t(s) * mn];
        if (bias) acc += bias[n];
        float* c = C + b * sC + m * ldc + n;
        *c = accumulate ? *c + acc : acc;
    }
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC>
static void launch_layout(const GemmArgs& g, bool va, bool vb) {
    dim3 grid(g.nwg), block(WM * WN * 64);
    hipStream_t s = rt().stream;
    if (va && vb)  hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, true>), grid, block, 0, s, g);
    else if (va)   hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, true, false>), grid, block, 0, s, g);
    else if (vb)   hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, false, true>), grid, block, 0, s, g);
    else           hipLaunchKernelGGL((sgemm_mfma<BM, BN, BK, WM, WN, AKC, BKC, false, false>), grid, block, 0, s, g);
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_config(const GemmArgs& base, bool akc, bool bkc, bool va, bool vb, int64_t batch) {
    GemmArgs g = base;
    g.tiles_m = int((g.M + BM - 1) / BM);
    g.tiles_n = int((g.N + BN - 1) / BN);
    const int64_t tiles = int64_t(g.tiles_m) * g.tiles_n * batch;
    // split-K when the tile grid alone cannot fill 256 CUs: aim at >= 512 workgroups, keep >= 2 K-tiles per slice
    int64_t slices = 1;
    if (tiles < 256 && g.K >= 4 * BK) {
        slices = (512 + tiles - 1) / tiles;
        const int64_t max_slices = g.K / (2 * BK);
        if (slices > max_slices) slices = max_slices;
        if (slices > 64) slices = 64;
        if (slices < 1) slices = 1;
    }
    g.k_per_slice = ((g.K + slices - 1) / slices + BK - 1) / BK * BK;
    slices = (g.K + g.k_per_slice - 1) / g.k_per_slice;
    g.k_slices = int(slices);
    g.nwg = int(tiles * slices);
    g.W = nullptr;
    if (slices > 1) {
        int rc = lg_malloc(reinterpret_cast<void**>(&g.W), size_t(batch * slices * g.M * g.N) * sizeof(float));
        if (rc != LG_OK) return rc;
    }
    if (akc && bkc) launch_layout<BM, BN, BK, WM, WN, true, true>(g, va, vb);
    else if (akc)   launch_layout<BM, BN, BK, WM, WN, true, false>(g, va, vb);
    else if (bkc)   launch_layout<BM, BN, BK, WM, WN, false, true>(g, va, vb);
    else            launch_layout<BM, BN, BK, WM, WN, false, false>(g, va, vb);
    if (slices > 1) {
        const int64_t total = batch * g.M * g.N;
        hipLaunchKernelGGL(splitk_combine, dim3(stream_grid(total)), dim3(256), 0, rt().stream, g.W, g.C, g.M, g.N, g.ldc, g.sC,
                           g.k_slices, total, g.accumulate, g.bias);
        return lg_free(g.W);     // stream-ordered: reused only by later launches
    }
    return LG_OK;
}

}  // namespace lg

using namespace lg;

static int gemm_impl(int transA, int transB, int64_t M, int64_t N, int64_t K,
                     const float* A, int64_t lda, int64_t strideA,
                     const float* B, int64_t ldb, int64_t strideB,
                     float* C, int64_t ldc, int64_t strideC,
                     int64_t batch, int accumulate, const float* bias) {
    LG_REQUIRE_INIT();
    LG_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "lg_gemm_f32: negative extent (M=%lld N=%lld K=%lld batch=%lld)",
           (long long)M, (long long)N, (long long)K, (long long)batch);
    if (M == 0 || N == 0 || batch == 0) return LG_OK;
    LG_ARG(A && B && C, "lg_gemm_f32: NULL operand");
    LG_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
           "lg_gemm_f32: leading dimension too small (lda=%lld ldb=%lld ldc=%lld for M=%lld N=%lld K=%lld tA=%d tB=%d)",
           (long long)lda, (long long)ldb, (long long)ldc, (long long)M, (long long)N, (long long)K, transA, transB);
    if (K == 0) {
        LG_ARG(bias == nullptr, "lg_gemm_bias_f32: K == 0 with a bias is not supported");
        if (accumulate) return LG_OK;
        // empty sum: C = 0
        int64_t shape[3] = {batch, M, N}, st[3] = {strideC, ldc, 1};
        return lg_fill_strided(4, 3, shape, C, st, 0);
    }

    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.accumulate = accumulate;
    g.bias = bias;
    static const char* group_env = getenv("LG_GEMM_GROUP");
    g.group_m = group_env ? atoi(group_env) : 8;
    if (g.group_m < 1) g.group_m = 1;

    const bool akc = !transA;   // A[m*lda + k]: k is the contiguous index
    const bool bkc = transB != 0;   // B[n*ldb + k]
    // float4 staging needs 16-byte aligned rows and whole float4s inside the matrix
    auto vec_ok = [](const float* p, int64_t ld, int64_t bstride, int64_t contiguous_extent) {
        return aligned16(p) && ld % 4 == 0 && bstride % 4 == 0 && contiguous_extent % 4 == 0;
    };
    const bool va = vec_ok(A, lda, strideA, akc ? K : M), vb = vec_ok(B, ldb, strideB, bkc ? K : N);

    // tile choice: largest tile that still yields enough workgroups for 256 CUs
    auto nblocks = [&](int64_t bm, int64_t bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
    LG_ARG(nblocks(32, 32) < (int64_t(1) << 30), "lg_gemm_f32: problem too large for one launch");
    // Tile choice.  Skinny outputs take the 64x32 / 32x64 tiles.  Otherwise three tiles compete:
    //   256x256 (16 waves, 1 WG/CU)  132-141 TFLOP/s at 4096^3   128x128 (4 waves)  ~124   64x64 (4 waves, + split-K)  ~107 at 3072^3
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
            const double c256 = cost(256, 256, 0.87), c128 = cost(128, 128, 0.78), c64 = cost(64, 64, 0.72);
            tile = (c256 <= c128 && c256 <= c64) ? 2 : (c128 <= c64 ? 0 : 9);
        }
        switch (tile) {
            case 1:  rc = launch_config<256, 128, 32, 4, 2>(g, akc, bkc, va, vb, batch); break;   // 8 waves (experiments only)
            case 2:  rc = launch_config<256, 256, 32, 4, 4>(g, akc, bkc, va, vb, batch); break;
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
