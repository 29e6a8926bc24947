// LAB - not part of the library (measured and not kept: profiles/r3/shortk_lab.txt).  It was built as one more translation unit of
// lightgrad_amd/csrc (copy it there, add it to CORE_SRCS, declare shortk_wants / shortk_launch in common.h and call them from gemm_impl).
// C[M x N] = A[M x K] @ B[N x K]^T + bias for a SHORT K (<= 128) and a very wide N: the projection of a hidden state onto
// a vocabulary (tiny-BERT's decoder, reference examples/bert.py:226-227: 1024 x 30522 logits from K = 128).
//
// The tiled kernel of gemm.hip gives every 64 x 64 output tile a workgroup of its own; with four K-steps per tile most of a
// workgroup's life is prologue (first loads) and epilogue - 84 TFLOP/s, 53 % of the fp32 MFMA peak, on this shape.  Here a
// workgroup keeps a 64-row panel of A in LDS for its whole life and walks along N: per step one 64-row tile of B comes in
// (global -> registers while the previous tile is multiplied, registers -> LDS between two barriers), four waves multiply
// panel x tile (each a 32 x 32 block over the whole K, operands as ds_read_b128, see mfma_lds.h) and store their block of C.
// Two workgroups share a CU (68 KiB of LDS each at K = 128): one's stores, LDS refill and barriers run under the other's MFMAs -
// with ONE workgroup of eight waves per CU both waves of a SIMD reach the epilogue together and the matrix cores idle
// (measured: 86.9 us against 95.3 us for the tiled kernel; this form: see profiles/r3/bert_gemm_bench).
//
// Workgroup ids: panel-major, `chunks` (a multiple of 8) ranges of N per panel - the workgroups that share a range of B have
// ids equal modulo 8, i.e. the same XCD and the same L2, which then fetches every B tile once.
#include "common.h"
#include "mfma_lds.h"

namespace lg {

struct ShortKArgs {
    const float* A;
    const float* B;
    float*       C;
    const float* bias;
    int64_t M, N;
    int64_t lda, ldb, ldc;
    int     K;
    int     tiles_n;        // 64-column tiles along N
    int     chunks;         // ranges of N per panel
    int     per_chunk;      // tiles per range
};

constexpr int kPanel = 64, kTileN = 64, kShortKMax = 128, kShortKThreads = 256;

__global__ void __launch_bounds__(kShortKThreads) sgemm_nt_shortk(ShortKArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = g.K, P = K + 4, Q = K / 4;                    // row pitch (floats), float4 per row
    float* As = lds;                                            // kPanel x P
    float* Bs = lds + kPanel * P;                               // kTileN x P
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;                    // the wave's 32 x 32 block of the 64 x 64 step
    const int panel = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
    const int64_t m0 = int64_t(panel) * kPanel;
    const int t0 = chunk * g.per_chunk;
    const int t1 = t0 + g.per_chunk < g.tiles_n ? t0 + g.per_chunk : g.tiles_n;
    if (t0 >= t1) return;

    // rows past the matrix are clamped to its last row: every load is valid, their results are never stored
    constexpr int NT = kShortKThreads;
    constexpr int NA = kPanel * (kShortKMax / 4) / NT, NB = kTileN * (kShortKMax / 4) / NT;
    af32x4 ra[NA], rb[NB];
    // The loads of the loop are inline asm: hipcc's own wait insertion drains every outstanding load at the loop's back-edge and
    // in front of each conditional store, which would put the tile's HBM round trip and sixteen store round trips in front of
    // the MFMAs.  Their destination registers are NOT protected until wait_loads() has run.
    auto load_b = [&](int t) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f0 = tid + i * NT, f = f0 < kTileN * Q ? f0 : 0;
            int64_t row = int64_t(t) * kTileN + f / Q;
            row = row < g.N ? row : g.N - 1;
            const float* src = g.B + row * g.ldb + (f % Q) * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[i]) : "v"(src));
        }
    };
    // every load of the loop (the next B tile, this tile's bias values) has landed - and the stores of the previous tile,
    // which are older, have left
    float bv = 0.f;
    auto wait_loads = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(rb[i]));
        asm volatile("" : "+v"(bv));
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * NT;
            if (f < kTileN * Q) *reinterpret_cast<af32x4*>(dst + (f / Q) * P + (f % Q) * 4) = rb[i];
        }
    };
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f0 = tid + i * NT, f = f0 < kPanel * Q ? f0 : 0;
        int64_t row = m0 + f / Q;
        row = row < g.M ? row : g.M - 1;
        ra[i] = *reinterpret_cast<const af32x4*>(g.A + row * g.lda + (f % Q) * 4);
    }
    load_b(t0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + i * NT;
        if (f < kPanel * Q) *reinterpret_cast<af32x4*>(As + (f / Q) * P + (f % Q) * 4) = ra[i];
    }
    wait_loads();
    store_b(Bs);
    __syncthreads();

    const float* Aw = As + 32 * wm * P;
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) load_b(t + 1);
        const int64_t col = int64_t(t) * kTileN + 32 * wn + r;
        if (g.bias) {
            const float* src = g.bias + (col < g.N ? col : g.N - 1);
            asm volatile("global_load_dword %0, %1, off" : "=v"(bv) : "v"(src));
        }
        af32x16 acc = zero16();
        wave_mma<true, true>(acc, Aw, P, Bs + 32 * wn * P, P, K, r, h);
        wait_loads();
        const int64_t row0 = m0 + 32 * wm;
        float* dst = g.C + row0 * g.ldc + col;
        if (row0 + 32 <= g.M && int64_t(t + 1) * kTileN <= g.N) {             // (wave-uniform) the whole block is inside C
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[int64_t(acc_row(e, h)) * g.ldc] = g.bias ? acc[e] + bv : acc[e];
        } else if (col < g.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = acc_row(e, h);
                if (row0 + rr < g.M) dst[int64_t(rr) * g.ldc] = g.bias ? acc[e] + bv : acc[e];
            }
        }
        __syncthreads();                                        // every wave has read tile t
        if (t + 1 < t1) store_b(Bs);
        __syncthreads();
    }
}

// used by gemm.hip: does this product belong here, and if so launch it
bool shortk_wants(bool akc, bool bkc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb) {
    if (!akc || !bkc || K < 8 || K > kShortKMax || K % 8 != 0) return false;
    if (lda % 4 != 0 || ldb % 4 != 0 || !aligned16(A) || !aligned16(B)) return false;
    const int64_t cus = rt().compute_units > 0 ? rt().compute_units : 256;
    const int64_t panels = (M + kPanel - 1) / kPanel, tiles_n = (N + kTileN - 1) / kTileN;
    // enough steps per workgroup to pay for its panel, and enough workgroups for the chip
    return M >= kPanel && panels * tiles_n >= 16 * cus && tiles_n >= 64 && panels <= 2 * cus;
}

int shortk_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  const float* bias) {
    const int64_t cus = rt().compute_units > 0 ? rt().compute_units : 256;
    ShortKArgs g{A, B, C, bias, M, N, lda, ldb, ldc, int(K), 0, 0, 0};
    const int64_t panels = (M + kPanel - 1) / kPanel;
    g.tiles_n = int((N + kTileN - 1) / kTileN);
    int64_t chunks = 2 * cus / panels;                           // two workgroups per CU
    chunks = chunks >= 8 ? chunks / 8 * 8 : 8;
    if (chunks > g.tiles_n) chunks = (g.tiles_n + 7) / 8 * 8;
    g.chunks = int(chunks);
    g.per_chunk = int((g.tiles_n + chunks - 1) / chunks);
    const size_t bytes = size_t(kPanel + kTileN) * size_t(K + 4) * sizeof(float);
    static size_t allowed = 0;
    if (bytes > allowed) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sgemm_nt_shortk), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
        if (e != hipSuccess) { set_error("lg_gemm_f32: %zu bytes of LDS refused: %s", bytes, hipGetErrorString(e)); return LG_EHIP; }
        allowed = bytes;
    }
    hipLaunchKernelGGL(sgemm_nt_shortk, dim3(unsigned(panels * chunks)), dim3(kShortKThreads), bytes, rt().stream, g);
    return LG_OK;
}

}  // namespace lg
