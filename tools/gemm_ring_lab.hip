// Lab: a barrier-free small-tile SGEMM for the launch- and latency-bound products of the MLP step.
//
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_ring_lab.hip -o tools/gemm_ring_lab.bin && tools/gemm_ring_lab.bin
//
// C[M][N] = A[M][K] @ B[N][K]^T (both operands K-contiguous: the forward product of nn.Linear, x @ W^T), 64x32 output tiles,
// 4 waves per workgroup = 2 (M) x 2 K-groups, one 32x32 accumulator per wave.  What differs from csrc/gemm_tile_body.inc:
//   * every wave stages ITS OWN operand fragments: no workgroup barrier in the K loop at all (one wave per SIMD: a barrier or
//     an `s_waitcnt` that stalls the wave is idle matrix-core time, 0.81 us per 64-k step against 0.43 us of MFMAs)
//   * global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write pass; a ring of three 8 KB slots
//     per wave with two K-steps in flight behind counted `vmcnt` waits
//   * the LDS image is lane-linear per DMA instruction (4 lanes x 16 B per row, 16 rows); the XOR swizzle that makes the
//     ds_read_b128 fragment reads conflict-free is applied on the SOURCE address (cdna_hip_programming.md 5.4 rule 21)
//   * DMA issue and fragment reads of the NEXT step are interleaved one per MFMA of the current step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef KGROUPS
#define KGROUPS 2
#endif
constexpr int BM = 64, BN = 32, KG = KGROUPS, BKS = 64, BKW = BKS / KG;      // k per step of the workgroup, per wave
constexpr int KH = BKW / 16, NDMA = 4 * KH, NQ = BKW / 8;               // 16-k halves, DMA instructions and 8-k groups per wave and step
constexpr int OP_BYTES = KH * 2048, SLOT_BYTES = 2 * OP_BYTES;            // A image + B image of one step of one wave
#ifndef RING_SLOTS
#define RING_SLOTS 3
#endif
constexpr int RING = RING_SLOTS;
constexpr unsigned OOB = 0x80000000u;

struct Args {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc, tiles_n;
    long long* trace;                        // -DLAB_TRACE: cycle stamps of one wave (workgroup 100, wave 0): 4 per K-step
};
#ifdef LAB_TRACE
#define STAMP(slot) do { if (blockIdx.x == 100 && wave == 0 && lane == 0) g.trace[step * 4 + (slot)] = clock64(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ u32x4 descriptor(const float* base) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    u32x4 d;
    d[0] = unsigned(p); d[1] = unsigned(p >> 32) & 0xffffu; d[2] = OOB; d[3] = 0x00020000u;
    return d;
}

// one LDS-DMA: 64 lanes x 16 bytes from (descriptor + per-lane byte offset) to LDS bytes [lds_dst + lane * 16, +16)
__device__ __forceinline__ void dma16(unsigned off, u32x4 desc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(off), "s"(lds_dst), "s"(desc) : "memory", "m0");
}

__global__ void __launch_bounds__(128 * KG) sgemm_ring_nt(Args g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];           // 4 waves x RING x 8 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: SGPRs
    const int wm = wave & 1, kg = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x % g.tiles_n;
    const int m0 = tm * BM + wm * 32, n0 = tn * BN;
    const unsigned ring0 = unsigned(wave) * RING * SLOT_BYTES;           // LDS byte address of this wave's ring (dynamic LDS starts at 0)

    // DMA lane geometry: instruction (operand, khalf, rowblock): lane l -> row rho = rowblock*16 + l/4, LDS chunk position l%4,
    // which receives the operand's chunk (l%4) ^ ((rho >> 2) & 3) of that row's 16-k half
    unsigned offA[KH][2], offB[KH][2];                                     // [khalf][rowblock], byte offsets from the step's base
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int rho = rb * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
            const unsigned kbytes = unsigned(kh * 16 + chunk * 4) * 4u;
            offA[kh][rb] = (m0 + rho < g.M) ? unsigned(rho) * unsigned(g.lda) * 4u + kbytes : OOB;
            offB[kh][rb] = (n0 + rho < g.N) ? unsigned(rho) * unsigned(g.ldb) * 4u + kbytes : OOB;
        }
    const float* Abase = g.A + size_t(m0) * g.lda + kg * BKW;            // + step * BKS
    const float* Bbase = g.B + size_t(n0) * g.ldb + kg * BKW;
    const int nsteps = (g.K + BKS - 1) / BKS;

    // k of this lane's chunk inside the wave's 32-k step, per (khalf, rowblock) - for the K tail predicate
    // per-step state of the DMA stream, advanced once per step (scalar): descriptors at the step's first k, ring slot
    u32x4 dsc[2];
    unsigned slot_base = 0;
    int issue_step = -1, issue_slot = 0;
    bool tail_step = false;
    const float* nextA = Abase;
    const float* nextB = Bbase;
    auto begin_step = [&]() {                                            // prepares step issue_step + 1
        ++issue_step;
        dsc[0] = descriptor(nextA);
        dsc[1] = descriptor(nextB);
        nextA += BKS;
        nextB += BKS;
        slot_base = ring0 + issue_slot * SLOT_BYTES;
        issue_slot = issue_slot + 1 == RING ? 0 : issue_slot + 1;
        tail_step = (issue_step + 1) * BKS > g.K;                        // some k of this step lie beyond K
    };
    auto issue = [&](int which) {                                        // which = 0..7: one DMA instruction of the prepared step
        const int op = which / (2 * KH), kh = (which / 2) % KH, rb = which & 1;
        unsigned off = op == 0 ? offA[kh][rb] : offB[kh][rb];
        if (tail_step) {
            const int rho = rb * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
            if (issue_step * BKS + kg * BKW + kh * 16 + chunk * 4 >= g.K) off = OOB;      // K is a multiple of 4 here (lab)
        }
        dma16(off, dsc[op], slot_base + op * OP_BYTES + kh * 2048 + rb * 1024);
    };
    // fragment of k-group q (8 k) of the step in `slot`: lane (r, h) reads 4 floats at k = 8q + 4h of row r
    const int lane_frag[2] = {r * 64 + ((h ^ ((r >> 2) & 3)) * 16), r * 64 + (((2 + h) ^ ((r >> 2) & 3)) * 16)};      // q even / odd
    auto frag_addr = [&](int slot, int op, int q) -> const f32x4* {
        return reinterpret_cast<const f32x4*>(lds + ring0 + slot * SLOT_BYTES + op * OP_BYTES + (q >> 1) * 2048 + lane_frag[q & 1]);
    };

#ifndef NACC
#define NACC 1
#endif
#ifndef MODE
#define MODE 0
#endif
    f32x16 accs[NACC];                       // independent accumulation chains (summed at the end): a lone dependent chain of
#pragma unroll                               // 32x32x2 MFMAs does not issue back to back
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[a][e] = 0.f;

    // prologue: steps 0 .. RING-1 in flight; fragments of step 0 into registers
#pragma unroll
    for (int s = 0; s < RING; ++s)
        if (s < nsteps) {
            begin_step();
#pragma unroll
            for (int w = 0; w < NDMA; ++w) issue(w);
        }
    f32x4 fa[2][NQ], fb[2][NQ];
    {
        static_assert(RING >= 3 && RING <= 5, "ring depth");
        if (nsteps >= RING) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA * (RING - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (short K: not the case this lab times)
#pragma unroll
        for (int q = 0; q < NQ; ++q) { fa[0][q] = *frag_addr(0, 0, q); fb[0][q] = *frag_addr(0, 1, q); }
    }
    for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int step = s + u;
            if (step < nsteps) {
                const int cur = u, nxt = u ^ 1;
                const bool more = step + 1 < nsteps, issue_more = step + RING < nsteps;
                // steps <= step+1 have landed once at most the newest step (8 instructions) is outstanding
                STAMP(0);
                if (more && MODE != 2 && MODE != 3) {
                    if (step + RING - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA * (RING - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail drains (a few steps of less overlap)
                }
                STAMP(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this step's fragments (read during the previous step)
                STAMP(2);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (MODE != 1 && MODE != 4) accs[e % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][e], fb[cur][q][e], accs[e % NACC], 0, 0, 0);
                        if (MODE == 4) accs[0][e] += fa[cur][q][e] * fb[cur][q][e];
                        const int i = q * 4 + e;                         // one memory instruction per MFMA
                        if (i < 2 * NQ) {
                            if (more && MODE != 1 && MODE != 3) { if (i < NQ) fa[nxt][i] = *frag_addr((step + 1) % RING, 0, i); else fb[nxt][i - NQ] = *frag_addr((step + 1) % RING, 1, i - NQ); }
                        } else if (i < 2 * NQ + NDMA && issue_more && MODE != 2 && MODE != 3) {
                            if (i == 2 * NQ) begin_step();
                            issue(i - 2 * NQ);
                        }
                    }
                }
                STAMP(3);
            }
        }
    }
    f32x16 acc = accs[0];
#pragma unroll
    for (int a = 1; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += accs[a][e];
    // K-group 1 hands its accumulator to group 0 through LDS (the rings are free now)
    __syncthreads();
    float* x = reinterpret_cast<float*>(lds) + wm * 16 * 64 + lane;          // slot (group - 1, wm): 2 * 16 * 64 floats per group
    if (kg >= 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) x[(kg - 1) * 2048 + e * 64] = acc[e];
    }
    __syncthreads();
    if (kg >= 1) return;
#pragma unroll
    for (int gq = 0; gq < KG - 1; ++gq)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += x[gq * 2048 + e * 64];
    const int col = n0 + r;
    if (col < g.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < g.M) g.C[size_t(row) * g.ldc + col] = acc[e];
        }
    }
}


// ---- producer / consumer variant (-DPC=1): wave 4 issues ALL LDS-DMAs, waves 0-3 only read fragments and multiply ----------------
// Measured above: an instruction of a wave does not overlap that wave's own MFMAs (the step costs 16 x 64 cycles of MFMA plus ~30
// cycles per LDS fragment read plus ~60 per LDS-DMA) - but another wave's instructions do.  So the DMA stream moves to a wave of
// its own; consumers keep 16 MFMAs + 8 fragment reads per step.  Hand-off through LDS words (no s_barrier): full[slot][w] = step + 1
// once the slot's bytes have landed (producer, behind a counted vmcnt), freed[w] = steps whose fragments consumer w has in registers.
#ifndef PC_RING
#define PC_RING 4
#endif
#ifndef PC_AHEAD
#define PC_AHEAD 1
#endif
__global__ void __launch_bounds__(320) sgemm_pc_nt(Args g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int R = PC_RING, D = PC_AHEAD;                 // ring slots per consumer; steps the producer keeps in flight before it publishes
    static_assert(KG == 2 && D >= 1 && D < R, "geometry");
    constexpr unsigned FLAGS = 4u * R * SLOT_BYTES;           // byte address of the hand-off words: full[R][4] then freed[4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x % g.tiles_n;
    const int nsteps = (g.K + BKS - 1) / BKS;
    volatile int* flags = reinterpret_cast<volatile int*>(lds + FLAGS);
    if (tid < 4 * R + 4) flags[tid] = 0;
    __syncthreads();
    if (wave == 4) {
        // ---------------- producer ----------------
        const int n0 = tn * BN;
        unsigned offA[2][KH][2], offB[KH][2];                // [wm][khalf][rowblock]
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int rho = rb * 16 + (lane >> 2);
                const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
                const unsigned kbytes = unsigned(kh * 16 + chunk * 4) * 4u;
#pragma unroll
                for (int wm = 0; wm < 2; ++wm)
                    offA[wm][kh][rb] = (tm * BM + wm * 32 + rho < g.M) ? unsigned(rho) * unsigned(g.lda) * 4u + kbytes : OOB;
                offB[kh][rb] = (n0 + rho < g.N) ? unsigned(rho) * unsigned(g.ldb) * 4u + kbytes : OOB;
            }
        auto publish = [&](int step) {                        // the slot of `step` of all four consumers is full
            if (lane < 4) flags[(step % R) * 4 + lane] = step + 1;
        };
        for (int s = 0; s < nsteps; ++s) {
            if (s >= R) {                                     // the slot must have been read: freed[w] >= s - R + 1 for every consumer
                for (;;) {
                    const int f = lane < 4 ? flags[4 * R + lane] : 0x7fffffff;
                    if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f < s - R + 1) == 0)) break;
                }
            }
            const bool tail = (s + 1) * BKS > g.K;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int wm = w & 1, kg = w >> 1;
                const u32x4 da = descriptor(g.A + size_t(tm * BM + wm * 32) * g.lda + kg * BKW + size_t(s) * BKS);
                const u32x4 db = descriptor(g.B + size_t(n0) * g.ldb + kg * BKW + size_t(s) * BKS);
                const unsigned base = unsigned(w) * R * SLOT_BYTES + (s % R) * SLOT_BYTES;
#pragma unroll
                for (int which = 0; which < NDMA; ++which) {
                    const int op = which / (2 * KH), kh = (which / 2) % KH, rb = which & 1;
                    unsigned off = op == 0 ? offA[wm][kh][rb] : offB[kh][rb];
                    if (tail) {
                        const int rho = rb * 16 + (lane >> 2);
                        const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
                        if (s * BKS + kg * BKW + kh * 16 + chunk * 4 >= g.K) off = OOB;
                    }
                    dma16(off, op == 0 ? da : db, base + op * OP_BYTES + kh * 2048 + rb * 1024);
                }
            }
            if (s >= D) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * NDMA * D) : "memory");
                publish(s - D);
            }
        }
        // drain: the last step
        static_assert(D == 1 && 4 * NDMA * D <= 63, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(nsteps - 1);
        return;
    }
    // ---------------- consumers ----------------
    const int wm = wave & 1, kg = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = tm * BM + wm * 32, n0 = tn * BN;
    const unsigned ring0 = unsigned(wave) * R * SLOT_BYTES;
    const int lane_frag[2] = {r * 64 + ((h ^ ((r >> 2) & 3)) * 16), r * 64 + (((2 + h) ^ ((r >> 2) & 3)) * 16)};
    auto frag_addr = [&](int slot, int op, int q) -> const f32x4* {
        return reinterpret_cast<const f32x4*>(lds + ring0 + slot * SLOT_BYTES + op * OP_BYTES + (q >> 1) * 2048 + lane_frag[q & 1]);
    };
    auto wait_full = [&](int step) {
        while (flags[(step % R) * 4 + wave] < step + 1) { }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 fa[2][NQ], fb[2][NQ];
    wait_full(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { fa[0][q] = *frag_addr(0, 0, q); fb[0][q] = *frag_addr(0, 1, q); }
    for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int step = s + u;
            if (step < nsteps) {
                const int cur = u, nxt = u ^ 1;
                const bool more = step + 1 < nsteps;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this step's fragments are in registers ...
                if (lane == 0) flags[4 * R + wave] = step + 1;          // ... so its slot may be refilled
                if (more) wait_full(step + 1);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][e], fb[cur][q][e], acc, 0, 0, 0);
                        const int i = q * 4 + e;
                        if (i < 2 * NQ && more) { if (i < NQ) fa[nxt][i] = *frag_addr((step + 1) % R, 0, i); else fb[nxt][i - NQ] = *frag_addr((step + 1) % R, 1, i - NQ); }
                    }
                }
            }
        }
    }
    // K-group 1 hands its accumulator to group 0 through LDS.  Only the four consumer waves take part: a counter instead of s_barrier
    float* x = reinterpret_cast<float*>(lds) + wm * 16 * 64 + lane;
    volatile int* done = flags + 4 * R + 4;                  // (zeroed with the flags? it lies beyond them: set below)
    if (kg == 1) {
        // the rings are being reused as exchange space: every consumer must have finished its LAST fragment reads - each wave's own
        // are done (lgkmcnt(0) below), and group 0's region [0, 2 * 4 KB) belongs to consumers 0 / 1 (wm = 0 / 1 of group 0) themselves
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    (void)done;
    __builtin_amdgcn_s_waitcnt(0);
    // consumers synchronise among themselves through freed[]: all four have posted freed == nsteps when their loops are over
    for (;;) {
        const int f = lane < 4 ? flags[4 * R + lane] : 0x7fffffff;
        if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f < nsteps) == 0)) break;
    }
    float* xx = reinterpret_cast<float*>(lds + FLAGS + 256) + wm * 16 * 64 + lane;      // exchange space behind the flags
    (void)x;
    if (kg == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xx[e * 64] = acc[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flags[(wave & 1)] = -1;               // full[0][wm] reused: "group 1's accumulator of row block wm is in LDS"
        return;
    }
    while (flags[wm] != -1) { }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += xx[e * 64];
    const int col = n0 + r;
    if (col < g.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < g.M) g.C[size_t(row) * g.ldc + col] = acc[e];
        }
    }
}

int main() {
    const int M = 1024, N = 512, K = getenv("LAB_K") ? atoi(getenv("LAB_K")) : 784;
    std::vector<float> a(size_t(M) * K), b(size_t(N) * K), c(size_t(M) * N);
    srand(1);
    for (auto& v : a) v = float(rand()) / RAND_MAX * 2 - 1;
    for (auto& v : b) v = float(rand()) / RAND_MAX * 2 - 1;
    float *da, *db, *dc;
    CK(hipMalloc(&da, a.size() * 4 + 64)); CK(hipMalloc(&db, b.size() * 4 + 64)); CK(hipMalloc(&dc, c.size() * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    long long* dtrace = nullptr;
    CK(hipMalloc(&dtrace, 4096 * 8));
    CK(hipMemset(dtrace, 0, 4096 * 8));
    Args g{da, db, dc, M, N, K, K, K, N, (N + BN - 1) / BN, dtrace};
    const int grid = ((M + BM - 1) / BM) * g.tiles_n;
#ifdef PC
    const size_t lds_bytes = size_t(4) * PC_RING * SLOT_BYTES + 256 + 2 * 16 * 64 * 4;
#define LAB_KERNEL sgemm_pc_nt
#define LAB_THREADS 320
#else
    const size_t lds_bytes = size_t(2 * KG) * RING * SLOT_BYTES;
#define LAB_KERNEL sgemm_ring_nt
#define LAB_THREADS (128 * KG)
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(LAB_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
    hipLaunchKernelGGL(LAB_KERNEL, dim3(grid), dim3(LAB_THREADS), lds_bytes, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, norm = 0, err = 0;
    for (int t = 0; t < 4000; ++t) {
        const int i = rand() % M, j = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += double(a[size_t(i) * K + k]) * b[size_t(j) * K + k];
        const double d = c[size_t(i) * N + j] - ref;
        worst = fmax(worst, fabs(d)); norm += ref * ref; err += d * d;
    }
    printf("ring NT %dx%dx%d: max |err| %.3e, relative Frobenius (4000 samples) %.3e\n", M, N, K, worst, sqrt(err / norm));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(LAB_KERNEL, dim3(grid), dim3(LAB_THREADS), lds_bytes, 0, g);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %.2f us per launch back to back (%d workgroups, %d K-steps of %d): %.1f TFLOP/s\n", 1e3 * ms / 50, grid, (K + BKS - 1) / BKS, BKS,
               2.0 * M * N * K / (ms / 50 * 1e-3) / 1e12);
    }
#ifdef LAB_TRACE
    {
        std::vector<long long> tr(4096);
        CK(hipMemcpy(tr.data(), dtrace, 4096 * 8, hipMemcpyDeviceToHost));
        const int ns = (K + BKS - 1) / BKS;
        printf("  one wave, cycles (clock64) per K-step: wait for the DMA of the next step | wait for this step's LDS fragments | MFMAs + issue of reads and DMAs | to the next step\n");
        for (int s = 0; s < ns; ++s)
            printf("    step %2d: %6lld %6lld %6lld %6lld\n", s, tr[s * 4 + 1] - tr[s * 4], tr[s * 4 + 2] - tr[s * 4 + 1], tr[s * 4 + 3] - tr[s * 4 + 2],
                   s + 1 < ns ? tr[(s + 1) * 4] - tr[s * 4 + 3] : 0LL);
    }
#endif
    return sqrt(err / norm) < 1e-5 ? 0 : 1;
}
