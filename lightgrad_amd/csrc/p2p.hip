// Peer-window gradient exchange (include/lghip_p2p.h): the data-parallel exchange step of BASELINE config #4 as
// ordinary kernels on the compute stream - no collective library, no second stream, nothing a hipGraph cannot record.
// The reference has no distributed code; contract: SURVEY.md 8e.
//
// Memory.  A rank's WINDOW is one uncached device allocation (hipDeviceMallocUncached: never held in an L2, so a
// store that arrives over xGMI - or from another process on the same GPU - is what the next load returns):
//     flag1   [max_chunks][16] int   flag1[c][s] = epoch of the last push of chunk c by rank s (written by rank s)
//     flag2   [max_chunks]     int   epoch of the last reduced chunk c (written by the chunk's owner)
//     recv    [nranks][cap]    float slot s = rank s's contribution to the chunks THIS rank owns
//     reduced [cap]            float the summed bucket, each chunk written by its owner
// Peers map it with hipIpcOpenMemHandle.  Remote accesses are stores only (posted writes suit xGMI); all loads are
// local.  Payload stores are 16-byte `sc0 sc1` (system scope, write-through) and drained by every storing wave before
// the workgroup's one flag store; payload loads are `sc0 sc1` after a poll + workgroup barrier
// (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the hand-off table, at system instead of agent scope).
//
// Epochs.  A flag holds the number of exchanges its chunk has been through.  Workgroup c keeps that number for chunk c in
// memory of its own rank (local[kEpochBase + c]: read when it starts, written when it ends - by the same workgroup index in
// every launch, so there is nothing to agree on inside a launch, no arrival ticket and no "last workgroup").  All ranks
// issue the same sequence of launches, so chunk c's count is the same everywhere.  The optimizer's step number is kept the
// same way: step[2 + c] is workgroup c's private copy, workgroup 0 mirrors it into step[0].
//
// Hazards between launches (no flag is ever reset, epochs only grow): a rank starts its next launch only after the
// current one has finished, i.e. after every owner has read the slots of this exchange (its "reduced" flags say so) - so
// the next pushes never overtake these reads; an owner stores the next "reduced" values of a chunk only after EVERY rank
// has pushed that chunk again, i.e. after every rank has finished reading the current ones.
//
// Deadlock freedom: workgroup c of a rank waits only for workgroup c of other ranks, and pushes before it waits.  The
// host keeps a launch at <= kMaxChunksPerLaunch workgroups (all resident at once) by giving a workgroup several
// 1024-float pieces.  One rank per GPU: while a rank's exchange launch waits, nothing else of that rank needs the device
// (its stream is serial).  Rank processes that SHARE a GPU must each run on CUs of their own (LG_CU_MASK, runtime.hip):
// measured on MI355X, a rank's waiting workgroups spread over all CUs can keep the OTHER rank's kernels - the very ones
// that lead to the launch being waited for - off the device (tests/test_hip_dist.py at MNIST-MLP size: 3 of 3 runs stuck
// until the wait gave up without the masks, 0 of 7 with them).
#include "common.h"
#include "adam_common.h"
#include "../../include/lghip_p2p.h"
#include <cstdlib>
#include <vector>

namespace lg {

constexpr int     kPiece = 1024;                 // floats per piece: one float4 per thread of a 256-thread workgroup
constexpr int     kMaxChunksPerLaunch = 448;
constexpr int     kFlag1Stride = 16;             // ints per chunk: one 64-byte line
constexpr int     kEpochBase = 32;               // local[kEpochBase + c]: exchanges chunk c has been through
constexpr int     SC_SYS = 1 | 16;               // sc0 sc1 on the raw buffer builtins

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P2PCtx {
    int     rank, nranks;
    int64_t cap;                                 // floats per slot
    float*  recv[LG_P2P_MAX_RANKS];              // recv[r]: rank r's window, slot 0
    float*  reduced[LG_P2P_MAX_RANKS];
    int*    flag1[LG_P2P_MAX_RANKS];
    int*    flag2[LG_P2P_MAX_RANKS];
    int*    local;                               // this rank only: [1] dead, [2..6] timeout record, [kEpochBase + c] epoch of chunk c
    int*    status;                              // device status flag (runtime.hip)
    int64_t spin_ticks;                          // a wait gives up after this many ticks of wall_clock64 (100 MHz)
};

__device__ __forceinline__ void st_sys(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int  ld_sys(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// epochs are counted modulo 2^32 (a chunk reaches 2^31 exchanges after ~36 h at 16.7 k steps/s): "flag has reached want" is
// the sign of the 32-bit DIFFERENCE, formed in unsigned arithmetic (signed overflow would be undefined behaviour and lets the
// compiler fold the test into `seen >= want`, true right after the wrap)
__device__ __forceinline__ bool reached(int seen, int want) { return int32_t(uint32_t(seen) - uint32_t(want)) >= 0; }
__device__ __forceinline__ int  next_epoch(int done) { return int(uint32_t(done) + 1u); }

// wait until *flag has reached epoch `want`; gives up after x.spin_ticks.  The FIRST wait that gives up leaves a record for
// the error message: local[2..6] = chunk, epoch wanted, value seen, kind (1 = a peer's push, 2 = the owner's sum), the rank
// waited for - and marks the communicator dead (local[1], never reset: lg_p2p_free is the only way out).  On a dead
// communicator every wait returns after 32 polls AND raises the status flag again: a launch that was already enqueued (or is
// replayed from a hipGraph) after the first report is reported too, never taken for an exchange that happened.
__device__ __forceinline__ void spin_ge(const int* flag, int want, const P2PCtx& x, int chunk, int kind, int from) {
    if (reached(ld_sys(flag), want)) return;
    const int64_t t0 = wall_clock64();
    for (int n = 1;; ++n) {
        __builtin_amdgcn_s_sleep(1);
        if (reached(ld_sys(flag), want)) return;
        if ((n & 31) == 0) {
            if (__hip_atomic_load(x.local + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_fetch_or(x.status, LG_STATUS_P2P_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            if (wall_clock64() - t0 > x.spin_ticks) {
                if (__hip_atomic_exchange(x.local + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    x.local[2] = chunk; x.local[3] = want; x.local[4] = ld_sys(flag); x.local[5] = kind; x.local[6] = from;
                }
                __hip_atomic_fetch_or(x.status, LG_STATUS_P2P_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
    }
}

// four consecutive floats at element `elem` of a slot of `cap` floats: one 16-byte access when `vec`, else dwords
__device__ __forceinline__ void slot_store(float* base, int64_t cap, int64_t elem, int nvalid, bool vec, const float (&v)[4]) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, int(cap * 4), 0x00020000);
    if (vec && nvalid == 4) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(v[e]);
        __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, int(elem * 4), 0, SC_SYS);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), rsrc, int((elem + e) * 4), 0, SC_SYS);
    }
}

__device__ __forceinline__ void slot_load(const float* base, int64_t cap, int64_t elem, int nvalid, bool vec, float (&v)[4]) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, int(cap * 4), 0x00020000);
    if (vec && nvalid == 4) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(elem * 4), 0, SC_SYS);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(w[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e < nvalid ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, int((elem + e) * 4), 0, SC_SYS)) : 0.f;
    }
}

// plain (local bucket) accesses of the same shape
__device__ __forceinline__ void local_load(const float* p, int nvalid, bool vec, float (&v)[4]) {
    if (vec && nvalid == 4) {
        const float4 w = *reinterpret_cast<const float4*>(p);
        v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e < nvalid ? p[e] : 0.f;
    }
}

__device__ __forceinline__ void local_store(float* p, int nvalid, bool vec, const float (&v)[4]) {
    if (vec && nvalid == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) p[e] = v[e];
    }
}

// The exchange of ONE chunk by ONE workgroup (all 256 threads call it).  The chunk is `pieces` pieces of 1024 floats
// starting at element `first` of the bucket `g` (bucket-absolute = slot-relative index), of which elements < `end`
// exist.  On return g[first .. end) holds the reduced values on every rank (the owner's bits), and `last` this thread's
// four reduced values of the chunk's LAST piece (the only one in small launches): a caller that goes on computing with
// them need not wait for its own stores to g.
template <bool kMax>
__device__ __forceinline__ void exchange_chunk(const P2PCtx& x, int epoch, int chunk, float* g, int64_t first, int64_t end, int pieces, bool vec,
                                               float (&last)[4]) {
    const int tid = threadIdx.x, me = x.rank, n = x.nranks, owner = chunk % n;
    if (owner != me) {
        float* slot = x.recv[owner] + int64_t(me) * x.cap;
        for (int q = 0; q < pieces; ++q) {
            const int64_t elem = first + int64_t(q) * kPiece + tid * 4;
            const int nvalid = end - elem >= 4 ? 4 : (end > elem ? int(end - elem) : 0);
            if (nvalid > 0) {
                float v[4];
                local_load(g + elem, nvalid, vec, v);
                slot_store(slot, x.cap, elem, nvalid, vec, v);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) {
            st_sys(x.flag1[owner] + int64_t(chunk) * kFlag1Stride + me, epoch);
            spin_ge(x.flag2[me] + chunk, epoch, x, chunk, 2, owner);
        }
        __syncthreads();
        for (int q = 0; q < pieces; ++q) {
            const int64_t elem = first + int64_t(q) * kPiece + tid * 4;
            const int nvalid = end - elem >= 4 ? 4 : (end > elem ? int(end - elem) : 0);
            if (nvalid > 0) {
                float v[4];
                slot_load(x.reduced[me], x.cap, elem, nvalid, vec, v);
                local_store(g + elem, nvalid, vec, v);
                if (q == pieces - 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) last[e] = v[e];
                }
            }
        }
        return;
    }
    if (tid < n && tid != me) spin_ge(x.flag1[me] + int64_t(chunk) * kFlag1Stride + tid, epoch, x, chunk, 1, tid);
    __syncthreads();
    for (int q = 0; q < pieces; ++q) {
        const int64_t elem = first + int64_t(q) * kPiece + tid * 4;
        const int nvalid = end - elem >= 4 ? 4 : (end > elem ? int(end - elem) : 0);
        if (nvalid > 0) {
            float part[LG_P2P_MAX_RANKS][4];
#pragma unroll
            for (int r = 0; r < LG_P2P_MAX_RANKS; ++r)
                if (r < n) {
                    if (r == me) local_load(g + elem, nvalid, vec, part[r]);
                    else slot_load(x.recv[me] + int64_t(r) * x.cap, x.cap, elem, nvalid, vec, part[r]);
                }
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = part[0][e];
#pragma unroll
            for (int r = 1; r < LG_P2P_MAX_RANKS; ++r)            // rank order: the same sum in every run
                if (r < n) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = kMax ? fmaxf(v[e], part[r][e]) : v[e] + part[r][e];
                }
            local_store(g + elem, nvalid, vec, v);
            if (q == pieces - 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) last[e] = v[e];
            }
#pragma unroll
            for (int r = 0; r < LG_P2P_MAX_RANKS; ++r)
                if (r < n && r != me) slot_store(x.reduced[r], x.cap, elem, nvalid, vec, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < n && tid != me) st_sys(x.flag2[tid] + chunk, epoch);
}

template <bool kMax>
__global__ void __launch_bounds__(256) p2p_allreduce(P2PCtx x, float* buf, int64_t n, int pieces, int vec) {
    const int c = blockIdx.x;
    const int epoch = next_epoch(__hip_atomic_load(x.local + kEpochBase + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int64_t first = int64_t(c) * pieces * kPiece;
    float last[4];
    exchange_chunk<kMax>(x, epoch, c, buf, first, n, pieces, vec != 0, last);
    if (threadIdx.x == 0) __hip_atomic_store(x.local + kEpochBase + c, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the optimizer launch with the exchange in front ------------------------------------------------------------------
constexpr int kP2PMaxSegments = 64;
struct P2PSegments {
    int     nseg, nseg_total, first;             // as AdamSegments (optim.hip)
    int     pieces;                              // 1024-float pieces per workgroup
    int     total_chunks;                        // workgroups with work in this launch
    int     chunk_base[kP2PMaxSegments + 1];     // index of a segment's first chunk among the launch's chunks; [nseg] = all of them
    int64_t offsets[kP2PMaxSegments + 1];
};

__global__ void __launch_bounds__(256) adam_multi_p2p(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                      P2PSegments seg, AdamScalars c, int64_t* __restrict__ step, int step_slot_base, double b1, double b2,
                                                      int base_aligned, P2PCtx x) {
    __shared__ float inv_bias[2];
    // compact grid: one workgroup per chunk with work (a 2-D grid as wide as the longest parameter needs dispatches mostly
    // workgroups that return at once: optim.hip, AdamSegments)
    const int chunk = blockIdx.x;
    int j = 0;
    while (j + 1 < seg.nseg && chunk >= seg.chunk_base[j + 1]) ++j;  // uniform: scalar loads
    const int64_t begin = seg.offsets[j], end = seg.offsets[j + 1];
    const int64_t first = begin + int64_t(chunk - seg.chunk_base[j]) * seg.pieces * kPiece;
    const int epoch = next_epoch(__hip_atomic_load(x.local + kEpochBase + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int64_t* my_step = step + 2 + step_slot_base + chunk;          // this workgroup's own copy of the optimizer's step number
    int64_t steps_done = 0;
    if (threadIdx.x == 0) {
        steps_done = __hip_atomic_load(my_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double t = double(steps_done * seg.nseg_total + seg.first + j + 1);
        inv_bias[0] = float(1.0 / (1.0 - pow(b1, t)));
        inv_bias[1] = float(1.0 / (1.0 - pow(b2, t)));
    }
    const bool vec = base_aligned && (begin & 3) == 0;
    // the LAST piece's parameter and moments are on their way while the exchange waits for the other ranks
    const int64_t tail = first + int64_t(seg.pieces - 1) * kPiece + threadIdx.x * 4;
    const int tail_valid = end - tail >= 4 ? 4 : (end > tail ? int(end - tail) : 0);
    float P0[4], M0[4], V0[4];
    if (tail_valid > 0) {
        local_load(p + tail, tail_valid, vec, P0);
        local_load(m + tail, tail_valid, vec, M0);
        local_load(v + tail, tail_valid, vec, V0);
    }
    float last[4] = {0.f, 0.f, 0.f, 0.f};
    exchange_chunk<false>(x, epoch, chunk, g, first, end, seg.pieces, vec, last);       // barriers inside: inv_bias is visible after it
    c.inv_bias1 = inv_bias[0];
    c.inv_bias2 = inv_bias[1];
    for (int q = 0; q + 1 < seg.pieces; ++q) {
        const int64_t elem = first + int64_t(q) * kPiece + threadIdx.x * 4;
        const int nvalid = end - elem >= 4 ? 4 : (end > elem ? int(end - elem) : 0);
        if (nvalid > 0) {
            float G[4], P[4], M[4], V[4];
            local_load(g + elem, nvalid, vec, G);                  // this thread's own stores of the reduced values
            local_load(p + elem, nvalid, vec, P);
            local_load(m + elem, nvalid, vec, M);
            local_load(v + elem, nvalid, vec, V);
#pragma unroll
            for (int e = 0; e < 4; ++e) adam_elem(P[e], G[e], M[e], V[e], c);
            local_store(p + elem, nvalid, vec, P);
            local_store(m + elem, nvalid, vec, M);
            local_store(v + elem, nvalid, vec, V);
        }
    }
    if (tail_valid > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_elem(P0[e], last[e], M0[e], V0[e], c);
        local_store(p + tail, tail_valid, vec, P0);
        local_store(m + tail, tail_valid, vec, M0);
        local_store(v + tail, tail_valid, vec, V0);
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(x.local + kEpochBase + chunk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(my_step, steps_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (step_slot_base + chunk == 0) __hip_atomic_store(step, steps_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the copy readers see
    }
}

// ---- host state ---------------------------------------------------------------------------------------------------------
struct P2PState {
    bool    exported = false, connected = false;
    bool    failed = false;                      // a wait gave up and was reported: every later lg_p2p_* launch is refused (LG_ECOMM)
    int     rank = -1, nranks = 0;
    int64_t cap = 0;
    int     max_chunks = 0;
    char*   window = nullptr;
    size_t  off_flag1 = 0, off_flag2 = 0, off_recv = 0, off_reduced = 0, bytes = 0;
    void*   peer[LG_P2P_MAX_RANKS] = {};
    int*    local = nullptr;
    const char* memory_kind = "";
    int64_t spin_ticks = int64_t(LG_P2P_TIMEOUT_S) * 100000000;
};

static P2PState& st() {
    static P2PState s;
    return s;
}

static size_t round256(size_t b) { return (b + 255) & ~size_t(255); }

static P2PCtx make_ctx() {
    P2PState& S = st();
    P2PCtx x;
    memset(&x, 0, sizeof(x));
    x.rank = S.rank;
    x.nranks = S.nranks;
    x.cap = S.cap;
    for (int r = 0; r < S.nranks; ++r) {
        char* base = r == S.rank ? S.window : static_cast<char*>(S.peer[r]);
        x.flag1[r] = reinterpret_cast<int*>(base + S.off_flag1);
        x.flag2[r] = reinterpret_cast<int*>(base + S.off_flag2);
        x.recv[r] = reinterpret_cast<float*>(base + S.off_recv);
        x.reduced[r] = reinterpret_cast<float*>(base + S.off_reduced);
    }
    x.local = S.local;
    x.status = rt().status_dev;
    x.spin_ticks = S.spin_ticks;
    return x;
}

// for the error text of a wait that gave up (runtime.hip: check_device_status, after the stream has been synchronised);
// from here on the communicator is FAILED on the host too: no later launch may pass for an exchange (the device-side dead
// flag makes the waits of already enqueued launches short, this makes new ones impossible)
void p2p_describe_timeout(char* out, size_t len) {
    P2PState& S = st();
    S.failed = true;
    int rec[8] = {};
    if (!S.local || hipMemcpy(rec, S.local, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(out, len, "no record");
        return;
    }
    snprintf(out, len, "rank %d of %d: chunk %d waited for %s rank %d to reach exchange %d of that chunk and saw %d", S.rank, S.nranks,
             rec[2], rec[5] == 1 ? "the push of" : "the sum from", rec[6], rec[3], rec[4]);
}

}  // namespace lg

using namespace lg;

#define LG_P2P_ALIVE(who)                                                                                              \
    do {                                                                                                               \
        if (st().failed) {                                                                                             \
            set_error("%s: this peer-window communicator lost a peer earlier (a wait gave up and was reported): nothing is " \
                      "exchanged any more - lg_p2p_disconnect / lg_p2p_free, then start over", who);                   \
            return LG_ECOMM;                                                                                           \
        }                                                                                                              \
    } while (0)

extern "C" int lg_p2p_export(int rank, int nranks, int64_t capacity_floats, char handle[LG_P2P_HANDLE_BYTES]) {
    LG_REQUIRE_INIT();
    P2PState& S = st();
    LG_ARG(!S.exported, "lg_p2p_export: this process already has a window (lg_p2p_free first)");
    LG_ARG(handle != nullptr, "lg_p2p_export: NULL handle buffer");
    LG_ARG(nranks >= 1 && nranks <= LG_P2P_MAX_RANKS && rank >= 0 && rank < nranks, "lg_p2p_export: rank %d of %d (at most %d ranks: one node)",
           rank, nranks, LG_P2P_MAX_RANKS);
    LG_ARG(capacity_floats >= 1 && capacity_floats <= (int64_t(1) << 28), "lg_p2p_export: capacity %lld floats outside 1 .. 2^28", (long long)capacity_floats);
    static_assert(sizeof(hipIpcMemHandle_t) <= LG_P2P_HANDLE_BYTES, "hipIpcMemHandle_t does not fit the handle buffer");
    S.cap = (capacity_floats + kPiece - 1) / kPiece * kPiece;
    S.max_chunks = int(S.cap / kPiece) + kP2PMaxSegments;
    S.off_flag1 = 0;
    S.off_flag2 = round256(S.off_flag1 + size_t(S.max_chunks) * kFlag1Stride * 4);
    S.off_recv = round256(S.off_flag2 + size_t(S.max_chunks) * 4);
    S.off_reduced = round256(S.off_recv + size_t(nranks) * S.cap * 4);
    S.bytes = round256(S.off_reduced + size_t(S.cap) * 4);
    void* w = nullptr;
    S.memory_kind = "uncached";
    if (hipExtMallocWithFlags(&w, S.bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        S.memory_kind = "fine-grained";
        if (hipExtMallocWithFlags(&w, S.bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            S.memory_kind = "hipMalloc";
            LG_HIP(hipMalloc(&w, S.bytes));
        }
    }
    S.window = static_cast<char*>(w);
    LG_HIP(hipMemset(S.window, 0, S.bytes));
    const size_t local_bytes = size_t(kEpochBase + S.max_chunks) * sizeof(int);
    LG_HIP(hipMalloc(reinterpret_cast<void**>(&S.local), local_bytes));
    LG_HIP(hipMemset(S.local, 0, local_bytes));
    LG_HIP(hipDeviceSynchronize());                               // the zeros are in place before any peer learns the handle
    hipIpcMemHandle_t h;
    LG_HIP(hipIpcGetMemHandle(&h, S.window));
    memset(handle, 0, LG_P2P_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    S.rank = rank;
    S.nranks = nranks;
    S.exported = true;
    if (const char* ms = getenv("LG_P2P_TIMEOUT_MS"))             // tests of the lost-peer path
        if (atoll(ms) > 0) S.spin_ticks = atoll(ms) * 100000;
    return LG_OK;
}

extern "C" int lg_p2p_connect(const char* handles) {
    LG_REQUIRE_INIT();
    P2PState& S = st();
    LG_ARG(S.exported && !S.connected, "lg_p2p_connect: call lg_p2p_export first (once)");
    LG_ARG(handles != nullptr, "lg_p2p_connect: NULL");
    for (int r = 0; r < S.nranks; ++r) {
        if (r == S.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + size_t(r) * LG_P2P_HANDLE_BYTES, sizeof(h));
        hipError_t e = hipIpcOpenMemHandle(&S.peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; ++q)
                if (q != S.rank && S.peer[q]) { (void)hipIpcCloseMemHandle(S.peer[q]); S.peer[q] = nullptr; }
            set_error("lg_p2p_connect: hipIpcOpenMemHandle of rank %d's window failed: %s", r, hipGetErrorString(e));
            return LG_ECOMM;
        }
    }
    S.connected = true;
    return LG_OK;
}

extern "C" int lg_p2p_rank(int* rank, int* nranks, int64_t* capacity_floats) {
    P2PState& S = st();
    LG_ARG(S.exported, "lg_p2p_rank: no window");
    if (rank) *rank = S.rank;
    if (nranks) *nranks = S.nranks;
    if (capacity_floats) *capacity_floats = S.cap;
    return LG_OK;
}

extern "C" int lg_p2p_state(int* failed, char memory_kind[32]) {
    P2PState& S = st();
    LG_ARG(S.exported, "lg_p2p_state: no window");
    if (failed) *failed = S.failed ? 1 : 0;
    if (memory_kind) snprintf(memory_kind, 32, "%s", S.memory_kind);
    return LG_OK;
}

// tests only: pretend every chunk has been through `epoch` exchanges (own window's flags and this rank's counts).  Every
// rank calls it with the same value while no launch is in flight, and the ranks synchronise on the host before the next one.
extern "C" int lg_p2p_debug_seed_epochs(int epoch) {
    LG_REQUIRE_INIT();
    P2PState& S = st();
    LG_ARG(S.connected, "lg_p2p_debug_seed_epochs: lg_p2p_export / lg_p2p_connect first");
    LG_HIP(hipStreamSynchronize(rt().stream));
    std::vector<int> flags(size_t(S.max_chunks) * kFlag1Stride, epoch);
    LG_HIP(hipMemcpy(S.window + S.off_flag1, flags.data(), size_t(S.max_chunks) * kFlag1Stride * sizeof(int), hipMemcpyHostToDevice));
    LG_HIP(hipMemcpy(S.window + S.off_flag2, flags.data(), size_t(S.max_chunks) * sizeof(int), hipMemcpyHostToDevice));
    LG_HIP(hipMemcpy(S.local + kEpochBase, flags.data(), size_t(S.max_chunks) * sizeof(int), hipMemcpyHostToDevice));
    LG_HIP(hipDeviceSynchronize());
    return LG_OK;
}

extern "C" int lg_p2p_allreduce_f32(float* buf, int64_t n, int op) {
    LG_REQUIRE_INIT();
    P2PState& S = st();
    LG_ARG(S.connected, "lg_p2p_allreduce_f32: lg_p2p_export / lg_p2p_connect first");
    LG_P2P_ALIVE("lg_p2p_allreduce_f32");
    LG_ARG(n >= 0 && (n == 0 || buf), "lg_p2p_allreduce_f32: bad buffer");
    LG_ARG(op == LG_P2P_SUM || op == LG_P2P_MAX, "lg_p2p_allreduce_f32: unknown op %d", op);
    const P2PCtx x = make_ctx();
    const int vec = aligned16(buf) ? 1 : 0;
    for (int64_t done = 0; done < n; done += S.cap) {             // a launch moves at most one window's worth
        const int64_t len = n - done < S.cap ? n - done : S.cap;
        const int64_t units = (len + kPiece - 1) / kPiece;
        const int pieces = int((units + kMaxChunksPerLaunch - 1) / kMaxChunksPerLaunch);
        const unsigned grid = unsigned((units + pieces - 1) / pieces);
        if (op == LG_P2P_SUM) hipLaunchKernelGGL(p2p_allreduce<false>, dim3(grid), dim3(256), 0, rt().stream, x, buf + done, len, pieces, vec);
        else                  hipLaunchKernelGGL(p2p_allreduce<true>, dim3(grid), dim3(256), 0, rt().stream, x, buf + done, len, pieces, vec);
        LG_CHECK_LAUNCH();
    }
    return LG_OK;
}

extern "C" int lg_p2p_adam_multi_dev_f32(float* p, float* g, float* m, float* v, int nseg, const int64_t* offsets,
                                         double lr, double b1, double b2, double eps, int64_t* step, int64_t step_slots, double gscale,
                                         int belief) {
    LG_REQUIRE_INIT();
    P2PState& S = st();
    LG_ARG(S.connected, "lg_p2p_adam_multi_dev_f32: lg_p2p_export / lg_p2p_connect first");
    LG_P2P_ALIVE("lg_p2p_adam_multi_dev_f32");
    LG_ARG(nseg >= 1, "lg_p2p_adam_multi_dev_f32: %d segments", nseg);
    LG_ARG(p && g && m && v && step && offsets, "lg_p2p_adam_multi_dev_f32: NULL pointer");
    for (int j = 0; j < nseg; ++j) LG_ARG(offsets[j + 1] >= offsets[j], "lg_p2p_adam_multi_dev_f32: offsets must be non-decreasing");
    LG_ARG(offsets[0] >= 0 && offsets[nseg] <= S.cap, "lg_p2p_adam_multi_dev_f32: the bucket (%lld floats) exceeds the window (%lld)",
           (long long)offsets[nseg], (long long)S.cap);
    const P2PCtx x = make_ctx();
    const AdamScalars c = adam_scalars(lr, b1, b2, eps, 0.0, 0.0, gscale, belief);
    const int base_aligned = (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) ? 1 : 0;
    int slot_base = 0;
    for (int first = 0; first < nseg; first += kP2PMaxSegments) {  // groups of segments: one launch each
        const int count = nseg - first < kP2PMaxSegments ? nseg - first : kP2PMaxSegments;
        P2PSegments seg;
        seg.nseg = count;
        seg.nseg_total = nseg;
        seg.first = first;
        int64_t units = 0, longest = 0;
        for (int j = 0; j < count; ++j) {
            const int64_t len = offsets[first + j + 1] - offsets[first + j];
            units += (len + kPiece - 1) / kPiece;
            if (len > longest) longest = len;
        }
        for (int j = 0; j <= count; ++j) seg.offsets[j] = offsets[first + j];
        if (longest == 0) continue;
        // pieces per workgroup: smallest count that keeps the launch's workgroups with work within the resident limit
        int pieces = int((units + kMaxChunksPerLaunch - 1) / kMaxChunksPerLaunch);
        for (;; ++pieces) {
            int64_t chunks = 0;
            for (int j = 0; j < count; ++j) chunks += (offsets[first + j + 1] - offsets[first + j] + int64_t(pieces) * kPiece - 1) / (int64_t(pieces) * kPiece);
            if (chunks <= kMaxChunksPerLaunch + count) break;
        }
        seg.pieces = pieces;
        int total = 0;
        for (int j = 0; j < count; ++j) {
            seg.chunk_base[j] = total;
            total += int((offsets[first + j + 1] - offsets[first + j] + int64_t(pieces) * kPiece - 1) / (int64_t(pieces) * kPiece));
        }
        seg.chunk_base[count] = total;
        seg.total_chunks = total;
        LG_ARG(total <= S.max_chunks, "lg_p2p_adam_multi_dev_f32: %d chunks exceed the window's %d flags", total, S.max_chunks);
        LG_ARG(slot_base + total <= step_slots, "lg_p2p_adam_multi_dev_f32: %d chunks need as many step slots, the caller gave %lld (lghip_p2p.h)",
               slot_base + total, (long long)step_slots);
        hipLaunchKernelGGL(adam_multi_p2p, dim3(unsigned(total)), dim3(256), 0, rt().stream, p, g, m, v, seg, c, step, slot_base, b1, b2, base_aligned, x);
        LG_CHECK_LAUNCH();
        slot_base += total;
    }
    return LG_OK;
}

extern "C" int lg_p2p_disconnect(void) {
    P2PState& S = st();
    if (!S.connected) return LG_OK;
    LG_HIP(hipStreamSynchronize(rt().stream));
    for (int r = 0; r < S.nranks; ++r)
        if (r != S.rank && S.peer[r]) { (void)hipIpcCloseMemHandle(S.peer[r]); S.peer[r] = nullptr; }
    S.connected = false;
    return LG_OK;
}

extern "C" int lg_p2p_free(void) {
    P2PState& S = st();
    if (!S.exported) return LG_OK;
    LG_ARG(!S.connected, "lg_p2p_free: lg_p2p_disconnect first");
    (void)hipFree(S.window);
    (void)hipFree(S.local);
    S = P2PState();
    return LG_OK;
}
