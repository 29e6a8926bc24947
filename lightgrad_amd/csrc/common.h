// Internal helpers shared by the translation units of liblghip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/lghip.h"

namespace lg {

// thread-local error text returned by lg_last_error()
void set_error(const char* fmt, ...);

struct Runtime {
    bool        ready = false;
    int         device = -1;
    hipStream_t stream = nullptr;
    int         compute_units = 0;
    // arrival counters of the split-K GEMM (one per output tile): zero between launches - the workgroup that
    // arrives last at a tile folds the partial products and resets the counter
    int*        gemm_tickets = nullptr;
    int         n_gemm_tickets = 0;
    // hand-off counters of the fused attention backward (attention.hip): two per (batch, head) pair, zero between launches
    int*        attn_flags = nullptr;
    int         n_attn_pairs = 1 << 14;
    // device status flag: one int in pinned, device-mapped host memory.  Kernels that meet an index / label out of
    // range OR a bit into it (system scope); lg_sync / lg_memcpy_d2h read it from the host side after their stream
    // synchronisation and report LG_EINDEX once.
    int*        status_host = nullptr;
    int*        status_dev = nullptr;
};

constexpr int LG_STATUS_BAD_INDEX = 1;
constexpr int LG_STATUS_HANDOFF_TIMEOUT = 4; // a workgroup of attention.hip's backward gave up waiting for its producers
constexpr int LG_STATUS_P2P_TIMEOUT = 2;     // a wait of the peer-window exchange (p2p.hip) gave up: a peer is gone

// after a stream synchronisation: turn a raised status flag into an error (and clear it)
int check_device_status(const char* who);
Runtime& rt();

// true between lg_graph_begin and lg_graph_end
bool capturing();

// lg_gemm_group_* (csrc/gemm.hip): is a group open / launch everything it has queued - GEMM products and the LayerNorm
// parameter gradients that ride along (csrc/rowwise.hip).  Called before anything that exposes results.
bool gemm_group_is_open();
int gemm_group_flush_pending();
int ln_group_flush_pending();
bool ln_group_writes(const void* ptr);
struct TailGroup;
bool tail_take(TailGroup* out);       // rowwise.hip: the queued LayerNorm / scatter jobs as one argument block ...
int tail_taken();                     // ... and, once launched, out of the queue

// p2p.hip: what the first wait of the peer-window exchange that gave up was waiting for
void p2p_describe_timeout(char* out, size_t len);

}  // namespace lg

#define LG_REQUIRE_INIT()                                                       \
    do {                                                                        \
        if (!lg::rt().ready) {                                                  \
            lg::set_error("%s: lg_init() has not been called", __func__);       \
            return LG_ENOTINIT;                                                 \
        }                                                                       \
    } while (0)

#define LG_HIP(expr)                                                            \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            lg::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(_e)); \
            return LG_EHIP;                                                     \
        }                                                                       \
    } while (0)

#define LG_CHECK_LAUNCH()                                                       \
    do {                                                                        \
        hipError_t _e = hipGetLastError();                                      \
        if (_e != hipSuccess) {                                                 \
            lg::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(_e)); \
            return LG_EHIP;                                                     \
        }                                                                       \
    } while (0)

#define LG_ARG(cond, ...)                                                       \
    do {                                                                        \
        if (!(cond)) {                                                          \
            lg::set_error(__VA_ARGS__);                                         \
            return LG_EINVAL;                                                   \
        }                                                                       \
    } while (0)

namespace lg {

// ---- strided iteration descriptor ------------------------------------------
// Up to NOPS operands over a common (collapsed) index space.  Dimension 0 is the
// outermost.  Passed to kernels by value (kernarg segment).
constexpr int kMaxOps = 6;

struct IterDesc {
    int     ndim;                           // after collapsing, >= 1
    int64_t numel;
    int64_t shape[LG_MAX_DIMS];
    int64_t stride[kMaxOps][LG_MAX_DIMS];   // elements; 0 = broadcast
};

// Build a collapsed descriptor: drops size-1 dims and merges neighbouring dims that
// are jointly contiguous for every operand.  strides[i] == nullptr marks an unused
// operand (its strides become 0).  Returns false if ndim is out of range or any
// extent is negative.
bool build_iter(int ndim, const int64_t* shape, const int64_t* const* strides, int nops, IterDesc& out);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid size for a memory-bound kernel: ONE work item (one float4) per thread.  Measured on MI355X for
// c = a + b over 512 MiB tensors (tools/stream_bench.hip): one float4 per thread over 131072 workgroups streams
// 6.08 TB/s, a grid-stride loop over 2048 workgroups 5.06 TB/s, 4-8 float4 per thread 5.5-5.6 TB/s.  The
// kernels keep their grid-stride loop only to stay correct when the cap below is hit (> 4G work items).
inline unsigned stream_grid(int64_t work_items, int block = 256) {
    int64_t need = (work_items + block - 1) / block;
    int64_t cap = int64_t(1) << 22;
    if (need < 1) need = 1;
    return static_cast<unsigned>(need < cap ? need : cap);
}

}  // namespace lg
