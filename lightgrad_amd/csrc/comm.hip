// liblghip_comm.so: RCCL collectives on the compute stream of liblghip.so (lghip_comm.h).
// The reference has no distributed code; this is the data-parallel exchange
// step of BASELINE config #4: ONE fp32 all-reduce per step over the flat gradient bucket.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include "../../include/lghip.h"
#include "../../include/lghip_comm.h"

static thread_local char g_cerr[1024] = "";
static ncclComm_t g_comm = nullptr;
static int g_rank = -1, g_nranks = 0;
// The gradient exchange runs on its own stream, forked from and joined to the compute stream with events, so that
// it overlaps the backward kernels still to come (the input-gradient GEMM of the first layer).  While the compute
// stream is being captured the two event hand-offs pull this stream into the capture: the collective becomes a
// parallel branch of the hipGraph and a replayed step costs no host call for it.
static hipStream_t g_comm_stream = nullptr;
static hipEvent_t  g_fork_ev = nullptr, g_join_ev = nullptr;

static void cerr(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cerr, sizeof(g_cerr), fmt, ap);
    va_end(ap);
}

#define LG_CHIP(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            cerr("%s: %s failed: %s", __func__, #expr, hipGetErrorString(_e));         \
            return LG_EHIP;                                                            \
        }                                                                              \
    } while (0)

#define LG_NCCL(expr)                                                                  \
    do {                                                                               \
        ncclResult_t _r = (expr);                                                      \
        if (_r != ncclSuccess) {                                                       \
            cerr("%s: %s failed: %s", __func__, #expr, ncclGetErrorString(_r));        \
            return LG_ECOMM;                                                           \
        }                                                                              \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= LG_COMM_ID_BYTES, "ncclUniqueId does not fit the id buffer");

extern "C" {

const char* lg_comm_last_error(void) { return g_cerr; }

int lg_comm_get_unique_id(char id[LG_COMM_ID_BYTES]) {
    if (!id) { cerr("lg_comm_get_unique_id: NULL"); return LG_EINVAL; }
    ncclUniqueId uid;
    LG_NCCL(ncclGetUniqueId(&uid));
    memset(id, 0, LG_COMM_ID_BYTES);
    memcpy(id, &uid, sizeof(uid));
    return LG_OK;
}

int lg_comm_init(int rank, int nranks, const char id[LG_COMM_ID_BYTES]) {
    if (g_comm) { cerr("lg_comm_init: communicator already initialised"); return LG_EINVAL; }
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) { cerr("lg_comm_init: bad arguments (rank %d of %d)", rank, nranks); return LG_EINVAL; }
    if (lg_stream() == nullptr) { cerr("lg_comm_init: lg_init() has not been called"); return LG_ENOTINIT; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    LG_NCCL(ncclCommInitRank(&g_comm, nranks, uid, rank));
    g_rank = rank;
    g_nranks = nranks;
    if (!g_comm_stream) {
        int lo = 0, hi = 0;                     // numerically lowest = highest priority: the few workgroups of the
        LG_CHIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // collective get their CUs while a GEMM fills the chip
        LG_CHIP(hipStreamCreateWithPriority(&g_comm_stream, hipStreamNonBlocking, hi));
        LG_CHIP(hipEventCreateWithFlags(&g_fork_ev, hipEventDisableTiming));
        LG_CHIP(hipEventCreateWithFlags(&g_join_ev, hipEventDisableTiming));
    }
    return LG_OK;
}

// The first collective of a new communicator, on the COMMUNICATION stream, awaited by polling an event: a collective that
// never completes (a peer that fell out during init) is reported after `timeout_s` instead of blocking the caller, and the
// compute stream - which the other forms of the exchange need - has nothing of it queued.
int lg_comm_selftest(double timeout_s) {
    if (!g_comm) { cerr("lg_comm_selftest: communicator not initialised"); return LG_ENOTINIT; }
    float* token = nullptr;
    LG_CHIP(hipMalloc(reinterpret_cast<void**>(&token), sizeof(float)));
    const float one = 1.0f;
    LG_CHIP(hipMemcpy(token, &one, sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t done = nullptr;
    LG_CHIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    {
        ncclResult_t r = ncclAllReduce(token, token, 1, ncclFloat32, ncclSum, g_comm, g_comm_stream);
        if (r != ncclSuccess) {
            cerr("lg_comm_selftest: ncclAllReduce failed: %s", ncclGetErrorString(r));
            (void)hipEventDestroy(done);
            (void)hipFree(token);
            return LG_ECOMM;
        }
    }
    LG_CHIP(hipEventRecord(done, g_comm_stream));
    const double step_us = 200.0;
    double waited_us = 0.0;
    for (;;) {
        hipError_t q = hipEventQuery(done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            cerr("lg_comm_selftest: hipEventQuery failed: %s", hipGetErrorString(q));
            return LG_EHIP;                       // token and event are left alone: the stream may still refer to them
        }
        if (waited_us > timeout_s * 1e6) {
            cerr("lg_comm_selftest: the first all-reduce over %d ranks did not complete within %.0f s", g_nranks, timeout_s);
            return LG_ECOMM;                      // likewise left alone
        }
        struct timespec ts = {0, long(step_us * 1000)};
        nanosleep(&ts, nullptr);
        waited_us += step_us;
    }
    float sum = 0.f;
    LG_CHIP(hipMemcpy(&sum, token, sizeof(float), hipMemcpyDeviceToHost));
    (void)hipEventDestroy(done);
    (void)hipFree(token);
    if (sum != float(g_nranks)) {
        cerr("lg_comm_selftest: the all-reduce of 1.0 over %d ranks gave %g", g_nranks, double(sum));
        return LG_ECOMM;
    }
    return LG_OK;
}

int lg_comm_fork(void) {
    if (!g_comm) { cerr("lg_comm_fork: communicator not initialised"); return LG_ENOTINIT; }
    hipStream_t compute = static_cast<hipStream_t>(lg_stream());
    LG_CHIP(hipEventRecord(g_fork_ev, compute));
    LG_CHIP(hipStreamWaitEvent(g_comm_stream, g_fork_ev, 0));
    return LG_OK;
}

int lg_comm_join(void) {
    if (!g_comm) { cerr("lg_comm_join: communicator not initialised"); return LG_ENOTINIT; }
    hipStream_t compute = static_cast<hipStream_t>(lg_stream());
    LG_CHIP(hipEventRecord(g_join_ev, g_comm_stream));
    LG_CHIP(hipStreamWaitEvent(compute, g_join_ev, 0));
    return LG_OK;
}

int lg_comm_rank(int* rank, int* nranks) {
    if (!g_comm) { cerr("lg_comm_rank: communicator not initialised"); return LG_ENOTINIT; }
    if (rank) *rank = g_rank;
    if (nranks) *nranks = g_nranks;
    return LG_OK;
}

// experiments only (LG_COMM_DEBUG_KERNEL=1): a one-wave kernel behind every collective, on the collective's stream.  With ONE
// rank ncclAllReduce in place enqueues nothing, so a captured "forked" exchange is an empty branch and says nothing about what
// a branch with a kernel in it costs in a hipGraph; this makes the branch real on a single-GPU box (bench.py --force-comm).
__global__ void comm_debug_touch(float* buf) {
    if (threadIdx.x == 0) buf[0] = buf[0] * 1.0f;
}

static int allreduce_on(hipStream_t stream, float* buf, int64_t n, int op, const char* who) {
    if (!g_comm) { cerr("%s: communicator not initialised", who); return LG_ENOTINIT; }
    if (n < 0 || (n > 0 && !buf)) { cerr("%s: bad buffer", who); return LG_EINVAL; }
    if (op != LG_COMM_SUM && op != LG_COMM_MAX) { cerr("%s: unknown op %d", who, op); return LG_EINVAL; }
    if (n == 0) return LG_OK;
    LG_NCCL(ncclAllReduce(buf, buf, size_t(n), ncclFloat32, op == LG_COMM_SUM ? ncclSum : ncclMax, g_comm, stream));
    static const char* dbg = getenv("LG_COMM_DEBUG_KERNEL");
    if (dbg && atoi(dbg) == 1) hipLaunchKernelGGL(comm_debug_touch, dim3(1), dim3(64), 0, stream, buf);
    return LG_OK;
}

int lg_comm_allreduce_f32(float* buf, int64_t n, int op) {
    return allreduce_on(static_cast<hipStream_t>(lg_stream()), buf, n, op, "lg_comm_allreduce_f32");
}

int lg_comm_allreduce_forked_f32(float* buf, int64_t n, int op) {
    return allreduce_on(g_comm_stream, buf, n, op, "lg_comm_allreduce_forked_f32");
}

int lg_comm_broadcast_f32(float* buf, int64_t n, int root) {
    if (!g_comm) { cerr("lg_comm_broadcast_f32: communicator not initialised"); return LG_ENOTINIT; }
    if (n < 0 || (n > 0 && !buf) || root < 0 || root >= g_nranks) { cerr("lg_comm_broadcast_f32: bad arguments"); return LG_EINVAL; }
    if (n == 0) return LG_OK;
    LG_NCCL(ncclBroadcast(buf, buf, size_t(n), ncclFloat32, root, g_comm, static_cast<hipStream_t>(lg_stream())));
    return LG_OK;
}

int lg_comm_destroy(void) {
    if (!g_comm) return LG_OK;
    (void)hipStreamSynchronize(static_cast<hipStream_t>(lg_stream()));
    if (g_comm_stream) (void)hipStreamSynchronize(g_comm_stream);
    LG_NCCL(ncclCommDestroy(g_comm));
    g_comm = nullptr;
    g_rank = -1;
    g_nranks = 0;
    return LG_OK;
}

}  // extern "C"
