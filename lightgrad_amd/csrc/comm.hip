// liblghip_comm.so: RCCL collectives on the compute stream of liblghip.so (lghip_comm.h).
// The reference has no distributed code; this is the data-parallel exchange
// step of BASELINE config #4: ONE fp32 all-reduce per step over the flat gradient bucket.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/lghip.h"
#include "../../include/lghip_comm.h"

static thread_local char g_cerr[1024] = "";
static ncclComm_t g_comm = nullptr;
static int g_rank = -1, g_nranks = 0;

static void cerr(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cerr, sizeof(g_cerr), fmt, ap);
    va_end(ap);
}

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
    return LG_OK;
}

int lg_comm_rank(int* rank, int* nranks) {
    if (!g_comm) { cerr("lg_comm_rank: communicator not initialised"); return LG_ENOTINIT; }
    if (rank) *rank = g_rank;
    if (nranks) *nranks = g_nranks;
    return LG_OK;
}

int lg_comm_allreduce_f32(float* buf, int64_t n, int op) {
    if (!g_comm) { cerr("lg_comm_allreduce_f32: communicator not initialised"); return LG_ENOTINIT; }
    if (n < 0 || (n > 0 && !buf)) { cerr("lg_comm_allreduce_f32: bad buffer"); return LG_EINVAL; }
    if (op != LG_COMM_SUM && op != LG_COMM_MAX) { cerr("lg_comm_allreduce_f32: unknown op %d", op); return LG_EINVAL; }
    if (n == 0) return LG_OK;
    LG_NCCL(ncclAllReduce(buf, buf, size_t(n), ncclFloat32, op == LG_COMM_SUM ? ncclSum : ncclMax, g_comm,
                          static_cast<hipStream_t>(lg_stream())));
    return LG_OK;
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
    LG_NCCL(ncclCommDestroy(g_comm));
    g_comm = nullptr;
    g_rank = -1;
    g_nranks = 0;
    return LG_OK;
}

}  // extern "C"
