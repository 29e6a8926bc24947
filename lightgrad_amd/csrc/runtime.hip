// Runtime of liblghip.so: device binding, the single compute stream, a caching
// stream-ordered allocator, transfers, events and hipGraph capture.
//
// Design analog in the reference: OpenCLDevice = context + in-order queue +
// cl.tools.MemoryPool (opencl/device.py:68-115).  Nothing is translated from it:
// the pool below is a best-fit free list keyed by size, safe without events
// because every consumer of a block is enqueued on the one library stream.
#include "common.h"
#include "../../include/lghip_p2p.h"
#include <cstdarg>
#include <map>
#include <unordered_map>
#include <vector>
#include <mutex>

namespace lg {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

Runtime& rt() {
    static Runtime r;
    return r;
}

// ---------------------------------------------------------------------------
// allocator
// ---------------------------------------------------------------------------
struct Block {
    size_t bytes;        // rounded size
    int    graph;        // 0 = general pool, >0 = pinned to that captured graph
};

struct GraphRec {
    hipGraph_t      graph = nullptr;
    hipGraphExec_t  exec = nullptr;
    int             id = 0;
    std::multimap<size_t, void*> free_blocks;   // blocks owned by the graph that are currently unused
};

struct Pool {
    std::mutex mu;
    std::unordered_map<void*, Block> live;            // handed out
    std::multimap<size_t, void*>     free_blocks;     // cached, general
    std::unordered_map<void*, Block> all;             // every hipMalloc'ed block (live or cached)
    uint64_t reserved = 0, in_use = 0, hip_mallocs = 0;
    // graph capture state
    GraphRec* capture = nullptr;
    int next_graph_id = 1;
    std::unordered_map<int, GraphRec*> graphs;
};

static Pool& pool() {
    static Pool p;
    return p;
}

bool capturing() { return pool().capture != nullptr; }

int check_device_status(const char* who) {
    Runtime& R = rt();
    if (R.status_host == nullptr) return LG_OK;
    const int status = __atomic_exchange_n(R.status_host, 0, __ATOMIC_ACQ_REL);
    if (status == 0) return LG_OK;
    if (status & LG_STATUS_P2P_TIMEOUT) {
        char what[256];
        p2p_describe_timeout(what, sizeof(what));
        set_error("%s: a peer-window exchange launched earlier gave up waiting for another rank (lghip_p2p.h; %s); the gradient "
                  "buckets and parameters of this rank are not to be trusted (device status %d)", who, what, status);
        return LG_ECOMM;
    }
    if (status & LG_STATUS_HANDOFF_TIMEOUT) {
        // (attention.hip's backward, gemm.hip's product + head rows.  The tickets such a launch leaves behind are not at zero: reset
        //  the pool so that later launches start clean - the stream is idle here, status is only looked at after a synchronisation)
        if (R.gemm_tickets) (void)hipMemset(R.gemm_tickets, 0, size_t(R.n_gemm_tickets) * sizeof(int));
        set_error("%s: a launch whose workgroups wait for other workgroups of the same launch (the attention backward, the hidden layer's "
                  "product followed by the output layer's rows) gave up waiting (2 s; device status %d): its results are not to be trusted",
                  who, status);
        return LG_EHIP;
    }
    set_error("%s: a kernel launched earlier met an index or label outside its axis (device status %d); results of that "
              "launch hold all-ones bytes / NaN where the index was bad", who, status);
    return LG_EINDEX;
}

static size_t round_size(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes <= (1u << 20)) return (bytes + 511) & ~size_t(511);          // 512 B granules up to 1 MiB
    return (bytes + (2u << 20) - 1) & ~size_t((2u << 20) - 1);             // 2 MiB granules above
}

// best fit from a free list; large requests refuse blocks that would waste > 25 %
static void* take_from(std::multimap<size_t, void*>& fl, size_t need) {
    auto it = fl.lower_bound(need);
    if (it == fl.end()) return nullptr;
    if (need > (1u << 20) && it->first > need + need / 4) return nullptr;
    void* p = it->second;
    fl.erase(it);
    return p;
}

static void trim_locked(Pool& P) {
    for (auto& kv : P.free_blocks) {
        (void)hipFree(kv.second);
        P.reserved -= kv.first;
        P.all.erase(kv.second);
    }
    P.free_blocks.clear();
}

}  // namespace lg

using namespace lg;

extern "C" {

const char* lg_last_error(void) { return g_err; }

const char* lg_version(void) { return "liblghip 0.1 gfx950 " __DATE__; }

int lg_device_count(int* count) {
    LG_ARG(count != nullptr, "lg_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return LG_OK;
}

int lg_peer_info(int device, int peer, int* can_access, int* link_type, int* hops) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    LG_ARG(device >= 0 && device < n && peer >= 0 && peer < n, "lg_peer_info: devices %d, %d of %d visible", device, peer, n);
    int can = device == peer ? 1 : 0;
    if (device != peer && hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) { (void)hipGetLastError(); can = 0; }
    uint32_t type = 0, hop = 0;
    int t = -1, h = -1;
    if (device != peer) {
        if (hipExtGetLinkTypeAndHopCount(device, peer, &type, &hop) == hipSuccess) { t = int(type); h = int(hop); }
        else (void)hipGetLastError();
    } else {
        h = 0;
    }
    if (can_access) *can_access = can;
    if (link_type) *link_type = t;
    if (hops) *hops = h;
    return LG_OK;
}

int lg_init(int device) {
    Runtime& R = rt();
    if (R.ready) {
        LG_ARG(R.device == device, "lg_init: already bound to device %d (asked for %d)", R.device, device);
        return LG_OK;
    }
    int n = 0;
    LG_HIP(hipGetDeviceCount(&n));
    LG_ARG(device >= 0 && device < n, "lg_init: device %d out of range (%d visible)", device, n);
    LG_HIP(hipSetDevice(device));
    // LG_CU_MASK=<k>/<n>: the compute stream runs on part k of n equal parts of the CUs.  For rank processes that SHARE a GPU
    // (tests and rehearsals of the data-parallel path on a one-GPU box): each rank then owns its CUs like a GPU of its own,
    // and a kernel of one rank that waits for another (csrc/p2p.hip) can never keep that rank's kernels off the device.
    int part = -1, parts = 0;
    if (const char* spec = getenv("LG_CU_MASK")) {
        LG_ARG(sscanf(spec, "%d/%d", &part, &parts) == 2 && parts >= 1 && part >= 0 && part < parts, "lg_init: LG_CU_MASK=%s is not <k>/<n> with 0 <= k < n", spec);
    }
    if (parts > 1) {
        hipDeviceProp_t prop0;
        LG_HIP(hipGetDeviceProperties(&prop0, device));
        const int cus = prop0.multiProcessorCount, words = (cus + 31) / 32;
        LG_ARG(parts <= cus, "lg_init: LG_CU_MASK asks for %d parts of %d CUs", parts, cus);
        std::vector<uint32_t> mask(words, 0u);
        for (int cu = part * cus / parts; cu < (part + 1) * cus / parts; ++cu) mask[cu / 32] |= 1u << (cu % 32);
        LG_HIP(hipExtStreamCreateWithCUMask(&R.stream, uint32_t(words), mask.data()));
    } else {
        LG_HIP(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
    }
    hipDeviceProp_t prop;
    LG_HIP(hipGetDeviceProperties(&prop, device));
    R.compute_units = prop.multiProcessorCount;
    R.n_gemm_tickets = 1 << 16;
    LG_HIP(hipMalloc(reinterpret_cast<void**>(&R.gemm_tickets), size_t(R.n_gemm_tickets) * sizeof(int)));
    LG_HIP(hipMemset(R.gemm_tickets, 0, size_t(R.n_gemm_tickets) * sizeof(int)));    // synchronous: ordered before any launch
    LG_HIP(hipMalloc(reinterpret_cast<void**>(&R.attn_flags), size_t(R.n_attn_pairs) * 2 * sizeof(int)));
    LG_HIP(hipMemset(R.attn_flags, 0, size_t(R.n_attn_pairs) * 2 * sizeof(int)));
    LG_HIP(hipHostMalloc(reinterpret_cast<void**>(&R.status_host), 64, hipHostMallocMapped));
    R.status_host[0] = 0;
    LG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&R.status_dev), R.status_host, 0));
    R.device = device;
    R.ready = true;
    return LG_OK;
}

int lg_device(int* device) {
    LG_REQUIRE_INIT();
    LG_ARG(device != nullptr, "lg_device: NULL");
    *device = rt().device;
    return LG_OK;
}

int lg_device_info(lg_device_info_t* out) {
    LG_REQUIRE_INIT();
    LG_ARG(out != nullptr, "lg_device_info: NULL");
    hipDeviceProp_t prop;
    LG_HIP(hipGetDeviceProperties(&prop, rt().device));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", prop.gcnArchName);
    // gcnArchName looks like "gfx950:sramecc+:xnack-": keep the target id only
    if (char* colon = strchr(out->arch, ':')) *colon = '\0';
    if (out->name[0] == '\0') snprintf(out->name, sizeof(out->name), "AMD Instinct (%s)", out->arch);
    out->compute_units = prop.multiProcessorCount;
    out->clock_mhz = prop.clockRate / 1000;
    out->wavefront_size = prop.warpSize;
    out->lds_bytes_per_cu = static_cast<int32_t>(prop.maxSharedMemoryPerMultiProcessor);
    out->hbm_bytes = prop.totalGlobalMem;
    out->l2_bytes = prop.l2CacheSize;
    return LG_OK;
}

void* lg_stream(void) { return rt().ready ? static_cast<void*>(rt().stream) : nullptr; }

// queued work (lg_gemm_group_*: weight-gradient products, LayerNorm parameter gradients, embedding scatter-adds) is launched
// before anything that exposes results to the host, to another graph or to the allocator's trim
static int flush_queued() { return lg::gemm_group_flush_pending(); }

int lg_sync(void) {
    LG_REQUIRE_INIT();
    LG_ARG(!capturing(), "lg_sync: not allowed while capturing a graph");
    { const int rc = flush_queued(); if (rc != LG_OK) return rc; }
    LG_HIP(hipStreamSynchronize(rt().stream));
    return check_device_status("lg_sync");
}

// ---- allocator ---------------------------------------------------------------

int lg_malloc(void** ptr, size_t bytes) {
    LG_REQUIRE_INIT();
    LG_ARG(ptr != nullptr, "lg_malloc: ptr is NULL");
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    // 16 bytes of slack behind every block: a float4 load that starts inside a tensor's last row may run up to 12 bytes
    // past its end (GEMM operands whose contiguous extent is not a multiple of 4); the bytes are never used
    size_t need = round_size(bytes + 16);
    void* p = nullptr;
    int tag = P.capture ? P.capture->id : 0;
    if (P.capture) p = take_from(P.capture->free_blocks, need);
    if (!p) p = take_from(P.free_blocks, need);
    size_t got = need;
    if (p) {
        got = P.all[p].bytes;
    } else {
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // out of memory: give cached blocks back and retry once
            if (!P.capture) (void)hipStreamSynchronize(rt().stream);
            trim_locked(P);
            e = hipMalloc(&p, need);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_error("lg_malloc: hipMalloc(%zu) failed: %s (reserved %llu B, in use %llu B)", need,
                          hipGetErrorString(e), (unsigned long long)P.reserved, (unsigned long long)P.in_use);
                return LG_ENOMEM;
            }
        }
        P.hip_mallocs++;
        P.reserved += need;
        P.all[p] = Block{need, 0};
    }
    P.all[p].graph = tag;
    P.live[p] = Block{got, tag};
    P.in_use += got;
    *ptr = p;
    return LG_OK;
}

int lg_free(void* ptr) {
    if (ptr == nullptr) return LG_OK;
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.live.find(ptr);
    LG_ARG(it != P.live.end(), "lg_free: %p was not allocated by lg_malloc (or freed twice)", ptr);
    Block b = it->second;
    P.live.erase(it);
    P.in_use -= b.bytes;
    if (b.graph != 0) {
        // memory a captured graph reads or writes stays pinned to that graph
        auto g = P.graphs.find(b.graph);
        if (g != P.graphs.end()) {
            g->second->free_blocks.emplace(b.bytes, ptr);
            return LG_OK;
        }
        P.all[ptr].graph = 0;   // graph already destroyed: back to the general pool
    }
    P.free_blocks.emplace(b.bytes, ptr);
    return LG_OK;
}

int lg_pool_trim(void) {
    LG_REQUIRE_INIT();
    LG_ARG(!capturing(), "lg_pool_trim: not allowed while capturing a graph");
    { const int rc = flush_queued(); if (rc != LG_OK) return rc; }
    LG_HIP(hipStreamSynchronize(rt().stream));
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    trim_locked(P);
    return LG_OK;
}

int lg_pool_stats(uint64_t* reserved_bytes, uint64_t* in_use_bytes, uint64_t* hip_malloc_calls) {
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    if (reserved_bytes) *reserved_bytes = P.reserved;
    if (in_use_bytes) *in_use_bytes = P.in_use;
    if (hip_malloc_calls) *hip_malloc_calls = P.hip_mallocs;
    return LG_OK;
}

// ---- transfers -----------------------------------------------------------------

int lg_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    LG_REQUIRE_INIT();
    if (bytes == 0) return LG_OK;
    LG_ARG(dst && src, "lg_memcpy_h2d: NULL pointer");
    LG_ARG(!capturing(), "lg_memcpy_h2d: host transfers cannot be captured; upload into a static tensor before lg_graph_launch");
    // pageable source: the runtime stages it, so the host buffer is reusable on return
    LG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, rt().stream));
    LG_HIP(hipStreamSynchronize(rt().stream));
    return LG_OK;
}

// Asynchronous upload through a ring of pinned staging buffers: the host data is copied into pinned memory
// (so `src` is reusable on return), the DMA to the device is stream-ordered and nobody waits for it.  This is how a
// training loop feeds a new batch into the static input tensor of a captured graph without stalling the stream.
namespace {
struct Staging {
    void*      host = nullptr;
    size_t     bytes = 0;
    hipEvent_t done = nullptr;
    bool       busy = false;
};
constexpr int kStagingSlots = 4;
Staging g_staging[kStagingSlots];
int g_staging_next = 0;
}  // namespace

int lg_memcpy_h2d_async(void* dst, const void* src, size_t bytes) {
    LG_REQUIRE_INIT();
    if (bytes == 0) return LG_OK;
    LG_ARG(dst && src, "lg_memcpy_h2d_async: NULL pointer");
    LG_ARG(!capturing(), "lg_memcpy_h2d_async: host transfers cannot be captured; upload between graph launches");
    Staging& st = g_staging[g_staging_next];
    g_staging_next = (g_staging_next + 1) % kStagingSlots;
    if (st.busy) {                       // the DMA that last used this slot must have finished
        LG_HIP(hipEventSynchronize(st.done));
        st.busy = false;
    }
    if (st.bytes < bytes) {
        if (st.host) LG_HIP(hipHostFree(st.host));
        st.host = nullptr;
        st.bytes = 0;
        size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
        LG_HIP(hipHostMalloc(&st.host, want, hipHostMallocDefault));
        st.bytes = want;
    }
    if (!st.done) LG_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    memcpy(st.host, src, bytes);
    LG_HIP(hipMemcpyAsync(dst, st.host, bytes, hipMemcpyHostToDevice, rt().stream));
    LG_HIP(hipEventRecord(st.done, rt().stream));
    st.busy = true;
    return LG_OK;
}

int lg_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    LG_REQUIRE_INIT();
    if (bytes == 0) return LG_OK;
    LG_ARG(dst && src, "lg_memcpy_d2h: NULL pointer");
    LG_ARG(!capturing(), "lg_memcpy_d2h: host transfers cannot be captured");
    { const int rc = flush_queued(); if (rc != LG_OK) return rc; }
    LG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, rt().stream));
    LG_HIP(hipStreamSynchronize(rt().stream));
    return check_device_status("lg_memcpy_d2h");
}

int lg_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    LG_REQUIRE_INIT();
    if (bytes == 0) return LG_OK;
    LG_ARG(dst && src, "lg_memcpy_d2d: NULL pointer");
    LG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt().stream));
    return LG_OK;
}

// ---- events ----------------------------------------------------------------------

int lg_event_create(void** ev) {
    LG_REQUIRE_INIT();
    LG_ARG(ev != nullptr, "lg_event_create: NULL");
    hipEvent_t e;
    LG_HIP(hipEventCreate(&e));
    *ev = e;
    return LG_OK;
}

int lg_event_record(void* ev) {
    LG_REQUIRE_INIT();
    LG_ARG(ev != nullptr, "lg_event_record: NULL");
    LG_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), rt().stream));
    return LG_OK;
}

int lg_event_elapsed_ms(void* start, void* stop, float* ms) {
    LG_REQUIRE_INIT();
    LG_ARG(start && stop && ms, "lg_event_elapsed_ms: NULL");
    LG_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    LG_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return LG_OK;
}

int lg_event_destroy(void* ev) {
    if (!ev) return LG_OK;
    LG_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return LG_OK;
}

// ---- graphs --------------------------------------------------------------------------

int lg_graph_begin(void) {
    LG_REQUIRE_INIT();
    { const int rc = flush_queued(); if (rc != LG_OK) return rc; }
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    LG_ARG(P.capture == nullptr, "lg_graph_begin: a capture is already in progress");
    // relaxed mode: hipMalloc from the pool is legal while capturing
    LG_HIP(hipStreamBeginCapture(rt().stream, hipStreamCaptureModeRelaxed));
    GraphRec* g = new GraphRec();
    g->id = P.next_graph_id++;
    P.graphs[g->id] = g;
    P.capture = g;
    return LG_OK;
}

int lg_graph_end(void** graph_exec) {
    LG_REQUIRE_INIT();
    LG_ARG(graph_exec != nullptr, "lg_graph_end: NULL");
    (void)flush_queued();                // queued launches belong to the capture that queued them
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    LG_ARG(P.capture != nullptr, "lg_graph_end: no capture in progress");
    GraphRec* g = P.capture;
    P.capture = nullptr;
    hipError_t e = hipStreamEndCapture(rt().stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("lg_graph_end: %s", hipGetErrorString(e));
        // release what the failed capture pinned
        for (auto& kv : g->free_blocks) { P.all[kv.second].graph = 0; P.free_blocks.emplace(kv.first, kv.second); }
        if (g->graph) (void)hipGraphDestroy(g->graph);
        P.graphs.erase(g->id);
        delete g;
        return LG_EHIP;
    }
    *graph_exec = g;
    return LG_OK;
}

int lg_graph_launch(void* graph_exec) {
    LG_REQUIRE_INIT();
    LG_ARG(graph_exec != nullptr, "lg_graph_launch: NULL");
    LG_ARG(!capturing(), "lg_graph_launch: not allowed while capturing");
    { const int rc = flush_queued(); if (rc != LG_OK) return rc; }
    GraphRec* g = static_cast<GraphRec*>(graph_exec);
    LG_HIP(hipGraphLaunch(g->exec, rt().stream));
    return LG_OK;
}

int lg_graph_kernel_count(void* graph_exec, int* kernels) {
    LG_REQUIRE_INIT();
    LG_ARG(graph_exec != nullptr && kernels != nullptr, "lg_graph_kernel_count: NULL");
    GraphRec* g = static_cast<GraphRec*>(graph_exec);
    size_t n = 0;
    LG_HIP(hipGraphGetNodes(g->graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) LG_HIP(hipGraphGetNodes(g->graph, nodes.data(), &n));
    int count = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        LG_HIP(hipGraphNodeGetType(nodes[i], &type));
        if (type == hipGraphNodeTypeKernel) ++count;
    }
    *kernels = count;
    return LG_OK;
}

int lg_graph_destroy(void* graph_exec) {
    if (!graph_exec) return LG_OK;
    LG_REQUIRE_INIT();
    GraphRec* g = static_cast<GraphRec*>(graph_exec);
    LG_HIP(hipStreamSynchronize(rt().stream));
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& kv : g->free_blocks) {
        P.all[kv.second].graph = 0;
        P.free_blocks.emplace(kv.first, kv.second);
    }
    // live blocks still tagged with this graph fall back to the general pool when freed (lg_free)
    P.graphs.erase(g->id);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return LG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// strided iteration descriptor
// ---------------------------------------------------------------------------
namespace lg {

bool build_iter(int ndim, const int64_t* shape, const int64_t* const* strides, int nops, IterDesc& out) {
    if (ndim < 0 || ndim > LG_MAX_DIMS || nops > kMaxOps) return false;
    int64_t shp[LG_MAX_DIMS];
    int64_t st[kMaxOps][LG_MAX_DIMS];
    int n = 0;
    int64_t numel = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] < 0) return false;
        numel *= shape[d];
        if (shape[d] == 1) continue;   // size-1 dims never move the pointer
        shp[n] = shape[d];
        for (int o = 0; o < nops; ++o) st[o][n] = strides[o] ? strides[o][d] : 0;
        ++n;
    }
    // merge dim i+1 into dim i when every operand walks them as one run
    int m = 0;
    for (int d = 0; d < n; ++d) {
        if (m > 0) {
            bool mergeable = true;
            for (int o = 0; o < nops; ++o)
                if (st[o][m - 1] != st[o][d] * shp[d]) { mergeable = false; break; }
            if (mergeable) {
                shp[m - 1] *= shp[d];
                for (int o = 0; o < nops; ++o) st[o][m - 1] = st[o][d];
                continue;
            }
        }
        shp[m] = shp[d];
        for (int o = 0; o < nops; ++o) st[o][m] = st[o][d];
        ++m;
    }
    if (m == 0) {   // scalar (or all dims of size 1)
        m = 1;
        shp[0] = 1;
        for (int o = 0; o < nops; ++o) st[o][0] = 0;
    }
    out.ndim = m;
    out.numel = numel;
    for (int d = 0; d < LG_MAX_DIMS; ++d) {
        out.shape[d] = d < m ? shp[d] : 1;
        for (int o = 0; o < kMaxOps; ++o) out.stride[o][d] = (d < m && o < nops) ? st[o][d] : 0;
    }
    return true;
}

}  // namespace lg
