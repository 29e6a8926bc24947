// Probe: can two PROCESSES that share ONE MI355X exchange data through hipIpc memory handles from inside running kernels?
// ("Next round" 1: the device-side gradient exchange needs a test bed on the one-GPU boxes of this pool.)
//
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe.bin && tools/ipc_probe.bin
//
// The parent forks two rank processes BEFORE any HIP call (it never touches the GPU itself).  Each rank allocates a buffer,
// exports it with hipIpcGetMemHandle through a file, opens the peer's handle, and then runs
//   1. ping-pong: one wave per rank, system-scope flag stores into the PEER's buffer, bounded spin on its own flag;
//   2. bulk:      256 workgroups per rank push 1 MiB into the peer's buffer with write-through stores, signal per chunk, wait for
//                 the peer's chunk and check every word - for 200 epochs with changing values (a stale cache line fails).
// for three kinds of allocation (hipMalloc, fine-grained, uncached).  Every spin ends after 2 s of wall clock at the latest.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <unistd.h>
#include <sys/wait.h>
#include <sys/stat.h>
#include <dirent.h>
#include <time.h>

#define CK(expr)                                                                                             \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            printf("[rank %d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            fflush(stdout);                                                                                  \
            return 10;                                                                                       \
        }                                                                                                    \
    } while (0)

static int g_rank = -1;
static std::string g_dir;

constexpr int      kChunks = 256, kChunkFloats = 1024;            // 256 x 4 KiB = 1 MiB payload
constexpr int64_t  kSpinTicks = 200000000;                        // wall_clock64 runs at 100 MHz: 2 s
// layout of a rank's buffer (floats / ints): [payload 1 MiB][chunk flags kChunks ints, one per 128-B line][ping flag]
constexpr size_t   kFlagStride = 32;                              // ints: 128 bytes
constexpr size_t   kPayloadBytes = size_t(kChunks) * kChunkFloats * 4;
constexpr size_t   kBufBytes = kPayloadBytes + (kChunks + 2) * kFlagStride * 4;

__device__ __forceinline__ void st_sys(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int  ld_sys(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// returns false on timeout
__device__ __forceinline__ bool spin_until(const int* flag, int want) {
    const int64_t t0 = wall_clock64();
    while (ld_sys(flag) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > kSpinTicks) return false;
    }
    return true;
}

// result[0] = timed-out flag, result[1] = ticks for `iters` round trips
__global__ void pingpong(int* mine, int* peer, int rank, int iters, int64_t* result) {
    if (threadIdx.x != 0) return;
    int* my_flag = mine + kPayloadBytes / 4 + kChunks * kFlagStride;
    int* peer_flag = peer + kPayloadBytes / 4 + kChunks * kFlagStride;
    const int64_t t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (rank == 0) {
            st_sys(peer_flag, it);
            if (!spin_until(my_flag, it)) { result[0] = it; return; }
        } else {
            if (!spin_until(my_flag, it)) { result[0] = it; return; }
            st_sys(peer_flag, it);
        }
    }
    result[1] = wall_clock64() - t0;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t word_of(int epoch, int rank, int idx) { return uint32_t(epoch) * 2654435761u + uint32_t(rank) * 40503u + uint32_t(idx); }

// one epoch: every workgroup pushes its chunk into the peer's payload, signals, waits for the peer's chunk, checks it
// result[0] timeouts, result[2] mismatching words
__global__ void __launch_bounds__(256) bulk(int* mine, int* peer, int rank, int epoch, int64_t* result, int plain_loads) {
    const int w = blockIdx.x, tid = threadIdx.x;
    constexpr int SC0 = 1, SC1 = 16;
    const auto dst = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(peer) + size_t(w) * kChunkFloats * 4, 0, kChunkFloats * 4, 0x00020000);
    u32x4 v;
    for (int e = 0; e < 4; ++e) v[e] = word_of(epoch, rank, w * kChunkFloats + tid * 4 + e);
    __builtin_amdgcn_raw_buffer_store_b128(v, dst, tid * 16, 0, SC0 | SC1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (tid == 0) {
        st_sys(peer + kPayloadBytes / 4 + w * kFlagStride, epoch);
        ok = spin_until(mine + kPayloadBytes / 4 + w * kFlagStride, epoch) ? 1 : 0;
        if (!ok) atomicAdd(reinterpret_cast<unsigned long long*>(result), 1ULL);
    }
    __syncthreads();
    if (!ok) return;
    u32x4 got;
    if (plain_loads) {
        got = reinterpret_cast<const u32x4*>(mine)[w * (kChunkFloats / 4) + tid];
    } else {
        const auto src = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(mine) + size_t(w) * kChunkFloats * 4, 0, kChunkFloats * 4, 0x00020000);
        got = __builtin_amdgcn_raw_buffer_load_b128(src, tid * 16, 0, SC0 | SC1);
    }
    int bad = 0;
    for (int e = 0; e < 4; ++e) bad += got[e] != word_of(epoch, 1 - rank, w * kChunkFloats + tid * 4 + e);
    if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(result + 2), (unsigned long long)bad);
}

static bool write_file(const std::string& path, const void* data, size_t n) {
    std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    fwrite(data, 1, n, f);
    fclose(f);
    return rename(tmp.c_str(), path.c_str()) == 0;
}

static bool read_file(const std::string& path, void* data, size_t n, double timeout_s) {
    for (int i = 0; i < int(timeout_s * 1000); ++i) {
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            size_t got = fread(data, 1, n, f);
            fclose(f);
            if (got == n) return true;
        }
        usleep(1000);
    }
    return false;
}

static bool file_barrier(const char* tag) {
    char one = 1, other = 0;
    if (!write_file(g_dir + "/bar_" + tag + "_" + std::to_string(g_rank), &one, 1)) return false;
    return read_file(g_dir + "/bar_" + tag + "_" + std::to_string(1 - g_rank), &other, 1, 60.0);
}

static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int run_kind(int kind, const char* name) {
    void* mine = nullptr;
    hipError_t e = kind == 0 ? hipMalloc(&mine, kBufBytes)
                             : hipExtMallocWithFlags(&mine, kBufBytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("[rank %d] %s: allocation failed: %s\n", g_rank, name, hipGetErrorString(e)); (void)hipGetLastError(); return 1; }
    CK(hipMemset(mine, 0, kBufBytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hm, hp;
    e = hipIpcGetMemHandle(&hm, mine);
    if (e != hipSuccess) { printf("[rank %d] %s: hipIpcGetMemHandle failed: %s\n", g_rank, name, hipGetErrorString(e)); (void)hipGetLastError(); return 1; }
    std::string tag = std::string("h") + std::to_string(kind);
    if (!write_file(g_dir + "/" + tag + "_" + std::to_string(g_rank), &hm, sizeof(hm))) return 11;
    if (!read_file(g_dir + "/" + tag + "_" + std::to_string(1 - g_rank), &hp, sizeof(hp), 60.0)) { printf("[rank %d] %s: no handle from the peer\n", g_rank, name); return 12; }
    void* peer = nullptr;
    e = hipIpcOpenMemHandle(&peer, hp, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        printf("[rank %d] %s: hipIpcOpenMemHandle failed: %s\n", g_rank, name, hipGetErrorString(e));
        (void)hipGetLastError();
        file_barrier((tag + "x").c_str());
        (void)hipFree(mine);
        return 1;
    }
    int64_t* result = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&result), 64, hipHostMallocMapped));
    memset(result, 0, 64);
    if (!file_barrier((tag + "a").c_str())) return 13;

    // 1. ping-pong
    const int iters = 2000;
    double t0 = now_s();
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, static_cast<int*>(mine), static_cast<int*>(peer), g_rank, iters, result);
    CK(hipDeviceSynchronize());
    double wall = now_s() - t0;
    if (result[0]) printf("[rank %d] %-12s ping-pong TIMED OUT at iteration %lld (no concurrent progress of the two processes' kernels?) wall %.2f s\n", g_rank, name, (long long)result[0], wall);
    else printf("[rank %d] %-12s ping-pong: %d round trips, %.3f us each (in-kernel clock), wall %.3f s\n", g_rank, name, iters, result[1] / 100.0 / iters, wall);
    fflush(stdout);
    const bool pp_ok = result[0] == 0;
    if (!file_barrier((tag + "b").c_str())) return 14;

    // 2. bulk exchange, 200 epochs, sc0 sc1 loads then plain loads
    for (int plain = 0; plain < 2 && pp_ok; ++plain) {
        memset(result, 0, 64);
        CK(hipMemset(mine, 0, kBufBytes));
        CK(hipDeviceSynchronize());
        if (!file_barrier((tag + "c" + std::to_string(plain)).c_str())) return 15;
        const int epochs = 200;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int ep = 1; ep <= epochs; ++ep)
            hipLaunchKernelGGL(bulk, dim3(kChunks), dim3(256), 0, 0, static_cast<int*>(mine), static_cast<int*>(peer), g_rank, ep, result, plain);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("[rank %d] %-12s bulk 1 MiB each way x %d epochs (%s loads): %.2f us per epoch, chunk timeouts %lld, wrong words %lld\n", g_rank, name, epochs,
               plain ? "plain" : "sc0 sc1", 1e3 * ms / epochs, (long long)result[0], (long long)result[2]);
        fflush(stdout);
        if (!file_barrier((tag + "d" + std::to_string(plain)).c_str())) return 16;
    }
    CK(hipIpcCloseMemHandle(peer));
    if (!file_barrier((tag + "e").c_str())) return 17;
    CK(hipFree(mine));
    (void)hipHostFree(result);
    return 0;
}

static int rank_main(int rank) {
    g_rank = rank;
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    const char* want = getenv("IPC_PROBE_TWO_DEVICES");            // set on a multi-GPU node: rank r on device r
    const int dev = (want && atoi(want) && ndev > 1) ? rank : 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("[rank %d] pid %d on device %d of %d: %s (%s), HSA_ENABLE_IPC_MODE_LEGACY=%s\n", rank, getpid(), dev, ndev, prop.name, prop.gcnArchName,
           getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "(unset)");
    fflush(stdout);
    int rc = 0;
    const char* names[3] = {"hipMalloc", "fine-grained", "uncached"};
    for (int kind = 0; kind < 3; ++kind) {
        int r = run_kind(kind, names[kind]);
        if (r >= 10) return r;                                    // protocol failure: stop
        rc |= r;
    }
    return rc;
}

int main() {
    char tmpl[] = "/tmp/ipc_probe_XXXXXX";
    if (!mkdtemp(tmpl)) { perror("mkdtemp"); return 2; }
    g_dir = tmpl;
    pid_t pids[2];
    for (int r = 0; r < 2; ++r) {
        pids[r] = fork();                                         // before any HIP call: the children own the GPU, not this process
        if (pids[r] == 0) _exit(rank_main(r));
    }
    int rc = 0;
    for (int r = 0; r < 2; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        printf("rank %d exit code %d\n", r, code);
        if (code) rc = code;
    }
    if (DIR* d = opendir(g_dir.c_str())) {                        // no exec of `rm` from here
        while (dirent* ent = readdir(d))
            if (ent->d_name[0] != '.') unlink((g_dir + "/" + ent->d_name).c_str());
        closedir(d);
    }
    rmdir(g_dir.c_str());
    return rc;
}
