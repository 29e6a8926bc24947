// Does a hipGraph with parallel branches run its kernel nodes concurrently on gfx950 / ROCm 7.2, and what does a
// fork/join cost?  Two "latency-bound GEMM like" kernels (128 workgroups spinning ~20 us) captured serially and forked.
//   hipcc --offload-arch=gfx950 -O2 -o exp/fork_lab tools/fork_lab.hip && exp/fork_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(float* p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

static double replay_us(hipGraphExec_t exec, hipStream_t s, int n) {
    for (int i = 0; i < 5; ++i) hipGraphLaunch(exec, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipGraphLaunch(exec, s);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
    float *a, *b, *c;
    CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22)); CK(hipMalloc(&c, 1 << 22));
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, j1, j2;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
    for (int iters : {2000, 8000}) {
        for (int blocks : {128, 1024}) {
            hipGraph_t g; hipGraphExec_t e1, e2, e3, e4;
            // one kernel
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
            spin<<<blocks, 256, 0, s0>>>(a, iters);
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&e1, g, nullptr, nullptr, 0));
            // three kernels, serial
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
            spin<<<blocks, 256, 0, s0>>>(a, iters); spin<<<blocks, 256, 0, s0>>>(b, iters); spin<<<blocks, 256, 0, s0>>>(c, iters);
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&e2, g, nullptr, nullptr, 0));
            // three kernels, forked over three streams
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
            CK(hipEventRecord(fork, s0));
            CK(hipStreamWaitEvent(s1, fork, 0)); CK(hipStreamWaitEvent(s2, fork, 0));
            spin<<<blocks, 256, 0, s0>>>(a, iters); spin<<<blocks, 256, 0, s1>>>(b, iters); spin<<<blocks, 256, 0, s2>>>(c, iters);
            CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2));
            CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&e3, g, nullptr, nullptr, 0));
            // chain of 6: k, fork(3), k, fork(3) ... a step-like pattern: k - (k|k|k) - k - (k|k|k)
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
            for (int rep = 0; rep < 2; ++rep) {
                spin<<<blocks, 256, 0, s0>>>(a, iters);
                CK(hipEventRecord(fork, s0));
                CK(hipStreamWaitEvent(s1, fork, 0)); CK(hipStreamWaitEvent(s2, fork, 0));
                spin<<<blocks, 256, 0, s0>>>(a, iters); spin<<<blocks, 256, 0, s1>>>(b, iters); spin<<<blocks, 256, 0, s2>>>(c, iters);
                CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2));
                CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
            }
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&e4, g, nullptr, nullptr, 0));
            printf("iters=%d blocks=%d: one %.1f us | three serial %.1f us | three forked %.1f us | (k + 3 forked) x2 %.1f us\n", iters, blocks,
                   replay_us(e1, s0, 200), replay_us(e2, s0, 200), replay_us(e3, s0, 200), replay_us(e4, s0, 200));
            // eager multi-stream (no graph)
            for (int w = 0; w < 2; ++w) {
                hipStreamSynchronize(s0);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 200; ++i) {
                    hipEventRecord(fork, s0);
                    hipStreamWaitEvent(s1, fork, 0); hipStreamWaitEvent(s2, fork, 0);
                    spin<<<blocks, 256, 0, s0>>>(a, iters); spin<<<blocks, 256, 0, s1>>>(b, iters); spin<<<blocks, 256, 0, s2>>>(c, iters);
                    hipEventRecord(j1, s1); hipEventRecord(j2, s2);
                    hipStreamWaitEvent(s0, j1, 0); hipStreamWaitEvent(s0, j2, 0);
                }
                hipStreamSynchronize(s0);
                if (w) printf("    eager fork/join of three: %.1f us per round\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
            }
        }
    }
    return 0;
}
