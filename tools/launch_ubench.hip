// What does a per-frame chain cost beyond its kernels?  A chain of N dependent kernels of ~D us each on one stream, issued (a) as N
// plain launches, (b) as one captured HIP graph, (c) as plain launches with a second-stream fork / join in the middle (event record +
// two stream waits, the blur's side stream), wall time per chain through the final hipStreamSynchronize; and the host time of
// the launch calls alone.  Decides whether graph replay / fewer launches can pay on this runtime.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_ubench.hip -o /tmp/launch_ubench && /tmp/launch_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long* sink, int iters, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7) {
    unsigned long long v = threadIdx.x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < iters; i++) v = v * 6364136223846793005ull + 1442695040888963407ull;
    if (v == 42) sink[0] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 12, reps = 400;
    unsigned long long* sink;
    CK(hipMalloc(&sink, 64));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    for (int iters : {0, 600, 1500}) {   // empty, ~3 us, ~8 us kernels
        auto chain = [&](hipStream_t st) { for (int k = 0; k < N; k++) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, st, sink, iters, k, 1, 2, 3, 4, 5, 6, 7); };
        // kernel duration alone
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        chain(s); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(t0, s)); for (int r = 0; r < 50; r++) chain(s); CK(hipEventRecord(t1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        const double per_kernel_back_to_back = ms * 1e3 / (50.0 * N);
        std::vector<double> plain, issue, graph, gissue, fork;
        for (int r = 0; r < reps; r++) {
            const double a = now_us(); chain(s); const double b = now_us(); CK(hipStreamSynchronize(s)); const double c = now_us();
            plain.push_back(c - a); issue.push_back(b - a);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(s); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; r++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        for (int r = 0; r < reps; r++) {
            const double a = now_us(); CK(hipGraphLaunch(ge, s)); const double b = now_us(); CK(hipStreamSynchronize(s)); const double c = now_us();
            graph.push_back(c - a); gissue.push_back(b - a);
        }
        for (int r = 0; r < reps; r++) {   // fork / join in the middle of the chain
            const double a = now_us();
            for (int k = 0; k < N; k++) {
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, sink, iters, k, 1, 2, 3, 4, 5, 6, 7);
                if (k == 1) { (void)hipEventRecord(e1, s); (void)hipStreamWaitEvent(s2, e1, 0); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s2, sink, iters, k, 1, 2, 3, 4, 5, 6, 7); (void)hipEventRecord(e2, s2); }
                if (k == N - 3) (void)hipStreamWaitEvent(s, e2, 0);
            }
            CK(hipStreamSynchronize(s));
            fork.push_back(now_us() - a);
        }
        std::printf("N=%d iters=%d: kernel back-to-back %.2f us each | plain chain %.1f us (issue %.1f) | graph %.1f us (issue %.1f) | plain + fork/join (one more kernel on a side stream) %.1f us\n",
                    N, iters, per_kernel_back_to_back, median(plain), median(issue), median(graph), median(gissue), median(fork));
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
