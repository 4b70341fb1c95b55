// i8 MFMA issue-rate micro-benchmark for gfx950 (v_mfma_i32_32x32x32_i8, 1-4 independent accumulators, 1-4 waves per SIMD):
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate   -> 37.5 cycles@2.4GHz per instruction, 4.3 POPS
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
    v16i acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = r + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu) {
    int* d; hipMalloc(&d, 256 * 64 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * wgs_per_cu, iters = 500;
    k<NACC><<<blocks, 256>>>(d, 5);
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 4 * iters * 8 * NACC;  // mfma per all waves
    double per_simd = n / 1024;
    printf("NACC %d, %d waves/SIMD: %.3f ms, %.1f cycles@2.4GHz per MFMA per SIMD, %.0f TOPS\n", NACC, wgs_per_cu, ms, ms * 1e-3 * 2.4e9 / per_simd, n * 65536 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<2>(4); run<1>(4); return 0; }
