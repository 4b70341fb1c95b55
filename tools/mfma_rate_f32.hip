// fp32 / i8 MFMA issue-rate micro-benchmark for gfx950 (the shapes gauss7_mfma_kernel could use):
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_f32.hip -o /tmp/mfma_rate_f32 && /tmp/mfma_rate_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a = (float)threadIdx.x, b = (float)blockIdx.x + 1.f;
    v16f acc[NACC];
    v4f acc4[NACC];
    v16i acci[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) { acc[i][r] = r + i; acci[i][r] = r + i; if (r < 4) acc4[i][r] = r; }
    v4i ai = {(int)threadIdx.x, 2, 3, 4}, bi = {5, 6, 7, (int)blockIdx.x};
    v8s ab = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, bb = {1, 2, 3, 4, 5, 6, 7, (short)blockIdx.x};
    v8h ah, bh2;
    for (int r = 0; r < 8; r++) { ah[r] = (_Float16)(r + (int)threadIdx.x % 3); bh2[r] = (_Float16)(r + 1); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                if (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
                if (KIND == 2) acci[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, acci[i], 0, 0, 0);
                if (KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                if (KIND == 4) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh2, acc[i], 0, 0, 0);
            }
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r] + (float)acci[i][r] + (r < 4 ? acc4[i][r] : 0.f);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NACC> void run(const char* name, int wgs_per_cu, double flops_per) {
    float* d; hipMalloc(&d, 256 * 64 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * wgs_per_cu, iters = 300;
    k<KIND, NACC><<<blocks, 256>>>(d, 5);
    hipEventRecord(e0); k<KIND, NACC><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 4 * iters * 8 * NACC;
    printf("%-26s NACC %d, %d waves/SIMD: %7.3f ms, %6.1f cycles@2.4GHz per MFMA per SIMD, %7.1f TFLOP/s\n", name, NACC, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / (n / 1024), n * flops_per / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    run<0, 1>("f32_32x32x2_f32", 1, 4096); run<0, 2>("f32_32x32x2_f32", 1, 4096); run<0, 1>("f32_32x32x2_f32", 4, 4096); run<0, 2>("f32_32x32x2_f32", 4, 4096);
    run<1, 1>("f32_16x16x4_f32", 1, 2048); run<1, 4>("f32_16x16x4_f32", 1, 2048); run<1, 4>("f32_16x16x4_f32", 4, 2048);
    run<2, 1>("i32_32x32x32_i8", 1, 65536); run<2, 2>("i32_32x32x32_i8", 4, 65536);
    run<3, 1>("f32_32x32x16_bf16", 1, 32768); run<3, 2>("f32_32x32x16_bf16", 4, 32768);
    run<4, 1>("f32_32x32x16_f16", 1, 32768); run<4, 2>("f32_32x32x16_f16", 4, 32768);
    return 0;
}
