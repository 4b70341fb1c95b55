// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the extractor kernels use
// (MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of the bytes of a 16 B/lane stream; other widths are uncalibrated).
// Every kernel moves exactly kBytes (1 GiB, past the 256 MB Infinity Cache) in one pattern:
//   calib_read_dword   4 B per lane, coalesced (FAST staging, blur, pyramid rows)      calib_read_x4   16 B per lane
//   calib_write_dword  4 B per lane (pyramid / blur stores)                            calib_write_x4  16 B per lane
//   calib_write_8B     8 B records (candidate slots)
//   calib_read_x3_overlap  12 B per lane every 4.8 B (pyramid band kernels: the span read is again kBytes)
// Run under  rocprofv3 --pmc FETCH_SIZE ...  and  --pmc WRITE_SIZE ...  (tools/pmc_run.sh); ratio = counter KB * 1024 / 2^30.
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr size_t kBytes = 1ull << 30;
__global__ void calib_read_dword(const unsigned* __restrict__ p, unsigned* __restrict__ out, size_t n) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read_x4(const uint4* __restrict__ p, unsigned* __restrict__ out, size_t n) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write_dword(unsigned* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void calib_write_x4(uint4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{(unsigned)i, 1, 2, 3};
}
__global__ void calib_write_8B(uint2* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = uint2{(unsigned)i, 7};
}
// the pyramid band kernels' source read: 12 B per lane, lanes 4.8 B apart (windows overlap), rows of 1280 B
struct u3 { unsigned x, y, z; };
__global__ void calib_read_x3_overlap(const unsigned char* __restrict__ p, unsigned* __restrict__ out, size_t n_groups) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_groups; i += (size_t)gridDim.x * blockDim.x) {
        const u3 v = *reinterpret_cast<const u3*>(p + ((i * 24 / 5) & ~(size_t)3));
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    void *a = nullptr, *b = nullptr;
    unsigned* out = nullptr;
    if (hipMalloc(&a, kBytes) || hipMalloc(&b, kBytes) || hipMalloc((void**)&out, 64)) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, kBytes); hipMemset(b, 2, kBytes);
    const dim3 g(2048), t(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read_dword, g, t, 0, 0, (const unsigned*)a, out, kBytes / 4);
        hipLaunchKernelGGL(calib_read_x4, g, t, 0, 0, (const uint4*)b, out, kBytes / 16);
        hipLaunchKernelGGL(calib_write_dword, g, t, 0, 0, (unsigned*)a, kBytes / 4);
        hipLaunchKernelGGL(calib_write_x4, g, t, 0, 0, (uint4*)b, kBytes / 16);
        hipLaunchKernelGGL(calib_write_8B, g, t, 0, 0, (uint2*)a, kBytes / 8);
        hipLaunchKernelGGL(calib_read_x3_overlap, g, t, 0, 0, (const unsigned char*)b, out, (kBytes - 64) * 5 / 24);
    }
    hipDeviceSynchronize();
    printf("moved %zu bytes per kernel\n", kBytes);
    return 0;
}
