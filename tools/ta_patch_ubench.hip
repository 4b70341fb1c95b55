// Vector-memory front-end cost of a patch gather (describe_kernel's access pattern), gfx950:
//   mode 0: 37 rows x 10 dwords, lane = dword of the patch (6 global_load_dword per patch)
//   mode 1: 37 rows x 4 x 16 bytes, lane = 16-byte granule (3 global_load_dwordx4 per patch, 16-byte aligned)
//   mode 2: 37 rows x 3 x 16 bytes (2 instructions; what a 48-byte window would need)
//   mode 3 / 4: mode 0 with non-temporal / sc1 loads (mode 4 waits after every load: a lower bound on its rate only)
//   mode 6: the image stored as 16 x 8 pixel tiles of 128 bytes (one cache line each): the 37 x 37 patch touches 3-4 x 5-6
//           tiles (18.6 on average) instead of 37 rows x 1.28 lines = 47; a wave loads two whole tiles per instruction
// Every workgroup works inside one 3 MB "image" chosen by blockIdx % 8, so that the lines come from the XCD's L2 as in the
// real kernel.  Prints ns per patch and patches/s.
//   hipcc --offload-arch=gfx950 -O3 tools/ta_patch_ubench.hip -o /tmp/ta_patch && /tmp/ta_patch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kPitch = 1280, kRows = 376 * 6, kImgBytes = kPitch * kRows;   // ~2.9 MB

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ base, int n_img, int per_wave, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* img = base + (size_t)(blockIdx.x % n_img) * kImgBytes;
    uint32_t acc = 0;
    for (int k = 0; k < per_wave; k++) {
        // modes 0..4: every wave its own random patch; mode 5 (= mode 0's loads): the four waves of a workgroup take
        // neighbouring patches (34 px apart in x, up to 12 rows in y), as keypoints sorted by image tile would be
        const uint32_t h = MODE == 5 ? hash(blockIdx.x * 977u + k) : hash((blockIdx.x * 4 + wave) * 977u + k);
        int x = 24 + (int)(h % (kPitch - 200)), y = 24 + (int)((h >> 12) % (kRows - 80));
        if (MODE == 5) { x += 34 * wave; y += 12 * (wave & 1); }
        const uint8_t* p = img + (size_t)(y - 18) * kPitch;
        if (MODE == 0 || MODE == 5) {
            const int px0 = (x - 18) & ~3;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int idx = lane + 64 * it, row = min(idx / 10, 36), col = idx % 10;
                acc ^= *reinterpret_cast<const uint32_t*>(p + row * kPitch + px0 + 4 * col);
            }
        } else if (MODE == 3) {   // mode 0 with non-temporal loads
            const int px0 = (x - 18) & ~3;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int idx = lane + 64 * it, row = min(idx / 10, 36), col = idx % 10;
                acc ^= __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p + row * kPitch + px0 + 4 * col));
            }
        } else if (MODE == 4) {   // mode 0 with sc1 loads (served by L2, no L1 allocation)
            const int px0 = (x - 18) & ~3;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int idx = lane + 64 * it, row = min(idx / 10, 36), col = idx % 10;
                uint32_t v;
                const uint32_t* q = reinterpret_cast<const uint32_t*>(p + row * kPitch + px0 + 4 * col);
                asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
                acc ^= v;
            }
        } else if (MODE == 6) {
            constexpr int kTilesPerRow = kPitch / 16;
            const int tx0 = (x - 18) >> 4, ty0 = (y - 18) >> 3;
            const int ntx = ((x + 18) >> 4) - tx0 + 1, nty = ((y + 18) >> 3) - ty0 + 1, nt = ntx * nty;   // <= 4 x 6
#pragma unroll
            for (int it = 0; it < 12; it++) {
                const int t = min(2 * it + (lane >> 5), nt - 1);
                const int tq = ntx == 4 ? t >> 2 : (t * 43) >> 7;   // t / ntx for ntx in {3, 4}, t < 24
                const int ty = ty0 + tq, tx = tx0 + t - tq * ntx;
                if (2 * it < nt)   // wave-uniform
                    acc ^= *reinterpret_cast<const uint32_t*>(img + ((size_t)(ty * kTilesPerRow + tx) << 7) + 4 * (lane & 31));
            }
        } else if (MODE == 7) {   // tiles again, 16 bytes per lane: eight whole tiles per instruction, 3 instructions
            constexpr int kTilesPerRow = kPitch / 16;
            const int tx0 = (x - 18) >> 4, ty0 = (y - 18) >> 3;
            const int ntx = ((x + 18) >> 4) - tx0 + 1, nty = ((y + 18) >> 3) - ty0 + 1, nt = ntx * nty;
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int t = min(8 * it + (lane >> 3), nt - 1);
                const int tq = ntx == 4 ? t >> 2 : (t * 43) >> 7;
                const int ty = ty0 + tq, tx = tx0 + t - tq * ntx;
                if (8 * it < nt) {
                    const uint4 v = *reinterpret_cast<const uint4*>(img + ((size_t)(ty * kTilesPerRow + tx) << 7) + 16 * (lane & 7));
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
            }
        } else if (MODE == 1) {
            const int px0 = (x - 18) & ~15;
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int idx = lane + 64 * it, row = min(idx >> 2, 36), col = idx & 3;
                const uint4 v = *reinterpret_cast<const uint4*>(p + row * kPitch + px0 + 16 * col);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        } else {
            const int px0 = (x - 18) & ~15;
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int idx = lane + 64 * it, row = min(idx / 3, 36), col = idx % 3;
                const uint4 v = *reinterpret_cast<const uint4*>(p + row * kPitch + px0 + 16 * col);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const int n_img = 8, blocks = 256 * 24, per_wave = 64;
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, (size_t)n_img * kImgBytes); hipMalloc(&o, 64);
    hipMemset(d, 1, (size_t)n_img * kImgBytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 8; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 3) hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 5) hipLaunchKernelGGL(gather<5>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 6) hipLaunchKernelGGL(gather<6>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            if (mode == 7) hipLaunchKernelGGL(gather<7>, dim3(blocks), dim3(256), 0, 0, d, n_img, per_wave, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double patches = (double)blocks * 4 * per_wave;
        printf("mode %d: %.3f ms for %.0f patches -> %.1f G patches/s, %.2f cycles@2.4GHz per patch per CU\n", mode, best, patches,
               patches / best / 1e6, best * 1e-3 * 2.4e9 * 256 / patches);
    }
    return 0;
}
