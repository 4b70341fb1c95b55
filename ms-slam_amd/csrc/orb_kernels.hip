// Hand-written gfx950 (CDNA4, wave64) kernels of the ORB extractor.  Integer / bit work only — no MFMA.
//
//   pyr_resize_kernel   ComputePyramid            ORBextractor.cc:1170-1195  (cv::resize INTER_LINEAR, 8-bit)
//   fast_cells_kernel   cell loop + cv::FAST      ORBextractor.cc:805-872    (FAST-9/16 score, 3x3 NMS, th fallback)
//   cand_* kernels      vToDistributeKeys order   ORBextractor.cc:863-867    (row-major cell / scan order compaction)
//   gauss7_kernel       GaussianBlur 7x7 s=2      ORBextractor.cc:1132-1133  (Q8.8 separable, reflect-101; LDS-free strips)
//   describe_kernel     IC_Angle + rBRIEF         ORBextractor.cc:76-146,894-895,1138
//
// Arithmetic follows SURVEY.md Appendix A / oracle/cvprims.h exactly (bit-exact contract).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "orb_device.h"
#include "sincosf_restated.h"

namespace msorb {

// ------------------------------------------------------------------------------------------------
// Pyramid level l from level l-1.  One thread = 4 horizontally adjacent destination pixels of one
// row, written as one aligned 32-bit store (pitch is a multiple of 64).  block = 64 x 4.
// ------------------------------------------------------------------------------------------------
// single_stage: the generic FixedPtCast rounding (S0*b0 + S1*b1 + (1 << 21)) >> 22 instead of VResizeLinear<uchar>'s two
// stages (Semantics::resize_single_stage; only this kernel serves the variant).
__global__ __launch_bounds__(256) void pyr_resize_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                         const ResizeTap* __restrict__ tx,
                                                         const ResizeTap* __restrict__ ty, int single_stage) {
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= dst.h || dx0 >= dst.w) return;
    const ResizeTap vy = ty[dy];
    const uint8_t* s0 = src.base + (size_t)img * src.img_stride + (size_t)vy.i0 * src.pitch;
    const uint8_t* s1 = src.base + (size_t)img * src.img_stride + (size_t)vy.i1 * src.pitch;
    const int b0 = vy.c0, b1 = vy.c1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        if (dx < dst.w) {
            const ResizeTap vx = tx[dx];
            const int h0 = s0[vx.i0] * vx.c0 + s0[vx.i1] * vx.c1;
            const int h1 = s1[vx.i0] * vx.c0 + s1[vx.i1] * vx.c1;
            const int v = single_stage ? (h0 * b0 + h1 * b1 + (1 << 21)) >> 22
                                       : (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)dy * dst.pitch + dx0;
    *reinterpret_cast<uint32_t*>(d) = packed;
}

// Aligned streaming variant: the six source pixels a 4-pixel group can touch ([sx(dx0), sx(dx0+3)+1]) lie inside
// three aligned dwords per source row, so a thread issues 6 coalesced dword loads + 2 x 16-byte tap loads instead
// of 16 byte loads; taps pick their two neighbouring bytes with v_alignbyte on a selected dword pair.
__device__ __forceinline__ uint32_t pick2(uint32_t w0, uint32_t w1, uint32_t w2, int o) {
    // bytes o, o+1 of the 12-byte window {w2,w1,w0} in the low 16 bits (o in [0, 10])
    const uint32_t lo = o < 4 ? w0 : (o < 8 ? w1 : w2);
    const uint32_t hi = o < 4 ? w1 : (o < 8 ? w2 : 0u);
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(o & 3));
}
__global__ __launch_bounds__(256) void pyr_resize_aligned_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                                 const ResizeTap* __restrict__ tx,
                                                                 const ResizeTap* __restrict__ ty) {
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= dst.h || dx0 >= dst.w) return;
    const ResizeTap vy = ty[dy];
    // the 4 x-taps of this group: 32 contiguous bytes (the table is padded to a multiple of 4 entries)
    const uint4 ta = reinterpret_cast<const uint4*>(tx + dx0)[0];
    const uint4 tb = reinterpret_cast<const uint4*>(tx + dx0)[1];
    const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};  // per tap: {i0|i1<<16, c0|c1<<16}
    const int base = (int)(tw[0] & 0xffffu) & ~3;  // aligned column of the first source pixel
    const uint8_t* s0 = src.base + (size_t)img * src.img_stride + (size_t)vy.i0 * src.pitch + base;
    const uint8_t* s1 = src.base + (size_t)img * src.img_stride + (size_t)vy.i1 * src.pitch + base;
    const uint32_t a0 = reinterpret_cast<const uint32_t*>(s0)[0], a1 = reinterpret_cast<const uint32_t*>(s0)[1],
                   a2 = reinterpret_cast<const uint32_t*>(s0)[2];
    const uint32_t b0w = reinterpret_cast<const uint32_t*>(s1)[0], b1w = reinterpret_cast<const uint32_t*>(s1)[1],
                   b2w = reinterpret_cast<const uint32_t*>(s1)[2];
    const int b0 = vy.c0, b1 = vy.c1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int i0 = (int)(tw[2 * i] & 0xffffu), i1 = (int)(tw[2 * i] >> 16);
        const int c0 = (int)(int16_t)(tw[2 * i + 1] & 0xffffu), c1 = (int)(int16_t)(tw[2 * i + 1] >> 16);
        const int o = i0 - base;
        const uint32_t pa = pick2(a0, a1, a2, o), pb = pick2(b0w, b1w, b2w, o);
        // second tap = next pixel, except at the right edge where i1 == i0 (and c1 == 0)
        const int sh = (i1 != i0) ? 8 : 0;
        const int h0 = (int)(pa & 255u) * c0 + (int)((pa >> sh) & 255u) * c1;
        const int h1 = (int)(pb & 255u) * c0 + (int)((pb >> sh) & 255u) * c1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 255) << (8 * i);
    }
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)dy * dst.pitch + dx0;
    *reinterpret_cast<uint32_t*>(d) = packed;
}

// Row-streaming variant for batches: a wave produces R consecutive output rows of its 64 column groups.  The x taps
// are decoded once per thread instead of once per output dword, and the horizontally interpolated source rows are
// kept in registers: at scale 1.2 consecutive output rows share one of their two source rows (i0(dy+1) == i1(dy) five
// times out of six), so each source row is fetched and interpolated once, not twice.  The raw dwords of the next
// source row are prefetched while the current output row is blended.  Same integer arithmetic as the kernels above.
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
template <int R>
__global__ __launch_bounds__(256) void pyr_resize_rows_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                              const ResizeTap* __restrict__ tx,
                                                              const ResizeTap* __restrict__ ty) {
    const int img = blockIdx.z;
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)) * R);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy0 >= dst.h || dx0 >= dst.w) return;
    const int dy_end = min(dy0 + R, dst.h);
    const uint4 ta = reinterpret_cast<const uint4*>(tx + dx0)[0];
    const uint4 tb = reinterpret_cast<const uint4*>(tx + dx0)[1];
    const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};  // per tap: {i0|i1<<16, c0|c1<<16}
    const int base = (int)(tw[0] & 0xffffu) & ~3;  // aligned column of the first source pixel
    // per tap: two byte-permute selectors that fetch the tap's two source pixels out of the 12-byte window {w2,w1,w0}
    // into the low bytes of the two 16-bit halves (0x0c = constant zero), and the weights as a u16 pair: the horizontal
    // interpolation is then perm, perm, or, dot2.  At the right edge i1 == i0 and c1 == 0: whatever the second selector
    // picks is multiplied by zero.
    uint32_t sel01[4], sel2[4], cw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t o = (tw[2 * i] & 0xffffu) - (uint32_t)base;  // 0..10
        const uint32_t a1 = o < 8 ? o : 0x0cu, b1 = o + 1 < 8 ? o + 1 : 0x0cu;
        const uint32_t a2 = o >= 8 ? o - 8 : 0x0cu, b2 = o + 1 >= 8 ? o + 1 - 8 : 0x0cu;
        sel01[i] = a1 | (0x0cu << 8) | (b1 << 16) | (0x0cu << 24);
        sel2[i] = a2 | (0x0cu << 8) | (b2 << 16) | (0x0cu << 24);
        cw[i] = tw[2 * i + 1];  // c0 | c1 << 16, both in [0, 2048]
    }
    const uint8_t* sb = src.base + (size_t)img * src.img_stride;
    struct Raw { uint32_t w0, w1, w2; };
    auto fetch = [&](int sy) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(sb + (size_t)sy * src.pitch + (size_t)(uint32_t)base);
        return Raw{q[0], q[1], q[2]};
    };
    auto hrow = [&](const Raw& r, int H[4]) {  // (src[i0]*c0 + src[i1]*c1) >> 4 for the 4 columns of this thread
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t px = __builtin_amdgcn_perm(r.w1, r.w0, sel01[i]) | __builtin_amdgcn_perm(0u, r.w2, sel2[i]);
            H[i] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, px), __builtin_bit_cast(ushort2v, cw[i]), 0u, false) >> 4);
        }
    };
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)(uint32_t)dx0;
    const int last_src = src.h - 1;
    int HA[4], HB[4];   // interpolated source rows ia (upper) and ib (lower) of the current output row
    int ia = -1, ib = -1;
    Raw pre = fetch(ty[dy0].i0);
    int ipre = ty[dy0].i0;  // source row held raw in `pre`
    for (int dy = dy0; dy < dy_end; dy++) {
        const ResizeTap vy = ty[dy];  // wave-uniform: scalar loads
        const int n0 = vy.i0, n1 = vy.i1;
        if (n0 == ib) {               // common case: the lower row of the previous output row becomes the upper one
#pragma unroll
            for (int i = 0; i < 4; i++) HA[i] = HB[i];
        } else if (n0 != ia) {
            const Raw r = (n0 == ipre) ? pre : fetch(n0);
            hrow(r, HA);
        }
        ia = n0;
        if (n1 == n0) {               // bottom clamp: both taps on one row
#pragma unroll
            for (int i = 0; i < 4; i++) HB[i] = HA[i];
        } else {
            const Raw r = (n1 == ipre) ? pre : fetch(n1);
            hrow(r, HB);
        }
        ib = n1;
        if (dy + 1 < dy_end) {        // the next output row needs ib (held) and, almost always, ib + 1
            ipre = min(ib + 1, last_src);
            pre = fetch(ipre);
        }
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = (((vy.c0 * HA[i]) >> 16) + ((vy.c1 * HB[i]) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(d + (size_t)dy * dst.pitch) = packed;
    }
}

// SDWA forms the compiler does not pick by itself: a 24-bit multiply by one u16 half of a register, and the sum of two
// registers' high halves.  Only source selects are used (a partial destination write would need wait states).
__device__ __forceinline__ uint32_t sdwa_mul_lo(uint32_t b, uint32_t h) {
    uint32_t r;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(b), "v"(h));
    return r;
}
__device__ __forceinline__ uint32_t sdwa_mul_hi(uint32_t b, uint32_t h) {
    uint32_t r;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(b), "v"(h));
    return r;
}
__device__ __forceinline__ uint32_t sdwa_hi_sum(uint32_t x, uint32_t y) {   // (x >> 16) + (y >> 16)
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Row-band variant (batches, scale factors up to 1.25): a wave produces R consecutive output rows of its 64 column groups in
// two phases without any data-dependent control flow in between:
//   A  all source rows the band touches (rows i0(dy0) .. i1(dy0 + R - 1): <= kSrc of them) are fetched with every load in
//      flight at once, interpolated horizontally (v_perm + v_dot2_u32_u16 per pixel) and parked as 4 x u16 per lane
//      in LDS — used only as storage a lane can index at run time: a lane reads back exactly what it wrote, so no
//      barrier and no sharing;
//   B  every output row reads its two parked rows by (wave-uniform) index and blends them with SDWA half-word operands.
// The y taps of the band come in up front with the first loads.  Compared with pyr_resize_rows_kernel: no per-row wait for a
// tap record, no register shuffling between "upper" and "lower" rows, 14 instead of 25 VALU lane-operations per pixel.
template <int R, int kSrc>
__device__ __forceinline__ void pyr_band_tile(const LevelView& src, const LevelView& dst, uint8_t* __restrict__ dst_base,
                                              const ResizeTap* __restrict__ tx, const ResizeTap* __restrict__ ty, const int img,
                                              const int bx, const int dy0, const int lane, uint2 (*park_w)[64]) {
    const int dx0 = (bx * 64 + lane) * 4;
    if (dy0 >= dst.h) return;   // wave-uniform
    const int n_out = min(R, dst.h - dy0);
    // y taps of the band (wave-uniform addresses: scalar loads)
    uint32_t ti[R], tc[R];      // i0 | i1 << 16, c0 | c1 << 16
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint2 v = reinterpret_cast<const uint2*>(ty)[__builtin_amdgcn_readfirstlane(dy0 + min(k, n_out - 1))];
        ti[k] = __builtin_amdgcn_readfirstlane(v.x);
        tc[k] = __builtin_amdgcn_readfirstlane(v.y);
    }
    const int s_lo = (int)(ti[0] & 0xffffu);
    int s_hi = s_lo;
#pragma unroll
    for (int k = 0; k < R; k++) s_hi = max(s_hi, (int)(ti[k] >> 16));
    const int n_src = s_hi - s_lo + 1;   // <= kSrc (checked on the host)
    const bool active = dx0 < dst.w;
    // x taps of this column group -> byte selectors and weights (as pyr_resize_rows_kernel)
    const int dxc = active ? dx0 : 0;
    const uint4 ta = reinterpret_cast<const uint4*>(tx + dxc)[0];
    const uint4 tb = reinterpret_cast<const uint4*>(tx + dxc)[1];
    const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};  // per tap: {i0|i1<<16, c0|c1<<16}
    const int base = (int)(tw[0] & 0xffffu) & ~3;  // aligned column of the first source pixel
    // The 4 taps of a lane start within 5 source pixels of the first one (scale <= 1.25): two v_alignbyte bring the 12-byte
    // window to "first tap at byte 0", then one v_perm per tap puts its two pixels into the u16 halves for v_dot2_u32_u16.
    const uint32_t o0 = (tw[0] & 0xffffu) - (uint32_t)base;   // 0..3
    uint32_t sel[4], cw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t r = (tw[2 * i] & 0xffffu) - (uint32_t)base - o0;  // 0..5; at the right edge i1 == i0 and c1 == 0
        sel[i] = r | (0x0cu << 8) | ((r + 1) << 16) | (0x0cu << 24);
        cw[i] = tw[2 * i + 1];  // c0 | c1 << 16, both in [0, 2048]
    }
    const uint8_t* sb = src.base + (size_t)img * src.img_stride + (size_t)(uint32_t)base;
    // phase A
    uint32_t raw[kSrc][3];
#pragma unroll
    for (int r = 0; r < kSrc; r++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(sb + (size_t)min(s_lo + r, s_hi) * src.pitch);
        raw[r][0] = q[0]; raw[r][1] = q[1]; raw[r][2] = q[2];
    }
#pragma unroll
    for (int r = 0; r < kSrc; r++) {
        if (r < n_src) {   // wave-uniform
            const uint32_t lo = __builtin_amdgcn_alignbyte(raw[r][1], raw[r][0], o0);
            const uint32_t hi = __builtin_amdgcn_alignbyte(raw[r][2], raw[r][1], o0);
            uint32_t H[4];   // (src[i0]*c0 + src[i1]*c1) >> 4 <= 32640
#pragma unroll
            for (int i = 0; i < 4; i++)
                H[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(hi, lo, sel[i])),
                                              __builtin_bit_cast(ushort2v, cw[i]), 0u, false) >> 4;
            park_w[r][lane] = uint2{H[0] | (H[1] << 16), H[2] | (H[3] << 16)};
        }
    }
    // phase B: ((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2) >> 2 with the u16 halves picked by SDWA operand selects
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)(uint32_t)dx0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        if (k < n_out) {   // wave-uniform
            const int n0 = (int)(ti[k] & 0xffffu) - s_lo, n1 = (int)(ti[k] >> 16) - s_lo;
            const uint32_t b0 = tc[k] & 0xffffu, b1 = tc[k] >> 16;
            const uint2 A = park_w[n0][lane], B = park_w[n1][lane];
            const uint32_t t0 = sdwa_hi_sum(sdwa_mul_lo(b0, A.x), sdwa_mul_lo(b1, B.x));
            const uint32_t t1 = sdwa_hi_sum(sdwa_mul_hi(b0, A.x), sdwa_mul_hi(b1, B.x));
            const uint32_t t2 = sdwa_hi_sum(sdwa_mul_lo(b0, A.y), sdwa_mul_lo(b1, B.y));
            const uint32_t t3 = sdwa_hi_sum(sdwa_mul_hi(b0, A.y), sdwa_mul_hi(b1, B.y));
            const ushort2v two = {2, 2};
            const ushort2v p01 = (__builtin_bit_cast(ushort2v, t0 | (t1 << 16)) + two) >> 2;
            const ushort2v p23 = (__builtin_bit_cast(ushort2v, t2 | (t3 << 16)) + two) >> 2;
            const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
            if (active) *reinterpret_cast<uint32_t*>(d + (size_t)(dy0 + k) * dst.pitch) = packed;
        }
    }
}

template <int R, int kSrc>
__device__ __forceinline__ void pyr_band_tile_reg(const LevelView& src, const LevelView& dst, uint8_t* __restrict__ dst_base,
                                              const ResizeTap* __restrict__ tx, const ResizeTap* __restrict__ ty, const int img,
                                              const int bx, const int dy0, const int lane) {
    const int dx0 = (bx * 64 + lane) * 4;
    if (dy0 >= dst.h) return;   // wave-uniform
    const int n_out = min(R, dst.h - dy0);
    // y taps of the band (wave-uniform addresses: scalar loads)
    uint32_t ti[R], tc[R];      // i0 | i1 << 16, c0 | c1 << 16
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint2 v = reinterpret_cast<const uint2*>(ty)[__builtin_amdgcn_readfirstlane(dy0 + min(k, n_out - 1))];
        ti[k] = __builtin_amdgcn_readfirstlane(v.x);
        tc[k] = __builtin_amdgcn_readfirstlane(v.y);
    }
    const int s_lo = (int)(ti[0] & 0xffffu);
    int s_hi = s_lo;
#pragma unroll
    for (int k = 0; k < R; k++) s_hi = max(s_hi, (int)(ti[k] >> 16));
    const int n_src = s_hi - s_lo + 1;   // <= kSrc (checked on the host)
    const bool active = dx0 < dst.w;
    // x taps of this column group -> byte selectors and weights (as pyr_resize_rows_kernel)
    const int dxc = active ? dx0 : 0;
    const uint4 ta = reinterpret_cast<const uint4*>(tx + dxc)[0];
    const uint4 tb = reinterpret_cast<const uint4*>(tx + dxc)[1];
    const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};  // per tap: {i0|i1<<16, c0|c1<<16}
    const int base = (int)(tw[0] & 0xffffu) & ~3;  // aligned column of the first source pixel
    // The 4 taps of a lane start within 5 source pixels of the first one (scale <= 1.25): two v_alignbyte bring the 12-byte
    // window to "first tap at byte 0", then one v_perm per tap puts its two pixels into the u16 halves for v_dot2_u32_u16.
    const uint32_t o0 = (tw[0] & 0xffffu) - (uint32_t)base;   // 0..3
    uint32_t sel[4], cw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t r = (tw[2 * i] & 0xffffu) - (uint32_t)base - o0;  // 0..5; at the right edge i1 == i0 and c1 == 0
        sel[i] = r | (0x0cu << 8) | ((r + 1) << 16) | (0x0cu << 24);
        cw[i] = tw[2 * i + 1];  // c0 | c1 << 16, both in [0, 2048]
    }
    const uint8_t* sb = src.base + (size_t)img * src.img_stride + (size_t)(uint32_t)base;
    // phase A (the parked rows live in registers: phase B picks them with wave-uniform indices, which the compiler turns into
    // s_set_gpr_idx + v_mov — one VALU move per operand instead of an LDS round trip, and the kernel needs no LDS at all)
    uint32_t park_x[kSrc], park_y[kSrc];
    uint32_t raw[kSrc][3];
#pragma unroll
    for (int r = 0; r < kSrc; r++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(sb + (size_t)min(s_lo + r, s_hi) * src.pitch);
        raw[r][0] = q[0]; raw[r][1] = q[1]; raw[r][2] = q[2];
    }
#pragma unroll
    for (int r = 0; r < kSrc; r++) {
        if (r < n_src) {   // wave-uniform
            const uint32_t lo = __builtin_amdgcn_alignbyte(raw[r][1], raw[r][0], o0);
            const uint32_t hi = __builtin_amdgcn_alignbyte(raw[r][2], raw[r][1], o0);
            uint32_t H[4];   // (src[i0]*c0 + src[i1]*c1) >> 4 <= 32640
#pragma unroll
            for (int i = 0; i < 4; i++)
                H[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(hi, lo, sel[i])),
                                              __builtin_bit_cast(ushort2v, cw[i]), 0u, false) >> 4;
            park_x[r] = H[0] | (H[1] << 16);
            park_y[r] = H[2] | (H[3] << 16);
        } else {
            park_x[r] = 0; park_y[r] = 0;
        }
    }
    // phase B: ((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2) >> 2 with the u16 halves picked by SDWA operand selects
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)(uint32_t)dx0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        if (k < n_out) {   // wave-uniform
            const int n0 = (int)(ti[k] & 0xffffu) - s_lo, n1 = (int)(ti[k] >> 16) - s_lo;
            const uint32_t b0 = tc[k] & 0xffffu, b1 = tc[k] >> 16;
            const uint2 A = uint2{park_x[n0], park_y[n0]}, B = uint2{park_x[n1], park_y[n1]};
            const uint32_t t0 = sdwa_hi_sum(sdwa_mul_lo(b0, A.x), sdwa_mul_lo(b1, B.x));
            const uint32_t t1 = sdwa_hi_sum(sdwa_mul_hi(b0, A.x), sdwa_mul_hi(b1, B.x));
            const uint32_t t2 = sdwa_hi_sum(sdwa_mul_lo(b0, A.y), sdwa_mul_lo(b1, B.y));
            const uint32_t t3 = sdwa_hi_sum(sdwa_mul_hi(b0, A.y), sdwa_mul_hi(b1, B.y));
            const ushort2v two = {2, 2};
            const ushort2v p01 = (__builtin_bit_cast(ushort2v, t0 | (t1 << 16)) + two) >> 2;
            const ushort2v p23 = (__builtin_bit_cast(ushort2v, t2 | (t3 << 16)) + two) >> 2;
            const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
            if (active) *reinterpret_cast<uint32_t*>(d + (size_t)(dy0 + k) * dst.pitch) = packed;
        }
    }
}

// LDS-free form of the band kernel (same arithmetic, the parked rows in registers): beside the other batch's FAST / quadtree /
// describe workgroups, which keep a CU's LDS filled to within a few KB, a workgroup that asks for no LDS starts as soon as
// two wave slots are free — the LDS form's levels were stretched from 0.05 to 0.8 ms there (profiles/round2_timeline_pipelined.txt).
template <int R, int kSrc, int WAVES = 2>
__global__ __launch_bounds__(64 * WAVES) void pyr_resize_bandreg_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                                        const ResizeTap* __restrict__ tx,
                                                                        const ResizeTap* __restrict__ ty) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * WAVES + wave) * R);
    pyr_band_tile_reg<R, kSrc>(src, dst, dst_base, tx, ty, blockIdx.z, blockIdx.x, dy0, lane);
}

template <int R, int kSrc, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) void pyr_resize_band_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                                     const ResizeTap* __restrict__ tx,
                                                                     const ResizeTap* __restrict__ ty) {
    __shared__ uint2 park[WAVES][kSrc][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * WAVES + wave) * R);
    pyr_band_tile<R, kSrc>(src, dst, dst_base, tx, ty, blockIdx.z, blockIdx.x, dy0, lane, park[wave]);
}

// The small top levels in ONE launch: a 16-wave workgroup per image walks down the levels, its waves taking the (band, column
// chunk) tiles of a level in turn; a workgroup-wide barrier separates the levels.  As separate launches each of these levels costs 13-16 us
// for 5-9 us of work — ramp-up, tap fetch, load latency and drain of a dependent launch do not shrink with the level.
struct PyrTailArgs {
    LevelView lv[4];            // lv[0] = source of the first tail level, lv[i + 1] = i-th tail level
    const ResizeTap* tx[3];
    const ResizeTap* ty[3];
    int nl;                     // tail levels (<= 3)
};
template <int R, int kSrc>
__global__ __launch_bounds__(1024) void pyr_resize_tail_kernel(PyrTailArgs A) {
    __shared__ uint2 park[16][kSrc][64];
    const int img = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    for (int li = 0; li < A.nl; li++) {
        const LevelView src = A.lv[li], dst = A.lv[li + 1];
        const int gx = (dst.w + 255) / 256, nb = (dst.h + R - 1) / R;
        for (int t = wave; t < gx * nb; t += 16) {
            const int band = t / gx, bx = t - band * gx;
            pyr_band_tile<R, kSrc>(src, dst, const_cast<uint8_t*>(dst.base), A.tx[li], A.ty[li], img, bx, band * R, lane, park[wave]);
        }
        __syncthreads();   // workgroup-scope release / acquire: all waves of a workgroup share the CU's L1 (write-through), and
                           // the level just written was never read before (an agent-scope fence here writes back / invalidates
                           // the whole L2 per workgroup and level: measured 0.34 ms instead of 0.02)
    }
}

// LDS-DMA form of the band kernel.  What limits pyr_resize_band_kernel is neither HBM nor VALU but the vector-memory
// front end: a lane there asks for 12 bytes every 4.8 bytes, i.e. 768 requested bytes per 307 new ones, and the texture
// addresser retires roughly 16 requested bytes per cycle and CU (tools/fetch_calib.hip: 3.3 TB/s for exactly this pattern
// against 6.6 TB/s for 16 B/lane).  Here every source byte is requested once: the band's source rows come in as 16-byte
// granules, 21 per row (336 B) and three rows per global_load_lds_dwordx4, straight into the wave's LDS slab; the lanes then
// pick their 12-byte windows out of LDS.  From there on it is the band kernel (phase A -> parked u16 rows -> phase B).
// Needs 16-byte aligned rows (base, pitch, image stride) and a horizontal scale <= 1.22 (the 64 windows of a wave must fit
// 336 bytes); launch_pyr_resize falls back to the band kernel otherwise.
template <int R>
__global__ __launch_bounds__(256) void pyr_resize_dma_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                             const ResizeTap* __restrict__ tx,
                                                             const ResizeTap* __restrict__ ty) {
    constexpr int kRowB = 336, kGran = kRowB / 16, kSrc = 12;   // 3 rows x 21 granules = 63 lanes per LDS-DMA instruction
    __shared__ __attribute__((aligned(16))) uint8_t slab[4][kSrc * kRowB];
    __shared__ uint2 park[4][kSrc][64];
    const int img = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + wave) * R);
    const int dx0 = (blockIdx.x * 64 + lane) * 4;
    if (dy0 >= dst.h) return;   // wave-uniform
    const int n_out = min(R, dst.h - dy0);
    uint32_t ti[R], tc[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint2 v = reinterpret_cast<const uint2*>(ty)[__builtin_amdgcn_readfirstlane(dy0 + min(k, n_out - 1))];
        ti[k] = __builtin_amdgcn_readfirstlane(v.x);
        tc[k] = __builtin_amdgcn_readfirstlane(v.y);
    }
    const int s_lo = (int)(ti[0] & 0xffffu);
    int s_hi = s_lo;
#pragma unroll
    for (int k = 0; k < R; k++) s_hi = max(s_hi, (int)(ti[k] >> 16));
    const int n_src = s_hi - s_lo + 1;   // <= kSrc (checked on the host)
    const bool active = dx0 < dst.w;
    const int dxc = active ? dx0 : 0;
    const uint4 ta = reinterpret_cast<const uint4*>(tx + dxc)[0];
    const uint4 tb = reinterpret_cast<const uint4*>(tx + dxc)[1];
    const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
    const int base = (int)(tw[0] & 0xffffu) & ~3;
    const int xg = __builtin_amdgcn_readfirstlane(base) & ~15;   // lane 0 (always active): first granule of the chunk
    const uint32_t o0 = (tw[0] & 0xffffu) - (uint32_t)base;
    uint32_t sel[4], cw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t r = (tw[2 * i] & 0xffffu) - (uint32_t)base - o0;
        sel[i] = r | (0x0cu << 8) | ((r + 1) << 16) | (0x0cu << 24);
        cw[i] = tw[2 * i + 1];
    }
    // stage: lane -> (row within the group of three, granule); granules past the end of the row re-read its last one
    {
        const int row3 = (lane * 49) >> 10, col = lane - row3 * kGran;
        const uint8_t* sb = src.base + (size_t)img * src.img_stride + (size_t)min(xg + col * 16, src.pitch - 16);
#pragma unroll
        for (int j = 0; j < kSrc / 3; j++) {
            if (j < 3 || j * 3 < n_src) {   // wave-uniform; the first 9 rows are always requested (clamped rows are harmless)
                const uint8_t* g = sb + (size_t)min(s_lo + j * 3 + row3, s_hi) * src.pitch;
                if (lane < 63)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)&slab[wave][j * 3 * kRowB], 16, 0, 0);
            }
        }
    }
    const int woff = active ? base - xg : 0;   // this lane's window inside a staged row: 0 .. 324
    // phase A
#pragma unroll
    for (int r = 0; r < kSrc; r++) {
        if (r < 9 || r < n_src) {   // wave-uniform; rows 0..8 unconditionally, so that their LDS reads can be batched
            const uint32_t* q = reinterpret_cast<const uint32_t*>(&slab[wave][r * kRowB + woff]);
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, o0);
            const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, o0);
            uint32_t H[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                H[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(hi, lo, sel[i])),
                                              __builtin_bit_cast(ushort2v, cw[i]), 0u, false) >> 4;
            park[wave][r][lane] = uint2{H[0] | (H[1] << 16), H[2] | (H[3] << 16)};
        }
    }
    // phase B
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)(uint32_t)dx0;
    auto out_row = [&](int k) {
        const int n0 = (int)(ti[k] & 0xffffu) - s_lo, n1 = (int)(ti[k] >> 16) - s_lo;
        const uint32_t b0 = tc[k] & 0xffffu, b1 = tc[k] >> 16;
        const uint2 A = park[wave][n0][lane], B = park[wave][n1][lane];
        const uint32_t t0 = sdwa_hi_sum(sdwa_mul_lo(b0, A.x), sdwa_mul_lo(b1, B.x));
        const uint32_t t1 = sdwa_hi_sum(sdwa_mul_hi(b0, A.x), sdwa_mul_hi(b1, B.x));
        const uint32_t t2 = sdwa_hi_sum(sdwa_mul_lo(b0, A.y), sdwa_mul_lo(b1, B.y));
        const uint32_t t3 = sdwa_hi_sum(sdwa_mul_hi(b0, A.y), sdwa_mul_hi(b1, B.y));
        const ushort2v two = {2, 2};
        const ushort2v p01 = (__builtin_bit_cast(ushort2v, t0 | (t1 << 16)) + two) >> 2;
        const ushort2v p23 = (__builtin_bit_cast(ushort2v, t2 | (t3 << 16)) + two) >> 2;
        const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
        if (active) *reinterpret_cast<uint32_t*>(d + (size_t)(dy0 + k) * dst.pitch) = packed;
    };
    if (n_out == R) {   // wave-uniform: every band but the last of an image
#pragma unroll
        for (int k = 0; k < R; k++) out_row(k);
    } else {
#pragma unroll
        for (int k = 0; k < R; k++)
            if (k < n_out) out_row(k);
    }
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 on one reference cell ROI per workgroup (cell loop + cv::FAST, ORBextractor.cc:805-872).
//   phase 0  stage the ROI (<= 76x76 bytes) in LDS, keeping the global 4-byte column phase so that every
//            later access to a group of 4 horizontally adjacent pixels is one aligned ds_read_b32
//   phase 1  per 4-pixel group: cheap necessary test on the two antipodal compass pairs at minTh from five
//            dwords (rows -3, 0, +3 and the +-3 column shifts by v_alignbyte); every (pixel, polarity) that
//            passes is appended to an LDS work list (wave prefix sum, one LDS atomic per wave)
//   phase 2  full arc test for the work list only — all lanes busy; one polarity per entry: a pixel cannot
//            be a dark and a bright corner at once, so S = A' - 1 with A' = max over the 16 arcs of the min
//            over 9 contiguous signed contrasts of the entry's polarity (v_min3/v_max3 network)
//   phase 3  strict 3x3 NMS inside the ROI's detection area; iniTh set if non-empty, else minTh set;
//            survivors written in scan order (ascending y, then x) to the cell's fixed slot run
// The FAST score is threshold independent for detected corners (cornerScore returns
// max(th, A, -B) - 1 and a corner has max(A,-B) > th), so one score plane serves both thresholds:
// corner at th  <=>  S >= th, and NMS at iniTh keeps exactly the minTh survivors with S >= iniTh.
// ------------------------------------------------------------------------------------------------
constexpr int kTileFront = 4;        // bytes in front of the tile so that column -4..-1 reads stay in bounds
// LDS geometry of the FAST kernel.  GeoLarge covers any legal cell (ROI <= 76 x 76); GeoSmall covers ROIs up to
// 46 x 57 — every cell of the KITTI / EuRoC / 4Seasons geometries — in 11 KB instead of 23 KB, which lifts the
// residency from 6 to 8 workgroups per CU.
struct GeoLarge {
    static constexpr int kTileRows = 77;    // ROI rows <= 76 (+1 spare row for harmless over-reads)
    static constexpr int kTilePitch = 84;   // bytes: 3 (column phase) + 76 (ROI) + over-read slack, multiple of 4
    static constexpr int kScorePitch = 88;  // 4 (apron group) + 4*20 (groups) + 4
    static constexpr int kScoreRows = 73;   // detection rows <= 70, +1 apron above, +1 below, +1 spare
    static constexpr int kMaxDet = 70;      // max detection height
    static constexpr int kWorkCap = 4096;   // work-list entries per chunk
    static constexpr int kThreads = 256;    // <= 6 tasks per thread (8 mask bits each in a 64-bit word)
    static constexpr int kWordsPerRow = 3;  // bitmap words per detection row (<= 80 columns)
    static constexpr int kMaxR = 6;         // detection rows per thread: ceil(70 / (256 / 20))
    static constexpr int kMinWaves = 4;     // waves per SIMD the register allocation must allow
};
struct GeoSmall {
    static constexpr int kTileRows = 58;    // ROI rows <= 57
    static constexpr int kTilePitch = 52;   // 4 (first group offset) + 4*11 (groups) + 4
    static constexpr int kScorePitch = 52;
    static constexpr int kScoreRows = 54;   // detection rows <= 51
    static constexpr int kMaxDet = 51;
    static constexpr int kWorkCap = 1536;   // 9.7 KB per workgroup in total -> 16 workgroups (32 waves) per CU
    static constexpr int kThreads = 128;    // 2 waves per cell: the task / work-list loops run fuller than with 4 (64 measured slower)
    static constexpr int kWordsPerRow = 2;  // <= 44 columns
    static constexpr int kMaxR = 5;         // detection rows per thread: ceil(51 / (128 / 11))
    static constexpr int kMinWaves = 8;     // 16 workgroups x 2 waves per CU: <= 64 VGPRs
};

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }
// The arc network below is 32 x min3 + 8 x max3.  Written with min()/max() the compiler re-associates it into ~48 two-input
// v_min_i32 + 12 max (all issue at the same slow-class VALU rate as the three-input forms), so the three-input
// instructions are spelled out.
__device__ __forceinline__ int vmin3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int vmax3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// max over the 16 arcs of 9 contiguous circle pixels of min(sgn * (v - p)); p = LDS pointer to the centre
template <class GEO>
__device__ __forceinline__ int fast_arc_contrast(const uint8_t* p, int sgn) {
    // circle offsets (x,y): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    const int sv = sgn * (int)p[0], ns = -sgn;
    int d[16];
    d[0] = (int)p[3 * GEO::kTilePitch] * ns + sv;
    d[1] = (int)p[3 * GEO::kTilePitch + 1] * ns + sv;
    d[2] = (int)p[2 * GEO::kTilePitch + 2] * ns + sv;
    d[3] = (int)p[1 * GEO::kTilePitch + 3] * ns + sv;
    d[4] = (int)p[3] * ns + sv;
    d[5] = (int)p[-1 * GEO::kTilePitch + 3] * ns + sv;
    d[6] = (int)p[-2 * GEO::kTilePitch + 2] * ns + sv;
    d[7] = (int)p[-3 * GEO::kTilePitch + 1] * ns + sv;
    d[8] = (int)p[-3 * GEO::kTilePitch] * ns + sv;
    d[9] = (int)p[-3 * GEO::kTilePitch - 1] * ns + sv;
    d[10] = (int)p[-2 * GEO::kTilePitch - 2] * ns + sv;
    d[11] = (int)p[-1 * GEO::kTilePitch - 3] * ns + sv;
    d[12] = (int)p[-3] * ns + sv;
    d[13] = (int)p[1 * GEO::kTilePitch - 3] * ns + sv;
    d[14] = (int)p[2 * GEO::kTilePitch - 2] * ns + sv;
    d[15] = (int)p[3 * GEO::kTilePitch - 1] * ns + sv;
    int mn3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) mn3[i] = vmin3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    int A = -512;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {  // arcs i..i+8 and i+1..i+9
        const int a0 = vmin3(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]);
        const int a1 = vmin3(mn3[(i + 1) & 15], mn3[(i + 4) & 15], mn3[(i + 7) & 15]);
        A = vmax3(A, a0, a1);
    }
    return A;
}

// the same with a window pointer w = centre - 3 * pitch - 3: every ds_read offset is non-negative, which matters when the tile
// lives in dynamic LDS (runtime base: negative offsets cannot be folded into the instruction's unsigned offset field)
template <class GEO>
__device__ __forceinline__ int fast_arc_contrast_win(const uint8_t* w, int sgn) {
    constexpr int P = GEO::kTilePitch;
    const int sv = sgn * (int)w[3 * P + 3], ns = -sgn;
    int d[16];
    d[0] = (int)w[6 * P + 3] * ns + sv;
    d[1] = (int)w[6 * P + 4] * ns + sv;
    d[2] = (int)w[5 * P + 5] * ns + sv;
    d[3] = (int)w[4 * P + 6] * ns + sv;
    d[4] = (int)w[3 * P + 6] * ns + sv;
    d[5] = (int)w[2 * P + 6] * ns + sv;
    d[6] = (int)w[1 * P + 5] * ns + sv;
    d[7] = (int)w[0 * P + 4] * ns + sv;
    d[8] = (int)w[0 * P + 3] * ns + sv;
    d[9] = (int)w[0 * P + 2] * ns + sv;
    d[10] = (int)w[1 * P + 1] * ns + sv;
    d[11] = (int)w[2 * P + 0] * ns + sv;
    d[12] = (int)w[3 * P + 0] * ns + sv;
    d[13] = (int)w[4 * P + 0] * ns + sv;
    d[14] = (int)w[5 * P + 1] * ns + sv;
    d[15] = (int)w[6 * P + 2] * ns + sv;
    int mn3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) mn3[i] = vmin3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    int A = -512;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const int a0 = vmin3(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]);
        const int a1 = vmin3(mn3[(i + 1) & 15], mn3[(i + 4) & 15], mn3[(i + 7) & 15]);
        A = vmax3(A, a0, a1);
    }
    return A;
}

// inclusive prefix sum over the wave with DPP adds only (no LDS crossbar round trips): shifts inside each row of 16
// lanes, then the row totals are chained through lanes 15 / 31 (lanes shifted in from outside a row read 0)
__device__ __forceinline__ int wave_incl_scan(int v, int) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
__device__ __forceinline__ uint32_t as_u32(short2v v) { return __builtin_bit_cast(uint32_t, v); }


// block-wide exclusive prefix of a per-thread count (4 waves); returns the grand total through *total
template <int WAVES>
__device__ __forceinline__ int block_excl_scan(int cnt, int lane, int wave, int* wave_tot /* LDS[WAVES] */, int* total) {
    const int incl = wave_incl_scan(cnt, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const int c = wave_tot[w];
        if (w < wave) before += c;
        tot += c;
    }
    *total = tot;
    return before + incl - cnt;
}

template <bool ALIGNED, class GEO>
__global__ __launch_bounds__(GEO::kThreads, GEO::kMinWaves) void fast_cells_kernel(PyramidView pyr, const CellDesc* __restrict__ cells,
                                                         int ini_th, int min_th, int slots_per_image,
                                                         Cand16* __restrict__ slots, int* __restrict__ cell_count,
                                                         int n_cells, uint32_t gx_magic, int debug_stop) {
    constexpr int T = GEO::kThreads, P = GEO::kTilePitch, SP = GEO::kScorePitch;
    constexpr int kScoreBytes = (GEO::kScoreRows * SP + 15) & ~15;
    constexpr int kBitWords = GEO::kWordsPerRow * GEO::kMaxDet;
    static_assert(kBitWords <= T, "one bitmap word per thread");
    __shared__ __attribute__((aligned(16))) uint8_t tile_mem[kTileFront + GEO::kTileRows * P + 8];
    __shared__ __attribute__((aligned(16))) uint8_t score[kScoreBytes];
    __shared__ uint16_t work[GEO::kWorkCap];
    __shared__ int wave_tot[2][T / 64];
    __shared__ uint32_t kbits[kBitWords];
    __shared__ int kprefix[kBitWords];
    __shared__ uint16_t lut[32];
    __shared__ uint16_t tbase[T];  // per thread: (tile row of its first detection row) << 7 | tile column of its group
    uint8_t* const tile = tile_mem + kTileFront;

    // XCD-aware order: consecutive workgroups are dealt round-robin to the 8 XCDs (each with a private L2); remap
    // the linear id so that every XCD works through one contiguous run of (image, cell) pairs and neighbouring
    // cells — which share their 3-pixel halo and 128-byte lines — hit the same L2.
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned chunk = (total + 7) >> 3;
    unsigned wg = (lin & 7u) * chunk + (lin >> 3);
    if (total & 7u) wg = lin;  // ragged totals keep the plain order (bench / test geometries are multiples of 8 images)
    const int img = gx_magic ? (int)__umulhi(wg, gx_magic) : (int)(wg / gridDim.x);  // host-checked exact reciprocal
    const int cell_id = (int)(wg - (unsigned)img * gridDim.x);
    const CellDesc cd = cells[cell_id];
    const LevelView lv = pyr.lv[cd.level];
    const int rw = cd.rw, rh = cd.rh;
    const int dh = rh - 6;                // detection rows (FAST skips 3 px on every side of its input)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ga = cd.x0 & ~3;            // tile column 0 = level column ga (keeps the 4-byte phase)
    const int x_lo = cd.x0 + 3, x_hi = cd.x0 + rw - 3;
    const int gx0 = x_lo & ~3;            // first 4-pixel group (may start left of x_lo)
    const int G = cd.G;                   // groups per detection row = (x_hi - gx0 + 3) >> 2
    const int c_lo = gx0 - ga;            // tile column of the first group (multiple of 4)
    const uint32_t magic = cd.g_magic;    // n / G == (n * magic) >> 20 for n < 2^20 / G

    // phase 0: stage the ROI.  Lane column c = dword of the tile row, kRowsPerPass rows per pass; all passes' loads are issued
    // back to back (one memory round trip per cell).  Rows past the ROI are clamped to its last row (an unconditional load of
    // a row that exists; what lands in the tile rows below the ROI is never used), so no pass needs a predicate.
    const uint8_t* src = lv.base + (size_t)img * lv.img_stride + (size_t)cd.y0 * lv.pitch;  // wave-uniform: SGPR base
    if (ALIGNED) {
        constexpr int kColLanes = P <= 64 ? 16 : 32;  // dwords per tile row: <= 13 (GeoSmall) / <= 21 (GeoLarge)
        constexpr int kRowsPerPass = T / kColLanes;
        constexpr int kPasses = (GEO::kTileRows - 1 + kRowsPerPass - 1) / kRowsPerPass;  // rh <= kTileRows - 1
        const int c = tid & (kColLanes - 1), r0 = tid / kColLanes;
        if (c < cd.ndw) {  // dwords per tile row = (x0 + rw - ga + 3) >> 2
            const uint32_t col = (uint32_t)(ga + 4 * c);
            uint32_t v[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; p++) {
                const uint32_t row = (uint32_t)min(r0 + p * kRowsPerPass, rh - 1);
                v[p] = *reinterpret_cast<const uint32_t*>(src + (__umul24(row, (uint32_t)lv.pitch) + col));
            }
            uint8_t* l0 = &tile[r0 * P + 4 * c];
#pragma unroll
            for (int p = 0; p < kPasses; p++)
                if ((p + 1) * kRowsPerPass <= GEO::kTileRows || r0 + p * kRowsPerPass < GEO::kTileRows)
                    *reinterpret_cast<uint32_t*>(l0 + p * kRowsPerPass * P) = v[p];
        }
    } else {
        const int off = cd.x0 - ga;
        const uint32_t bmagic = cd.rw_magic;
        for (int i = tid; i < rh * rw; i += T) {
            const int y = (int)(__umul24((uint32_t)i, bmagic) >> 20), x = i - (int)__umul24((uint32_t)y, (uint32_t)rw);
            tile[(int)__umul24((uint32_t)y, P) + off + x] = src[(size_t)(__umul24((uint32_t)y, (uint32_t)lv.pitch) + (uint32_t)(cd.x0 + x))];
        }
    }
    // score plane and keep-bitmap start at zero (16-byte stores, no loop)
    auto clear_planes = [&]() {
#pragma unroll
        for (int k = 0; k < (kScoreBytes / 16 + T - 1) / T; k++)
            if ((k + 1) * T <= kScoreBytes / 16 || tid + k * T < kScoreBytes / 16)
                reinterpret_cast<uint4*>(score)[tid + k * T] = uint4{0, 0, 0, 0};
        if (tid < kBitWords) kbits[tid] = 0;
    };
    clear_planes();
    if (tid < 32) lut[tid] = (uint16_t)((((tid >> 1) & 3) << 7) + (tid >> 3) + ((tid & 1) << 15));  // flag bit -> work entry offset

    // Quick-test mapping: thread (strip, g) owns the 4-pixel column group g of R consecutive detection rows, so the column
    // clipping is a per-thread constant and the rows above / below come out of one register window.
    // Two ways to deal the detection rows: uniformly (strip = tid / G over the whole workgroup, R rows each), or per wave
    // (64 / G strips inside every wave; wave w takes rw[w] rows per thread from row yw[w] on) when that needs fewer iterations
    // of the row loop — 39 rows over 2 x 6 strips are 4 + 3 instead of 4 + 4 (host-decided per cell, block-uniform).
    int g_own, R, y_b;
    if ((T == 128 ? cd.by_wave[0] : cd.by_wave[1]) == 0) {
        const int strip = (int)(__umul24((uint32_t)tid, magic) >> 20);
        g_own = tid - strip * G;
        R = T == 128 ? cd.R128 : cd.R256;               // wave-uniform
        y_b = strip * R;                                // first detection row of the thread
    } else {
        const int sw = __builtin_amdgcn_readfirstlane(wave) * 8;
        const int sl = (int)(__umul24((uint32_t)lane, magic) >> 20);
        g_own = lane - sl * G;
        R = (int)(((T == 128 ? cd.rw128 : cd.rw256) >> sw) & 255u);    // wave-uniform (scalar)
        const int y_w = (int)(((T == 128 ? cd.yw128 : cd.yw256) >> sw) & 255u);
        y_b = sl < (int)cd.spw ? y_w + sl * R : dh;     // lanes past the wave's last strip: no rows
    }
    const int nrows = min(max(dh - y_b, 0), R);         // 0 for the threads beyond the last strip
    const int c_own = c_lo + 4 * g_own;                 // tile column of pixel 0 of the group
    uint32_t Hm;                                        // 0x80 in every byte whose pixel lies inside [x_lo, x_hi)
    {
        const int xg = ga + c_own;
        const int vlo = min(max(x_lo - xg, 0), 4), vhi = min(max(x_hi - xg, 0), 4);
        Hm = (0x80808080u << (8 * vlo)) & (uint32_t)(0x0080808080ull >> (8 * (4 - vhi)));  // shifts by 32 must give 0
        if (vlo >= 4) Hm = 0;
    }
    tbase[tid] = (uint16_t)(((y_b + 3) << 7) | c_own);
    // Quick test at threshold th for the thread's rows: wA / wB receive 2 flags (dark, bright) per pixel, bit 8 j + 2 k (+ 1)
    // for pixel j of row k (rows 0..3 in wA, 4.. in wB).  All four pixels of a group are tested at once on raw bytes:
    //   A = sat0(v - t), B = sat255(v + t) per byte (v_pk_sub_u16 clamp on the even / odd bytes),
    //   p < v - t  <=>  p + (255 - A) + 1 <= 255  <=>  bit 7 of v_lerp_u8(P, ~A, 1) clear,
    //   p > v + t  <=>  p + (255 - B) >= 256      <=>  bit 7 of v_lerp_u8(P, ~B, 0) set          (~B = sat0(~v - t)),
    // i.e. one instruction per compass point, polarity and 4 pixels.  Every 9-arc contains one pixel of each antipodal
    // pair: a corner needs the predicate for (up OR down) AND (left OR right).
    auto quick_test = [&](int th, uint32_t& wA, uint32_t& wB) {
        wA = 0; wB = 0;
        const uint8_t* colp = &tile[(int)__umul24((uint32_t)y_b, P) + c_own];
        uint32_t cw[GEO::kMaxR + 6], lw[GEO::kMaxR], rw_[GEO::kMaxR];
#pragma unroll
        for (int r = 0; r < GEO::kMaxR + 6; r++)
            if (r < 9 || r - 6 < R) cw[r] = *reinterpret_cast<const uint32_t*>(colp + r * P);
#pragma unroll
        for (int k = 0; k < GEO::kMaxR; k++)
            if (k < 3 || k < R) {
                lw[k] = *reinterpret_cast<const uint32_t*>(colp + (k + 3) * P - 4);
                rw_[k] = *reinterpret_cast<const uint32_t*>(colp + (k + 3) * P + 4);
            }
        const ushort2v t2 = __builtin_bit_cast(ushort2v, (uint32_t)th * 0x00010001u);
#pragma unroll
        for (int k = 0; k < GEO::kMaxR; k++) {
            if (k >= 3 && k >= R) break;  // wave-uniform: the cells of the BASELINE geometries have R = 3 or 4
            const uint32_t V = cw[k + 3], U = cw[k], D = cw[k + 6];
            const uint32_t R3 = __builtin_amdgcn_alignbyte(rw_[k], V, 3);  // p[x+3] per byte
            const uint32_t L3 = __builtin_amdgcn_alignbyte(V, lw[k], 1);   // p[x-3] per byte
            // A = sat0(v - t), ~B = sat0(~v - t) on the even / odd bytes (16-bit lanes cannot borrow from each other)
            const uint32_t Ve = V & 0x00ff00ffu, Vo = (V >> 8) & 0x00ff00ffu;
            const uint32_t Ae = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Ve), t2));
            const uint32_t Ao = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Vo), t2));
            const uint32_t Qd = ~(Ae | (Ao << 8));
            const uint32_t Be = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Ve ^ 0x00ff00ffu), t2));
            const uint32_t Bo = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Vo ^ 0x00ff00ffu), t2));
            const uint32_t Qb = Be | (Bo << 8);
            const uint32_t one = 0x01010101u;
            // dark: bit 7 SET means "not darker"
            const uint32_t X = (__builtin_amdgcn_lerp(U, Qd, one) & __builtin_amdgcn_lerp(D, Qd, one)) |
                               (__builtin_amdgcn_lerp(L3, Qd, one) & __builtin_amdgcn_lerp(R3, Qd, one));
            const uint32_t Y = (__builtin_amdgcn_lerp(U, Qb, 0u) | __builtin_amdgcn_lerp(D, Qb, 0u)) &
                               (__builtin_amdgcn_lerp(L3, Qb, 0u) | __builtin_amdgcn_lerp(R3, Qb, 0u));
            const uint32_t hm = k < nrows ? Hm : 0u;
            const uint32_t z = (Y & hm) | ((~X & hm) >> 1);   // bits 8j+6 (dark), 8j+7 (bright)
            if (k < 4) wA |= z >> (6 - 2 * k);
            else wB |= z >> (6 - 2 * (k - 4));
        }
    };
    __syncthreads();
    if (debug_stop == 1) return;

    // Threshold passes (ORBextractor.cc:826,843-847): iniThFAST first; only a cell that ends up with no keypoint at all
    // is redone at minThFAST.  NMS at a threshold only sees the corners of that threshold (the others score 0 there),
    // so the first pass needs nothing below iniThFAST — half the quick-test survivors and arc tests of a minTh pass.
    const int sc_off = 4 - c_lo;  // score column = tile column + sc_off  (first group at score column 4)
    Cand16* const out = slots + (size_t)img * slots_per_image + cd.slot_off;
    int n_emitted = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int th = pass ? min_th : ini_th;
        // phase 1
        uint32_t wA, wB;
        quick_test(th, wA, wB);
        const int cnt = __popc(wA) + __popc(wB);
        int n_work = 0;
        const int my_base = block_excl_scan<T / 64>(cnt, lane, wave, wave_tot[0], &n_work);
        if (debug_stop == 2) return;

        // phase 2: every (pixel, polarity) that passed goes to the work list; the list is then processed with all lanes busy
        // (S = A' - 1 with A' = max over the 16 arcs of the min over 9 contiguous signed contrasts of the entry's polarity)
        // (the append loop runs as long as the busiest lane of the wave has flags left, so it only packs thread id and flag
        // bit; the list's consumers — all lanes busy — turn that into tile coordinates through two small tables)
        const uint32_t baseA = (uint32_t)(((y_b + 3) << 7) | c_own), baseB = baseA + (4u << 7);
        int w_begin = 0, n_corner = 0;   // this wave's stretch of the list, its corners (wave-uniform)
        if (n_work <= GEO::kWorkCap) {
            uint16_t* wp = &work[my_base];
            const uint32_t idA = (uint32_t)tid << 5, idB = idA | (1u << (5 + (T == 128 ? 7 : 8)));
            for (uint32_t w = wA; w; w &= w - 1) *wp++ = (uint16_t)(idA | (uint32_t)__builtin_ctz(w));
            for (uint32_t w = wB; w; w &= w - 1) *wp++ = (uint16_t)(idB | (uint32_t)__builtin_ctz(w));
            __syncthreads();
            if (debug_stop == 3) return;
            // every wave takes one contiguous stretch of the list and leaves the corners it finds packed at the front of that
            // stretch (it has read more entries than it has written): NMS and emission then loop over corners only — about a
            // third of the quick-test survivors — instead of skipping the other two thirds lane by lane
            w_begin = wave * ((((n_work + T / 64 - 1) / (T / 64)) + 63) & ~63);
            const int w_end = min(w_begin + ((((n_work + T / 64 - 1) / (T / 64)) + 63) & ~63), n_work);
            n_corner = 0;
            for (int i0 = w_begin; i0 < w_end; i0 += 64) {   // wave-uniform
                const int i = i0 + lane;
                bool corner = false;
                int ce = 0;
                if (i < w_end) {
                    const int id = work[i];
                    const int e = tbase[(id >> 5) & (T - 1)] + lut[id & 31] + ((id >> (5 + (T == 128 ? 7 : 8))) << 9);
                    const int ty = (e >> 7) & 127, tx = e & 127;
                    const int A = fast_arc_contrast<GEO>(&tile[(int)__umul24((uint32_t)ty, P) + tx], (e & 0x8000) ? -1 : 1);
                    if (A > th) {
                        score[(int)__umul24((uint32_t)(ty - 2), SP) + tx + sc_off] = (uint8_t)(A - 1);  // score row = y + 1
                        ce = e & 0x3FFF;
                        corner = true;
                    }
                }
                const unsigned long long bm = __ballot(corner);
                if (corner)
                    work[w_begin + n_corner + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = (uint16_t)ce;
                n_corner += __popcll(bm);
            }
        } else {
            // saturated cell (more quick-test survivors than the list holds): every thread scores its own survivors
#pragma unroll 1
            for (int half = 0; half < 2; half++)
#pragma unroll 1
                for (uint32_t w = half ? wB : wA; w; w &= w - 1) {
                    const int e = lut[__builtin_ctz(w)] + (half ? baseB : baseA);
                    const int ty = (e >> 7) & 127, tx = e & 127;
                    const int A = fast_arc_contrast<GEO>(&tile[(int)__umul24((uint32_t)ty, P) + tx], (e & 0x8000) ? -1 : 1);
                    if (A > th) score[(int)__umul24((uint32_t)(ty - 2), SP) + tx + sc_off] = (uint8_t)(A - 1);
                }
        }
        __syncthreads();
        if (debug_stop == 4) return;

        if (n_work <= GEO::kWorkCap) {
            // phase 3 (common case: the whole work list fitted): NMS and ordered emission driven by the corner list.
            // kept corners set a bit in a row-major bitmap of the detection area; the rank of a corner in scan order is
            // the popcount of the bits before it (prefix over the bitmap words, one word per thread).
            constexpr int wpr = GEO::kWordsPerRow;  // bitmap words per detection row
            const int nwords = dh * wpr;
            uint32_t mine_keep = 0;  // per-thread record of the corners it owns: list slots tid, tid + T, ... (cap / T <= 32)
            {
                int slot = 0;
                for (int m0 = 0; m0 < n_corner; m0 += 64, slot++) {
                    if (m0 + lane >= n_corner) continue;
                    const int e = work[w_begin + m0 + lane];
                    const int ty = e >> 7, tx = e & 127;
                    const uint8_t* q = &score[(int)__umul24((uint32_t)(ty - 2), SP) + tx + sc_off];
                    const int sv = q[0];
                    int m = max3i(q[-SP - 1], q[-SP], q[-SP + 1]);
                    m = max3i(m, q[-1], q[1]);
                    m = max(m, max3i(q[SP - 1], q[SP], q[SP + 1]));
                    if (sv > m) {
                        const int bx = tx - c_lo, by = ty - 3;       // column inside the group span, detection row
                        const int b = by * (32 * wpr) + bx;
                        atomicOr(&kbits[b >> 5], 1u << (b & 31));
                        mine_keep |= 1u << slot;
                    }
                }
            }
            __syncthreads();
            const uint32_t myword = tid < nwords ? kbits[tid] : 0u;
            int n_out = 0;
            const int wprefix = block_excl_scan<T / 64>(__popc(myword), lane, wave, wave_tot[1], &n_out);
            if (tid < nwords) kprefix[tid] = wprefix;
            __syncthreads();
            {
                int slot = 0;
                for (int m0 = 0; m0 < n_corner; m0 += 64, slot++) {
                    if (!(mine_keep & (1u << slot))) continue;
                    const int e = work[w_begin + m0 + lane];
                    const int ty = e >> 7, tx = e & 127;
                    const int b = (ty - 3) * (32 * wpr) + (tx - c_lo);
                    const int rank = kprefix[b >> 5] + __popc(kbits[b >> 5] & ((1u << (b & 31)) - 1u));
                    Cand16 c;
                    c.x = (uint16_t)(ga + tx - kMinBorder);
                    c.y = (uint16_t)(cd.y0 + ty - kMinBorder);
                    c.score = score[(int)__umul24((uint32_t)(ty - 2), SP) + tx + sc_off];
                    c.pad = 0;
                    out[rank] = c;
                }
            }
            n_emitted = n_out;
        } else {
            // saturated cell: strict 3x3 NMS by scanning the score plane; one task = one group of one detection row in scan
            // order, balanced consecutive task ranges per thread (thread t owns tasks [t*n/T, (t+1)*n/T)), 4 flag bits per task
            const int n_task = dh * G;
            const int t_begin = (int)(__umul24((uint32_t)tid, (uint32_t)n_task) / T),
                      t_end = (int)(__umul24((uint32_t)tid + 1u, (uint32_t)n_task) / T);
            uint64_t keep = 0;
            {
                int y = (int)(__umul24((uint32_t)t_begin, magic) >> 20);
                int g = t_begin - y * G;
                for (int task = t_begin, k = 0; task < t_end; task++, k++) {
                    const uint8_t* sp = &score[(y + 1) * SP + 4 + 4 * g];
                    const uint32_t S = *reinterpret_cast<const uint32_t*>(sp);
                    if (S) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int sv = (S >> (8 * j)) & 255;
                            if (sv) {
                                const uint8_t* q = sp + j;
                                int m = max3i(q[-SP - 1], q[-SP], q[-SP + 1]);
                                m = max3i(m, q[-1], q[1]);
                                m = max(m, max3i(q[SP - 1], q[SP], q[SP + 1]));
                                if (sv > m) keep |= 1ull << (4 * k + j);
                            }
                        }
                    }
                    if (++g == G) { g = 0; y++; }
                }
            }
            // ordered emission — a thread's tasks are consecutive in scan order, so the block-wide prefix of the
            // per-thread counts is the rank in (ascending y, then x) order
            int n_out = 0;
            int pos = block_excl_scan<T / 64>(__popcll(keep), lane, wave, wave_tot[1], &n_out);
            if (keep) {
                int y = (int)(__umul24((uint32_t)t_begin, magic) >> 20);
                int g = t_begin - y * G;
                uint64_t m = keep;
                for (int task = t_begin; task < t_end; task++, m >>= 4) {
                    uint32_t m4 = (uint32_t)m & 15u;
                    while (m4) {
                        const int j = __ffs(m4) - 1;
                        m4 &= m4 - 1;
                        Cand16 c;
                        c.x = (uint16_t)(gx0 + 4 * g + j - kMinBorder);
                        c.y = (uint16_t)(cd.y0 + 3 + y - kMinBorder);
                        c.score = score[(y + 1) * SP + 4 + 4 * g + j];
                        c.pad = 0;
                        out[pos++] = c;
                    }
                    if (++g == G) { g = 0; y++; }
                }
            }
            n_emitted = n_out;
        }
        if (n_emitted > 0 || pass == 1 || ini_th == min_th) break;
        __syncthreads();
        clear_planes();   // (kept corners there were none; the scores of the failed pass must not leak into the next)
        __syncthreads();
    }
    if (tid == 0) cell_count[(size_t)img * n_cells + cell_id] = n_emitted;
}

// ------------------------------------------------------------------------------------------------
// FAST over a STRIP of up to K horizontally adjacent cells (StripDesc, orb_host.h) in one workgroup.
// The detection areas of the cells of a cell row are contiguous, so staging, the quick test, the work list and the arc scores
// treat the strip as one wide cell — one set-up and two block scans per wave for K / (T / 64) cells instead of one per
// two waves and cell, fuller lanes in every loop, the 3-pixel column halo staged once per strip instead of once per cell.
// What the reference defines per cell stays per cell:
//   * NMS (cv::FAST runs on the cell's ROI, so a corner on the first / last detection column has no neighbour beyond it):
//     the score plane gives every cell two apron columns of its own (score column = tile column + 2 * cell), never written;
//   * emission order and slot run: kept corners set a bit in the cell's own row bitmap (64 bits per cell and detection row);
//     rank = popcount prefix inside the cell;
//   * the minThFAST fallback (ORBextractor.cc:843-847): a cell-activity mask selects the cells whose pixels the quick test
//     lets through; the strip is run again at minThFAST for the cells that came out empty;
//   * a strip with more quick-test survivors than the work list holds is redone one cell at a time (same mask; the list holds
//     the flags of any single cell by construction).
// ------------------------------------------------------------------------------------------------
template <int K>
struct StripGeo {
    static constexpr int kMaxDw = (K * kStripMaxCellW + 6 + 3 + 3) / 4;  // dwords per staged ROI row (4-byte phase included)
    static constexpr int P = 4 * kMaxDw + 8;                              // tile pitch, bytes
    static constexpr int SP = (P + 2 * K + 3) & ~3;                       // score pitch (two apron columns per cell)
    static constexpr int kTilePitch = P;                                  // (fast_arc_contrast's name for it)
};
struct StripLds {  // byte offsets of the dynamic LDS carve (host-computed from the geometry's tallest ROI / largest cell)
    int tile, score, score_bytes, work, work_cap, kbits, kprefix, total;
};
template <int K>
StripLds strip_lds_layout(int max_rh, int work_cap) {
    StripLds L;
    int o = 0;
    // (the tile comes last: the kernel addresses it through a pointer biased by -(3 rows + 3 bytes), which must stay inside LDS)
    L.score = o; L.score_bytes = ((max_rh - 6 + 3) * StripGeo<K>::SP + 15) & ~15; o += L.score_bytes;
    L.work_cap = (work_cap + 7) & ~7;
    L.work = o; o += L.work_cap * 2;
    // the row bitmaps and their prefix live in the head of the TILE: the tile is dead once the arc scores are written; a strip that
    // has to run again (cells redone at minThFAST, or one cell at a time) stages its ROI again
    const int max_words = std::min(K * std::max(max_rh - 6, 1), kStripThreads);   // bitmap rows: cells x detection rows
    L.tile = o;
    L.kbits = o;
    L.kprefix = o + 2 * max_words * 4;
    o += std::max((kTileFront + (max_rh + 1) * StripGeo<K>::P + 8 + 15) & ~15, 2 * max_words * 4 + (((max_words + 1) * 4 + 15) & ~15));
    L.total = (o + 15) & ~15;
    return L;
}

template <int K, int T>
__global__ __launch_bounds__(T) void fast_strip_kernel(PyramidView pyr, const StripDesc* __restrict__ strips, int n_strips, int ini_th,
                                                       int min_th, int slots_per_image, Cand16* __restrict__ slots,
                                                       int* __restrict__ cell_count, int n_cells, uint32_t gx_magic, StripLds L, int debug_stop) {
    using GEO = StripGeo<K>;
    constexpr int P = GEO::P, SP = GEO::SP, kMaxR = kStripMaxR;
    constexpr int kIdBits = T == 256 ? 8 : 7;
    static_assert(T == 128 || T == 256, "entry ids hold 7 or 8 thread bits");
    extern __shared__ __attribute__((aligned(16))) uint8_t strip_mem[];
    __shared__ int wave_tot[2][T / 64];
    __shared__ uint16_t lut[32];
    __shared__ uint16_t tbase[T];
    __shared__ int slot_off_s[K];
    uint8_t* const tile = strip_mem + L.tile + kTileFront;
    const uint8_t* const tile_win = tile - 3 * GEO::P - 3;   // a pixel's 7 x 7 window starts here + its tile offset
    uint8_t* const score = strip_mem + L.score;
    uint16_t* const work = reinterpret_cast<uint16_t*>(strip_mem + L.work);
    uint32_t* const kbits = reinterpret_cast<uint32_t*>(strip_mem + L.kbits);     // [cell * dh + row][2]
    int* const kprefix = reinterpret_cast<int*>(strip_mem + L.kprefix);           // [cell * dh + row], + total at the end

    // XCD-aware order (see fast_cells_kernel)
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned chunk = (total + 7) >> 3;
    unsigned wg = (lin & 7u) * chunk + (lin >> 3);
    if (total & 7u) wg = lin;
    const int img = gx_magic ? (int)__umulhi(wg, gx_magic) : (int)(wg / gridDim.x);
    const int strip_id = (int)(wg - (unsigned)img * gridDim.x);
    const StripDesc sd = strips[strip_id];
    const LevelView lv = pyr.lv[sd.level];
    const int rw = sd.rw, rh = sd.rh, dh = rh - 6, ncell = sd.ncell, w_cell = sd.w_cell;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ga = sd.x0 & ~3;
    const int x_lo = sd.x0 + 3, x_hi = sd.x0 + rw - 3;
    const int gx0 = x_lo & ~3;
    const int G = sd.G;
    const int c_lo = gx0 - ga;
    const uint32_t magic = sd.g_magic, wc_magic = sd.wc_magic;

    // phase 0: stage the ROI with 16-byte lanes: lane (row r0 + 16 p, quad c) copies dwords 4c .. 4c + 3 of its row — 3-4 load
    // instructions per thread for the whole strip, all in flight at once (the level rows are only 4-byte aligned for these loads,
    // which the memory pipeline splits; the tile rows are written as dwords).  A quad may run up to 12 bytes past the ROI: still
    // inside the level row (ROIs end 16 pixels before the border).
    auto stage_tile = [&]() {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint8_t* src = lv.base + (size_t)img * lv.img_stride + (size_t)sd.y0 * lv.pitch + ga;
        const int nq = (sd.ndw + 3) >> 2;
        constexpr int kRowsPerPass = T / 16, kPasses = (58 + kRowsPerPass - 1) / kRowsPerPass;
        const int c = tid & 15, r0 = tid >> 4;
        if (c < nq) {
            u32x4 v[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; p++) {
                const uint32_t row = (uint32_t)min(r0 + p * kRowsPerPass, rh - 1);
                __builtin_memcpy(&v[p], __builtin_assume_aligned(src + (__umul24(row, (uint32_t)lv.pitch) + 16u * (uint32_t)c), 4), 16);
            }
            uint32_t* l0 = reinterpret_cast<uint32_t*>(tile + r0 * P + 16 * c);
#pragma unroll
            for (int p = 0; p < kPasses; p++)
                if (r0 + p * kRowsPerPass < rh) {
                    uint32_t* d = l0 + p * kRowsPerPass * (P / 4);
                    d[0] = v[p].x; d[1] = v[p].y; d[2] = v[p].z; d[3] = v[p].w;
                }
        }
    };
    stage_tile();
    if (tid < 32) lut[tid] = (uint16_t)((((tid >> 1) & 3) << 8) + (tid >> 3) + ((tid & 1) << 15));  // flag bit -> work entry offset
    if (tid < K) slot_off_s[tid] = strips[strip_id].slot_off[tid];   // (indexing the register copy by lane would put it in scratch)

    // quick-test mapping: thread (strip of rows, group) — uniform split
    const int strip = (int)(__umul24((uint32_t)tid, magic) >> 20);
    const int g_own = tid - strip * G;
    const int R = sd.R;
    const int y_b = strip * R;
    const int nrows = min(max(dh - y_b, 0), R);
    const int c_own = c_lo + 4 * g_own;
    // per-pixel masks of the group: 0x80 in the bytes whose pixel lies in the detection range, split by the (at most two) cells
    // the group touches; cA = cell of the group's first detection pixel
    uint32_t HmA, HmB;
    int cA;
    {
        const int xg = ga + c_own;
        const int vlo = min(max(x_lo - xg, 0), 4), vhi = min(max(x_hi - xg, 0), 4);
        uint32_t Hm = (0x80808080u << (8 * vlo)) & (uint32_t)(0x0080808080ull >> (8 * (4 - vhi)));  // shifts by 32 must give 0
        if (vlo >= 4) Hm = 0;
        cA = min((int)(__umul24((uint32_t)max(xg + vlo - x_lo, 0), wc_magic) >> 20), ncell - 1);
        const int nb = min(max(x_lo + (cA + 1) * w_cell - xg, 0), 4);          // pixels of the group in front of cell cA + 1
        const uint32_t low = (uint32_t)((1ull << (8 * nb)) - 1ull);
        HmA = Hm & low; HmB = Hm & ~low;
    }
    tbase[tid] = (uint16_t)(((y_b + 3) << 8) | c_own);
    auto quick_test = [&](int th, uint32_t Hm, uint32_t& wA, uint32_t& wB) {
        wA = 0; wB = 0;
        const uint8_t* colp = &tile[(int)__umul24((uint32_t)y_b, P) + c_own];
        uint32_t cw[kMaxR + 6], lw[kMaxR], rw_[kMaxR];
#pragma unroll
        for (int r = 0; r < kMaxR + 6; r++)
            if (r < 9 || r - 6 < R) cw[r] = *reinterpret_cast<const uint32_t*>(colp + r * P);
#pragma unroll
        for (int k = 0; k < kMaxR; k++)
            if (k < 3 || k < R) {
                lw[k] = *reinterpret_cast<const uint32_t*>(colp + (k + 3) * P - 4);
                rw_[k] = *reinterpret_cast<const uint32_t*>(colp + (k + 3) * P + 4);
            }
        const ushort2v t2 = __builtin_bit_cast(ushort2v, (uint32_t)th * 0x00010001u);
#pragma unroll
        for (int k = 0; k < kMaxR; k++) {
            if (k >= 3 && k >= R) break;  // workgroup-uniform
            const uint32_t V = cw[k + 3], U = cw[k], D = cw[k + 6];
            const uint32_t R3 = __builtin_amdgcn_alignbyte(rw_[k], V, 3);
            const uint32_t L3 = __builtin_amdgcn_alignbyte(V, lw[k], 1);
            const uint32_t Ve = V & 0x00ff00ffu, Vo = (V >> 8) & 0x00ff00ffu;
            const uint32_t Ae = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Ve), t2));
            const uint32_t Ao = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Vo), t2));
            const uint32_t Qd = ~(Ae | (Ao << 8));
            const uint32_t Be = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Ve ^ 0x00ff00ffu), t2));
            const uint32_t Bo = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, Vo ^ 0x00ff00ffu), t2));
            const uint32_t Qb = Be | (Bo << 8);
            const uint32_t one = 0x01010101u;
            const uint32_t X = (__builtin_amdgcn_lerp(U, Qd, one) & __builtin_amdgcn_lerp(D, Qd, one)) |
                               (__builtin_amdgcn_lerp(L3, Qd, one) & __builtin_amdgcn_lerp(R3, Qd, one));
            const uint32_t Y = (__builtin_amdgcn_lerp(U, Qb, 0u) | __builtin_amdgcn_lerp(D, Qb, 0u)) &
                               (__builtin_amdgcn_lerp(L3, Qb, 0u) | __builtin_amdgcn_lerp(R3, Qb, 0u));
            const uint32_t hm = k < nrows ? Hm : 0u;
            const uint32_t z = (Y & hm) | ((~X & hm) >> 1);
            if (k < 4) wA |= z >> (6 - 2 * k);
            else wB |= z >> (6 - 2 * (k - 4));
        }
    };
    const int score_words = L.score_bytes >> 4;
    const int n_words = ncell * dh;          // bitmap rows (64 bits each) of the strip, <= T
    auto clear_score = [&]() {
        for (int i = tid; i < score_words; i += T) reinterpret_cast<uint4*>(score)[i] = uint4{0, 0, 0, 0};
    };
    const int sc_off = 4 - c_lo;
    Cand16* const out_img = slots + (size_t)img * slots_per_image;
    int* const cnt_img = cell_count + (size_t)img * n_cells + sd.cell0;

    // control: cells still to do at the current threshold; a saturated run is repeated one cell at a time
    uint32_t todo = (1u << ncell) - 1u, empties = 0;
    int th = ini_th;
    bool minpass = false, single = false, tile_dirty = false;
    for (;;) {
        if (todo == 0) {
            if (minpass || ini_th == min_th || empties == 0) break;
            todo = empties; empties = 0; th = min_th; minpass = true; single = false;
        }
        const uint32_t mask = single ? (todo & (0u - todo)) : todo;
        __syncthreads();       // staging done / the previous run's planes, lists and counts are no longer read
        if (debug_stop == 1) return;
        if (tile_dirty) {      // a run before this one put its bitmaps into the tile
            stage_tile();
            __syncthreads();
        }
        tile_dirty = true;
        clear_score();
        const uint32_t Hm = (((mask >> cA) & 1u) ? HmA : 0u) | (((mask >> (cA + 1)) & 1u) ? HmB : 0u);
        uint32_t wA, wB;
        quick_test(th, Hm, wA, wB);
        const int cnt = __popc(wA) + __popc(wB);
        int n_work = 0;
        const int my_base = block_excl_scan<T / 64>(cnt, lane, wave, wave_tot[0], &n_work);   // (its barrier also covers the clear)
        if (debug_stop == 2) return;
        if (n_work > L.work_cap) {
            tile_dirty = false;   // nothing was written over the tile
            if (!single) { single = true; continue; }
            // cannot happen (the list holds any single cell's flags); give the cell up rather than loop
            if (tid == 0) cnt_img[__builtin_ctz(mask)] = 0;
            todo &= ~mask;
            continue;
        }
        {
            uint16_t* wp = &work[my_base];
            const uint32_t idA = (uint32_t)tid << 5, idB = idA | (1u << (5 + kIdBits));
            for (uint32_t w = wA; w; w &= w - 1) *wp++ = (uint16_t)(idA | (uint32_t)__builtin_ctz(w));
            for (uint32_t w = wB; w; w &= w - 1) *wp++ = (uint16_t)(idB | (uint32_t)__builtin_ctz(w));
        }
        __syncthreads();
        if (debug_stop == 3) return;
        // arc scores, all lanes busy.  A wave works through the chunks wave, wave + T / 64, ... of the list and leaves the corners
        // it finds (cell << 14 | tile row << 8 | score column) packed at the front of its OWN chunks — it has read more entries
        // than it writes, so nothing unread is overwritten —: NMS and emission then loop over corners only (a third of the list).
        const int wbase = wave * 64;
        int n_corner = 0;                                    // wave-uniform
        for (int i0 = wbase; i0 < n_work; i0 += T) {
            const int i = i0 + lane;
            bool corner = false;
            int ce = 0;
            if (i < n_work) {
                const int id = work[i];
                const int e = tbase[(id >> 5) & (T - 1)] + lut[id & 31] + ((id >> (5 + kIdBits)) << 10);
                const int ty = (e >> 8) & 127, tx = e & 255;
                const int A = fast_arc_contrast_win<GEO>(tile_win + ((int)__umul24((uint32_t)ty, P) + tx), (e & 0x8000) ? -1 : 1);
                if (A > th) {
                    const int cell = (int)(__umul24((uint32_t)(tx + (ga - x_lo)), wc_magic) >> 20);
                    const int sx = tx + sc_off + 2 * cell;
                    score[(int)__umul24((uint32_t)(ty - 2), SP) + sx] = (uint8_t)(A - 1);
                    ce = (cell << 14) | ((e & 0x3F00) + sx);
                    corner = true;
                }
            }
            const unsigned long long bm = __ballot(corner);
            if (corner) {
                const int m = n_corner + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                work[(m >> 6) * T + wbase + (m & 63)] = (uint16_t)ce;
            }
            n_corner += __popcll(bm);
        }
        __syncthreads();
        if (debug_stop == 4) return;
        if (tid < n_words) *reinterpret_cast<uint2*>(&kbits[2 * tid]) = uint2{0, 0};   // (in the tile, which is dead from here on)
        __syncthreads();
        // NMS over the wave's own corners; kept corners set a bit in their cell's row bitmap
        uint32_t mine_keep = 0;
        {
            int slot = 0;
            for (int m0 = 0; m0 < n_corner; m0 += 64, slot++) {
                if (m0 + lane >= n_corner) continue;
                const int e = work[slot * T + wbase + lane];
                const int cell = e >> 14, ty = (e >> 8) & 63, sx = e & 255;
                const uint8_t* q = score + ((int)__umul24((uint32_t)(ty - 3), SP) + sx - 1);   // top-left neighbour
                const int sv = q[SP + 1];
                int m = max3i(q[0], q[1], q[2]);
                m = max3i(m, q[SP], q[SP + 2]);
                m = max(m, max3i(q[2 * SP], q[2 * SP + 1], q[2 * SP + 2]));
                if (sv > m) {
                    const int bit = sx - sc_off - 2 * cell + (ga - x_lo) - cell * w_cell;   // column inside the cell's detection area, < 64
                    const int wi = cell * dh + (ty - 3);
                    atomicOr(&kbits[2 * wi + (bit >> 5)], 1u << (bit & 31));
                    mine_keep |= 1u << slot;
                }
            }
        }
        __syncthreads();
        const uint32_t w0 = tid < n_words ? kbits[2 * tid] : 0u, w1 = tid < n_words ? kbits[2 * tid + 1] : 0u;
        int n_out = 0;
        const int wprefix = block_excl_scan<T / 64>(__popc(w0) + __popc(w1), lane, wave, wave_tot[1], &n_out);
        if (tid < n_words) kprefix[tid] = wprefix;
        if (tid == 0) kprefix[n_words] = n_out;
        __syncthreads();
        {
            int slot = 0;
            for (int m0 = 0; m0 < n_corner; m0 += 64, slot++) {
                if (!(mine_keep & (1u << slot))) continue;
                const int e = work[slot * T + wbase + lane];
                const int cell = e >> 14, ty = (e >> 8) & 63, sx = e & 255;
                const int tx = sx - sc_off - 2 * cell;
                const int bit = tx + (ga - x_lo) - cell * w_cell;
                const int wi = cell * dh + (ty - 3);
                const uint32_t b0 = kbits[2 * wi], b1 = kbits[2 * wi + 1];
                const int before = bit < 32 ? __popc(b0 & ((1u << bit) - 1u)) : __popc(b0) + __popc(b1 & ((1u << (bit - 32)) - 1u));
                const int rank = kprefix[wi] - kprefix[cell * dh] + before;
                Cand16 c;
                c.x = (uint16_t)(ga + tx - kMinBorder);
                c.y = (uint16_t)(sd.y0 + ty - kMinBorder);
                c.score = score[(int)__umul24((uint32_t)(ty - 2), SP) + sx];
                c.pad = 0;
                out_img[slot_off_s[cell] + rank] = c;
            }
        }
        // per-cell totals straight from the prefix (every thread reads the K + 1 cell boundaries: no further barrier)
        {
            int prev = 0;   // kprefix[0]
#pragma unroll
            for (int c = 0; c < K; c++) {
                if (c >= ncell) break;
                const int nxt = kprefix[(c + 1) * dh];
                if ((mask >> c) & 1u) {
                    if (tid == c) cnt_img[c] = nxt - prev;
                    if (!minpass && nxt == prev) empties |= 1u << c;
                }
                prev = nxt;
            }
        }
        todo &= ~mask;
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: per image exclusive scan of the cell counts (cells are already in the
// reference's row-major cell order), a one-block scan over images, then a gather.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_scan_cells_kernel(const int* __restrict__ cell_count, int n_cells,
                                                              const int* __restrict__ level_cell_begin, int nlevels,
                                                              int* __restrict__ cell_off, int* __restrict__ level_count,
                                                              int* __restrict__ img_total) {
    __shared__ int wave_tot[4];
    __shared__ int lvl[kMaxLevels];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* cnt = cell_count + (size_t)img * n_cells;
    int* off = cell_off + (size_t)img * n_cells;
    const int per = (n_cells + 255) / 256;
    const int b = tid * per, e = min(b + per, n_cells);
    // the thread's counts are read once (the first kCache of them stay in registers, their loads in flight together)
    constexpr int kCache = 4;
    int c[kCache];
#pragma unroll
    for (int k = 0; k < kCache; k++) c[k] = b + k < e ? cnt[b + k] : 0;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kCache; k++) s += c[k];
    for (int i = b + kCache; i < e; i++) s += cnt[i];
    if (tid < kMaxLevels) lvl[tid] = 0;
    int total = 0;
    int acc = block_excl_scan<4>(s, lane, wave, wave_tot, &total);
    if (tid == 0) img_total[img] = total;
    // offsets + per-level totals (a thread's cells are consecutive: they span at most a few levels)
    int cur_l = 0, cur_t = 0;
    auto visit = [&](int i, int v) {
        off[i] = acc;
        acc += v;
        while (i >= level_cell_begin[cur_l + 1]) {
            if (cur_t) atomicAdd(&lvl[cur_l], cur_t);
            cur_t = 0;
            cur_l++;
        }
        cur_t += v;
    };
#pragma unroll
    for (int k = 0; k < kCache; k++)
        if (b + k < e) visit(b + k, c[k]);
    for (int i = b + kCache; i < e; i++) visit(i, cnt[i]);
    if (cur_t) atomicAdd(&lvl[cur_l], cur_t);
    __syncthreads();
    if (tid < nlevels) level_count[(size_t)img * nlevels + tid] = lvl[tid];
}

__global__ __launch_bounds__(256) void cand_scan_images_kernel(const int* __restrict__ img_total, int n_images,
                                                               int* __restrict__ img_base /* n_images+1 */) {
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_images + 255) / 256;
    const int b = tid * per, e = min(b + per, n_images);
    const int first = b < e ? img_total[b] : 0;
    int s = first;
    for (int i = b + 1; i < e; i++) s += img_total[i];
    int total = 0;
    int acc = block_excl_scan<4>(s, lane, wave, wave_tot, &total);
    if (tid == 0) img_base[n_images] = total;
    if (b < e) { img_base[b] = acc; acc += first; }
    for (int i = b + 1; i < e; i++) { img_base[i] = acc; acc += img_total[i]; }
}

// 8 cells per 256-thread workgroup, 32 lanes per cell (a cell keeps ~20-40 candidates: one block of 64 threads per cell was
// 230 k nearly empty workgroups per batch, 56 us of launch machinery for 48 MB)
constexpr int kGatherCells = 16;   // 8 groups of 32 lanes, two cells each (their loads in flight together)
__global__ __launch_bounds__(256) void cand_gather_kernel(const CellDesc* __restrict__ cells, int n_cells,
                                                          int slots_per_image, const Cand16* __restrict__ slots,
                                                          const int* __restrict__ cell_count,
                                                          const int* __restrict__ cell_off,
                                                          const int* __restrict__ img_base, Cand16* __restrict__ compact) {
    const int c0 = blockIdx.x * kGatherCells + 2 * (int)(threadIdx.x >> 5), img = blockIdx.y, l = threadIdx.x & 31;
    if (c0 >= n_cells) return;
    const int c1 = min(c0 + 1, n_cells - 1);
    const int* cnt = cell_count + (size_t)img * n_cells;
    const int* off = cell_off + (size_t)img * n_cells;
    const int n0 = cnt[c0], n1 = c0 + 1 < n_cells ? cnt[c1] : 0;
    const int o0 = off[c0], o1 = off[c1], s0 = cells[c0].slot_off, s1 = cells[c1].slot_off, base = img_base[img];
    const Cand16* sp = slots + (size_t)img * slots_per_image;
    Cand16* d = compact + base;
    Cand16 v0{}, v1{};
    if (l < n0) v0 = sp[s0 + l];
    if (l < n1) v1 = sp[s1 + l];
    if (l < n0) d[o0 + l] = v0;
    if (l < n1) d[o1 + l] = v1;
    for (int i = l + 32; i < n0; i += 32) d[o0 + i] = sp[s0 + i];
    for (int i = l + 32; i < n1; i += 32) d[o1 + i] = sp[s1 + i];
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, Q8.8 fixed point [18,34,48,56,48,34,18], BORDER_REFLECT_101, per level.
// Tile = 64 x 16 output pixels per 256-thread block; 70 x 22 input pixels staged in LDS.
// ------------------------------------------------------------------------------------------------
// LDS-free streaming form: one thread owns a 4-pixel-wide column strip and walks kGaussRows + 6 input
// rows.  Per row it loads the 12 bytes [x0-4, x0+8) as aligned dwords (each load instruction is one fully
// coalesced 256-byte request per wave), forms the four horizontal sums with v_alignbyte + v_dot4_u32_u8 and
// scatters them into seven rotating vertical accumulators; a row is complete six input rows later and is
// stored as one aligned dword.  All levels of all images go in ONE launch (block -> level table).
// ALIGNED = every source row starts on a 4-byte boundary (our own planes, or a caller buffer with such a
// pitch); otherwise a fourth dword + funnel shift by the row's byte phase.  Column groups whose window
// reaches the right border (reflect-101 gather, byte loads) are left to gauss7_edge_kernel so that the
// streaming waves stay divergence free and the unrolled body stays small (instruction cache).
constexpr int kGaussRows = 35;  // 6 warm-up rows + 5 x 7 steady rows
struct GaussTaps { uint32_t k[7]; };   // the generic blur kernels take the Q8 taps at run time (Semantics::gauss_taps)

struct BlurPlan {
    int block_begin[kMaxLevels + 1];  // first blockIdx.x of each level
    int bx_count[kMaxLevels];         // blocks per strip-row of the level
    int nlevels;
};

__device__ __forceinline__ int refl101(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); }

template <bool ALIGNED>
__device__ __forceinline__ void gauss_row_sums(const uint8_t* __restrict__ sb, int pitch, int h, int yy, int x0,
                                               uint32_t hsum[4], uint32_t kGaussLo, uint32_t kGaussHi) {
    yy = refl101(yy, h);
    yy = min(max(yy, 0), h - 1);  // rows past the image only feed outputs that are never stored
    const uint8_t* rp = sb + (size_t)yy * pitch;
    uint32_t w0, w1, w2;
    if (ALIGNED) {
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(rp + x0);
        w1 = ap[0]; w2 = ap[1];
        w0 = x0 ? ap[-1] : 0u;
    } else {
        const uint32_t ph = (uint32_t)(reinterpret_cast<uintptr_t>(rp) & 3);
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(rp - ph + x0);
        const uint32_t d1 = ap[0], d2 = ap[1], d3 = ap[2];
        const uint32_t d0 = x0 ? ap[-1] : 0u;
        w0 = __builtin_amdgcn_alignbyte(d1, d0, ph);
        w1 = __builtin_amdgcn_alignbyte(d2, d1, ph);
        w2 = __builtin_amdgcn_alignbyte(d3, d2, ph);
    }
    if (x0 == 0) w0 = __builtin_amdgcn_perm(0u, w1, 0x01020300u);  // p[-1..-3] = p[1..3]
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t a = j == 3 ? w1 : __builtin_amdgcn_alignbyte(w1, w0, j + 1);
        const uint32_t b = j == 3 ? w2 : __builtin_amdgcn_alignbyte(w2, w1, j + 1);
        hsum[j] = __builtin_amdgcn_udot4(b, kGaussHi, __builtin_amdgcn_udot4(a, kGaussLo, 0u, false), false);
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void gauss7_kernel(PyramidView src, PyramidView dst, BlurPlan plan, GaussTaps T) {
    const uint32_t K[7] = {T.k[0], T.k[1], T.k[2], T.k[3], T.k[4], T.k[5], T.k[6]};
    const uint32_t kGaussLo = K[0] | (K[1] << 8) | (K[2] << 16) | (K[3] << 24), kGaussHi = K[4] | (K[5] << 8) | (K[6] << 16);
    int level = 0;
    while (level + 1 < plan.nlevels && (int)blockIdx.x >= plan.block_begin[level + 1]) level++;
    const int rem = blockIdx.x - plan.block_begin[level];
    const int bx = rem % plan.bx_count[level], by = rem / plan.bx_count[level];
    const LevelView sv = src.lv[level], dv = dst.lv[level];
    const int img = blockIdx.y;
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (by * 4 + (threadIdx.x >> 6)) * kGaussRows;
    if (x0 + 16 > sv.w || y0 >= sv.h) return;  // right-border groups belong to gauss7_edge_kernel
    const uint8_t* sb = sv.base + (size_t)img * sv.img_stride;
    uint8_t* db = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride + x0;
    uint32_t acc[7][4];
    uint32_t hs[4];
    // warm-up: input rows 0..5 (image rows y0-3 .. y0+2) open accumulators 0..5
#pragma unroll
    for (int r = 0; r < 6; r++) {
        gauss_row_sums<ALIGNED>(sb, sv.pitch, sv.h, y0 - 3 + r, x0, hs, kGaussLo, kGaussHi);
#pragma unroll
        for (int t = 0; t <= r; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[r - t][j] = (t == 0 ? 0u : acc[r - t][j]) + K[t] * hs[j];
    }
    // steady state: input row r = 6 + 7*it + u completes output row o = r - 6 and opens accumulator r % 7
    for (int it = 0; it < kGaussRows / 7; it++) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int r = 6 + 7 * it + u;
            gauss_row_sums<ALIGNED>(sb, sv.pitch, sv.h, y0 - 3 + r, x0, hs, kGaussLo, kGaussHi);
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const int a = (6 + u - t) % 7;  // == (r - t) % 7
#pragma unroll
                for (int j = 0; j < 4; j++) acc[a][j] = (t == 0 ? 0u : acc[a][j]) + K[t] * hs[j];
            }
            const int o = r - 6, a = u % 7;  // (r - 6) % 7 == u
            if (y0 + o < sv.h) {
                uint32_t packed = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) packed |= min((acc[a][j] + 32768u) >> 16, 255u) << (8 * j);   // saturate_cast (taps summing to 257)
                *reinterpret_cast<uint32_t*>(db + (size_t)(y0 + o) * dv.pitch) = packed;
            }
        }
    }
}

// Aligned fast path of the blur: the same strips, but a lane loads only ITS dword of a row and takes the 4 bytes on
// either side from the neighbouring lanes (DPP wave shifts) — one global load per row instead of three — and the loads
// of the next seven rows are in flight while the current seven are accumulated.  Lanes 0 and 63 of a wave only provide
// halo bytes: a wave stores 62 four-pixel groups per row.
constexpr int kGaussLanesOut = 62;
// One strip of the streaming blur.  EDGE = the column block holds the group at x0 == 0 or the right border (reflect-101
// fix-ups by byte permutation); the common interior blocks are compiled without them (a block-uniform `if` inside the row
// loop is if-converted into per-row v_cndmask work by the compiler, so the two cases are separate instantiations).
// Per row and lane: 2 DPP moves (neighbour dwords), 10 v_dot4 (the 7 taps of the 4 pixels against the three dwords
// {w0,w1,w2} with the kernel shifted inside the constants: no v_alignbyte), 4 v_cvt, 28 exact fp32 FMAs as 14 v_pk_fma_f32,
// and for the output row 4 FMAs + 3 v_perm.  Rounding without a conversion: the accumulator is opened with the +32768 of
// (acc + 32768) >> 16 already in it, and under round-toward-zero fma(acc, 2^-16, 2^23) = 2^23 + floor(acc / 65536) exactly
// (0 <= acc < 2^24: every product and partial sum is an integer below 2^24, all other FMAs are exact in any rounding
// mode): the result byte is the low byte of the float's bit pattern.
template <int ROWS, bool EDGE>
__device__ __forceinline__ void gauss7_strip(const uint8_t* __restrict__ sb, uint8_t* __restrict__ db, int src_pitch, int dst_pitch,
                                             int h, int y0, uint32_t xl, int x0, bool store, uint32_t sel_w1, uint32_t sel_w2a,
                                             uint32_t sel_w2b) {
    auto load_row = [&](int r) {  // input row r of the strip = image row y0 - 3 + r (reflect-101, then clamped)
        int yy = refl101(y0 - 3 + r, h);
        yy = min(max(yy, 0), h - 1);
        return *reinterpret_cast<const uint32_t*>(sb + (uint32_t)yy * (uint32_t)src_pitch + xl);   // a level plane of one image is < 4 GB
    };
    // taps k = {18,34,48,56,48,34,18}; window bytes 0..11 = {w0,w1,w2}; pixel j (byte 4 + j) = sum_t k[t] * B[1 + j + t]
    constexpr uint32_t k0 = 18, k1 = 34, k2 = 48, k3 = 56;
    constexpr uint32_t A0 = (k0 << 8) | (k1 << 16) | (k2 << 24), B0 = k3 | (k2 << 8) | (k1 << 16) | (k0 << 24);                 // j = 0
    constexpr uint32_t A1 = (k0 << 16) | (k1 << 24), B1 = k2 | (k3 << 8) | (k2 << 16) | (k1 << 24), C1 = k0;                    // j = 1
    constexpr uint32_t A2 = (k0 << 24), B2 = k1 | (k2 << 8) | (k3 << 16) | (k2 << 24), C2 = k1 | (k0 << 8);                     // j = 2
    constexpr uint32_t B3 = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), C3 = k2 | (k1 << 8) | (k0 << 16);                         // j = 3
    auto row_sums = [&](uint32_t w1, float hf[4]) {
        // bound_ctrl: lanes without a source (0 for wave_shr, 63 for wave_shl) read 0 — they only provide halo bytes
        uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        if (EDGE) {
            if (x0 == 0) w0 = __builtin_amdgcn_perm(0u, w1, 0x01020300u);  // p[-1..-3] = p[1..3]
            // right border: identity selectors in the lanes that need no fix
            const uint32_t n2 = __builtin_amdgcn_perm(w1, w0, sel_w2a) | __builtin_amdgcn_perm(0u, w2, sel_w2b);
            w1 = __builtin_amdgcn_perm(w1, w0, sel_w1);
            w2 = n2;
        }
        const uint32_t h0 = __builtin_amdgcn_udot4(w1, B0, __builtin_amdgcn_udot4(w0, A0, 0u, false), false);
        const uint32_t h1 = __builtin_amdgcn_udot4(w2, C1, __builtin_amdgcn_udot4(w1, B1, __builtin_amdgcn_udot4(w0, A1, 0u, false), false), false);
        const uint32_t h2 = __builtin_amdgcn_udot4(w2, C2, __builtin_amdgcn_udot4(w1, B2, __builtin_amdgcn_udot4(w0, A2, 0u, false), false), false);
        const uint32_t h3 = __builtin_amdgcn_udot4(w2, C3, __builtin_amdgcn_udot4(w1, B3, 0u, false), false);
        hf[0] = (float)h0; hf[1] = (float)h1; hf[2] = (float)h2; hf[3] = (float)h3;
    };
    constexpr float Kf[7] = {18.f, 34.f, 48.f, 56.f, 48.f, 34.f, 18.f};
    float acc[7][4];
    float hf[4];
    uint32_t warm[6], nxt[7];
#pragma unroll
    for (int r = 0; r < 6; r++) warm[r] = load_row(r);
#pragma unroll
    for (int u = 0; u < 7; u++) nxt[u] = load_row(6 + u);
    // warm-up: input rows 0..5 open accumulators 0..5
#pragma unroll
    for (int r = 0; r < 6; r++) {
        row_sums(warm[r], hf);
#pragma unroll
        for (int t = 0; t <= r; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[r - t][j] = __fmaf_rn(Kf[t], hf[j], t == 0 ? 32768.0f : acc[r - t][j]);
    }
    // steady state: input row r = 6 + 7*it + u completes output row o = r - 6 and opens accumulator r % 7
    for (int it = 0; it < ROWS / 7; it++) {
        if (y0 + 7 * it >= h) break;   // wave-uniform: the last strip of a level ends with the level (4.8 % of all rows otherwise)
        uint32_t cur[7];
#pragma unroll
        for (int u = 0; u < 7; u++) cur[u] = nxt[u];
        if (it + 1 < ROWS / 7) {
#pragma unroll
            for (int u = 0; u < 7; u++) nxt[u] = load_row(6 + 7 * (it + 1) + u);
        }
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int r = 6 + 7 * it + u;
            row_sums(cur[u], hf);
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const int a = (6 + u - t) % 7;  // == (r - t) % 7
#pragma unroll
                for (int j = 0; j < 4; j++) acc[a][j] = __fmaf_rn(Kf[t], hf[j], t == 0 ? 32768.0f : acc[a][j]);
            }
            const int o = r - 6, a = u % 7;  // (r - 6) % 7 == u
            if (y0 + o < h) {   // wave-uniform
                uint32_t q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = __float_as_uint(__fmaf_rn(acc[a][j], 1.0f / 65536.0f, 8388608.0f));   // RTZ (see above)
                const uint32_t packed = __builtin_amdgcn_perm(__builtin_amdgcn_perm(q[3], q[2], 0x0c0c0400u),
                                                              __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0400u), 0x05040100u);
                if (store) *reinterpret_cast<uint32_t*>(db + (uint32_t)(y0 + o) * (uint32_t)dst_pitch + (uint32_t)x0) = packed;
            }
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(256) void gauss7_stream_kernel(PyramidView src, PyramidView dst, BlurPlan plan, int per_xcd,
                                                            int total_blocks) {
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (private L2 each).  A wave-row reads 256 bytes that
    // start 4 bytes before a multiple of 248, i.e. three 128-byte lines of which the outer two are shared with the
    // neighbouring column blocks; giving every XCD one contiguous run of (image, block) pairs keeps those neighbours —
    // and the strips above / below — on one L2 (FETCH_SIZE 334 -> 242 MB per launch).
    const int tile = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= total_blocks) return;
    const int blocks_per_image = plan.block_begin[plan.nlevels];
    const int img = tile / blocks_per_image, blk = tile - img * blocks_per_image;
    int level = 0;
    while (level + 1 < plan.nlevels && blk >= plan.block_begin[level + 1]) level++;
    const int rem = blk - plan.block_begin[level];
    const int bx = rem % plan.bx_count[level], by = rem / plan.bx_count[level];
    const LevelView sv = src.lv[level], dv = dst.lv[level];
    const int lane = threadIdx.x & 63;
    const int x0 = (bx * kGaussLanesOut + lane - 1) * 4;
    // the wave index through readfirstlane: everything derived from y0 (row reflection, row offsets, the row bound of the
    // stores) is then scalar work
    const int y0 = (by * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * ROWS;
    if (y0 >= sv.h) return;  // wave-uniform
    // lanes 0 / 63 and lanes past the row: halo only — except that lane 63 can take the row's last group: columns x0 + 4 .. are
    // all beyond the border then, i.e. reflections out of {w1, w0}, and the missing right neighbour is not needed (a row of
    // 62 k + 1 groups — KITTI's 1241 and 499 pixel levels — takes k blocks instead of k + 1)
    const bool store = lane >= 1 && (lane <= kGaussLanesOut || x0 + 4 >= sv.w) && x0 < sv.w;
    // Right border (reflect-101): a group with x0 + 7 > w needs pixels beyond column w-1.  Their mirror images
    // p[2(w-1) - x] lie at most 3 columns left of w-1, i.e. inside the lane's own 12-byte window {w2,w1,w0} = columns
    // x0-4 .. x0+7, so the fix is a byte permutation of the window that depends on w - x0 only: three selectors per lane,
    // computed once; applied only in the column block that contains the border (block-uniform branch).
    const bool border_block = (bx + 1) * kGaussLanesOut * 4 + 7 > sv.w;  // some stored lane of this block has x0 + 7 > w
    uint32_t sel_w1 = 0x07060504u, sel_w2a = 0x0c0c0c0cu, sel_w2b = 0x03020100u;  // identity: w1 = w1, w2 = w2
    if (border_block && x0 + 7 > sv.w && x0 < sv.w) {
        sel_w1 = 0; sel_w2a = 0; sel_w2b = 0;
        for (int b = 0; b < 8; b++) {
            const int x = x0 + b;
            int idx = b + 4;                                        // window index of column x (x0-4 -> 0)
            if (x >= sv.w) idx = max(2 * (sv.w - 1) - x, x0 - 4) - x0 + 4;
            if (b < 4) sel_w1 |= (uint32_t)idx << (8 * b);          // sources of w1 lie in {w1,w0}: index 0..7
            else {
                sel_w2a |= (uint32_t)(idx < 8 ? idx : 0x0c) << (8 * (b - 4));       // from {w1,w0}
                sel_w2b |= (uint32_t)(idx >= 8 ? idx - 8 : 0x0c) << (8 * (b - 4));  // from w2
            }
        }
    }
    const uint32_t xl = (uint32_t)min(max(x0, 0), sv.pitch - 4);
    const uint8_t* sb = sv.base + (size_t)img * sv.img_stride;
    uint8_t* db = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;
    __builtin_amdgcn_s_setreg(0x801 /* hwreg(HW_REG_MODE, 0, 2): FP32 rounding */, 3 /* toward zero */);
    if (border_block || bx == 0)   // block-uniform
        gauss7_strip<ROWS, true>(sb, db, sv.pitch, dv.pitch, sv.h, y0, xl, x0, store, sel_w1, sel_w2a, sel_w2b);
    else
        gauss7_strip<ROWS, false>(sb, db, sv.pitch, dv.pitch, sv.h, y0, xl, x0, store, sel_w1, sel_w2a, sel_w2b);
}

// Right-border column groups (x0 + 16 > w: at most four groups = 16 columns per row): per-byte reflect-101 gather.
// One 64-thread block covers 58 output rows: thread r first forms the horizontal sums of input row y0-3+r for the
// border columns (all rows in parallel: one memory round trip), then thread r < 58 finishes output row y0+r from LDS.
constexpr int kEdgeRows = 58;
__global__ __launch_bounds__(64) void gauss7_edge_kernel(PyramidView src, PyramidView dst, GaussTaps T) {
    __shared__ uint16_t hs[64][16];
    const int level = blockIdx.y, img = blockIdx.z;
    const LevelView sv = src.lv[level], dv = dst.lv[level];
    const int first = sv.w >= 16 ? ((sv.w - 16) / 4 + 1) * 4 : 0;  // first x0 with x0 + 16 > w
    const int ncol = sv.w - first;                                  // 1..16 border columns
    const int y0 = blockIdx.x * kEdgeRows;
    if (y0 >= sv.h) return;
    const int r = threadIdx.x;
    const uint8_t* sb = sv.base + (size_t)img * sv.img_stride;
    uint8_t* db = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;
    const int K[7] = {(int)T.k[0], (int)T.k[1], (int)T.k[2], (int)T.k[3], (int)T.k[4], (int)T.k[5], (int)T.k[6]};
    {
        int yy = refl101(y0 - 3 + r, sv.h);
        yy = min(max(yy, 0), sv.h - 1);
        const uint8_t* rp = sb + (size_t)yy * sv.pitch;
        uint32_t px[22];
#pragma unroll
        for (int o = 0; o < 22; o++) px[o] = rp[min(max(refl101(first - 3 + o, sv.w), 0), sv.w - 1)];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t a = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) a += K[k] * px[j + k];
            hs[r][j] = (uint16_t)a;
        }
    }
    __syncthreads();
    if (r < kEdgeRows && y0 + r < sv.h) {
        for (int j = 0; j < ncol; j++) {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) acc += (uint32_t)K[k] * hs[r + k][j];
            db[(size_t)(y0 + r) * dv.pitch + first + j] = (uint8_t)min((acc + 32768u) >> 16, 255u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: one wave64 per selected keypoint, 4 keypoints per block.
// ------------------------------------------------------------------------------------------------
struct PatchTables {
    int8_t pattern[256 * 4];  // (x0,y0,x1,y1) per pair
    int8_t umax[16];          // half-width of the circular patch per |v| (ORBextractor.cc:447-468)
};
__constant__ PatchTables c_tab;

void upload_patch_tables(const int8_t* pattern, const int* umax, hipStream_t stream) {
    PatchTables t;
    for (int i = 0; i < 1024; i++) t.pattern[i] = pattern[i];
    for (int v = 0; v <= kHalfPatch; v++) t.umax[v] = (int8_t)umax[v];
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tab), &t, sizeof(t), 0, hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);
}

// cv::fastAtan2 — separate multiply/add, no contraction (oracle/cvprims.h fast_atan2); fma != 0: the Horner steps contracted
// (Semantics::atan2_fma).
__device__ __forceinline__ float fast_atan2_deg(float y, float x, int fma) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    auto poly = [&](float cc, float cc2) {
        if (fma) return __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, cc2, p5), cc2, p3), cc2, p1), cc);
        return __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, cc2), p5), cc2), p3), cc2), p1), cc);
    };
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = poly(c, c2);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, poly(c, c2));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Sum over the 64 lanes with DPP adds only (no LDS crossbar): xor-butterfly inside each row of 16 lanes, then the row
// totals are chained through lanes 15 / 31 into row 3; lane 63 holds the total, returned as a scalar.
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140 /* row_mirror */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}

// cvRound for |x| < 2^22 as one fp32 add and one integer subtract (both fast-class VALU ops; v_rndne_f32 + v_cvt_i32_f32
// are two slow-class ones): adding 1.5 * 2^23 leaves round-to-nearest-even of x in the low mantissa bits.
__device__ __forceinline__ int rint_small(float x) { return __float_as_int(__fadd_rn(x, 12582912.0f)) - 0x4B400000; }

constexpr int kPatchPitch = 44;  // bytes per staged patch row: 11 dwords
constexpr int kKpPerWave = 4;  // keypoints handled back to back by one wave (amortises the per-lane table loads)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void describe_kernel(PyramidView pyr, PyramidView blur, const SelRec* __restrict__ sel,
                                                       const int* __restrict__ sel_count, int sel_stride,
                                                       LevelScale scales, msorb_keypoint* __restrict__ kps,
                                                       uint8_t* __restrict__ desc, int out_stride, int atan2_fma) {
    __shared__ __attribute__((aligned(16))) uint8_t patch[4 * kKpPerWave][37 * kPatchPitch + 4];  // one slot per (wave, keypoint)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; give every image to ONE XCD so that the
    // overlapping keypoint patches of an image are served by a single L2 instead of being fetched by all eight.
    int img = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7u) == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        img = (int)((j / gridDim.x) * 8 + xcd);
        bx = (int)(j % gridDim.x);
    }
    const int lane = threadIdx.x & 63;
    const int k_first = __builtin_amdgcn_readfirstlane((bx * 4 + (int)(threadIdx.x >> 6)) * kKpPerWave);  // wave-uniform -> SALU
    const int n_sel = sel_count[img];
    if (k_first >= n_sel) return;
    // per-lane constants, loaded once.  IC-angle patch: the 31 rows are read as 9 aligned dwords each (279 slots, slot =
    // lane + 64 t); after the byte re-alignment below, slot (row, col < 8) holds the pixels u = 4 col - 15 .. 4 col - 12 of
    // row v = row - 15.  wu = (u + 16) per byte inside the circle (0 outside), vm = 1 / 0: two udot4 give sum(u I), sum(I).
    uint32_t wu[5], vm[5];
    uint32_t rc[5], bc[6];  // slot -> row | byte column << 8 (lane constants; keeps the /9, /10 out of the keypoint loop)
#pragma unroll
    for (int it = 0; it < 6; it++) {  // slots past row 36 repeat a slot of row 36 (same address, same value: no predication)
        const int idx = lane + 64 * it;
        bc[it] = (uint32_t)min(idx / 10, 36) | ((uint32_t)(4 * (idx % 10)) << 8);
    }
    uint32_t vrow03 = 0;  // v of slots 0..3, one signed byte each
    int vrow4 = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) {
        // slot layout: a load instruction covers 7 whole rows of 9 dwords in lanes 0..62 (lane 63 idles), so "the next
        // dword of the row" is always the next lane of the same register: 5 x 7 = 35 >= 31 rows in the same 5 loads
        const int row = lane < 63 ? 7 * t + lane / 9 : 31, col = lane % 9;
        uint32_t a = 0, m = 0;
        if (row < 31 && col < 8) {
            const int d = c_tab.umax[row < 15 ? 15 - row : row - 15];
            for (int bb = 0; bb < 4; bb++) {
                const int u = 4 * col + bb - 15;
                if (u >= -d && u <= d) { a |= (uint32_t)(u + 16) << (8 * bb); m |= 1u << (8 * bb); }
            }
        }
        wu[t] = a; vm[t] = m;
        rc[t] = (uint32_t)min(row, 30) | ((uint32_t)(4 * col + 4) << 8);  // slots of no row: weights 0, any valid address
        if (t < 4) vrow03 |= (uint32_t)((row - 15) & 255) << (8 * t);
        else vrow4 = row - 15;
    }
    // the 4 pattern pairs of this lane as floats (lane constants: decoded once per wave, not once per keypoint)
    float patx0[4], paty0[4], patx1[4], paty1[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t pw = *reinterpret_cast<const uint32_t*>(&c_tab.pattern[(w * 64 + lane) * 4]);
        patx0[w] = (float)(int8_t)(pw & 255u); paty0[w] = (float)(int8_t)((pw >> 8) & 255u);
        patx1[w] = (float)(int8_t)((pw >> 16) & 255u); paty1[w] = (float)(int8_t)(pw >> 24);
    }
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.0);

    // Software pipeline over the wave's keypoints: while keypoint k is being processed the 11 patch loads of keypoint
    // k+1 are already in flight (and the record of k+2 is being fetched) — the kernel is bound by the latency of these
    // scattered loads, not by arithmetic.
    struct Loads { uint32_t rp[5], rsh[5], bp[6]; int poff; };
    auto scalar_rec = [](const SelRec& v) {
        // every lane loaded the same record: move it to scalar registers so that everything derived from it (level view,
        // row pointers, strides) is SALU work and the loads use an SGPR base + 32-bit lane offset
        SelRec r;
        uint32_t w[3];
        memcpy(w, &v, sizeof(w));
        for (int i = 0; i < 3; i++) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
        memcpy(&r, w, sizeof(w));
        return r;
    };
    auto issue = [&](const SelRec& r, Loads& L) {
        // Both patches depend only on (x, y, level).  The 256 test pairs gather 512 bytes from the blurred 37x37
        // neighbourhood (|offset| <= 18 after rotation); a direct gather touches ~35 cache lines per load instruction,
        // so that patch goes to LDS with row-coherent dword loads: 37 rows x 10 dwords from the 4-byte boundary below
        // x-18 = 370 slots, slot = lane + 64 it.  The 31x31 IC-angle patch: 31 rows x 9 dwords, slot = lane + 64 t.
        const LevelView lv = pyr.lv[r.level];
        const LevelView bv = blur.lv[r.level];
        const int px0 = (r.x - 18) & ~3;
        L.poff = (r.x - 18) - px0;
        const uint8_t* brow = bv.base + (size_t)img * bv.img_stride + (size_t)(r.y - 18) * bv.pitch + px0;
        // (level 0 may be the caller's own image with any row stride: the 4-byte phase is taken per row)
        // addresses = scalar base + 32-bit lane offset (the global_load saddr form: no 64-bit VALU address math)
        const uint8_t* rrow = lv.base + (size_t)img * lv.img_stride + (size_t)(r.y - 15) * lv.pitch + (r.x - 15) - 4;
        const uint32_t rlow = (uint32_t)reinterpret_cast<uintptr_t>(rrow);
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const uint32_t row = rc[t] & 255u, col4p4 = rc[t] >> 8;
#ifdef MSORB_DESC_EXP_RAW_FLAT    // timing experiment only (wrong results): every IC-angle row from one line
            const uint32_t o = 0u * row;
#else
            const uint32_t o = __umul24(row, (uint32_t)lv.pitch);  // full-rate 24-bit multiply
#endif
            L.rsh[t] = (rlow + o) & 3u;
            L.rp[t] = *reinterpret_cast<const uint32_t*>(rrow + (size_t)(o + col4p4 - L.rsh[t]));
        }
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t row = bc[it] & 255u, col4 = bc[it] >> 8;
#ifdef MSORB_DESC_EXP_BLUR_FLAT   // timing experiment only (wrong results): every blurred row from one line
            L.bp[it] = *reinterpret_cast<const uint32_t*>(brow + (size_t)(col4));
            (void)row;
#elif defined(MSORB_DESC_EXP_BLUR_TILED)   // timing experiment only (wrong results): addresses of a 16 x 8-pixel tiled plane
            {
                const uint32_t yy = min((uint32_t)(r.y - 18) + row, (uint32_t)((bv.h & ~7) - 1)), xx = (uint32_t)px0 + col4;
                const uint32_t off = (yy >> 3) * (uint32_t)(bv.pitch * 8) + (xx >> 4) * 128u + (yy & 7u) * 16u + (xx & 15u);
                L.bp[it] = *reinterpret_cast<const uint32_t*>(bv.base + (size_t)img * bv.img_stride + off);
            }
#else
            L.bp[it] = *reinterpret_cast<const uint32_t*>(brow + (size_t)(__umul24(row, (uint32_t)bv.pitch) + col4));
#endif
        }
    };
    const SelRec* recs = sel + (size_t)img * sel_stride;
    SelRec r_cur = scalar_rec(recs[k_first]);
    Loads L_next;
    issue(r_cur, L_next);
    SelRec r_pre = recs[min(k_first + 1, n_sel - 1)];
    // Phase A, per keypoint: moments of the IC-angle patch (wave-uniform totals) and the blurred patch into the keypoint's
    // own LDS slot.  Phase V, once per wave: angle, cos, sin of all four keypoints at once — lane kk computes keypoint kk,
    // so the atan2 polynomial and the double-precision sincos run once per wave instead of once per keypoint.
    // Phase B, per keypoint: steered BRIEF from its LDS slot.
    // The last wave of an image repeats the image's last keypoint in its unused places (no exit inside the sequence: the
    // four keypoints are one straight line of code, so the pre-issued loads stay in flight across them; an early exit made
    // the compiler copy them at the merge points, i.e. wait for them); only the stores are conditional.
    SelRec R[kKpPerWave];
    int M10[kKpPerWave], M01[kKpPerWave], POFF[kKpPerWave];
    uint8_t* const lp0 = patch[(threadIdx.x >> 6) * kKpPerWave];
#pragma unroll
    for (int kk = 0; kk < kKpPerWave; kk++) {
        const int k = k_first + kk;
        R[kk] = r_cur;
        const Loads L = L_next;
        if (kk + 1 < kKpPerWave) {
            r_cur = scalar_rec(r_pre);
            issue(r_cur, L_next);
            r_pre = recs[min(k + 2, n_sel - 1)];
        }
        const uint32_t* rp = L.rp;
        const uint32_t* rsh = L.rsh;
        const uint32_t* bp = L.bp;
        POFF[kk] = L.poff;

        // IC_Angle: integer moments over the 749-pixel circular patch (un-blurred level).  The next lane holds the
        // following 4 bytes of the row: alignbyte undoes the 4-byte
        // alignment of the loads, so the per-lane weights do not depend on the keypoint.
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rp[t], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            const uint32_t px = __builtin_amdgcn_alignbyte(nx, rp[t], rsh[t]);
            const int sI = (int)__builtin_amdgcn_udot4(px, vm[t], 0u, false);
            m10 += (int)__builtin_amdgcn_udot4(px, wu[t], 0u, false) - 16 * sI;
            m01 += (t < 4 ? (int)(int8_t)(vrow03 >> (8 * t)) : vrow4) * sI;
        }
        M10[kk] = wave_sum_dpp(m10);  // wave-uniform (SGPR) totals
        M01[kk] = wave_sum_dpp(m01);
        uint8_t* lp = lp0 + kk * (37 * kPatchPitch + 4);
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t row = bc[it] & 255u, col4 = bc[it] >> 8;
            *reinterpret_cast<uint32_t*>(lp + row * kPatchPitch + col4) = bp[it];
        }
    }
    // Phase V
    int m10v = M10[0], m01v = M01[0];
#pragma unroll
    for (int kk = 1; kk < kKpPerWave; kk++)
        if (lane == kk) { m10v = M10[kk]; m01v = M01[kk]; }
    const float angle_v = fast_atan2_deg((float)m01v, (float)m10v, atan2_fma);
    float a_v, b_v;
    glibc_sincosf<true>(__fmul_rn(angle_v, factor_pi), &b_v, &a_v);  // a = cos, b = sin (ORBextractor.cc:112)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Phase B
#pragma unroll
    for (int kk = 0; kk < kKpPerWave; kk++) {
        const SelRec r = R[kk];
        const float angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_v), kk));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_v), kk));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_v), kk));
        const uint8_t* bc = lp0 + kk * (37 * kPatchPitch + 4) + 18 * kPatchPitch + 18 + POFF[kk];
        unsigned long long word[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float x0 = patx0[w], y0 = paty0[w], x1 = patx1[w], y1 = paty1[w];
            // cvRound(x*b + y*a), cvRound(x*a - y*b) with the contraction order of oracle/orb_extractor_oracle.cc
            const int r0 = rint_small(__fmaf_rn(x0, b, __fmul_rn(y0, a)));
            const int q0 = rint_small(__fmaf_rn(x0, a, -__fmul_rn(y0, b)));
            const int r1 = rint_small(__fmaf_rn(x1, b, __fmul_rn(y1, a)));
            const int q1 = rint_small(__fmaf_rn(x1, a, -__fmul_rn(y1, b)));
            const int t0 = bc[r0 * kPatchPitch + q0];
            const int t1 = bc[r1 * kPatchPitch + q1];
            word[w] = __ballot(t0 < t1);
        }
        if (k_first + kk >= n_sel) break;  // wave-uniform; nothing but the stores is left
        if (lane < 4) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)img * out_stride + r.dst) * 32);
            d[lane] = lane == 0 ? word[0] : lane == 1 ? word[1] : lane == 2 ? word[2] : word[3];
        }
        if (lane == 0) {
            msorb_keypoint kp;
            const float sc = scales.scale[r.level];
            kp.x = r.level ? __fmul_rn((float)r.x, sc) : (float)r.x;   // keypoint->pt *= scale (ORBextractor.cc:1149-1151)
            kp.y = r.level ? __fmul_rn((float)r.y, sc) : (float)r.y;
            kp.size = scales.patch[r.level];
            kp.angle = angle;
            kp.response = (float)r.score;
            kp.octave = r.level;
            kp.class_id = -1;
            kps[(size_t)img * out_stride + r.dst] = kp;
        }
    }
}

// ---- launch wrappers (called from extractor.hip) --------------------------------------------------
// Level 0 of a batch whose rows are not 4-byte aligned (e.g. tightly packed 1241-pixel rows): copied once into the
// handle's aligned level-0 planes, so that every later kernel takes its aligned variant (the byte-granular variants of
// FAST / blur / pyramid are 1.7-2x slower than this copy costs).
__global__ __launch_bounds__(256) void stage_level0_kernel(const uint8_t* __restrict__ src, size_t row_stride, size_t image_stride,
                                                           uint8_t* __restrict__ dst, int dst_pitch, size_t dst_image_stride,
                                                           int cols) {
    const int g = blockIdx.x * 256 + threadIdx.x;  // 4-pixel group of the row
    const int x = 4 * g;
    if (x >= cols) return;
    const uint8_t* s = src + (size_t)blockIdx.z * image_stride + (size_t)blockIdx.y * row_stride + x;
    uint32_t v = s[0];
    if (x + 1 < cols) v |= (uint32_t)s[1] << 8;
    if (x + 2 < cols) v |= (uint32_t)s[2] << 16;
    if (x + 3 < cols) v |= (uint32_t)s[3] << 24;
    *reinterpret_cast<uint32_t*>(dst + (size_t)blockIdx.z * dst_image_stride + (size_t)blockIdx.y * dst_pitch + x) = v;
}
void launch_stage_level0(const LevelView& src, uint8_t* dst, int dst_pitch, size_t dst_image_stride, int n_images, hipStream_t s) {
    hipLaunchKernelGGL(stage_level0_kernel, dim3((src.w + 1023) / 1024, src.h, n_images), dim3(256), 0, s, src.base, (size_t)src.pitch,
                       src.img_stride, dst, dst_pitch, dst_image_stride, src.w);
}
void launch_pyr_resize(const LevelView& src, const LevelView& dst, uint8_t* dst_base, const ResizeTap* tx,
                       const ResizeTap* ty, int n_images, hipStream_t s, int single_stage) {
    dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, n_images);
    if (single_stage) {   // Semantics::resize_single_stage: served by the generic kernel only
        hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty, 1);
        return;
    }
    // aligned variant: source rows start on 4-byte boundaries and may be read up to the next multiple of 4 past w
    const bool aligned = (reinterpret_cast<uintptr_t>(src.base) & 3) == 0 && (src.pitch & 3) == 0 &&
                         (src.img_stride & 3) == 0 && src.pitch >= ((src.w + 3) & ~3) + 8 &&
                         (reinterpret_cast<uintptr_t>(tx) & 15) == 0;
    constexpr int R = 8;  // rows per wave
    // test aids: MSORB_PYR_SINGLE -> one-row kernels, MSORB_PYR_ROWS -> row-streaming kernel, MSORB_PYR_DMA -> LDS-DMA band kernel
    // (default = the register band kernel: 0.168 against 0.180 ms per 256 KITTI images for the LDS-DMA form once both share
    // pyr_band_tile's arithmetic — and 25 instead of 40 KB of LDS per workgroup beside the other batch's kernels)
    const bool rows_env = !getenv("MSORB_PYR_SINGLE");
    const bool band_env = !getenv("MSORB_PYR_ROWS");
    const bool dma_env = getenv("MSORB_PYR_DMA") != nullptr && !getenv("MSORB_PYR_BAND");
    // the band kernels park the source rows of a band of R output rows: at most floor((R - 1) * scale) + 3 of them (<= 12),
    // and decode the 4 taps of a lane out of an 8-byte window (horizontal scale <= 1.25)
    const double sy = (double)src.h / (double)dst.h, sx = (double)src.w / (double)dst.w;
    const bool band_ok = aligned && rows_env && band_env && n_images >= 16 && (int)std::floor((R - 1) * sy) + 3 <= 12 && sx <= 1.25;
    // the LDS-DMA kernel stages 336-byte row segments as 16-byte granules: 16-byte aligned rows, 64 windows within 336 bytes
    const bool dma_ok = band_ok && dma_env && (reinterpret_cast<uintptr_t>(src.base) & 15) == 0 && (src.pitch & 15) == 0 &&
                        (src.img_stride & 15) == 0 && src.pitch >= 336 && 252.0 * sx + 28.0 <= 336.0;
    const dim3 band_grid((dst.w + 255) / 256, (dst.h + 4 * R - 1) / (4 * R), n_images);
    if (dma_ok) hipLaunchKernelGGL(pyr_resize_dma_kernel<R>, band_grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
    else if (band_ok) {
        // two waves (12 KB of LDS) per workgroup: beside the other batch's FAST / quadtree / describe, which fill a CU's LDS
        // to within 5-12 KB, small workgroups find room sooner (1.310 against 1.324 ms per step with 4 waves, 1.336 with 1)
        static const int bw = getenv("MSORB_PYR_BAND_WAVES") ? atoi(getenv("MSORB_PYR_BAND_WAVES")) : 2;
        const bool lds_form = getenv("MSORB_PYR_BAND_LDS") != nullptr;   // the round-2 form (rows parked in LDS); test aid
        if (!lds_form) {
            if (bw == 1) hipLaunchKernelGGL((pyr_resize_bandreg_kernel<R, 12, 1>), dim3((dst.w + 255) / 256, (dst.h + R - 1) / R, n_images), dim3(64), 0, s, src, dst, dst_base, tx, ty);
            else if (bw == 4) hipLaunchKernelGGL((pyr_resize_bandreg_kernel<R, 12, 4>), band_grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
            else hipLaunchKernelGGL((pyr_resize_bandreg_kernel<R, 12, 2>), dim3((dst.w + 255) / 256, (dst.h + 2 * R - 1) / (2 * R), n_images), dim3(128), 0, s, src, dst, dst_base, tx, ty);
        }
        else if (bw == 1) hipLaunchKernelGGL((pyr_resize_band_kernel<R, 12, 1>), dim3((dst.w + 255) / 256, (dst.h + R - 1) / R, n_images), dim3(64), 0, s, src, dst, dst_base, tx, ty);
        else if (bw == 2) hipLaunchKernelGGL((pyr_resize_band_kernel<R, 12, 2>), dim3((dst.w + 255) / 256, (dst.h + 2 * R - 1) / (2 * R), n_images), dim3(128), 0, s, src, dst, dst_base, tx, ty);
        else hipLaunchKernelGGL((pyr_resize_band_kernel<R, 12, 4>), band_grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
    }
    else if (aligned && rows_env && n_images >= 16)
        hipLaunchKernelGGL(pyr_resize_rows_kernel<R>, dim3((dst.w + 255) / 256, (dst.h + 4 * R - 1) / (4 * R), n_images), dim3(256),
                           0, s, src, dst, dst_base, tx, ty);
    else if (aligned) hipLaunchKernelGGL(pyr_resize_aligned_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
    else hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty, 0);
}
// ComputePyramid for a batch: levels 1 .. n-1, each from the one above (ORBextractor.cc:1179-1193), one launch per level.
// MSORB_PYR_TAIL=1 puts the last up to three levels into one launch (pyr_resize_tail_kernel) — measured on MI355X, 256 KITTI
// images: 33 us instead of 42 us for the three launches with the stage alone on the GPU, but the whole step gets slower
// (1.366 vs 1.352 ms): a 16-wave workgroup with 98 KB of LDS per CU keeps the other stream's kernels out.  Off by default.
void launch_pyramid(const PyramidView& pyr, const ResizeTap* taps, const size_t* tap_x_off, const size_t* tap_y_off, int n_images,
                    hipStream_t s, const Semantics& sem) {
    const int nl = pyr.nlevels;
    constexpr int R = 8;
    int tail = 0;
    if (!sem.resize_single_stage && n_images >= 64 && getenv("MSORB_PYR_TAIL") && !getenv("MSORB_PYR_SINGLE") && !getenv("MSORB_PYR_ROWS")) {
        while (tail < 3 && nl - 1 - tail >= 2) {   // at least level 1 stays a launch of its own
            const int l = nl - 1 - tail;
            const LevelView &src = pyr.lv[l - 1], &dst = pyr.lv[l];
            const ResizeTap* tx = taps + tap_x_off[l];
            const bool aligned = (reinterpret_cast<uintptr_t>(src.base) & 3) == 0 && (src.pitch & 3) == 0 && (src.img_stride & 3) == 0 &&
                                 src.pitch >= ((src.w + 3) & ~3) + 8 && (reinterpret_cast<uintptr_t>(tx) & 15) == 0;
            const double sy = (double)src.h / (double)dst.h, sx = (double)src.w / (double)dst.w;
            if (!aligned || (int)std::floor((R - 1) * sy) + 3 > 12 || sx > 1.25 || (size_t)dst.w * dst.h > 100000) break;
            tail++;
        }
    }
    for (int l = 1; l < nl - tail; l++)
        launch_pyr_resize(pyr.lv[l - 1], pyr.lv[l], const_cast<uint8_t*>(pyr.lv[l].base), taps + tap_x_off[l], taps + tap_y_off[l], n_images, s,
                          sem.resize_single_stage);
    if (tail) {
        PyrTailArgs A{};
        A.nl = tail;
        const int first = nl - tail;
        A.lv[0] = pyr.lv[first - 1];
        for (int i = 0; i < tail; i++) {
            A.lv[i + 1] = pyr.lv[first + i];
            A.tx[i] = taps + tap_x_off[first + i];
            A.ty[i] = taps + tap_y_off[first + i];
        }
        hipLaunchKernelGGL((pyr_resize_tail_kernel<R, 12>), dim3(n_images), dim3(1024), 0, s, A);
    }
}
void launch_fast_cells(const PyramidView& pyr, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                       int slots_per_image, Cand16* slots, int* cell_count, int n_images, bool small_cells, hipStream_t s) {
    bool aligned = true;
    for (int l = 0; l < pyr.nlevels; l++) {
        const LevelView& v = pyr.lv[l];
        aligned = aligned && (reinterpret_cast<uintptr_t>(v.base) & 3) == 0 && (v.pitch & 3) == 0 && (v.img_stride & 3) == 0;
    }
    static const int dbg = getenv("MSORB_FAST_DEBUG_STOP") ? atoi(getenv("MSORB_FAST_DEBUG_STOP")) : 0;  // profiling only
    const dim3 grid(n_cells, n_images);
    // q = mulhi(n, gx_magic) == n / n_cells for every n < n_cells * n_images (checked here, once per shape)
    uint32_t gx_magic = (uint32_t)((0x100000000ull + (unsigned)n_cells - 1) / (unsigned)n_cells);
    {
        const unsigned long long total = (unsigned long long)n_cells * (unsigned)n_images;
        const unsigned long long e = (unsigned long long)gx_magic * (unsigned)n_cells - 0x100000000ull;  // < n_cells
        if (n_cells < 2 || total >= 0x100000000ull || e * total >= 0x100000000ull) gx_magic = 0;  // kernel divides instead
    }
#define MSORB_FAST_LAUNCH(AL, GEO)                                                                                        \
    hipLaunchKernelGGL((fast_cells_kernel<AL, GEO>), grid, dim3(GEO::kThreads), 0, s, pyr, cells, ini_th, min_th, slots_per_image, slots, \
                       cell_count, n_cells, gx_magic, dbg)
    if (small_cells) { if (aligned) MSORB_FAST_LAUNCH(true, GeoSmall); else MSORB_FAST_LAUNCH(false, GeoSmall); }
    else { if (aligned) MSORB_FAST_LAUNCH(true, GeoLarge); else MSORB_FAST_LAUNCH(false, GeoLarge); }
#undef MSORB_FAST_LAUNCH
}
bool launch_fast_strips(const PyramidView& pyr, const StripDesc* strips, int n_strips, int n_small, const int* max_rh, const int* work_cap,
                        int n_cells, int ini_th, int min_th, int slots_per_image, Cand16* slots, int* cell_count, int n_images,
                        hipStream_t s) {
    for (int l = 0; l < pyr.nlevels; l++) {
        const LevelView& v = pyr.lv[l];
        if ((reinterpret_cast<uintptr_t>(v.base) & 3) != 0 || (v.pitch & 3) != 0 || (v.img_stride & 3) != 0) return false;  // dword staging
    }
    if (n_strips < 1) return false;
    static const int dbg = getenv("MSORB_FAST_DEBUG_STOP") ? atoi(getenv("MSORB_FAST_DEBUG_STOP")) : 0;  // profiling only
    static const int cap_env = getenv("MSORB_STRIP_CAP") ? atoi(getenv("MSORB_STRIP_CAP")) : 0;          // tuning only
    StripLds L[2];
    const int count[2] = {n_small, n_strips - n_small};
    for (int c = 0; c < 2; c++) {
        if (count[c] <= 0) continue;
        L[c] = strip_lds_layout<kStripCells>(max_rh[c], std::max(work_cap[c], cap_env));
        if (L[c].total > 60 * 1024 || L[c].work_cap > 8192) return false;
    }
    // two launches, one per LDS class (orb_host.cc): first the bulk with the small footprint, then the few tall strips
    for (int c = 0, first = 0; c < 2; first += count[c], c++) {
        const int n = count[c];
        if (n <= 0) continue;
        uint32_t gx_magic = (uint32_t)((0x100000000ull + (unsigned)n - 1) / (unsigned)n);
        {
            const unsigned long long total = (unsigned long long)n * (unsigned)n_images;
            const unsigned long long e = (unsigned long long)gx_magic * (unsigned)n - 0x100000000ull;
            if (n < 2 || total >= 0x100000000ull || e * total >= 0x100000000ull) gx_magic = 0;
        }
        hipLaunchKernelGGL((fast_strip_kernel<kStripCells, kStripThreads>), dim3(n, n_images), dim3(kStripThreads), (size_t)L[c].total, s, pyr,
                           strips + first, n, ini_th, min_th, slots_per_image, slots, cell_count, n_cells, gx_magic, L[c], dbg);
    }
    return true;
}
void launch_cand_compact(const CellDesc* cells, int n_cells, const int* level_cell_begin, int nlevels,
                         int slots_per_image, const Cand16* slots, const int* cell_count, int* cell_off,
                         int* level_count, int* img_total, int* img_base, Cand16* compact, int n_images,
                         hipStream_t s) {
    hipLaunchKernelGGL(cand_scan_cells_kernel, dim3(n_images), dim3(256), 0, s, cell_count, n_cells, level_cell_begin,
                       nlevels, cell_off, level_count, img_total);
    hipLaunchKernelGGL(cand_scan_images_kernel, dim3(1), dim3(256), 0, s, img_total, n_images, img_base);
    hipLaunchKernelGGL(cand_gather_kernel, dim3((n_cells + kGatherCells - 1) / kGatherCells, n_images), dim3(256), 0, s, cells, n_cells, slots_per_image,
                       slots, cell_count, cell_off, img_base, compact);
}
// ------------------------------------------------------------------------------------------------
// The same 7 x 7 Gaussian on the matrix cores — an OPTION (`MSORB_BLUR_MFMA=1`), bit-equal to the VALU kernels, kept for what
// it measured: a separable blur is two banded matrix products, the chain is bound by VALU issue and nothing in the front-end
// uses the MFMA pipe.  The kernel needs 58 M VALU instructions per 256 images instead of gauss7_stream_kernel's 99 M and
// 0.035 ms of matrix-pipe time — and 0.32 ms alone on the GPU against 0.18, 1.41 against 1.29 ms per pipelined step: it reads
// every pixel three times (64 source columns per 32 outputs, strips of one image spread over the XCDs: 1.13 GB FETCH against
// 0.45) and its waves are long dependent chains (load -> MFMA -> byte planes -> MFMA -> LDS turn -> store, with loads and stores
// sharing the in-order vmcnt).  Two findings on the way: the fp32 MFMAs (v_mfma_f32_32x32x2_f32, 142 TFLOP/s measured,
// tools/mfma_rate_f32.hip) run at exactly the vector ALUs' FMA rate and did not overlap with other waves' VALU work — the first,
// fp32 form of the vertical pass cost more than the kernel it replaced —, and 16-byte stores of 32-byte row pieces from four
// waves at four times cost 0.22 ms against whole 128-byte lines from one workgroup.
//   horizontal pass  H = P x Bh :  v_mfma_i32_32x32x32_i8 x 2 (K = 64 source columns X-16 .. X+47 for the 32 output columns
//       X .. X+31).  A operand = pixels: lane (row r = lane & 31, half = lane >> 5) loads 16 consecutive bytes of its row for
//       each K step, xor 0x80 makes them signed (p - 128); B operand = the band matrix, built once per wave: byte j of
//       lane (column n, half) is the weight of source column x for output column X + n — tap(x - (X + n) + 3) plus, in the first /
//       last strip, the taps that BORDER_REFLECT_101 folds back onto x; columns that do not exist weigh 0.  Both operands use the
//       same lane -> (row / column, k slot) rule, so the order of k inside the instruction does not matter.
//   vertical pass  V = Bv x H :  i8 again (the fp32 MFMAs run on the vector ALUs' own FMA units — 142 TFLOP/s, the VALU rate —,
//       so a fp32 form of this pass, built first, cost more vector time than the kernel it replaced; the i8 / bf16 forms run on
//       the matrix cores proper).  H is 16 bits wide: it is fed as two byte planes, hi = H >> 8 and lo = H & 255, two MFMAs each
//       (tile j, and rows 0 .. 6 of tile j + 1), and the planes are recombined as 256 * acc_hi + acc_lo.  The H tile sits in the
//       accumulator layout — lane (column n, half) holds rows 4 half + (i & 3) + 8 (i >> 2) in register i — and the byte planes
//       keep that order: byte 4 q + b of the operand = register 4 q + b, with the weights (a per-lane constant) in the same
//       order.  An output block covers rows 32 j + 4 .. 32 j + 35, so it needs tile j and tile j + 1 and no tile above.
//   Rows are reflected by LOADING the reflected source row (the horizontal pass is row independent), so no block has special
//   weights; out = (V + 32768) >> 16 (saturated when the taps sum to more than 256).
//   One wave = one 32-column strip x plan.chunk output blocks; 1 KB of LDS per wave (output transpose), no workgroup barrier.
// ------------------------------------------------------------------------------------------------
typedef int bm_v4i __attribute__((ext_vector_type(4)));
typedef int bm_v16i __attribute__((ext_vector_type(16)));
typedef float bm_v16f __attribute__((ext_vector_type(16)));
struct BlurMfmaPlan {
    int task_begin[kMaxLevels + 1];  // first task of each level (per image)
    int strips[kMaxLevels];          // groups of four 32-column strips of the level
    int nblocks[kMaxLevels];         // output blocks j = -1 .. nblocks - 2 of the level
    int nlevels;
    int chunk;                       // output blocks (32 rows each) per wave
};

__device__ __forceinline__ uint32_t bm_tap(int t, unsigned long long K) {   // k[t] for 0 <= t <= 6, else 0
    return (t >= 0 && t <= 6) ? (uint32_t)(K >> (8 * t)) & 255u : 0u;
}

template <bool SAT>
__global__ __launch_bounds__(256, 6) void gauss7_mfma_kernel(PyramidView src, PyramidView dst, BlurMfmaPlan plan, GaussTaps T, int tasks_per_image, int n_images, int dbg) {
    // the workgroup's output tile (32 rows x 4 strips = 128 bytes per row) on its way to row-major, double buffered
    __shared__ __attribute__((aligned(16))) uint32_t wg_tile[2][32 * 32];
    const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31, wv = threadIdx.x >> 6;
    if (dbg & 8) __builtin_amdgcn_s_setprio(3);   // experiment
    // One workgroup = four ADJACENT strips (128 columns) x plan.chunk blocks, its four waves in step: the finished 32 x 128 tile is
    // stored as whole 128-byte lines.  (Every wave storing its own 32-byte row pieces left the L2 with quarter lines from four
    // waves at four different times: 0.22 of that version's 0.42 ms.)  The grid may be smaller than the task list.
    const int n_all = tasks_per_image * n_images;
    int parity = 0;
    for (int gt = (int)blockIdx.x; gt < n_all; gt += (int)gridDim.x) {
    const int img = gt / tasks_per_image, task = gt - img * tasks_per_image;
    int level = 0;
    while (level + 1 < plan.nlevels && task >= plan.task_begin[level + 1]) level++;
    const int idx = task - plan.task_begin[level];
    const int n_groups = plan.strips[level];            // groups of four strips
    const int rc = idx / n_groups, sg = idx - rc * n_groups;
    const int cs = 4 * sg + wv;                         // (a strip past the level's last one works on columns >= w: nothing is stored)
    const LevelView sv = src.lv[level], dv = dst.lv[level];
    const int w = sv.w, h = sv.h, X = 32 * cs;
    const uint8_t* sb = sv.base + (size_t)img * sv.img_stride;
    uint8_t* db = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;
    unsigned long long K = 0;
    uint32_t ksum = 0;
#pragma unroll
    for (int t = 0; t < 7; t++) { K |= (unsigned long long)(T.k[t] & 255u) << (8 * t); ksum += T.k[t]; }

    // B operand of the horizontal pass (band matrix): 2 K steps x 16 bytes per lane
    bm_v4i bh[2];
    const int xn = X + c;
    const bool edge = X == 0 || X + 32 + 3 >= w;   // wave-uniform: some output column of the strip reaches across a border
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t word = 0;
            const int x0 = X - 16 + 32 * s + 16 * half + 4 * d;
            if (!edge) {
                // interior: bytes t0 .. t0 + 3 of the zero-padded tap sequence, t0 = x0 - xn + 3
                const int t0 = x0 - xn + 3;
                if (t0 > -4 && t0 < 7) word = t0 >= 0 ? (uint32_t)(K >> (8 * t0)) : (uint32_t)(K << (8 * -t0));
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int x = x0 + b;
                    uint32_t wgt = 0;
                    if (x >= 0 && x < w) {
                        wgt = bm_tap(x - xn + 3, K);
                        if (x >= 1) wgt += bm_tap(-x - xn + 3, K);                       // source column -x reflects onto x
                        if (x <= w - 2) wgt += bm_tap(2 * (w - 1) - x - xn + 3, K);      // source column 2 (w - 1) - x too
                    }
                    word |= (wgt & 255u) << (8 * b);
                }
            }
            bh[s][d] = (int)word;
        }
    // Weights of the vertical pass (B operand: lane = output row m = c of the block, byte 4 q + b <-> H row rho(half, 4 q + b) of
    // tile j, resp. 32 + rho of tile j + 1), for output row 32 j + 4 + m
    bm_v4i wv_own, wv_next;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t wo = 0, wn = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int rho = 4 * half + b + 8 * q;
            wo |= bm_tap(rho - c - 1, K) << (8 * b);
            if (q == 0) wn |= bm_tap(31 + rho - c, K) << (8 * b);     // rows 0 .. 6 of the next tile sit in dword 0 of both halves
        }
        wv_own[q] = (int)wo; wv_next[q] = (int)wn;
    }
    bm_v16i czero;
#pragma unroll
    for (int i = 0; i < 16; i++) czero[i] = 0;
    // Bookkeeping of the signed bytes.  Pixels enter as a = p - 128, so the row sums come out as H' = H - 128 ksum.  H' (16 bits,
    // signed) is split into hi = H' >> 8 (signed byte) and lo = H' & 255, fed as lo - 128: with every output's weights summing to
    // ksum,  V + 32768 = 256 * acc_hi + acc_lo + 128 ksum + 128 ksum^2 + 32768,  and the pixel is bits 16 .. 23 of that.
    const uint32_t round_c = 128u * ksum + 128u * ksum * ksum + 32768u;

    // source columns of this lane's two 16-byte loads (clamped loads only ever hold columns of weight 0)
    int xoff[2];
#pragma unroll
    for (int s = 0; s < 2; s++) xoff[s] = min(max(X - 16 + 32 * s + 16 * half, 0), sv.pitch - 16);
    struct Px { bm_v4i a0, a1; };
    struct Hp { bm_v4i hi, lo; };   // an H tile as the two byte planes of the vertical pass's A operand
    auto load_rows = [&](int J) {   // pixel rows 32 J .. 32 J + 31 (reflected), this lane's 2 x 16 bytes
        int y = 32 * J + c;
        y = y < 0 ? -y : (y >= h ? 2 * (h - 1) - y : y);
        y = min(max(y, 0), h - 1);            // (far outside: feeds only outputs that are not stored)
        const uint8_t* rp = sb + (size_t)y * sv.pitch;
        Px p;
        p.a0 = *reinterpret_cast<const bm_v4i*>(rp + xoff[0]);
        p.a1 = *reinterpret_cast<const bm_v4i*>(rp + xoff[1]);
        return p;
    };
    auto h_tile = [&](Px p) {   // the horizontal pass of those rows, split into byte planes
#pragma unroll
        for (int d = 0; d < 4; d++) { p.a0[d] ^= (int)0x80808080u; p.a1[d] ^= (int)0x80808080u; }
        bm_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(p.a0, bh[0], czero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(p.a1, bh[1], acc, 0, 0, 0);
        Hp t;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t t0 = __builtin_amdgcn_perm((uint32_t)acc[4 * q + 1], (uint32_t)acc[4 * q], 0x05040100u);       // lo0 hi0 lo1 hi1
            const uint32_t t1 = __builtin_amdgcn_perm((uint32_t)acc[4 * q + 3], (uint32_t)acc[4 * q + 2], 0x05040100u);   // lo2 hi2 lo3 hi3
            t.lo[q] = (int)(__builtin_amdgcn_perm(t1, t0, 0x06040200u) ^ 0x80808080u);
            t.hi[q] = (int)__builtin_amdgcn_perm(t1, t0, 0x07050301u);
        }
        return t;
    };
    // The vertical product is formed TRANSPOSED (H as the A operand, the weights as B: same registers, same constants): a lane owns
    // one output ROW (lane & 31) and the columns 4 half + 8 q + 0..3 of the strip.
    auto out_block = [&](int j, const Hp& Hc, const Hp& Hn) {
        bm_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Hc.hi, wv_own, czero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Hn.hi, wv_next, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = (int)(((uint32_t)acc[i] << 8) + round_c);   // the lo plane accumulates on top of 256 * hi + constant
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Hc.lo, wv_own, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Hn.lo, wv_next, acc, 0, 0, 0);
        // acc[4 q + b] = row (lane & 31), column 4 half + 8 q + b of the wave's 32 x 32 tile: into the workgroup tile (dword column
        // 8 wave + half + 2 q), barrier, then wave w stores rows 8 w .. 8 w + 7 of the whole 128-byte-wide tile, 16 bytes per lane.
        uint32_t* const tw = wg_tile[parity] + (lane & 31) * 32 + 8 * wv + half;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t sum[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                sum[b] = (uint32_t)acc[4 * q + b];    // V + 32768: the pixel is bits 16 .. 23
                if (SAT) sum[b] = min(sum[b], 0x00FFFFFFu);
            }
            tw[2 * q] = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);   // the four bytes 2
        }
        __syncthreads();
        const int row = 8 * wv + (lane >> 3);
        const bm_v4i row16 = *reinterpret_cast<const bm_v4i*>(wg_tile[parity] + row * 32 + 4 * (lane & 7));
        parity ^= 1;    // (the other buffer is free: its readers passed this block's barrier)
        const int y = 32 * j + 4 + row, x = 128 * sg + 16 * (lane & 7);
        if (y >= 0 && y < h && !(dbg & 1)) {
            uint8_t* op = db + (size_t)y * dv.pitch + x;
            if (x + 15 < w) *reinterpret_cast<bm_v4i*>(op) = row16;
            else
                for (int b = 0; b < 16; b++)
                    if (x + b < w) op[b] = (uint8_t)((uint32_t)row16[b >> 2] >> (8 * (b & 3)));
        }
    };

    const int jb = -1 + rc * plan.chunk, je = min(jb + plan.chunk, plan.nblocks[level] - 1);
    // Loads and stores share one in-order counter on this chip (vmcnt), so a wave that waits for the next tile's rows also waits
    // for its last store: a deep prefetch ring inside the wave (built, measured slower) does not hide that — many waves per SIMD
    // do.  The loop therefore keeps its register footprint small: two H tiles alternate, the rows of the next tile are requested
    // one block ahead.
    Hp Ha = h_tile(load_rows(jb)), Hb;
    Px pn = load_rows(jb + 1);
    for (int j = jb; j < je; j += 2) {
        Hb = h_tile(pn);
        pn = load_rows(j + 2);
        out_block(j, Ha, Hb);
        if (j + 1 >= je) break;
        Ha = h_tile(pn);
        pn = load_rows(j + 3);
        out_block(j + 1, Hb, Ha);
    }
    }
}

int launch_gauss7(const PyramidView& src, const PyramidView& dst, int n_images, hipStream_t s, const Semantics& sem) {
    BlurPlan plan{};
    plan.nlevels = src.nlevels;
    bool aligned = true;
    for (int l = 0; l < src.nlevels; l++) {
        const LevelView& v = src.lv[l];
        aligned = aligned && (reinterpret_cast<uintptr_t>(v.base) & 3) == 0 && (v.pitch & 3) == 0 && (v.img_stride & 3) == 0;
    }
    // matrix-core form (batches on 16-byte aligned planes; any taps): see gauss7_mfma_kernel
    {
        const char* e = getenv("MSORB_BLUR_MFMA");   // read per call: the tests run both forms in one process
        bool mfma = e ? atoi(e) != 0 : false;
        GaussTaps Tm;
        uint32_t ksum = 0;
        for (int i = 0; i < 7; i++) { Tm.k[i] = (uint32_t)sem.gauss_taps[i]; ksum += Tm.k[i]; }
        for (int l = 0; l < src.nlevels && mfma; l++) {
            const LevelView& v = src.lv[l];
            const LevelView& d = dst.lv[l];
            mfma = (reinterpret_cast<uintptr_t>(v.base) & 15) == 0 && (v.pitch & 15) == 0 && (v.img_stride & 15) == 0 && v.w >= 8 && v.h >= 8 &&
                   v.pitch >= 16 && d.w == v.w && d.h == v.h;
        }
        for (int i = 0; i < 7 && mfma; i++) mfma = Tm.k[i] <= 64;   // a folded border weight (two taps) must fit a signed byte
        // (taps summing to more than 256 — a semantics variant — would need a 17-bit row sum: the VALU kernels take them)
        if (mfma && ksum <= 256) {
            BlurMfmaPlan mp{};
            mp.nlevels = src.nlevels;
            static const int chunk_env = getenv("MSORB_BLUR_MFMA_CHUNK") ? atoi(getenv("MSORB_BLUR_MFMA_CHUNK")) : 0;   // tuning only
            mp.chunk = chunk_env > 0 ? chunk_env : 16;
            static const int bm_dbg = getenv("MSORB_BLUR_MFMA_DEBUG") ? atoi(getenv("MSORB_BLUR_MFMA_DEBUG")) : 0;   // timing experiments only
            int total = 0;
            for (int l = 0; l < src.nlevels; l++) {
                const LevelView& v = src.lv[l];
                mp.task_begin[l] = total;
                mp.strips[l] = ((v.w + 31) / 32 + 3) / 4;   // groups of four 32-column strips
                const int jlast = v.h > 36 ? (v.h - 36 + 31) / 32 : 0;   // 32 jlast + 35 >= h - 1
                mp.nblocks[l] = jlast + 2;                                // blocks -1 .. jlast
                total += mp.strips[l] * ((mp.nblocks[l] + mp.chunk - 1) / mp.chunk);
            }
            mp.task_begin[src.nlevels] = total;
            static const int wgs_env = getenv("MSORB_BLUR_MFMA_WGS") ? atoi(getenv("MSORB_BLUR_MFMA_WGS")) : 0;   // tuning only
            const int n_wg = std::min(wgs_env > 0 ? wgs_env : 1 << 20, total * n_images);   // (a smaller grid: the workgroups loop over the tasks)
            hipLaunchKernelGGL(gauss7_mfma_kernel<false>, dim3(n_wg), dim3(256), 0, s, src, dst, mp, Tm, total, n_images, bm_dbg);
            return 1;
        }
    }
    static const bool stream_env = !getenv("MSORB_BLUR_GENERIC");  // tuning / test aid
    // the streaming kernel has the default taps folded into its v_dot4 constants; other taps (Semantics::gauss_taps) take the
    // generic kernels, which read them at run time
    const bool stream = aligned && stream_env && sem.default_taps();
    GaussTaps T;
    for (int i = 0; i < 7; i++) T.k[i] = (uint32_t)sem.gauss_taps[i];
    // strip height 35: 21..35 rows measure the same (0.29 ms / 256 images), 70 and 140 are slower (too few waves)
    const int rows = kGaussRows;
    int total = 0, max_h = 0;
    for (int l = 0; l < src.nlevels; l++) {
        const LevelView& v = src.lv[l];
        plan.block_begin[l] = total;
        // aligned path: 62 stored groups per wave; generic path: 64 groups per wave
        const int main_groups = stream ? (v.w + 3) / 4 : (v.w >= 16 ? (v.w - 16) / 4 + 1 : 0);  // stream kernel: every group
        // (62 stored lanes per wave, + the row's last group on lane 63 if it is exactly one group more)
        plan.bx_count[l] = stream ? max(1, (main_groups - 1 + kGaussLanesOut - 1) / kGaussLanesOut) : (v.w + 255) / 256;
        const int strips = (v.h + rows - 1) / rows;
        total += plan.bx_count[l] * ((strips + 3) / 4);
        max_h = max(max_h, v.h);
    }
    plan.block_begin[src.nlevels] = total;
    if (stream) {
        const int all = total * n_images, per_xcd = (all + 7) / 8;
        hipLaunchKernelGGL(gauss7_stream_kernel<kGaussRows>, dim3(per_xcd * 8), dim3(256), 0, s, src, dst, plan, per_xcd, all);
    }
    else if (aligned) hipLaunchKernelGGL(gauss7_kernel<true>, dim3(total, n_images), dim3(256), 0, s, src, dst, plan, T);
    else hipLaunchKernelGGL(gauss7_kernel<false>, dim3(total, n_images), dim3(256), 0, s, src, dst, plan, T);
    if (!stream)  // the streaming kernel handles the right border itself
        hipLaunchKernelGGL(gauss7_edge_kernel, dim3((max_h + kEdgeRows - 1) / kEdgeRows, src.nlevels, n_images), dim3(64), 0, s, src, dst, T);
    return 0;
}
void launch_describe(const PyramidView& pyr, const PyramidView& blur, const SelRec* sel, const int* sel_count,
                     int sel_stride, const LevelScale& scales, msorb_keypoint* kps, uint8_t* desc, int out_stride,
                     int max_sel, int n_images, hipStream_t s, const Semantics& sem) {
    if (max_sel <= 0) return;
    hipLaunchKernelGGL(describe_kernel, dim3((max_sel + 4 * kKpPerWave - 1) / (4 * kKpPerWave), n_images), dim3(256), 0, s, pyr, blur, sel, sel_count,
                       sel_stride, scales, kps, desc, out_stride, sem.atan2_fma);
}

}  // namespace msorb
