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

#include <algorithm>
#include <string>
#include <stdint.h>
#include <stdlib.h>

#include "lds_limit.h"
#include "orb_device.h"
#include "gauss7_stream_device.h"
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

// Row-band kernel (batches, scale factors up to 1.25): a wave produces R consecutive output rows of its 64 column groups in
// two phases without any data-dependent control flow in between:
//   A  all source rows the band touches (rows i0(dy0) .. i1(dy0 + R - 1): <= kSrc of them) are fetched with every load in
//      flight at once, interpolated horizontally (v_perm + v_dot2_u32_u16 per pixel) and parked as 4 x u16 per lane in
//      REGISTERS (s_set_gpr_idx picks; no LDS: beside the other batch's FAST / quadtree / describe workgroups, which keep a
//      CU's LDS filled to within a few KB, a workgroup that asks for none starts as soon as two wave slots are free);
//   B  every output row reads its two parked rows by (wave-uniform) index and blends them with SDWA half-word operands.
// The y taps of the band come in up front with the first loads.  14 VALU lane-operations per pixel.
// (Retired forms with their measurements — row streaming, rows parked in LDS, LDS-DMA staging, the top levels as one launch:
// tools/experiments/README.md.)
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

template <int R, int kSrc, int WAVES = 2>
__global__ __launch_bounds__(64 * WAVES) void pyr_resize_bandreg_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                                        const ResizeTap* __restrict__ tx,
                                                                        const ResizeTap* __restrict__ ty, uint32_t per_img_magic,
                                                                        uint32_t gx_magic) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (private L2 each); give every XCD whole images, so that
    // the source rows two neighbouring bands share and the 12-byte windows neighbouring lanes re-request are fetched into one
    // L2 instead of several (batches of a multiple of 8 images; others keep the plain order)
    unsigned bx = blockIdx.x, by = blockIdx.y, img = blockIdx.z;
    if ((gridDim.z & 7u) == 0) {
        const unsigned per_img = gridDim.x * gridDim.y;
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        // (host-checked exact reciprocals, exact_div_magic: two divisions per wave were 4 % of this kernel's instructions)
        const unsigned q = per_img_magic ? __umulhi(j, per_img_magic) : j / per_img, rem = j - q * per_img;
        img = q * 8u + xcd;
        by = gx_magic ? __umulhi(rem, gx_magic) : rem / gridDim.x;
        bx = rem - by * gridDim.x;
    }
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(by * WAVES + wave) * R);
    pyr_band_tile_reg<R, kSrc>(src, dst, dst_base, tx, ty, (int)img, (int)bx, dy0, lane);
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

// FAST corner score: max over the 16 arcs of 9 contiguous circle pixels of min(sgn * (v - p)); p = LDS pointer to the centre.
// circle offsets (x,y): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
// (Rounds 1-4 ran the network on 32-bit lanes: 16 v_mad + 32 v_min3_i32 + 8 v_max3_i32; `git show f40730a` has it.)
// Packed form (round 5): the sixteen signed contrasts live two to a register as 16-bit halves,
// D[k] = (d[k], d[k + 8]), so "eight positions further round the circle" is the other half of the same register and costs
// nothing (VOP3P op_sel picks the halves of an operand: D[k + 8] is D[k] read swapped).  OpenCV's own decomposition
// (`cornerScore<16>`: arcs i and i + 1 share d[i+1 .. i+8], max(min(c, a), min(c, b)) = min(c, max(a, b))) on packed pairs:
//   m2[j] = min(d[j], d[j+1]), m4[j] = min(m2[j], m2[j+2]) (j odd), c8[i] = min(m4[i+1], m4[i+5]),
//   e[i] = max(d[i], d[i+9]), pair[i] = min(c8[i], e[i]) (i even), A = max over pair[]
// = 5 x 4 packed instructions + 4 for the maximum, after 8 v_lshl_or (pack) + 8 v_pk_mad_i16 (sign and centre): 40
// VALU instructions per work entry against 16 v_mad + 32 v_min3 + 8 v_max3 = 56.  |d| <= 255: nothing saturates.
#define MSORB_PK2(NAME, OPC)                                                                                                   \
    __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b) {                                                         \
        uint32_t r; asm(OPC " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }                                             \
    __device__ __forceinline__ uint32_t NAME##_sw(uint32_t a, uint32_t b) { /* b's halves swapped */                           \
        uint32_t r; asm(OPC " %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
MSORB_PK2(pk_min16, "v_pk_min_i16")
MSORB_PK2(pk_max16, "v_pk_max_i16")
#undef MSORB_PK2
__device__ __forceinline__ uint32_t pk_mad16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}

template <class GEO>
__device__ __forceinline__ int fast_arc_contrast_pk(const uint8_t* p, int sgn) {
    constexpr int P = GEO::kTilePitch;
    const uint32_t sv = ((uint32_t)(sgn * (int)p[0]) & 0xffffu) * 0x10001u, ns = ((uint32_t)(-sgn) & 0xffffu) * 0x10001u;
    uint32_t D[8];
    D[0] = pk_mad16(((uint32_t)p[-3 * P] << 16) | p[3 * P], ns, sv);            // (0,3)   | (0,-3)
    D[1] = pk_mad16(((uint32_t)p[-3 * P - 1] << 16) | p[3 * P + 1], ns, sv);    // (1,3)   | (-1,-3)
    D[2] = pk_mad16(((uint32_t)p[-2 * P - 2] << 16) | p[2 * P + 2], ns, sv);    // (2,2)   | (-2,-2)
    D[3] = pk_mad16(((uint32_t)p[-1 * P - 3] << 16) | p[1 * P + 3], ns, sv);    // (3,1)   | (-3,-1)
    D[4] = pk_mad16(((uint32_t)p[-3] << 16) | p[3], ns, sv);                    // (3,0)   | (-3,0)
    D[5] = pk_mad16(((uint32_t)p[1 * P - 3] << 16) | p[-1 * P + 3], ns, sv);    // (3,-1)  | (-3,1)
    D[6] = pk_mad16(((uint32_t)p[2 * P - 2] << 16) | p[-2 * P + 2], ns, sv);    // (2,-2)  | (-2,2)
    D[7] = pk_mad16(((uint32_t)p[3 * P - 1] << 16) | p[-3 * P + 1], ns, sv);    // (1,-3)  | (-1,3)
    const uint32_t m2_1 = pk_min16(D[1], D[2]), m2_3 = pk_min16(D[3], D[4]), m2_5 = pk_min16(D[5], D[6]),
                   m2_7 = pk_min16_sw(D[7], D[0]);
    const uint32_t m4_1 = pk_min16(m2_1, m2_3), m4_3 = pk_min16(m2_3, m2_5), m4_5 = pk_min16(m2_5, m2_7),
                   m4_7 = pk_min16_sw(m2_7, m2_1);
    const uint32_t c8_0 = pk_min16(m4_1, m4_5), c8_2 = pk_min16(m4_3, m4_7), c8_4 = pk_min16_sw(m4_5, m4_1),
                   c8_6 = pk_min16_sw(m4_7, m4_3);
    const uint32_t e0 = pk_max16_sw(D[0], D[1]), e2 = pk_max16_sw(D[2], D[3]), e4 = pk_max16_sw(D[4], D[5]),
                   e6 = pk_max16_sw(D[6], D[7]);
    const uint32_t a = pk_max16(pk_max16(pk_min16(c8_0, e0), pk_min16(c8_2, e2)), pk_max16(pk_min16(c8_4, e4), pk_min16(c8_6, e6)));
    return (int)(short)(pk_max16_sw(a, a) & 0xffffu);
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

// The kernel's body, on a workgroup's LINEAR index among `total` = cells_x * n_images workgroups (so that the frame kernel below can
// run it on a sub-range of its grid).
template <bool ALIGNED, class GEO>
__device__ __forceinline__ void fast_cells_body(const PyramidView& pyr, const CellDesc* __restrict__ cells,
                                                         int ini_th, int min_th, int slots_per_image,
                                                         Cand16* __restrict__ slots, int* __restrict__ cell_count,
                                                         int n_cells, uint32_t gx_magic, const unsigned lin, const unsigned grid_x, const unsigned total) {
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
    const unsigned chunk = (total + 7) >> 3;
    unsigned wg = (lin & 7u) * chunk + (lin >> 3);
    if (total & 7u) wg = lin;  // ragged totals keep the plain order (bench / test geometries are multiples of 8 images)
    const int img = gx_magic ? (int)__umulhi(wg, gx_magic) : (int)(wg / grid_x);  // host-checked exact reciprocal
    const int cell_id = (int)(wg - (unsigned)img * grid_x);
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
                    const int A = fast_arc_contrast_pk<GEO>(&tile[(int)__umul24((uint32_t)ty, P) + tx], (e & 0x8000) ? -1 : 1);
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
                    const int A = fast_arc_contrast_pk<GEO>(&tile[(int)__umul24((uint32_t)ty, P) + tx], (e & 0x8000) ? -1 : 1);
                    if (A > th) score[(int)__umul24((uint32_t)(ty - 2), SP) + tx + sc_off] = (uint8_t)(A - 1);
                }
        }
        __syncthreads();

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

template <bool ALIGNED, class GEO>
__global__ __launch_bounds__(GEO::kThreads, GEO::kMinWaves) void fast_cells_kernel(PyramidView pyr, const CellDesc* __restrict__ cells,
                                                         int ini_th, int min_th, int slots_per_image,
                                                         Cand16* __restrict__ slots, int* __restrict__ cell_count,
                                                         int n_cells, uint32_t gx_magic) {
    fast_cells_body<ALIGNED, GEO>(pyr, cells, ini_th, min_th, slots_per_image, slots, cell_count, n_cells, gx_magic,
                                  blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.x * gridDim.y);
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: per image exclusive scan of the cell counts (cells are already in the
// reference's row-major cell order), a one-block scan over images, then a gather.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_scan_cells_kernel(const int* __restrict__ cell_count, int n_cells,
                                                              const int* __restrict__ level_cell_begin, int nlevels,
                                                              int* __restrict__ cell_off, int* __restrict__ level_count,
                                                              int* __restrict__ img_total, int* __restrict__ img_base,
                                                              int fixed_stride) {
    __shared__ int wave_tot[4];
    __shared__ int lvl[kMaxLevels];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fixed_stride > 0: image i's compacted candidates start at i * fixed_stride (no scan over the images, no second launch)
    if (fixed_stride > 0 && tid == 0) {
        img_base[img] = img * fixed_stride;
        if (img == (int)gridDim.x - 1) img_base[gridDim.x] = (int)gridDim.x * fixed_stride;
    }
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
constexpr int kGatherThreads = 256;   // (64 / 128 measured inside the pipelined step: no different)
constexpr int kGatherCells = kGatherThreads / 16;   // groups of 32 lanes, two cells each (their loads in flight together)
__global__ __launch_bounds__(kGatherThreads) void cand_gather_kernel(const CellDesc* __restrict__ cells, int n_cells,
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

// Compaction of a FRAME (1-4 images) as ONE launch: every gather workgroup repeats the scan over its image's cell counts (a few
// hundred integers from L2, the offsets kept in LDS) instead of waiting for a scan launch — the frame chain loses a launch and the
// boundary in front of it; workgroup 0 of an image also writes what the scan kernel writes (cell_off, level_count, img_total,
// img_base).  Image i's run starts at i * fixed_stride, as in the two-launch form with packed = false.
constexpr int kFrameCompactCells = 2048;   // cells of one image the LDS offset table holds (KITTI 902, EuRoC 700)
__global__ __launch_bounds__(kGatherThreads) void cand_compact_frame_kernel(const CellDesc* __restrict__ cells, int n_cells,
                                                                            const int* __restrict__ level_cell_begin, int nlevels, int fixed_stride,
                                                                            const Cand16* __restrict__ slots, const int* __restrict__ cell_count,
                                                                            int* __restrict__ cell_off, int* __restrict__ level_count,
                                                                            int* __restrict__ img_total, int* __restrict__ img_base,
                                                                            Cand16* __restrict__ compact) {
    static_assert(kGatherThreads == 256, "the scan below is written for four waves");
    __shared__ int wave_tot[4];
    __shared__ int lvl[kMaxLevels];
    __shared__ int off_s[kFrameCompactCells];
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool writer = blockIdx.x == 0;
    const int* cnt = cell_count + (size_t)img * n_cells;
    int* off = cell_off + (size_t)img * n_cells;
    const int per = (n_cells + 255) / 256;
    const int b = tid * per, e = min(b + per, n_cells);
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
    int cur_l = 0, cur_t = 0;
    auto visit = [&](int i, int v) {
        off_s[i] = acc;
        if (writer) {
            off[i] = acc;
            while (i >= level_cell_begin[cur_l + 1]) {
                if (cur_t) atomicAdd(&lvl[cur_l], cur_t);
                cur_t = 0;
                cur_l++;
            }
            cur_t += v;
        }
        acc += v;
    };
#pragma unroll
    for (int k = 0; k < kCache; k++)
        if (b + k < e) visit(b + k, c[k]);
    for (int i = b + kCache; i < e; i++) visit(i, cnt[i]);
    if (writer && cur_t) atomicAdd(&lvl[cur_l], cur_t);
    __syncthreads();
    if (writer) {
        if (tid < nlevels) level_count[(size_t)img * nlevels + tid] = lvl[tid];
        if (tid == 0) {
            img_total[img] = total;
            img_base[img] = img * fixed_stride;
            if (img == (int)gridDim.y - 1) img_base[gridDim.y] = (int)gridDim.y * fixed_stride;
        }
    }
    // the gather of cand_gather_kernel, offsets from LDS
    const int c0 = blockIdx.x * kGatherCells + 2 * (int)(threadIdx.x >> 5), l = threadIdx.x & 31;
    if (c0 >= n_cells) return;
    const int c1 = min(c0 + 1, n_cells - 1);
    const int n0 = cnt[c0], n1 = c0 + 1 < n_cells ? cnt[c1] : 0;
    const int o0 = off_s[c0], o1 = off_s[c1], s0 = cells[c0].slot_off, s1 = cells[c1].slot_off;
    const Cand16* sp = slots + (size_t)img * fixed_stride;
    Cand16* d = compact + (size_t)img * fixed_stride;
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
    uint8_t* db = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;   // tiled plane (blur_tile_off)
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
                *reinterpret_cast<uint32_t*>(db + blur_tile_off((uint32_t)x0, (uint32_t)(y0 + o), (uint32_t)dv.pitch)) = packed;
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
    gauss7_stream_body<ROWS>(src, dst, plan, tile, (int)(threadIdx.x >> 6));
}

// FAST and the blur of a FRAME (1-4 images) as ONE launch: both only need the pyramid, and as two launches the blur ran on a side
// stream — an event record, two stream waits and a launch per frame, and a fork / join that costs the chain ~8 us on this
// runtime whatever runs in it (tools/launch_ubench.hip, profiles/round6_launch_ubench.txt).  Workgroups [0, n_fast) run
// fast_cells_body on their cell, the rest gauss7_stream_body: GEO::kThreads / 64 waves of a blur block each (the blur's waves are
// independent).  The blur's registers (88) set the allocation: no minimum-waves bound here — a frame does not fill the chip.
template <bool ALIGNED, class GEO>
__global__ __launch_bounds__(GEO::kThreads) void frame_fast_blur_kernel(PyramidView pyr, const CellDesc* __restrict__ cells, int ini_th, int min_th,
                                                                        int slots_per_image, Cand16* __restrict__ slots, int* __restrict__ cell_count,
                                                                        int n_cells, uint32_t gx_magic, unsigned n_fast, PyramidView blur, BlurPlan plan,
                                                                        int blur_blocks) {
    if (blockIdx.x < n_fast) {
        fast_cells_body<ALIGNED, GEO>(pyr, cells, ini_th, min_th, slots_per_image, slots, cell_count, n_cells, gx_magic, blockIdx.x, (unsigned)n_cells, n_fast);
        return;
    }
    constexpr int kWavesPerWg = GEO::kThreads / 64, kWgPerTile = 4 / kWavesPerWg;   // 128 threads: two workgroups per blur block
    const unsigned b = blockIdx.x - n_fast;
    const int tile = (int)(b / kWgPerTile);
    if (tile >= blur_blocks) return;
    gauss7_stream_body<kGaussRows>(pyr, blur, plan, tile, (int)(b % kWgPerTile) * kWavesPerWg + (int)(threadIdx.x >> 6));
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
            db[blur_tile_off((uint32_t)(first + j), (uint32_t)(y0 + r), (uint32_t)dv.pitch)] = (uint8_t)min((acc + 32768u) >> 16, 255u);
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

hipError_t download_patch_tables(int8_t* pattern, int8_t* umax, hipStream_t stream) {
    PatchTables t;
    hipError_t e = hipMemcpyFromSymbolAsync(&t, HIP_SYMBOL(c_tab), sizeof(t), 0, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    memcpy(pattern, t.pattern, sizeof(t.pattern));
    memcpy(umax, t.umax, sizeof(t.umax));
    return hipSuccess;
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

constexpr int kBlkSlot = 1600;  // one keypoint's blurred neighbourhood in LDS: 10 x 10 blocks of 16 bytes, block (a, b) at (10 a + b) * 16
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from any 4-byte boundary
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kKpPerWave = 4;   // keypoints handled back to back by one wave (amortises the per-lane table loads)
#ifndef MSORB_DESC_RAW_DEPTH
#define MSORB_DESC_RAW_DEPTH 4  // keypoints whose IC-angle patch loads are in flight beyond the one being consumed (4 = all of the wave's, measured best: 0.2777 (r4 kernel) / 0.273 / 0.270 / 0.264 / 0.265 ms for r4 / 1 / 2 / 3 / 4)
#endif

// LDS-DMA (global_load_lds_dwordx4): lane l fetches the 16 bytes at sbase + voff(l) and the hardware writes them to
// LDS[lds_dst + 16 l] — no VGPR is the destination, so a load in flight costs no register.  M0 carries the wave-uniform
// destination; it is compiler-reserved, so it is saved, set and restored inside the one statement (guide section 5.7).  The
// compiler does not count this load: the reader waits for vmcnt itself (glds_wait_all).
__device__ __forceinline__ void glds16(const uint8_t* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// v_pk_add_f32 with the second operand's low half broadcast from a scalar pair, and v_mad_u32_u24 with a scalar multiplier
// (the compiler splits the one into two v_add_f32 and the other into v_mul_u32_u24 + v_add3_u32 when left to itself)
constexpr unsigned long long kRoundMagic = 0x4B4000004B400000ull;   // 12582912.0f twice: 1.5 * 2^23
__device__ __forceinline__ f32x2 pk_add_bcast(f32x2 v, unsigned long long s) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "s"(s)); return r;
}
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t s, uint32_t c) {
    uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r;
}

constexpr int kDescWaves = 4;   // waves per descriptor workgroup (2: 0.262 ms alone; 8: 0.239 alone but 1.088 instead of 1.057 ms per pipelined step — a 51 KB workgroup finds room later)
// kTap = Semantics::brief_tap: which product of the rotated tap  x*b + y*a / x*a - y*b  (ORBextractor.cc:117-119) the reference's
// compiler fused — 0: the first (fma(x, b, y*a); default), 1: the second (fma(y, a, x*b)), 2: none.  A template parameter: the
// default costs what it did before the variants existed.
template <int kTap>
__global__ __launch_bounds__(64 * kDescWaves) __attribute__((amdgpu_waves_per_eu(6, 6))) void describe_kernel(PyramidView pyr, PyramidView blur, const SelRec* __restrict__ sel,
                                                       const int* __restrict__ sel_count, int sel_stride,
                                                       LevelScale scales, msorb_keypoint* __restrict__ kps,
                                                       uint8_t* __restrict__ desc, int out_stride, int atan2_fma, uint32_t gx_magic) {
    __shared__ __attribute__((aligned(16))) uint8_t patch[kDescWaves * kKpPerWave][kBlkSlot];  // one slot per (wave, keypoint)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; give every image to ONE XCD so that the
    // overlapping keypoint patches of an image are served by a single L2 instead of being fetched by all eight.
    int img = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7u) == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        const unsigned q = gx_magic ? __umulhi(j, gx_magic) : j / gridDim.x;   // host-checked exact reciprocal (a division costs 25 VALU instructions per wave)
        img = (int)(q * 8 + xcd);
        bx = (int)(j - q * gridDim.x);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k_first = (bx * kDescWaves + wave) * kKpPerWave;  // wave-uniform -> SALU
    const int n_sel = sel_count[img];
    if (k_first >= n_sel) return;
    // IC-angle patch (raw level, registers): 16-byte loads, three lanes per patch row (lane = 3 row' + seg, 21 rows per load
    // instruction, lane 63 idles): 2 load instructions per keypoint.  The texture addresser spends its cycles per lane address,
    // not per byte (PMC, round 4), so the same bytes in a quarter of the addresses is what shortened this kernel in round 4.
    // Row v = row - 15 from the 4-byte boundary at or below x - 15; after the byte re-alignment below, dword col = 4 seg + d
    // (col < 8) holds the pixels u = 4 col - 15 .. 4 col - 12.  wu = (u + 16) per byte inside the circle (0 outside), vm = 1 / 0:
    // two udot4 chains give sum(u I), sum(I).  seg 2 only feeds the alignment of col 7.
    const int seg = lane % 3, row3 = lane / 3;
    uint32_t wu[2][4], vm[2][4];
    uint32_t rrow_c[2];   // clamped IC-angle row of this lane in load t (rows of no patch row repeat the last one: same line)
    int vrow[2];
    // The weights of a dword, all four bytes at once: byte b of U is u + 16 = 16 seg + 4 d + 1 + b (1 .. 48); a pixel is inside
    // the circle iff 16 - dd <= U_b <= 16 + dd, tested per byte through bit 7 of U + (0x7f - hi) ("above hi") and of
    // U + (0x80 - lo) ("at or above lo") — no carries: 48 + 128 < 256.  dwords past column 7 (U_b >= 33 > hi) and rows past
    // 30 (no lower bound added: bit 7 stays clear) come out empty by themselves.
    const uint32_t U0 = (uint32_t)(16 * seg + 1) * 0x01010101u + 0x03020100u;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int row = lane < 63 ? 21 * t + row3 : 63;
        const bool valid = row < 31;
        const int dd = c_tab.umax[valid ? (row < 15 ? 15 - row : row - 15) : 0];
        const uint32_t above = (uint32_t)(0x7f - 16 - dd) * 0x01010101u;
        const uint32_t atlo = valid ? (uint32_t)(0x80 - 16 + dd) * 0x01010101u : 0u;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t U = U0 + (uint32_t)d * 0x04040404u;
            const uint32_t in7 = (U + atlo) & ~(U + above) & 0x80808080u;
            const uint32_t m = in7 >> 7;
            wu[t][d] = U & (in7 | (in7 - m));
            vm[t][d] = m;
        }
        rrow_c[t] = (uint32_t)min(row, 30);
        vrow[t] = row - 15;
    }
    const uint32_t seg16 = 16u * (uint32_t)seg;
    // Blurred neighbourhood (blocked plane, orb_device.h blur_tile_off): the 37 x 37 pixels a descriptor samples lie in 10 x 10
    // blocks of 4 x 4 pixels wherever they start.  They go straight from L2 / HBM into the keypoint's LDS slot by LDS-DMA: lane l
    // of instruction t owns block l + 64 t = (block row (l + 64 t) / 10, block column (l + 64 t) % 10), which the hardware puts
    // at slot + 16 (l + 64 t): the slot is block-indexed, a pixel (rr, qq) of the 40 x 40 block area sits at
    // ((rr >> 2) 10 + (qq >> 2)) 16 + (rr & 3) 4 + (qq & 3).  Ten neighbouring lanes read 160 contiguous bytes.  (Rounds 1-4
    // staged these blocks through registers and re-wrote them row-major: 8 VGPRs per keypoint in flight, 8 conditional ds_write.)
    uint32_t bbrow[2], bbcol16[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int slot = lane + 64 * t;
        bbrow[t] = (uint32_t)min(slot / 10, 9);
        bbcol16[t] = (uint32_t)(slot % 10) * 16u;
    }
    // the 4 pattern pairs of this lane as floats (lane constants: decoded once per wave, not once per keypoint)
    // (kept as (tap 0, tap 1) pairs: the two taps of a test go through the rotation as packed fp32, v_pk_mul / v_pk_fma / v_pk_add)
    f32x2 patx[4], paty[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t pw = *reinterpret_cast<const uint32_t*>(&c_tab.pattern[(w * 64 + lane) * 4]);
        patx[w] = f32x2{(float)(int8_t)(pw & 255u), (float)(int8_t)((pw >> 16) & 255u)};
        paty[w] = f32x2{(float)(int8_t)((pw >> 8) & 255u), (float)(int8_t)(pw >> 24)};
    }
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.0);

    // The wave's four selection records, fetched together and moved to scalar registers: everything derived from them (level
    // view, row pointers, strides, the LDS-DMA bases) is SALU work, and no keypoint's addresses wait for an earlier keypoint.
    // The last wave of an image repeats the image's last keypoint in its unused places (no exit inside the sequence: the four
    // keypoints are one straight line of code; only the stores are conditional).
    const SelRec* recs = sel + (size_t)img * sel_stride;
    SelRec R[kKpPerWave];
    {
        SelRec rv[kKpPerWave];
#pragma unroll
        for (int kk = 0; kk < kKpPerWave; kk++) rv[kk] = recs[min(k_first + kk, n_sel - 1)];
#pragma unroll
        for (int kk = 0; kk < kKpPerWave; kk++) {
            uint32_t w[3];
            memcpy(w, &rv[kk], sizeof(w));
            for (int i = 0; i < 3; i++) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
            memcpy(&R[kk], w, sizeof(w));
        }
    }
    uint8_t* const lp0 = patch[wave * kKpPerWave];
    // LDS byte address of the wave's first slot (a generic pointer to __shared__ is aperture base | offset: C-style cast to address space 3)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lp0);
    // blurred blocks of all four keypoints: on their way before anything else (no register holds them)
    int RQ[kKpPerWave];   // (18 + oy) << 8 | (18 + poff): where the keypoint sits inside its 40 x 40 block area
#pragma unroll
    for (int kk = 0; kk < kKpPerWave; kk++) {
        const SelRec& r = R[kk];
        const LevelView bv = blur.lv[r.level];
        RQ[kk] = ((18 + ((r.y - 18) & 3)) << 8) | (18 + ((r.x - 18) & 3));
        const uint8_t* bbase = bv.base + (size_t)img * bv.img_stride + (size_t)((r.y - 18) >> 2) * ((size_t)bv.pitch * 4) +
                               (size_t)((r.x - 18) >> 2) * 16;
        const uint32_t pitch4 = (uint32_t)bv.pitch * 4u;
        const uint32_t dst = lds0 + (uint32_t)kk * kBlkSlot;
        glds16(bbase, __umul24(bbrow[0], pitch4) + bbcol16[0], dst);
        if (lane < 100 - 64) glds16(bbase, __umul24(bbrow[1], pitch4) + bbcol16[1], dst + 1024u);
    }
    // IC-angle patch loads, MSORB_DESC_RAW_DEPTH keypoints ahead of the one whose moments are being summed
    struct Raw { u32x4u rp[2]; uint32_t rsh[2]; };
    auto issue_raw = [&](const SelRec& r, Raw& L) {
        // 31 rows x 48 bytes (36 used) from the 4-byte boundary at or below x - 15.  The bytes past the patch are in the same
        // image: a keypoint is >= 19 pixels from the border.  (Level 0 may be the caller's own image with any row stride: the
        // 4-byte phase is taken per row.)  Addresses = scalar base + 32-bit lane offset (the global_load saddr form).
        const LevelView lv = pyr.lv[r.level];
        const uint8_t* rrow = lv.base + (size_t)img * lv.img_stride + (size_t)(r.y - 15) * lv.pitch + (r.x - 15) - 4;
        const uint32_t rlow = (uint32_t)reinterpret_cast<uintptr_t>(rrow);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t o = __umul24(rrow_c[t], (uint32_t)lv.pitch);  // full-rate 24-bit multiply
            L.rsh[t] = (rlow + o) & 3u;
            if (t == 0 ? lane < 63 : lane < 3 * (31 - 21))   // rows 0..20 / 21..30, three lanes each
                L.rp[t] = *reinterpret_cast<const u32x4u*>(rrow + (size_t)(o + seg16 + 4u - L.rsh[t]));
        }
    };
    Raw Lq[kKpPerWave];
#pragma unroll
    for (int kk = 0; kk < MSORB_DESC_RAW_DEPTH && kk < kKpPerWave; kk++) issue_raw(R[kk], Lq[kk]);
    // Phase A, per keypoint: moments of the IC-angle patch (wave-uniform totals).  Phase V, once per wave: angle, cos, sin of all
    // four keypoints at once — lane kk computes keypoint kk, so the atan2 polynomial and the double-precision sincos run once
    // per wave instead of once per keypoint.  Phase B, per keypoint: steered BRIEF from its LDS slot.
    int M10[kKpPerWave], M01[kKpPerWave];
#pragma unroll
    for (int kk = 0; kk < kKpPerWave; kk++) {
        if (kk + MSORB_DESC_RAW_DEPTH < kKpPerWave) issue_raw(R[kk + MSORB_DESC_RAW_DEPTH], Lq[kk + MSORB_DESC_RAW_DEPTH]);
        const Raw& L = Lq[kk];
        // IC_Angle: integer moments over the 749-pixel circular patch (un-blurred level).  The next dword of the row is the
        // next register, and for the last one the next lane's first: alignbyte undoes the 4-byte alignment of the loads, so
        // the per-lane weights do not depend on the keypoint.
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const u32x4u v = L.rp[t];
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            const uint32_t sh = L.rsh[t];
            const uint32_t px0 = __builtin_amdgcn_alignbyte(v.y, v.x, sh), px1 = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
            const uint32_t px2 = __builtin_amdgcn_alignbyte(v.w, v.z, sh), px3 = __builtin_amdgcn_alignbyte(nx, v.w, sh);
            uint32_t sI = __builtin_amdgcn_udot4(px0, vm[t][0], 0u, false);
            sI = __builtin_amdgcn_udot4(px1, vm[t][1], sI, false);
            sI = __builtin_amdgcn_udot4(px2, vm[t][2], sI, false);
            sI = __builtin_amdgcn_udot4(px3, vm[t][3], sI, false);
            uint32_t su = __builtin_amdgcn_udot4(px0, wu[t][0], 0u, false);
            su = __builtin_amdgcn_udot4(px1, wu[t][1], su, false);
            su = __builtin_amdgcn_udot4(px2, wu[t][2], su, false);
            su = __builtin_amdgcn_udot4(px3, wu[t][3], su, false);
            m10 += (int)su - 16 * (int)sI;
            m01 += vrow[t] * (int)sI;
        }
        M10[kk] = wave_sum_dpp(m10);  // wave-uniform (SGPR) totals
        M01[kk] = wave_sum_dpp(m01);
    }
    // Phase V
    int m10v = M10[0], m01v = M01[0];
#pragma unroll
    for (int kk = 1; kk < kKpPerWave; kk++)
        if (lane == kk) { m10v = M10[kk]; m01v = M01[kk]; }
    const float angle_v = fast_atan2_deg((float)m01v, (float)m10v, atan2_fma);
    float a_v, b_v;
    glibc_sincosf<true>(__fmul_rn(angle_v, factor_pi), &b_v, &a_v);  // a = cos, b = sin (ORBextractor.cc:112)
    glds_wait_all();   // the four slots have landed (the wave reads only what its own lanes' DMA wrote: no barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Phase B
#pragma unroll
    for (int kk = 0; kk < kKpPerWave; kk++) {
        const SelRec r = R[kk];
        const float angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_v), kk));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_v), kk));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_v), kk));
        // rr = cvRound(x*b + y*a) + 18 + oy, qq = cvRound(x*a - y*b) + 18 + poff (contraction order of oracle/orb_extractor_oracle.cc:
        // fma(x, b, y*a) and fma(x, a, -(y*b)); y * (-b) is -(y*b) bit for bit), both taps of a test at once as packed fp32.
        // cvRound + offset in one subtract: the magic-number rounding leaves the integer in the low mantissa bits.  Then the
        // block-indexed address 160 (rr >> 2) + 4 (rr & 3) + 16 (qq >> 2) + (qq & 3) = 4 rr + 144 (rr >> 2) + qq + 12 (qq >> 2) as
        // two v_mad_u32_u24 and a v_lshl_add.  The wave's slot base rides in qq: the slots of wave w start 6400 w bytes into
        // `patch`, and qq + 1600 w contributes 12 (400 w) + 1600 w = 6400 w to the sum (1600 w is a multiple of 4: the shift
        // takes it whole), so the address is relative to the array and keypoint kk's slot is an immediate offset.
        static_assert((kKpPerWave * kBlkSlot) % 16 == 0, "a wave's slots must start on a multiple of 16 bytes for the base to ride in qq");
        const uint8_t* bc = &patch[0][0] + kk * kBlkSlot;
        const int r_bias = 0x4B400000 - (RQ[kk] >> 8), q_bias = 0x4B400000 - (RQ[kk] & 255) - wave * (kKpPerWave * kBlkSlot / 4);
        const f32x2 a2 = {a, a}, b2 = {b, b}, nb2 = {-b, -b};
        unsigned long long word[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            f32x2 rv, qv;
            if (kTap == 1) {        // second product fused: fma(y, a, x*b), fma(-y, b, x*a)  (y * (-b) is -(y*b) bit for bit)
                rv = __builtin_elementwise_fma(paty[w], a2, patx[w] * b2);
                qv = __builtin_elementwise_fma(paty[w], nb2, patx[w] * a2);
            } else if (kTap == 2) { // no contraction: (x*b) + (y*a), (x*a) - (y*b)  (-ffp-contract=off keeps these apart)
                rv = patx[w] * b2 + paty[w] * a2;
                qv = patx[w] * a2 + paty[w] * nb2;
            } else {                // first product fused: fma(x, b, y*a), fma(x, a, -(y*b))
                rv = __builtin_elementwise_fma(patx[w], b2, paty[w] * a2);
                qv = __builtin_elementwise_fma(patx[w], a2, paty[w] * nb2);
            }
            const f32x2 rf = pk_add_bcast(rv, kRoundMagic);
            const f32x2 qf = pk_add_bcast(qv, kRoundMagic);
            int tv[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t rr = (uint32_t)(__float_as_int(rf[j]) - r_bias), qq = (uint32_t)(__float_as_int(qf[j]) - q_bias);
                const uint32_t off = (rr << 2) + mad24s(rr >> 2, 144u, mad24s(qq >> 2, 12u, qq));
                tv[j] = bc[off];
            }
            word[w] = __ballot(tv[0] < tv[1]);
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
// Copies between pinned host memory and device memory done by the compute units instead of the SDMA engines, for the
// per-frame calls: a frame's transfers are 0.1-1 MB, where an SDMA copy costs ~8-10 us of fixed latency before it moves a byte
// and another ~10 us before the next command of the stream starts (profiles/round4_frame_trace.txt: three uploads = 68 us in
// front of the first kernel).  A kernel that reads / writes the mapped host pointer (hipHostMalloc memory is device accessible)
// queues behind and in front of the other kernels of the stream like any launch (~1.5 us boundaries) and moves the bytes at PCIe
// rate.  16 bytes per lane, a grid-stride loop; both pointers 16-byte aligned; the last bytes % 16 bytes go one by one (the first
// lanes of workgroup 0), so exactly `bytes` bytes are read and written whatever follows them in either buffer.
__global__ __launch_bounds__(256) void blit16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, unsigned tail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail)
        reinterpret_cast<uint8_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t*>(src + n16)[threadIdx.x];
}
// A per-frame transfer between a PINNED host block and device memory on stream s: the copy kernel when both pointers are 16-byte
// aligned (whole hipMalloc / hipHostMalloc blocks and 16-byte offsets into them), hipMemcpyAsync otherwise — or always under MSORB_FRAME_COPIES=sdma (read once per process: the
// A/B switch of this choice).
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    static const bool sdma = [] { const char* e = getenv("MSORB_FRAME_COPIES"); return e && std::string(e) == "sdma"; }();
    if (bytes == 0) return hipSuccess;
    if (sdma || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) return hipMemcpyAsync(dst, src, bytes, kind, s);
    launch_blit(dst, src, bytes, s);
    return hipSuccess;
}
void launch_blit(void* dst, const void* src, size_t bytes, hipStream_t s) {
    const size_t n16 = bytes / 16;
    if (bytes == 0) return;
    const int blocks = (int)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(blit16_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src), n16,
                       (unsigned)(bytes & 15));
}
void launch_stage_level0(const LevelView& src, uint8_t* dst, int dst_pitch, size_t dst_image_stride, int n_images, hipStream_t s) {
    hipLaunchKernelGGL(stage_level0_kernel, dim3((src.w + 1023) / 1024, src.h, n_images), dim3(256), 0, s, src.base, (size_t)src.pitch,
                       src.img_stride, dst, dst_pitch, dst_image_stride, src.w);
}
// magic with mulhi(n, magic) == n / d for every n < total, or 0 when there is none of this form (the kernel divides instead)
static uint32_t exact_div_magic(unsigned d, unsigned long long total) {
    if (d < 2 || total >= 0x100000000ull) return 0;
    const uint32_t magic = (uint32_t)((0x100000000ull + d - 1) / d);
    const unsigned long long e = (unsigned long long)magic * d - 0x100000000ull;  // < d
    return e * total >= 0x100000000ull ? 0 : magic;
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
    // the band kernel parks the source rows of a band of R output rows: at most floor((R - 1) * scale) + 3 of them (<= 12),
    // and decodes the 4 taps of a lane out of an 8-byte window (horizontal scale <= 1.25); batches only (a frame or two are
    // launch latency, not throughput: the one-row kernel has the shorter dependent chain)
    const double sy = (double)src.h / (double)dst.h, sx = (double)src.w / (double)dst.w;
    const bool band_ok = aligned && n_images >= 16 && (int)std::floor((R - 1) * sy) + 3 <= 12 && sx <= 1.25;
    if (band_ok) {   // two waves per workgroup: small workgroups find room sooner beside the other batch's kernels
        const unsigned gx = (unsigned)(dst.w + 255) / 256, gy = (unsigned)(dst.h + 2 * R - 1) / (2 * R);
        const unsigned long long total = (unsigned long long)gx * gy * (unsigned)n_images;
        hipLaunchKernelGGL((pyr_resize_bandreg_kernel<R, 12, 2>), dim3(gx, gy, n_images), dim3(128), 0,
                           s, src, dst, dst_base, tx, ty, exact_div_magic(gx * gy, total), exact_div_magic(gx, (unsigned long long)gx * gy));
    }
    else if (aligned) hipLaunchKernelGGL(pyr_resize_aligned_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
    else hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty, 0);
}
// ------------------------------------------------------------------------------------------------
// The pyramid of a frame as one launch (TowerPlan, orb_device.h).  grid = (tile x, tile y, image), 1024 threads.
// A level's region of a tile lives in LDS as rows of `pitch` bytes starting at column need0 (a multiple of 4: dword columns of
// the region are dword columns of the plane).  The arithmetic of an output dword is pyr_resize_aligned_kernel's, its two source
// rows read from LDS instead of global memory.
// ------------------------------------------------------------------------------------------------
struct TowerTaps { const ResizeTap* x[kMaxLevels]; const ResizeTap* y[kMaxLevels]; };
enum { kTpX0 = 0, kTpX1, kTpY0, kTpY1, kTpOx0, kTpOx1, kTpOy0, kTpOy1, kTpW, kTpH, kTpPitch, kTpPlaneLo, kTpPlaneHi, kTpTapXLo, kTpTapXHi, kTpTapYLo, kTpTapYHi, kTpCount };
__device__ __forceinline__ int tower_div(int i, int wd, uint32_t magic) { return wd == 1 ? i : (int)__umulhi((uint32_t)i, magic); }   // i / wd for i, wd < 2^16, magic = ceil(2^32 / wd)
#ifndef MSORB_TOWER_THREADS
#define MSORB_TOWER_THREADS 1024
#endif
constexpr int kTowerThreads = MSORB_TOWER_THREADS;
__global__ __launch_bounds__(kTowerThreads) void pyr_tower_kernel(PyramidView pyr, TowerPlan plan, TowerTaps taps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tower_lds[];
    __shared__ int lvp[kMaxLevels][kTpCount];   // per level: what this tile needs to know about it
    const int tx = blockIdx.x, ty = blockIdx.y, img = blockIdx.z, tid = threadIdx.x;
    uint8_t* const buf_even = tower_lds;
    uint8_t* const buf_odd = tower_lds + plan.lds_even;
    uint2* const tap_lds = reinterpret_cast<uint2*>(tower_lds + plan.lds_even + plan.lds_odd);
    // The per-level parameters sit in kernel-argument memory: indexed by a level that is only known at run time they would be
    // fetched level by level, each fetch a scalar-cache miss in front of that level's work (24 us for seven levels whatever the
    // tile size).  Thread l fetches level l's once; everybody reads them from LDS.
    if (tid < pyr.nlevels) {
        const int l = tid;
        const LevelView& v = pyr.lv[l];
        int* q = lvp[l];
        q[kTpX0] = plan.x[l].need0[tx]; q[kTpX1] = plan.x[l].need1[tx]; q[kTpY0] = plan.y[l].need0[ty]; q[kTpY1] = plan.y[l].need1[ty];
        q[kTpOx0] = tower_own_x(tx, plan.ntx, v.w); q[kTpOx1] = tower_own_x(tx + 1, plan.ntx, v.w);
        q[kTpOy0] = tower_own_y(ty, plan.nty, v.h); q[kTpOy1] = tower_own_y(ty + 1, plan.nty, v.h);
        q[kTpW] = v.w; q[kTpH] = v.h; q[kTpPitch] = v.pitch;
        const uintptr_t plane = reinterpret_cast<uintptr_t>(v.base) + (size_t)img * v.img_stride;
        q[kTpPlaneLo] = (int)(uint32_t)plane; q[kTpPlaneHi] = (int)(uint32_t)(plane >> 32);
        const uintptr_t px = reinterpret_cast<uintptr_t>(taps.x[l]), py = reinterpret_cast<uintptr_t>(taps.y[l]);
        q[kTpTapXLo] = (int)(uint32_t)px; q[kTpTapXHi] = (int)(uint32_t)(px >> 32);
        q[kTpTapYLo] = (int)(uint32_t)py; q[kTpTapYHi] = (int)(uint32_t)(py >> 32);
    }
    __syncthreads();
    auto ptr_of = [&](int l, int lo) { return (uintptr_t)(uint32_t)lvp[l][lo] | ((uintptr_t)(uint32_t)lvp[l][lo + 1] << 32); };
    // Everything that comes from global memory is requested before anything is waited for (a workgroup is alone on its CU: a
    // load it waits for is ~1 us during which nothing else happens): the taps of the tile's regions of all levels — segment
    // 2 (l - 1) = the x taps [need0, need1) of level l, 2 (l - 1) + 1 its y taps — and the tile's region of level 0.
    int seg_begin[2 * kMaxLevels + 1];
    seg_begin[0] = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) {
        const bool on = l < pyr.nlevels;
        seg_begin[2 * l - 1] = seg_begin[2 * l - 2] + (on ? lvp[l][kTpX1] - lvp[l][kTpX0] : 0);
        seg_begin[2 * l] = seg_begin[2 * l - 1] + (on ? (lvp[l][kTpY1] - lvp[l][kTpY0] + 1) & ~1 : 0);   // (even: the x taps are read four at a time, 16-byte aligned)
    }
    const int n_taps = seg_begin[2 * kMaxLevels - 2];
    constexpr int kTapRounds = 2048 / kTowerThreads;   // x kTowerThreads >= the taps of a tile (checked by the host: build_tower_plan)
    uint2 tap_v[kTapRounds];
#pragma unroll
    for (int k = 0; k < kTapRounds; k++) {
        const int i = tid + k * kTowerThreads;
        int lsel = 1, axis = 0, off = 0;
#pragma unroll
        for (int l = 1; l < kMaxLevels; l++) {
            if (l < pyr.nlevels && i >= seg_begin[2 * l - 2]) { lsel = l; axis = 0; off = seg_begin[2 * l - 2]; }
            if (l < pyr.nlevels && i >= seg_begin[2 * l - 1]) { lsel = l; axis = 1; off = seg_begin[2 * l - 1]; }
        }
        const ResizeTap* src = reinterpret_cast<const ResizeTap*>(ptr_of(lsel, axis ? kTpTapYLo : kTpTapXLo)) + lvp[lsel][axis ? kTpY0 : kTpX0];
        tap_v[k] = i < n_taps ? *reinterpret_cast<const uint2*>(src + (i - off)) : uint2{0u, 0u};
    }
    {
        const int x0 = lvp[0][kTpX0], x1 = lvp[0][kTpX1], y0 = lvp[0][kTpY0], y1 = lvp[0][kTpY1], pitch = lvp[0][kTpPitch];
        const int wd = (x1 - x0) >> 2, lp = (x1 - x0) + 8, total = wd * (y1 - y0);
        const uint32_t magic = wd > 1 ? (uint32_t)(0xFFFFFFFFu / (uint32_t)wd) + 1u : 0u;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(ptr_of(0, kTpPlaneLo));
        constexpr int kStage = 4096 / kTowerThreads;
        for (int i0 = 0; i0 < total; i0 += kStage * kTowerThreads) {
            uint32_t val[kStage];
            int dst[kStage];
#pragma unroll
            for (int k = 0; k < kStage; k++) {
                const int i = i0 + k * kTowerThreads + tid;
                const int r = tower_div(i, wd, magic), c = i - r * wd;
                dst[k] = i < total ? r * lp + 4 * c : -1;
                val[k] = i < total ? *reinterpret_cast<const uint32_t*>(src + (size_t)(y0 + r) * pitch + x0 + 4 * c) : 0u;
            }
#pragma unroll
            for (int k = 0; k < kStage; k++)
                if (dst[k] >= 0) *reinterpret_cast<uint32_t*>(buf_even + dst[k]) = val[k];
        }
    }
#pragma unroll
    for (int k = 0; k < kTapRounds; k++)
        if (tid + k * kTowerThreads < n_taps) tap_lds[tid + k * kTowerThreads] = tap_v[k];
    __syncthreads();
    int xb = 0;   // this level's x taps in tap_lds (its y taps follow them)
    for (int l = 1; l < pyr.nlevels; l++) {
        const uint8_t* sbuf = (l & 1) ? buf_even : buf_odd;
        uint8_t* dbuf = (l & 1) ? buf_odd : buf_even;
        const int sx0 = lvp[l - 1][kTpX0], sy0 = lvp[l - 1][kTpY0];
        const int sp = (lvp[l - 1][kTpX1] - sx0) + 8;
        const int x0 = lvp[l][kTpX0], x1 = lvp[l][kTpX1], y0 = lvp[l][kTpY0], y1 = lvp[l][kTpY1];
        const int ox0 = lvp[l][kTpOx0], ox1 = lvp[l][kTpOx1], oy0 = lvp[l][kTpOy0], oy1 = lvp[l][kTpOy1];
        const int dw = lvp[l][kTpW], dpitch = lvp[l][kTpPitch];
        const int wd = (x1 - x0) >> 2, dp = (x1 - x0) + 8, total = wd * (y1 - y0);
        const uint32_t magic = wd > 1 ? (uint32_t)(0xFFFFFFFFu / (uint32_t)wd) + 1u : 0u;
        const uint2* __restrict__ tapx = tap_lds + xb;               // entry j = tap of column x0 + j
        const uint2* __restrict__ tapy = tap_lds + xb + (x1 - x0);   // entry j = tap of row y0 + j
        xb += (x1 - x0) + ((y1 - y0 + 1) & ~1);
        uint8_t* dplane = reinterpret_cast<uint8_t*>(ptr_of(l, kTpPlaneLo));
        for (int i = tid; i < total; i += kTowerThreads) {
            const int r = tower_div(i, wd, magic), c = i - r * wd;
            const int dy = y0 + r, dx0 = x0 + 4 * c;
            const uint2 vyw = tapy[r];
            const int vy_i0 = (int)(vyw.x & 0xffffu), vy_i1 = (int)(vyw.x >> 16);
            const int b0 = (int)(int16_t)(vyw.y & 0xffffu), b1 = (int)(int16_t)(vyw.y >> 16);
            const uint4 ta = reinterpret_cast<const uint4*>(tapx + 4 * c)[0];
            const uint4 tb = reinterpret_cast<const uint4*>(tapx + 4 * c)[1];
            const uint32_t tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};  // per tap: {i0|i1<<16, c0|c1<<16}
            const int base = (int)(tw[0] & 0xffffu) & ~3;  // aligned column of the first source pixel
            const uint8_t* s0 = sbuf + (vy_i0 - sy0) * sp + (base - sx0);
            const uint8_t* s1 = sbuf + (vy_i1 - sy0) * sp + (base - sx0);
            const uint32_t a0 = reinterpret_cast<const uint32_t*>(s0)[0], a1 = reinterpret_cast<const uint32_t*>(s0)[1],
                           a2 = reinterpret_cast<const uint32_t*>(s0)[2];
            const uint32_t b0w = reinterpret_cast<const uint32_t*>(s1)[0], b1w = reinterpret_cast<const uint32_t*>(s1)[1],
                           b2w = reinterpret_cast<const uint32_t*>(s1)[2];
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i0 = (int)(tw[2 * k] & 0xffffu), i1 = (int)(tw[2 * k] >> 16);
                const int c0 = (int)(int16_t)(tw[2 * k + 1] & 0xffffu), c1 = (int)(int16_t)(tw[2 * k + 1] >> 16);
                const int o = i0 - base;
                const uint32_t pa = pick2(a0, a1, a2, o), pb = pick2(b0w, b1w, b2w, o);
                const int sh = (i1 != i0) ? 8 : 0;   // second tap = next pixel, except at the right edge where i1 == i0 (and c1 == 0)
                const int h0 = (int)(pa & 255u) * c0 + (int)((pa >> sh) & 255u) * c1;
                const int h1 = (int)(pb & 255u) * c0 + (int)((pb >> sh) & 255u) * c1;
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                if (dx0 + k < dw) packed |= (uint32_t)(v & 255) << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(dbuf + r * dp + 4 * c) = packed;
            if (dy >= oy0 && dy < oy1 && dx0 >= ox0 && dx0 < ox1) *reinterpret_cast<uint32_t*>(dplane + (size_t)dy * dpitch + dx0) = packed;
        }
        __syncthreads();
    }
}

// Regions top-down: the top level's tiles are its `own` ranges; a level's region = its own range and whatever the region of the
// level above reads of it (first tap of the first pixel .. second tap of the last).  x ranges grow to dword columns.
bool build_tower_plan(TowerPlan& plan, int nlevels, const int* w, const int* h, const int* pitch, const std::vector<std::vector<ResizeTap>>& taps_x,
                      const std::vector<std::vector<ResizeTap>>& taps_y, size_t lds_limit) {
    plan = TowerPlan{};
    if (nlevels < 2) return false;
    const int ntx = kTowerTiles, nty = w[0] >= 2 * h[0] ? kTowerTiles / 2 : kTowerTiles;
    for (int l = 0; l < nlevels; l++)
        if (pitch[l] < ((w[l] + 3) & ~3)) return false;
    size_t lds[2] = {0, 0};
    for (int tyi = 0; tyi < kTowerTiles; tyi++)
        for (int l = nlevels - 1; l >= 0; l--) {
            int a = l ? tower_own_y(tyi, nty, h[l]) : 32767, b = l ? tower_own_y(tyi + 1, nty, h[l]) : -1;
            if (l + 1 < nlevels && plan.y[l + 1].need1[tyi] > plan.y[l + 1].need0[tyi]) {
                const auto& t = taps_y[l + 1];
                a = std::min<int>(a, t[plan.y[l + 1].need0[tyi]].i0);
                b = std::max<int>(b, t[std::min<int>(plan.y[l + 1].need1[tyi], h[l + 1]) - 1].i1 + 1);
            }
            if (b <= a) { a = 0; b = 0; }
            plan.y[l].need0[tyi] = (int16_t)a; plan.y[l].need1[tyi] = (int16_t)b;
        }
    for (int txi = 0; txi < kTowerTiles; txi++)
        for (int l = nlevels - 1; l >= 0; l--) {
            int a = l ? tower_own_x(txi, ntx, w[l]) : 32767, b = l ? tower_own_x(txi + 1, ntx, w[l]) : -1;
            if (l + 1 < nlevels && plan.x[l + 1].need1[txi] > plan.x[l + 1].need0[txi]) {
                const auto& t = taps_x[l + 1];
                a = std::min<int>(a, t[std::min<int>(plan.x[l + 1].need0[txi], w[l + 1] - 1)].i0);
                // (an output dword reads three source dwords from the aligned column of its first tap: up to 11 bytes past it)
                b = std::max<int>(b, (t[std::min<int>(plan.x[l + 1].need1[txi], w[l + 1]) - 1].i1 + 1));
            }
            if (b <= a) { a = 0; b = 0; }
            a &= ~3; b = (b + 3) & ~3;
            plan.x[l].need0[txi] = (int16_t)a; plan.x[l].need1[txi] = (int16_t)b;
        }
    for (int l = 0; l < nlevels; l++)
        for (int tyi = 0; tyi < nty; tyi++)
            for (int txi = 0; txi < ntx; txi++) {
                const size_t bytes = (size_t)(plan.x[l].need1[txi] - plan.x[l].need0[txi] + 8) * (plan.y[l].need1[tyi] - plan.y[l].need0[tyi]) + 16;
                lds[l & 1] = std::max(lds[l & 1], (bytes + 15) & ~size_t(15));
            }
    size_t n_taps = 0;   // taps of a tile's regions, all levels (the kernel fetches them in two rounds of 1024)
    for (int tyi = 0; tyi < nty; tyi++)
        for (int txi = 0; txi < ntx; txi++) {
            size_t t = 0;
            for (int l = 1; l < nlevels; l++) t += (size_t)(plan.x[l].need1[txi] - plan.x[l].need0[txi]) + ((plan.y[l].need1[tyi] - plan.y[l].need0[tyi] + 1) & ~1);
            n_taps = std::max(n_taps, t);
        }
    if (n_taps > 2048 || lds[0] + lds[1] + (n_taps + 8) * sizeof(ResizeTap) > lds_limit) return false;
    plan.ntx = ntx; plan.nty = nty;
    plan.lds_even = (int)lds[0]; plan.lds_odd = (int)lds[1]; plan.lds_taps = (int)((n_taps + 8) * sizeof(ResizeTap));
    return true;
}
bool launch_pyramid_tower(const PyramidView& pyr, const TowerPlan& plan, const ResizeTap* taps, const size_t* tap_x_off, const size_t* tap_y_off,
                          int n_images, hipStream_t s) {
    static const bool off = getenv("MSORB_PYR_TOWER") && atoi(getenv("MSORB_PYR_TOWER")) == 0;
    const LevelView& v0 = pyr.lv[0];   // level 0 may be the caller's buffer: dword rows, readable up to the next multiple of 4 past w
    if (off || plan.ntx <= 0 || (reinterpret_cast<uintptr_t>(v0.base) & 3) || (v0.pitch & 3) || (v0.img_stride & 3) || v0.pitch < ((v0.w + 3) & ~3) ||
        (reinterpret_cast<uintptr_t>(taps) & 31))
        return false;
    const size_t lds = (size_t)plan.lds_even + plan.lds_odd + plan.lds_taps;
    if ((long long)lds > dynamic_lds_room(reinterpret_cast<const void*>(pyr_tower_kernel))) return false;   // (the room already excludes the kernel's static __shared__)
    TowerTaps t{};
    for (int l = 1; l < pyr.nlevels; l++) { t.x[l] = taps + tap_x_off[l]; t.y[l] = taps + tap_y_off[l]; }
    hipLaunchKernelGGL(pyr_tower_kernel, dim3(plan.ntx, plan.nty, n_images), dim3(kTowerThreads), lds, s, pyr, plan, t);
    return true;
}
// ComputePyramid for a batch: levels 1 .. n-1, each from the one above (ORBextractor.cc:1179-1193), one launch per level.
void launch_pyramid(const PyramidView& pyr, const ResizeTap* taps, const size_t* tap_x_off, const size_t* tap_y_off, int n_images,
                    hipStream_t s, const Semantics& sem) {
    for (int l = 1; l < pyr.nlevels; l++)
        launch_pyr_resize(pyr.lv[l - 1], pyr.lv[l], const_cast<uint8_t*>(pyr.lv[l].base), taps + tap_x_off[l], taps + tap_y_off[l], n_images, s,
                          sem.resize_single_stage);
}
void launch_fast_cells(const PyramidView& pyr, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                       int slots_per_image, Cand16* slots, int* cell_count, int n_images, bool small_cells, hipStream_t s) {
    bool aligned = true;
    for (int l = 0; l < pyr.nlevels; l++) {
        const LevelView& v = pyr.lv[l];
        aligned = aligned && (reinterpret_cast<uintptr_t>(v.base) & 3) == 0 && (v.pitch & 3) == 0 && (v.img_stride & 3) == 0;
    }
    const dim3 grid(n_cells, n_images);
    const uint32_t gx_magic = exact_div_magic((unsigned)n_cells, (unsigned long long)n_cells * (unsigned)n_images);
#define MSORB_FAST_LAUNCH(AL, GEO)                                                                                        \
    hipLaunchKernelGGL((fast_cells_kernel<AL, GEO>), grid, dim3(GEO::kThreads), 0, s, pyr, cells, ini_th, min_th, slots_per_image, slots, \
                       cell_count, n_cells, gx_magic)
    if (small_cells) { if (aligned) MSORB_FAST_LAUNCH(true, GeoSmall); else MSORB_FAST_LAUNCH(false, GeoSmall); }
    else { if (aligned) MSORB_FAST_LAUNCH(true, GeoLarge); else MSORB_FAST_LAUNCH(false, GeoLarge); }
#undef MSORB_FAST_LAUNCH
}
// packed: the images' candidate runs follow each other without gaps (one contiguous read-back for the host quadtree: a scan
// over the images in a launch of its own); otherwise image i's run starts at i * slots_per_image — two launches instead of three
void launch_cand_compact(const CellDesc* cells, int n_cells, const int* level_cell_begin, int nlevels,
                         int slots_per_image, const Cand16* slots, const int* cell_count, int* cell_off,
                         int* level_count, int* img_total, int* img_base, Cand16* compact, int n_images,
                         hipStream_t s, bool packed, bool frame_form) {
    if (!packed && frame_form && n_images <= 4 && n_cells <= kFrameCompactCells) {   // a frame: scan + gather as one launch
        hipLaunchKernelGGL(cand_compact_frame_kernel, dim3((n_cells + kGatherCells - 1) / kGatherCells, n_images), dim3(kGatherThreads), 0, s, cells, n_cells,
                           level_cell_begin, nlevels, slots_per_image, slots, cell_count, cell_off, level_count, img_total, img_base, compact);
        return;
    }
    hipLaunchKernelGGL(cand_scan_cells_kernel, dim3(n_images), dim3(256), 0, s, cell_count, n_cells, level_cell_begin,
                       nlevels, cell_off, level_count, img_total, img_base, packed ? 0 : slots_per_image);
    if (packed) hipLaunchKernelGGL(cand_scan_images_kernel, dim3(1), dim3(256), 0, s, img_total, n_images, img_base);
    hipLaunchKernelGGL(cand_gather_kernel, dim3((n_cells + kGatherCells - 1) / kGatherCells, n_images), dim3(kGatherThreads), 0, s, cells, n_cells, slots_per_image,
                       slots, cell_count, cell_off, img_base, compact);
}
static bool levels_aligned(const PyramidView& v) {
    bool aligned = true;
    for (int l = 0; l < v.nlevels; l++) {
        const LevelView& lv = v.lv[l];
        aligned = aligned && (reinterpret_cast<uintptr_t>(lv.base) & 3) == 0 && (lv.pitch & 3) == 0 && (lv.img_stride & 3) == 0;
    }
    return aligned;
}
// The blur kernels' block table: per level bx_count column blocks x ceil(strips / 4) row blocks of four wave-strips.
static BlurPlan make_blur_plan(const PyramidView& src, int n_images, bool stream, int* total_out, int* max_h_out) {
    BlurPlan plan{};
    plan.nlevels = src.nlevels;
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
    for (int l = 0; l < src.nlevels; l++)
        plan.bx_magic[l] = exact_div_magic((unsigned)plan.bx_count[l], (unsigned long long)(plan.block_begin[l + 1] - plan.block_begin[l]));
    for (int l = src.nlevels; l < kMaxLevels; l++) plan.bx_magic[l] = 0;
    plan.image_magic = exact_div_magic((unsigned)total, (unsigned long long)total * (unsigned)n_images + 8);
    *total_out = total;
    *max_h_out = max_h;
    return plan;
}
int launch_gauss7(const PyramidView& src, const PyramidView& dst, int n_images, hipStream_t s, const Semantics& sem) {
    const bool aligned = levels_aligned(src);
    // the streaming kernel has the default taps folded into its v_dot4 constants; other taps (Semantics::gauss_taps) take the
    // generic kernels, which read them at run time
    const bool stream = aligned && sem.default_taps();
    GaussTaps T;
    for (int i = 0; i < 7; i++) T.k[i] = (uint32_t)sem.gauss_taps[i];
    int total = 0, max_h = 0;
    const BlurPlan plan = make_blur_plan(src, n_images, stream, &total, &max_h);
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
bool make_frame_blur_job(const PyramidView& src, const PyramidView& dst, int n_images, const Semantics& sem, FrameBlurJob* job) {
    if (!levels_aligned(src) || !sem.default_taps()) return false;
    int total = 0, max_h = 0;
    job->src = src; job->dst = dst;
    job->plan = make_blur_plan(src, n_images, true, &total, &max_h);
    job->blocks = total * n_images;
    return true;
}
// FAST + blur of a frame as one launch (frame_fast_blur_kernel); false: the conditions of the one-launch form do not hold (rows
// not 4-byte aligned, non-default Gaussian taps) and nothing was launched — the caller issues the two launches.
bool launch_frame_fast_blur(const PyramidView& pyr, const PyramidView& blur, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                            int slots_per_image, Cand16* slots, int* cell_count, int n_images, bool small_cells, hipStream_t s,
                            const Semantics& sem) {
    if (!levels_aligned(pyr) || !sem.default_taps()) return false;
    int total = 0, max_h = 0;
    const BlurPlan plan = make_blur_plan(pyr, n_images, true, &total, &max_h);
    const int blur_blocks = total * n_images;
    const unsigned n_fast = (unsigned)n_cells * (unsigned)n_images;
    const uint32_t gx_magic = exact_div_magic((unsigned)n_cells, (unsigned long long)n_cells * (unsigned)n_images);
    if (small_cells)
        hipLaunchKernelGGL((frame_fast_blur_kernel<true, GeoSmall>), dim3(n_fast + (unsigned)blur_blocks * (256 / GeoSmall::kThreads)), dim3(GeoSmall::kThreads), 0, s,
                           pyr, cells, ini_th, min_th, slots_per_image, slots, cell_count, n_cells, gx_magic, n_fast, blur, plan, blur_blocks);
    else
        hipLaunchKernelGGL((frame_fast_blur_kernel<true, GeoLarge>), dim3(n_fast + (unsigned)blur_blocks * (256 / GeoLarge::kThreads)), dim3(GeoLarge::kThreads), 0, s,
                           pyr, cells, ini_th, min_th, slots_per_image, slots, cell_count, n_cells, gx_magic, n_fast, blur, plan, blur_blocks);
    return true;
}
void launch_describe(const PyramidView& pyr, const PyramidView& blur, const SelRec* sel, const int* sel_count,
                     int sel_stride, const LevelScale& scales, msorb_keypoint* kps, uint8_t* desc, int out_stride,
                     int max_sel, int n_images, hipStream_t s, const Semantics& sem) {
    if (max_sel <= 0) return;
    const unsigned gx = (unsigned)((max_sel + kDescWaves * kKpPerWave - 1) / (kDescWaves * kKpPerWave));
    const uint32_t magic = exact_div_magic(gx, (unsigned long long)gx * (unsigned)n_images);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(gx, n_images), dim3(64 * kDescWaves), 0, s, pyr, blur, sel, sel_count, sel_stride, scales, kps, desc,
                           out_stride, sem.atan2_fma, magic);
    };
    if (sem.brief_tap == 1) go(describe_kernel<1>);
    else if (sem.brief_tap == 2) go(describe_kernel<2>);
    else go(describe_kernel<0>);
}

}  // namespace msorb
