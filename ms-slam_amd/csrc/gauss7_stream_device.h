// The streaming form of the 7x7 Gaussian (ORBextractor.cc:1132-1133: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in OpenCV's
// Q8.8 fixed point, taps [18, 34, 48, 56, 48, 34, 18]) as DEVICE FUNCTIONS: one wave blurs one strip, waves are independent (no
// barrier, no LDS), so the body can run as a kernel of its own (orb_kernels.hip gauss7_stream_kernel: batches), beside FAST's cells
// (frame_fast_blur_kernel) or beside the keypoint selection of a frame (quadtree_kernels.hip: the selection keeps 16 of the chip's
// 256 CUs busy for 35-40 us — the blur of the frame's levels, which only the descriptor stage reads, runs on the others).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"

namespace msorb {

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

constexpr int kGaussRows = 28;  // 6 warm-up rows + 4 x 7 steady rows; a multiple of 4: the blurred plane is written in blocks of four rows
struct GaussTaps { uint32_t k[7]; };   // the generic blur kernels take the Q8 taps at run time (Semantics::gauss_taps)

struct BlurPlan {
    int block_begin[kMaxLevels + 1];  // first blockIdx.x of each level
    int bx_count[kMaxLevels];         // blocks per strip-row of the level
    int nlevels;
    uint32_t bx_magic[kMaxLevels];    // exact_div_magic(bx_count[l], blocks of the level), 0 = divide
    uint32_t image_magic;             // exact_div_magic(blocks per image, blocks of the launch)
};

__device__ __forceinline__ int refl101(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); }

// Aligned fast path of the blur: the same strips, but a lane loads only ITS dword of a row and takes the 4 bytes on
// either side from the neighbouring lanes (DPP wave shifts) — one global load per row instead of three — and the loads
// of the next seven rows are in flight while the current seven are accumulated.  Lanes 0 and 63 of a wave only provide
// halo bytes: a wave stores 62 four-pixel groups per row.
constexpr int kGaussLanesOut = 62;
// One strip of the streaming blur.  EDGE = the column block holds the group at x0 == 0 or the right border (reflect-101
// fix-ups by byte permutation); the common interior blocks are compiled without them (a block-uniform `if` inside the row
// loop is if-converted into per-row v_cndmask work by the compiler, so the two cases are separate instantiations).
// Per row and lane: 2 DPP moves (neighbour dwords), 10 v_dot4 (the 7 taps of the 4 pixels against the three dwords
// {w0,w1,w2} with the kernel shifted inside the constants: no v_alignbyte), then the vertical pass in integers (below): 4 packs,
// 12 v_dot2_u32_u16, 4 v_mad_u32_u24, and for the output row 3 v_perm — 35 instructions per four pixels of a row (51 with the
// fp32 accumulators of rounds 1-3).  Exactly OpenCV's fixed-point arithmetic: u16 row sums, u32 column sums, (x + 2^15) >> 16.
template <int ROWS, bool EDGE>
__device__ __forceinline__ void gauss7_strip(const uint8_t* __restrict__ sb, uint8_t* __restrict__ db, int src_pitch, int dst_pitch,
                                             int h, int y0, uint32_t xl, int x0, bool store, uint32_t sel_w1, uint32_t sel_w2a,
                                             uint32_t sel_w2b) {
    static_assert(ROWS % 28 == 0, "strips start on a multiple of 4 rows and the row loop is unrolled over 4 x 7 rows");
    uint8_t* const dblk = db + blur_tile_off((uint32_t)max(x0, 0), 0u, (uint32_t)dst_pitch);   // the lane's 4 x 4 block column
    uint32_t blk[4] = {0u, 0u, 0u, 0u};   // packed output rows of the block being filled (a level's last block may be flushed part-filled: rows past h are zero)
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
    auto row_sums = [&](uint32_t w1, uint32_t hs[4]) {
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
        hs[0] = __builtin_amdgcn_udot4(w1, B0, __builtin_amdgcn_udot4(w0, A0, 0u, false), false);
        hs[1] = __builtin_amdgcn_udot4(w2, C1, __builtin_amdgcn_udot4(w1, B1, __builtin_amdgcn_udot4(w0, A1, 0u, false), false), false);
        hs[2] = __builtin_amdgcn_udot4(w2, C2, __builtin_amdgcn_udot4(w1, B2, __builtin_amdgcn_udot4(w0, A2, 0u, false), false), false);
        hs[3] = __builtin_amdgcn_udot4(w2, C3, __builtin_amdgcn_udot4(w1, B3, 0u, false), false);
    };
    // Vertical pass in integers, two taps per instruction: a row sum is < 2^16 (255 * 256), so consecutive rows of a column
    // travel as one register P_r = h_r | h_{r+1} << 16 and output row o = 2^15 + dot2(P_o, k0 k1) + dot2(P_{o+2}, k2 k3) +
    // dot2(P_{o+4}, k4 k5) + k6 h_{o+6}: per input row and pixel 1 pack + 3 v_dot2_u32_u16 + 1 v_mad_u32_u24 (rounds 1-3 ran 7
    // fp32 FMAs + a conversion here).  Accumulator o % 7 is opened by row o + 1 and closed by row o + 6.
    const ushort2v K01 = {(unsigned short)k0, (unsigned short)k1}, K23 = {(unsigned short)k2, (unsigned short)k3},
                   K45 = {(unsigned short)k2, (unsigned short)k1};
    uint32_t acc[7][4];
    uint32_t hs[4], hp[4] = {0, 0, 0, 0};
    uint32_t warm[6], nxt[7];
#pragma unroll
    for (int r = 0; r < 6; r++) warm[r] = load_row(r);
#pragma unroll
    for (int u = 0; u < 7; u++) nxt[u] = load_row(6 + u);
    // warm-up: input rows 0..5
#pragma unroll
    for (int r = 0; r < 6; r++) {
        row_sums(warm[r], hs);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (r >= 1) {
                const ushort2v P = __builtin_bit_cast(ushort2v, hp[j] | (hs[j] << 16));
                acc[(r - 1) % 7][j] = __builtin_amdgcn_udot2(P, K01, 32768u, false);
                if (r >= 3) acc[(r - 3) % 7][j] = __builtin_amdgcn_udot2(P, K23, acc[(r - 3) % 7][j], false);
                if (r >= 5) acc[(r - 5) % 7][j] = __builtin_amdgcn_udot2(P, K45, acc[(r - 5) % 7][j], false);
            }
            hp[j] = hs[j];
        }
    }
    // steady state: input row r = 6 + 7*it + u completes output row o = r - 6
#pragma unroll
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
            row_sums(cur[u], hs);
            uint32_t q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const ushort2v P = __builtin_bit_cast(ushort2v, hp[j] | (hs[j] << 16));
                q[j] = __umul24(hs[j], k0) + acc[u % 7][j];                                      // (r - 6) % 7 == u: closes row o
                acc[(5 + u) % 7][j] = __builtin_amdgcn_udot2(P, K01, 32768u, false);             // (r - 1) % 7: opens row r - 1
                acc[(3 + u) % 7][j] = __builtin_amdgcn_udot2(P, K23, acc[(3 + u) % 7][j], false);
                acc[(1 + u) % 7][j] = __builtin_amdgcn_udot2(P, K45, acc[(1 + u) % 7][j], false);
                hp[j] = hs[j];
            }
            const int o = r - 6;
            if (y0 + o < h) {   // wave-uniform
                // (x + 2^15) >> 16 of a value < 2^24: byte 2 of each sum
                const uint32_t packed = __builtin_amdgcn_perm(__builtin_amdgcn_perm(q[3], q[2], 0x0c0c0602u),
                                                              __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0602u), 0x05040100u);
                // blocked plane (orb_device.h blur_tile_off): four output rows of the lane's columns are one 16-byte store, 8 lanes
                // one full line; y0 is a multiple of 4, so o & 3 is the row inside the block.  The level's last rows flush a
                // partly filled block (the rows below h in it are never read).
                blk[o & 3] = packed;
                if ((o & 3) == 3 || y0 + o == h - 1) {
                    if (store) *reinterpret_cast<uint4*>(dblk + (uint32_t)((y0 + o) >> 2) * ((uint32_t)dst_pitch * 4u)) = make_uint4(blk[0], blk[1], blk[2], blk[3]);
                }
            }
        }
    }
}

// One wave's strip: `tile` = the 256-thread block the wave belongs to in the streaming kernel's numbering, wave_in_block = 0..3
// (waves are independent: no barrier, no LDS).
template <int ROWS>
__device__ __forceinline__ void gauss7_stream_body(const PyramidView& src, const PyramidView& dst, const BlurPlan& plan, const int tile, const int wave_in_block) {
    const int blocks_per_image = plan.block_begin[plan.nlevels];
    const int img = plan.image_magic ? (int)__umulhi((uint32_t)tile, plan.image_magic) : tile / blocks_per_image, blk = tile - img * blocks_per_image;
    int level = 0;
    while (level + 1 < plan.nlevels && blk >= plan.block_begin[level + 1]) level++;
    const int rem = blk - plan.block_begin[level];
    const int by = plan.bx_magic[level] ? (int)__umulhi((uint32_t)rem, plan.bx_magic[level]) : rem / plan.bx_count[level];
    const int bx = rem - by * plan.bx_count[level];
    const LevelView sv = src.lv[level], dv = dst.lv[level];
    const int lane = threadIdx.x & 63;
    const int x0 = (bx * kGaussLanesOut + lane - 1) * 4;
    // the wave index through readfirstlane: everything derived from y0 (row reflection, row offsets, the row bound of the
    // stores) is then scalar work
    const int y0 = (by * 4 + __builtin_amdgcn_readfirstlane(wave_in_block)) * ROWS;
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
    if (border_block || bx == 0)   // block-uniform
        gauss7_strip<ROWS, true>(sb, db, sv.pitch, dv.pitch, sv.h, y0, xl, x0, store, sel_w1, sel_w2a, sel_w2b);
    else
        gauss7_strip<ROWS, false>(sb, db, sv.pitch, dv.pitch, sv.h, y0, xl, x0, store, sel_w1, sel_w2a, sel_w2b);
}

// The blur of a frame's levels as a job another launch can carry (quadtree_kernels.hip): block b of `blocks` is the 256-thread
// block b of gauss7_stream_kernel's numbering (four wave-strips).  make_frame_blur_job (orb_kernels.hip): false = the streaming
// form does not serve this call (rows not 4-byte aligned, non-default Gaussian taps).
struct FrameBlurJob {
    PyramidView src, dst;
    BlurPlan plan;
    int blocks = 0;
};
bool make_frame_blur_job(const PyramidView& src, const PyramidView& dst, int n_images, const Semantics& sem, FrameBlurJob* job);

}  // namespace msorb
