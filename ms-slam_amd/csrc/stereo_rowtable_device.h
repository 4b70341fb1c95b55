// vRowIndices of Frame::ComputeStereoMatches (Frame.cc:757-776) as a CSR over the rows of level 0, built by ONE workgroup:
// for every right keypoint iR the rows [floor(y - r), ceil(y + r)] with r = 2 * scale[octave] receive the entry
// {iR | octave << 24, bits of x}.  Shared by stereo_rowtable_kernel (matcher.hip: keys from the extractor's keypoint array) and
// the stereo frame's selection-layout kernel (quadtree_kernels.hip: keys straight from the selection records, so the table of a
// frame is built beside the layout instead of in a kernel of its own behind the descriptor stage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msorb {

// rt: LDS, 2 * rows0 + 1 ints.  key(iR, x, y, octave) -> the right keypoint iR in level-0 coordinates (kp.pt, kp.octave).
// Every thread of the T-thread workgroup calls this (it contains __syncthreads).
template <int T, class KeyFn>
__device__ __forceinline__ void stereo_rowtable_build(int* rt, int t, int rows0, int nR, const float* __restrict__ scale,
                                                      int* __restrict__ row_begin, int2* __restrict__ row_list, int row_cap, KeyFn key) {
    int* cnt = rt;             // [rows0] counts -> cursors
    int* beg = rt + rows0;     // [rows0 + 1] begins
    for (int r = t; r < rows0; r += T) cnt[r] = 0;
    // each right keypoint's row band, computed once: the first kBandCache rounds of the T-strided loop keep it in
    // registers (all their loads in flight together), later rounds (more than 2048 right keypoints) recompute it
    constexpr int kBandCache = 2048 / T;
    int band[kBandCache];    // minr | maxr << 16, -1 = no keypoint
    int2 entry[kBandCache];  // the keypoint's table entry: {iR | octave << 24, bits of x}
    auto band_of = [&](int iR, int2& e) -> int {
        float x, y;
        int octave;
        key(iR, x, y, octave);
        e = int2{iR | (octave << 24), __float_as_int(x)};
        const float r = __fmul_rn(2.0f, scale[octave]);
        const int maxr = min((int)ceilf(__fadd_rn(y, r)), rows0 - 1), minr = max((int)floorf(__fsub_rn(y, r)), 0);
        return maxr >= minr ? (minr | (maxr << 16)) : -1;
    };
#pragma unroll
    for (int k = 0; k < kBandCache; k++) {
        const int iR = t + k * T;
        band[k] = iR < nR ? band_of(iR, entry[k]) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBandCache; k++)
        if (band[k] >= 0)
            for (int y = band[k] & 0xffff; y <= (band[k] >> 16); y++) atomicAdd(&cnt[y], 1);
    for (int iR = t + kBandCache * T; iR < nR; iR += T) {
        int2 e;
        const int bd = band_of(iR, e);
        if (bd >= 0)
            for (int y = bd & 0xffff; y <= (bd >> 16); y++) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    // exclusive scan of cnt over the rows (rows0 is a few hundred: one wave, sequential chunks)
    if (t < 64) {
        int carry = 0;
        for (int base = 0; base < rows0; base += 64) {
            const int r = base + t;
            const int v = r < rows0 ? cnt[r] : 0;
            int inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(inc, off);
                if (t >= off) inc += u;
            }
            if (r < rows0) beg[r] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (t == 0) beg[rows0] = carry;
    }
    __syncthreads();
    for (int r = t; r <= rows0; r += T) row_begin[r] = beg[r];
    for (int r = t; r < rows0; r += T) cnt[r] = beg[r];
    __syncthreads();
    auto fill = [&](const int2& e, int bd) {
        for (int y = bd & 0xffff; y <= (bd >> 16); y++) {
            const int pos = atomicAdd(&cnt[y], 1);
            if (pos < row_cap) row_list[pos] = e;
        }
    };
#pragma unroll
    for (int k = 0; k < kBandCache; k++)
        if (band[k] >= 0) fill(entry[k], band[k]);
    for (int iR = t + kBandCache * T; iR < nR; iR += T) {
        int2 e;
        const int bd = band_of(iR, e);
        if (bd >= 0) fill(e, bd);
    }
}

// A stereo FRAME does without the table: the selection-layout kernel (quadtree_kernels.hip) writes one BAND record per right
// keypoint — the rows [floor(y - r), ceil(y + r)] it would be entered in (Frame.cc:757-776), its octave and its x — and the
// association of a left keypoint tests every right keypoint's band against its row (stereo_match_one, matcher.hip): 2 000 x 2 000
// 8-byte tests cost the match kernel ~2 us, the table (counts, scan, scattered fill, by one workgroup) cost the frame 13.
struct StereoRowJob {
    int right_img;      // image of the launch whose keypoints are the right eye's (1 for a stereo frame)
    int rows0;          // rows of level 0 (< 4096)
    int2* band;         // [capacity]: {minr | maxr << 12 | octave << 24, bits of x}; minr > maxr = no row
    int* level_begin;   // [kMaxLevels + 1]: first record of each octave (the association scans the three octaves it can accept)
    int* n_oob;         // zeroed here (the association's out-of-bounds counter), may be nullptr
};
__device__ __forceinline__ int2 stereo_band_record(float x, float y, int octave, const float* __restrict__ scale, int rows0) {
    const float r = __fmul_rn(2.0f, scale[octave]);
    const int maxr = min((int)ceilf(__fadd_rn(y, r)), rows0 - 1), minr = max((int)floorf(__fsub_rn(y, r)), 0);
    return int2{(maxr >= minr ? (minr | (maxr << 12)) : 1) | (octave << 24), __float_as_int(x)};
}

}  // namespace msorb
