// Hamming matching kernels + the flat C ABI that ORBmatcher::SearchBy* and Frame::ComputeStereoMatches
// call into (include/msorb.h, "Matcher" section).  Integer / bit work: XOR + popcount, wave64 reductions.
//
//   window_topk_kernel   GetFeaturesInArea + best/second argmin   Frame.cc:589-655, ORBmatcher.cc:84-120, 2006-2033
//   list_top2_kernel     explicit candidate lists (BoW searches)   ORBmatcher.cc:288-330 idiom
//   stereo_match_kernel  row-band argmin + 11-offset SAD           Frame.cc:743-897
//
// Tie-breaks: the reference scans candidates sequentially with strict '<'.  That equals a lexicographic
// minimum over (distance, position-in-scan), so every lane keeps packed 64-bit keys
// dist<<40 | scan position and the wave reduces with min — exact, order independent.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msorb.h"
#include "matcher_device.h"
#include "stereo_rowtable_device.h"

namespace msorb {

__device__ __forceinline__ int hamming256(const uint64_t a[4], const uint64_t* __restrict__ b) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The same reductions with DPP row operations only (no LDS crossbar round trip per step): xor-butterfly inside each row of 16
// lanes, then the row results are chained through lanes 15 / 31 into row 3; lane 63 holds the result, returned wave-uniform.
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140 /* row_mirror */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t x) {
    int v = (int)x;
    auto mn = [](int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); };
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // lanes outside the row mask keep their own value
    v = mn(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

constexpr uint64_t kNoKey = ~0ull;

// per-lane sorted insertion into a 4-deep (key, idx) list
__device__ __forceinline__ void topk_insert(uint64_t key, int idx, uint64_t k[kTopK], int id[kTopK]) {
    if (key >= k[kTopK - 1]) return;
    k[kTopK - 1] = key; id[kTopK - 1] = idx;
#pragma unroll
    for (int i = kTopK - 1; i > 0; i--) {
        if (k[i] < k[i - 1]) {
            const uint64_t tk = k[i]; k[i] = k[i - 1]; k[i - 1] = tk;
            const int ti = id[i]; id[i] = id[i - 1]; id[i - 1] = ti;
        }
    }
}

// One group of kWinLanes lanes (16 = a DPP row, or 4) per query.  Walks the query's grid window exactly like
// Frame::GetFeaturesInArea: cells ix (outer) / iy (inner) ascending, cell contents in insertion order; group lane c owns
// window cell c, c+16, ...  (A tracking window covers 6-20 cells with about one keypoint each: with a whole wave per query
// three quarters of the lanes had no cell; per 64 frames x 4096 queries 0.67 -> 0.2 ms.)
// blockIdx.y = frame of a batch (frame_stride keypoints / q_stride queries apart in every array; 0 / 0 for one frame).
// kCount: the number of Hamming distances evaluated is added to *n_eval (measurement runs only).
template <int kWinLanes>
__device__ __forceinline__ uint64_t group_min_u64(uint64_t v) {
#pragma unroll
    for (int o = kWinLanes / 2; o > 0; o >>= 1) {
        const uint64_t t = __shfl_xor(v, o, kWinLanes);
        v = t < v ? t : v;
    }
    return v;
}
// kGate: F.gate_kp holds the keypoints the level band and the kQFuseGate test read (msorb_fuse_search_gated); a template parameter so
// that the searches of every frame pay nothing for it.
template <bool kCount, int kWinLanes, bool kGate = false>
__global__ __launch_bounds__(256) void window_topk_kernel(FrameView F, const WinQuery* __restrict__ q,
                                                          const uint8_t* __restrict__ qdesc, int q_begin, int q_end,
                                                          TopK* __restrict__ out, int frame_stride, int q_stride,
                                                          unsigned long long* __restrict__ n_eval) {
    constexpr int kGroups = 256 / kWinLanes;
    const int qi_raw = q_begin + blockIdx.x * kGroups + (threadIdx.x / kWinLanes);
    const int lane = threadIdx.x & (kWinLanes - 1);
    const int gshift = (threadIdx.x & 63) & ~(kWinLanes - 1);  // first wave lane of this group
    const bool live = qi_raw < q_end;
    const int qi = live ? qi_raw : q_end - 1;
    if (blockIdx.y) {
        const size_t fo = (size_t)blockIdx.y * frame_stride, qo = (size_t)blockIdx.y * q_stride;
        F.kp += fo; F.desc += fo * 32; F.cell_idx += fo; F.occupied += fo;
        F.cell_begin += (size_t)blockIdx.y * (kGridCols * kGridRows + 1);
        q += qo; qdesc += qo * 32; out += qo;
    }
    int n_pairs = 0;
    const WinQuery Q = q[qi];
    uint64_t k[kTopK];
    int id[kTopK];
#pragma unroll
    for (int i = 0; i < kTopK; i++) { k[i] = kNoKey; id[i] = -1; }
    bool any = live && (Q.flags & kQValid) != 0;
    int minCX = 0, maxCX = -1, minCY = 0, maxCY = -1;
    if (any) {  // Frame.cc:597-619
        minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.x, F.minX), Q.r), F.gridWInv)));
        maxCX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.x, F.minX), Q.r), F.gridWInv)));
        minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.y, F.minY), Q.r), F.gridHInv)));
        maxCY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.y, F.minY), Q.r), F.gridHInv)));
        any = !(minCX >= kGridCols || maxCX < 0 || minCY >= kGridRows || maxCY < 0);
    }
    if (any) {
        const uint64_t* qd = reinterpret_cast<const uint64_t*>(qdesc + (size_t)qi * 32);
        const uint64_t a[4] = {qd[0], qd[1], qd[2], qd[3]};
        const int ncy = maxCY - minCY + 1;
        const int ncell = (maxCX - minCX + 1) * ncy;
        const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
        for (int c = lane; c < ncell; c += kWinLanes) {
            const int ix = minCX + c / ncy, iy = minCY + c % ncy;
            const int cell = ix * kGridRows + iy;
            const int b = F.cell_begin[cell], e = F.cell_begin[cell + 1];
            for (int j = b; j < e; j++) {
                const int idx = F.cell_idx[j];
                const KpLite kp = F.kp[idx];
                const KpLite g = kGate ? F.gate_kp[idx] : kp;
                if (check_levels) {
                    if (g.octave < Q.min_level) continue;
                    if (Q.max_level >= 0 && g.octave > Q.max_level) continue;
                }
                if (!(fabsf(__fsub_rn(kp.x, Q.x)) < Q.r && fabsf(__fsub_rn(kp.y, Q.y)) < Q.r)) continue;
                if ((Q.flags & kQSkipOccupied) && F.occupied[idx]) continue;       // ORBmatcher.cc:88-90
                if (Q.flags & kQFuseGate) {  // reprojection-error gate of ORBmatcher::Fuse, ORBmatcher.cc:1520-1545
                    const float ex = __fsub_rn(Q.x, g.x), ey = __fsub_rn(Q.y, g.y);
                    float e2 = __fmaf_rn(ex, ex, __fmul_rn(ey, ey));
                    double lim = 5.99;
                    if (g.u_right >= 0) {
                        const float er = __fsub_rn(Q.ur, g.u_right);
                        e2 = __fmaf_rn(er, er, e2);
                        lim = 7.8;
                    }
                    if ((double)__fmul_rn(e2, F.inv_sigma2[g.octave]) > lim) continue;
                } else if (!(Q.flags & kQNoUr) && kp.u_right > 0 && fabsf(__fsub_rn(Q.ur, kp.u_right)) > Q.r) continue;  // :92-97
                const int d = hamming256(a, reinterpret_cast<const uint64_t*>(F.desc + (size_t)idx * 32));
                if (kCount) n_pairs++;
                topk_insert(((uint64_t)d << 40) | ((uint64_t)c << 20) | (uint64_t)(j - b), idx, k, id);
            }
        }
    }
    // merge the group's sorted lists: kTopK rounds of "global minimum head pops".  A group without candidates skips it
    // (wave-uniform only when all four groups are empty, but the rounds are cheap next to the gathers above).
    TopK res;
    const bool group_any = ((__ballot(k[0] != kNoKey) >> gshift) & ((1ull << kWinLanes) - 1)) != 0;
#pragma unroll
    for (int r = 0; r < kTopK; r++) {
        const uint64_t m = group_min_u64<kWinLanes>(k[0]);
        int widx = -1;
        if (m != kNoKey && k[0] == m) {  // keys are unique (scan position), exactly one lane of the group matches
            widx = id[0];
#pragma unroll
            for (int i = 0; i < kTopK - 1; i++) { k[i] = k[i + 1]; id[i] = id[i + 1]; }
            k[kTopK - 1] = kNoKey; id[kTopK - 1] = -1;
        }
        const unsigned owner = (unsigned)((__ballot(widx >= 0) >> gshift) & ((1ull << kWinLanes) - 1));
        const int src = owner ? __ffs((int)owner) - 1 : 0;
        res.idx[r] = owner ? __shfl(widx, src, kWinLanes) : -1;
        res.dist[r] = owner ? (int)(m >> 40) : 256;
    }
    (void)group_any;
    if (live && lane == 0) out[qi] = res;
    if (kCount) {
        const int tot = wave_sum_i32(n_pairs);
        if ((threadIdx.x & 63) == 0 && tot) atomicAdd(n_eval, (unsigned long long)tot);
    }
}

// Explicit candidate lists (CSR): best / second-best in list order.
__global__ __launch_bounds__(256) void list_top2_kernel(const uint8_t* __restrict__ qdesc, const uint8_t* __restrict__ tdesc,
                                                        const int* __restrict__ cand_begin, const int* __restrict__ cand_idx,
                                                        int n_queries, int* __restrict__ best_idx, int* __restrict__ best_dist,
                                                        int* __restrict__ second_idx, int* __restrict__ second_dist) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= n_queries) return;
    const uint64_t* qd = reinterpret_cast<const uint64_t*>(qdesc + (size_t)qi * 32);
    const uint64_t a[4] = {qd[0], qd[1], qd[2], qd[3]};
    uint64_t k0 = kNoKey, k1 = kNoKey;
    const int b = cand_begin[qi], e = cand_begin[qi + 1];
    for (int j = b + lane; j < e; j += 64) {
        const int idx = cand_idx[j];
        const int d = hamming256(a, reinterpret_cast<const uint64_t*>(tdesc + (size_t)idx * 32));
        const uint64_t key = ((uint64_t)d << 40) | (uint64_t)(j - b);
        if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key;
    }
    const uint64_t m0 = wave_min_u64(k0);
    if (k0 == m0 && m0 != kNoKey) { k0 = k1; k1 = kNoKey; }  // owner pops
    const uint64_t m1 = wave_min_u64(k0);
    if (lane == 0) {
        best_idx[qi] = m0 == kNoKey ? -1 : cand_idx[b + (int)(m0 & 0xffffffffffull)];
        best_dist[qi] = m0 == kNoKey ? 256 : (int)(m0 >> 40);
        second_idx[qi] = m1 == kNoKey ? -1 : cand_idx[b + (int)(m1 & 0xffffffffffull)];
        second_dist[qi] = m1 == kNoKey ? 256 : (int)(m1 >> 40);
    }
}

// Frame::ComputeStereoMatches (Frame.cc:743-897), one wave per left keypoint.
//  1. candidates = right keypoints whose row band [floor(y-r), ceil(y+r)], r = 2*scale[octave], contains
//     row (int)vL (the reference's vRowIndices table, ascending iR), octave within +-1, uR in [uL-maxD, uL];
//     best = lexicographic min (dist, iR), accepted if dist < TH_HIGH(100)            (:805-826)
//  2. if best < 75: L1 SAD of the 11x11 window at 11 horizontal offsets on the pyramid level, parabola
//     sub-pixel fit, disparity gates                                                  (:829-897)
// The median-based rejection (:899-912) is serial and done by the host on the SAD values written here.
// G = the geometry tables (kernel argument memory: indexed by level without a private copy), V = this image pair
struct StereoPairView {
    const msorb_keypoint *kpL, *kpR;
    const uint8_t *descL, *descR;
    int nR;
    float *u_right, *depth;
    int *sad, *n_oob;
    const int* row_begin;
    const int2* row_list;
    const int2* band;           // no row table: one band record per right keypoint (stereo frames), or nullptr: bands from the keypoints
    const int* band_level_begin;   // [MSORB_MAX_LEVELS + 1] first record of each octave (level-major order; = nR past the last level), or nullptr
    size_t img;                 // image index of the pair inside the batch arrays (0 for the per-frame call)
    const size_t *img_strideL, *img_strideR;  // per level, nullptr for the per-frame call
};
__device__ __forceinline__ void stereo_match_one(const StereoArgs& G, const StereoPairView& A, const int iL, const int lane, const bool active = true) {
    if (!active) return;
    const msorb_keypoint kpL = A.kpL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const int row = (int)vL;
    const float minD = 0.f, maxD = __fdiv_rn(G.mbf, G.mb);
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    // key = distance << 22 | iR (iR < 2^22): one 32-bit DPP min-reduction gives the reference's (distance, iR) minimum
    constexpr uint32_t kNoKey32 = 0xFFFFFFFFu;
    uint32_t best = kNoKey32;
    float best_x = 0.f;   // x of this lane's best candidate
    if (row >= 0 && row < G.rows0 && !(maxU < 0)) {
        const uint64_t* dl = reinterpret_cast<const uint64_t*>(A.descL + (size_t)iL * 32);
        const uint64_t a[4] = {dl[0], dl[1], dl[2], dl[3]};
        // candidates: the row's entry of the vRowIndices table (:760-770) when one was built (any order: the result is
        // the lexicographic minimum of (distance, iR)), else every right keypoint with the band test done here
        const int2* list = A.row_begin ? A.row_list + A.row_begin[row] : nullptr;
        const int n_cand = A.row_begin ? A.row_begin[row + 1] - A.row_begin[row] : A.nR;
        if (!list && A.band) {
            // Every right keypoint's band against this row (8 bytes each: the 2 000 of a KITTI frame are 16 KB per left keypoint, from
            // L2 / L1), four records per lane in flight; the few that pass row, octave and x range (the table's row entry after its
            // filters) are compacted into a per-wave LDS list by ballot, and the descriptor distances are taken over the dense list
            // — with the distance inside the scan every one of the 32 tests of a lane was its own divergent round of descriptor loads.
            __shared__ int2 cand_all[4][512];
            int2* const cand = cand_all[threadIdx.x >> 6];
            int n_list = 0;   // wave-uniform
            auto drain = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int j = lane; j < n_list; j += 64) {
                    const int2 e = cand[j];
                    const int d = hamming256(a, reinterpret_cast<const uint64_t*>(A.descR + (size_t)e.x * 32));
                    const uint32_t key = ((uint32_t)d << 22) | (uint32_t)e.x;
                    if (key < best) { best = key; best_x = __int_as_float(e.y); }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                n_list = 0;
            };
            // (the records are in level-major order and only octaves levelL - 1 .. levelL + 1 can pass: their stretch of the list)
            const int j_begin = A.band_level_begin ? A.band_level_begin[max(levelL - 1, 0)] : 0;
            const int j_end = A.band_level_begin ? min(A.band_level_begin[min(levelL + 2, MSORB_MAX_LEVELS)], n_cand) : n_cand;
            for (int j0 = j_begin; j0 < j_end; j0 += 256) {
                int2 e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int j = j0 + 64 * k + lane; e[k] = A.band[j < j_end ? j : 0]; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int iR = j0 + 64 * k + lane;
                    const int minr = e[k].x & 0xfff, maxr = (e[k].x >> 12) & 0xfff, octR = (int)((uint32_t)e[k].x >> 24);
                    const float xR = __int_as_float(e[k].y);
                    const bool pass = iR < j_end && row >= minr && row <= maxr && !(octR < levelL - 1 || octR > levelL + 1) && (xR >= minU && xR <= maxU);
                    const unsigned long long m = __ballot(pass);
                    if (pass) cand[n_list + __popcll(m & ((1ull << lane) - 1ull))] = int2{iR, e[k].y};
                    n_list += __popcll(m);
                }
                if (n_list > 256) drain();   // (room for the next four ballots)
            }
            drain();
        } else
        for (int j = lane; j < n_cand; j += 64) {
            int iR, octR;
            float xR;
            if (list) {   // the table entry carries what the filters need: one dependent load less per candidate
                const int2 e = list[j];
                iR = e.x & 0xffffff; octR = (int)((uint32_t)e.x >> 24); xR = __int_as_float(e.y);
            } else {
                iR = j;
                const msorb_keypoint kr = A.kpR[iR];
                const float r = __fmul_rn(2.0f, G.scale[kr.octave]);
                const int maxr = (int)ceilf(__fadd_rn(kr.y, r)), minr = (int)floorf(__fsub_rn(kr.y, r));
                if (row < minr || row > maxr) continue;
                octR = kr.octave; xR = kr.x;
            }
            if (octR < levelL - 1 || octR > levelL + 1) continue;
            if (!(xR >= minU && xR <= maxU)) continue;
            const int d = hamming256(a, reinterpret_cast<const uint64_t*>(A.descR + (size_t)iR * 32));
            const uint32_t key = ((uint32_t)d << 22) | (uint32_t)iR;
            if (key < best) { best = key; best_x = xR; }
        }
    }
    const uint32_t mine = best;
    best = wave_min_u32_dpp(best);
    const int bestDist = best == kNoKey32 ? 256 : (int)(best >> 22);
    if (bestDist < kThHigh && bestDist < (kThHigh + kThLow) / 2) {
        // x of the winner comes from the lane that holds it (keys are unique): no keypoint load behind the reduction
        const unsigned long long owner = __ballot(mine == best);
        const float uR0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_x), (int)__builtin_ctzll(owner)));
        const float sf = G.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, sf));
        const float scaledvL = roundf(__fmul_rn(kpL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const int cols = G.cols[levelL], rows = G.rows[levelL];
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w), xR0 = (int)(scaleduR0 - L - w);
        const bool ok = !(iniu < 0 || endu >= cols) && y0 >= 0 && y0 + 2 * w + 1 <= rows && xL0 >= 0 &&
                        xL0 + 2 * w + 1 <= cols && xR0 >= 0 && (int)(scaleduR0 + L + w + 1) <= cols;
        if (ok) {
            const uint8_t* pl = G.pyrL[levelL] + (A.img_strideL ? A.img * A.img_strideL[levelL] : 0) + (size_t)y0 * G.pitchL[levelL] + xL0;
            const uint8_t* pr = G.pyrR[levelL] + (A.img_strideR ? A.img * A.img_strideR[levelL] : 0) + (size_t)y0 * G.pitchR[levelL] + xR0;
            // The 11 x 11 left window and the 11 x 21 right strip go to LDS as whole dwords (3 load instructions per wave; the
            // byte-per-lane form needed 24, each a separate trip through the texture addresser — that, not the arithmetic,
            // was the kernel's bottleneck).  Then lane (row r = lane & 15, offset group lane >> 4) forms, in three passes, the
            // row SAD of offset inc = 4 pass + (lane >> 4) with three v_sad_u8 on byte-aligned dwords (v_alignbyte), and a
            // 4-step DPP add inside each row of 16 lanes sums the 11 window rows.
            __shared__ uint32_t win_all[4][11 * 12];   // per wave and window row: 8 right dwords, 4 left dwords
            uint32_t* win = win_all[threadIdx.x >> 6];
            const int pitchL = G.pitchL[levelL], pitchR = G.pitchR[levelL];
            const uint32_t phL = (uint32_t)(reinterpret_cast<uintptr_t>(pl) & 3), phR = (uint32_t)(reinterpret_cast<uintptr_t>(pr) & 3);
            // A dword may not reach past the end of the level plane of this image (level 0 can be the caller's own buffer): such a
            // dword is read 1..3 bytes (or more: then nothing of it is needed) earlier and shifted back, its missing top bytes —
            // beyond the plane, never part of a window — come out as zero.  With 4-byte aligned rows this never triggers.
            const uint8_t* planeL = G.pyrL[levelL] + (A.img_strideL ? A.img * A.img_strideL[levelL] : 0);
            const uint8_t* planeR = G.pyrR[levelL] + (A.img_strideR ? A.img * A.img_strideR[levelL] : 0);
            const uintptr_t limL = reinterpret_cast<uintptr_t>(planeL) + (size_t)rows * pitchL - 4;
            const uintptr_t limR = reinterpret_cast<uintptr_t>(planeR) + (size_t)rows * pitchR - 4;
            auto load_dword = [](uintptr_t a, uintptr_t lim) {
                const uintptr_t b = min(a, lim);
                const uint32_t v = *reinterpret_cast<const uint32_t*>(b);
                const uint32_t back = (uint32_t)(a - b);
                return back == 0 ? v : (back < 4 ? v >> (8 * back) : 0u);
            };
            {
                const int r = lane >> 3, c = lane & 7;
                const uintptr_t a0 = reinterpret_cast<uintptr_t>(pr - phR) + (size_t)r * pitchR + 4 * c;
                const uint32_t v0 = load_dword(a0, limR);                                          // rows 0..7 of the right strip
                uint32_t v1 = 0, v2 = 0;
                const uintptr_t a1 = reinterpret_cast<uintptr_t>(pr - phR) + (size_t)(r + 8) * pitchR + 4 * c;
                if (lane < 24) v1 = load_dword(a1, limR);                                           // rows 8..10
                const int rl = lane >> 2, cl = lane & 3;
                const uintptr_t a2 = reinterpret_cast<uintptr_t>(pl - phL) + (size_t)rl * pitchL + 4 * cl;
                if (lane < 44) v2 = load_dword(a2, limL);                                           // the left window
                win[r * 12 + c] = v0;
                if (lane < 24) win[(r + 8) * 12 + c] = v1;
                if (lane < 44) win[rl * 12 + 8 + cl] = v2;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int wr = lane & 15, grp = lane >> 4;
            const bool row_ok = wr < 11;
            const uint32_t* wrow = win + (row_ok ? wr : 0) * 12;
            const uint32_t l0 = __builtin_amdgcn_alignbyte(wrow[9], wrow[8], phL), l1 = __builtin_amdgcn_alignbyte(wrow[10], wrow[9], phL);
            const uint32_t l2 = __builtin_amdgcn_alignbyte(wrow[11], wrow[10], phL) & 0x00ffffffu;   // 11 bytes = 4 + 4 + 3
            int sad[11];
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                const int inc = 4 * pass + grp;                       // < 11 except group 3 of the last pass
                const uint32_t off = phR + (uint32_t)min(inc, 10), d0 = off >> 2;
                const uint32_t r0 = __builtin_amdgcn_alignbyte(wrow[d0 + 1], wrow[d0], off);
                const uint32_t r1 = __builtin_amdgcn_alignbyte(wrow[d0 + 2], wrow[d0 + 1], off);
                const uint32_t r2 = __builtin_amdgcn_alignbyte(wrow[d0 + 3], wrow[d0 + 2], off) & 0x00ffffffu;
                uint32_t sv = __builtin_amdgcn_sad_u8(l0, r0, 0u);
                sv = __builtin_amdgcn_sad_u8(l1, r1, sv);
                sv = __builtin_amdgcn_sad_u8(l2, r2, sv);
                int v = row_ok ? (int)sv : 0;
                v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x140 /* row_mirror */, 0xf, 0xf, false);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * pass + j < 11) sad[4 * pass + j] = __builtin_amdgcn_readlane(v, 16 * j);
            }
            int bestS = 0x7fffffff, bestinc = 0;
#pragma unroll
            for (int inc = 0; inc < 11; inc++)
                if (sad[inc] < bestS) { bestS = sad[inc]; bestinc = inc - L; }
            if (!(bestinc == -L || bestinc == L)) {
                float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int inc = 1; inc < 10; inc++)
                    if (inc == bestinc + L) { d1 = (float)sad[inc - 1]; d2 = (float)sad[inc]; d3 = (float)sad[inc + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(G.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        out_d = __fdiv_rn(G.mbf, disparity);
                        out_u = bestuR;
                        out_sad = bestS;
                    }
                }
            }
        } else if (lane == 0 && !(iniu < 0 || endu >= cols)) {
            atomicAdd(A.n_oob, 1);
        }
    }
    if (lane == 0) {
        A.u_right[iL] = out_u;
        A.depth[iL] = out_d;
        A.sad[iL] = out_sad;
    }
}

__global__ __launch_bounds__(256) void stereo_match_kernel(StereoArgs A) {
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= A.nL) return;
    StereoPairView V{};
    V.kpL = A.kpL; V.kpR = A.kpR; V.descL = A.descL; V.descR = A.descR; V.nR = A.nR;
    V.u_right = A.u_right; V.depth = A.depth; V.sad = A.sad; V.n_oob = A.n_oob;
    V.row_begin = A.row_begin; V.row_list = A.row_list;
    stereo_match_one(A, V, iL, threadIdx.x & 63);
}

// vRowIndices (Frame.cc:756-770) for one right image per workgroup, as CSR over the image rows: thread per right
// keypoint, LDS counters per row, block scan, fill.  The order inside a row is arbitrary (see stereo_match_one).
// T threads: 256 for batches (one workgroup per pair, many pairs), 1024 for a frame or two (the table of a single pair is
// latency: 17 -> ~9 us with four times the threads)
template <int T>
__global__ __launch_bounds__(T) void stereo_rowtable_kernel(StereoBatchArgs B) {
    extern __shared__ int rt[];  // [rows0] counts -> cursors, [rows0 + 1] begins
    const int pair = blockIdx.x, t = threadIdx.x;
    const int rows0 = B.A.rows0;
    const size_t img = (size_t)pair * B.pair_step;
    const int nR = B.countsR ? B.countsR[img] : B.A.nR;
    const msorb_keypoint* kpR = B.A.kpR + img * B.capacity;
    if (t == 0 && B.A.n_oob) B.A.n_oob[pair] = 0;   // the association's out-of-bounds counter starts here: no memset launch in front of a frame
    stereo_rowtable_build<T>(rt, t, rows0, nR, B.A.scale, B.row_begin + (size_t)pair * (rows0 + 1), B.row_list + (size_t)pair * B.row_cap, B.row_cap,
                             [&](int iR, float& x, float& y, int& octave) {
                                 const msorb_keypoint kr = kpR[iR];
                                 x = kr.x; y = kr.y; octave = kr.octave;
                             });
}

// The same for every stereo pair of a batch (pair p = images 2p / 2p+1 of msorb_extract_batch): blockIdx.y = pair.
__global__ __launch_bounds__(256) void stereo_match_batch_kernel(StereoBatchArgs B) {
    const int pair = blockIdx.y;
    const size_t img = (size_t)pair * B.pair_step;
    const int nL = B.countsL[img], nR = B.countsR[img];
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= nL) return;
    StereoPairView V{};
    V.kpL = B.A.kpL + img * B.capacity;
    V.kpR = B.A.kpR + img * B.capacity;
    V.descL = B.A.descL + img * B.capacity * 32;
    V.descR = B.A.descR + img * B.capacity * 32;
    V.nR = nR;
    V.u_right = B.A.u_right + (size_t)pair * B.capacity;
    V.depth = B.A.depth + (size_t)pair * B.capacity;
    V.sad = B.A.sad + (size_t)pair * B.capacity;
    V.n_oob = B.A.n_oob + pair;
    V.row_begin = B.row_begin ? B.row_begin + (size_t)pair * (B.A.rows0 + 1) : nullptr;
    V.row_list = B.row_list ? B.row_list + (size_t)pair * B.row_cap : nullptr;
    V.band = B.band ? B.band + (size_t)pair * B.capacity : nullptr;
    V.band_level_begin = B.band ? B.band_level_begin : nullptr;
    V.img = img;
    V.img_strideL = B.img_strideL; V.img_strideR = B.img_strideR;
    stereo_match_one(B.A, V, iL, threadIdx.x & 63, iL < nL);
}

// Batches: sixteen lanes per left keypoint, four keypoints per wave.  A keypoint's association is a chain of five dependent
// memory round trips (keypoint -> row bounds -> row entries -> right descriptors -> window rows) with a handful of
// instructions between them: at one keypoint per wave the kernel ran at full occupancy and still waited (0.235 ms per 128
// pairs); four keypoints per wave have four chains in flight for the same wave slot.  Same results as stereo_match_one:
// the best candidate is the lexicographic minimum of (distance, iR) whatever the order, the SAD / parabola arithmetic is the
// same.  Lane l of a group: candidates l, l + 16, ... of the row entry; then window row l (11 of 16 lanes) with the 11 offsets
// as static byte shifts of its own 32-byte strip row — no LDS staging — and a 4-step DPP row reduction of the row SADs
// (two 16-bit sums per register: a window SAD is < 2^15).
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t x) {   // minimum over the 16 lanes of a DPP row, in every lane
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));
    return x;
}
__device__ __forceinline__ uint32_t row16_sum_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false);
    return x;
}
// 16 bytes from a 4-byte aligned address without touching anything past `lim` (the address of the plane's last dword): the
// dwords beyond come out as the clamped dword shifted down (their bytes inside the plane) or zero — never part of a window
__device__ __forceinline__ void load16_in_plane(uintptr_t a, uintptr_t lim, uint32_t out[4]) {
    if (a + 12 <= lim) {
        const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(a);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uintptr_t ad = a + 4 * d, b = min(ad, lim);
            const uint32_t v = *reinterpret_cast<const uint32_t*>(b);
            const uint32_t back = (uint32_t)(ad - b);
            out[d] = back == 0 ? v : (back < 4 ? v >> (8 * back) : 0u);
        }
    }
}
__global__ __launch_bounds__(256) void stereo_match_quad_kernel(StereoBatchArgs B) {
    const StereoArgs& G = B.A;
    const int pair = blockIdx.y;
    const size_t img = (size_t)pair * B.pair_step;
    const int nL = B.countsL[img];
    const int l = threadIdx.x & 15, grp_lane0 = (int)(threadIdx.x & 63u) & ~15;
    const int iL = blockIdx.x * 16 + (int)(threadIdx.x >> 4);
    if (iL >= nL) return;   // whole groups leave: every DPP row below has all 16 lanes or none
    const msorb_keypoint* kpLp = G.kpL + img * B.capacity + iL;
    const uint8_t* descR = G.descR + img * B.capacity * 32;
    const int* row_begin = B.row_begin + (size_t)pair * (G.rows0 + 1);
    const int2* row_list = B.row_list + (size_t)pair * B.row_cap;
    const float uL = kpLp->x, vL = kpLp->y;
    const int levelL = kpLp->octave;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const int row = (int)vL;
    const float minD = 0.f, maxD = __fdiv_rn(G.mbf, G.mb);
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    constexpr uint32_t kNoKey32 = 0xFFFFFFFFu;
    uint32_t best = kNoKey32;
    float best_x = 0.f;
    if (row >= 0 && row < G.rows0 && !(maxU < 0)) {
        const uint4* dl = reinterpret_cast<const uint4*>(G.descL + (img * B.capacity + iL) * 32);
        const uint4 a0 = dl[0], a1 = dl[1];
        const int rb = row_begin[row], n_cand = row_begin[row + 1] - rb;
        const int2* list = row_list + rb;
        for (int base = 0; base < n_cand; base += 64) {   // four entries per lane and round, their loads issued together
            int2 e[4];
            bool pass[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = base + 16 * k + l;
                pass[k] = j < n_cand;
                e[k] = list[pass[k] ? j : 0];
            }
            uint4 d0[4], d1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int octR = (int)((uint32_t)e[k].x >> 24);
                const float xR = __int_as_float(e[k].y);
                pass[k] = pass[k] && !(octR < levelL - 1 || octR > levelL + 1) && (xR >= minU && xR <= maxU);
                const uint4* dr = reinterpret_cast<const uint4*>(descR + (size_t)(e[k].x & 0xffffff) * 32);
                if (pass[k]) { d0[k] = dr[0]; d1[k] = dr[1]; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!pass[k]) continue;
                const int d = __popc(a0.x ^ d0[k].x) + __popc(a0.y ^ d0[k].y) + __popc(a0.z ^ d0[k].z) + __popc(a0.w ^ d0[k].w) +
                              __popc(a1.x ^ d1[k].x) + __popc(a1.y ^ d1[k].y) + __popc(a1.z ^ d1[k].z) + __popc(a1.w ^ d1[k].w);
                const uint32_t key = ((uint32_t)d << 22) | (uint32_t)(e[k].x & 0xffffff);
                if (key < best) { best = key; best_x = __int_as_float(e[k].y); }
            }
        }
    }
    const uint32_t mine = best;
    best = row16_min_u32(best);
    const int bestDist = best == kNoKey32 ? 256 : (int)(best >> 22);
    if (bestDist < kThHigh && bestDist < (kThHigh + kThLow) / 2) {
        // x of the winner from the lane of the group that holds it (keys are unique)
        const unsigned long long owner = __ballot(mine == best) >> grp_lane0;
        const int src = grp_lane0 + (int)__builtin_ctz((uint32_t)owner & 0xffffu);
        const float uR0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(best_x)));
        const float sf = G.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(uL, sf));
        const float scaledvL = roundf(__fmul_rn(vL, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const int cols = G.cols[levelL], rows = G.rows[levelL];
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w), xR0 = (int)(scaleduR0 - L - w);
        const bool ok = !(iniu < 0 || endu >= cols) && y0 >= 0 && y0 + 2 * w + 1 <= rows && xL0 >= 0 &&
                        xL0 + 2 * w + 1 <= cols && xR0 >= 0 && (int)(scaleduR0 + L + w + 1) <= cols;
        if (ok) {
            const int pitchL = G.pitchL[levelL], pitchR = G.pitchR[levelL];
            const uint8_t* planeL = G.pyrL[levelL] + img * B.img_strideL[levelL];
            const uint8_t* planeR = G.pyrR[levelL] + img * B.img_strideR[levelL];
            const uintptr_t limL = reinterpret_cast<uintptr_t>(planeL) + (size_t)rows * pitchL - 4;
            const uintptr_t limR = reinterpret_cast<uintptr_t>(planeR) + (size_t)rows * pitchR - 4;
            const int wr = min(l, 10);   // lanes 11..15 repeat row 10 and count for nothing
            const uintptr_t pl = reinterpret_cast<uintptr_t>(planeL + (size_t)(y0 + wr) * pitchL + xL0);
            const uintptr_t pr = reinterpret_cast<uintptr_t>(planeR + (size_t)(y0 + wr) * pitchR + xR0);
            const uint32_t phL = (uint32_t)(pl & 3), phR = (uint32_t)(pr & 3);
            uint32_t lw[4], rw[8];
            load16_in_plane(pl - phL, limL, lw);
            load16_in_plane(pr - phR, limR, rw);
            load16_in_plane(pr - phR + 16, limR, rw + 4);
            // byte 0 of the normalised rows = first pixel of the window row / strip row
            const uint32_t l0 = __builtin_amdgcn_alignbyte(lw[1], lw[0], phL), l1 = __builtin_amdgcn_alignbyte(lw[2], lw[1], phL);
            const uint32_t l2 = __builtin_amdgcn_alignbyte(lw[3], lw[2], phL) & 0x00ffffffu;   // 11 bytes = 4 + 4 + 3
            uint32_t rn[7];
#pragma unroll
            for (int d = 0; d < 7; d++) rn[d] = __builtin_amdgcn_alignbyte(rw[d + 1], rw[d], phR);   // 28 bytes >= 21
            uint32_t rs[11];
#pragma unroll
            for (int inc = 0; inc < 11; inc++) {
                const int d0 = inc >> 2, sh = inc & 3;
                const uint32_t r0 = __builtin_amdgcn_alignbyte(rn[d0 + 1], rn[d0], sh);
                const uint32_t r1 = __builtin_amdgcn_alignbyte(rn[d0 + 2], rn[d0 + 1], sh);
                const uint32_t r2 = __builtin_amdgcn_alignbyte(d0 + 3 < 7 ? rn[d0 + 3] : 0u, rn[d0 + 2], sh) & 0x00ffffffu;
                uint32_t sv = __builtin_amdgcn_sad_u8(l0, r0, 0u);
                sv = __builtin_amdgcn_sad_u8(l1, r1, sv);
                sv = __builtin_amdgcn_sad_u8(l2, r2, sv);
                rs[inc] = l < 11 ? sv : 0u;
            }
            int sad[11];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t t = row16_sum_u32(rs[2 * k] | (rs[2 * k + 1] << 16));
                sad[2 * k] = (int)(t & 0xffffu); sad[2 * k + 1] = (int)(t >> 16);
            }
            sad[10] = (int)row16_sum_u32(rs[10]);
            int bestS = 0x7fffffff, bestinc = 0;
#pragma unroll
            for (int inc = 0; inc < 11; inc++)
                if (sad[inc] < bestS) { bestS = sad[inc]; bestinc = inc - L; }
            if (!(bestinc == -L || bestinc == L)) {
                float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int inc = 1; inc < 10; inc++)
                    if (inc == bestinc + L) { d1 = (float)sad[inc - 1]; d2 = (float)sad[inc]; d3 = (float)sad[inc + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(G.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        out_d = __fdiv_rn(G.mbf, disparity);
                        out_u = bestuR;
                        out_sad = bestS;
                    }
                }
            }
        } else if (l == 0 && !(iniu < 0 || endu >= cols)) {
            atomicAdd(G.n_oob + pair, 1);
        }
    }
    if (l == 0) {
        const size_t o = (size_t)pair * B.capacity + iL;
        G.u_right[o] = out_u;
        G.depth[o] = out_d;
        G.sad[o] = out_sad;
    }
}

// Frame.cc:899-912 for one pair per workgroup: median = vDistIdx[size/2].first of the ascending (SAD, iL) list — the
// value of rank size/2 — found by a two-level histogram select (SAD <= 121*255 < 2^15); every match whose SAD is not
// below thDist = 1.5f*1.4f*median is withdrawn (the reference walks the sorted list from the end and stops at the first
// smaller value: the same set).
__device__ __forceinline__ void stereo_median_body(const int pair, const int* __restrict__ countsL, const int* __restrict__ countsR,
                                                   int pair_step, int capacity,
                                                   const int* __restrict__ sad_all, float* __restrict__ u_right_all,
                                                   float* __restrict__ depth_all, int* __restrict__ counts_out) {
    __shared__ int hist[256];
    __shared__ int sel[3];  // chosen high bin, rank inside it, number of valid SADs
    const int t = threadIdx.x;
    const int nL = countsL[(size_t)pair * pair_step];
    if (counts_out && t < 2) counts_out[2 * pair + t] = t ? countsR[(size_t)pair * pair_step] : nL;
    const int* sad = sad_all + (size_t)pair * capacity;
    float* u_right = u_right_all + (size_t)pair * capacity;
    float* depth = depth_all + (size_t)pair * capacity;
    hist[t] = 0;
    constexpr int kSadCache = 8;  // the first 2048 SADs of the pair stay in registers for all three passes
    int sv[kSadCache];
#pragma unroll
    for (int k = 0; k < kSadCache; k++) { const int i = t + k * 256; sv[k] = i < nL ? sad[i] : -1; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSadCache; k++)
        if (sv[k] >= 0) atomicAdd(&hist[sv[k] >> 7], 1);
    for (int i = t + kSadCache * 256; i < nL; i += 256) {
        const int v = sad[i];
        if (v >= 0) atomicAdd(&hist[v >> 7], 1);
    }
    __syncthreads();
    // rank search over 256 bins by all threads: inclusive prefix (wave shuffles + wave totals), the bin whose prefix
    // interval holds rank total/2 announces itself
    __shared__ int wave_tot[4];
    auto find_bin = [&](int nbins, int k_or_half) {  // k_or_half < 0: rank = total / 2
        const int v = t < nbins ? hist[t] : 0;
        int inc = v;
        const int lane = t & 63, w = t >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wave_tot[w] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int c = wave_tot[i]; if (i < w) before += c; total += c; }
        const int k = k_or_half < 0 ? total / 2 : k_or_half;
        const int excl = before + inc - v;
        if (v > 0 && excl <= k && k < excl + v) { sel[0] = t; sel[1] = k - excl; }
        if (t == 0) sel[2] = total;
        __syncthreads();
    };
    find_bin(256, -1);
    const int hb = sel[0], kk = sel[1], total = sel[2];
    if (total == 0) return;  // vDistIdx empty: nothing to reject (the reference would index an empty vector)
    __syncthreads();
    hist[t] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSadCache; k++)
        if (sv[k] >= 0 && (sv[k] >> 7) == hb) atomicAdd(&hist[sv[k] & 127], 1);
    for (int i = t + kSadCache * 256; i < nL; i += 256) {
        const int v = sad[i];
        if (v >= 0 && (v >> 7) == hb) atomicAdd(&hist[v & 127], 1);
    }
    __syncthreads();
    find_bin(128, kk);
    const float median = (float)((hb << 7) | sel[0]);
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
#pragma unroll
    for (int k = 0; k < kSadCache; k++) {
        const int i = t + k * 256;
        if (sv[k] >= 0 && !((float)sv[k] < thDist)) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    }
    for (int i = t + kSadCache * 256; i < nL; i += 256) {
        const int v = sad[i];
        if (v >= 0 && !((float)v < thDist)) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    }
}

__global__ __launch_bounds__(256) void stereo_median_kernel(const int* __restrict__ countsL, const int* __restrict__ countsR,
                                                            int pair_step, int capacity,
                                                            const int* __restrict__ sad_all, float* __restrict__ u_right_all,
                                                            float* __restrict__ depth_all, int* __restrict__ counts_out) {
    stereo_median_body(blockIdx.x, countsL, countsR, pair_step, capacity, sad_all, u_right_all, depth_all, counts_out);
}

// A stereo frame without a sink: the median rule and the frame's read-back as ONE launch (the host is bound by its launches on
// this chain).  Workgroup 0 applies the rule and then copies what the rule touches — the block's tail from `n16_head` on:
// u_right, depth, counters —; the other workgroups copy the keypoints and descriptors in front of it, which the rule never
// touches.  No dependency between workgroups.  dst: the pinned block, src: the device block, whole 16-byte units.
__global__ __launch_bounds__(256) void stereo_median_readback_kernel(const int* __restrict__ countsL, const int* __restrict__ countsR,
                                                                     int pair_step, int capacity, const int* __restrict__ sad_all,
                                                                     float* u_right_all, float* depth_all, int* counts_out,
                                                                     uint4* __restrict__ dst, const uint4* src, size_t n16_head,
                                                                     size_t n16_all) {
    if (blockIdx.x == 0) {
        stereo_median_body(0, countsL, countsR, pair_step, capacity, sad_all, u_right_all, depth_all, counts_out);
        __threadfence_block();
        __syncthreads();
        for (size_t i = n16_head + threadIdx.x; i < n16_all; i += 256) dst[i] = src[i];
    } else {
        for (size_t i = (size_t)(blockIdx.x - 1) * 256 + threadIdx.x; i < n16_head; i += (size_t)(gridDim.x - 1) * 256) dst[i] = src[i];
    }
}

// Every candidate of a window with its distance, in the reference's scan order — for searches whose accept rule depends
// on what EARLIER queries did to the candidates (ORBmatcher::SearchForInitialization skips a train already matched at a
// smaller or equal distance, ORBmatcher.cc:791-792): the device evaluates all Hamming distances, the host replays the
// sequential rule on the lists.  One thread per query walks GetFeaturesInArea (Frame.cc:589-655) itself; FILL = false
// counts, FILL = true writes (index, distance) pairs at list_begin[q].
template <bool FILL>
__global__ __launch_bounds__(64) void window_list_kernel(FrameView F, const WinQuery* __restrict__ q, const uint8_t* __restrict__ qdesc,
                                                         int n, int* __restrict__ count, const int* __restrict__ list_begin,
                                                         int2* __restrict__ list) {
    const int qi = blockIdx.x * 64 + threadIdx.x;
    if (qi >= n) return;
    const WinQuery Q = q[qi];
    int m = 0;
    if (Q.flags & kQValid) {
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.x, F.minX), Q.r), F.gridWInv)));
        const int maxCX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.x, F.minX), Q.r), F.gridWInv)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.y, F.minY), Q.r), F.gridHInv)));
        const int maxCY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.y, F.minY), Q.r), F.gridHInv)));
        if (!(minCX >= kGridCols || maxCX < 0 || minCY >= kGridRows || maxCY < 0)) {
            const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
            uint64_t a[4] = {0, 0, 0, 0};
            int2* out = nullptr;
            if (FILL) {
                const uint64_t* qd = reinterpret_cast<const uint64_t*>(qdesc + (size_t)qi * 32);
                a[0] = qd[0]; a[1] = qd[1]; a[2] = qd[2]; a[3] = qd[3];
                out = list + list_begin[qi];
            }
            for (int ix = minCX; ix <= maxCX; ix++)
                for (int iy = minCY; iy <= maxCY; iy++) {
                    const int cell = ix * kGridRows + iy;
                    for (int j = F.cell_begin[cell]; j < F.cell_begin[cell + 1]; j++) {
                        const int idx = F.cell_idx[j];
                        const KpLite kp = F.kp[idx];
                        if (check_levels) {
                            if (kp.octave < Q.min_level) continue;
                            if (Q.max_level >= 0 && kp.octave > Q.max_level) continue;
                        }
                        if (!(fabsf(__fsub_rn(kp.x, Q.x)) < Q.r && fabsf(__fsub_rn(kp.y, Q.y)) < Q.r)) continue;
                        if (FILL) out[m] = make_int2(idx, hamming256(a, reinterpret_cast<const uint64_t*>(F.desc + (size_t)idx * 32)));
                        m++;
                    }
                }
        }
    }
    if (!FILL) count[qi] = m;
}
void launch_window_list(const FrameView& F, const WinQuery* q, const uint8_t* qdesc, int n, int* count, const int* list_begin,
                        int2* list, bool fill, hipStream_t s) {
    if (n <= 0) return;
    if (fill) hipLaunchKernelGGL(window_list_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, s, F, q, qdesc, n, count, list_begin, list);
    else hipLaunchKernelGGL(window_list_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, s, F, q, qdesc, n, count, list_begin, list);
}

void launch_window_topk(const FrameView& F, const WinQuery* q, const uint8_t* qdesc, int q_begin, int q_end,
                        TopK* out, hipStream_t s, int n_frames, int frame_stride, int q_stride, unsigned long long* n_eval, int lanes) {
    const int n = q_end - q_begin;
    if (n <= 0 || n_frames <= 0) return;
    // lanes per query: 4 for the small windows of SearchLocalPoints (th 1-2: 3-11 grid cells with about one keypoint each —
    // 64 frames x 4096 queries 0.224 ms with 16 lanes, 0.127 with 4), 16 for the wide ones (motion model, relocalisation, Fuse)
#define MSORB_WIN_LAUNCH(COUNT, LANES)                                                                                          \
    hipLaunchKernelGGL((window_topk_kernel<COUNT, LANES>), dim3((n + 256 / LANES - 1) / (256 / LANES), n_frames), dim3(256), 0, s, F, q, \
                       qdesc, q_begin, q_end, out, frame_stride, q_stride, n_eval)
    // (a single frame's few thousand queries are latency, not throughput: 12.5 us with 16 lanes, 19.7 with 4)
    if (F.gate_kp) {   // Fuse(..., bRight = true): a few thousand queries of one KeyFrame
        hipLaunchKernelGGL((window_topk_kernel<false, 16, true>), dim3((n + 15) / 16, n_frames), dim3(256), 0, s, F, q, qdesc, q_begin, q_end, out,
                           frame_stride, q_stride, n_eval);
        return;
    }
    if (lanes <= 4 && (long long)n * n_frames >= 32768) { if (n_eval) MSORB_WIN_LAUNCH(true, 4); else MSORB_WIN_LAUNCH(false, 4); }
    else { if (n_eval) MSORB_WIN_LAUNCH(true, 16); else MSORB_WIN_LAUNCH(false, 16); }
#undef MSORB_WIN_LAUNCH
}
void launch_list_top2(const uint8_t* qdesc, const uint8_t* tdesc, const int* cand_begin, const int* cand_idx, int nq,
                      int* bi, int* bd, int* si, int* sd, hipStream_t s) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(list_top2_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, qdesc, tdesc, cand_begin, cand_idx, nq, bi,
                       bd, si, sd);
}
void launch_stereo_match(const StereoArgs& a, hipStream_t s) {
    if (a.nL <= 0) return;
    hipLaunchKernelGGL(stereo_match_kernel, dim3((a.nL + 3) / 4), dim3(256), 0, s, a);
}
void launch_stereo_match_batch(const StereoBatchArgs& b, int n_pairs, int max_left, hipStream_t s, bool row_table_built) {
    if (n_pairs <= 0 || max_left <= 0) return;
    if (row_table_built) {}   // a stereo frame: band records from the selection-layout launch (StereoRowJob) instead of a table
    else if (n_pairs <= 4) hipLaunchKernelGGL(stereo_rowtable_kernel<1024>, dim3(n_pairs), dim3(1024), (size_t)(2 * b.A.rows0 + 1) * sizeof(int), s, b);
    else hipLaunchKernelGGL(stereo_rowtable_kernel<256>, dim3(n_pairs), dim3(256), (size_t)(2 * b.A.rows0 + 1) * sizeof(int), s, b);
    // batches whose row table exists: four keypoints per wave (stereo_match_quad_kernel); frames: one per wave, every wave slot used
    if (n_pairs > 4 && b.row_begin) hipLaunchKernelGGL(stereo_match_quad_kernel, dim3((max_left + 15) / 16, n_pairs), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(stereo_match_batch_kernel, dim3((max_left + 3) / 4, n_pairs), dim3(256), 0, s, b);
    if (b.median_with_readback) return;   // the caller follows up with launch_stereo_median_readback
    hipLaunchKernelGGL(stereo_median_kernel, dim3(n_pairs), dim3(256), 0, s, b.countsL, b.countsR, b.pair_step, b.capacity, b.A.sad, b.A.u_right,
                       b.A.depth, b.counts_out);
}
void launch_stereo_median_readback(const StereoBatchArgs& b, void* dst, const void* src, size_t head_bytes, size_t all_bytes, hipStream_t s) {
    const size_t n16_head = head_bytes / 16, n16_all = (all_bytes + 15) / 16;
    const int blocks = 1 + (int)std::min<size_t>((n16_head + 255) / 256, 1024);
    hipLaunchKernelGGL(stereo_median_readback_kernel, dim3(blocks), dim3(256), 0, s, b.countsL, b.countsR, b.pair_step, b.capacity, b.A.sad,
                       b.A.u_right, b.A.depth, b.counts_out, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src), n16_head, n16_all);
}

// Dense brute-force top-2 (SURVEY.md K6 "dense mode"; the knnMatch(k=2) shape of Frame.cc:1076): every query
// of a frame against every train descriptor of the same frame, candidates scanned in index order with strict
// '<' (best = lexicographic min of (distance, index), second = the runner-up).  One thread per query, the
// frame's train descriptors staged once per workgroup in LDS (<= 2048 x 32 B = 64 KB) and broadcast-read.
constexpr int kDenseMaxTrain = 2048;
// QPL = queries per lane: one LDS broadcast read of a train descriptor serves QPL pairs per lane (QPL independent
// popcount-accumulate chains).  SPLIT = the train range is split over this many 256-thread parts of a workgroup (merged at
// the end): SPLIT times the waves for the same LDS tile — a batch of 128 frames has too few waves per SIMD otherwise.
template <int kDenseQPL, int kDenseSplit>
__global__ __launch_bounds__(256 * kDenseSplit) void dense_top2_kernel(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t,
                                                        const int* __restrict__ n_q, const int* __restrict__ n_t,
                                                        int q_stride, int t_stride, int* __restrict__ best_idx,
                                                        int* __restrict__ best_dist, int* __restrict__ second_dist) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tile[];
    const int frame = blockIdx.y;
    const int nq = n_q[frame], nt = min(n_t[frame], kDenseMaxTrain);
    const int q0 = blockIdx.x * (256 * kDenseQPL);
    if (q0 >= nq) return;
    const uint64_t* ts = reinterpret_cast<const uint64_t*>(t + (size_t)frame * t_stride * 32);
    for (int i = threadIdx.x; i < nt * 4; i += 256 * kDenseSplit) tile[i] = ts[i];
    __syncthreads();
    const int part = threadIdx.x >> 8, l = threadIdx.x & 255;
    const int per = (nt + kDenseSplit - 1) / kDenseSplit;
    const int jb = part * per, je = min(jb + per, nt);
    uint32_t a[kDenseQPL][8];
    uint32_t k0[kDenseQPL], k1[kDenseQPL];  // (dist << 16) | index
#pragma unroll
    for (int u = 0; u < kDenseQPL; u++) {
        const int qi = min(q0 + u * 256 + l, nq - 1);  // lanes past the end repeat the last query, no store
        const uint4* qp = reinterpret_cast<const uint4*>(q + ((size_t)frame * q_stride + qi) * 32);
        const uint4 lo = qp[0], hi = qp[1];
        a[u][0] = lo.x; a[u][1] = lo.y; a[u][2] = lo.z; a[u][3] = lo.w; a[u][4] = hi.x; a[u][5] = hi.y; a[u][6] = hi.z; a[u][7] = hi.w;
        k0[u] = k1[u] = 0xFFFFFFFFu;
    }
    const uint4* tile4 = reinterpret_cast<const uint4*>(tile);
    auto score = [&](const uint4& tl, const uint4& th, int j) {
        const uint32_t tw[8] = {tl.x, tl.y, tl.z, tl.w, th.x, th.y, th.z, th.w};
#pragma unroll
        for (int u = 0; u < kDenseQPL; u++) {
            // the distance as ONE accumulate chain: v_bcnt_u32_b32 d, x, d adds the population count to its third operand
            // (written with 64-bit popcounts the compiler emits eight counts from zero plus adds to join them)
            // (spelled as the instruction: left to the compiler the chain is re-associated into independent counts joined by
            // v_add3_u32 — three more slow-class instructions per pair, 23 instead of 19 + the top-2 update)
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const uint32_t x = a[u][w] ^ tw[w];
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(d));
            }
            const uint32_t key = (d << 16) | (uint32_t)j;
            // (k0 <= k1) + key -> the two smallest: second = median of the three (one v_med3_u32 instead of max + min)
            // (written in the min/max form the backend folds into v_med3_u32).  Gating this on "d below the second-best
            // distance" was measured slower: with 64 lanes x QPL queries some lane improves on most steps.
            const uint32_t m = min(max(k0[u], k1[u]), max(min(k0[u], k1[u]), key));
            k0[u] = min(k0[u], key);
            k1[u] = m;
        }
    };
    // (a hand-made software pipeline over groups of four trains — next group's LDS reads in flight while the current one is
    // scored — measured slower: 1.36-1.46 against 1.65-1.73 Tpairs/s; register copies and one wave per SIMD less)
    // (unrolled by hand: the inline-asm chain keeps the loop unroller away)
    int j = jb;
    for (; j + 4 <= je; j += 4) {
        score(tile4[2 * j], tile4[2 * j + 1], j);
        score(tile4[2 * j + 2], tile4[2 * j + 3], j + 1);
        score(tile4[2 * j + 4], tile4[2 * j + 5], j + 2);
        score(tile4[2 * j + 6], tile4[2 * j + 7], j + 3);
    }
    for (; j < je; j++) score(tile4[2 * j], tile4[2 * j + 1], j);
    // merge the parts: keys are unique (distinct indices), so top-2 of the union = {min(a0,b0), min(max(a0,b0), min(a1,b1))}
    __syncthreads();  // everyone is done reading the tile; it becomes the exchange buffer
    uint32_t* xch = reinterpret_cast<uint32_t*>(tile);
    if (part > 0) {
#pragma unroll
        for (int u = 0; u < kDenseQPL; u++) {
            xch[((part - 1) * kDenseQPL + u) * 512 + l] = k0[u];
            xch[((part - 1) * kDenseQPL + u) * 512 + 256 + l] = k1[u];
        }
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int u = 0; u < kDenseQPL; u++) {
#pragma unroll
        for (int p2 = 1; p2 < kDenseSplit; p2++) {
            const uint32_t b0 = xch[((p2 - 1) * kDenseQPL + u) * 512 + l], b1 = xch[((p2 - 1) * kDenseQPL + u) * 512 + 256 + l];
            const uint32_t lo = min(k0[u], b0), hi = max(k0[u], b0);
            k1[u] = min(hi, min(k1[u], b1));
            k0[u] = lo;
        }
        const int qi = q0 + u * 256 + l;
        if (qi >= nq) continue;
        const size_t o = (size_t)frame * q_stride + qi;
        // a train at distance 256 is never taken: the scan starts from bestDist = 256 with a strict '<' (oracle orc_dense_top2)
        best_idx[o] = k0[u] >= (256u << 16) ? -1 : (int)(k0[u] & 0xFFFF);
        best_dist[o] = k0[u] >= (256u << 16) ? 256 : (int)(k0[u] >> 16);
        second_dist[o] = k1[u] == 0xFFFFFFFFu ? 256 : (int)(k1[u] >> 16);
    }
}

// ------------------------------------------------------------------------------------------------
// The same dense top-2 on the matrix cores.  A Hamming distance is a dot product in disguise: with train bits encoded as
// +-32 and query bits as -+32 (int8), sum_k a_k * b_k = 1024 * (2 d - 256), so
//     v_mfma_i32_32x32x32_i8 x 8 (K = 256 bits)  with  C = 262144 + train index   gives   acc = 2048 * d + train index,
// i.e. the MFMA accumulator IS the key (d << 11 | j) the top-2 bookkeeping orders by — 8 x 8 xor / popcount VALU
// instructions per 64 pairs become 8 MFMAs per 1024 pairs, and the VALU is left with v_med3 + v_min per pair.
//   A operand = 32 trains (rows), B operand = 32 queries (columns): a lane owns query column (lane & 31) and sees the 16 train
//   rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of every 32 x 32 tile, so its top-2 update is sequential over its registers;
//   lanes l and l + 32 are merged at the end.  Both operands use the same lane -> (row / column, k-group) rule, so the k order
//   inside the instruction does not matter.
//   Workgroup = 4 waves x 64 queries (two 32-query B fragments per wave, expanded once into 64 VGPRs).  Per tile of 32 trains
//   the workgroup expands 1 KB of descriptor bits into the 8 KB A-fragment image in LDS (double buffered): thread (row, k-step)
//   reads one dword of train `row` — prefetched from L2 four tiles ahead — and turns its four bytes into 4 x 8 int8 through a
//   256-entry LDS table.  Small workgroups on purpose: four of them share a CU and drift apart, so one's MFMAs run under
//   another's top-2 / expansion VALU work (one 16-wave workgroup in barrier lock-step measured 4.3 Tpairs/s).
constexpr int kMfmaWaves = 4, kMfmaThreads = kMfmaWaves * 64, kMfmaAhead = 4;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 4 bits -> 4 int8: pair = value for a clear bit | value for a set bit << 8; three instructions (mul24, and, perm)
__device__ __forceinline__ uint32_t expand4(uint32_t nib, uint32_t pair) {
    const uint32_t sp = __umul24(nib, 0x00204081u) & 0x01010101u;   // bit k of the nibble -> byte k = 0 / 1 (no collisions, no carries)
    return __builtin_amdgcn_perm(0u, pair, sp);                     // byte k = pair.byte[sp.byte[k]]
}
constexpr uint32_t kTrainPair = 0xE0u | (0x20u << 8);   // train bits: clear = -32, set = +32
constexpr uint32_t kQueryPair = 0x20u | (0xE0u << 8);   // query bits negated: clear = +32, set = -32
__device__ __forceinline__ v4i expand16(uint32_t bits, uint32_t pair) {
    v4i r;
    r.x = (int)expand4(bits & 15u, pair); r.y = (int)expand4((bits >> 4) & 15u, pair);
    r.z = (int)expand4((bits >> 8) & 15u, pair); r.w = (int)expand4((bits >> 12) & 15u, pair);
    return r;
}

// Four 32-query B fragments per wave (128 VGPRs of queries, 64 of accumulators, 32 of A fragments: two waves per SIMD).
// A wave issues its instructions in order, so MFMAs only run under VALU work that sits BETWEEN them in the instruction
// stream: the tile is processed block by block (8 dependent MFMAs per 32-query block), and the top-2 update of block b - 1
// (v_med3 + v_min per pair) is interleaved with the MFMAs of block b — across the tile boundary too: the last block of a
// tile is consumed under the first block of the next.  sched_group_barrier pins the pattern (1 MFMA, 4 VALU).
__global__ __launch_bounds__(kMfmaThreads, 2) void dense_top2_mfma_kernel(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t,
                                                                          const int* __restrict__ n_q, const int* __restrict__ n_t,
                                                                          int q_stride, int t_stride, int* __restrict__ best_idx,
                                                                          int* __restrict__ best_dist, int* __restrict__ second_dist) {
    constexpr int NB = 4;
    __shared__ __attribute__((aligned(16))) uint4 frag[2][512];   // [buffer][k-step * 64 + lane] = the 16 bytes that lane reads
    __shared__ uint2 lut[256];                                    // byte of descriptor bits -> 8 int8 (train convention)
    const int frame = blockIdx.y;
    const int nq = n_q[frame], nt = min(n_t[frame], kDenseMaxTrain);
    const int q0 = blockIdx.x * (kMfmaWaves * NB * 32);
    if (q0 >= nq) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    lut[tid] = uint2{expand4(tid & 15u, kTrainPair), expand4((uint32_t)tid >> 4, kTrainPair)};
    // B fragments: query column (lane & 31) of the wave's 32-query blocks, k-group lane >> 5: bits [32 s + 16 g, +16)
    v4i bq[NB][8];
#pragma unroll
    for (int blk = 0; blk < NB; blk++) {
        const int qi = min(q0 + (wave * NB + blk) * 32 + (lane & 31), nq - 1);
        const uint4* qp = reinterpret_cast<const uint4*>(q + ((size_t)frame * q_stride + qi) * 32);
        const uint4 lo = qp[0], hi = qp[1];   // the whole descriptor: dword s8 holds the bits of k-step s8, its half (lane >> 5) this lane's
        const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) bq[blk][s8] = expand16((dw[s8] >> (16 * (lane >> 5))) & 0xffffu, kQueryPair);
    }
    const int n_tiles = (nt + 31) / 32;
    // expansion role of this thread: train row tid & 31 of the tile, k-step tid >> 5: descriptor bytes [4 s, 4 s + 4) ->
    // k-group 0 chunk (bytes 0,1) at frag[s * 64 + row] and k-group 1 chunk (bytes 2,3) at frag[s * 64 + 32 + row]
    const int e_row = tid & 31, e_step = tid >> 5;
    const uint8_t* tbase = t + (size_t)frame * t_stride * 32 + 4 * e_step;
    auto fetch = [&](int tile) {   // rows / tiles past the end re-read the last train: their keys are forced high below
        return *reinterpret_cast<const uint32_t*>(tbase + (size_t)min(tile * 32 + e_row, nt - 1) * 32);
    };
    auto expand_tile = [&](uint32_t w, int buf) {
        const uint2 e0 = lut[w & 255u], e1 = lut[(w >> 8) & 255u], e2 = lut[(w >> 16) & 255u], e3 = lut[w >> 24];
        frag[buf][e_step * 64 + e_row] = uint4{e0.x, e0.y, e1.x, e1.y};
        frag[buf][e_step * 64 + 32 + e_row] = uint4{e2.x, e2.y, e3.x, e3.y};
    };
    uint32_t pre[kMfmaAhead];
    if (nt > 0) {
#pragma unroll
        for (int i = 0; i < kMfmaAhead; i++) pre[i] = fetch(1 + i);
        const uint32_t w0 = fetch(0);
        __syncthreads();   // lut
        expand_tile(w0, 0);
    }
    __syncthreads();
    const int lane_row = 262144 + 4 * (lane >> 5);   // C operand = 262144 + global train index of accumulator register r
    int k0[NB], k1[NB];
#pragma unroll
    for (int blk = 0; blk < NB; blk++) k0[blk] = k1[blk] = 0x7FFFFFFF;
    v16i acc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; blk++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[blk][r] = 0x7FFFFFFF;   // "previous tile" of the first one: no effect on the top-2
    auto top2 = [&](int blk) {
#pragma unroll
        for (int r = 0; r < 16; r++) {   // (k0 <= k1) + key -> the two smallest: second = median of the three
            k1[blk] = min(max(k0[blk], k1[blk]), max(min(k0[blk], k1[blk]), acc[blk][r]));
            k0[blk] = min(k0[blk], acc[blk][r]);
        }
    };
    for (int tile0 = 0; tile0 < n_tiles; tile0 += kMfmaAhead) {
#pragma unroll
        for (int u = 0; u < kMfmaAhead; u++) {
            const int tile = tile0 + u;
            if (tile >= n_tiles) break;   // block-uniform
            const int buf = tile & 1;
            // all eight A fragments of the tile in flight at once
            const uint4* fr = &frag[buf][lane];
            v4i a[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) { const uint4 av = fr[s8 * 64]; a[s8] = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w}; }
            if (tile + 1 < n_tiles) {
                expand_tile(pre[u], buf ^ 1);
                pre[u] = fetch(tile + 1 + kMfmaAhead);
            }
            v16i c;
#pragma unroll
            for (int r = 0; r < 16; r++) c[r] = lane_row + tile * 32 + ((r & 3) + 8 * (r >> 2));
            if (tile == n_tiles - 1) {   // the last tile may be partial: rows past nt get a key no real pair reaches
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (c[r] - 262144 >= nt) c[r] = 0x40000000;
            }
#pragma unroll
            for (int blk = 0; blk < NB; blk++) {
                // consume the block computed just before (of the previous tile for blk == 0) ...
                const int prev = (blk + NB - 1) % NB;
                top2(prev);
                // ... under this block's MFMAs (they overwrite acc[blk], consumed one step earlier)
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++)
                    acc[blk] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s8], bq[blk][s8], s8 == 0 ? c : acc[blk], 0, 0, 0);
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
                }
            }
            __syncthreads();   // tile + 1 is expanded, and everyone is done reading buffer `buf`
        }
    }
    top2(NB - 1);   // the last block of the last tile
    // merge the two halves of the wave (same query, complementary train rows); keys are unique
#pragma unroll
    for (int blk = 0; blk < NB; blk++) {
        const int o0 = __shfl_xor(k0[blk], 32), o1 = __shfl_xor(k1[blk], 32);
        const int lo = min(k0[blk], o0), hi = max(k0[blk], o0);
        k1[blk] = min(hi, min(k1[blk], o1));
        k0[blk] = lo;
        const int qi = q0 + (wave * NB + blk) * 32 + (lane & 31);
        if (lane < 32 && qi < nq) {
            const size_t o = (size_t)frame * q_stride + qi;
            // a train at distance 256 is never taken: the scan starts from bestDist = 256 with a strict '<' (oracle orc_dense_top2)
            best_idx[o] = k0[blk] >= (256 << 11) ? -1 : (k0[blk] & 2047);
            best_dist[o] = k0[blk] >= (256 << 11) ? 256 : (k0[blk] >> 11);
            second_dist[o] = k1[blk] >= (256 << 11) ? 256 : (k1[blk] >> 11);   // also the masked rows: 0x40000000 + (a sum within +-262144)
        }
    }
}

template <int QPL, int SPLIT>
static void launch_dense_variant(const uint8_t* q, const uint8_t* t, const int* n_q, const int* n_t, int n_frames, int q_stride,
                                 int t_stride, int max_q, int max_t, int* bi, int* bd, int* sd, hipStream_t s) {
    // tile of the train descriptors; at least the exchange buffer of the final merge
    const size_t lds = std::max<size_t>((size_t)min(max_t, kDenseMaxTrain) * 32, (size_t)(SPLIT - 1) * QPL * 512 * 4);
    const int per_block = 256 * QPL;
    hipLaunchKernelGGL((dense_top2_kernel<QPL, SPLIT>), dim3((max_q + per_block - 1) / per_block, n_frames), dim3(256 * SPLIT), lds, s,
                       q, t, n_q, n_t, q_stride, t_stride, bi, bd, sd);
}
// formulation 0: matrix cores (dense_top2_mfma_kernel); 1: xor + popcount (dense_top2_kernel<2, 4>, BASELINE north_star's form).
// Measured on MI355X (128 frames x 2000 x 2000), popcount <QPL, SPLIT>: 2,1 1.54  2,2 1.65  2,4 1.72  4,2 1.68  4,4 1.73  1,4 1.57
// Tpairs/s; matrix cores 5.1.
void launch_dense_top2(const uint8_t* q, const uint8_t* t, const int* n_q, const int* n_t, int n_frames, int q_stride,
                       int t_stride, int max_q, int max_t, int* bi, int* bd, int* sd, hipStream_t s, int formulation) {
    if (n_frames <= 0 || max_q <= 0) return;
    if (formulation == 0) {
        const int per_block = kMfmaWaves * 128;
        hipLaunchKernelGGL(dense_top2_mfma_kernel, dim3((max_q + per_block - 1) / per_block, n_frames), dim3(kMfmaThreads), 0, s, q, t, n_q, n_t,
                           q_stride, t_stride, bi, bd, sd);
        return;
    }
    launch_dense_variant<2, 4>(q, t, n_q, n_t, n_frames, q_stride, t_stride, max_q, max_t, bi, bd, sd, s);
}

}  // namespace msorb
