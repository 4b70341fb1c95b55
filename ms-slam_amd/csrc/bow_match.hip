// ORBmatcher::SearchByBoW (src/ORBmatcher.cc:223-421 pinhole branch, :872-1016, :1018-1166) for a batch of
// (KeyFrame, Frame) / (KeyFrame, KeyFrame) pairs.
//
// The reference merge-walks the two DBoW2::FeatureVectors (:239-243, :385-392); for every node both have, each
// query of the node (in list order) scans the node's trains that are still unclaimed (:274-275, :937), keeps
// best / second best with strict '<' (:283-292) and, when accepted (:332-336), claims its best train.  A feature
// sits in exactly one node of its vector, so claims never cross nodes: the nodes are independent problems and the
// order dependence lives inside one node only.  Here: one wavefront per (pair, common node); the node's trains are
// spread over the lanes (position p = chunk*64 + lane, chunk 0's descriptors stay in registers), the node's queries
// are visited in list order, each one a wave-wide masked top-2 (key = dist<<20 | position, so the minimum is the
// first strict minimum of the reference's scan), the claim is one bit in LDS.  The rotation histogram and
// ComputeThreeMaxima (:340-353, :396-418) are replayed on the host in the reference's visiting order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/msorb.h"
#include "matcher_device.h"

namespace msorb {
void set_last_error(const std::string& s);
// pinned host <-> device on a stream by the copy kernel (orb_kernels.hip; hipMemcpyAsync for unaligned pointers / MSORB_FRAME_COPIES=sdma):
// a block of 100-300 KB is across before an SDMA copy has started
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
}
using msorb::set_last_error;
using msorb::kHistoLength;

namespace {

struct BowItem {  // one (pair, common node)
    int base1, base2;  // first row of the pair's set 1 / set 2 in the descriptor (and static per-feature) arrays
    int b1, n1l;       // the node's query list inside feat1
    int b2, n2l;       // the node's train list inside feat2
    int pair;
    int m1, m2;        // first entry of the pair's set 1 / set 2 in the PER-CALL arrays (visit / availability flags, match12):
                       // equal to base1 / base2 when everything is staged per call, different when the descriptors live in a
                       // resident KeyFrame store
};

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

constexpr int kNoKey = 0x7fffffff;

__global__ __launch_bounds__(64) void bow_match_kernel(const BowItem* __restrict__ items, const uint4* __restrict__ desc1,
                                                       const uint4* __restrict__ desc2, const uint8_t* __restrict__ valid1,
                                                       const uint8_t* __restrict__ avail2, const int* __restrict__ feat1,
                                                       const int* __restrict__ feat2, int th_low, int flags,
                                                       float nnratio, int* __restrict__ match12, int* __restrict__ best1 = nullptr) {
    // flags: bit 0 = inclusive (best <= th_low, :332; else best < th_low, :959), bit 1 = no ratio test (the right-camera arm of
    // SearchByBoW(pKF, F) on a two-camera frame, :357-359: `|| true`).  best1 (optional): per query feature the distance of its best
    // unclaimed train at the time it is processed — before the threshold and the ratio test — or 256.
    const int inclusive = flags & 1;
    const bool no_ratio = (flags & 2) != 0;
    extern __shared__ unsigned free_bits[];  // bit p: train at list position p is unclaimed
    const BowItem it = items[blockIdx.x];
    const int lane = threadIdx.x;
    const int chunks = (it.n2l + 63) >> 6;
    for (int w = lane; w < chunks * 2; w += 64) free_bits[w] = 0;
    __syncthreads();
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    for (int c = 0; c < chunks; c++) {
        const int p = c * 64 + lane;
        bool fr = false;
        if (p < it.n2l) {
            const int f2 = feat2[it.b2 + p], idx2 = it.base2 + f2;
            fr = avail2[it.m2 + f2] != 0;
            if (c == 0) { t0 = desc2[(size_t)idx2 * 2]; t1 = desc2[(size_t)idx2 * 2 + 1]; }
        }
        const unsigned long long m = __ballot(fr);
        if (lane == 0) { free_bits[2 * c] = (unsigned)m; free_bits[2 * c + 1] = (unsigned)(m >> 32); }
    }
    __syncthreads();
    for (int k1 = 0; k1 < it.n1l; k1++) {
        const int f1 = feat1[it.b1 + k1], idx1 = it.base1 + f1;
        if (!valid1[it.m1 + f1]) continue;  // wave uniform
        const uint4 q0 = desc1[(size_t)idx1 * 2], q1 = desc1[(size_t)idx1 * 2 + 1];
        int key = kNoKey, second = 256;
        for (int c = 0; c < chunks; c++) {
            const int p = c * 64 + lane;
            if (p < it.n2l && ((free_bits[p >> 5] >> (p & 31)) & 1u)) {
                int dist;
                if (c == 0) dist = hamming256(q0, q1, t0, t1);
                else {
                    const int idx2 = it.base2 + feat2[it.b2 + p];
                    dist = hamming256(q0, q1, desc2[(size_t)idx2 * 2], desc2[(size_t)idx2 * 2 + 1]);
                }
                if (dist < 256) {  // bestDist1 starts at 256 with a strict '<' (:265-267, :283)
                    const int kk = (dist << 20) | p;
                    if (kk < key) { second = min(second, key >> 20); key = kk; }
                    else second = min(second, dist);
                }
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int ok = __shfl_xor(key, off), os = __shfl_xor(second, off);
            second = min(min(second, os), max(key, ok) >> 20);
            key = min(key, ok);
        }
        second = min(second, 256);
        const int best = key >> 20;
        if (best1 && lane == 0) best1[it.m1 + f1] = key != kNoKey ? best : 256;
        const bool pass = key != kNoKey && (inclusive ? best <= th_low : best < th_low) &&
                          (no_ratio || (float)best < nnratio * (float)second);  // :332-336, :959-961
        if (pass) {
            const int p = key & 0xFFFFF;
            if (lane == 0) {
                match12[it.m1 + f1] = feat2[it.b2 + p];
                free_bits[p >> 5] &= ~(1u << (p & 31));
            }
            __syncthreads();
        }
    }
}

struct TriConst {  // per pair
    float F[9];
    float ep[2];
};

// SearchForTriangulation's node problem (:1230-1358): best-only, bestDist starts at TH_LOW and a candidate replaces
// the best when dist <= bestDist (:1277) — the LAST minimum of the scan — provided it passes the epipole-distance
// gate (:1283-1291, mono-mono only) and Pinhole::epipolarConstrain (Pinhole.cpp:107-131).  The gates do not depend on
// the running best, so the winner is the minimum of key = dist<<20 | (0xFFFFF - position) over the gated candidates.
__global__ __launch_bounds__(64) void triangulation_match_kernel(
    const BowItem* __restrict__ items, const TriConst* __restrict__ consts, const uint4* __restrict__ desc1,
    const uint4* __restrict__ desc2, const uint8_t* __restrict__ flags1, const uint8_t* __restrict__ flags2,
    const float2* __restrict__ xy1, const float4* __restrict__ tr2, const int* __restrict__ feat1,
    const int* __restrict__ feat2, int coarse, int* __restrict__ match12) {
    extern __shared__ unsigned free_bits[];
    const BowItem it = items[blockIdx.x];
    const TriConst K = consts[it.pair];
    const int lane = threadIdx.x;
    const int chunks = (it.n2l + 63) >> 6;
    for (int w = lane; w < chunks * 2; w += 64) free_bits[w] = 0;
    __syncthreads();
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    float4 r0 = make_float4(0, 0, 0, 0);
    bool s0 = false;
    for (int c = 0; c < chunks; c++) {
        const int p = c * 64 + lane;
        bool fr = false;
        if (p < it.n2l) {
            const int f2 = feat2[it.b2 + p], idx2 = it.base2 + f2;
            const uint8_t fl = flags2[it.m2 + f2];
            fr = (fl & 1) != 0;
            if (c == 0) { t0 = desc2[(size_t)idx2 * 2]; t1 = desc2[(size_t)idx2 * 2 + 1]; r0 = tr2[idx2]; s0 = (fl & 2) != 0; }
        }
        const unsigned long long m = __ballot(fr);
        if (lane == 0) { free_bits[2 * c] = (unsigned)m; free_bits[2 * c + 1] = (unsigned)(m >> 32); }
    }
    __syncthreads();
    for (int k1 = 0; k1 < it.n1l; k1++) {
        const int f1 = feat1[it.b1 + k1], idx1 = it.base1 + f1;
        const uint8_t fl1 = flags1[it.m1 + f1];
        if (!(fl1 & 1)) continue;  // wave uniform
        const bool stereo1 = (fl1 & 2) != 0;
        const uint4 q0 = desc1[(size_t)idx1 * 2], q1 = desc1[(size_t)idx1 * 2 + 1];
        const float2 P1 = xy1[idx1];
        // l = x1' F12 (Pinhole.cpp:114-117)
        const float a = __fadd_rn(__fmaf_rn(P1.x, K.F[0], __fmul_rn(P1.y, K.F[3])), K.F[6]);
        const float b = __fadd_rn(__fmaf_rn(P1.x, K.F[1], __fmul_rn(P1.y, K.F[4])), K.F[7]);
        const float cc = __fadd_rn(__fmaf_rn(P1.x, K.F[2], __fmul_rn(P1.y, K.F[5])), K.F[8]);
        const float den = __fmaf_rn(a, a, __fmul_rn(b, b));
        int key = kNoKey;
        for (int c = 0; c < chunks; c++) {
            const int p = c * 64 + lane;
            if (p < it.n2l && ((free_bits[p >> 5] >> (p & 31)) & 1u)) {
                int dist;
                float4 r;
                bool stereo2;
                if (c == 0) { dist = hamming256(q0, q1, t0, t1); r = r0; stereo2 = s0; }
                else {
                    const int f2 = feat2[it.b2 + p], idx2 = it.base2 + f2;
                    dist = hamming256(q0, q1, desc2[(size_t)idx2 * 2], desc2[(size_t)idx2 * 2 + 1]);
                    r = tr2[idx2];
                    stereo2 = (flags2[it.m2 + f2] & 2) != 0;
                }
                bool ok = dist <= msorb::kThLow;  // :1277
                if (ok && !stereo1 && !stereo2) {  // :1283-1291
                    const float ex = __fsub_rn(K.ep[0], r.x), ey = __fsub_rn(K.ep[1], r.y);
                    if (__fmaf_rn(ex, ex, __fmul_rn(ey, ey)) < r.z) ok = false;
                }
                if (ok && !coarse) {  // Pinhole.cpp:119-130
                    const float num = __fadd_rn(__fmaf_rn(a, r.x, __fmul_rn(b, r.y)), cc);
                    if (den == 0.0f) ok = false;
                    else {
                        const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                        ok = (double)dsqr < 3.84 * (double)r.w;
                    }
                }
                if (ok) key = min(key, (dist << 20) | (0xFFFFF - p));
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) key = min(key, __shfl_xor(key, off));
        if (key != kNoKey) {
            const int p = 0xFFFFF - (key & 0xFFFFF);
            if (lane == 0) {
                match12[it.m1 + f1] = feat2[it.b2 + p];
                free_bits[p >> 5] &= ~(1u << (p & 31));
            }
            __syncthreads();
        }
    }
}

// Every (query, train) of a node with distance <= th_low among the visited queries / eligible trains, in the reference's scan order
// (queries in list order, trains in list order): the candidates of a search whose accept test lives with the CALLER
// (msorb_search_for_triangulation_cb: GeometricCamera::epipolarConstrain of a camera model this library does not hold).
// FILL = false counts per item, FILL = true writes (query, train, dist, train position) at list + begin[item].
template <bool FILL>
__global__ __launch_bounds__(64) void node_candidates_kernel(const BowItem* __restrict__ items, const uint4* __restrict__ desc1,
                                                             const uint4* __restrict__ desc2, const uint8_t* __restrict__ valid1,
                                                             const uint8_t* __restrict__ avail2, const int* __restrict__ feat1,
                                                             const int* __restrict__ feat2, int th_low, int* __restrict__ count,
                                                             const int* __restrict__ begin, int4* __restrict__ list) {
    const BowItem it = items[blockIdx.x];
    const int lane = threadIdx.x;
    const int chunks = (it.n2l + 63) >> 6;
    int total = 0;
    int4* out = FILL ? list + begin[blockIdx.x] : nullptr;
    for (int k1 = 0; k1 < it.n1l; k1++) {
        const int f1 = feat1[it.b1 + k1];
        if (!valid1[it.m1 + f1]) continue;  // wave uniform
        const int idx1 = it.base1 + f1;
        const uint4 q0 = desc1[(size_t)idx1 * 2], q1 = desc1[(size_t)idx1 * 2 + 1];
        for (int c = 0; c < chunks; c++) {
            const int p = c * 64 + lane;
            bool ok = false;
            int f2 = 0, dist = 0;
            if (p < it.n2l) {
                f2 = feat2[it.b2 + p];
                if (avail2[it.m2 + f2]) {
                    const int idx2 = it.base2 + f2;
                    dist = hamming256(q0, q1, desc2[(size_t)idx2 * 2], desc2[(size_t)idx2 * 2 + 1]);
                    ok = dist <= th_low;
                }
            }
            const unsigned long long m = __ballot(ok);
            if (FILL && ok) out[total + __popcll(m & ((1ull << lane) - 1))] = make_int4(f1, f2, dist, p);
            total += __popcll(m);
        }
    }
    if (!FILL && lane == 0) count[blockIdx.x] = total;
}

// The rotation-consistency filter of a pair on the device (resident-KeyFrame entries): histogram of round((angle1 - angle2) / 12
// degrees) over the pair's raw matches, ComputeThreeMaxima (ORBmatcher.cc:2277-2318) on the 30 counts, matches outside the
// three fullest bins withdrawn (:396-418, :1360-1381), match21 and the count written.  The survivors do not depend on the
// order the reference visits the matches in — only the counts enter — so no replay on the host is needed (the host loop
// over 2000 features costs 24 us per pair: seven times the device part of a 32-pair batch).  One workgroup per pair.
struct PairPost {
    int m1, n1, m2, n2;   // the pair's slices of the per-call match12 / match21 arrays
    int a1, a2;           // first entry of the pair's set 1 / set 2 in angle1 / angle2
};
__global__ __launch_bounds__(256) void pair_histogram_kernel(const PairPost* __restrict__ posts, const float* __restrict__ angle1,
                                                            const float* __restrict__ angle2, int check_orientation,
                                                            int* __restrict__ match12, int* __restrict__ match21,
                                                            int* __restrict__ nmatches) {
    __shared__ int hist[kHistoLength];
    __shared__ int ind[3];
    __shared__ int total;
    const PairPost P = posts[blockIdx.x];
    const int t = threadIdx.x;
    if (t < kHistoLength) hist[t] = 0;
    if (t == 0) total = 0;
    __syncthreads();
    auto bin_of = [&](int i, int idx2) {
        float rot = __fsub_rn(angle1[P.a1 + i], angle2[P.a2 + idx2]);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHistoLength));
        if (bin == kHistoLength) bin = 0;
        return (bin >= 0 && bin < kHistoLength) ? bin : -1;   // NaN / out-of-range angle: the reference asserts
    };
    if (check_orientation) {
        for (int i = t; i < P.n1; i += 256) {
            const int idx2 = match12[P.m1 + i];
            if (idx2 < 0) continue;
            const int b = bin_of(i, idx2);
            if (b >= 0) atomicAdd(&hist[b], 1);
        }
        __syncthreads();
        if (t == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < kHistoLength; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if (max3 < 0.1f * (float)max1) { i3 = -1; }
            ind[0] = i1; ind[1] = i2; ind[2] = i3;
        }
        __syncthreads();
    }
    int kept = 0;
    for (int i = t; i < P.n1; i += 256) {
        const int idx2 = match12[P.m1 + i];
        if (idx2 < 0) continue;
        bool keep = true;
        if (check_orientation) {
            const int b = bin_of(i, idx2);
            keep = b >= 0 && (b == ind[0] || b == ind[1] || b == ind[2]);
        }
        if (!keep) { match12[P.m1 + i] = -1; continue; }
        kept++;
        if (match21 && idx2 < P.n2) match21[P.m2 + idx2] = i;
    }
    atomicAdd(&total, kept);
    __syncthreads();
    if (t == 0) nmatches[blockIdx.x] = total;
}

struct Scratch {
    int device = -1;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    char *d = nullptr, *h = nullptr;
    size_t cap = 0;
    void release() {
        if (device < 0 || hipSetDevice(device) != hipSuccess) return;
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
        d = h = nullptr; s = nullptr; e0 = e1 = nullptr; cap = 0; device = -1;
    }
    hipError_t acquire(int dev, size_t total) {
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess && device != dev) {
            release();
            device = dev;
            e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreate(&e0);
            if (e == hipSuccess) e = hipEventCreate(&e1);
        }
        if (e == hipSuccess && total > cap) {
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            d = h = nullptr; cap = 0;
            e = hipMalloc((void**)&d, total + total / 2);
            if (e == hipSuccess) e = hipHostMalloc((void**)&h, total + total / 2, hipHostMallocDefault);
            if (e == hipSuccess) cap = total + total / 2;
        }
        return e;
    }
    ~Scratch() { release(); }
};

struct FeatVec {  // DBoW2::FeatureVector as CSR
    int nodes;
    const int *node, *begin, *feat;
};
struct Common { int r1, r2; };  // rows of the two vectors holding the same node id

bool check_feature_vector(int n, const FeatVec& v, std::vector<uint8_t>& seen) {
    if (v.nodes < 0 || (v.nodes > 0 && (!v.node || !v.begin))) return false;
    if (v.nodes == 0) return true;
    if (v.begin[0] < 0) return false;
    for (int r = 0; r < v.nodes; r++) {
        if (v.begin[r + 1] < v.begin[r]) return false;
        if (r > 0 && v.node[r] <= v.node[r - 1]) return false;
    }
    if (v.begin[v.nodes] > v.begin[0] && !v.feat) return false;
    seen.assign((size_t)n, 0);
    for (int k = v.begin[0]; k < v.begin[v.nodes]; k++) {
        const int i = v.feat[k];
        if (i < 0 || i >= n || seen[i]) return false;
        seen[i] = 1;
    }
    return true;
}

// the merge walk of :239-243 / :385-392: nodes both vectors hold (with non-empty lists), ascending
bool merge_walk(const FeatVec& a, const FeatVec& b, std::vector<Common>& out, size_t& totf1, size_t& totf2, int& max_chunks) {
    int i = 0, j = 0;
    while (i < a.nodes && j < b.nodes) {
        if (a.node[i] == b.node[j]) {
            const int l1 = a.begin[i + 1] - a.begin[i], l2 = b.begin[j + 1] - b.begin[j];
            if (l1 > 0 && l2 > 0) {
                if (l2 >= (1 << 20)) return false;
                out.push_back({i, j});
                totf1 += (size_t)l1;
                totf2 += (size_t)l2;
                max_chunks = std::max(max_chunks, (l2 + 63) >> 6);
            }
            i++; j++;
        } else if (a.node[i] < b.node[j]) i++;
        else j++;
    }
    return true;
}

// copies the common nodes' lists of one pair behind k1 / k2 and appends their items
void stage_lists(const FeatVec& a, const FeatVec& b, const std::vector<Common>& common, int pi, size_t r1, size_t r2, int* f1,
                 int* f2, size_t& k1, size_t& k2, BowItem* items, size_t& ni) {
    for (const Common& c : common) {
        const int l1 = a.begin[c.r1 + 1] - a.begin[c.r1], l2 = b.begin[c.r2 + 1] - b.begin[c.r2];
        std::memcpy(f1 + k1, a.feat + a.begin[c.r1], (size_t)l1 * 4);
        std::memcpy(f2 + k2, b.feat + b.begin[c.r2], (size_t)l2 * 4);
        items[ni++] = BowItem{(int)r1, (int)r2, (int)k1, l1, (int)k2, l2, pi, (int)r1, (int)r2};
        k1 += (size_t)l1;
        k2 += (size_t)l2;
    }
}

// The rotation histogram in the reference's visiting order (:340-353 / :1342-1354) and the ComputeThreeMaxima filter
// (:396-418 / :1360-1381).  m = the kernel's raw matches of this pair; angle(i1, i2) = the two keypoint angles.
template <class Angles>
int replay_histogram(const FeatVec& a, const std::vector<Common>& common, const int* m, int check_orientation, Angles angle,
                     int* match12) {
    // rotHist[bin] only decides which matches survive (the three fullest bins, ComputeThreeMaxima): a bin index per match
    // and 30 counters replace the reference's 30 vectors — this runs once per pair on the calling thread, and with the
    // vectors it cost 23 us per pair, seven times the whole device part of a 32-pair batch
    static thread_local std::vector<int8_t> bin_of;
    int sizes[kHistoLength] = {0};
    const float factor = 1.0f / kHistoLength;
    int nm = 0, n_feat = 0;
    for (const Common& c : common) n_feat = std::max(n_feat, a.begin[c.r1 + 1]);
    if (check_orientation && (int)bin_of.size() < n_feat) bin_of.resize(n_feat);
    for (const Common& c : common)
        for (int k = a.begin[c.r1]; k < a.begin[c.r1 + 1]; k++) {
            const int idx1 = a.feat[k], idx2 = m[idx1];
            if (idx2 < 0) continue;
            match12[idx1] = idx2;
            nm++;
            if (check_orientation) {
                float a1, a2;
                angle(idx1, idx2, a1, a2);
                float rot = a1 - a2;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == kHistoLength) bin = 0;
                if (bin >= 0 && bin < kHistoLength) { bin_of[k] = (int8_t)bin; sizes[bin]++; }
                else { match12[idx1] = -1; nm--; }  // NaN / out-of-range angle: the reference asserts
            }
        }
    if (check_orientation) {
        int ind[3];
        msorb_three_maxima(sizes, kHistoLength, ind);
        for (const Common& c : common)
            for (int k = a.begin[c.r1]; k < a.begin[c.r1 + 1]; k++) {
                const int idx1 = a.feat[k];
                if (match12[idx1] < 0) continue;
                const int bin = bin_of[k];
                if (bin != ind[0] && bin != ind[1] && bin != ind[2]) { match12[idx1] = -1; nm--; }
            }
    }
    return nm;
}

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

int no_device() {
    set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
    return MSORB_E_NO_DEVICE;
}
int hip_fail(Scratch& scr, const char* what, hipError_t e) {
    set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    scr.release();
    return MSORB_E_HIP;
}

}  // namespace

namespace {
// msorb_search_by_bow's body.  flags: bit 0 inclusive, bit 1 no ratio test; best1 (optional): per pair the kernel's best1 output.
int search_by_bow_impl(int device, msorb_bow_pair* pairs, int n_pairs, int th_low, int flags, float nnratio, int check_orientation,
                       float* elapsed_ms, std::vector<std::vector<int>>* best1) {
    const int inclusive = flags;   // (passed through to the kernel, which reads the bits)
    if (best1) { best1->assign(n_pairs, {}); for (int pi = 0; pi < n_pairs; pi++) (*best1)[pi].assign(std::max(pairs[pi].n1, 0), 256); }
    if (elapsed_ms) *elapsed_ms = 0;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSORB_E_INVALID;
    if (n_pairs == 0) return MSORB_OK;
    std::vector<std::vector<Common>> common(n_pairs);
    std::vector<FeatVec> fa(n_pairs), fb(n_pairs);
    std::vector<uint8_t> seen;
    size_t tot1 = 0, tot2 = 0, totf1 = 0, totf2 = 0, n_items = 0;
    int max_chunks = 1;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_pair& P = pairs[pi];
        P.nmatches = 0;
        fa[pi] = FeatVec{P.fv1_nodes, P.fv1_node, P.fv1_begin, P.fv1_feat};
        fb[pi] = FeatVec{P.fv2_nodes, P.fv2_node, P.fv2_begin, P.fv2_feat};
        if (P.n1 < 0 || P.n2 < 0 || (!P.match12 && P.n1 > 0) || (P.n1 > 0 && (!P.desc1 || !P.valid1)) || (P.n2 > 0 && !P.desc2) ||
            (check_orientation && ((P.n1 > 0 && !P.angle1) || (P.n2 > 0 && !P.angle2))) ||
            !check_feature_vector(P.n1, fa[pi], seen) || !check_feature_vector(P.n2, fb[pi], seen) ||
            !merge_walk(fa[pi], fb[pi], common[pi], totf1, totf2, max_chunks)) {
            set_last_error("search_by_bow: pair " + std::to_string(pi) +
                           ": bad sizes / null arrays / feature vector not ascending, out of range or with a repeated feature");
            return MSORB_E_INVALID;
        }
        n_items += common[pi].size();
        tot1 += (size_t)P.n1;
        tot2 += (size_t)P.n2;
    }
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_pair& P = pairs[pi];
        for (int i = 0; i < P.n1; i++) P.match12[i] = -1;
        if (P.match21) for (int j = 0; j < P.n2; j++) P.match21[j] = -1;
    }
    if (n_items == 0) return MSORB_OK;
    if (tot1 > (size_t)INT32_MAX / 2 || tot2 > (size_t)INT32_MAX / 2) return MSORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return no_device();
    // ---- staging: [desc1 | desc2 | feat1 | feat2 | items | valid1 | avail2] in, [match12] out ----
    const size_t o_d1 = 0, o_d2 = o_d1 + tot1 * 32, o_f1 = o_d2 + tot2 * 32, o_f2 = o_f1 + up16(totf1 * 4),
                 o_it = o_f2 + up16(totf2 * 4), o_v1 = o_it + up16(n_items * sizeof(BowItem)), o_a2 = o_v1 + up16(tot1),
                 in_bytes = o_a2 + up16(tot2), o_m = in_bytes, o_b = o_m + up16(tot1 * 4), total = o_b + (best1 ? up16(tot1 * 4) : 0);
    static thread_local Scratch scr;
    hipError_t e = scr.acquire(device, total);
    if (e != hipSuccess) return hip_fail(scr, "search_by_bow", e);
    {
        char* h = scr.h;
        size_t r1 = 0, r2 = 0, k1 = 0, k2 = 0, ni = 0;
        for (int pi = 0; pi < n_pairs; pi++) {
            const msorb_bow_pair& P = pairs[pi];
            if (P.n1) std::memcpy(h + o_d1 + r1 * 32, P.desc1, (size_t)P.n1 * 32);
            if (P.n2) std::memcpy(h + o_d2 + r2 * 32, P.desc2, (size_t)P.n2 * 32);
            if (P.n1) std::memcpy(h + o_v1 + r1, P.valid1, (size_t)P.n1);
            if (P.n2) {
                if (P.avail2) std::memcpy(h + o_a2 + r2, P.avail2, (size_t)P.n2);
                else std::memset(h + o_a2 + r2, 1, (size_t)P.n2);
            }
            stage_lists(fa[pi], fb[pi], common[pi], pi, r1, r2, (int*)(h + o_f1), (int*)(h + o_f2), k1, k2,
                        (BowItem*)(h + o_it), ni);
            r1 += (size_t)P.n1;
            r2 += (size_t)P.n2;
        }
    }
    hipStream_t s = scr.s;
    char* d = scr.d;
    e = msorb::small_copy(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, tot1 * 4, s);
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bow_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s,
                           (const BowItem*)(d + o_it), (const uint4*)(d + o_d1), (const uint4*)(d + o_d2),
                           (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2), (const int*)(d + o_f1),
                           (const int*)(d + o_f2), th_low, inclusive, nnratio, (int*)(d + o_m), best1 ? (int*)(d + o_b) : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = msorb::small_copy(scr.h + o_m, d + o_m, (best1 ? o_b - o_m : 0) + tot1 * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_by_bow", e);
    if (best1) {   // (features of nodes the two vectors do not share were never visited: they keep 256)
        size_t r = 0;
        for (int pi = 0; pi < n_pairs; pi++) {
            const int* b = (const int*)(scr.h + o_b) + r;
            for (const Common& c : common[pi])
                for (int k = fa[pi].begin[c.r1]; k < fa[pi].begin[c.r1 + 1]; k++) {
                    const int i1 = fa[pi].feat[k];
                    if (pairs[pi].valid1[i1]) (*best1)[pi][i1] = b[i1];
                }
            r += (size_t)pairs[pi].n1;
        }
    }
    // the raw matches are read feature by feature below: out of the pinned block first (one streaming copy) — scattered 4-byte
    // reads of pinned host memory cost ~10 ns each, 0.8 ms for a 32-pair batch
    static thread_local std::vector<int> m_local;
    m_local.resize(tot1);
    std::memcpy(m_local.data(), scr.h + o_m, tot1 * 4);
    const int* m_all = m_local.data();
    size_t r1 = 0;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_pair& P = pairs[pi];
        P.nmatches = replay_histogram(fa[pi], common[pi], m_all + r1, check_orientation,
                                      [&](int i1, int i2, float& a1, float& a2) { a1 = P.angle1[i1]; a2 = P.angle2[i2]; },
                                      P.match12);
        r1 += (size_t)P.n1;
        if (P.match21)
            for (int i = 0; i < P.n1; i++)
                if (P.match12[i] >= 0) P.match21[P.match12[i]] = i;
    }
    return MSORB_OK;
}
}  // namespace

extern "C" int msorb_search_by_bow(int device, msorb_bow_pair* pairs, int n_pairs, int th_low, int inclusive, float nnratio,
                                   int check_orientation, float* elapsed_ms) {
    return search_by_bow_impl(device, pairs, n_pairs, th_low, inclusive ? 1 : 0, nnratio, check_orientation, elapsed_ms, nullptr);
}

// ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) on a two-camera frame (F.Nleft != -1), ORBmatcher.cc:223-421 with the arms of
// :276-309 and :357-382: inside a BoW node every KeyFrame feature keeps a best / second over the frame's LEFT features and a best
// over its RIGHT features (rows >= n_left), both among the features no earlier KeyFrame feature has claimed.  The left match is
// taken as on a one-camera frame (<= TH_LOW, ratio test); the right match — only looked at when the LEFT best distance was <= TH_LOW,
// whatever the ratio test said (:330 encloses :357) — is taken at <= TH_LOW without a ratio test (`|| true`).  Left and right claims
// touch disjoint features, so the two arms are two runs of the node kernel: the left one also reports every KeyFrame feature's
// left best distance, which gates the right one.  Both feed ONE rotation histogram, per KeyFrame feature the left entry first.
//   pair: as msorb_search_by_bow (set 1 = the KeyFrame, set 2 = the frame's n2 = N features, left camera first; avail2 unused);
//   match21[n2]: the KeyFrame feature matched to frame feature j (vpMapPointMatches[j] = its map point), -1 none.
extern "C" int msorb_search_by_bow_rig(int device, msorb_bow_pair* pair, int n_left, int th_low, float nnratio, int check_orientation) {
    if (!pair || n_left < 0 || n_left > pair->n2 || !pair->match21 || (pair->n1 > 0 && !pair->valid1)) return MSORB_E_INVALID;
    msorb_bow_pair& P = *pair;
    P.nmatches = 0;
    FeatVec fb{P.fv2_nodes, P.fv2_node, P.fv2_begin, P.fv2_feat}, fa{P.fv1_nodes, P.fv1_node, P.fv1_begin, P.fv1_feat};
    std::vector<uint8_t> seen;
    if (P.n1 < 0 || P.n2 < 0 || !check_feature_vector(P.n2, fb, seen) || !check_feature_vector(P.n1, fa, seen) ||
        (check_orientation && ((P.n1 > 0 && !P.angle1) || (P.n2 > 0 && !P.angle2)))) {
        set_last_error("search_by_bow_rig: bad sizes / null arrays / feature vector not ascending, out of range or with a repeated feature");
        return MSORB_E_INVALID;
    }
    // the frame's FeatureVector split by camera (the node ids and the order inside a node stay)
    std::vector<int> nodeL, beginL{0}, featL, nodeR, beginR{0}, featR;
    for (int r = 0; r < fb.nodes; r++) {
        const size_t l0 = featL.size(), r0 = featR.size();
        for (int k = fb.begin[r]; k < fb.begin[r + 1]; k++) (fb.feat[k] < n_left ? featL : featR).push_back(fb.feat[k]);
        if (featL.size() > l0) { nodeL.push_back(fb.node[r]); beginL.push_back((int)featL.size()); }
        if (featR.size() > r0) { nodeR.push_back(fb.node[r]); beginR.push_back((int)featR.size()); }
    }
    std::vector<int> m12L(std::max(P.n1, 1), -1), m12R(std::max(P.n1, 1), -1), m21(std::max(P.n2, 1), -1);
    msorb_bow_pair A = P;
    A.avail2 = nullptr; A.match12 = m12L.data(); A.match21 = nullptr;
    A.fv2_nodes = (int)nodeL.size(); A.fv2_node = nodeL.data(); A.fv2_begin = beginL.data(); A.fv2_feat = featL.data();
    std::vector<std::vector<int>> best1;
    int rc = search_by_bow_impl(device, &A, 1, th_low, 1, nnratio, 0, nullptr, &best1);
    if (rc) return rc;
    std::vector<uint8_t> validR(std::max(P.n1, 1), 0);
    for (int i = 0; i < P.n1; i++) validR[i] = P.valid1[i] && best1[0][i] <= th_low;                // :330
    msorb_bow_pair B = P;
    B.valid1 = validR.data(); B.avail2 = nullptr; B.match12 = m12R.data(); B.match21 = nullptr;
    B.fv2_nodes = (int)nodeR.size(); B.fv2_node = nodeR.data(); B.fv2_begin = beginR.data(); B.fv2_feat = featR.data();
    rc = search_by_bow_impl(device, &B, 1, th_low, 1 | 2, nnratio, 0, nullptr, nullptr);
    if (rc) return rc;
    // the rotation histogram in the reference's order: the nodes both full vectors hold, the KeyFrame's features of a node in list
    // order, the left match of a feature before its right match (:338-353, :361-378); then ComputeThreeMaxima (:396-418)
    for (int j = 0; j < P.n2; j++) P.match21[j] = -1;
    std::vector<std::pair<int, int>> hist[kHistoLength];   // (frame feature, KeyFrame feature)
    const float factor = 1.0f / kHistoLength;
    int nm = 0;
    int i = 0, j = 0;
    while (i < fa.nodes && j < fb.nodes) {
        if (fa.node[i] == fb.node[j]) {
            for (int k = fa.begin[i]; k < fa.begin[i + 1]; k++) {
                const int i1 = fa.feat[k];
                for (int side = 0; side < 2; side++) {
                    const int i2 = side ? m12R[i1] : m12L[i1];
                    if (i2 < 0) continue;
                    P.match21[i2] = i1;
                    nm++;
                    if (check_orientation) {
                        float rot = P.angle1[i1] - P.angle2[i2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == kHistoLength) bin = 0;
                        if (bin >= 0 && bin < kHistoLength) hist[bin].push_back({i2, i1});
                        else { P.match21[i2] = -1; nm--; }
                    }
                }
            }
            i++; j++;
        } else if (fa.node[i] < fb.node[j]) i++;
        else j++;
    }
    if (check_orientation) {
        int sizes[kHistoLength], ind[3];
        for (int b = 0; b < kHistoLength; b++) sizes[b] = (int)hist[b].size();
        msorb_three_maxima(sizes, kHistoLength, ind);
        for (int b = 0; b < kHistoLength; b++)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (auto& e : hist[b]) { P.match21[e.first] = -1; nm--; }
    }
    if (P.match12) {   // the left partner of every KeyFrame feature (the right one is in match21 only)
        for (int k = 0; k < P.n1; k++) P.match12[k] = -1;
        for (int f2 = 0; f2 < n_left; f2++) if (P.match21[f2] >= 0) P.match12[P.match21[f2]] = f2;
    }
    P.nmatches = nm;
    return MSORB_OK;
}

// ORBmatcher::SearchForTriangulation (:1168-1402) with the geometric test of :1332 left to the caller — the form the two-camera arms
// (:1294-1330: one of four relative poses and two camera models per candidate, KannalaBrandt8::epipolarConstrain) need, and any
// camera model this library does not restate.  The device lists, per common node, every (query, train) within th_low among the
// visited / eligible features (node_candidates_kernel: count, then fill); the replay below walks them in the reference's order.
// For one query the reference scans the trains in list order, keeps `bestDist` (initially th_low) and takes a train when
// dist <= bestDist and the test passes: the winner is the passing train of smallest distance, the LAST of equal ones — here the
// candidates sorted by (distance, position descending), the first unclaimed one that accept() passes.  accept() is a pure
// predicate of the two features (the reference's is: a const camera, two keypoints, a relative pose), so asking it for fewer
// or other candidates than the reference's running-minimum scan reaches changes nothing.
extern "C" int msorb_search_for_triangulation_cb(int device, msorb_bow_pair* pair, int th_low, int check_orientation, msorb_pair_accept accept,
                                                 void* ctx) {
    if (!pair || !accept || th_low < 0) return MSORB_E_INVALID;
    msorb_bow_pair& P = *pair;
    P.nmatches = 0;
    FeatVec fa{P.fv1_nodes, P.fv1_node, P.fv1_begin, P.fv1_feat}, fb{P.fv2_nodes, P.fv2_node, P.fv2_begin, P.fv2_feat};
    std::vector<Common> common;
    std::vector<uint8_t> seen;
    size_t totf1 = 0, totf2 = 0;
    int max_chunks = 1;
    if (P.n1 < 0 || P.n2 < 0 || (!P.match12 && P.n1 > 0) || (P.n1 > 0 && (!P.desc1 || !P.valid1)) || (P.n2 > 0 && !P.desc2) ||
        (check_orientation && ((P.n1 > 0 && !P.angle1) || (P.n2 > 0 && !P.angle2))) || !check_feature_vector(P.n1, fa, seen) ||
        !check_feature_vector(P.n2, fb, seen) || !merge_walk(fa, fb, common, totf1, totf2, max_chunks)) {
        set_last_error("search_for_triangulation_cb: bad sizes / null arrays / feature vector not ascending, out of range or with a repeated feature");
        return MSORB_E_INVALID;
    }
    for (int i = 0; i < P.n1; i++) P.match12[i] = -1;
    if (P.match21) for (int j = 0; j < P.n2; j++) P.match21[j] = -1;
    const size_t n_items = common.size(), tot1 = (size_t)P.n1, tot2 = (size_t)P.n2;
    if (n_items == 0) return MSORB_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return no_device();
    // ---- staging: [desc1 | desc2 | feat1 | feat2 | items | valid1 | avail2 | begin] in, [count] out; the lists in their own block ----
    const size_t o_d1 = 0, o_d2 = o_d1 + tot1 * 32, o_f1 = o_d2 + tot2 * 32, o_f2 = o_f1 + up16(totf1 * 4), o_it = o_f2 + up16(totf2 * 4),
                 o_v1 = o_it + up16(n_items * sizeof(BowItem)), o_a2 = o_v1 + up16(tot1), in_bytes = o_a2 + up16(tot2), o_cnt = in_bytes,
                 o_beg = o_cnt + up16(n_items * 4), total = o_beg + up16(n_items * 4);
    static thread_local Scratch scr, lists;
    hipError_t e = scr.acquire(device, total);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation_cb", e);
    {
        char* h = scr.h;
        size_t k1 = 0, k2 = 0, ni = 0;
        if (P.n1) { std::memcpy(h + o_d1, P.desc1, tot1 * 32); std::memcpy(h + o_v1, P.valid1, tot1); }
        if (P.n2) {
            std::memcpy(h + o_d2, P.desc2, tot2 * 32);
            if (P.avail2) std::memcpy(h + o_a2, P.avail2, tot2);
            else std::memset(h + o_a2, 1, tot2);
        }
        stage_lists(fa, fb, common, 0, 0, 0, (int*)(h + o_f1), (int*)(h + o_f2), k1, k2, (BowItem*)(h + o_it), ni);
    }
    hipStream_t s = scr.s;
    char* d = scr.d;
    e = msorb::small_copy(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(node_candidates_kernel<false>, dim3((unsigned)n_items), dim3(64), 0, s, (const BowItem*)(d + o_it), (const uint4*)(d + o_d1),
                           (const uint4*)(d + o_d2), (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2), (const int*)(d + o_f1),
                           (const int*)(d + o_f2), th_low, (int*)(d + o_cnt), (const int*)nullptr, (int4*)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = msorb::small_copy(scr.h + o_cnt, d + o_cnt, n_items * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation_cb", e);
    const int* cnt = (const int*)(scr.h + o_cnt);
    int* beg = (int*)(scr.h + o_beg);
    size_t n_cand = 0;
    for (size_t i = 0; i < n_items; i++) { beg[i] = (int)n_cand; n_cand += (size_t)cnt[i]; }
    std::vector<int> raw(tot1, -1);
    if (n_cand > 0) {
        if (n_cand > (size_t)INT32_MAX / 16) { set_last_error("search_for_triangulation_cb: too many candidates"); return MSORB_E_CAPACITY; }
        e = lists.acquire(device, n_cand * sizeof(int4));
        if (e != hipSuccess) return hip_fail(lists, "search_for_triangulation_cb", e);
        e = msorb::small_copy(d + o_beg, beg, n_items * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(node_candidates_kernel<true>, dim3((unsigned)n_items), dim3(64), 0, s, (const BowItem*)(d + o_it), (const uint4*)(d + o_d1),
                               (const uint4*)(d + o_d2), (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2), (const int*)(d + o_f1),
                               (const int*)(d + o_f2), th_low, (int*)nullptr, (const int*)(d + o_beg), (int4*)lists.d);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = msorb::small_copy(lists.h, lists.d, n_cand * sizeof(int4), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation_cb", e);
        // ---- the replay: nodes ascending, a node's queries in list order (:1230-1358) ----
        std::vector<int4> all((const int4*)lists.h, (const int4*)lists.h + n_cand);   // (out of the pinned block: read repeatedly below)
        std::vector<uint8_t> matched2(tot2, 0);                                        // vbMatched2 (:1212)
        std::vector<int4> group;
        size_t k = 0;
        while (k < n_cand) {
            size_t k_end = k;
            while (k_end < n_cand && all[k_end].x == all[k].x) k_end++;                // one query's candidates (contiguous: the fill order)
            group.assign(all.begin() + k, all.begin() + k_end);
            std::sort(group.begin(), group.end(), [](const int4& a, const int4& b) { return a.z != b.z ? a.z < b.z : a.w > b.w; });
            for (const int4& c : group) {
                if (matched2[c.y]) continue;                                            // :1262
                if (!accept(ctx, c.x, c.y)) continue;                                  // :1332
                raw[c.x] = c.y;
                matched2[c.y] = 1;                                                      // :1345
                break;
            }
            k = k_end;
        }
    }
    P.nmatches = replay_histogram(fa, common, raw.data(), check_orientation,
                                  [&](int i1, int i2, float& a1, float& a2) { a1 = P.angle1[i1]; a2 = P.angle2[i2]; }, P.match12);
    if (P.match21)
        for (int i = 0; i < P.n1; i++)
            if (P.match12[i] >= 0) P.match21[P.match12[i]] = i;
    return MSORB_OK;
}

extern "C" int msorb_search_for_triangulation(int device, msorb_triangulation_pair* pairs, int n_pairs, int coarse,
                                              int check_orientation, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSORB_E_INVALID;
    if (n_pairs == 0) return MSORB_OK;
    std::vector<std::vector<Common>> common(n_pairs);
    std::vector<FeatVec> fa(n_pairs), fb(n_pairs);
    std::vector<uint8_t> seen;
    size_t tot1 = 0, tot2 = 0, totf1 = 0, totf2 = 0, n_items = 0;
    int max_chunks = 1;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_triangulation_pair& P = pairs[pi];
        P.nmatches = 0;
        fa[pi] = FeatVec{P.fv1_nodes, P.fv1_node, P.fv1_begin, P.fv1_feat};
        fb[pi] = FeatVec{P.fv2_nodes, P.fv2_node, P.fv2_begin, P.fv2_feat};
        bool ok = P.n1 >= 0 && P.n2 >= 0 && (P.n1 == 0 || (P.match12 && P.desc1 && P.valid1 && P.stereo1 && P.kp1)) &&
                  (P.n2 == 0 || (P.desc2 && P.avail2 && P.stereo2 && P.kp2 && P.scale_factors2 && P.level_sigma2_2 &&
                                 P.n_levels2 > 0));
        for (int j = 0; ok && j < P.n2; j++) ok = P.kp2[j].octave >= 0 && P.kp2[j].octave < P.n_levels2;
        if (!ok || !check_feature_vector(P.n1, fa[pi], seen) || !check_feature_vector(P.n2, fb[pi], seen) ||
            !merge_walk(fa[pi], fb[pi], common[pi], totf1, totf2, max_chunks)) {
            set_last_error("search_for_triangulation: pair " + std::to_string(pi) +
                           ": bad sizes / null arrays / octave out of range / feature vector not ascending, out of range or "
                           "with a repeated feature");
            return MSORB_E_INVALID;
        }
        n_items += common[pi].size();
        tot1 += (size_t)P.n1;
        tot2 += (size_t)P.n2;
    }
    for (int pi = 0; pi < n_pairs; pi++)
        for (int i = 0; i < pairs[pi].n1; i++) pairs[pi].match12[i] = -1;
    if (n_items == 0) return MSORB_OK;
    if (tot1 > (size_t)INT32_MAX / 2 || tot2 > (size_t)INT32_MAX / 2) return MSORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return no_device();
    // ---- staging: [desc1 | desc2 | tr2 (x, y, 100*scale, sigma2) | xy1 | feat1 | feat2 | items | consts | flags1 | flags2] ----
    const size_t o_d1 = 0, o_d2 = o_d1 + tot1 * 32, o_t2 = o_d2 + tot2 * 32, o_x1 = o_t2 + tot2 * 16, o_f1 = o_x1 + up16(tot1 * 8),
                 o_f2 = o_f1 + up16(totf1 * 4), o_it = o_f2 + up16(totf2 * 4), o_c = o_it + up16(n_items * sizeof(BowItem)),
                 o_v1 = o_c + up16((size_t)n_pairs * sizeof(TriConst)), o_a2 = o_v1 + up16(tot1), in_bytes = o_a2 + up16(tot2),
                 o_m = in_bytes, total = o_m + up16(tot1 * 4);
    static thread_local Scratch scr;
    hipError_t e = scr.acquire(device, total);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation", e);
    {
        char* h = scr.h;
        size_t r1 = 0, r2 = 0, k1 = 0, k2 = 0, ni = 0;
        TriConst* consts = (TriConst*)(h + o_c);
        for (int pi = 0; pi < n_pairs; pi++) {
            const msorb_triangulation_pair& P = pairs[pi];
            if (P.n1) std::memcpy(h + o_d1 + r1 * 32, P.desc1, (size_t)P.n1 * 32);
            if (P.n2) std::memcpy(h + o_d2 + r2 * 32, P.desc2, (size_t)P.n2 * 32);
            float* xy = (float*)(h + o_x1) + 2 * r1;
            uint8_t* f1 = (uint8_t*)(h + o_v1) + r1;
            for (int i = 0; i < P.n1; i++) {
                xy[2 * i] = P.kp1[i].x;
                xy[2 * i + 1] = P.kp1[i].y;
                f1[i] = (uint8_t)((P.valid1[i] ? 1 : 0) | (P.stereo1[i] ? 2 : 0));
            }
            float* tr = (float*)(h + o_t2) + 4 * r2;
            uint8_t* f2 = (uint8_t*)(h + o_a2) + r2;
            for (int j = 0; j < P.n2; j++) {
                const int oct = P.kp2[j].octave;
                tr[4 * j] = P.kp2[j].x;
                tr[4 * j + 1] = P.kp2[j].y;
                tr[4 * j + 2] = 100 * P.scale_factors2[oct];  // :1287 (int * float -> float)
                tr[4 * j + 3] = P.level_sigma2_2[oct];        // :1332
                f2[j] = (uint8_t)((P.avail2[j] ? 1 : 0) | (P.stereo2[j] ? 2 : 0));
            }
            std::memcpy(consts[pi].F, P.F12, sizeof(P.F12));
            consts[pi].ep[0] = P.ep[0];
            consts[pi].ep[1] = P.ep[1];
            stage_lists(fa[pi], fb[pi], common[pi], pi, r1, r2, (int*)(h + o_f1), (int*)(h + o_f2), k1, k2,
                        (BowItem*)(h + o_it), ni);
            r1 += (size_t)P.n1;
            r2 += (size_t)P.n2;
        }
    }
    hipStream_t s = scr.s;
    char* d = scr.d;
    e = msorb::small_copy(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, tot1 * 4, s);
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(triangulation_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s,
                           (const BowItem*)(d + o_it), (const TriConst*)(d + o_c), (const uint4*)(d + o_d1),
                           (const uint4*)(d + o_d2), (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2),
                           (const float2*)(d + o_x1), (const float4*)(d + o_t2), (const int*)(d + o_f1), (const int*)(d + o_f2),
                           coarse, (int*)(d + o_m));
        e = hipGetLastError();
    }
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = msorb::small_copy(scr.h + o_m, d + o_m, tot1 * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation", e);
    // the raw matches are read feature by feature below: out of the pinned block first (one streaming copy) — scattered 4-byte
    // reads of pinned host memory cost ~10 ns each, 0.8 ms for a 32-pair batch
    static thread_local std::vector<int> m_local;
    m_local.resize(tot1);
    std::memcpy(m_local.data(), scr.h + o_m, tot1 * 4);
    const int* m_all = m_local.data();
    size_t r1 = 0;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_triangulation_pair& P = pairs[pi];
        P.nmatches = replay_histogram(
            fa[pi], common[pi], m_all + r1, check_orientation,
            [&](int i1, int i2, float& a1, float& a2) { a1 = P.kp1[i1].angle; a2 = P.kp2[i2].angle; }, P.match12);
        r1 += (size_t)P.n1;
    }
    return MSORB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Resident KeyFrames (DESIGN.md "what comes next" of round 1: the per-call entries above spend their time staging the
// same KeyFrames again and again — 30-40 us of kernel inside ~1 ms of copies).  A KeyFrame is matched against many
// others over its life (LocalMapping::CreateNewMapPoints against 10-20 neighbours per new KeyFrame, relocalisation and
// loop candidates against the current frame), while its descriptors, keypoints and FeatureVector are fixed once
// KeyFrame::ComputeBoW has run (until map sparsification compacts it: remove + add).  The store keeps them on the
// device; a search then uploads only the visit / availability flags (1 B per feature), the (pair, common node) work
// items and, for the KeyFrame-vs-Frame form, the frame; it downloads match12.
// ------------------------------------------------------------------------------------------------------------------
// Offsets inside one of the store's arenas (feature rows; FeatureVector entries): first fit over the free ranges, neighbours merged on
// release, the arena's end pulled back when its last range is released.  A KeyFrame that is removed (culled, compacted by map
// sparsification, evicted) gives its rows back — a sequence inserts and culls KeyFrames for as long as it runs (tests/soak_main.cc:
// 268 MB after 100 000 frames when removed rows stayed allocated).
struct RangeAlloc {
    size_t end = 0;                    // first offset past the highest range in use
    std::map<size_t, size_t> free_;    // offset -> length, disjoint, non-adjacent, all below `end`
    size_t take(size_t n) {
        if (n == 0) return 0;
        for (auto it = free_.begin(); it != free_.end(); ++it)
            if (it->second >= n) {
                const size_t off = it->first, rest = it->second - n;
                free_.erase(it);
                if (rest) free_[off + n] = rest;
                return off;
            }
        const size_t off = end;
        end += n;
        return off;
    }
    void give(size_t off, size_t n) {
        if (n == 0) return;
        auto nx = free_.lower_bound(off);
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { off = pv->first; n += pv->second; free_.erase(pv); }
        }
        if (nx != free_.end() && off + n == nx->first) { n += nx->second; free_.erase(nx); }
        if (off + n == end) end = off;
        else free_[off] = n;
    }
    size_t free_total() const { size_t t = 0; for (auto& e : free_) t += e.second; return t; }
};

struct msorb_kf_store {
    int device = 0;
    mutable std::shared_mutex mu;  // searches hold it shared (the buffers must not move under a running kernel), add / remove exclusive
    struct Entry {
        bool alive = false;
        int row0 = 0, n = 0, feat0 = 0, nfeat = 0;
        std::vector<int> node, begin;   // FeatureVector: node ids ascending, list r = feat[begin[r] .. begin[r+1]) (offsets relative to feat0)
        std::vector<int> feat;          // host copy of the lists (the rotation-histogram replay walks them)
        std::vector<float> angle;
    };
    std::vector<Entry> kf;
    std::vector<int> dead_ids;   // indices of `kf` whose KeyFrame was removed: handed out again by the next add
    int n_alive = 0;
    RangeAlloc rows_a, feats_a;  // which rows / FeatureVector entries of the device arrays are in use
    size_t rows_cap = 0, feats_cap = 0;
    uint4* d_desc = nullptr;    // 2 per row
    float2* d_xy = nullptr;     // keypoint position (SearchForTriangulation, set 1)
    float4* d_tr = nullptr;     // x, y, 100 * scale[octave], sigma2[octave] (SearchForTriangulation, set 2)
    float* d_angle = nullptr;   // keypoint angles (rotation histogram)
    int* d_feat = nullptr;
};

namespace {
template <class T>
hipError_t grow(T*& p, size_t used, size_t& cap, size_t need, size_t unit) {
    if (need <= cap) return hipSuccess;
    const size_t ncap = std::max(need, cap * 2 + 4096);
    T* q = nullptr;
    hipError_t e = hipMalloc((void**)&q, ncap * unit * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && used) e = hipMemcpy(q, p, used * unit * sizeof(T), hipMemcpyDeviceToDevice);
    if (p) (void)hipFree(p);
    p = q;
    cap = ncap;
    return e;
}
}  // namespace

extern "C" int msorb_kf_store_create(int device, msorb_kf_store** out) {
    if (!out) return MSORB_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return no_device();
    msorb_kf_store* s = new msorb_kf_store();
    s->device = device;
    *out = s;
    return MSORB_OK;
}

extern "C" void msorb_kf_store_destroy(msorb_kf_store* s) {
    if (!s) return;
    if (hipSetDevice(s->device) == hipSuccess) {
        if (s->d_desc) (void)hipFree(s->d_desc);
        if (s->d_xy) (void)hipFree(s->d_xy);
        if (s->d_tr) (void)hipFree(s->d_tr);
        if (s->d_feat) (void)hipFree(s->d_feat);
        if (s->d_angle) (void)hipFree(s->d_angle);
    }
    delete s;
}

extern "C" int msorb_kf_store_count(const msorb_kf_store* s) {
    if (!s) return MSORB_E_INVALID;
    std::shared_lock<std::shared_mutex> lk(s->mu);
    return s->n_alive;
}

extern "C" int msorb_kf_store_add(msorb_kf_store* s, int n, const msorb_keypoint* kps, const uint8_t* desc, int fv_nodes,
                                  const int* fv_node, const int* fv_begin, const int* fv_feat, const float* scale_factors,
                                  const float* level_sigma2, int n_levels, int* kf_id) {
    if (!s || !kf_id || n < 0 || (n > 0 && (!kps || !desc)) || !scale_factors || !level_sigma2 || n_levels < 1) return MSORB_E_INVALID;
    *kf_id = -1;
    FeatVec fv{fv_nodes, fv_node, fv_begin, fv_feat};
    std::vector<uint8_t> seen;
    if (!check_feature_vector(n, fv, seen)) { set_last_error("kf_store_add: feature vector not ascending, out of range or with a repeated feature"); return MSORB_E_INVALID; }
    for (int i = 0; i < n; i++)
        if (kps[i].octave < 0 || kps[i].octave >= n_levels) { set_last_error("kf_store_add: keypoint octave out of range"); return MSORB_E_INVALID; }
    const int f_lo = fv_nodes ? fv_begin[0] : 0, f_hi = fv_nodes ? fv_begin[fv_nodes] : 0, nf = f_hi - f_lo;
    std::vector<float> xy((size_t)2 * n), tr((size_t)4 * n);
    for (int i = 0; i < n; i++) {
        xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y;
        tr[4 * i] = kps[i].x; tr[4 * i + 1] = kps[i].y;
        tr[4 * i + 2] = 100 * scale_factors[kps[i].octave];   // ORBmatcher.cc:1287
        tr[4 * i + 3] = level_sigma2[kps[i].octave];          // :1332
    }
    std::unique_lock<std::shared_mutex> lk(s->mu);
    if (hipSetDevice(s->device) != hipSuccess) return MSORB_E_HIP;
    hipError_t e = hipSuccess;
    // rows / FeatureVector entries: a free range of an earlier KeyFrame first, the end of the arena otherwise
    const size_t used_rows = s->rows_a.end, used_feats = s->feats_a.end;   // what a reallocation has to carry over
    const size_t row0 = s->rows_a.take((size_t)n), feat0 = s->feats_a.take((size_t)nf);
    size_t cap2 = s->rows_cap, cap3 = s->rows_cap, cap1 = s->rows_cap, cap4 = s->rows_cap;
    e = grow(s->d_desc, used_rows, cap1, s->rows_a.end, 2);
    if (e == hipSuccess) e = grow(s->d_xy, used_rows, cap2, s->rows_a.end, 1);
    if (e == hipSuccess) e = grow(s->d_tr, used_rows, cap3, s->rows_a.end, 1);
    if (e == hipSuccess) e = grow(s->d_angle, used_rows, cap4, s->rows_a.end, 1);
    if (e == hipSuccess) s->rows_cap = std::min(std::min(cap1, cap4), std::min(cap2, cap3));
    if (e == hipSuccess) e = grow(s->d_feat, used_feats, s->feats_cap, s->feats_a.end, 1);
    if (e == hipSuccess && n) e = hipMemcpy(s->d_desc + 2 * row0, desc, (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(s->d_xy + row0, xy.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(s->d_tr + row0, tr.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && nf) e = hipMemcpy(s->d_feat + feat0, fv_feat + f_lo, (size_t)nf * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) {
        std::vector<float> ang(n);
        for (int i = 0; i < n; i++) ang[i] = kps[i].angle;
        e = hipMemcpy(s->d_angle + row0, ang.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        s->rows_a.give(row0, (size_t)n); s->feats_a.give(feat0, (size_t)nf);
        set_last_error(std::string("kf_store_add: ") + hipGetErrorString(e));
        return MSORB_E_HIP;
    }
    msorb_kf_store::Entry E;
    E.alive = true; E.row0 = (int)row0; E.n = n; E.feat0 = (int)feat0; E.nfeat = nf;
    E.node.assign(fv_node, fv_node + fv_nodes);
    E.begin.resize(fv_nodes + 1);
    for (int r = 0; r <= fv_nodes; r++) E.begin[r] = (fv_nodes ? fv_begin[r] : 0) - f_lo;
    E.feat.assign(fv_feat + f_lo, fv_feat + f_hi);
    E.angle.resize(n);
    for (int i = 0; i < n; i++) E.angle[i] = kps[i].angle;
    s->n_alive++;
    if (!s->dead_ids.empty()) {   // ids of removed KeyFrames come back: the table does not grow with the length of the sequence
        *kf_id = s->dead_ids.back();
        s->dead_ids.pop_back();
        s->kf[*kf_id] = std::move(E);
    } else {
        s->kf.push_back(std::move(E));
        *kf_id = (int)s->kf.size() - 1;
    }
    return MSORB_OK;
}

extern "C" int msorb_kf_store_remove(msorb_kf_store* s, int kf_id) {
    if (!s) return MSORB_E_INVALID;
    std::unique_lock<std::shared_mutex> lk(s->mu);
    if (kf_id < 0 || kf_id >= (int)s->kf.size() || !s->kf[kf_id].alive) { set_last_error("kf_store_remove: unknown KeyFrame id"); return MSORB_E_INVALID; }
    msorb_kf_store::Entry& E = s->kf[kf_id];
    E.alive = false;
    s->rows_a.give((size_t)E.row0, (size_t)E.n);       // the rows and the id are free for the next add (no kernel is running: the lock is exclusive)
    s->feats_a.give((size_t)E.feat0, (size_t)E.nfeat);
    E.n = 0; E.nfeat = 0;
    std::vector<int>().swap(E.node); E.begin.assign(1, 0); std::vector<float>().swap(E.angle); std::vector<int>().swap(E.feat);
    s->dead_ids.push_back(kf_id);
    s->n_alive--;
    return MSORB_OK;
}

// Rows of the device arrays in use / reserved (rows of 64 + 8 + 16 + 4 bytes): a store's footprint for long-run checks.
extern "C" int msorb_kf_store_rows(const msorb_kf_store* s, size_t* rows_in_use, size_t* rows_reserved) {
    if (!s || !rows_in_use || !rows_reserved) return MSORB_E_INVALID;
    std::shared_lock<std::shared_mutex> lk(s->mu);
    *rows_in_use = s->rows_a.end - s->rows_a.free_total();
    *rows_reserved = s->rows_cap;
    return MSORB_OK;
}

namespace {
// items of one pair from the stores' host copies of the two FeatureVectors; lists are addressed inside the resident /
// staged feat arrays through absolute offsets
void items_of(const std::vector<int>& node1, const std::vector<int>& begin1, int feat0_1, const int* node2, const int* begin2, int nodes2,
              int feat0_2, int base1, int base2, int m1, int m2, int pi, std::vector<BowItem>& items, std::vector<Common>& common,
              int& max_chunks, bool& ok) {
    int i = 0, j = 0;
    const int nodes1 = (int)node1.size();
    while (i < nodes1 && j < nodes2) {
        if (node1[i] == node2[j]) {
            const int l1 = begin1[i + 1] - begin1[i], l2 = begin2[j + 1] - begin2[j];
            if (l1 > 0 && l2 > 0) {
                if (l2 >= (1 << 20)) { ok = false; return; }
                common.push_back({i, j});
                items.push_back(BowItem{base1, base2, feat0_1 + begin1[i], l1, feat0_2 + begin2[j], l2, pi, m1, m2});
                max_chunks = std::max(max_chunks, (l2 + 63) >> 6);
            }
            i++; j++;
        } else if (node1[i] < node2[j]) i++;
        else j++;
    }
}
}  // namespace

extern "C" int msorb_search_by_bow_kf(msorb_kf_store* st, msorb_bow_kf_pair* pairs, int n_pairs, const msorb_bow_frame* frame,
                                      int th_low, int inclusive, float nnratio, int check_orientation, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (!st || n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSORB_E_INVALID;
    if (n_pairs == 0) return MSORB_OK;
    std::shared_lock<std::shared_mutex> lk(st->mu);
    std::vector<uint8_t> seen;
    FeatVec ff{0, nullptr, nullptr, nullptr};
    if (frame) {
        ff = FeatVec{frame->fv_nodes, frame->fv_node, frame->fv_begin, frame->fv_feat};
        if (frame->n < 0 || (frame->n > 0 && !frame->desc) || (check_orientation && frame->n > 0 && !frame->angle) ||
            !check_feature_vector(frame->n, ff, seen)) {
            set_last_error("search_by_bow_kf: bad frame arrays");
            return MSORB_E_INVALID;
        }
    }
    // per-call flag arrays: [valid1 of pair 0 | pair 1 | ...] and [avail2 of pair 0 | ...]
    std::vector<BowItem> items;
    std::vector<std::vector<Common>> common(n_pairs);
    std::vector<int> m1(n_pairs), m2(n_pairs);
    size_t tot1 = 0, tot2 = 0;
    int max_chunks = 1;
    const int fr_feat_lo = frame && frame->fv_nodes ? frame->fv_begin[0] : 0;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_kf_pair& P = pairs[pi];
        P.nmatches = 0;
        const bool vs_frame = P.kf2 < 0;
        if (P.kf1 < 0 || P.kf1 >= (int)st->kf.size() || !st->kf[P.kf1].alive || (vs_frame && !frame) ||
            (!vs_frame && (P.kf2 >= (int)st->kf.size() || !st->kf[P.kf2].alive)) || (vs_frame != (frame != nullptr))) {
            set_last_error("search_by_bow_kf: pair " + std::to_string(pi) + ": unknown KeyFrame id, or KeyFrame / frame trains mixed in one call");
            return MSORB_E_INVALID;
        }
        const msorb_kf_store::Entry& A = st->kf[P.kf1];
        const int n2 = vs_frame ? frame->n : st->kf[P.kf2].n;
        if ((A.n > 0 && (!P.valid1 || !P.match12))) { set_last_error("search_by_bow_kf: null valid1 / match12"); return MSORB_E_INVALID; }
        m1[pi] = (int)tot1; m2[pi] = (int)tot2;
        bool ok = true;
        if (vs_frame) {
            // the frame's lists are staged at feat offset 0 of the per-call frame block; descriptors at row 0 of that block
            std::vector<int> rel(frame->fv_nodes + 1);
            for (int r = 0; r <= frame->fv_nodes; r++) rel[r] = (frame->fv_nodes ? frame->fv_begin[r] : 0) - fr_feat_lo;
            items_of(A.node, A.begin, A.feat0, frame->fv_node, rel.data(), frame->fv_nodes, 0, A.row0, 0, m1[pi], m2[pi], pi, items,
                     common[pi], max_chunks, ok);
        } else {
            const msorb_kf_store::Entry& B = st->kf[P.kf2];
            items_of(A.node, A.begin, A.feat0, B.node.data(), B.begin.data(), (int)B.node.size(), B.feat0, A.row0, B.row0, m1[pi], m2[pi],
                     pi, items, common[pi], max_chunks, ok);
        }
        if (!ok) { set_last_error("search_by_bow_kf: node list too long"); return MSORB_E_INVALID; }
        tot1 += (size_t)A.n;
        tot2 += (size_t)n2;
    }
    if (items.empty()) {   // no common node anywhere: nothing is launched, every feature stays unmatched
        for (int pi = 0; pi < n_pairs; pi++) {
            msorb_bow_kf_pair& P = pairs[pi];
            const int n1 = st->kf[P.kf1].n, n2 = P.kf2 < 0 ? frame->n : st->kf[P.kf2].n;
            for (int i = 0; i < n1; i++) P.match12[i] = -1;
            if (P.match21) for (int j = 0; j < n2; j++) P.match21[j] = -1;
        }
        return MSORB_OK;
    }
    const size_t n_items = items.size();
    const size_t fr_rows = frame ? (size_t)frame->n : 0, fr_feats = frame && frame->fv_nodes ? (size_t)(frame->fv_begin[frame->fv_nodes] - fr_feat_lo) : 0;
    // staging: [frame desc | frame feat | frame angle | items | posts | valid1 | avail2] in, [match12 | match21 | nmatches] out
    const size_t o_fd = 0, o_ff = o_fd + fr_rows * 32, o_fa = o_ff + up16(fr_feats * 4), o_it = o_fa + up16(fr_rows * 4),
                 o_po = o_it + up16(n_items * sizeof(BowItem)), o_v1 = o_po + up16((size_t)n_pairs * sizeof(PairPost)),
                 o_a2 = o_v1 + up16(tot1), in_bytes = o_a2 + up16(tot2), o_m = in_bytes, o_m21 = o_m + up16(tot1 * 4),
                 o_nm = o_m21 + up16(tot2 * 4), total = o_nm + up16((size_t)n_pairs * 4), out_bytes = total - o_m;
    static thread_local Scratch scr;
    hipError_t e = scr.acquire(st->device, total);
    if (e != hipSuccess) return hip_fail(scr, "search_by_bow_kf", e);
    {
        char* h = scr.h;
        if (fr_rows) std::memcpy(h + o_fd, frame->desc, fr_rows * 32);
        if (fr_feats) std::memcpy(h + o_ff, frame->fv_feat + fr_feat_lo, fr_feats * 4);
        if (fr_rows && frame->angle) std::memcpy(h + o_fa, frame->angle, fr_rows * 4);
        std::memcpy(h + o_it, items.data(), n_items * sizeof(BowItem));
        PairPost* posts = (PairPost*)(h + o_po);
        for (int pi = 0; pi < n_pairs; pi++) {
            const msorb_bow_kf_pair& P = pairs[pi];
            const msorb_kf_store::Entry& A = st->kf[P.kf1];
            const int n1 = A.n, n2 = P.kf2 < 0 ? frame->n : st->kf[P.kf2].n;
            if (n1) std::memcpy(h + o_v1 + m1[pi], P.valid1, (size_t)n1);
            if (n2) {
                if (P.avail2) std::memcpy(h + o_a2 + m2[pi], P.avail2, (size_t)n2);
                else std::memset(h + o_a2 + m2[pi], 1, (size_t)n2);
            }
            posts[pi] = PairPost{m1[pi], n1, m2[pi], n2, A.row0, P.kf2 < 0 ? 0 : st->kf[P.kf2].row0};
        }
    }
    hipStream_t s = scr.s;
    char* d = scr.d;
    e = msorb::small_copy(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, o_nm - o_m, s);   // match12 and match21 = -1
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        const uint4* desc2 = frame ? (const uint4*)(d + o_fd) : st->d_desc;
        const int* feat2 = frame ? (const int*)(d + o_ff) : st->d_feat;
        hipLaunchKernelGGL(bow_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s, (const BowItem*)(d + o_it),
                           st->d_desc, desc2, (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2), st->d_feat, feat2, th_low,
                           inclusive ? 1 : 0, nnratio, (int*)(d + o_m));
        hipLaunchKernelGGL(pair_histogram_kernel, dim3((unsigned)n_pairs), dim3(256), 0, s, (const PairPost*)(d + o_po), st->d_angle,
                           frame ? (const float*)(d + o_fa) : st->d_angle, check_orientation, (int*)(d + o_m), (int*)(d + o_m21),
                           (int*)(d + o_nm));
        e = hipGetLastError();
    }
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = msorb::small_copy(scr.h + o_m, d + o_m, out_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_by_bow_kf", e);
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_kf_pair& P = pairs[pi];
        const int n1 = st->kf[P.kf1].n, n2 = P.kf2 < 0 ? frame->n : st->kf[P.kf2].n;
        if (n1) std::memcpy(P.match12, scr.h + o_m + (size_t)m1[pi] * 4, (size_t)n1 * 4);
        if (P.match21 && n2) std::memcpy(P.match21, scr.h + o_m21 + (size_t)m2[pi] * 4, (size_t)n2 * 4);
        P.nmatches = ((const int*)(scr.h + o_nm))[pi];
    }
    return MSORB_OK;
}

extern "C" int msorb_search_for_triangulation_kf(msorb_kf_store* st, msorb_triangulation_kf_pair* pairs, int n_pairs, int coarse,
                                                 int check_orientation, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (!st || n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSORB_E_INVALID;
    if (n_pairs == 0) return MSORB_OK;
    std::shared_lock<std::shared_mutex> lk(st->mu);
    std::vector<BowItem> items;
    std::vector<std::vector<Common>> common(n_pairs);
    std::vector<int> m1(n_pairs), m2(n_pairs);
    size_t tot1 = 0, tot2 = 0;
    int max_chunks = 1;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_triangulation_kf_pair& P = pairs[pi];
        P.nmatches = 0;
        if (P.kf1 < 0 || P.kf1 >= (int)st->kf.size() || !st->kf[P.kf1].alive || P.kf2 < 0 || P.kf2 >= (int)st->kf.size() ||
            !st->kf[P.kf2].alive) {
            set_last_error("search_for_triangulation_kf: pair " + std::to_string(pi) + ": unknown KeyFrame id");
            return MSORB_E_INVALID;
        }
        const msorb_kf_store::Entry &A = st->kf[P.kf1], &B = st->kf[P.kf2];
        if ((A.n > 0 && (!P.valid1 || !P.stereo1 || !P.match12)) || (B.n > 0 && (!P.avail2 || !P.stereo2))) {
            set_last_error("search_for_triangulation_kf: null flag arrays / match12");
            return MSORB_E_INVALID;
        }
        m1[pi] = (int)tot1; m2[pi] = (int)tot2;
        bool ok = true;
        items_of(A.node, A.begin, A.feat0, B.node.data(), B.begin.data(), (int)B.node.size(), B.feat0, A.row0, B.row0, m1[pi], m2[pi], pi,
                 items, common[pi], max_chunks, ok);
        if (!ok) { set_last_error("search_for_triangulation_kf: node list too long"); return MSORB_E_INVALID; }
        tot1 += (size_t)A.n;
        tot2 += (size_t)B.n;
    }
    if (items.empty()) {
        for (int pi = 0; pi < n_pairs; pi++)
            for (int i = 0; i < st->kf[pairs[pi].kf1].n; i++) pairs[pi].match12[i] = -1;
        return MSORB_OK;
    }
    const size_t n_items = items.size();
    // staging: [items | consts | posts | flags1 | flags2] in, [match12 | nmatches] out
    const size_t o_it = 0, o_c = o_it + up16(n_items * sizeof(BowItem)), o_po = o_c + up16((size_t)n_pairs * sizeof(TriConst)),
                 o_v1 = o_po + up16((size_t)n_pairs * sizeof(PairPost)), o_a2 = o_v1 + up16(tot1), in_bytes = o_a2 + up16(tot2),
                 o_m = in_bytes, o_nm = o_m + up16(tot1 * 4), total = o_nm + up16((size_t)n_pairs * 4), out_bytes = total - o_m;
    static thread_local Scratch scr;
    hipError_t e = scr.acquire(st->device, total);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation_kf", e);
    {
        char* h = scr.h;
        std::memcpy(h + o_it, items.data(), n_items * sizeof(BowItem));
        TriConst* consts = (TriConst*)(h + o_c);
        PairPost* posts = (PairPost*)(h + o_po);
        for (int pi = 0; pi < n_pairs; pi++) {
            const msorb_triangulation_kf_pair& P = pairs[pi];
            const msorb_kf_store::Entry &A = st->kf[P.kf1], &B = st->kf[P.kf2];
            uint8_t* f1 = (uint8_t*)(h + o_v1) + m1[pi];
            uint8_t* f2 = (uint8_t*)(h + o_a2) + m2[pi];
            for (int i = 0; i < A.n; i++) f1[i] = (uint8_t)((P.valid1[i] ? 1 : 0) | (P.stereo1[i] ? 2 : 0));
            for (int j = 0; j < B.n; j++) f2[j] = (uint8_t)((P.avail2[j] ? 1 : 0) | (P.stereo2[j] ? 2 : 0));
            std::memcpy(consts[pi].F, P.F12, sizeof(P.F12));
            consts[pi].ep[0] = P.ep[0];
            consts[pi].ep[1] = P.ep[1];
            posts[pi] = PairPost{m1[pi], A.n, m2[pi], B.n, A.row0, B.row0};
        }
    }
    hipStream_t s = scr.s;
    char* d = scr.d;
    e = msorb::small_copy(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, tot1 * 4, s);
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(triangulation_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s,
                           (const BowItem*)(d + o_it), (const TriConst*)(d + o_c), st->d_desc, st->d_desc, (const uint8_t*)(d + o_v1),
                           (const uint8_t*)(d + o_a2), st->d_xy, st->d_tr, st->d_feat, st->d_feat, coarse, (int*)(d + o_m));
        hipLaunchKernelGGL(pair_histogram_kernel, dim3((unsigned)n_pairs), dim3(256), 0, s, (const PairPost*)(d + o_po), st->d_angle,
                           st->d_angle, check_orientation, (int*)(d + o_m), (int*)nullptr, (int*)(d + o_nm));
        e = hipGetLastError();
    }
    if (e == hipSuccess && elapsed_ms) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = msorb::small_copy(scr.h + o_m, d + o_m, out_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation_kf", e);
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_triangulation_kf_pair& P = pairs[pi];
        const int n1 = st->kf[P.kf1].n;
        if (n1) std::memcpy(P.match12, scr.h + o_m + (size_t)m1[pi] * 4, (size_t)n1 * 4);
        P.nmatches = ((const int*)(scr.h + o_nm))[pi];
    }
    return MSORB_OK;
}
