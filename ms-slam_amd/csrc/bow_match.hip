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
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msorb.h"
#include "matcher_device.h"

namespace msorb {
void set_last_error(const std::string& s);
}
using msorb::set_last_error;
using msorb::kHistoLength;

namespace {

struct BowItem {  // one (pair, common node)
    int base1, base2;  // first row of the pair's set 1 / set 2 in the concatenated arrays
    int b1, n1l;       // the node's query list inside feat1
    int b2, n2l;       // the node's train list inside feat2
    int pair;
};

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

constexpr int kNoKey = 0x7fffffff;

__global__ __launch_bounds__(64) void bow_match_kernel(const BowItem* __restrict__ items, const uint4* __restrict__ desc1,
                                                       const uint4* __restrict__ desc2, const uint8_t* __restrict__ valid1,
                                                       const uint8_t* __restrict__ avail2, const int* __restrict__ feat1,
                                                       const int* __restrict__ feat2, int th_low, int inclusive,
                                                       float nnratio, int* __restrict__ match12) {
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
            const int idx2 = it.base2 + feat2[it.b2 + p];
            fr = avail2[idx2] != 0;
            if (c == 0) { t0 = desc2[(size_t)idx2 * 2]; t1 = desc2[(size_t)idx2 * 2 + 1]; }
        }
        const unsigned long long m = __ballot(fr);
        if (lane == 0) { free_bits[2 * c] = (unsigned)m; free_bits[2 * c + 1] = (unsigned)(m >> 32); }
    }
    __syncthreads();
    for (int k1 = 0; k1 < it.n1l; k1++) {
        const int idx1 = it.base1 + feat1[it.b1 + k1];
        if (!valid1[idx1]) continue;  // wave uniform
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
        const bool pass = key != kNoKey && (inclusive ? best <= th_low : best < th_low) &&
                          (float)best < nnratio * (float)second;  // :332-336, :959-961
        if (pass) {
            const int p = key & 0xFFFFF;
            if (lane == 0) {
                match12[idx1] = feat2[it.b2 + p];
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
            const int idx2 = it.base2 + feat2[it.b2 + p];
            const uint8_t fl = flags2[idx2];
            fr = (fl & 1) != 0;
            if (c == 0) { t0 = desc2[(size_t)idx2 * 2]; t1 = desc2[(size_t)idx2 * 2 + 1]; r0 = tr2[idx2]; s0 = (fl & 2) != 0; }
        }
        const unsigned long long m = __ballot(fr);
        if (lane == 0) { free_bits[2 * c] = (unsigned)m; free_bits[2 * c + 1] = (unsigned)(m >> 32); }
    }
    __syncthreads();
    for (int k1 = 0; k1 < it.n1l; k1++) {
        const int idx1 = it.base1 + feat1[it.b1 + k1];
        const uint8_t fl1 = flags1[idx1];
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
                    const int idx2 = it.base2 + feat2[it.b2 + p];
                    dist = hamming256(q0, q1, desc2[(size_t)idx2 * 2], desc2[(size_t)idx2 * 2 + 1]);
                    r = tr2[idx2];
                    stereo2 = (flags2[idx2] & 2) != 0;
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
                match12[idx1] = feat2[it.b2 + p];
                free_bits[p >> 5] &= ~(1u << (p & 31));
            }
            __syncthreads();
        }
    }
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
        items[ni++] = BowItem{(int)r1, (int)r2, (int)k1, l1, (int)k2, l2, pi};
        k1 += (size_t)l1;
        k2 += (size_t)l2;
    }
}

// The rotation histogram in the reference's visiting order (:340-353 / :1342-1354) and the ComputeThreeMaxima filter
// (:396-418 / :1360-1381).  m = the kernel's raw matches of this pair; angle(i1, i2) = the two keypoint angles.
template <class Angles>
int replay_histogram(const FeatVec& a, const std::vector<Common>& common, const int* m, int check_orientation, Angles angle,
                     int* match12) {
    std::vector<int> rotHist[kHistoLength];
    const float factor = 1.0f / kHistoLength;
    int nm = 0;
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
                if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(idx1);
                else { match12[idx1] = -1; nm--; }  // NaN / out-of-range angle: the reference asserts
            }
        }
    if (check_orientation) {
        int sizes[kHistoLength], ind[3];
        for (int i = 0; i < kHistoLength; i++) sizes[i] = (int)rotHist[i].size();
        msorb_three_maxima(sizes, kHistoLength, ind);
        for (int i = 0; i < kHistoLength; i++)
            if (i != ind[0] && i != ind[1] && i != ind[2])
                for (int idx1 : rotHist[i]) { match12[idx1] = -1; nm--; }
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

extern "C" int msorb_search_by_bow(int device, msorb_bow_pair* pairs, int n_pairs, int th_low, int inclusive, float nnratio,
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
                 in_bytes = o_a2 + up16(tot2), o_m = in_bytes, total = o_m + up16(tot1 * 4);
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
    e = hipMemcpyAsync(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, tot1 * 4, s);
    if (e == hipSuccess) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bow_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s,
                           (const BowItem*)(d + o_it), (const uint4*)(d + o_d1), (const uint4*)(d + o_d2),
                           (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2), (const int*)(d + o_f1),
                           (const int*)(d + o_f2), th_low, inclusive, nnratio, (int*)(d + o_m));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = hipMemcpyAsync(scr.h + o_m, d + o_m, tot1 * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_by_bow", e);
    const int* m_all = (const int*)(scr.h + o_m);
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
    e = hipMemcpyAsync(d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_m, 0xFF, tot1 * 4, s);
    if (e == hipSuccess) e = hipEventRecord(scr.e0, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(triangulation_match_kernel, dim3((unsigned)n_items), dim3(64), (size_t)max_chunks * 8, s,
                           (const BowItem*)(d + o_it), (const TriConst*)(d + o_c), (const uint4*)(d + o_d1),
                           (const uint4*)(d + o_d2), (const uint8_t*)(d + o_v1), (const uint8_t*)(d + o_a2),
                           (const float2*)(d + o_x1), (const float4*)(d + o_t2), (const int*)(d + o_f1), (const int*)(d + o_f2),
                           coarse, (int*)(d + o_m));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(scr.e1, s);
    if (e == hipSuccess) e = hipMemcpyAsync(scr.h + o_m, d + o_m, tot1 * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
    if (e != hipSuccess) return hip_fail(scr, "search_for_triangulation", e);
    const int* m_all = (const int*)(scr.h + o_m);
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
