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
    ~Scratch() { release(); }
};

bool check_feature_vector(int n, int nodes, const int* node, const int* begin, const int* feat, std::vector<uint8_t>& seen) {
    if (nodes < 0 || (nodes > 0 && (!node || !begin))) return false;
    if (nodes == 0) return true;
    if (begin[0] < 0) return false;
    for (int r = 0; r < nodes; r++) {
        if (begin[r + 1] < begin[r]) return false;
        if (r > 0 && node[r] <= node[r - 1]) return false;
    }
    if (begin[nodes] > begin[0] && !feat) return false;
    seen.assign((size_t)n, 0);
    for (int k = begin[0]; k < begin[nodes]; k++) {
        const int i = feat[k];
        if (i < 0 || i >= n || seen[i]) return false;
        seen[i] = 1;
    }
    return true;
}

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

extern "C" int msorb_search_by_bow(int device, msorb_bow_pair* pairs, int n_pairs, int th_low, int inclusive, float nnratio,
                                   int check_orientation, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSORB_E_INVALID;
    if (n_pairs == 0) return MSORB_OK;
    // ---- validation + merge walk (:239-243, :385-392): the nodes both vectors hold, ascending ----
    struct Common { int r1, r2; };
    std::vector<std::vector<Common>> common(n_pairs);
    std::vector<uint8_t> seen;
    size_t tot1 = 0, tot2 = 0, totf1 = 0, totf2 = 0, n_items = 0;
    int max_chunks = 1;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_pair& P = pairs[pi];
        P.nmatches = 0;
        if (P.n1 < 0 || P.n2 < 0 || (!P.match12 && P.n1 > 0) || (P.n1 > 0 && (!P.desc1 || !P.valid1)) || (P.n2 > 0 && !P.desc2) ||
            (check_orientation && ((P.n1 > 0 && !P.angle1) || (P.n2 > 0 && !P.angle2))) ||
            !check_feature_vector(P.n1, P.fv1_nodes, P.fv1_node, P.fv1_begin, P.fv1_feat, seen) ||
            !check_feature_vector(P.n2, P.fv2_nodes, P.fv2_node, P.fv2_begin, P.fv2_feat, seen)) {
            set_last_error("search_by_bow: pair " + std::to_string(pi) +
                           ": bad sizes / null arrays / feature vector not ascending, out of range or with a repeated feature");
            return MSORB_E_INVALID;
        }
        int a = 0, b = 0;
        while (a < P.fv1_nodes && b < P.fv2_nodes) {
            if (P.fv1_node[a] == P.fv2_node[b]) {
                const int l1 = P.fv1_begin[a + 1] - P.fv1_begin[a], l2 = P.fv2_begin[b + 1] - P.fv2_begin[b];
                if (l1 > 0 && l2 > 0) {
                    if (l2 >= (1 << 20)) { set_last_error("search_by_bow: node list too long"); return MSORB_E_INVALID; }
                    common[pi].push_back({a, b});
                    totf1 += (size_t)l1;
                    totf2 += (size_t)l2;
                    max_chunks = std::max(max_chunks, (l2 + 63) >> 6);
                }
                a++; b++;
            } else if (P.fv1_node[a] < P.fv2_node[b]) a++;
            else b++;
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
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    // ---- staging: [desc1 | desc2 | feat1 | feat2 | items | valid1 | avail2] in, [match12] out ----
    const size_t o_d1 = 0, o_d2 = o_d1 + tot1 * 32, o_f1 = o_d2 + tot2 * 32, o_f2 = o_f1 + up16(totf1 * 4),
                 o_it = o_f2 + up16(totf2 * 4), o_v1 = o_it + up16(n_items * sizeof(BowItem)), o_a2 = o_v1 + up16(tot1),
                 in_bytes = o_a2 + up16(tot2), o_m = in_bytes, total = o_m + up16(tot1 * 4);
    static thread_local Scratch scr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess && scr.device != device) {
        scr.release();
        scr.device = device;
        e = hipStreamCreateWithFlags(&scr.s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&scr.e0);
        if (e == hipSuccess) e = hipEventCreate(&scr.e1);
    }
    if (e == hipSuccess && total > scr.cap) {
        if (scr.d) (void)hipFree(scr.d);
        if (scr.h) (void)hipHostFree(scr.h);
        scr.d = scr.h = nullptr; scr.cap = 0;
        e = hipMalloc((void**)&scr.d, total + total / 2);
        if (e == hipSuccess) e = hipHostMalloc((void**)&scr.h, total + total / 2, hipHostMallocDefault);
        if (e == hipSuccess) scr.cap = total + total / 2;
    }
    if (e != hipSuccess) {
        set_last_error(std::string("search_by_bow: ") + hipGetErrorString(e));
        scr.release();
        return MSORB_E_HIP;
    }
    {
        char* h = scr.h;
        BowItem* items = (BowItem*)(h + o_it);
        int *f1 = (int*)(h + o_f1), *f2 = (int*)(h + o_f2);
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
            for (const Common& c : common[pi]) {
                const int l1 = P.fv1_begin[c.r1 + 1] - P.fv1_begin[c.r1], l2 = P.fv2_begin[c.r2 + 1] - P.fv2_begin[c.r2];
                std::memcpy(f1 + k1, P.fv1_feat + P.fv1_begin[c.r1], (size_t)l1 * 4);
                std::memcpy(f2 + k2, P.fv2_feat + P.fv2_begin[c.r2], (size_t)l2 * 4);
                items[ni++] = BowItem{(int)r1, (int)r2, (int)k1, l1, (int)k2, l2};
                k1 += (size_t)l1;
                k2 += (size_t)l2;
            }
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
    if (e != hipSuccess) {
        set_last_error(std::string("search_by_bow: ") + hipGetErrorString(e));
        scr.release();
        return MSORB_E_HIP;
    }
    // ---- host replay of the rotation histogram in the reference's visiting order (:340-353, :396-418) ----
    const int* m_all = (const int*)(scr.h + o_m);
    size_t r1 = 0;
    const float factor = 1.0f / kHistoLength;
    for (int pi = 0; pi < n_pairs; pi++) {
        msorb_bow_pair& P = pairs[pi];
        const int* m = m_all + r1;
        r1 += (size_t)P.n1;
        std::vector<int> rotHist[kHistoLength];
        int nm = 0;
        for (const Common& c : common[pi])
            for (int k = P.fv1_begin[c.r1]; k < P.fv1_begin[c.r1 + 1]; k++) {
                const int idx1 = P.fv1_feat[k], idx2 = m[idx1];
                if (idx2 < 0) continue;
                P.match12[idx1] = idx2;
                nm++;
                if (check_orientation) {
                    float rot = P.angle1[idx1] - P.angle2[idx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == kHistoLength) bin = 0;
                    if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(idx1);
                    else { P.match12[idx1] = -1; nm--; }  // NaN / out-of-range angle: the reference asserts
                }
            }
        if (check_orientation) {
            int sizes[kHistoLength], ind[3];
            for (int i = 0; i < kHistoLength; i++) sizes[i] = (int)rotHist[i].size();
            msorb_three_maxima(sizes, kHistoLength, ind);
            for (int i = 0; i < kHistoLength; i++)
                if (i != ind[0] && i != ind[1] && i != ind[2])
                    for (int idx1 : rotHist[i]) { P.match12[idx1] = -1; nm--; }
        }
        if (P.match21)
            for (int i = 0; i < P.n1; i++)
                if (P.match12[i] >= 0) P.match21[P.match12[i]] = i;
        P.nmatches = nm;
    }
    return MSORB_OK;
}
