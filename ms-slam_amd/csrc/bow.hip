// Bag-of-words transform of DBoW2 as ORB-SLAM3 uses it (Frame::ComputeBoW, src/Frame.cc:670-677 ->
// TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1123-1191 and :1218-1259).
//
// Device layout: the tree is stored in "children CSR" order — all children of one node are contiguous, in the
// order of the reference's `children` vector (ascending node id, TemplatedVocabulary.h:1392) — as three arrays
// indexed by position: 32-byte descriptor, ChildInfo{node id, first child position, child count, word id},
// double weight.  One level of the descent is then one coalesced read of k*32 B.
//
// Kernels:
//   bow_descend_kernel   one LPF-lane group (16 lanes for k<=16, else 32) per descriptor; per level every lane
//                        takes a child, XOR+popcount against the feature, (distance<<16 | child) min-reduced over
//                        the group = the reference's first-strict-minimum scan (:1237-1247).
//   bow_assemble_kernel  one workgroup per frame: builds BowVector and FeatureVector.  std::map insertion becomes
//                        a bitonic sort of (id<<32 | feature index) keys in LDS + run-length heads; the weight of
//                        a word is accumulated by repeated addition exactly like BowVector::addWeight
//                        (BowVector.cpp:36-48) and the L1/L2 norm is summed sequentially in ascending word order
//                        by one lane (BowVector.cpp:64-88), so every double is bit-identical to the reference's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/msorb.h"
#include "lds_limit.h"

namespace msorb {
// pinned host <-> device on a stream by the copy kernel (orb_kernels.hip; hipMemcpyAsync for unaligned pointers / MSORB_FRAME_COPIES=sdma)
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
}

namespace msorb {
void set_last_error(const std::string& s);
}
using msorb::set_last_error;

#define HIPCHK(expr)                                                           \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            return MSORB_E_HIP;                                                \
        }                                                                      \
    } while (0)

namespace {

constexpr int kMaxBowFeatures = 8192;  // per frame (LDS: 12 B per slot)
constexpr int kAsmThreads = 1024;

struct ChildInfo {
    int node, child_begin, child_count, word;
};

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), :1218-1259
template <int LPF>
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint8_t* __restrict__ desc, const int* __restrict__ counts,
                                                          int n_frames, int per_frame, int desc_stride, int out_stride,
                                                          const uint4* __restrict__ cdesc,
                                                          const ChildInfo* __restrict__ cinfo,
                                                          const double* __restrict__ cweight, int root_count,
                                                          int nid_level, int* __restrict__ feat_word,
                                                          int* __restrict__ feat_node, double* __restrict__ feat_weight) {
    const int sub = threadIdx.x % LPF;
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPF;
    const int frame = (int)(g / per_frame), i = (int)(g % per_frame);
    if (frame >= n_frames || i >= counts[frame]) return;
    const uint4* f = reinterpret_cast<const uint4*>(desc + ((size_t)frame * desc_stride + i) * 32);
    const uint4 f0 = f[0], f1 = f[1];
    int begin = 0, count = root_count, level = 0, nid = nid_level <= 0 ? 0 : -1, pos = 0;
    ChildInfo ci{0, 0, 0, 0};
    while (count > 0) {
        ++level;
        int key = 0x7fffffff;
        for (int c = sub; c < count; c += LPF) {
            const uint4* d = cdesc + (size_t)(begin + c) * 2;
            const int dist = hamming256(f0, f1, d[0], d[1]);
            key = min(key, (dist << 16) | c);
        }
#pragma unroll
        for (int off = LPF / 2; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, LPF));
        pos = begin + (key & 0xffff);
        ci = cinfo[pos];
        if (level == nid_level) nid = ci.node;
        begin = ci.child_begin;
        count = ci.child_count;
    }
    if (sub == 0) {
        const size_t o = (size_t)frame * out_stride + i;
        feat_word[o] = ci.word;
        feat_node[o] = nid;  // -1: level L-levelsup not reached (resolved by the assemble kernel)
        feat_weight[o] = cweight[pos];
    }
}

__device__ __forceinline__ int wave_incl_sum(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}
// exclusive block prefix sum; *total = block sum.  wtot: kAsmThreads/64 ints of LDS.
__device__ int block_excl_sum(int v, int* wtot, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_incl_sum(v);
    __syncthreads();
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kAsmThreads / 64; w++) {
        const int t = wtot[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}
// exclusive "last defined (>= 0) value before this thread", -1 if none
__device__ int block_excl_last(int v, int* wtot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off && inc < 0) inc = t;
    }
    __syncthreads();
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int prev = __shfl_up(inc, 1);
    if (lane == 0) prev = -1;
    if (prev < 0)
        for (int w = wave - 1; w >= 0; w--)
            if (wtot[w] >= 0) { prev = wtot[w]; break; }
    return prev;
}

// Ascending bitonic sort of P keys in LDS (P a power of two >= 128, all kAsmThreads threads call it, keys complete and a barrier
// passed before the call; ends with a barrier).  Comparators whose partners lie within 128 consecutive keys (j <= 64) run on
// registers — a wave holds a chunk as two keys per lane, partners by __shfl_xor — so only the j >= 128 steps of each merge are
// LDS passes with a workgroup barrier: 15 barriers for 2048 keys instead of 66.
__device__ __forceinline__ void bitonic_chunk_steps(unsigned long long& a, unsigned long long& b, int i0, int k, int j_first) {
    const int lane = threadIdx.x & 63;
    for (int j = j_first; j > 0; j >>= 1) {
        if (j == 64) {   // partner = the lane's other key (index i0 + 64)
            const bool up = (i0 & k) == 0;
            if ((a > b) == up) { const unsigned long long t = a; a = b; b = t; }
        } else {
            const unsigned long long pa = __shfl_xor(a, j, 64), pb = __shfl_xor(b, j, 64);
            const bool lower = (lane & j) == 0;
            const bool keep_min_a = lower == ((i0 & k) == 0), keep_min_b = lower == (((i0 + 64) & k) == 0);
            a = keep_min_a ? (a < pa ? a : pa) : (a > pa ? a : pa);
            b = keep_min_b ? (b < pb ? b : pb) : (b > pb ? b : pb);
        }
    }
}
__device__ void bitonic_sort(unsigned long long* keys, int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kAsmThreads / 64;
    const int n_chunks = P >> 7;
    // merges up to 128 keys: entirely inside a chunk
    for (int c = wave; c < n_chunks; c += kWaves) {
        const int i0 = (c << 7) + lane;
        unsigned long long a = keys[i0], b = keys[i0 + 64];
        for (int k = 2; k <= 128; k <<= 1) bitonic_chunk_steps(a, b, i0, k, k >> 1);
        keys[i0] = a; keys[i0 + 64] = b;
    }
    __syncthreads();
    for (int k = 256; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += kAsmThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int ixj = i | j;
                const unsigned long long a = keys[i], b = keys[ixj];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
            }
            __syncthreads();
        }
        for (int c = wave; c < n_chunks; c += kWaves) {
            const int i0 = (c << 7) + lane;
            unsigned long long a = keys[i0], b = keys[i0 + 64];
            bitonic_chunk_steps(a, b, i0, k, 64);
            keys[i0] = a; keys[i0 + 64] = b;
        }
        __syncthreads();
    }
}

constexpr int kPerMax = kMaxBowFeatures / kAsmThreads;  // slots per thread at the largest P

// transform(features, v, fv, levelsup) container assembly, :1141-1190
__global__ __launch_bounds__(kAsmThreads) void bow_assemble_kernel(
    const int* __restrict__ counts, int stride, int P, int tf_mode /* TF_IDF|TF: addWeight */, int must, int l2,
    const int* __restrict__ feat_word, int* __restrict__ feat_node, const double* __restrict__ feat_weight,
    int* __restrict__ bow_word, double* __restrict__ bow_value, int* __restrict__ n_bow, int* __restrict__ fv_node,
    int* __restrict__ fv_begin, int* __restrict__ fv_feat, int* __restrict__ n_fv) {
    extern __shared__ unsigned long long smem[];
    unsigned long long* keys = smem;
    int* headpos = reinterpret_cast<int*>(keys + P);
    __shared__ int wtot[kAsmThreads / 64];
    __shared__ double s_norm;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = counts[frame];
    feat_word += (size_t)frame * stride;
    feat_node += (size_t)frame * stride;
    feat_weight += (size_t)frame * stride;
    bow_word += (size_t)frame * stride;
    bow_value += (size_t)frame * stride;
    fv_node += (size_t)frame * stride;
    fv_feat += (size_t)frame * stride;
    fv_begin += (size_t)frame * (stride + 1);
    const int per = max(1, P / kAsmThreads);
    const int b = tid * per, e = min(b + per, P);

    // (a) node id of features that stopped above level L-levelsup: keeps the previous feature's (header note)
    {
        int last = -1;
        for (int i = b; i < e && i < n; i++) {
            const int v = feat_node[i];
            if (v >= 0) last = v;
        }
        int carry = block_excl_last(last, wtot);
        for (int i = b; i < e && i < n; i++) {
            const int v = feat_node[i];
            if (v >= 0) carry = v;
            else feat_node[i] = carry < 0 ? 0 : carry;
        }
    }
    // (b) BowVector: sort (word, feature) keys, run heads = map entries
    int nvalid = 0;
    for (int i = b; i < e; i++) {
        const bool valid = i < n && feat_weight[i] > 0;  // "not stopped", :1155
        keys[i] = valid ? ((unsigned long long)(unsigned)feat_word[i] << 32) | (unsigned)i : ~0ull;
        nvalid += valid;
    }
    int m;
    (void)block_excl_sum(nvalid, wtot, &m);
    __syncthreads();
    bitonic_sort(keys, P);
    int nh = 0;
    for (int i = b; i < e && i < m; i++) nh += (i == 0) || (keys[i] >> 32) != (keys[i - 1] >> 32);
    int nb;
    int rank = block_excl_sum(nh, wtot, &nb);
    {
        int r = rank;
        for (int i = b; i < e && i < m; i++)
            if ((i == 0) || (keys[i] >> 32) != (keys[i - 1] >> 32)) headpos[r++] = i;
        if (tid == 0) headpos[nb] = m;
    }
    __syncthreads();
    double vals[kPerMax];
    {
        int r = rank, c = 0;
        for (int i = b; i < e && i < m; i++)
            if ((i == 0) || (keys[i] >> 32) != (keys[i - 1] >> 32)) {
                const int cnt = headpos[r + 1] - i;
                const double w = feat_weight[(int)(unsigned)keys[i]];
                double v = w;                                     // insert(id, w)
                if (tf_mode) {
                    for (int k = 1; k < cnt; k++) v += w;         // vit->second += v, BowVector.cpp:42
                    if (!must) v /= (double)nb;                   // :1162-1168
                }
                bow_word[r] = (int)(keys[i] >> 32);
                if (c < kPerMax) vals[c] = v;
                c++;
                r++;
            }
    }
    __syncthreads();
    double* dv = reinterpret_cast<double*>(keys);
    for (int c = 0; c < nh && c < kPerMax; c++) dv[rank + c] = vals[c];
    __syncthreads();
    if (tid == 0) {
        double norm = 0.0;
        if (must) {  // BowVector::normalize, BowVector.cpp:64-88
            if (!l2) {
                for (int r = 0; r < nb; r++) norm += fabs(dv[r]);
            } else {
                for (int r = 0; r < nb; r++) norm = fma(dv[r], dv[r], norm);
                norm = sqrt(norm);
            }
        }
        s_norm = norm;
        n_bow[frame] = nb;
    }
    __syncthreads();
    {
        const double norm = s_norm;
        for (int r = tid; r < nb; r += kAsmThreads) bow_value[r] = (must && norm > 0.0) ? dv[r] / norm : dv[r];
    }
    __syncthreads();
    // (c) FeatureVector: sort (node, feature) keys; heads = map entries, members keep ascending feature order
    for (int i = b; i < e; i++) {
        const bool valid = i < n && feat_weight[i] > 0;
        keys[i] = valid ? ((unsigned long long)(unsigned)feat_node[i] << 32) | (unsigned)i : ~0ull;
    }
    __syncthreads();
    bitonic_sort(keys, P);
    nh = 0;
    for (int i = b; i < e && i < m; i++) nh += (i == 0) || (keys[i] >> 32) != (keys[i - 1] >> 32);
    int nf;
    rank = block_excl_sum(nh, wtot, &nf);
    for (int i = b; i < e && i < m; i++) {
        if ((i == 0) || (keys[i] >> 32) != (keys[i - 1] >> 32)) {
            fv_node[rank] = (int)(keys[i] >> 32);
            fv_begin[rank] = i;
            rank++;
        }
        fv_feat[i] = (int)(unsigned)keys[i];
    }
    if (tid == 0) {
        fv_begin[nf] = m;
        n_fv[frame] = nf;
    }
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct msorb_vocabulary {
    int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0, root_count = 0, max_children = 0;
    uint4* d_cdesc = nullptr;
    ChildInfo* d_cinfo = nullptr;
    double* d_cweight = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::recursive_mutex mu;  // scratch below is shared by all callers of this handle
    DevBuf<int> d_counts, d_feat_word, d_feat_node;
    DevBuf<double> d_feat_weight;
    // single-frame host path staging
    DevBuf<uint8_t> d_desc;
    DevBuf<int> d_out_i;
    DevBuf<double> d_out_d;
    uint8_t* h_pin = nullptr;   // pinned staging of the single-frame call: descriptors in, every output out, one synchronisation
    size_t h_pin_cap = 0;
    hipError_t ensure_pin(size_t n) {
        if (n <= h_pin_cap) return hipSuccess;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr; h_pin_cap = 0;
        const hipError_t e = hipHostMalloc((void**)&h_pin, n, hipHostMallocDefault);
        if (e == hipSuccess) h_pin_cap = n;
        return e;
    }
};

extern "C" {

int msorb_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int* parent,
                            const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights,
                            msorb_vocabulary** out) {
    if (!out) return MSORB_E_INVALID;
    *out = nullptr;
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 ||
        n_nodes < 1 || (n_nodes > 1 && (!parent || !is_leaf || !descriptors || !weights))) {
        set_last_error("vocabulary: bad header (k 0..20, L 1..10, scoring 0..5, weighting 0..3) or null arrays");
        return MSORB_E_INVALID;
    }
    for (int i = 1; i < n_nodes; i++)
        if (parent[i] < 0 || parent[i] >= n_nodes || parent[i] == i) {
            set_last_error("vocabulary: parent id out of range");
            return MSORB_E_INVALID;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    // children CSR by parent, children in ascending node id (= push_back order of the loader)
    std::vector<int> cnt(n_nodes, 0), begin(n_nodes + 1, 0), fill(n_nodes, 0), word(n_nodes, 0);
    for (int i = 1; i < n_nodes; i++) cnt[parent[i]]++;
    for (int i = 0; i < n_nodes; i++) begin[i + 1] = begin[i] + cnt[i];
    const int C = n_nodes - 1;
    std::vector<ChildInfo> info(std::max(C, 1));
    std::vector<uint8_t> cdesc((size_t)std::max(C, 1) * 32);
    std::vector<double> cw(std::max(C, 1));
    int n_words = 0, max_children = 0;
    for (int i = 1; i < n_nodes; i++)
        if (is_leaf[i]) word[i] = n_words++;
    for (int i = 0; i < n_nodes; i++) max_children = std::max(max_children, cnt[i]);
    if (max_children > 65535) {
        set_last_error("vocabulary: more than 65535 children under one node");
        return MSORB_E_CAPACITY;
    }
    for (int i = 1; i < n_nodes; i++) {
        const int pos = begin[parent[i]] + fill[parent[i]]++;
        info[pos] = ChildInfo{i, begin[i], cnt[i], word[i]};
        std::memcpy(&cdesc[(size_t)pos * 32], descriptors + (size_t)i * 32, 32);
        cw[pos] = weights[i];
    }
    HIPCHK(hipSetDevice(device));
    msorb_vocabulary* v = new msorb_vocabulary;
    v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->n_nodes = n_nodes; v->n_words = n_words; v->root_count = cnt[0]; v->max_children = max_children;
    hipError_t e = hipStreamCreateWithFlags(&v->s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&v->e0);
    if (e == hipSuccess) e = hipEventCreate(&v->e1);
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_cdesc, cdesc.size());
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_cinfo, info.size() * sizeof(ChildInfo));
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_cweight, cw.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(v->d_cdesc, cdesc.data(), cdesc.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_cinfo, info.data(), info.size() * sizeof(ChildInfo), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_cweight, cw.data(), cw.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_last_error(std::string("vocabulary upload: ") + hipGetErrorString(e));
        msorb_vocabulary_destroy(v);
        return MSORB_E_HIP;
    }
    *out = v;
    return MSORB_OK;
}

int msorb_vocabulary_load_text(int device, const char* path, msorb_vocabulary** out) {
    if (!out || !path) return MSORB_E_INVALID;
    *out = nullptr;
    FILE* f = std::fopen(path, "r");
    if (!f) {
        set_last_error(std::string("vocabulary: cannot open ") + path);
        return MSORB_E_INVALID;
    }
    std::vector<char> line(1 << 16);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    if (!std::fgets(line.data(), (int)line.size(), f) || std::sscanf(line.data(), "%d %d %d %d", &k, &L, &n1, &n2) != 4 ||
        k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        std::fclose(f);
        set_last_error("Vocabulary loading failure: This is not a correct text file!");  // TemplatedVocabulary.h:1361
        return MSORB_E_INVALID;
    }
    std::vector<int> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    bool bad = false;
    while (std::fgets(line.data(), (int)line.size(), f)) {
        char* p = line.data();
        char* q = nullptr;
        const long pid = std::strtol(p, &q, 10);
        if (q == p) continue;  // blank line
        p = q;
        const long is_leaf = std::strtol(p, &q, 10);
        if (q == p) { bad = true; break; }
        p = q;
        uint8_t d[32];
        for (int i = 0; i < 32 && !bad; i++) {
            const long b = std::strtol(p, &q, 10);
            if (q == p) bad = true;
            d[i] = (uint8_t)b;  // FORB::fromString, FORB.cpp:120-135
            p = q;
        }
        if (bad) break;
        const double w = std::strtod(p, &q);
        if (q == p) { bad = true; break; }
        parent.push_back((int)pid);
        leaf.push_back(is_leaf > 0);
        desc.insert(desc.end(), d, d + 32);
        weight.push_back(w);
    }
    std::fclose(f);
    if (bad) {
        set_last_error("vocabulary: malformed node line " + std::to_string(parent.size()));
        return MSORB_E_INVALID;
    }
    return msorb_vocabulary_create(device, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(),
                                   weight.data(), out);
}

void msorb_vocabulary_destroy(msorb_vocabulary* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->d_cdesc) (void)hipFree(v->d_cdesc);
    if (v->d_cinfo) (void)hipFree(v->d_cinfo);
    if (v->d_cweight) (void)hipFree(v->d_cweight);
    v->d_counts.release(); v->d_feat_word.release(); v->d_feat_node.release(); v->d_feat_weight.release();
    v->d_desc.release(); v->d_out_i.release(); v->d_out_d.release();
    if (v->h_pin) (void)hipHostFree(v->h_pin);
    if (v->e0) (void)hipEventDestroy(v->e0);
    if (v->e1) (void)hipEventDestroy(v->e1);
    if (v->s) (void)hipStreamDestroy(v->s);
    delete v;
}

int msorb_vocabulary_info(const msorb_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
    if (!v) return MSORB_E_INVALID;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return MSORB_OK;
}

static int bow_transform_enqueue(msorb_vocabulary* v, const uint8_t* d_descriptors, const int* h_counts, int n_frames,
                                 int desc_stride, int levelsup, int stride, int* d_bow_word, double* d_bow_value,
                                 int* d_n_bow, int* d_fv_node, int* d_fv_begin, int* d_fv_feat, int* d_n_fv,
                                 float* elapsed_ms, bool synchronise, int* d_fw = nullptr, int* d_fn = nullptr, double* d_fwt = nullptr);
int msorb_bow_transform_batch(msorb_vocabulary* v, const uint8_t* d_descriptors, const int* h_counts, int n_frames,
                              int desc_stride, int levelsup, int stride, int* d_bow_word, double* d_bow_value,
                              int* d_n_bow, int* d_fv_node, int* d_fv_begin, int* d_fv_feat, int* d_n_fv,
                              float* elapsed_ms) {
    return bow_transform_enqueue(v, d_descriptors, h_counts, n_frames, desc_stride, levelsup, stride, d_bow_word, d_bow_value, d_n_bow,
                                 d_fv_node, d_fv_begin, d_fv_feat, d_n_fv, elapsed_ms, true);
}
// synchronise = false: everything is enqueued on the vocabulary's stream, the caller appends its copies and waits once;
// d_fw / d_fn / d_fwt: where the per-feature word, node and weight go (default: the handle's scratch)
static int bow_transform_enqueue(msorb_vocabulary* v, const uint8_t* d_descriptors, const int* h_counts, int n_frames,
                                 int desc_stride, int levelsup, int stride, int* d_bow_word, double* d_bow_value,
                                 int* d_n_bow, int* d_fv_node, int* d_fv_begin, int* d_fv_feat, int* d_n_fv,
                                 float* elapsed_ms, bool synchronise, int* d_fw, int* d_fn, double* d_fwt) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (!v || n_frames < 0 || stride < 1 || stride > kMaxBowFeatures || desc_stride < 0 ||
        (n_frames > 0 && (!h_counts || !d_bow_word || !d_bow_value || !d_n_bow || !d_fv_node || !d_fv_begin ||
                          !d_fv_feat || !d_n_fv)))
        return MSORB_E_INVALID;
    int max_count = 0;
    for (int i = 0; i < n_frames; i++) {
        if (h_counts[i] < 0 || h_counts[i] > stride || h_counts[i] > desc_stride) {
            set_last_error("bow_transform: count exceeds stride (<= 8192 features per frame)");
            return MSORB_E_CAPACITY;
        }
        max_count = std::max(max_count, h_counts[i]);
    }
    if (max_count > 0 && (!d_descriptors || (reinterpret_cast<uintptr_t>(d_descriptors) & 15))) {
        set_last_error("bow_transform: descriptors must be 16-byte aligned device memory");
        return MSORB_E_INVALID;
    }
    if (n_frames == 0) return MSORB_OK;
    std::lock_guard<std::recursive_mutex> lock(v->mu);
    HIPCHK(hipSetDevice(v->device));
    hipStream_t s = v->s;
    if (v->n_words == 0 || max_count == 0) {  // empty(): containers stay empty (:1132-1135)
        HIPCHK(hipMemsetAsync(d_n_bow, 0, (size_t)n_frames * sizeof(int), s));
        HIPCHK(hipMemsetAsync(d_n_fv, 0, (size_t)n_frames * sizeof(int), s));
        HIPCHK(hipMemset2DAsync(d_fv_begin, (size_t)(stride + 1) * sizeof(int), 0, sizeof(int), n_frames, s));
        if (synchronise) HIPCHK(hipStreamSynchronize(s));
        return MSORB_OK;
    }
    const size_t total = (size_t)n_frames * stride;
    HIPCHK(v->d_counts.ensure(n_frames));
    if (!d_fw) {
        HIPCHK(v->d_feat_word.ensure(total));
        HIPCHK(v->d_feat_node.ensure(total));
        HIPCHK(v->d_feat_weight.ensure(total));
        d_fw = v->d_feat_word.p; d_fn = v->d_feat_node.p; d_fwt = v->d_feat_weight.p;
    }
    HIPCHK(hipMemcpyAsync(v->d_counts.p, h_counts, (size_t)n_frames * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(v->e0, s));
    const int lpf = v->max_children <= 16 ? 16 : 32;
    const long groups = (long)n_frames * max_count;
    const int threads = 256;
    const long blocks = (groups * lpf + threads - 1) / threads;
    const int nid_level = v->L - levelsup;
    if (lpf == 16)
        hipLaunchKernelGGL(bow_descend_kernel<16>, dim3((unsigned)blocks), dim3(threads), 0, s, d_descriptors,
                           v->d_counts.p, n_frames, max_count, desc_stride, stride, v->d_cdesc, v->d_cinfo, v->d_cweight,
                           v->root_count, nid_level, d_fw, d_fn, d_fwt);
    else
        hipLaunchKernelGGL(bow_descend_kernel<32>, dim3((unsigned)blocks), dim3(threads), 0, s, d_descriptors,
                           v->d_counts.p, n_frames, max_count, desc_stride, stride, v->d_cdesc, v->d_cinfo, v->d_cweight,
                           v->root_count, nid_level, d_fw, d_fn, d_fwt);
    int P = kAsmThreads;  // >= one slot per thread keeps the chunk arithmetic simple
    while (P < max_count) P <<= 1;
    const size_t lds = (size_t)P * 8 + (size_t)(P + 1) * 4;
    const int must = v->scoring != 5, l2 = v->scoring == 1, tf_mode = v->weighting == 0 || v->weighting == 1;
    if ((long long)lds > msorb::dynamic_lds_room(reinterpret_cast<const void*>(bow_assemble_kernel))) {   // raised once per device, never lowered
        set_last_error("bow_assemble: the device refuses " + std::to_string(lds) + " bytes of LDS per workgroup");
        return MSORB_E_HIP;
    }
    hipLaunchKernelGGL(bow_assemble_kernel, dim3(n_frames), dim3(kAsmThreads), lds, s, v->d_counts.p, stride, P, tf_mode,
                       must, l2, d_fw, d_fn, d_fwt, d_bow_word, d_bow_value, d_n_bow,
                       d_fv_node, d_fv_begin, d_fv_feat, d_n_fv);
    HIPCHK(hipEventRecord(v->e1, s));
    if (!synchronise) return MSORB_OK;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, v->e0, v->e1));
    return MSORB_OK;
}

int msorb_bow_transform(msorb_vocabulary* v, const uint8_t* descriptors, int n, int levelsup, int* bow_word,
                        double* bow_value, int* n_bow, int* fv_node, int* fv_begin, int* fv_feat, int* n_fv,
                        int* feat_word, int* feat_node, double* feat_weight) {
    if (!v || n < 0 || !n_bow || !n_fv || !fv_begin || (n > 0 && (!descriptors || !bow_word || !bow_value || !fv_node || !fv_feat)))
        return MSORB_E_INVALID;
    *n_bow = *n_fv = 0;
    fv_begin[0] = 0;
    if (n == 0) return MSORB_OK;
    if (n > kMaxBowFeatures) {
        set_last_error("bow_transform: more than 8192 features in one frame");
        return MSORB_E_CAPACITY;
    }
    const int stride = n;
    std::lock_guard<std::recursive_mutex> lock(v->mu);
    HIPCHK(hipSetDevice(v->device));
    // one device block and its pinned twin: [descriptors in][ints out 4 stride + 3][bow values][feat_word][feat_node][feat_weight]:
    // one copy in, the kernels, one copy out, one synchronisation
    const size_t n_int = (size_t)4 * stride + 3;
    const size_t o_int = ((size_t)n * 32 + 15) & ~(size_t)15, o_val = (o_int + n_int * sizeof(int) + 7) & ~(size_t)7,
                 o_fw = o_val + (size_t)stride * sizeof(double), o_fn = o_fw + (size_t)n * sizeof(int),
                 o_fwt = (o_fn + (size_t)n * sizeof(int) + 7) & ~(size_t)7, blk_bytes = o_fwt + (size_t)n * sizeof(double);
    HIPCHK(v->d_desc.ensure(blk_bytes));
    HIPCHK(v->ensure_pin(blk_bytes));
    uint8_t* const db = v->d_desc.p;
    uint8_t* const hp = v->h_pin;
    int* d_i = reinterpret_cast<int*>(db + o_int);
    double* d_d = reinterpret_cast<double*>(db + o_val);
    hipStream_t s = v->s;
    std::memcpy(hp, descriptors, (size_t)n * 32);
    HIPCHK(msorb::small_copy(db, hp, (size_t)n * 32, hipMemcpyHostToDevice, s));
    int* d_bow_word = d_i;
    int* d_fv_node = d_i + stride;
    int* d_fv_feat = d_i + 2 * stride;
    int* d_fv_begin = d_i + 3 * stride;       // stride + 1
    int* d_nb = d_i + 4 * stride + 1;
    int* d_nf = d_i + 4 * stride + 2;
    const int rc = bow_transform_enqueue(v, db, &n, 1, n, levelsup, stride, d_bow_word, d_d, d_nb, d_fv_node, d_fv_begin, d_fv_feat, d_nf,
                                         nullptr, false, reinterpret_cast<int*>(db + o_fw), reinterpret_cast<int*>(db + o_fn),
                                         reinterpret_cast<double*>(db + o_fwt));
    if (rc) return rc;
    const bool feats = v->n_words > 0;
    const bool want_feats = feats && (feat_word || feat_node || feat_weight);
    HIPCHK(msorb::small_copy(hp + o_int, db + o_int, (want_feats ? blk_bytes : o_fw) - o_int, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const int* hi = reinterpret_cast<const int*>(hp + o_int);
    const int nb = hi[4 * stride + 1], nf = hi[4 * stride + 2];
    *n_bow = nb;
    *n_fv = nf;
    std::memcpy(bow_word, hi, (size_t)nb * sizeof(int));
    std::memcpy(fv_node, hi + stride, (size_t)nf * sizeof(int));
    std::memcpy(fv_begin, hi + 3 * stride, (size_t)(nf + 1) * sizeof(int));
    std::memcpy(fv_feat, hi + 2 * stride, (size_t)hi[3 * stride + nf] * sizeof(int));
    if (nb) std::memcpy(bow_value, hp + o_val, (size_t)nb * sizeof(double));
    if (feats) {
        if (feat_word) std::memcpy(feat_word, hp + o_fw, (size_t)n * sizeof(int));
        if (feat_node) std::memcpy(feat_node, hp + o_fn, (size_t)n * sizeof(int));
        if (feat_weight) std::memcpy(feat_weight, hp + o_fwt, (size_t)n * sizeof(double));
    }
    return MSORB_OK;
}

}  // extern "C"
