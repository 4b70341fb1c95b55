// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:351-429) batched over map points.
//
// For a point with N observed descriptors the reference fills the N x N Hamming matrix (:399-410), sorts every
// row and takes vDists[0.5*(N-1)] as the row's median (:415-419), and keeps the first row with the strictly
// smallest median (:421-425).  Here: one W-lane segment of a wavefront per point (W = 8/16/32/64 >= N), lane j
// holds descriptor j in registers, row i's descriptor is a segment-uniform load, and the k-th smallest of the
// row is found by a 9-step radix select over __ballot masks (distances are 0..256) — no sort, no N x N storage.
// Points with more than 64 descriptors take a one-wave-per-point kernel with a 257-bin LDS histogram per row.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msorb.h"

namespace msorb {
void set_last_error(const std::string& s);
// pinned host <-> device on a stream by the copy kernel (orb_kernels.hip)
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
}
using msorb::set_last_error;

#define HIPCHK(expr)                                                           \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            cleanup();                                                         \
            return MSORB_E_HIP;                                                \
        }                                                                      \
    } while (0)

namespace {

struct DistinctScratch {   // grow-only, per calling thread; released at thread exit (a dead runtime is tolerated)
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
            e = hipMalloc((void**)&d, total + total / 2 + 64);
            if (e == hipSuccess) e = hipHostMalloc((void**)&h, total + total / 2 + 64, hipHostMallocDefault);
            if (e == hipSuccess) cap = total + total / 2;
        }
        return e;
    }
    ~DistinctScratch() { release(); }
};

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

template <int W>
__global__ __launch_bounds__(256) void distinct_small_kernel(const uint4* __restrict__ desc, const int* __restrict__ obs_begin,
                                                             const int* __restrict__ points, int n_list,
                                                             int* __restrict__ best_idx, int* __restrict__ best_median) {
    const int g = (blockIdx.x * 256 + threadIdx.x) / W;
    if (g >= n_list) return;
    const int lane = threadIdx.x & 63, j = lane % W, seg = lane - j;
    const int p = points[g];
    const int b = obs_begin[p], N = obs_begin[p + 1] - b;
    const unsigned long long seg_mask = (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull)) << seg;
    uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0;
    if (j < N) {
        m0 = desc[(size_t)(b + j) * 2];
        m1 = desc[(size_t)(b + j) * 2 + 1];
    }
    int best = INT_MAX, besti = 0;
    for (int i = 0; i < N; i++) {
        const uint4 r0 = desc[(size_t)(b + i) * 2], r1 = desc[(size_t)(b + i) * 2 + 1];
        const int d = hamming256(m0, m1, r0, r1);  // Distances[i][j]; the diagonal is 0 by construction
        unsigned long long cand = __ballot(j < N) & seg_mask;
        int k = (N - 1) >> 1;  // vDists[0.5*(N-1)], :419
        int med = 0;
#pragma unroll
        for (int bit = 8; bit >= 0; bit--) {
            const unsigned long long z = __ballot(((d >> bit) & 1) == 0) & cand;
            const int c = __popcll(z);
            if (k < c) cand = z;
            else { cand &= ~z; k -= c; med |= 1 << bit; }
        }
        if (med < best) { best = med; besti = i; }
    }
    if (j == 0) {
        best_idx[p] = besti;
        if (best_median) best_median[p] = best;
    }
}

__global__ __launch_bounds__(64) void distinct_big_kernel(const uint4* __restrict__ desc, const int* __restrict__ obs_begin,
                                                          const int* __restrict__ points, int* __restrict__ best_idx,
                                                          int* __restrict__ best_median) {
    __shared__ int hist[320];
    const int lane = threadIdx.x;
    const int p = points[blockIdx.x];
    const int b = obs_begin[p], N = obs_begin[p + 1] - b;
    int best = INT_MAX, besti = 0;
    for (int i = 0; i < N; i++) {
        for (int t = lane; t < 320; t += 64) hist[t] = 0;
        __syncthreads();
        const uint4 r0 = desc[(size_t)(b + i) * 2], r1 = desc[(size_t)(b + i) * 2 + 1];
        for (int j = lane; j < N; j += 64) {
            const uint4 c0 = desc[(size_t)(b + j) * 2], c1 = desc[(size_t)(b + j) * 2 + 1];
            atomicAdd(&hist[hamming256(c0, c1, r0, r1)], 1);
        }
        __syncthreads();
        int h[5], mine = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) { h[t] = hist[lane * 5 + t]; mine += h[t]; }
        int inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        const int k = (N - 1) >> 1;
        int excl = inc - mine;
        const bool owner = excl <= k && k < inc;
        int med = 0;
        if (owner) {
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (excl <= k && k < excl + h[t]) med = lane * 5 + t;
                excl += h[t];
            }
        }
        const unsigned long long ob = __ballot(owner);
        med = __shfl(med, __ffsll((long long)ob) - 1);
        if (med < best) { best = med; besti = i; }
        __syncthreads();
    }
    if (lane == 0) {
        best_idx[p] = besti;
        if (best_median) best_median[p] = best;
    }
}

}  // namespace

extern "C" int msorb_distinctive_descriptors(int device, const uint8_t* descriptors, const int* obs_begin, int n_points,
                                             int* best_idx, int* best_median, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (n_points < 0 || (n_points > 0 && (!obs_begin || !best_idx))) return MSORB_E_INVALID;
    if (n_points == 0) return MSORB_OK;
    if (obs_begin[0] < 0) return MSORB_E_INVALID;
    for (int p = 0; p < n_points; p++)
        if (obs_begin[p + 1] < obs_begin[p]) {
            set_last_error("distinctive_descriptors: obs_begin must be non-decreasing");
            return MSORB_E_INVALID;
        }
    const int total = obs_begin[n_points];
    if (total > 0 && !descriptors) return MSORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    // classes by segment width; points without descriptors are answered here (:393-394: early return)
    std::vector<int> lists[5];
    for (int p = 0; p < n_points; p++) {
        const int N = obs_begin[p + 1] - obs_begin[p];
        if (N == 0) {
            best_idx[p] = -1;
            if (best_median) best_median[p] = INT_MAX;
            continue;
        }
        lists[N <= 8 ? 0 : N <= 16 ? 1 : N <= 32 ? 2 : N <= 64 ? 3 : 4].push_back(p);
    }
    std::vector<int> all;
    int off[6] = {0};
    for (int c = 0; c < 5; c++) {
        off[c] = (int)all.size();
        all.insert(all.end(), lists[c].begin(), lists[c].end());
    }
    off[5] = (int)all.size();
    if (all.empty()) return MSORB_OK;

    // per-thread grow-only scratch: one pinned block, one device block, a stream and two events — a call is one upload
    // ([descriptors | obs_begin | point lists]), the kernels, one read-back ([best | median]).  (Rounds 1-5 created the stream and the
    // events, hipMalloc'ed five arrays and freed them again in EVERY call, and copied from pageable memory: 0.5 ms for 2 000 points
    // where the kernels take 20 us — and every hipFree synchronises the device under the other SLAM threads.)
    static thread_local DistinctScratch scr;
    auto cleanup = [&] {};
    const size_t o_beg = ((size_t)total * 32 + 15) & ~(size_t)15, o_pts = o_beg + (((size_t)(n_points + 1) * 4 + 15) & ~(size_t)15),
                 in_bytes = o_pts + ((all.size() * 4 + 15) & ~(size_t)15), o_best = in_bytes, o_med = o_best + (((size_t)n_points * 4 + 15) & ~(size_t)15),
                 bytes = o_med + (((size_t)n_points * 4 + 15) & ~(size_t)15);
    {
        const hipError_t e = scr.acquire(device, bytes);
        if (e != hipSuccess) { set_last_error(std::string("distinctive_descriptors: ") + hipGetErrorString(e)); scr.release(); return MSORB_E_HIP; }
    }
    hipStream_t s = scr.s;
    if (total) std::memcpy(scr.h, descriptors, (size_t)total * 32);
    std::memcpy(scr.h + o_beg, obs_begin, (size_t)(n_points + 1) * sizeof(int));
    std::memcpy(scr.h + o_pts, all.data(), all.size() * sizeof(int));
    HIPCHK(msorb::small_copy(scr.d, scr.h, in_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(scr.e0, s));
    const uint4* dd = reinterpret_cast<const uint4*>(scr.d);
    const int* d_begin = reinterpret_cast<const int*>(scr.d + o_beg);
    const int* d_points = reinterpret_cast<const int*>(scr.d + o_pts);
    int* d_best = reinterpret_cast<int*>(scr.d + o_best);
    int* d_med = reinterpret_cast<int*>(scr.d + o_med);
    auto launch_small = [&](auto kernel, int W, int c) {
        const int n = off[c + 1] - off[c];
        if (n == 0) return;
        const int per_block = 256 / W;
        hipLaunchKernelGGL(kernel, dim3((n + per_block - 1) / per_block), dim3(256), 0, s, dd, d_begin, d_points + off[c], n,
                           d_best, d_med);
    };
    launch_small(distinct_small_kernel<8>, 8, 0);
    launch_small(distinct_small_kernel<16>, 16, 1);
    launch_small(distinct_small_kernel<32>, 32, 2);
    launch_small(distinct_small_kernel<64>, 64, 3);
    if (off[5] > off[4])
        hipLaunchKernelGGL(distinct_big_kernel, dim3(off[5] - off[4]), dim3(64), 0, s, dd, d_begin, d_points + off[4], d_best,
                           d_med);
    HIPCHK(hipEventRecord(scr.e1, s));
    HIPCHK(msorb::small_copy(scr.h + o_best, scr.d + o_best, bytes - o_best, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1));
    const int* hb = reinterpret_cast<const int*>(scr.h + o_best);
    const int* hm = reinterpret_cast<const int*>(scr.h + o_med);
    for (int p : all) {
        best_idx[p] = hb[p];
        if (best_median) best_median[p] = hm[p];
    }
    return MSORB_OK;
}
