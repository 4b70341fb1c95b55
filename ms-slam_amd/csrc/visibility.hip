// Keyframe x map-point visibility / constraint matrix of MapSparsification::Sparsifying
// (MapSparsification.cc:58-151), assembled on the device as CSR (rows in the order the reference adds
// constraints, columns = map points in first-encounter order).  Sort-free: first-encounter order comes from
// an atomicMin + flag scan, the outside-keyframe rows from per-row bitmaps whose prefix popcount yields the
// ascending column order the reference's accumulation produces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msorb.h"

namespace msorb {
void set_last_error(const std::string& s);

// exclusive scan of one int per thread over a workgroup of up to 1024 threads (wave scans + the wave totals in LDS)
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot /* [16] shared */, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; i++) { const int x = wave_tot[i]; if (i < w) base += x; tot += x; }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}

// Column order = first-encounter order of the points over the slots (MapSparsification.cc:78-99): slot s opens a column iff it
// is the first slot of its point.  Three small launches instead of a one-block chunked scan (148 us for 60 000 slots):
// per-block counts of the "first" flags, one block scans the counts, every block ranks its own slots.
__global__ __launch_bounds__(256) void vis_flag_count_kernel(const int* __restrict__ slot_point, const int* __restrict__ first,
                                                             int n_slots, int* __restrict__ block_count) {
    __shared__ int wt[16];
    const int s = blockIdx.x * 256 + threadIdx.x;
    int f = 0;
    if (s < n_slots) { const int p = slot_point[s]; f = (p >= 0 && first[p] == s); }
    int tot;
    (void)block_exclusive_scan(f, wt, &tot);
    if (threadIdx.x == 0) block_count[blockIdx.x] = tot;
}
// column of every point; column tables; running max of observations
__global__ __launch_bounds__(256) void vis_columns_kernel(const int* __restrict__ slot_point, const int* __restrict__ first,
                                                          const int* __restrict__ block_base, int n_slots,
                                                          const int* __restrict__ point_nobs, int* __restrict__ col_of_point,
                                                          int* __restrict__ col_point, int* __restrict__ n_max_obs, int* __restrict__ n_cols) {
    __shared__ int wt[16];
    __shared__ int smax, sbase;
    if (threadIdx.x == 0) { smax = 0; sbase = 0; }
    __syncthreads();
    // columns opened by the blocks in front of this one: block_base holds the per-block COUNTS (vis_flag_count_kernel); a few
    // hundred values, summed here instead of scanned by a launch of their own
    {
        int part = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += block_base[i];
        if (part) atomicAdd(&sbase, part);
    }
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int p = s < n_slots ? slot_point[s] : -1;
    const int f = (p >= 0 && first[p] == s);
    int tot;
    const int ex = block_exclusive_scan(f, wt, &tot);   // (the scan synchronises: smax and sbase are final)
    const int rank = sbase + ex;
    if (p >= 0) atomicMax(&smax, point_nobs[p]);
    if (f) { col_of_point[p] = rank; col_point[rank] = p; }
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) *n_cols = sbase + tot;
    __syncthreads();
    if (threadIdx.x == 0 && smax > 0) atomicMax(n_max_obs, smax);
}

constexpr int kKfThreads = 1024;   // one workgroup per window keyframe: its ~2000 slots are two per thread (latency, not work, is the cost)
// Launch fusions (round 3, second half: 15 launches — 12 kernels and 3 memsets — were 90 us of a 196 us call; every dependent
// launch costs 5-6 us here whatever it does).  The three memsets are one kernel; steps without a dependency between them share a
// launch (block ranges); the one-block scans over a few hundred values are redone by every block that needs them.
__global__ __launch_bounds__(256) void vis_init_kernel(int* __restrict__ first, int n_points, int* __restrict__ kf_count, int n_kf,
                                                       unsigned* __restrict__ bitmap, size_t bitmap_words) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < (size_t)n_points; i += stride) first[i] = 0x7f7f7f7f;
    for (size_t i = i0; i < (size_t)n_kf; i += stride) kf_count[i] = 0;
    for (size_t i = i0; i < bitmap_words; i += stride) bitmap[i] = 0u;
}
// blocks [0, nb_first): vis_first over 1024 slots each; blocks [nb_first, nb_first + n_window_kf): vis_kf_count
__global__ __launch_bounds__(kKfThreads) void vis_first_kfcount_kernel(int nb_first, const int* __restrict__ slot_point, int n_slots,
                                                                       int* __restrict__ first, const int* __restrict__ kf_slot_begin,
                                                                       const int* __restrict__ slot_cell, int* __restrict__ kf_valid,
                                                                       int* __restrict__ kf_cells) {
    __shared__ int sv, sc;
    if ((int)blockIdx.x < nb_first) {
        const int s = blockIdx.x * kKfThreads + threadIdx.x;
        if (s >= n_slots) return;
        const int p = slot_point[s];
        if (p >= 0) atomicMin(&first[p], s);
        return;
    }
    const int k = blockIdx.x - nb_first;
    if (threadIdx.x == 0) { sv = 0; sc = 0; }
    __syncthreads();
    const int b = kf_slot_begin[k], e = kf_slot_begin[k + 1];
    int v = 0, c = 0;
    for (int s = b + threadIdx.x; s < e; s += kKfThreads) {
        if (slot_point[s] < 0) continue;
        v++;
        bool first_valid = true;
        for (int t = s - 1; t >= b && slot_cell[t] == slot_cell[s]; t--)
            if (slot_point[t] >= 0) { first_valid = false; break; }
        c += first_valid;
    }
    atomicAdd(&sv, v);
    atomicAdd(&sc, c);
    __syncthreads();
    if (threadIdx.x == 0) { kf_valid[k] = sv; kf_cells[k] = sc; }
}

// One block per window keyframe: emit its cell rows then its keyframe row.  Serial-in-order within the
// block's thread 0 would be O(slots); instead each thread ranks its slots with block-wide prefix sums.
__device__ __forceinline__ void vis_kf_rows_body(int k, int row0, int nnz0, int* wt /* shared [16] */,
                                                 const int* __restrict__ kf_slot_begin, const int* __restrict__ slot_point,
                                                 const int* __restrict__ slot_cell, const int* __restrict__ col_of_point,
                                                 const int* __restrict__ kf_valid, int N, int* __restrict__ row_begin,
                                                 int* __restrict__ row_kind, int* __restrict__ row_owner, float* __restrict__ row_rhs,
                                                 int* __restrict__ col_idx) {
    const int tid = threadIdx.x;
    const int b = kf_slot_begin[k], e = kf_slot_begin[k + 1];
    const int n = e - b;
    const int per = (n + kKfThreads - 1) / kKfThreads;
    const int tb = min(b + tid * per, e), te = min(tb + per, e);
    int v = 0, c = 0;
    for (int s = tb; s < te; s++) {
        if (slot_point[s] < 0) continue;
        v++;
        bool first_valid = true;
        for (int t = s - 1; t >= b && slot_cell[t] == slot_cell[s]; t--)
            if (slot_point[t] >= 0) { first_valid = false; break; }
        c += first_valid;
    }
    int total_c;
    const int ev = block_exclusive_scan(v, wt, nullptr);
    const int ec = block_exclusive_scan(c, wt, &total_c);
    int part_v_tid = ev, part_c_tid = ec;
    int rv = part_v_tid, rc = part_c_tid;  // valid slots / valid cells before this thread's chunk
    const int V = kf_valid[k];
    for (int s = tb; s < te; s++) {
        const int p = slot_point[s];
        if (p < 0) continue;
        bool first_valid = true;
        for (int t = s - 1; t >= b && slot_cell[t] == slot_cell[s]; t--)
            if (slot_point[t] >= 0) { first_valid = false; break; }
        const int col = col_of_point[p];
        col_idx[nnz0 + rv] = col;          // cell rows: the keyframe's valid slots in walk order
        col_idx[nnz0 + V + rv] = col;      // keyframe row: the same terms again
        if (first_valid) {
            const int r = row0 + rc;
            row_begin[r] = nnz0 + rv;
            row_kind[r] = 0; row_owner[r] = slot_cell[s]; row_rhs[r] = 1.0f;
            rc++;
        }
        rv++;
    }
    if (tid == 0) {  // the keyframe's own row follows its cell rows
        const int r = row0 + total_c;
        row_begin[r] = nnz0 + V;
        row_kind[r] = 1; row_owner[r] = k; row_rhs[r] = (float)N;
    }
}

// outside-keyframe rows: count, bitmap, emit
// (the column count lives on the device: scal[0]; the grid covers its host-side upper bound).  A few hundred keyframe counters
// take tens of thousands of increments: they are accumulated per workgroup in LDS (kLdsHist counters) and flushed once.
constexpr int kLdsHist = 8192;
__device__ __forceinline__ void vis_extra_count_body(int cblock, int* hist /* dynamic LDS, n_kf ints if n_kf <= kLdsHist */,
                                                     const int* __restrict__ col_point, const int* __restrict__ scal,
                                                     const int* __restrict__ obs_begin, const int* __restrict__ obs_kf,
                                                     const uint8_t* __restrict__ kf_in_window, int n_kf, int* __restrict__ kf_count) {
    const bool lds = n_kf <= kLdsHist;
    if (lds) {
        for (int i = threadIdx.x; i < n_kf; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    const int c = cblock * (int)blockDim.x + threadIdx.x;
    if (c < scal[0]) {
        const int p = col_point[c];
        for (int o = obs_begin[p]; o < obs_begin[p + 1]; o++) {
            const int kf = obs_kf[o];
            if (!kf_in_window[kf]) atomicAdd(lds ? &hist[kf] : &kf_count[kf], 1);
        }
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_kf; i += blockDim.x)
            if (hist[i]) atomicAdd(&kf_count[i], hist[i]);
    }
}
// blocks [0, n_window_kf): the rows of a window keyframe (rows / non-zeros in front of it summed from the counts of the keyframes
// before it: a window holds a few dozen; the last of them publishes the totals in scal[4], scal[5]);
// blocks [n_window_kf, ...): vis_extra_count over 1024 columns each.  Both only depend on the column step.
__global__ __launch_bounds__(kKfThreads) void vis_kf_rows_extra_count_kernel(
    int n_window_kf, const int* __restrict__ kf_slot_begin, const int* __restrict__ slot_point, const int* __restrict__ slot_cell,
    const int* __restrict__ col_of_point, const int* __restrict__ kf_valid, const int* __restrict__ kf_cells, int N,
    int* __restrict__ row_begin, int* __restrict__ row_kind, int* __restrict__ row_owner, float* __restrict__ row_rhs,
    int* __restrict__ col_idx, int* __restrict__ scal, const int* __restrict__ col_point, const int* __restrict__ obs_begin,
    const int* __restrict__ obs_kf, const uint8_t* __restrict__ kf_in_window, int n_kf, int* __restrict__ kf_count) {
    extern __shared__ int hist[];
    __shared__ int wt[16];
    __shared__ int s_row0, s_nnz0;
    if ((int)blockIdx.x >= n_window_kf) {
        vis_extra_count_body((int)blockIdx.x - n_window_kf, hist, col_point, scal, obs_begin, obs_kf, kf_in_window, n_kf, kf_count);
        return;
    }
    const int k = blockIdx.x;
    if (threadIdx.x == 0) {
        int rows = 0, nnz = 0;
        for (int i = 0; i < k; i++) { rows += kf_cells[i] + 1; nnz += 2 * kf_valid[i]; }
        s_row0 = rows; s_nnz0 = nnz;
        if (k == n_window_kf - 1) { scal[4] = rows + kf_cells[k] + 1; scal[5] = nnz + 2 * kf_valid[k]; }
    }
    __syncthreads();
    vis_kf_rows_body(k, s_row0, s_nnz0, wt, kf_slot_begin, slot_point, slot_cell, col_of_point, kf_valid, N, row_begin, row_kind, row_owner,
                     row_rhs, col_idx);
}
__global__ __launch_bounds__(1024) void vis_extra_scan_kernel(const int* __restrict__ kf_count, int n_kf,
                                                              int* __restrict__ kf_row /* rank among count>0 */,
                                                              int* __restrict__ kf_off /* nnz offset */,
                                                              int* __restrict__ totals /* [n_rows_c, nnz_c] */) {
    __shared__ int wt[16];
    __shared__ int carry_r, carry_n;
    if (threadIdx.x == 0) { carry_r = 0; carry_n = 0; }
    __syncthreads();
    for (int base = 0; base < n_kf; base += 1024) {
        const int i = base + threadIdx.x;
        const int cnt = i < n_kf ? kf_count[i] : 0;
        int tr, tn;
        const int er = block_exclusive_scan(cnt > 0 ? 1 : 0, wt, &tr);
        const int en = block_exclusive_scan(cnt, wt, &tn);
        const int cr = carry_r, cn = carry_n;
        if (i < n_kf) { kf_row[i] = cnt > 0 ? cr + er : -1; kf_off[i] = cn + en; }
        __syncthreads();
        if (threadIdx.x == 0) { carry_r = cr + tr; carry_n = cn + tn; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = carry_r; totals[1] = carry_n; }
}
// The result as ONE block: the eight scalars, then the six arrays side by side at their exact sizes (device scalars) — the
// host reads the block back with one copy sized by its upper bound and finds the arrays from the header.
__global__ __launch_bounds__(256) void vis_pack_kernel(const int* __restrict__ scal, const int* __restrict__ row_begin,
                                                       const int* __restrict__ row_kind, const int* __restrict__ row_owner,
                                                       const int* __restrict__ row_rhs, const int* __restrict__ col_point,
                                                       const int* __restrict__ col_idx, int* __restrict__ out) {
    const int ncols = scal[0], R = scal[4] + scal[2], nnz = scal[5] + scal[3];
    const int total = 4 * R + ncols + nnz;
    if (blockIdx.x == 0 && threadIdx.x < 8) out[threadIdx.x] = scal[threadIdx.x];
    out += 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int v;
        if (i < R) v = row_begin[i];
        else if (i < 2 * R) v = row_kind[i - R];
        else if (i < 3 * R) v = row_owner[i - 2 * R];
        else if (i < 4 * R) v = row_rhs[i - 3 * R];
        else if (i < 4 * R + ncols) v = col_point[i - 4 * R];
        else v = col_idx[i - 4 * R - ncols];
        out[i] = v;
    }
}
__global__ void vis_extra_bits_kernel(const int* __restrict__ col_point, const int* __restrict__ scal, const int* __restrict__ obs_begin,
                                      const int* __restrict__ obs_kf, const uint8_t* __restrict__ kf_in_window,
                                      const int* __restrict__ kf_row, int words, unsigned* __restrict__ bitmap) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= scal[0]) return;
    const int p = col_point[c];
    for (int o = obs_begin[p]; o < obs_begin[p + 1]; o++) {
        const int kf = obs_kf[o];
        if (!kf_in_window[kf]) atomicOr(&bitmap[(size_t)kf_row[kf] * words + (c >> 5)], 1u << (c & 31));
    }
}
// one block per outside keyframe with count > 0
__global__ __launch_bounds__(256) void vis_extra_emit_kernel(int n_kf, const int* __restrict__ kf_count,
                                                             const int* __restrict__ kf_row, const int* __restrict__ kf_off,
                                                             const int* __restrict__ kf_num_mps, int N, int words,
                                                             const unsigned* __restrict__ bitmap, const int* __restrict__ scal,
                                                             int* __restrict__ row_begin, int* __restrict__ row_kind,
                                                             int* __restrict__ row_owner, float* __restrict__ row_rhs,
                                                             int* __restrict__ col_idx) {
    __shared__ int part[256];
    const int kf = blockIdx.x, tid = threadIdx.x;
    if (kf >= n_kf || kf_count[kf] == 0) return;
    const int row_base = scal[4], nnz_base = scal[5];  // rows / non-zeros of the window keyframes (vis_kf_base_kernel)
    const unsigned* bm = bitmap + (size_t)kf_row[kf] * words;
    const int per = (words + 255) / 256;
    const int b = tid * per, e = min(b + per, words);
    int c = 0;
    for (int w = b; w < e; w++) c += __popc(bm[w]);
    part[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; }
    }
    __syncthreads();
    int pos = nnz_base + kf_off[kf] + part[tid];
    for (int w = b; w < e; w++) {
        unsigned m = bm[w];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            col_idx[pos++] = w * 32 + bit;
        }
    }
    if (tid == 0) {
        const int r = row_base + kf_row[kf];
        row_begin[r] = nnz_base + kf_off[kf];
        row_kind[r] = 2; row_owner[r] = kf;
        const float nTotal = (float)kf_num_mps[kf];
        row_rhs[r] = __fmul_rn(__fdiv_rn((float)kf_count[kf], nTotal), (float)N);  // MapSparsification.cc:146-147
    }
}

}  // namespace msorb

using namespace msorb;

#define VCHK(expr)                                                             \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            rc = MSORB_E_HIP;                                                  \
            goto done;                                                         \
        }                                                                      \
    } while (0)

extern "C" int msorb_visibility_csr(int device, int n_window_kf, const int* kf_slot_begin, const int* slot_point,
                                    const int* slot_cell, int n_points, const int* point_nobs, const int* obs_begin,
                                    const int* obs_kf, int n_kf_total, const uint8_t* kf_in_window, const int* kf_num_mps,
                                    int N, int n_max_obs_floor, int* n_cols, int* col_point, int cap_cols, int* n_rows,
                                    int* row_begin, int* row_kind, int* row_owner, float* row_rhs, int cap_rows,
                                    int* col_idx, int cap_nnz, int* nnz_out, float* obj_coef, int* n_max_obs) {
    if (n_window_kf < 0 || n_points < 0 || n_kf_total < 0 || cap_cols < 0 || cap_rows < 0 || cap_nnz < 0 || !kf_slot_begin ||
        !n_cols || !n_rows || !row_begin || !nnz_out || !n_max_obs || (n_points > 0 && (!point_nobs || !obs_begin)) ||
        (cap_rows > 0 && (!row_kind || !row_owner || !row_rhs)) || (cap_nnz > 0 && !col_idx) ||
        (cap_cols > 0 && (!col_point || !obj_coef)) || (n_kf_total > 0 && (!kf_in_window || !kf_num_mps)) ||
        (kf_slot_begin[n_window_kf] > 0 && (!slot_point || !slot_cell)) ||
        (n_points > 0 && obs_begin[n_points] > 0 && !obs_kf)) {
        set_last_error("msorb_visibility_csr: null argument");
        return MSORB_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    static const bool vis_timing = getenv("MSORB_VIS_TIMING") != nullptr;   // read once per process: wall-clock breakdown on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    std::chrono::steady_clock::time_point t1, t2, t3, t4;
    const auto t0 = now();
    const int S = kf_slot_begin[n_window_kf];
    const int n_obs = n_points ? obs_begin[n_points] : 0;
    int n_valid = 0;   // slots that hold a point: every size below is bounded by it
    {   // range checks as two branch-free reductions (they vectorise; the kernels index with these values)
        int bad_slot = 0, bad_obs = 0;
        for (int s = 0; s < S; s++) { bad_slot |= slot_point[s] >= n_points; n_valid += slot_point[s] >= 0; }
        for (int o = 0; o < n_obs; o++) bad_obs |= (obs_kf[o] < 0) | (obs_kf[o] >= n_kf_total);
        if (bad_slot) { set_last_error("slot_point out of range"); return MSORB_E_INVALID; }
        if (bad_obs) { set_last_error("obs_kf out of range"); return MSORB_E_INVALID; }
    }
    t1 = now();
    int rc = MSORB_OK;
    if (hipSetDevice(device) != hipSuccess) return MSORB_E_HIP;
    // Scratch kept per calling thread (grow-only), a private non-blocking stream, pinned staging.  Shape of a call: the
    // inputs are packed into ONE pinned block and uploaded with one copy; every kernel takes the sizes it depends on (columns,
    // rows / non-zeros of the window part) from device scalars and is launched over host-side upper bounds, so nothing
    // waits for the host; the six scalars come back with the first synchronisation, the six output arrays (exact sizes)
    // with the second.  (Round 2: nine pageable uploads and about ten synchronous read-backs, 0.74 ms for a 30-keyframe window.)
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        void* p[2] = {nullptr, nullptr};
        size_t cap[2] = {0, 0};
        void* h = nullptr;
        size_t hcap = 0;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            for (int i = 0; i < 2; i++) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
            if (h) (void)hipHostFree(h);
            h = nullptr; hcap = 0;
            if (s) (void)hipStreamDestroy(s);
            s = nullptr; device = -1;
        }
        hipError_t ensure(int i, size_t bytes) {
            if (bytes <= cap[i]) return hipSuccess;
            if (p[i]) (void)hipFree(p[i]);
            p[i] = nullptr; cap[i] = 0;
            const hipError_t e = hipMalloc(&p[i], bytes + bytes / 4);
            if (e == hipSuccess) cap[i] = bytes + bytes / 4;
            return e;
        }
        hipError_t ensure_host(size_t bytes) {
            if (bytes <= hcap) return hipSuccess;
            if (h) (void)hipHostFree(h);
            h = nullptr; hcap = 0;
            const hipError_t e = hipHostMalloc(&h, bytes + bytes / 4, hipHostMallocDefault);
            if (e == hipSuccess) hcap = bytes + bytes / 4;
            return e;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch scr;
    if (scr.device != device) {
        scr.release();
        if (hipStreamCreateWithFlags(&scr.s, hipStreamNonBlocking) != hipSuccess) { set_last_error("stream creation failed"); return MSORB_E_HIP; }
        scr.device = device;
    }
    hipStream_t const st = scr.s;
    // host-side upper bounds of the result sizes
    int n_outside = 0;   // keyframes outside the window: bound of the kind-2 rows
    for (int k = 0; k < n_kf_total; k++) n_outside += !kf_in_window[k];
    if (n_obs == 0) n_outside = 0;
    const int cols_max = std::min(n_valid, n_points), rows_ab_max = n_valid + n_window_kf, rows_max = rows_ab_max + n_outside;
    const size_t nnz_max = (size_t)2 * n_valid + (size_t)n_obs;
    const int words = (cols_max + 31) / 32;
    int* d = nullptr;
    int scal[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += (n + 3) & ~size_t(3); return o; };
    // inputs (one upload) ...
    const size_t o_slot_begin = take(n_window_kf + 1), o_slot_point = take(S), o_slot_cell = take(S), o_nobs = take(n_points),
                 o_obs_begin = take(n_points + 1), o_obs_kf = take(n_obs), o_num_mps = take(n_kf_total),
                 o_inwin = take(((size_t)n_kf_total + 3) / 4), o_scal = take(8), n_in = off;
    // ... work arrays ...
    const size_t o_first = take(n_points), o_rank = take(S), o_colofp = take(n_points), o_kfvalid = take(n_window_kf + 1),
                 o_kfcells = take(n_window_kf + 1),
                 o_kfcount = take(n_kf_total), o_kfrow = take(n_kf_total), o_kfoff = take(n_kf_total);
    // ... outputs
    const size_t o_rowbegin = take((size_t)rows_max + 1), o_rowkind = take(rows_max), o_rowowner = take(rows_max), o_rhs = take(rows_max),
                 o_colpoint = take((size_t)cols_max + 1), o_colidx = take(nnz_max),
                 o_pack = take((size_t)8 + 4 * rows_max + cols_max + nnz_max);
    const size_t bitmap_bytes = (size_t)n_outside * words * sizeof(unsigned);
    const size_t pack_max = (size_t)8 + 4 * rows_max + cols_max + nnz_max;
    VCHK(scr.ensure(0, std::max<size_t>(off, 1) * sizeof(int)));
    VCHK(scr.ensure(1, std::max<size_t>(bitmap_bytes, 4)));
    VCHK(scr.ensure_host(std::max<size_t>(std::max(n_in, pack_max), 1) * sizeof(int)));
    d = static_cast<int*>(scr.p[0]);
    {
        int* h = static_cast<int*>(scr.h);
        std::memcpy(h + o_slot_begin, kf_slot_begin, (size_t)(n_window_kf + 1) * sizeof(int));
        if (S) {
            std::memcpy(h + o_slot_point, slot_point, (size_t)S * sizeof(int));
            std::memcpy(h + o_slot_cell, slot_cell, (size_t)S * sizeof(int));
        }
        if (n_points) {
            std::memcpy(h + o_nobs, point_nobs, (size_t)n_points * sizeof(int));
            std::memcpy(h + o_obs_begin, obs_begin, (size_t)(n_points + 1) * sizeof(int));
            if (n_obs) std::memcpy(h + o_obs_kf, obs_kf, (size_t)n_obs * sizeof(int));
        }
        if (n_kf_total) {
            std::memcpy(h + o_num_mps, kf_num_mps, (size_t)n_kf_total * sizeof(int));
            std::memcpy(h + o_inwin, kf_in_window, n_kf_total);
        }
        for (int i = 0; i < 8; i++) h[o_scal + i] = 0;
        h[o_scal + 1] = n_max_obs_floor;
        VCHK(hipMemcpyAsync(d, h, n_in * sizeof(int), hipMemcpyHostToDevice, st));
    }
    t2 = now();
    {
        const uint8_t* d_inwin = reinterpret_cast<const uint8_t*>(d + o_inwin);
        float* d_rhs = reinterpret_cast<float*>(d + o_rhs);
        unsigned* d_bitmap = static_cast<unsigned*>(scr.p[1]);
        // 9 launches (round 2: 12 kernels + 3 memsets): init | first + window-keyframe counts | first-flags per block | columns |
        // window-keyframe rows + outside-keyframe counts | their scan | their bitmaps | their rows | pack
        const size_t bitmap_words = (cols_max && n_outside) ? bitmap_bytes / sizeof(unsigned) : 0;
        {
            const size_t n_init = std::max<size_t>(std::max<size_t>(n_points, n_kf_total), bitmap_words);
            if (n_init)
                hipLaunchKernelGGL(vis_init_kernel, dim3((unsigned)std::min<size_t>((n_init + 255) / 256, 2048)), dim3(256), 0, st, d + o_first,
                                   n_points, d + o_kfcount, n_kf_total, d_bitmap, bitmap_words);
        }
        const int nb_first = S ? (S + kKfThreads - 1) / kKfThreads : 0;
        if (nb_first + n_window_kf > 0)
            hipLaunchKernelGGL(vis_first_kfcount_kernel, dim3(nb_first + n_window_kf), dim3(kKfThreads), 0, st, nb_first, d + o_slot_point, S,
                               d + o_first, d + o_slot_begin, d + o_slot_cell, d + o_kfvalid, d + o_kfcells);
        if (S) {
            const int nb = (S + 255) / 256;
            hipLaunchKernelGGL(vis_flag_count_kernel, dim3(nb), dim3(256), 0, st, d + o_slot_point, d + o_first, S, d + o_rank);
            hipLaunchKernelGGL(vis_columns_kernel, dim3(nb), dim3(256), 0, st, d + o_slot_point, d + o_first, d + o_rank, S,
                               d + o_nobs, d + o_colofp, d + o_colpoint, d + o_scal + 1, d + o_scal);
        }
        const bool extra = cols_max && n_outside;
        {
            const int nb_cols = extra ? (cols_max + kKfThreads - 1) / kKfThreads : 0;
            if (n_window_kf + nb_cols > 0)
                hipLaunchKernelGGL(vis_kf_rows_extra_count_kernel, dim3(n_window_kf + nb_cols), dim3(kKfThreads),
                                   n_kf_total <= kLdsHist ? (size_t)n_kf_total * sizeof(int) : 0, st, n_window_kf, d + o_slot_begin,
                                   d + o_slot_point, d + o_slot_cell, d + o_colofp, d + o_kfvalid, d + o_kfcells, N, d + o_rowbegin,
                                   d + o_rowkind, d + o_rowowner, d_rhs, d + o_colidx, d + o_scal, d + o_colpoint, d + o_obs_begin,
                                   d + o_obs_kf, d_inwin, n_kf_total, d + o_kfcount);
        }
        if (extra) {
            hipLaunchKernelGGL(vis_extra_scan_kernel, dim3(1), dim3(1024), 0, st, d + o_kfcount, n_kf_total, d + o_kfrow,
                               d + o_kfoff, d + o_scal + 2);
            hipLaunchKernelGGL(vis_extra_bits_kernel, dim3((cols_max + 255) / 256), dim3(256), 0, st, d + o_colpoint, d + o_scal,
                               d + o_obs_begin, d + o_obs_kf, d_inwin, d + o_kfrow, words, d_bitmap);
            hipLaunchKernelGGL(vis_extra_emit_kernel, dim3(n_kf_total), dim3(256), 0, st, n_kf_total, d + o_kfcount,
                               d + o_kfrow, d + o_kfoff, d + o_num_mps, N, words, d_bitmap, d + o_scal, d + o_rowbegin,
                               d + o_rowkind, d + o_rowowner, d_rhs, d + o_colidx);
        }
        int* h = static_cast<int*>(scr.h);
        hipLaunchKernelGGL(vis_pack_kernel, dim3((unsigned)std::min<size_t>((pack_max + 255) / 256, 1024)), dim3(256), 0, st, d + o_scal,
                           d + o_rowbegin, d + o_rowkind, d + o_rowowner, reinterpret_cast<const int*>(d_rhs), d + o_colpoint,
                           d + o_colidx, d + o_pack);
        VCHK(hipMemcpyAsync(h, d + o_pack, pack_max * sizeof(int), hipMemcpyDeviceToHost, st));
        t3 = now();
        VCHK(hipStreamSynchronize(st));   // the call's only synchronisation
        t4 = now();
        VCHK(hipGetLastError());
        for (int i = 0; i < 8; i++) scal[i] = h[i];
        const int ncols = scal[0], nmax = scal[1], R = scal[4] + scal[2], NNZ = scal[5] + scal[3];
        if (ncols > cap_cols || R > cap_rows || NNZ > cap_nnz) { set_last_error("output capacity too small"); rc = MSORB_E_CAPACITY; goto done; }
        *n_cols = ncols; *n_rows = R; *nnz_out = NNZ; *n_max_obs = nmax;
        int* hb = h + 8;
        int *h_rb = hb, *h_rk = h_rb + R, *h_ro = h_rk + R, *h_cp = h_ro + R + R /* rhs in between */, *h_ci = h_cp + ncols;
        float* h_rhs = reinterpret_cast<float*>(h_ro + R);
        if (R) {
            std::memcpy(row_begin, h_rb, (size_t)R * sizeof(int));
            std::memcpy(row_kind, h_rk, (size_t)R * sizeof(int));
            std::memcpy(row_owner, h_ro, (size_t)R * sizeof(int));
            std::memcpy(row_rhs, h_rhs, (size_t)R * sizeof(float));
        }
        row_begin[R] = NNZ;
        if (NNZ) std::memcpy(col_idx, h_ci, (size_t)NNZ * sizeof(int));
        if (ncols) {
            std::memcpy(col_point, h_cp, (size_t)ncols * sizeof(int));
            for (int c = 0; c < ncols; c++) obj_coef[c] = (float)(nmax - point_nobs[col_point[c]]);  // MapSparsification.cc:95-96
        }
        if (vis_timing) {
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "vis timing us: checks %.1f  scratch+staging+upload-enqueue %.1f  launches-enqueue %.1f  wait %.1f  copy-out %.1f\n", us(t0, t1),
                    us(t1, t2), us(t2, t3), us(t3, t4), us(t4, now()));
        }
    }
done:
    return rc;
}
