// Keyframe x map-point visibility / constraint matrix of MapSparsification::Sparsifying
// (MapSparsification.cc:58-151), assembled on the device as CSR (rows in the order the reference adds
// constraints, columns = map points in first-encounter order).  Sort-free: first-encounter order comes from
// an atomicMin + flag scan, the outside-keyframe rows from per-row bitmaps whose prefix popcount yields the
// ascending column order the reference's accumulation produces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msorb.h"

namespace msorb {
void set_last_error(const std::string& s);

// first[p] = first slot (in window order) that references point p
__global__ void vis_first_kernel(const int* __restrict__ slot_point, int n_slots, int* __restrict__ first) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int p = slot_point[s];
    if (p >= 0) atomicMin(&first[p], s);
}

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
__global__ __launch_bounds__(1024) void vis_block_scan_kernel(int* __restrict__ block_count, int n_blocks, int* __restrict__ n_cols) {
    __shared__ int wt[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? block_count[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, wt, &tot);
        const int carry = carry_s;
        if (i < n_blocks) block_count[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_cols = carry_s;
}
// column of every point; column tables; running max of observations
__global__ __launch_bounds__(256) void vis_columns_kernel(const int* __restrict__ slot_point, const int* __restrict__ first,
                                                          const int* __restrict__ block_base, int n_slots,
                                                          const int* __restrict__ point_nobs, int* __restrict__ col_of_point,
                                                          int* __restrict__ col_point, int* __restrict__ n_max_obs) {
    __shared__ int wt[16];
    __shared__ int smax;
    if (threadIdx.x == 0) smax = 0;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int p = s < n_slots ? slot_point[s] : -1;
    const int f = (p >= 0 && first[p] == s);
    const int rank = block_base[blockIdx.x] + block_exclusive_scan(f, wt, nullptr);   // (the scan synchronises: smax is set)
    if (p >= 0) atomicMax(&smax, point_nobs[p]);
    if (f) { col_of_point[p] = rank; col_point[rank] = p; }
    __syncthreads();
    if (threadIdx.x == 0 && smax > 0) atomicMax(n_max_obs, smax);
}

// One block per window keyframe: counts of valid slots and valid cells.
constexpr int kKfThreads = 1024;   // one workgroup per window keyframe: its ~2000 slots are two per thread (latency, not work, is the cost)
__global__ __launch_bounds__(kKfThreads) void vis_kf_count_kernel(const int* __restrict__ kf_slot_begin,
                                                           const int* __restrict__ slot_point,
                                                           const int* __restrict__ slot_cell, int* __restrict__ kf_valid,
                                                           int* __restrict__ kf_cells) {
    __shared__ int sv, sc;
    const int k = blockIdx.x;
    if (threadIdx.x == 0) { sv = 0; sc = 0; }
    __syncthreads();
    const int b = kf_slot_begin[k], e = kf_slot_begin[k + 1];
    int v = 0, c = 0;
    for (int s = b + threadIdx.x; s < e; s += kKfThreads) {
        if (slot_point[s] < 0) continue;
        v++;
        // first valid slot of its cell <=> no earlier valid slot with the same cell (cells are contiguous runs)
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

// rows / non-zeros in front of every window keyframe (one thread: a window is <= a few dozen keyframes); totals in scal[4], scal[5]
__global__ void vis_kf_base_kernel(int n_kf, const int* __restrict__ kf_valid, const int* __restrict__ kf_cells,
                                   int* __restrict__ row_base, int* __restrict__ nnz_base, int* __restrict__ scal) {
    if (blockIdx.x || threadIdx.x) return;
    int rows = 0, nnz = 0;
    for (int k = 0; k < n_kf; k++) {
        row_base[k] = rows; nnz_base[k] = nnz;
        rows += kf_cells[k] + 1;
        nnz += 2 * kf_valid[k];
    }
    scal[4] = rows; scal[5] = nnz;
}

// One block per window keyframe: emit its cell rows then its keyframe row.  Serial-in-order within the
// block's thread 0 would be O(slots); instead each thread ranks its slots with block-wide prefix sums.
__global__ __launch_bounds__(kKfThreads) void vis_kf_rows_kernel(const int* __restrict__ kf_slot_begin,
                                                          const int* __restrict__ slot_point,
                                                          const int* __restrict__ slot_cell,
                                                          const int* __restrict__ col_of_point,
                                                          const int* __restrict__ kf_row_base /* rows before kf */,
                                                          const int* __restrict__ kf_nnz_base /* nnz before kf */,
                                                          const int* __restrict__ kf_valid, int N,
                                                          int* __restrict__ row_begin, int* __restrict__ row_kind,
                                                          int* __restrict__ row_owner, float* __restrict__ row_rhs,
                                                          int* __restrict__ col_idx) {
    __shared__ int wt[16];
    const int k = blockIdx.x, tid = threadIdx.x;
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
    const int nnz0 = kf_nnz_base[k], row0 = kf_row_base[k];
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
__global__ __launch_bounds__(256) void vis_extra_count_kernel(const int* __restrict__ col_point, const int* __restrict__ scal,
                                                              const int* __restrict__ obs_begin, const int* __restrict__ obs_kf,
                                                              const uint8_t* __restrict__ kf_in_window, int n_kf,
                                                              int* __restrict__ kf_count) {
    extern __shared__ int hist[];
    const bool lds = n_kf <= kLdsHist;
    if (lds) {
        for (int i = threadIdx.x; i < n_kf; i += 256) hist[i] = 0;
        __syncthreads();
    }
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < scal[0]) {
        const int p = col_point[c];
        for (int o = obs_begin[p]; o < obs_begin[p + 1]; o++) {
            const int kf = obs_kf[o];
            if (!kf_in_window[kf]) atomicAdd(lds ? &hist[kf] : &kf_count[kf], 1);
        }
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_kf; i += 256)
            if (hist[i]) atomicAdd(&kf_count[i], hist[i]);
    }
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
                 o_kfcells = take(n_window_kf + 1), o_rowbase = take(n_window_kf + 1), o_nnzbase = take(n_window_kf + 1),
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
    {
        const uint8_t* d_inwin = reinterpret_cast<const uint8_t*>(d + o_inwin);
        float* d_rhs = reinterpret_cast<float*>(d + o_rhs);
        unsigned* d_bitmap = static_cast<unsigned*>(scr.p[1]);
        if (n_points) VCHK(hipMemsetAsync(d + o_first, 0x7f, (size_t)n_points * sizeof(int), st));
        if (n_kf_total) VCHK(hipMemsetAsync(d + o_kfcount, 0, (size_t)n_kf_total * sizeof(int), st));
        if (S) {
            const int nb = (S + 255) / 256;
            hipLaunchKernelGGL(vis_first_kernel, dim3(nb), dim3(256), 0, st, d + o_slot_point, S, d + o_first);
            hipLaunchKernelGGL(vis_flag_count_kernel, dim3(nb), dim3(256), 0, st, d + o_slot_point, d + o_first, S, d + o_rank);
            hipLaunchKernelGGL(vis_block_scan_kernel, dim3(1), dim3(1024), 0, st, d + o_rank, nb, d + o_scal);
            hipLaunchKernelGGL(vis_columns_kernel, dim3(nb), dim3(256), 0, st, d + o_slot_point, d + o_first, d + o_rank, S,
                               d + o_nobs, d + o_colofp, d + o_colpoint, d + o_scal + 1);
        }
        if (n_window_kf) {
            hipLaunchKernelGGL(vis_kf_count_kernel, dim3(n_window_kf), dim3(kKfThreads), 0, st, d + o_slot_begin, d + o_slot_point,
                               d + o_slot_cell, d + o_kfvalid, d + o_kfcells);
            hipLaunchKernelGGL(vis_kf_base_kernel, dim3(1), dim3(64), 0, st, n_window_kf, d + o_kfvalid, d + o_kfcells, d + o_rowbase,
                               d + o_nnzbase, d + o_scal);
            hipLaunchKernelGGL(vis_kf_rows_kernel, dim3(n_window_kf), dim3(kKfThreads), 0, st, d + o_slot_begin, d + o_slot_point,
                               d + o_slot_cell, d + o_colofp, d + o_rowbase, d + o_nnzbase, d + o_kfvalid, N, d + o_rowbegin,
                               d + o_rowkind, d + o_rowowner, d_rhs, d + o_colidx);
        }
        if (cols_max && n_outside) {
            hipLaunchKernelGGL(vis_extra_count_kernel, dim3((cols_max + 255) / 256), dim3(256),
                               n_kf_total <= kLdsHist ? (size_t)n_kf_total * sizeof(int) : 0, st, d + o_colpoint, d + o_scal,
                               d + o_obs_begin, d + o_obs_kf, d_inwin, n_kf_total, d + o_kfcount);
            hipLaunchKernelGGL(vis_extra_scan_kernel, dim3(1), dim3(1024), 0, st, d + o_kfcount, n_kf_total, d + o_kfrow,
                               d + o_kfoff, d + o_scal + 2);
            VCHK(hipMemsetAsync(d_bitmap, 0, bitmap_bytes, st));
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
        VCHK(hipStreamSynchronize(st));   // the call's only synchronisation
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
    }
done:
    return rc;
}
