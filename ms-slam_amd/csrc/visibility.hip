// Keyframe x map-point visibility / constraint matrix of MapSparsification::Sparsifying
// (MapSparsification.cc:58-151), assembled on the device as CSR (rows in the order the reference adds
// constraints, columns = map points in first-encounter order).  Sort-free: first-encounter order comes from
// an atomicMin + flag scan, the outside-keyframe rows from per-row bitmaps whose prefix popcount yields the
// ascending column order the reference's accumulation produces.
#include <hip/hip_runtime.h>

#include <algorithm>
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

// Single-block exclusive scans (chunk per thread).  mode 0: is_first flags -> column index per slot + n_cols.
__global__ __launch_bounds__(1024) void vis_scan_first_kernel(const int* __restrict__ slot_point,
                                                              const int* __restrict__ first, int n_slots,
                                                              int* __restrict__ slot_rank /* exclusive rank among firsts */,
                                                              int* __restrict__ n_cols) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n_slots + 1023) / 1024;
    const int b = tid * per, e = min(b + per, n_slots);
    int c = 0;
    for (int s = b; s < e; s++) { const int p = slot_point[s]; c += (p >= 0 && first[p] == s); }
    part[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 1024; i++) { const int v = part[i]; part[i] = acc; acc += v; }
        *n_cols = acc;
    }
    __syncthreads();
    int acc = part[tid];
    for (int s = b; s < e; s++) {
        const int p = slot_point[s];
        slot_rank[s] = acc;
        acc += (p >= 0 && first[p] == s);
    }
}

// column of every point; column tables; running max of observations
__global__ void vis_columns_kernel(const int* __restrict__ slot_point, const int* __restrict__ first,
                                   const int* __restrict__ slot_rank, int n_slots, const int* __restrict__ point_nobs,
                                   int* __restrict__ col_of_point, int* __restrict__ col_point, int* __restrict__ n_max_obs) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int p = slot_point[s];
    if (p < 0) return;
    atomicMax(n_max_obs, point_nobs[p]);
    if (first[p] == s) {
        col_of_point[p] = slot_rank[s];
        col_point[slot_rank[s]] = p;
    }
}

// One block per window keyframe: counts of valid slots and valid cells.
__global__ __launch_bounds__(256) void vis_kf_count_kernel(const int* __restrict__ kf_slot_begin,
                                                           const int* __restrict__ slot_point,
                                                           const int* __restrict__ slot_cell, int* __restrict__ kf_valid,
                                                           int* __restrict__ kf_cells) {
    __shared__ int sv, sc;
    const int k = blockIdx.x;
    if (threadIdx.x == 0) { sv = 0; sc = 0; }
    __syncthreads();
    const int b = kf_slot_begin[k], e = kf_slot_begin[k + 1];
    int v = 0, c = 0;
    for (int s = b + threadIdx.x; s < e; s += 256) {
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

// One block per window keyframe: emit its cell rows then its keyframe row.  Serial-in-order within the
// block's thread 0 would be O(slots); instead each thread ranks its slots with block-wide prefix sums.
__global__ __launch_bounds__(256) void vis_kf_rows_kernel(const int* __restrict__ kf_slot_begin,
                                                          const int* __restrict__ slot_point,
                                                          const int* __restrict__ slot_cell,
                                                          const int* __restrict__ col_of_point,
                                                          const int* __restrict__ kf_row_base /* rows before kf */,
                                                          const int* __restrict__ kf_nnz_base /* nnz before kf */,
                                                          const int* __restrict__ kf_valid, int N,
                                                          int* __restrict__ row_begin, int* __restrict__ row_kind,
                                                          int* __restrict__ row_owner, float* __restrict__ row_rhs,
                                                          int* __restrict__ col_idx) {
    __shared__ int part_v[256], part_c[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int b = kf_slot_begin[k], e = kf_slot_begin[k + 1];
    const int n = e - b;
    const int per = (n + 255) / 256;
    const int tb = b + tid * per, te = min(tb + per, e);
    int v = 0, c = 0;
    for (int s = tb; s < te; s++) {
        if (slot_point[s] < 0) continue;
        v++;
        bool first_valid = true;
        for (int t = s - 1; t >= b && slot_cell[t] == slot_cell[s]; t--)
            if (slot_point[t] >= 0) { first_valid = false; break; }
        c += first_valid;
    }
    part_v[tid] = v; part_c[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int av = 0, ac = 0;
        for (int i = 0; i < 256; i++) {
            const int x = part_v[i], y = part_c[i];
            part_v[i] = av; part_c[i] = ac;
            av += x; ac += y;
        }
    }
    __syncthreads();
    int rv = part_v[tid], rc = part_c[tid];  // valid slots / valid cells before this thread's chunk
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
    if (tid == 255) {  // rc after the last chunk = number of valid cells
        const int r = row0 + rc;
        row_begin[r] = nnz0 + V;
        row_kind[r] = 1; row_owner[r] = k; row_rhs[r] = (float)N;
    }
}

// outside-keyframe rows: count, bitmap, emit
__global__ void vis_extra_count_kernel(const int* __restrict__ col_point, int n_cols, const int* __restrict__ obs_begin,
                                       const int* __restrict__ obs_kf, const uint8_t* __restrict__ kf_in_window,
                                       int* __restrict__ kf_count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const int p = col_point[c];
    for (int o = obs_begin[p]; o < obs_begin[p + 1]; o++) {
        const int kf = obs_kf[o];
        if (!kf_in_window[kf]) atomicAdd(&kf_count[kf], 1);
    }
}
__global__ __launch_bounds__(1024) void vis_extra_scan_kernel(const int* __restrict__ kf_count, int n_kf,
                                                              int* __restrict__ kf_row /* rank among count>0 */,
                                                              int* __restrict__ kf_off /* nnz offset */,
                                                              int* __restrict__ totals /* [n_rows_c, nnz_c] */) {
    __shared__ int pr[1024], pn[1024];
    const int tid = threadIdx.x;
    const int per = (n_kf + 1023) / 1024;
    const int b = tid * per, e = min(b + per, n_kf);
    int r = 0, n = 0;
    for (int i = b; i < e; i++) { r += kf_count[i] > 0; n += kf_count[i]; }
    pr[tid] = r; pn[tid] = n;
    __syncthreads();
    if (tid == 0) {
        int ar = 0, an = 0;
        for (int i = 0; i < 1024; i++) {
            const int x = pr[i], y = pn[i];
            pr[i] = ar; pn[i] = an;
            ar += x; an += y;
        }
        totals[0] = ar; totals[1] = an;
    }
    __syncthreads();
    int ar = pr[tid], an = pn[tid];
    for (int i = b; i < e; i++) {
        kf_row[i] = kf_count[i] > 0 ? ar : -1;
        kf_off[i] = an;
        ar += kf_count[i] > 0; an += kf_count[i];
    }
}
__global__ void vis_extra_bits_kernel(const int* __restrict__ col_point, int n_cols, const int* __restrict__ obs_begin,
                                      const int* __restrict__ obs_kf, const uint8_t* __restrict__ kf_in_window,
                                      const int* __restrict__ kf_row, int words, unsigned* __restrict__ bitmap) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
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
                                                             const unsigned* __restrict__ bitmap, int row_base, int nnz_base,
                                                             int* __restrict__ row_begin, int* __restrict__ row_kind,
                                                             int* __restrict__ row_owner, float* __restrict__ row_rhs,
                                                             int* __restrict__ col_idx) {
    __shared__ int part[256];
    const int kf = blockIdx.x, tid = threadIdx.x;
    if (kf >= n_kf || kf_count[kf] == 0) return;
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
    for (int s = 0; s < S; s++)
        if (slot_point[s] >= n_points) { set_last_error("slot_point out of range"); return MSORB_E_INVALID; }
    for (int o = 0; o < n_obs; o++)
        if (obs_kf[o] < 0 || obs_kf[o] >= n_kf_total) { set_last_error("obs_kf out of range"); return MSORB_E_INVALID; }
    int rc = MSORB_OK;
    if (hipSetDevice(device) != hipSuccess) return MSORB_E_HIP;
    // scratch kept per calling thread (grow-only) and a private non-blocking stream, like the other per-frame entries
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        void* p[4] = {nullptr, nullptr, nullptr, nullptr};
        size_t cap[4] = {0, 0, 0, 0};
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            for (int i = 0; i < 4; i++) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
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
        ~Scratch() { release(); }
    };
    static thread_local Scratch scr;
    if (scr.device != device) {
        scr.release();
        if (hipStreamCreateWithFlags(&scr.s, hipStreamNonBlocking) != hipSuccess) { set_last_error("stream creation failed"); return MSORB_E_HIP; }
        scr.device = device;
    }
    hipStream_t const st = scr.s;
    // one arena of ints
    int *d = nullptr;
    unsigned* d_bitmap = nullptr;
    float* d_rhs = nullptr;
    uint8_t* d_inwin = nullptr;
    std::vector<int> kf_valid(n_window_kf + 1), kf_cells(n_window_kf + 1), row_base(n_window_kf + 1), nnz_base(n_window_kf + 1);
    int totals[2] = {0, 0}, ncols = 0, nmax = 0, rows_ab = 0, nnz_ab = 0, words = 0;
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += (n + 3) & ~size_t(3); return o; };
    const size_t o_slot_begin = take(n_window_kf + 1), o_slot_point = take(S), o_slot_cell = take(S), o_nobs = take(n_points),
                 o_obs_begin = take(n_points + 1), o_obs_kf = take(n_obs), o_num_mps = take(n_kf_total), o_first = take(n_points),
                 o_rank = take(S), o_colofp = take(n_points), o_colpoint = take(S + 1), o_scal = take(8),
                 o_kfvalid = take(n_window_kf + 1), o_kfcells = take(n_window_kf + 1), o_rowbase = take(n_window_kf + 1),
                 o_nnzbase = take(n_window_kf + 1), o_kfcount = take(n_kf_total), o_kfrow = take(n_kf_total),
                 o_kfoff = take(n_kf_total), o_rowbegin = take((size_t)cap_rows + 1), o_rowkind = take(cap_rows),
                 o_rowowner = take(cap_rows), o_colidx = take(cap_nnz);
    VCHK(scr.ensure(0, std::max<size_t>(off, 1) * sizeof(int)));
    VCHK(scr.ensure(1, std::max<size_t>(cap_rows, 1) * sizeof(float)));
    VCHK(scr.ensure(2, std::max(n_kf_total, 1)));
    d = static_cast<int*>(scr.p[0]); d_rhs = static_cast<float*>(scr.p[1]); d_inwin = static_cast<uint8_t*>(scr.p[2]);
    VCHK(hipMemcpyAsync(d + o_slot_begin, kf_slot_begin, (n_window_kf + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    if (S) {
        VCHK(hipMemcpyAsync(d + o_slot_point, slot_point, S * sizeof(int), hipMemcpyHostToDevice, st));
        VCHK(hipMemcpyAsync(d + o_slot_cell, slot_cell, S * sizeof(int), hipMemcpyHostToDevice, st));
    }
    if (n_points) {
        VCHK(hipMemcpyAsync(d + o_nobs, point_nobs, n_points * sizeof(int), hipMemcpyHostToDevice, st));
        VCHK(hipMemcpyAsync(d + o_obs_begin, obs_begin, (n_points + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (n_obs) VCHK(hipMemcpyAsync(d + o_obs_kf, obs_kf, n_obs * sizeof(int), hipMemcpyHostToDevice, st));
        VCHK(hipMemsetAsync(d + o_first, 0x7f, n_points * sizeof(int), st));
    }
    if (n_kf_total) {
        VCHK(hipMemcpyAsync(d + o_num_mps, kf_num_mps, n_kf_total * sizeof(int), hipMemcpyHostToDevice, st));
        VCHK(hipMemcpyAsync(d_inwin, kf_in_window, n_kf_total, hipMemcpyHostToDevice, st));
        VCHK(hipMemsetAsync(d + o_kfcount, 0, n_kf_total * sizeof(int), st));
    }
    VCHK(hipMemsetAsync(d + o_scal, 0, 8 * sizeof(int), st));
    VCHK(hipMemcpyAsync(d + o_scal + 1, &n_max_obs_floor, sizeof(int), hipMemcpyHostToDevice, st));
    if (S) {
        hipLaunchKernelGGL(vis_first_kernel, dim3((S + 255) / 256), dim3(256), 0, st, d + o_slot_point, S, d + o_first);
        hipLaunchKernelGGL(vis_scan_first_kernel, dim3(1), dim3(1024), 0, st, d + o_slot_point, d + o_first, S, d + o_rank,
                           d + o_scal);
        hipLaunchKernelGGL(vis_columns_kernel, dim3((S + 255) / 256), dim3(256), 0, st, d + o_slot_point, d + o_first,
                           d + o_rank, S, d + o_nobs, d + o_colofp, d + o_colpoint, d + o_scal + 1);
    }
    if (n_window_kf)
        hipLaunchKernelGGL(vis_kf_count_kernel, dim3(n_window_kf), dim3(256), 0, st, d + o_slot_begin, d + o_slot_point,
                           d + o_slot_cell, d + o_kfvalid, d + o_kfcells);
    VCHK(hipMemcpyAsync(&ncols, d + o_scal, sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
    VCHK(hipMemcpyAsync(&nmax, d + o_scal + 1, sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
    if (n_window_kf) {
        VCHK(hipMemcpyAsync(kf_valid.data(), d + o_kfvalid, n_window_kf * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
        VCHK(hipMemcpyAsync(kf_cells.data(), d + o_kfcells, n_window_kf * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
    }
    for (int k = 0; k < n_window_kf; k++) {
        row_base[k] = rows_ab; nnz_base[k] = nnz_ab;
        rows_ab += kf_cells[k] + 1;
        nnz_ab += 2 * kf_valid[k];
    }
    if (ncols > cap_cols || rows_ab > cap_rows || nnz_ab > cap_nnz) { set_last_error("output capacity too small"); rc = MSORB_E_CAPACITY; goto done; }
    if (n_window_kf) {
        VCHK(hipMemcpyAsync(d + o_rowbase, row_base.data(), n_window_kf * sizeof(int), hipMemcpyHostToDevice, st));
        VCHK(hipMemcpyAsync(d + o_nnzbase, nnz_base.data(), n_window_kf * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(vis_kf_rows_kernel, dim3(n_window_kf), dim3(256), 0, st, d + o_slot_begin, d + o_slot_point,
                           d + o_slot_cell, d + o_colofp, d + o_rowbase, d + o_nnzbase, d + o_kfvalid, N, d + o_rowbegin,
                           d + o_rowkind, d + o_rowowner, d_rhs, d + o_colidx);
    }
    if (ncols && n_kf_total) {
        hipLaunchKernelGGL(vis_extra_count_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, d + o_colpoint, ncols,
                           d + o_obs_begin, d + o_obs_kf, d_inwin, d + o_kfcount);
        hipLaunchKernelGGL(vis_extra_scan_kernel, dim3(1), dim3(1024), 0, st, d + o_kfcount, n_kf_total, d + o_kfrow,
                           d + o_kfoff, d + o_scal + 2);
        VCHK(hipMemcpyAsync(totals, d + o_scal + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
        if (rows_ab + totals[0] > cap_rows || nnz_ab + totals[1] > cap_nnz) { set_last_error("output capacity too small"); rc = MSORB_E_CAPACITY; goto done; }
        if (totals[0]) {
            words = (ncols + 31) / 32;
            VCHK(scr.ensure(3, (size_t)totals[0] * words * sizeof(unsigned)));
            d_bitmap = static_cast<unsigned*>(scr.p[3]);
            VCHK(hipMemsetAsync(d_bitmap, 0, (size_t)totals[0] * words * sizeof(unsigned), st));
            hipLaunchKernelGGL(vis_extra_bits_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, d + o_colpoint, ncols,
                               d + o_obs_begin, d + o_obs_kf, d_inwin, d + o_kfrow, words, d_bitmap);
            hipLaunchKernelGGL(vis_extra_emit_kernel, dim3(n_kf_total), dim3(256), 0, st, n_kf_total, d + o_kfcount,
                               d + o_kfrow, d + o_kfoff, d + o_num_mps, N, words, d_bitmap, rows_ab, nnz_ab, d + o_rowbegin,
                               d + o_rowkind, d + o_rowowner, d_rhs, d + o_colidx);
        }
    }
    VCHK(hipStreamSynchronize(st));
    {
        const int R = rows_ab + totals[0], NNZ = nnz_ab + totals[1];
        *n_cols = ncols; *n_rows = R; *nnz_out = NNZ; *n_max_obs = nmax;
        if (R) {
            VCHK(hipMemcpyAsync(row_begin, d + o_rowbegin, R * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
            VCHK(hipMemcpyAsync(row_kind, d + o_rowkind, R * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
            VCHK(hipMemcpyAsync(row_owner, d + o_rowowner, R * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
            VCHK(hipMemcpyAsync(row_rhs, d_rhs, R * sizeof(float), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
        }
        row_begin[R] = NNZ;
        if (NNZ) VCHK(hipMemcpyAsync(col_idx, d + o_colidx, NNZ * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
        if (ncols) {
            VCHK(hipMemcpyAsync(col_point, d + o_colpoint, ncols * sizeof(int), hipMemcpyDeviceToHost, st)); VCHK(hipStreamSynchronize(st));
            for (int c = 0; c < ncols; c++) obj_coef[c] = (float)(nmax - point_nobs[col_point[c]]);  // MapSparsification.cc:95-96
        }
    }
done:
    return rc;
}
