// The tracking front-end of one frame as ONE device-resident chain (BASELINE configs[2]):
//
//   Frame::Frame(imLeft, imRight, ...)       Frame.cc:119-137   extraction of both eyes + ComputeStereoMatches
//   Frame::AssignFeaturesToGrid              Frame.cc:385-416   (PosInGrid :657-667)        -> frame_grid_kernel
//   Tracking::SearchLocalPoints              Tracking.cc:3343-3388
//     Frame::isInFrustum per local map point Frame.cc:512-571                                -> local_points_kernel
//     ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints)  ORBmatcher.cc:43-142
//                                                                                            -> window_topk_kernel + host replay
//
// The extractor's outputs never leave the device on their way into the matcher: the grid is built from the device
// keypoints (counting sort that keeps the ascending keypoint index inside a cell = the reference's push_back order), the
// frustum test writes the window queries the search kernel reads, and one block comes back to the host for the
// sequential claim replay (matcher_host.h).  Entries: msorb_frame_set_device, msorb_extract_stereo_frame,
// msorb_search_local_points, msorb_track_frontend, msorb_track_batch (include/msorb.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "frustum_device.h"
#include "lds_limit.h"
#include "matcher_host.h"

using namespace msorb;

namespace {

constexpr int kNCell = kGridCols * kGridRows;
constexpr int kGridThreads = 1024;

// ---------------------------------------------------------------------------------------------------------------------
// TrackWithMotionModel's search, ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1941-2152):
// the per-point part of the loop up to the window (:1962-1998) — one thread per last-frame keypoint, window query out.
//   x3Dc = Tcw * x3Dw is Sophus' SE3 action (se3.hpp:321-324 -> so3.hpp:358-367): uv = q.vec x p; uv += uv;
//   p + q.w * uv + q.vec x uv, then + translation.  Float convention (as for isInFrustum): of a difference of two
//   products the first one is fused (a*b - c*d -> fma(a, b, -(c*d))), a product added to a term is fused (p + w*uv ->
//   fma(w, uv, p)); sums stay sums.  Which products an actual -O3 -march=native build contracts is the compiler's choice (GCC 11
//   SLP-vectorises this very expression with fmsub in some lanes and fnmadd in others): msorb_search_by_projection_frames takes
//   coordinates projected by the caller's own build when the last ulp matters.
// ---------------------------------------------------------------------------------------------------------------------
struct LastFrameArgs {
    float qx, qy, qz, qw, tx, ty, tz;
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float th;
    int forward, backward, n;
    float scale[MSORB_MAX_LEVELS];
    const float* pos_w;      // [3n]
    const int* octave;       // [n]
    const uint8_t* flags;    // [n] bit 0: the keypoint holds a map point that is not an outlier
    WinQuery* q;
    float *u, *v, *ur;
    uint8_t* valid;
};

__device__ __forceinline__ float diff_of_products(float a, float b, float c, float d) { return __fmaf_rn(a, b, -__fmul_rn(c, d)); }

__device__ __forceinline__ void last_frame_point(const LastFrameArgs& A, const int i) {
    if (i >= A.n) return;
    WinQuery w;
    w.x = 0; w.y = 0; w.r = 0; w.ur = 0; w.min_level = 0; w.max_level = 0; w.flags = 0; w.pad[0] = w.pad[1] = w.pad[2] = 0;
    float u = 0.0f, v = 0.0f, ur = 0.0f;
    uint8_t ok = 0;
    if (A.flags[i] & 1) {
        const float px = A.pos_w[3 * i], py = A.pos_w[3 * i + 1], pz = A.pos_w[3 * i + 2];
        float uvx = diff_of_products(A.qy, pz, A.qz, py), uvy = diff_of_products(A.qz, px, A.qx, pz), uvz = diff_of_products(A.qx, py, A.qy, px);
        uvx = __fadd_rn(uvx, uvx); uvy = __fadd_rn(uvy, uvy); uvz = __fadd_rn(uvz, uvz);
        const float cx_ = diff_of_products(A.qy, uvz, A.qz, uvy), cy_ = diff_of_products(A.qz, uvx, A.qx, uvz), cz_ = diff_of_products(A.qx, uvy, A.qy, uvx);
        const float xc = __fadd_rn(__fadd_rn(__fmaf_rn(A.qw, uvx, px), cx_), A.tx);
        const float yc = __fadd_rn(__fadd_rn(__fmaf_rn(A.qw, uvy, py), cy_), A.ty);
        const float zc = __fadd_rn(__fadd_rn(__fmaf_rn(A.qw, uvz, pz), cz_), A.tz);
        // :1973 `const float invzc = 1.0/x3Dc(2)`: the double quotient rounded to float IS the float quotient (53 >= 2*24 + 2)
        const float invzc = __fdiv_rn(1.0f, zc);
        if (!(invzc < 0.0f)) {
            u = __fadd_rn(__fdiv_rn(__fmul_rn(A.fx, xc), zc), A.cx);   // Pinhole::project, Pinhole.cpp:43-49
            v = __fadd_rn(__fdiv_rn(__fmul_rn(A.fy, yc), zc), A.cy);
            if (!(u < A.min_x || u > A.max_x) && !(v < A.min_y || v > A.max_y)) {   // :1980-1983
                const int oct = A.octave[i];
                ok = 1;
                ur = __fmaf_rn(-A.mbf, invzc, u);                                  // :2019
                w.x = u; w.y = v; w.ur = ur;
                w.r = __fmul_rn(A.th, A.scale[oct]);                               // :1989
                if (A.forward) { w.min_level = (int16_t)oct; w.max_level = -1; }  // :1993-1998
                else if (A.backward) { w.min_level = 0; w.max_level = (int16_t)oct; }
                else { w.min_level = (int16_t)(oct - 1); w.max_level = (int16_t)(oct + 1); }
                w.flags = kQValid | kQSkipOccupied;
            }
        }
    }
    A.q[i] = w;
    A.u[i] = u; A.v[i] = v; A.ur[i] = ur; A.valid[i] = ok;
}
__global__ __launch_bounds__(256) void last_frame_kernel(LastFrameArgs A) { last_frame_point(A, blockIdx.x * blockDim.x + threadIdx.x); }

// ---------------------------------------------------------------------------------------------------------------------
// Frame::AssignFeaturesToGrid (Frame.cc:385-416) for a batch of frames, one workgroup per frame.  Sources: the extractor's
// device outputs (28-byte keypoints, 32-byte descriptors, mvuRight).  Products: the matcher's train arrays (KpLite,
// descriptors, occupancy cleared) and mGrid as CSR (cell = ix * 48 + iy, ascending keypoint index inside a cell).
// ---------------------------------------------------------------------------------------------------------------------
struct GridArgs {
    const msorb_keypoint* kps;   // frame b: kps + b * src_stride
    const uint8_t* desc;         //          desc + b * src_stride * 32
    const float* u_right;        //          u_right + b * ur_stride (nullptr: -1 everywhere)
    size_t src_stride, ur_stride;
    const int* counts;           // n of frame b = counts[b * count_step] (nullptr: n_fixed)
    int count_step, n_fixed;
    float minX, minY, gridWInv, gridHInv;
    KpLite* kp;                  // outputs, frame b: + b * dst_stride (cell_begin: + b * (kNCell + 1))
    uint8_t* desc_out;
    uint8_t* occ;
    int* cell_begin;
    int* cell_idx;
    int* n_out;                  // [b]: clamped keypoint count (nullptr: not written)
    int dst_stride;              // also the upper bound of n
};

// n_grid_blocks < gridDim.x: the blocks behind the frames' project the last frame's map points (last_frame_point, kGridThreads
// each) — TrackWithMotionModel's projection needs nothing of the grid, so it rides the same launch instead of a kernel of its own
// behind it (a tracking frame: one launch and ~10 us less on the chain).
__global__ __launch_bounds__(kGridThreads) void frame_grid_kernel(GridArgs A, LastFrameArgs P, int n_grid_blocks) {
    extern __shared__ int lds[];
    if ((int)blockIdx.x >= n_grid_blocks) {
        last_frame_point(P, ((int)blockIdx.x - n_grid_blocks) * kGridThreads + (int)threadIdx.x);
        return;
    }
    int* const cnt = lds;                    // kNCell + 1
    int* const cur = cnt + (kNCell + 1);     // kNCell
    uint16_t* const cell_of = reinterpret_cast<uint16_t*>(cur + kNCell);  // dst_stride
    uint16_t* const lst = cell_of + A.dst_stride;                        // dst_stride
    __shared__ int wave_tot[kGridThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = A.counts ? A.counts[(size_t)b * A.count_step] : A.n_fixed;
    n = max(0, min(n, A.dst_stride));
    const msorb_keypoint* kps = A.kps + (size_t)b * A.src_stride;
    const float* ur = A.u_right ? A.u_right + (size_t)b * A.ur_stride : nullptr;
    KpLite* kp_out = A.kp + (size_t)b * A.dst_stride;
    uint8_t* occ = A.occ + (size_t)b * A.dst_stride;
    int* cell_begin = A.cell_begin + (size_t)b * (kNCell + 1);
    int* cell_idx = A.cell_idx + (size_t)b * A.dst_stride;
    for (int c = tid; c <= kNCell; c += kGridThreads) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kGridThreads) {
        const msorb_keypoint k = kps[i];
        // PosInGrid, Frame.cc:657-667: round() half away from zero on the float product
        const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, A.minX), A.gridWInv));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, A.minY), A.gridHInv));
        const bool in = !(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows);
        const int cell = px * kGridRows + py;
        cell_of[i] = in ? (uint16_t)cell : (uint16_t)0xFFFF;
        if (in) atomicAdd(&cnt[cell], 1);
        kp_out[i] = KpLite{k.x, k.y, ur ? ur[i] : -1.0f, k.octave};
        occ[i] = 0;
    }
    {  // descriptors: 2 x 16 bytes per keypoint
        const uint4* src = reinterpret_cast<const uint4*>(A.desc + (size_t)b * A.src_stride * 32);
        uint4* dst = reinterpret_cast<uint4*>(A.desc_out + (size_t)b * A.dst_stride * 32);
        for (int i = tid; i < 2 * n; i += kGridThreads) dst[i] = src[i];
    }
    __syncthreads();
    // exclusive scan of the kNCell counters: 3 consecutive cells per thread, wave scan, wave totals
    const int c0 = tid * 3;
    const int v0 = cnt[c0], v1 = cnt[c0 + 1], v2 = cnt[c0 + 2];
    const int mine = v0 + v1 + v2;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += wave_tot[w];
    const int e0 = base + incl - mine;
    __syncthreads();  // every thread has read its counters
    cnt[c0] = e0; cnt[c0 + 1] = e0 + v0; cnt[c0 + 2] = e0 + v0 + v1;
    cur[c0] = e0; cur[c0 + 1] = e0 + v0; cur[c0 + 2] = e0 + v0 + v1;
    if (tid == kGridThreads - 1) cnt[kNCell] = e0 + mine;
    __syncthreads();
    for (int c = tid; c <= kNCell; c += kGridThreads) cell_begin[c] = cnt[c];
    for (int i = tid; i < n; i += kGridThreads) {
        const int c = cell_of[i];
        if (c != 0xFFFF) lst[atomicAdd(&cur[c], 1)] = (uint16_t)i;
    }
    __syncthreads();
    // the reference pushes indices in ascending order: restore it inside every cell.  Short runs (the normal case: about
    // one keypoint per cell): insertion sort by the cell's thread.  Long runs (keypoints piled up in a cell) are noted in a
    // small list and ranked afterwards by a whole wave each (indices are unique) through cell_of, which is free from here on
    // — a pile of k keypoints costs k^2 / 64 steps per lane instead of k^2 on one thread; frames without piles (every real
    // frame) pay one counter read for it.
    constexpr int kShortRun = 32, kMaxLong = 64;
    __shared__ int n_long, long_cell[kMaxLong];
    if (tid == 0) n_long = 0;
    __syncthreads();
    for (int c = tid; c < kNCell; c += kGridThreads) {
        const int s = cnt[c], e = cnt[c + 1];
        if (e - s > kShortRun) {
            const int k = atomicAdd(&n_long, 1);
            if (k < kMaxLong) { long_cell[k] = c; continue; }   // (beyond the list — > 64 piles of > 32 keypoints — the thread sorts it itself)
        }
        for (int i = s + 1; i < e; i++) {
            const uint16_t v = lst[i];
            int j = i - 1;
            while (j >= s && lst[j] > v) { lst[j + 1] = lst[j]; j--; }
            lst[j + 1] = v;
        }
    }
    __syncthreads();
    const int nl = min(n_long, kMaxLong);   // block-uniform
    if (nl > 0) {
        const int wave = tid >> 6, lane = tid & 63;
        for (int k = wave; k < nl; k += kGridThreads / 64) {
            const int c = long_cell[k], s = cnt[c], e = cnt[c + 1];
            for (int i = s + lane; i < e; i += 64) {
                const uint16_t v = lst[i];
                int r = 0;
                for (int j = s; j < e; j++) r += lst[j] < v;
                cell_of[s + r] = v;
            }
        }
        __syncthreads();
        for (int k = wave; k < nl; k += kGridThreads / 64) {
            const int c = long_cell[k], s = cnt[c], e = cnt[c + 1];
            for (int i = s + lane; i < e; i += 64) lst[i] = cell_of[i];
        }
        __syncthreads();
    }
    const int total = cnt[kNCell];
    for (int j = tid; j < total; j += kGridThreads) cell_idx[j] = lst[j];
    if (A.n_out && tid == 0) A.n_out[b] = n;
}

// dynamic LDS of frame_grid_kernel for frames of up to n_cap keypoints (its static __shared__ comes on top: dynamic_lds_room
// subtracts what hipFuncGetAttributes reports for the kernel)
size_t frame_grid_lds(int n_cap) { return (size_t)(2 * kNCell + 1) * sizeof(int) + (size_t)n_cap * 2 * sizeof(uint16_t); }

int launch_frame_grid(const GridArgs& A, int n_frames, hipStream_t s, const LastFrameArgs* proj = nullptr) {
    const int limit = msorb::frame_grid_max_keypoints();
    if (limit < 0) return MSORB_E_HIP;
    if (A.dst_stride > limit) {
        set_last_error("frame grid: more than " + std::to_string(limit) + " keypoints per frame");
        return MSORB_E_CAPACITY;
    }
    const size_t lds = frame_grid_lds(A.dst_stride);   // <= the kernel's room: frame_grid_max_keypoints raised the limit once, for good
    const int n_proj = proj && proj->n > 0 ? (proj->n + kGridThreads - 1) / kGridThreads : 0;
    hipLaunchKernelGGL(frame_grid_kernel, dim3(n_frames + n_proj), dim3(kGridThreads), lds, s, A, proj ? *proj : LastFrameArgs{}, n_frames);
    return MSORB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Tracking::SearchLocalPoints' loop (Tracking.cc:3343-3361: isInFrustum for every local map point the loop reaches) fused
// with the query set-up of ORBmatcher::SearchByProjection (ORBmatcher.cc:52-72): one thread per map point, SoA in, the
// reference's per-point scratch (mbTrackInView, mTrackProjX/Y/XR, mTrackDepth, mnTrackScaleLevel, mTrackViewCos) and the
// window query out.  blockIdx.y = frame of a batch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint8_t kMpVisit = 1, kMpBad = 2, kMpSparsified = 4;

struct LocalPointsArgs {
    const msorb_frustum* frustum;  // [n_frames] (device)
    float cos_limit;
    int m;                          // map points per frame (stride of every array below)
    const float *pos_w, *normal, *max_d, *min_d;  // [frame][3m], [frame][3m], [frame][m], [frame][m]
    const uint8_t* flags;           // kMpVisit | kMpBad | kMpSparsified
    float scale[MSORB_MAX_LEVELS];  // mvScaleFactors
    float th;
    int far_points;
    float th_far;
    // outputs
    WinQuery* q;
    float *proj_x, *proj_y, *proj_xr, *depth, *view_cos;
    int* level;
    uint8_t* in_view;
};

__global__ __launch_bounds__(256) void local_points_kernel(LocalPointsArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.m) return;
    const size_t o = (size_t)blockIdx.y * A.m + i;
    const msorb_frustum F = A.frustum[blockIdx.y];
    const uint8_t fl = A.flags[o];
    FrustumOut r;
    r.in_view = 0; r.px = -1.0f; r.py = -1.0f; r.pxr = 0.0f; r.depth = 0.0f; r.vc = 0.0f; r.level = 0;
    if (fl & kMpVisit)
        r = frustum_point(F, A.cos_limit, A.pos_w[3 * o], A.pos_w[3 * o + 1], A.pos_w[3 * o + 2], A.normal[3 * o],
                          A.normal[3 * o + 1], A.normal[3 * o + 2], A.max_d[o], A.min_d[o]);
    else { r.px = 0.0f; r.py = 0.0f; }  // not visited: the reference leaves the point's scratch alone; zeros here
    A.in_view[o] = r.in_view;
    A.proj_x[o] = r.px; A.proj_y[o] = r.py; A.proj_xr[o] = r.pxr; A.depth[o] = r.depth; A.level[o] = r.level; A.view_cos[o] = r.vc;
    WinQuery w;
    w.x = 0; w.y = 0; w.r = 0; w.ur = 0; w.min_level = 0; w.max_level = 0; w.flags = 0; w.pad[0] = w.pad[1] = w.pad[2] = 0;
    // ORBmatcher.cc:57-64: mbTrackInView, bFarPoints && mTrackDepth > thFarPoints, isBad()
    const bool valid = r.in_view && !(A.far_points && r.depth > A.th_far) && !(fl & kMpBad);
    if (valid) {
        float rad = ((double)r.vc > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:215-221
        if (A.th != 1.0f) rad = __fmul_rn(rad, A.th);       // :68-69
        w.x = r.px; w.y = r.py;
        w.r = __fmul_rn(rad, A.scale[r.level]);             // :71
        w.ur = r.pxr;
        w.min_level = (int16_t)(r.level - 1);
        w.max_level = (int16_t)r.level;
        w.flags = kQValid | ((fl & kMpSparsified) ? 0 : kQSkipOccupied);
    }
    A.q[o] = w;
}


// packed transfer blocks of the last frame's points (n entries)
struct LastLayout {
    size_t o_pos, o_oct, o_desc, o_flags, in_bytes;         // input block
    size_t o_topk, o_u, o_v, o_ur, o_valid, out_bytes;      // output block
    explicit LastLayout(size_t n) {
        o_pos = 0; o_oct = o_pos + 12 * n; o_desc = o_oct + 4 * n; o_flags = o_desc + 32 * n; in_bytes = (o_flags + n + 63) & ~(size_t)63;
        o_topk = 0; o_u = o_topk + sizeof(TopK) * n; o_v = o_u + 4 * n; o_ur = o_v + 4 * n; o_valid = o_ur + 4 * n;
        out_bytes = (o_valid + n + 63) & ~(size_t)63;
    }
};

// packed transfer blocks of one frame's local map points (m entries): one H2D, one D2H
struct MpLayout {
    size_t o_pos, o_nrm, o_max, o_min, o_desc, o_flags, o_frustum, in_bytes;       // input block (the frustum rides at its end: one upload)
    size_t o_topk, o_px, o_py, o_pxr, o_depth, o_level, o_vc, o_inview, out_bytes;  // output block
    explicit MpLayout(size_t m) {
        o_pos = 0; o_nrm = o_pos + 12 * m; o_max = o_nrm + 12 * m; o_min = o_max + 4 * m; o_desc = o_min + 4 * m;
        o_flags = o_desc + 32 * m; o_frustum = (o_flags + m + 63) & ~(size_t)63;
        in_bytes = (o_frustum + sizeof(msorb_frustum) + 63) & ~(size_t)63;
        o_topk = 0; o_px = o_topk + sizeof(TopK) * m; o_py = o_px + 4 * m; o_pxr = o_py + 4 * m; o_depth = o_pxr + 4 * m;
        o_level = o_depth + 4 * m; o_vc = o_level + 4 * m; o_inview = o_vc + 4 * m; out_bytes = (o_inview + m + 63) & ~(size_t)63;
    }
};

}  // namespace

// state the local-points chain keeps on the frame handle
struct msorb_frame_track {
    DBuf<uint8_t> d_in, d_out;
    DBuf<msorb_frustum> d_frustum;
    HBuf<uint8_t> h_in, h_out;
    hipEvent_t ev_in = nullptr;
    // TrackWithMotionModel's search: the last frame's points, resident between calls (msorb_frame_set_last_points)
    DBuf<uint8_t> d_last, d_last_out;
    HBuf<uint8_t> h_last, h_last_out;
    std::vector<float> last_angle;   // host copies the replay reads
    int last_n = -1;                 // -1: no table set
    hipEvent_t ev_last = nullptr;
    void release() {
        d_in.release(); d_out.release(); d_frustum.release(); h_in.release(); h_out.release();
        d_last.release(); d_last_out.release(); h_last.release(); h_last_out.release();
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_last) (void)hipEventDestroy(ev_last);
        ev_in = ev_last = nullptr;
    }
};

namespace msorb {
// Largest keypoint count the grid kernel takes on the current device: its LDS holds 2 x 3073 ints and two uint16 per keypoint
// (160 KB per workgroup on gfx950 -> 32768; the uint16 indices stop at 65535 anyway).  -1: the attribute query failed.
int frame_grid_max_keypoints() {
    const long long dyn = dynamic_lds_room(reinterpret_cast<const void*>(frame_grid_kernel));   // device LDS - the kernel's static part
    if (dyn < 0) {
        set_last_error("frame grid: cannot query / raise the kernel's LDS limit");
        return -1;
    }
    const long long room = dyn - (long long)frame_grid_lds(0);
    return (int)std::max<long long>(0, std::min<long long>(room / (2 * (long long)sizeof(uint16_t)), 32768));
}
// A set that failed half way leaves the handle EMPTY — no keypoints, an empty (host-authoritative) grid — never a mix of the
// new count with old arrays: every search on it finds nothing, msorb_frame_features_in_area returns nothing.
void frame_invalidate(msorb_frame* f) {
    f->N = 0;
    f->kps.clear();
    f->u_right.clear();
    f->cell_begin.assign(kNCell + 1, 0);
    f->cell_idx.clear();
    f->host_grid_valid = true;
}
void frame_track_release(msorb_frame* f) {
    if (f->track) { f->track->release(); delete f->track; f->track = nullptr; }
}
// cell_begin / cell_idx on the host (msorb_frame_features_in_area, msorb_frame_grid) for a frame whose grid was built on the device
int frame_host_grid(msorb_frame* f) {
    std::lock_guard<std::mutex> lk(f->grid_mu);   // two threads may query one frame: the fetch happens once
    if (f->host_grid_valid) return MSORB_OK;
    const int ncell = kGridCols * kGridRows;
    f->cell_begin.assign(ncell + 1, 0);
    f->cell_idx.clear();
    if (f->d_cell_begin.p) {
        HIPCHK(hipMemcpyAsync(f->cell_begin.data(), f->d_cell_begin.p, (size_t)(ncell + 1) * sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        f->cell_idx.assign(f->cell_begin[ncell], 0);
        if (!f->cell_idx.empty()) {
            HIPCHK(hipMemcpyAsync(f->cell_idx.data(), f->d_cell_idx.p, f->cell_idx.size() * sizeof(int), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipStreamSynchronize(f->stream));
        }
    }
    f->host_grid_valid = true;
    return MSORB_OK;
}
}  // namespace msorb

namespace msorb {
// Sets the frame from device arrays: enqueue only, on stream s.  n_cap = upper bound of the keypoint count.
int enqueue_frame_from_device(msorb_frame* f, hipStream_t s, const msorb_keypoint* d_kps, const uint8_t* d_desc,
                              const float* d_u_right, const int* d_count, int n_fixed, int n_cap, float min_x, float max_x,
                              float min_y, float max_y, const float* scale_factors, int nlevels, const LastFrameProjector* proj) {
    int rc;
    n_cap = std::max(n_cap, 1);
    if ((rc = f->d_kp.ensure(n_cap)) || (rc = f->d_desc.ensure((size_t)n_cap * 32)) || (rc = f->d_cell_begin.ensure(kNCell + 1)) ||
        (rc = f->d_cell_idx.ensure(n_cap)) || (rc = f->d_occ.ensure(n_cap)) || (rc = f->d_n.ensure(1)))
        return rc;
    f->nlevels = nlevels;
    f->minX = min_x; f->maxX = max_x; f->minY = min_y; f->maxY = max_y;
    f->gridWInv = static_cast<float>(kGridCols) / (max_x - min_x);  // Frame.cc:147-148
    f->gridHInv = static_cast<float>(kGridRows) / (max_y - min_y);
    f->scale.assign(scale_factors, scale_factors + nlevels);
    GridArgs A{};
    A.kps = d_kps; A.desc = d_desc; A.u_right = d_u_right;
    A.src_stride = 0; A.ur_stride = 0;
    A.counts = d_count; A.count_step = 0; A.n_fixed = n_fixed;
    A.minX = min_x; A.minY = min_y; A.gridWInv = f->gridWInv; A.gridHInv = f->gridHInv;
    A.kp = f->d_kp.p; A.desc_out = f->d_desc.p; A.occ = f->d_occ.p; A.cell_begin = f->d_cell_begin.p; A.cell_idx = f->d_cell_idx.p;
    A.n_out = f->d_n.p;
    A.dst_stride = n_cap;
    f->host_grid_valid = false;
    if (proj) {   // the motion-model projection rides the grid launch (frame_grid_kernel): its arguments need the frame fields set above
        LastFrameArgs P{};
        if ((rc = proj->prepare(proj->ctx, s, &P))) return rc;
        return launch_frame_grid(A, 1, s, &P);
    }
    return launch_frame_grid(A, 1, s);
}
}  // namespace msorb

namespace {

int ensure_track(msorb_frame* f) {
    if (f->track) return MSORB_OK;
    msorb_frame_track* t = new msorb_frame_track();
    if (hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t->ev_last, hipEventDisableTiming) != hipSuccess) {
        t->release();
        delete t;
        set_last_error("frame: cannot create the staging events");
        return MSORB_E_HIP;
    }
    f->track = t;
    return MSORB_OK;
}

// the host copies the accept replay reads (octave / angle of a train keypoint) and the count
void finish_frame_host(msorb_frame* f, const msorb_keypoint* kps, int n, const float* u_right) {
    f->N = n;
    f->kps.assign(kps, kps + n);
    if (u_right) f->u_right.assign(u_right, u_right + n); else f->u_right.assign(n, -1.0f);
}

struct LocalPointsCall {
    const msorb_frustum* frustum;
    float cos_limit;
    int m;
    const float *pos_w, *normal, *max_d, *min_d;
    const uint8_t *visit, *bad, *sparsified, *mp_desc;
    const int* obs;
    float th;
    int far_points;
    float th_far, nnratio;
};

int check_local_points(const msorb_frame* f, const LocalPointsCall& c) {
    if (!f || !c.frustum || c.m < 0 || c.frustum->n_scale_levels < 1 || c.frustum->n_scale_levels > MSORB_MAX_LEVELS ||
        (c.m > 0 && (!c.pos_w || !c.normal || !c.max_d || !c.min_d || !c.bad || !c.sparsified || !c.mp_desc || !c.obs)))
        return MSORB_E_INVALID;
    return MSORB_OK;
}

// stage + upload the map points of one call (on f->stream, completion in f->track->ev_in)
int upload_local_points(msorb_frame* f, const LocalPointsCall& c) {
    int rc;
    if ((rc = ensure_track(f))) return rc;
    msorb_frame_track& T = *f->track;
    const size_t m = (size_t)c.m;
    const MpLayout L(m);
    if ((rc = T.d_in.ensure(L.in_bytes)) || (rc = T.d_out.ensure(L.out_bytes)) || (rc = T.h_in.ensure(L.in_bytes)) ||
        (rc = T.h_out.ensure(L.out_bytes)) || (rc = T.d_frustum.ensure(1)) || (rc = f->d_q.ensure(m)) || (rc = f->h_topk.ensure(m)))
        return rc;
    uint8_t* h = T.h_in.p;
    std::memcpy(h + L.o_pos, c.pos_w, 12 * m);
    std::memcpy(h + L.o_nrm, c.normal, 12 * m);
    std::memcpy(h + L.o_max, c.max_d, 4 * m);
    std::memcpy(h + L.o_min, c.min_d, 4 * m);
    std::memcpy(h + L.o_desc, c.mp_desc, 32 * m);
    for (size_t i = 0; i < m; i++)
        h[L.o_flags + i] = (uint8_t)(((!c.visit || c.visit[i]) ? kMpVisit : 0) | (c.bad[i] ? kMpBad : 0) | (c.sparsified[i] ? kMpSparsified : 0));
    std::memcpy(h + L.o_frustum, c.frustum, sizeof(msorb_frustum));
    hipStream_t s = f->stream;
    HIPCHK(small_copy(T.d_in.p, h, L.in_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(T.ev_in, s));
    return MSORB_OK;
}

// lanes per query of the window kernel for SearchLocalPoints' radii: 4 * th * scale of a middle level (RadiusByViewingCos)
int local_points_lanes(const msorb_frame* f, float th) {
    const float mid = f->scale.empty() ? 1.0f : f->scale[f->scale.size() / 2];
    return window_lanes_for(4.0f * th * mid, f->gridWInv, f->gridHInv);
}

// frustum + queries + round 0 of the window search + the read-back, enqueued on stream s (which must already be ordered
// behind the upload and behind whatever produced the frame's device arrays).  d_occ must hold the occupancy snapshot.
int enqueue_local_points(msorb_frame* f, const LocalPointsCall& c, hipStream_t s) {
    msorb_frame_track& T = *f->track;
    const size_t m = (size_t)c.m;
    if (!m) return MSORB_OK;
    const MpLayout L(m);
    LocalPointsArgs A{};
    A.cos_limit = c.cos_limit;
    A.m = c.m;
    uint8_t* di = T.d_in.p;
    A.frustum = reinterpret_cast<const msorb_frustum*>(di + L.o_frustum);
    uint8_t* dout = T.d_out.p;
    A.pos_w = reinterpret_cast<const float*>(di + L.o_pos); A.normal = reinterpret_cast<const float*>(di + L.o_nrm);
    A.max_d = reinterpret_cast<const float*>(di + L.o_max); A.min_d = reinterpret_cast<const float*>(di + L.o_min);
    A.flags = di + L.o_flags;
    for (int l = 0; l < MSORB_MAX_LEVELS; l++) A.scale[l] = l < (int)f->scale.size() ? f->scale[l] : 0.0f;
    A.th = c.th; A.far_points = c.far_points; A.th_far = c.th_far;
    A.q = f->d_q.p;
    A.proj_x = reinterpret_cast<float*>(dout + L.o_px); A.proj_y = reinterpret_cast<float*>(dout + L.o_py);
    A.proj_xr = reinterpret_cast<float*>(dout + L.o_pxr); A.depth = reinterpret_cast<float*>(dout + L.o_depth);
    A.level = reinterpret_cast<int*>(dout + L.o_level); A.view_cos = reinterpret_cast<float*>(dout + L.o_vc);
    A.in_view = dout + L.o_inview;
    hipLaunchKernelGGL(local_points_kernel, dim3((unsigned)((m + 255) / 256), 1), dim3(256), 0, s, A);
    launch_window_topk(f->view(), f->d_q.p, di + L.o_desc, 0, c.m, reinterpret_cast<TopK*>(dout + L.o_topk), s, 1, 0, 0, nullptr,
                       local_points_lanes(f, c.th));
    HIPCHK(small_copy(T.h_out.p, dout, L.out_bytes, hipMemcpyDeviceToHost, s));
    return MSORB_OK;
}

struct LocalPointsOut {
    uint8_t* track_in_view;
    float *proj_x, *proj_y, *proj_xr, *track_depth;
    int* scale_level;
    float* view_cos;
};

// after the synchronisation: scatter the per-point scratch, replay the claims in map-point order
int replay_local_points(msorb_frame* f, const LocalPointsCall& c, std::vector<uint8_t>& occ, int* frame_mp, const LocalPointsOut& o,
                        int* nmatches, int* rounds) {
    *nmatches = 0;
    if (rounds) *rounds = 0;
    const size_t m = (size_t)c.m;
    if (!m) return MSORB_OK;
    msorb_frame_track& T = *f->track;
    const MpLayout L(m);
    const uint8_t* ho = T.h_out.p;
    const uint8_t* in_view = ho + L.o_inview;
    const float* depth = reinterpret_cast<const float*>(ho + L.o_depth);
    if (o.track_in_view) std::memcpy(o.track_in_view, in_view, m);
    if (o.proj_x) std::memcpy(o.proj_x, ho + L.o_px, 4 * m);
    if (o.proj_y) std::memcpy(o.proj_y, ho + L.o_py, 4 * m);
    if (o.proj_xr) std::memcpy(o.proj_xr, ho + L.o_pxr, 4 * m);
    if (o.track_depth) std::memcpy(o.track_depth, depth, 4 * m);
    if (o.scale_level) std::memcpy(o.scale_level, ho + L.o_level, 4 * m);
    if (o.view_cos) std::memcpy(o.view_cos, ho + L.o_vc, 4 * m);
    std::memcpy(f->h_topk.p, ho + L.o_topk, sizeof(TopK) * m);
    std::vector<uint8_t> flags(m);
    for (size_t i = 0; i < m; i++) {
        const bool valid = in_view[i] && !(c.far_points && depth[i] > c.th_far) && !c.bad[i];
        flags[i] = valid ? (uint8_t)(kQValid | (c.sparsified[i] ? 0 : kQSkipOccupied)) : 0;
    }
    int nm = 0;
    const float nnratio = c.nnratio;
    auto accept = [&](int qi, const int* idx, const int* dist, int n, int* new_occ) -> int {
        if (n == 0) return -1;
        const int bestDist = dist[0], bestIdx = idx[0];
        const int bestLevel = f->kps[bestIdx].octave;
        const int bestDist2 = n > 1 ? dist[1] : 256;
        const int bestLevel2 = n > 1 ? f->kps[idx[1]].octave : -1;
        if (bestDist <= kThHigh) {  // ORBmatcher.cc:122-141
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) return -1;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                frame_mp[bestIdx] = qi;
                nm++;
                *new_occ = c.obs[qi] > 0;
                return bestIdx;
            }
        }
        return -1;
    };
    // resync rounds (rare) read the query descriptors from f->d_qdesc: alias the uploaded block
    const int rc = run_window_search(f, c.m, nullptr, flags.data(), nullptr, occ, 2, accept, true, rounds, T.d_in.p + L.o_desc,
                                     local_points_lanes(f, c.th));
    *nmatches = nm;
    return rc;
}

int initial_occupancy(const msorb_frame* f, int n, const int* frame_mp, const int* obs, int m, std::vector<uint8_t>& occ, bool* any) {
    occ.assign(n, 0);
    *any = false;
    if (!frame_mp) return MSORB_OK;
    for (int i = 0; i < n; i++) {
        if (frame_mp[i] >= m) { set_last_error("frame_mp holds an id outside the map-point table"); return MSORB_E_INVALID; }
        occ[i] = frame_mp[i] >= 0 && obs[frame_mp[i]] > 0;
        *any |= occ[i] != 0;
    }
    return MSORB_OK;
}

// ---- TrackWithMotionModel's search on the resident last-frame table ----
struct LastFrameCall {
    const msorb_motion_model* mm;
    const int* obs;
    int n_obs;
    float th;
    int check_orientation;
};

// projection + queries + round 0 of the window search + the read-back, enqueued on stream s (ordered behind whatever produced
// the frame's device arrays; the table's upload is waited for here).  d_occ must hold the occupancy snapshot.
// buffers, the wait for the table's upload, the projection kernel's arguments
int prepare_last_frame(msorb_frame* f, const LastFrameCall& c, hipStream_t s, LastFrameArgs& A) {
    msorb_frame_track& T = *f->track;
    const size_t n = (size_t)T.last_n;
    A = LastFrameArgs{};
    if (!n) return MSORB_OK;
    const LastLayout L(n);
    int rc;
    if ((rc = f->d_q.ensure(n)) || (rc = T.d_last_out.ensure(L.out_bytes)) || (rc = T.h_last_out.ensure(L.out_bytes)) ||
        (rc = f->h_topk.ensure(n)))
        return rc;
    HIPCHK(hipStreamWaitEvent(s, T.ev_last, 0));
    const msorb_motion_model& m = *c.mm;
    A.qx = m.q[0]; A.qy = m.q[1]; A.qz = m.q[2]; A.qw = m.q[3];
    A.tx = m.t[0]; A.ty = m.t[1]; A.tz = m.t[2];
    A.fx = m.fx; A.fy = m.fy; A.cx = m.cx; A.cy = m.cy; A.mbf = m.mbf;
    A.min_x = f->minX; A.max_x = f->maxX; A.min_y = f->minY; A.max_y = f->maxY;
    A.th = c.th; A.forward = m.forward; A.backward = m.backward; A.n = T.last_n;
    for (int l = 0; l < MSORB_MAX_LEVELS; l++) A.scale[l] = l < (int)f->scale.size() ? f->scale[l] : 0.0f;
    const uint8_t* di = T.d_last.p;
    uint8_t* dout = T.d_last_out.p;
    A.pos_w = reinterpret_cast<const float*>(di + L.o_pos); A.octave = reinterpret_cast<const int*>(di + L.o_oct);
    A.flags = di + L.o_flags;
    A.q = f->d_q.p;
    A.u = reinterpret_cast<float*>(dout + L.o_u); A.v = reinterpret_cast<float*>(dout + L.o_v);
    A.ur = reinterpret_cast<float*>(dout + L.o_ur); A.valid = dout + L.o_valid;
    return MSORB_OK;
}
// projection + queries + round 0 of the window search + the read-back, enqueued on stream s (ordered behind whatever produced
// the frame's device arrays; the table's upload is waited for here).  d_occ must hold the occupancy snapshot.
// projected: the projection already ran with the frame's grid launch (prepare_last_frame through LastFrameProjector)
int enqueue_last_frame(msorb_frame* f, const LastFrameCall& c, hipStream_t s, bool projected = false) {
    msorb_frame_track& T = *f->track;
    const size_t n = (size_t)T.last_n;
    if (!n) return MSORB_OK;
    const LastLayout L(n);
    if (!projected) {
        LastFrameArgs A{};
        int rc;
        if ((rc = prepare_last_frame(f, c, s, A))) return rc;
        hipLaunchKernelGGL(last_frame_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A);
    }
    const uint8_t* di = T.d_last.p;
    uint8_t* dout = T.d_last_out.p;
    const float mid = f->scale.empty() ? 1.0f : f->scale[f->scale.size() / 2];
    launch_window_topk(f->view(), f->d_q.p, di + L.o_desc, 0, T.last_n, reinterpret_cast<TopK*>(dout + L.o_topk), s, 1, 0, 0, nullptr,
                       window_lanes_for(c.th * mid, f->gridWInv, f->gridHInv));
    HIPCHK(small_copy(T.h_last_out.p, dout, L.out_bytes, hipMemcpyDeviceToHost, s));
    return MSORB_OK;
}

// after the synchronisation: the sequential claims in last-frame order (:2035-2038) and the rotation histogram (:2040-2057,
// :2129-2149); occ = the occupancy round 0 ran against
int replay_last_frame(msorb_frame* f, const LastFrameCall& c, std::vector<uint8_t>& occ, int* cur_mp, int* nmatches, uint8_t* proj_valid,
                      float* proj_u, float* proj_v, float* proj_ur) {
    *nmatches = 0;
    msorb_frame_track& T = *f->track;
    const size_t n = (size_t)T.last_n;
    if (!n) return MSORB_OK;
    const LastLayout L(n);
    const uint8_t* ho = T.h_last_out.p;
    const uint8_t* valid = ho + L.o_valid;
    if (proj_valid) std::memcpy(proj_valid, valid, n);
    if (proj_u) std::memcpy(proj_u, ho + L.o_u, 4 * n);
    if (proj_v) std::memcpy(proj_v, ho + L.o_v, 4 * n);
    if (proj_ur) std::memcpy(proj_ur, ho + L.o_ur, 4 * n);
    std::memcpy(f->h_topk.p, ho + L.o_topk, sizeof(TopK) * n);
    std::vector<uint8_t> flags(n);
    for (size_t i = 0; i < n; i++) flags[i] = valid[i] ? (uint8_t)(kQValid | kQSkipOccupied) : 0;
    int nm = 0;
    std::vector<int> rotHist[kHistoLength];
    const float factor = 1.0f / kHistoLength;
    const int* obs = c.obs;
    auto accept = [&](int qi, const int* idx, const int* dist, int nc, int* new_occ) -> int {
        if (nc == 0 || dist[0] > kThHigh) return -1;   // :2035
        const int bestIdx2 = idx[0];
        cur_mp[bestIdx2] = qi;
        nm++;
        if (c.check_orientation) {
            float rot = T.last_angle[qi] - f->kps[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == kHistoLength) bin = 0;
            if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(bestIdx2);
        }
        *new_occ = obs[qi] > 0;
        return bestIdx2;
    };
    const float mid = f->scale.empty() ? 1.0f : f->scale[f->scale.size() / 2];
    const int rc = run_window_search(f, T.last_n, nullptr, flags.data(), nullptr, occ, 1, accept, true, nullptr, T.d_last.p + L.o_desc,
                                     window_lanes_for(c.th * mid, f->gridWInv, f->gridHInv));
    if (rc) return rc;
    if (c.check_orientation) {
        int sizes[kHistoLength], ind[3];
        for (int i = 0; i < kHistoLength; i++) sizes[i] = (int)rotHist[i].size();
        msorb_three_maxima(sizes, kHistoLength, ind);
        for (int i = 0; i < kHistoLength; i++)
            if (i != ind[0] && i != ind[1] && i != ind[2])
                for (int k : rotHist[i]) { cur_mp[k] = -1; nm--; }
    }
    *nmatches = nm;
    return MSORB_OK;
}

int check_last_frame(const msorb_frame* f, const LastFrameCall& c, bool frame_is_set) {
    if (!f || !c.mm || !f->track || f->track->last_n < 0) {
        if (f && (!f->track || f->track->last_n < 0)) set_last_error("no last-frame table on this frame (msorb_frame_set_last_points)");
        return MSORB_E_INVALID;
    }
    if (f->track->last_n > 0 && (!c.obs || c.n_obs < f->track->last_n)) {
        set_last_error("observation table shorter than the last-frame table");
        return MSORB_E_INVALID;
    }
    if (frame_is_set && f->nlevels < 1) { set_last_error("frame not set"); return MSORB_E_INVALID; }
    return MSORB_OK;
}

}  // namespace

extern "C" {

int msorb_frame_set_last_points(msorb_frame* f, int n, const uint8_t* has_point, const float* pos_w, const int* octave,
                                const float* angle, const uint8_t* mp_desc) {
    if (!f || n < 0 || (n > 0 && (!has_point || !pos_w || !octave || !angle || !mp_desc))) return MSORB_E_INVALID;
    for (int i = 0; i < n; i++)
        if (has_point[i] && (octave[i] < 0 || octave[i] >= MSORB_MAX_LEVELS)) { set_last_error("last-frame octave out of range"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(f->device));
    int rc;
    if ((rc = ensure_track(f))) return rc;
    msorb_frame_track& T = *f->track;
    T.last_n = -1;   // stays unset if anything below fails
    const LastLayout L((size_t)n);
    if ((rc = T.d_last.ensure(L.in_bytes)) || (rc = T.h_last.ensure(L.in_bytes))) return rc;
    // a previous upload may still read the staging block (the call that enqueued it has not necessarily synchronised)
    HIPCHK(hipEventSynchronize(T.ev_last));
    uint8_t* h = T.h_last.p;
    const size_t m = (size_t)n;
    if (m) {
        std::memcpy(h + L.o_pos, pos_w, 12 * m);
        std::memcpy(h + L.o_oct, octave, 4 * m);
        std::memcpy(h + L.o_desc, mp_desc, 32 * m);
        for (size_t i = 0; i < m; i++) h[L.o_flags + i] = has_point[i] ? 1 : 0;
        HIPCHK(small_copy(T.d_last.p, h, L.in_bytes, hipMemcpyHostToDevice, f->stream));
    }
    HIPCHK(hipEventRecord(T.ev_last, f->stream));
    T.last_angle.assign(angle, angle + n);
    T.last_n = n;
    return MSORB_OK;
}

int msorb_frame_last_points_count(const msorb_frame* f) { return f && f->track ? f->track->last_n : -1; }

int msorb_search_last_frame(msorb_frame* f, const msorb_motion_model* mm, const int* obs, int n_obs, int* cur_mp, float th,
                            int check_orientation, int* nmatches, uint8_t* proj_valid, float* proj_u, float* proj_v, float* proj_ur) {
    const LastFrameCall c{mm, obs, n_obs, th, check_orientation};
    if (!nmatches || n_obs < 0) return MSORB_E_INVALID;
    *nmatches = 0;
    int rc = check_last_frame(f, c, true);
    if (rc) return rc;
    if (f->N > 0 && !cur_mp) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    std::vector<uint8_t> occ(f->N, 0);
    for (int i = 0; i < f->N; i++) {   // :2011-2013: a keypoint whose map point has observations is never taken
        if (cur_mp[i] >= n_obs) { set_last_error("cur_mp holds an id outside the observation table"); return MSORB_E_INVALID; }
        occ[i] = cur_mp[i] >= 0 && obs[cur_mp[i]] > 0;
    }
    hipStream_t s = f->stream;
    if (f->N) {
        if ((rc = f->h_in.ensure((size_t)f->N + 64))) return rc;
        std::memcpy(f->h_in.p, occ.data(), f->N);
        HIPCHK(small_copy(f->d_occ.p, f->h_in.p, f->N, hipMemcpyHostToDevice, s));
    }
    if ((rc = enqueue_last_frame(f, c, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return replay_last_frame(f, c, occ, cur_mp, nmatches, proj_valid, proj_u, proj_v, proj_ur);
}

int msorb_frame_set_device(msorb_frame* f, const msorb_keypoint* d_keypoints, int n, const uint8_t* d_descriptors,
                           const float* d_u_right, float min_x, float max_x, float min_y, float max_y,
                           const float* scale_factors, int nlevels) {
    if (!f || n < 0 || (n > 0 && (!d_keypoints || !d_descriptors)) || !scale_factors || nlevels < 1 || nlevels > MSORB_MAX_LEVELS ||
        !(max_x > min_x) || !(max_y > min_y))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    int rc;
    if ((rc = enqueue_frame_from_device(f, f->stream, d_keypoints, d_descriptors, d_u_right, nullptr, n, n, min_x, max_x, min_y,
                                        max_y, scale_factors, nlevels)))
        return rc;
    // the replay's host copies: keypoints (octave, angle) and mvuRight
    std::vector<msorb_keypoint> kps(n);
    std::vector<float> ur(n);
    if (n) {
        HIPCHK(hipMemcpyAsync(kps.data(), d_keypoints, (size_t)n * sizeof(msorb_keypoint), hipMemcpyDeviceToHost, f->stream));
        if (d_u_right) HIPCHK(hipMemcpyAsync(ur.data(), d_u_right, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    }
    HIPCHK(hipStreamSynchronize(f->stream));
    HIPCHK(hipGetLastError());
    finish_frame_host(f, kps.data(), n, d_u_right ? ur.data() : nullptr);
    return MSORB_OK;
}

int msorb_frame_grid(msorb_frame* f, int* cell_begin, int* cell_idx, int capacity, int* n_assigned) {
    if (!f || !cell_begin || !n_assigned) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    int rc;
    if ((rc = frame_host_grid(f))) return rc;
    std::memcpy(cell_begin, f->cell_begin.data(), (size_t)(kNCell + 1) * sizeof(int));
    *n_assigned = f->cell_begin[kNCell];
    if (*n_assigned > capacity || (*n_assigned > 0 && !cell_idx)) return MSORB_E_CAPACITY;
    if (*n_assigned) std::memcpy(cell_idx, f->cell_idx.data(), (size_t)*n_assigned * sizeof(int));
    return MSORB_OK;
}

namespace {
struct FrameSinkCtx {
    msorb_frame* f;
    float min_x, max_x, min_y, max_y;
    const float* scale;
    int nlevels;
    // the chained search (msorb_track_frontend); nullptr: the frame only
    const LocalPointsCall* lp;
    const uint8_t* occ0;  // initial occupancy to upload (nullptr: all free, cleared by the grid kernel)
    const LastFrameCall* lf;  // the chained motion-model search (msorb_track_frontend_motion); nullptr: none
};
int frame_sink(void* ctx, const StereoDeviceOutputs& o) {
    FrameSinkCtx& C = *static_cast<FrameSinkCtx*>(ctx);
    const LastFrameProjector proj{[](void* ctx, hipStream_t s, void* args) {
                                      FrameSinkCtx& X = *static_cast<FrameSinkCtx*>(ctx);
                                      return prepare_last_frame(X.f, *X.lf, s, *static_cast<LastFrameArgs*>(args));
                                  },
                                  ctx};
    int rc = enqueue_frame_from_device(C.f, o.stream, o.kps_left, o.desc_left, o.u_right, o.n_left, 0, o.capacity, C.min_x, C.max_x,
                                       C.min_y, C.max_y, C.scale, C.nlevels, C.lf ? &proj : nullptr);
    if (rc) return rc;
    if (C.lf) return enqueue_last_frame(C.f, *C.lf, o.stream, /*projected=*/true);
    if (!C.lp) return rc;
    HIPCHK(hipStreamWaitEvent(o.stream, C.f->track->ev_in, 0));
    return enqueue_local_points(C.f, *C.lp, o.stream);
}
}  // namespace

int msorb_extract_stereo_frame(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                               size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left,
                               uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right,
                               int capacity, float* u_right, float* depth, int* n_oob, float min_x, float max_x, float min_y,
                               float max_y) {
    if (!h || !f || !(max_x > min_x) || !(max_y > min_y)) return MSORB_E_INVALID;
    if (extractor_device(h) != f->device) { set_last_error("extractor and frame live on different devices"); return MSORB_E_INVALID; }
    float scale[MSORB_MAX_LEVELS];
    int rc = msorb_extractor_tables(h, scale, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    FrameSinkCtx C{f, min_x, max_x, min_y, max_y, scale, extractor_levels(h), nullptr, nullptr, nullptr};
    rc = extract_stereo_sink(h, left, right, rows, cols, stride_left, stride_right, mb, mbf, kps_left, desc_left, n_left, kps_right,
                             desc_right, n_right, capacity, u_right, depth, n_oob, frame_sink, &C);
    if (rc) { frame_invalidate(f); return rc; }
    finish_frame_host(f, kps_left, *n_left, u_right);
    return MSORB_OK;
}

int msorb_search_local_points(msorb_frame* f, const msorb_frustum* frustum, float viewing_cos_limit, int m, const float* pos_w,
                              const float* normal, const float* max_distance, const float* min_distance, const uint8_t* visit,
                              const uint8_t* bad, const uint8_t* sparsified, const uint8_t* mp_desc, const int* obs, int* frame_mp,
                              float th, int far_points, float th_far_points, float nnratio, uint8_t* track_in_view, float* proj_x,
                              float* proj_y, float* proj_xr, float* track_depth, int* scale_level, float* view_cos, int* nmatches) {
    const LocalPointsCall c{frustum, viewing_cos_limit, m, pos_w, normal, max_distance, min_distance, visit, bad, sparsified, mp_desc,
                            obs, th, far_points, th_far_points, nnratio};
    int rc = check_local_points(f, c);
    if (rc || !nmatches || (f->N > 0 && !frame_mp)) return MSORB_E_INVALID;
    *nmatches = 0;
    if (frustum->n_scale_levels > f->nlevels) { set_last_error("frustum has more scale levels than the frame"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(f->device));
    std::vector<uint8_t> occ;
    bool any = false;
    if ((rc = initial_occupancy(f, f->N, frame_mp, obs, m, occ, &any))) return rc;
    if ((rc = upload_local_points(f, c))) return rc;
    hipStream_t s = f->stream;
    if (f->N) {  // the occupancy snapshot of round 0
        if ((rc = f->h_in.ensure((size_t)f->N + 64))) return rc;
        std::memcpy(f->h_in.p, occ.data(), f->N);
        HIPCHK(small_copy(f->d_occ.p, f->h_in.p, f->N, hipMemcpyHostToDevice, s));
    }
    if ((rc = enqueue_local_points(f, c, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const LocalPointsOut o{track_in_view, proj_x, proj_y, proj_xr, track_depth, scale_level, view_cos};
    return replay_local_points(f, c, occ, frame_mp, o, nmatches, nullptr);
}

int msorb_track_frontend(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                         size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left,
                         int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right,
                         float* depth, int* n_oob, float min_x, float max_x, float min_y, float max_y, const msorb_frustum* frustum,
                         float viewing_cos_limit, int m, const float* pos_w, const float* normal, const float* max_distance,
                         const float* min_distance, const uint8_t* visit, const uint8_t* bad, const uint8_t* sparsified,
                         const uint8_t* mp_desc, const int* obs, int* frame_mp, float th, int far_points, float th_far_points,
                         float nnratio, uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr, float* track_depth,
                         int* scale_level, float* view_cos, int* nmatches, int* rounds) {
    const LocalPointsCall c{frustum, viewing_cos_limit, m, pos_w, normal, max_distance, min_distance, visit, bad, sparsified, mp_desc,
                            obs, th, far_points, th_far_points, nnratio};
    if (!h || !f || !(max_x > min_x) || !(max_y > min_y) || !nmatches || !frame_mp) return MSORB_E_INVALID;
    int rc = check_local_points(f, c);
    if (rc) return rc;
    *nmatches = 0;
    if (extractor_device(h) != f->device) { set_last_error("extractor and frame live on different devices"); return MSORB_E_INVALID; }
    if (frustum->n_scale_levels > extractor_levels(h)) { set_last_error("frustum has more scale levels than the extractor"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(f->device));
    float scale[MSORB_MAX_LEVELS];
    if ((rc = msorb_extractor_tables(h, scale, nullptr, nullptr, nullptr, nullptr))) return rc;
    f->scale.assign(scale, scale + extractor_levels(h));  // the query set-up reads it before the sink has run
    if ((rc = upload_local_points(f, c))) return rc;     // rides PCIe while the extraction kernels run
    FrameSinkCtx C{f, min_x, max_x, min_y, max_y, scale, extractor_levels(h), &c, nullptr, nullptr};
    rc = extract_stereo_sink(h, left, right, rows, cols, stride_left, stride_right, mb, mbf, kps_left, desc_left, n_left, kps_right,
                             desc_right, n_right, capacity, u_right, depth, n_oob, frame_sink, &C);
    if (rc) { frame_invalidate(f); return rc; }
    finish_frame_host(f, kps_left, *n_left, u_right);
    // a new frame holds no map points (Frame.cc:139: mvpMapPoints = vector<MapPoint*>(N, nullptr)): round 0 ran against an
    // all-free occupancy, which the grid kernel cleared on the device
    for (int i = 0; i < f->N; i++) frame_mp[i] = -1;
    std::vector<uint8_t> occ(f->N, 0);
    const LocalPointsOut o{track_in_view, proj_x, proj_y, proj_xr, track_depth, scale_level, view_cos};
    return replay_local_points(f, c, occ, frame_mp, o, nmatches, rounds);
}

int msorb_track_frontend_motion(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                                size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left,
                                uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right,
                                int capacity, float* u_right, float* depth, int* n_oob, float min_x, float max_x, float min_y,
                                float max_y, const msorb_motion_model* mm, const int* obs, int* cur_mp, float th,
                                int check_orientation, int* nmatches) {
    if (!h || !f || !(max_x > min_x) || !(max_y > min_y) || !nmatches || !cur_mp) return MSORB_E_INVALID;
    *nmatches = 0;
    const LastFrameCall c{mm, obs, f->track ? f->track->last_n : 0, th, check_orientation};
    int rc = check_last_frame(f, c, false);
    if (rc) return rc;
    if (extractor_device(h) != f->device) { set_last_error("extractor and frame live on different devices"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(f->device));
    float scale[MSORB_MAX_LEVELS];
    if ((rc = msorb_extractor_tables(h, scale, nullptr, nullptr, nullptr, nullptr))) return rc;
    FrameSinkCtx C{f, min_x, max_x, min_y, max_y, scale, extractor_levels(h), nullptr, nullptr, &c};
    rc = extract_stereo_sink(h, left, right, rows, cols, stride_left, stride_right, mb, mbf, kps_left, desc_left, n_left, kps_right,
                             desc_right, n_right, capacity, u_right, depth, n_oob, frame_sink, &C);
    if (rc) { frame_invalidate(f); return rc; }
    finish_frame_host(f, kps_left, *n_left, u_right);
    // a new frame holds no map points: round 0 ran against the all-free occupancy the grid kernel left on the device
    for (int i = 0; i < f->N; i++) cur_mp[i] = -1;
    std::vector<uint8_t> occ(f->N, 0);
    return replay_last_frame(f, c, occ, cur_mp, nmatches, nullptr, nullptr, nullptr, nullptr);
}

// Batched, device-resident form of the same chain (offline throughput, measurement): every frame's features are the
// DEVICE outputs of msorb_extract_batch (+ msorb_stereo_matches_batch), every frame brings its own pose and m map points.
// Grid, frustum + queries and the window search of all frames are three launches; the per-query candidate lists stay on
// the device (a new frame holds no map points, so the lists are final up to the sequential claim replay).
int msorb_track_batch(int device, int n_frames, const msorb_keypoint* d_keypoints, const uint8_t* d_descriptors,
                      const float* d_u_right, const int* d_counts, int frame_step, int capacity, float min_x, float max_x,
                      float min_y, float max_y, const float* scale_factors, int nlevels, const msorb_frustum* frusta,
                      float viewing_cos_limit, int m, const float* d_pos_w, const float* d_normal, const float* d_max_distance,
                      const float* d_min_distance, const uint8_t* d_flags, const uint8_t* d_mp_desc, float th, int far_points,
                      float th_far_points, int* d_topk, uint8_t* d_track_in_view, int* d_cell_begin, int* d_cell_idx,
                      float* elapsed_ms, unsigned long long* n_pairs) {
    if (elapsed_ms) elapsed_ms[0] = elapsed_ms[1] = elapsed_ms[2] = 0;
    if (n_pairs) *n_pairs = 0;
    if (n_frames < 0 || m < 0 || capacity < 1 || frame_step < 1 || !scale_factors || nlevels < 1 || nlevels > MSORB_MAX_LEVELS ||
        !(max_x > min_x) || !(max_y > min_y) ||
        (n_frames > 0 && (!d_keypoints || !d_descriptors || !d_counts || !frusta)) ||
        (n_frames > 0 && m > 0 && (!d_pos_w || !d_normal || !d_max_distance || !d_min_distance || !d_flags || !d_mp_desc || !d_topk)))
        return MSORB_E_INVALID;
    if (n_frames == 0) return MSORB_OK;
    for (int b = 0; b < n_frames; b++)
        if (frusta[b].n_scale_levels < 1 || frusta[b].n_scale_levels > nlevels) { set_last_error("frustum scale levels out of range"); return MSORB_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        hipEvent_t ev[4] = {};
        DBuf<KpLite> kp;
        DBuf<uint8_t> desc, occ, in_view;
        DBuf<int> cell_begin, cell_idx, level;
        DBuf<WinQuery> q;
        DBuf<float> proj;
        DBuf<msorb_frustum> fr;
        DBuf<unsigned long long> cnt;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            kp.release(); desc.release(); occ.release(); in_view.release(); cell_begin.release(); cell_idx.release(); level.release();
            q.release(); proj.release(); fr.release(); cnt.release();
            for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            if (s) (void)hipStreamDestroy(s);
            s = nullptr; device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch S;
    if (S.device != device) {
        S.release();
        S.device = device;   // release() frees on this device; a failure below releases again and leaves device == -1
        hipError_t e = hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking);
        for (auto& ev : S.ev)
            if (e == hipSuccess) e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            S.release();
            set_last_error(std::string("msorb_track_batch: stream / event creation: ") + hipGetErrorString(e));
            return MSORB_E_HIP;
        }
    }
    const size_t B = (size_t)n_frames, cap = (size_t)capacity, M = (size_t)m;
    int rc;
    if ((rc = S.kp.ensure(B * cap)) || (rc = S.desc.ensure(B * cap * 32)) || (rc = S.occ.ensure(B * cap)) ||
        (rc = S.cell_begin.ensure(B * (kNCell + 1))) || (rc = S.cell_idx.ensure(B * cap)) || (rc = S.fr.ensure(B)) ||
        (rc = S.q.ensure(B * M)) || (rc = S.proj.ensure(5 * B * M)) || (rc = S.level.ensure(B * M)) ||
        (rc = S.in_view.ensure(B * M)) || (rc = S.cnt.ensure(1)))
        return rc;
    hipStream_t s = S.s;
    HIPCHK(hipMemcpyAsync(S.fr.p, frusta, B * sizeof(msorb_frustum), hipMemcpyHostToDevice, s));
    if (n_pairs) HIPCHK(hipMemsetAsync(S.cnt.p, 0, sizeof(unsigned long long), s));
    GridArgs G{};
    G.kps = d_keypoints; G.desc = d_descriptors; G.u_right = d_u_right;
    G.src_stride = (size_t)frame_step * cap; G.ur_stride = cap;
    G.counts = d_counts; G.count_step = frame_step; G.n_fixed = 0;
    G.minX = min_x; G.minY = min_y;
    G.gridWInv = static_cast<float>(kGridCols) / (max_x - min_x);
    G.gridHInv = static_cast<float>(kGridRows) / (max_y - min_y);
    G.kp = S.kp.p; G.desc_out = S.desc.p; G.occ = S.occ.p;
    G.cell_begin = d_cell_begin ? d_cell_begin : S.cell_begin.p;
    G.cell_idx = d_cell_idx ? d_cell_idx : S.cell_idx.p;
    G.n_out = nullptr; G.dst_stride = capacity;
    HIPCHK(hipEventRecord(S.ev[0], s));
    if ((rc = launch_frame_grid(G, n_frames, s))) return rc;
    HIPCHK(hipEventRecord(S.ev[1], s));
    if (m > 0) {
        LocalPointsArgs A{};
        A.frustum = S.fr.p; A.cos_limit = viewing_cos_limit; A.m = m;
        A.pos_w = d_pos_w; A.normal = d_normal; A.max_d = d_max_distance; A.min_d = d_min_distance; A.flags = d_flags;
        for (int l = 0; l < MSORB_MAX_LEVELS; l++) A.scale[l] = l < nlevels ? scale_factors[l] : 0.0f;
        A.th = th; A.far_points = far_points; A.th_far = th_far_points;
        A.q = S.q.p;
        A.proj_x = S.proj.p; A.proj_y = S.proj.p + B * M; A.proj_xr = S.proj.p + 2 * B * M; A.depth = S.proj.p + 3 * B * M;
        A.view_cos = S.proj.p + 4 * B * M; A.level = S.level.p;
        A.in_view = d_track_in_view ? d_track_in_view : S.in_view.p;
        hipLaunchKernelGGL(local_points_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)n_frames), dim3(256), 0, s, A);
        HIPCHK(hipEventRecord(S.ev[2], s));
        FrameView V{};
        V.kp = S.kp.p; V.desc = S.desc.p; V.cell_begin = G.cell_begin; V.cell_idx = G.cell_idx; V.occupied = S.occ.p;
        V.minX = min_x; V.minY = min_y; V.gridWInv = G.gridWInv; V.gridHInv = G.gridHInv; V.n = capacity;
        for (int l = 0; l < MSORB_MAX_LEVELS; l++) V.inv_sigma2[l] = 0.0f;
        static_assert(sizeof(TopK) == 16 * sizeof(int), "d_topk layout");
        const float mid = scale_factors[nlevels / 2];
        launch_window_topk(V, S.q.p, d_mp_desc, 0, m, reinterpret_cast<TopK*>(d_topk), s, n_frames, capacity, m,
                           n_pairs ? S.cnt.p : nullptr, window_lanes_for(4.0f * th * mid, G.gridWInv, G.gridHInv));
        HIPCHK(hipEventRecord(S.ev[3], s));
    }
    if (n_pairs) HIPCHK(hipMemcpyAsync(n_pairs, S.cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (elapsed_ms) {
        HIPCHK(hipEventElapsedTime(&elapsed_ms[0], S.ev[0], S.ev[1]));
        if (m > 0) {
            HIPCHK(hipEventElapsedTime(&elapsed_ms[1], S.ev[1], S.ev[2]));
            HIPCHK(hipEventElapsedTime(&elapsed_ms[2], S.ev[2], S.ev[3]));
        }
    }
    return MSORB_OK;
}

}  // extern "C"
