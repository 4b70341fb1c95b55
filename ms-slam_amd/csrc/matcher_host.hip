// Matcher side of the C ABI: frame handle (features + 64x48 grid on the device), the sequential replay
// that reproduces the reference's in-loop side effects, stereo matching orchestration.
//
// Sequential side effects.  SearchByProjection assigns F.mvpMapPoints[bestIdx] inside
// its loop and later map points skip keypoints that hold a map point with Observations() > 0
// (ORBmatcher.cc:88-90,129).  The device computes, for every query at once, the kTopK best candidates
// against an occupancy SNAPSHOT; the host then replays the accept rules in query order, dropping
// candidates claimed since the snapshot.  If a query's list is exhausted (all but <2 of a full list were
// claimed) or a keypoint was freed (only possible through the mbSparsified bypass), the snapshot is
// refreshed and the kernel re-run from that query on — the result is always the reference's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "matcher_device.h"
#include "orb_device.h"

using namespace msorb;

#include "matcher_host.h"

extern "C" {

int msorb_frame_create(int device, msorb_frame** out) {
    if (!out) return MSORB_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    msorb_frame* f = new msorb_frame();
    f->device = device;
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) { delete f; return MSORB_E_HIP; }
    *out = f;
    return MSORB_OK;
}

void msorb_frame_destroy(msorb_frame* f) {
    if (!f) return;
    // called from thread-exit / static destructors too (the host classes' per-thread caches): when the HIP runtime has
    // already shut down nothing can be freed any more — and nothing needs to be
    if (hipSetDevice(f->device) != hipSuccess) { delete f; return; }
    if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
    f->d_kp.release(); f->d_win.release(); f->d_desc.release(); f->d_occ.release(); f->d_qdesc.release(); f->d_cell_begin.release();
    f->d_cell_idx.release(); f->d_q.release(); f->d_topk.release(); f->h_in.release(); f->h_topk.release(); f->d_n.release(); f->d_stage.release(); f->d_init_cnt.release(); f->d_init_beg.release(); f->d_init_list.release();
    frame_track_release(f);
    delete f;
}

int msorb_frame_set(msorb_frame* f, const msorb_keypoint* kps, int n, const uint8_t* desc, const float* u_right,
                    float min_x, float max_x, float min_y, float max_y, const float* scale_factors, int nlevels) {
    if (!f || n < 0 || (n > 0 && (!kps || !desc)) || !scale_factors || nlevels < 1 || nlevels > MSORB_MAX_LEVELS ||
        !(max_x > min_x) || !(max_y > min_y))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    // everything that can refuse the call is checked BEFORE the handle is touched; a failure after that empties it
    const int limit = frame_grid_max_keypoints();
    if (limit < 0) return MSORB_E_HIP;
    if (n > limit) {
        set_last_error("msorb_frame_set: " + std::to_string(n) + " keypoints, the device grid takes " + std::to_string(limit));
        return MSORB_E_CAPACITY;
    }
    struct Guard {   // any early return below leaves an empty handle
        msorb_frame* f;
        bool ok = false;
        ~Guard() { if (!ok) frame_invalidate(f); }
    } guard{f};
    // AssignFeaturesToGrid (Frame.cc:385-416) runs on the device (frame_grid_kernel, track.hip): the features are staged as
    // they are — cv::KeyPoint records, descriptor rows, mvuRight — and the kernel derives the train arrays and the grid
    int rc;
    const size_t kb = (size_t)n * sizeof(msorb_keypoint), db = (size_t)n * 32, ub = u_right ? (size_t)n * sizeof(float) : 0;
    const size_t o_desc = (kb + 63) & ~(size_t)63, o_ur = o_desc + ((db + 63) & ~(size_t)63), total = o_ur + ub + 64;
    if ((rc = f->h_in.ensure(total)) || (rc = f->d_stage.ensure(total))) return rc;
    hipStream_t s = f->stream;
    if (n) {
        std::memcpy(f->h_in.p, kps, kb);
        std::memcpy(f->h_in.p + o_desc, desc, db);
        if (u_right) std::memcpy(f->h_in.p + o_ur, u_right, ub);
        HIPCHK(small_copy(f->d_stage.p, f->h_in.p, o_ur + ub, hipMemcpyHostToDevice, s));   // (the copy kernel: a pinned block of ~130 KB is up before an SDMA copy has started)
    }
    if ((rc = enqueue_frame_from_device(f, s, reinterpret_cast<const msorb_keypoint*>(f->d_stage.p), f->d_stage.p + o_desc,
                                        u_right ? reinterpret_cast<const float*>(f->d_stage.p + o_ur) : nullptr, nullptr, n, n, min_x,
                                        max_x, min_y, max_y, scale_factors, nlevels)))
        return rc;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    f->N = n;
    f->kps.assign(kps, kps + n);
    if (u_right) f->u_right.assign(u_right, u_right + n); else f->u_right.assign(n, -1.0f);
    guard.ok = true;
    return MSORB_OK;
}

// Telemetry of the claim-replaying searches (SearchByProjection forms): how many device rounds the last one needed (1 unless a
// candidate list was exhausted by earlier claims; every extra round re-runs the kernel from the first such query on, so the
// count is bounded by the number of queries) and the totals since the handle was created.
int msorb_frame_search_rounds(const msorb_frame* f, long long* total_rounds, long long* total_searches) {
    if (!f) return MSORB_E_INVALID;
    if (total_rounds) *total_rounds = f->total_rounds;
    if (total_searches) *total_searches = f->total_searches;
    return f->last_rounds;
}

int msorb_frame_features_in_area(const msorb_frame* f, float x, float y, float r, int min_level, int max_level,
                                 int* out, int capacity, int* n_out) {
    if (!f || !n_out) return MSORB_E_INVALID;
    int n = 0;
    *n_out = 0;
    {   // grid built on the device (msorb_frame_set / _set_device / msorb_extract_stereo_frame): fetched once, under the frame's lock
        HIPCHK(hipSetDevice(f->device));
        const int rc = frame_host_grid(const_cast<msorb_frame*>(f));
        if (rc) return rc;
    }
    const int minCX = std::max(0, (int)std::floor((x - f->minX - r) * f->gridWInv));
    if (minCX >= kGridCols) return MSORB_OK;
    const int maxCX = std::min(kGridCols - 1, (int)std::ceil((x - f->minX + r) * f->gridWInv));
    if (maxCX < 0) return MSORB_OK;
    const int minCY = std::max(0, (int)std::floor((y - f->minY - r) * f->gridHInv));
    if (minCY >= kGridRows) return MSORB_OK;
    const int maxCY = std::min(kGridRows - 1, (int)std::ceil((y - f->minY + r) * f->gridHInv));
    if (maxCY < 0) return MSORB_OK;
    const bool check = (min_level > 0) || (max_level >= 0);
    for (int ix = minCX; ix <= maxCX; ix++)
        for (int iy = minCY; iy <= maxCY; iy++) {
            const int c = ix * kGridRows + iy;
            for (int j = f->cell_begin[c]; j < f->cell_begin[c + 1]; j++) {
                const msorb_keypoint& kp = f->kps[f->cell_idx[j]];
                if (check) {
                    if (kp.octave < min_level) continue;
                    if (max_level >= 0 && kp.octave > max_level) continue;
                }
                if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) {
                    if (n < capacity) out[n] = f->cell_idx[j];
                    n++;
                }
            }
        }
    *n_out = n;
    return n > capacity ? MSORB_E_CAPACITY : MSORB_OK;
}

int msorb_search_by_projection_mps(msorb_frame* f, int M, const uint8_t* track_in_view, const uint8_t* bad,
                                   const uint8_t* sparsified, const float* proj_x, const float* proj_y,
                                   const float* proj_xr, const float* track_depth, const int* level,
                                   const float* view_cos, const uint8_t* mp_desc, const int* obs, int* frame_mp,
                                   float th, int far_points, float th_far, float nnratio, int* nmatches) {
    if (!f || M < 0 || !nmatches || (M > 0 && (!track_in_view || !bad || !sparsified || !proj_x || !proj_y || !proj_xr ||
                                               !track_depth || !level || !view_cos || !mp_desc || !obs)) ||
        (f->N > 0 && !frame_mp))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    *nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<WinQuery> q(M);
    for (int i = 0; i < M; i++) {
        WinQuery w{};
        bool valid = track_in_view[i] && !(far_points && track_depth[i] > th_far) && !bad[i];
        if (valid && (level[i] < 0 || level[i] >= f->nlevels)) { set_last_error("predicted level out of range"); return MSORB_E_INVALID; }
        if (valid) {
            float r = (view_cos[i] > 0.998) ? 2.5 : 4.0;  // RadiusByViewingCos, ORBmatcher.cc:215-221
            if (bFactor) r *= th;
            w.x = proj_x[i]; w.y = proj_y[i];
            w.r = r * f->scale[level[i]];
            w.ur = proj_xr[i];
            w.min_level = (int16_t)(level[i] - 1);
            w.max_level = (int16_t)level[i];
            w.flags = kQValid | (sparsified[i] ? 0 : kQSkipOccupied);
        }
        q[i] = w;
    }
    std::vector<uint8_t> occ(f->N);
    for (int i = 0; i < f->N; i++) {
        if (frame_mp[i] >= M) { set_last_error("frame_mp holds an id outside the map-point table"); return MSORB_E_INVALID; }
        occ[i] = frame_mp[i] >= 0 && obs[frame_mp[i]] > 0;
    }
    int nm = 0;
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
                *new_occ = obs[qi] > 0;
                return bestIdx;
            }
        }
        return -1;
    };
    const int rc = run_window_search(f, M, q.data(), nullptr, mp_desc, occ, 2, accept);
    *nmatches = nm;
    return rc;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint>&, th, bFarPoints, thFarPoints) on a TWO-CAMERA frame (F.Nleft != -1:
// the KannalaBrandt8 stereo rig), ORBmatcher.cc:43-213 with both arms: per map point a LEFT pass (mbTrackInView: window in the left
// camera's grid, best / second over the left keypoints, :61-142) and a RIGHT pass (mbTrackInViewR: window in the right camera's grid,
// :144-210).  The two cameras are two device frames (left: F.mvKeys[0, Nleft), right: F.mvKeysRight — what Frame::GetFeaturesInArea
// walks for such a frame, Frame.cc:589-655 with bRight); F.mvpMapPoints is one array of n_left + n_right entries.  What couples
// the passes, all replayed here in map-point order:
//   * a left match also claims the right keypoint it is stereo-matched with (mvLeftToRightMatch, :130-134), a right match the left
//     one (mvRightToLeftMatch, :196-200): the occupancy each LATER map point sees on either side;
//   * a left pass that fails its ratio test `continue`s the map-point loop (:125-126): the right pass of that point is skipped;
//   * the right pass does not scale its radius by th (:148) and has no mbSparsified bypass (:169-171).
// Both device searches run against occupancy snapshots; the replay drops candidates claimed since and re-runs a side from the first
// query whose list is exhausted (run_window_search's rule, per side).
namespace {
struct RigSide {
    msorb_frame* f = nullptr;
    std::vector<WinQuery> q;
    std::vector<uint8_t> occ, snap;
    std::vector<int8_t> diff;   // occupancy now vs the snapshot the side's lists were computed against
    int n_freed = 0, next = 0, fresh_from = 0, lanes = 16, rounds = 0;
    bool pristine = true;   // no occupancy change since the last round: only then is the list of query `fresh_from` exact as it stands.  (The
                            // OTHER camera's pass of the same map point can change this side between its round and its first query: a left
                            // match of a point without observations frees the right partner it overwrites, ORBmatcher.cc:130-134.)
    uint8_t* h_occ = nullptr;
    int prepare(int M, const uint8_t* mp_desc) {   // queries + query descriptors to the device, once
        int rc;
        if ((rc = f->d_q.ensure(M)) || (rc = f->d_qdesc.ensure((size_t)M * 32)) || (rc = f->d_topk.ensure(M)) || (rc = f->d_occ.ensure(std::max(f->N, 1))))
            return rc;
        const size_t qb = (size_t)M * sizeof(WinQuery), db = (size_t)M * 32;
        if ((rc = f->h_in.ensure(qb + db + (size_t)f->N + 64)) || (rc = f->h_topk.ensure(M))) return rc;
        double sum = 0;
        int nv = 0;
        for (int i = 0; i < M; i++)
            if (q[i].flags & kQValid) { sum += q[i].r; nv++; }
        if (nv) lanes = window_lanes_for((float)(sum / nv), f->gridWInv, f->gridHInv);
        std::memcpy(f->h_in.p, q.data(), qb);
        std::memcpy(f->h_in.p + qb, mp_desc, db);
        HIPCHK(small_copy(f->d_q.p, f->h_in.p, qb, hipMemcpyHostToDevice, f->stream));
        HIPCHK(small_copy(f->d_qdesc.p, f->h_in.p + qb, db, hipMemcpyHostToDevice, f->stream));
        h_occ = f->h_in.p + qb + db;
        diff.assign(f->N, 0);
        return MSORB_OK;
    }
    int round(int M) {   // the side's lists for queries [next, M) against its occupancy as it is now
        if (f->N > 0 && next < M) {
            std::memcpy(h_occ, occ.data(), f->N);
            HIPCHK(small_copy(f->d_occ.p, h_occ, f->N, hipMemcpyHostToDevice, f->stream));
            launch_window_topk(f->view(), f->d_q.p, f->d_qdesc.p, next, M, f->d_topk.p, f->stream, 1, 0, 0, nullptr, lanes);
            HIPCHK(small_copy(f->h_topk.p + next, f->d_topk.p + next, (size_t)(M - next) * sizeof(TopK), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipStreamSynchronize(f->stream));
        }
        snap = occ;
        std::fill(diff.begin(), diff.end(), 0);
        n_freed = 0;
        fresh_from = next;
        pristine = true;
        freed.clear();
        rounds++;
        return MSORB_OK;
    }
    std::vector<int> freed;   // keypoints freed since the round (entries whose diff is no longer negative were claimed again)
    void set_occ(int idx, int v) {
        pristine = false;
        occ[idx] = (uint8_t)v;
        const int8_t d = (int8_t)((int)occ[idx] - (int)snap[idx]);
        if (diff[idx] < 0) n_freed--;
        diff[idx] = d;
        if (d < 0) { n_freed++; freed.push_back(idx); }
    }
    // a keypoint that was occupied at the round and is free now is missing from the lists of exactly those queries whose window
    // (box and level band, as window_topk_kernel tests them) holds it: only such a query needs a new round
    bool window_holds_a_freed_keypoint(const WinQuery& w) const {
        for (int idx : freed) {
            if (diff[idx] >= 0) continue;
            const msorb_keypoint& kp = f->kps[idx];
            if (kp.octave < w.min_level || (w.max_level >= 0 && kp.octave > w.max_level)) continue;
            if (fabsf(kp.x - w.x) < w.r && fabsf(kp.y - w.y) < w.r) return true;
        }
        return false;
    }
    // the exact candidate prefix of query qi (>= need entries unless the true candidate set is smaller); false: the list cannot be
    // trusted any more (exhausted by claims, or a keypoint was freed): the side needs a new round from qi
    bool prefix(int qi, int need, int* idx, int* dist, int* n_out) const {
        const bool skip = q[qi].flags & kQSkipOccupied;
        if (f->N <= 0) { *n_out = 0; return true; }
        const bool stale = qi > fresh_from || !pristine;   // claims may lie between the round and this query
        if (skip && n_freed > 0 && stale && window_holds_a_freed_keypoint(q[qi])) return false;
        const TopK& t = f->h_topk.p[qi];
        int n = 0, n_dev = 0;
        for (int k = 0; k < kTopK; k++) {
            if (t.idx[k] < 0) break;
            n_dev++;
            if (skip && diff[t.idx[k]] > 0) continue;
            idx[n] = t.idx[k]; dist[n] = t.dist[k]; n++;
        }
        if (n < need && n < n_dev && n_dev == kTopK && stale) return false;
        *n_out = n;
        return true;
    }
};
}  // namespace

int msorb_search_by_projection_mps_rig(msorb_frame* left, msorb_frame* right, int M, const uint8_t* track_in_view, const uint8_t* track_in_view_r,
                                       const uint8_t* bad, const uint8_t* sparsified, const float* proj_x, const float* proj_y,
                                       const float* proj_xr, const float* proj_yr, const float* track_depth, const int* level, const int* level_r,
                                       const float* view_cos, const float* view_cos_r, const uint8_t* mp_desc, const int* obs,
                                       const int* left_to_right, const int* right_to_left, int* frame_mp, float th, int far_points, float th_far,
                                       float nnratio, int* nmatches) {
    if (!left || !right || M < 0 || !nmatches ||
        (M > 0 && (!track_in_view || !track_in_view_r || !bad || !sparsified || !proj_x || !proj_y || !proj_xr || !proj_yr || !track_depth || !level ||
                   !level_r || !view_cos || !view_cos_r || !mp_desc || !obs)) ||
        (left->N + right->N > 0 && !frame_mp) || (left->N > 0 && !left_to_right) || (right->N > 0 && !right_to_left) || left->device != right->device)
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(left->device));
    *nmatches = 0;
    const int NL = left->N, NR = right->N;
    const bool bFactor = th != 1.0;
    RigSide L, R;
    L.f = left; R.f = right;
    L.q.assign(M, WinQuery{}); R.q.assign(M, WinQuery{});
    std::vector<uint8_t> live(M, 0);
    for (int i = 0; i < M; i++) {
        if (!track_in_view[i] && !track_in_view_r[i]) continue;          // :50-51
        if (far_points && track_depth[i] > th_far) continue;             // :53-54
        if (bad[i]) continue;                                            // :56-57
        live[i] = 1;
        if (track_in_view[i]) {
            if (level[i] < 0 || level[i] >= left->nlevels) { set_last_error("predicted level out of range"); return MSORB_E_INVALID; }
            float r = (view_cos[i] > 0.998) ? 2.5 : 4.0;                 // RadiusByViewingCos, :215-221
            if (bFactor) r *= th;
            WinQuery& w = L.q[i];
            w.x = proj_x[i]; w.y = proj_y[i]; w.r = r * left->scale[level[i]]; w.ur = 0.0f;
            w.min_level = (int16_t)(level[i] - 1); w.max_level = (int16_t)level[i];
            w.flags = kQValid | (sparsified[i] ? 0 : kQSkipOccupied);
        }
        if (track_in_view_r[i] && level_r[i] != -1) {                    // :144-146
            if (level_r[i] < 0 || level_r[i] >= right->nlevels) { set_last_error("predicted level (right camera) out of range"); return MSORB_E_INVALID; }
            const float r = (view_cos_r[i] > 0.998) ? 2.5 : 4.0;         // (:147: not scaled by th)
            WinQuery& w = R.q[i];
            w.x = proj_xr[i]; w.y = proj_yr[i]; w.r = r * right->scale[level_r[i]]; w.ur = 0.0f;
            w.min_level = (int16_t)(level_r[i] - 1); w.max_level = (int16_t)level_r[i];
            w.flags = kQValid | kQSkipOccupied;
        }
    }
    L.occ.assign(NL, 0); R.occ.assign(NR, 0);
    for (int i = 0; i < NL + NR; i++) {
        if (frame_mp[i] >= M) { set_last_error("frame_mp holds an id outside the map-point table"); return MSORB_E_INVALID; }
        const uint8_t o = frame_mp[i] >= 0 && obs[frame_mp[i]] > 0;
        if (i < NL) L.occ[i] = o; else R.occ[i - NL] = o;
    }
    for (int i = 0; i < NL; i++) if (left_to_right[i] < -1 || left_to_right[i] >= NR) { set_last_error("left_to_right out of range"); return MSORB_E_INVALID; }
    for (int i = 0; i < NR; i++) if (right_to_left[i] < -1 || right_to_left[i] >= NL) { set_last_error("right_to_left out of range"); return MSORB_E_INVALID; }
    if (M == 0) return MSORB_OK;
    int rc;
    if ((rc = L.prepare(M, mp_desc)) || (rc = R.prepare(M, mp_desc))) return rc;
    int nm = 0;
    // one side's accept rule on a candidate prefix: -1 no match, -2 the ratio test failed (:125-126 / :191-192), else the keypoint
    auto pick = [&](const msorb_frame* f, const int* idx, const int* dist, int n) -> int {
        if (n == 0) return -1;
        const int bestDist = dist[0], bestIdx = idx[0];
        const int bestLevel = f->kps[bestIdx].octave;
        const int bestDist2 = n > 1 ? dist[1] : 256;
        const int bestLevel2 = n > 1 ? f->kps[idx[1]].octave : -1;
        if (bestDist > kThHigh) return -1;
        if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) return -2;
        return bestIdx;   // (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2 holds here)
    };
    bool needL = true, needR = true;
    std::vector<uint8_t> skip_right(M, 0);   // the left pass of the point `continue`d the loop
    for (;;) {
        if (needL && (rc = L.round(M))) return rc;
        if (needR && (rc = R.round(M))) return rc;
        needL = needR = false;
        for (int i = std::min(L.next, R.next); i < M; i++) {
            int idx[kTopK], dist[kTopK], n = 0;
            if (i >= L.next) {
                if (live[i] && (L.q[i].flags & kQValid)) {
                    if (!L.prefix(i, 2, idx, dist, &n)) { needL = true; break; }
                    const int b = pick(left, idx, dist, n);
                    if (b == -2) skip_right[i] = 1;
                    if (b >= 0) {
                        frame_mp[b] = i; nm++;
                        L.set_occ(b, obs[i] > 0);
                        if (left_to_right[b] != -1) {                    // :130-134
                            frame_mp[NL + left_to_right[b]] = i; nm++;
                            R.set_occ(left_to_right[b], obs[i] > 0);
                        }
                    }
                }
                L.next = i + 1;
            }
            if (i >= R.next) {
                if (live[i] && !skip_right[i] && (R.q[i].flags & kQValid)) {
                    if (!R.prefix(i, 2, idx, dist, &n)) { needR = true; break; }
                    const int b = pick(right, idx, dist, n);
                    if (b >= 0) {
                        if (right_to_left[b] != -1) {                    // :196-200
                            frame_mp[right_to_left[b]] = i; nm++;
                            L.set_occ(right_to_left[b], obs[i] > 0);
                        }
                        frame_mp[NL + b] = i; nm++;
                        R.set_occ(b, obs[i] > 0);
                    }
                }
                R.next = i + 1;
            }
        }
        if (!needL && !needR) break;
    }
    left->last_rounds = L.rounds; left->total_rounds += L.rounds; left->total_searches++;
    right->last_rounds = R.rounds; right->total_rounds += R.rounds; right->total_searches++;
    *nmatches = nm;
    return MSORB_OK;
}

namespace {
// Shared body of the two frame-against-projected-points searches: SearchByProjection(Current, Last, th, bMono)
// (:1941-2152) and SearchByProjection(Current, pKF, sAlreadyFound, th, ORBdist) (:2154-2275).  ur == nullptr: no
// mvuRight test (the KeyFrame form has none); obs == nullptr: a keypoint holding any map point is taken (:2214-2215)
// instead of "a map point with observations" (:2011-2013).
int search_projected(msorb_frame* f, int NL, const uint8_t* valid, const float* u, const float* v, const float* ur,
                     const int* octave, const float* angle, const uint8_t* mp_desc, const int* ids, const int* obs, int n_obs,
                     int* cur_mp, float th, int forward, int backward, int check_orientation, float accept_dist, int* nmatches,
                     bool band_below = false, std::vector<std::pair<int, int>>* push_log = nullptr) {
    // push_log: the (query, keypoint) pairs the rotation histogram would receive, in the order of the loop, INSTEAD of the histogram
    // (the two-camera form merges the logs of its two cameras into one histogram)
    *nmatches = 0;
    std::vector<WinQuery> q(NL);
    for (int i = 0; i < NL; i++) {
        WinQuery w{};
        if (valid[i]) {
            const int oct = octave[i];
            if (oct < 0 || oct >= f->nlevels) { set_last_error("octave out of range"); return MSORB_E_INVALID; }
            w.x = u[i]; w.y = v[i];
            w.r = th * f->scale[oct];  // ORBmatcher.cc:1989 / :2202
            w.ur = ur ? ur[i] : 0.0f;
            if (forward) { w.min_level = (int16_t)oct; w.max_level = -1; }
            else if (backward) { w.min_level = 0; w.max_level = (int16_t)oct; }
            else if (band_below) { w.min_level = (int16_t)(oct - 1); w.max_level = (int16_t)oct; }  // :505-507
            else { w.min_level = (int16_t)(oct - 1); w.max_level = (int16_t)(oct + 1); }
            w.flags = kQValid | kQSkipOccupied | (ur ? 0 : kQNoUr);
        }
        q[i] = w;
    }
    std::vector<uint8_t> occ(f->N);
    if (obs) {  // every id that indexes the observation table must lie inside it
        for (int i = 0; i < f->N; i++)
            if (cur_mp[i] >= n_obs) { set_last_error("cur_mp holds an id outside the observation table"); return MSORB_E_INVALID; }
        for (int i = 0; i < NL; i++)
            if (valid[i] && (ids[i] < 0 || ids[i] >= n_obs)) { set_last_error("map-point id outside the observation table"); return MSORB_E_INVALID; }
    }
    for (int i = 0; i < f->N; i++) occ[i] = cur_mp[i] >= 0 && (!obs || obs[cur_mp[i]] > 0);
    int nm = 0;
    std::vector<int> rotHist[kHistoLength];
    const float factor = 1.0f / kHistoLength;
    auto accept = [&](int qi, const int* idx, const int* dist, int n, int* new_occ) -> int {
        if (n == 0) return -1;
        if ((float)dist[0] <= accept_dist) {  // ORBmatcher.cc:2035-2057 / :2229-2247 / :521
            const int bestIdx2 = idx[0];
            cur_mp[bestIdx2] = ids[qi];
            nm++;
            if (push_log) push_log->push_back({qi, bestIdx2});
            else if (check_orientation && angle) {
                float rot = angle[qi] - f->kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == kHistoLength) bin = 0;
                if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(bestIdx2);
            }
            *new_occ = !obs || obs[ids[qi]] > 0;
            return bestIdx2;
        }
        return -1;
    };
    const int rc = run_window_search(f, NL, q.data(), nullptr, mp_desc, occ, 1, accept);
    if (rc) return rc;
    if (check_orientation && !push_log) {  // ORBmatcher.cc:2129-2149 / :2253-2272
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
}  // namespace

int msorb_search_by_projection_frames(msorb_frame* f, int NL, const uint8_t* valid, const float* u, const float* v,
                                      const float* ur, const int* last_octave, const float* last_angle,
                                      const uint8_t* mp_desc, const int* last_mp, const int* obs, int n_obs, int* cur_mp,
                                      float th, int forward, int backward, int check_orientation, int* nmatches) {
    if (!f || NL < 0 || n_obs < 0 || !nmatches || (NL > 0 && (!valid || !u || !v || !ur || !last_octave || !last_angle || !mp_desc ||
                                                 !last_mp || !obs)) || (f->N > 0 && !cur_mp))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    return search_projected(f, NL, valid, u, v, ur, last_octave, last_angle, mp_desc, last_mp, obs, n_obs, cur_mp, th, forward,
                            backward, check_orientation, (float)kThHigh, nmatches);
}

// SearchByProjection(Current, Last, th, bMono) on a two-camera CurrentFrame (ORBmatcher.cc:1941-2152, right arm :2059-2124).  The
// two arms claim disjoint halves of CurrentFrame.mvpMapPoints, so each is search_projected on its camera's frame; what couples
// them: the right window of a last keypoint is only searched when its LEFT window held a candidate (:2003-2004 `continue`s before
// the right arm is reached — GetFeaturesInArea before any occupancy test: the host walk of the same grid), and one rotation
// histogram receives both arms' matches, per last keypoint the left one first (:2054, :2121 — right entries as idx + Nleft).
int msorb_search_by_projection_frames_rig(msorb_frame* left, msorb_frame* right, int NLast, const uint8_t* valid, const float* u, const float* v,
                                          const float* u_r, const float* v_r, const int* last_octave, const float* last_angle,
                                          const uint8_t* mp_desc, const int* last_mp, const int* obs, int n_obs, int* cur_mp, float th,
                                          int forward, int backward, int check_orientation, int* nmatches) {
    if (!left || !right || NLast < 0 || n_obs < 0 || !nmatches || left->device != right->device ||
        (NLast > 0 && (!valid || !u || !v || !u_r || !v_r || !last_octave || !last_angle || !mp_desc || !last_mp || !obs)) ||
        (left->N + right->N > 0 && !cur_mp))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(left->device));
    *nmatches = 0;
    const int NL = left->N;
    std::vector<uint8_t> valid_r(NLast, 0);
    {
        int rc0;
        if ((rc0 = frame_host_grid(left))) return rc0;   // the left camera's grid on the host, fetched once
    }
    // `vIndices2.empty()` of the left window (:2001-2004), whatever its keypoints hold: GetFeaturesInArea's walk (Frame.cc:589-655) cut
    // short at the first keypoint found
    auto left_window_has_a_keypoint = [&](float x, float y, float r, int min_level, int max_level) {
        const msorb_frame* f = left;
        const int minCX = std::max(0, (int)std::floor((x - f->minX - r) * f->gridWInv));
        if (minCX >= kGridCols) return false;
        const int maxCX = std::min(kGridCols - 1, (int)std::ceil((x - f->minX + r) * f->gridWInv));
        if (maxCX < 0) return false;
        const int minCY = std::max(0, (int)std::floor((y - f->minY - r) * f->gridHInv));
        if (minCY >= kGridRows) return false;
        const int maxCY = std::min(kGridRows - 1, (int)std::ceil((y - f->minY + r) * f->gridHInv));
        if (maxCY < 0) return false;
        const bool check = (min_level > 0) || (max_level >= 0);
        for (int ix = minCX; ix <= maxCX; ix++)
            for (int iy = minCY; iy <= maxCY; iy++) {
                const int c = ix * kGridRows + iy;
                for (int j = f->cell_begin[c]; j < f->cell_begin[c + 1]; j++) {
                    const msorb_keypoint& kp = f->kps[f->cell_idx[j]];
                    if (check) {
                        if (kp.octave < min_level) continue;
                        if (max_level >= 0 && kp.octave > max_level) continue;
                    }
                    if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) return true;
                }
            }
        return false;
    };
    for (int i = 0; i < NLast; i++) {
        if (!valid[i]) continue;
        const int oct = last_octave[i];
        if (oct < 0 || oct >= left->nlevels) { set_last_error("octave out of range"); return MSORB_E_INVALID; }
        const float radius = th * left->scale[oct];
        const int lo = forward ? oct : backward ? 0 : oct - 1, hi = forward ? -1 : backward ? oct : oct + 1;
        valid_r[i] = left_window_has_a_keypoint(u[i], v[i], radius, lo, hi);
    }
    std::vector<std::pair<int, int>> pl, pr;
    int nl = 0, nr = 0, rc;
    if ((rc = search_projected(left, NLast, valid, u, v, nullptr, last_octave, last_angle, mp_desc, last_mp, obs, n_obs, cur_mp, th, forward,
                               backward, 0, (float)kThHigh, &nl, false, &pl)))
        return rc;
    if ((rc = search_projected(right, NLast, valid_r.data(), u_r, v_r, nullptr, last_octave, last_angle, mp_desc, last_mp, obs, n_obs,
                               cur_mp + NL, th, forward, backward, 0, (float)kThHigh, &nr, false, &pr)))
        return rc;
    int nm = nl + nr;
    if (check_orientation) {
        std::vector<int> rotHist[kHistoLength];
        const float factor = 1.0f / kHistoLength;
        auto push = [&](int qi, float kp_angle, int entry) {
            float rot = last_angle[qi] - kp_angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == kHistoLength) bin = 0;
            if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(entry);
        };
        size_t a = 0, b = 0;   // both logs are in query order: merge, the left entry of a query first
        while (a < pl.size() || b < pr.size()) {
            if (b >= pr.size() || (a < pl.size() && pl[a].first <= pr[b].first)) { push(pl[a].first, left->kps[pl[a].second].angle, pl[a].second); a++; }
            else { push(pr[b].first, right->kps[pr[b].second].angle, NL + pr[b].second); b++; }
        }
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

int msorb_search_by_projection_kf(msorb_frame* f, int n, const uint8_t* valid, const float* u, const float* v,
                                  const int* predicted_level, const float* kf_angle, const uint8_t* mp_desc, const int* mp_id,
                                  int* cur_mp, float th, int orb_dist, int check_orientation, int* nmatches) {
    if (!f || n < 0 || !nmatches || (n > 0 && (!valid || !u || !v || !predicted_level || !kf_angle || !mp_desc || !mp_id)) ||
        (f->N > 0 && !cur_mp))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    return search_projected(f, n, valid, u, v, nullptr, predicted_level, kf_angle, mp_desc, mp_id, nullptr, 0, cur_mp, th, 0, 0,
                            check_orientation, (float)orb_dist, nmatches);
}

int msorb_search_by_projection_sim3(msorb_frame* f, int n, const uint8_t* valid, const float* u, const float* v,
                                    const int* predicted_level, const uint8_t* mp_desc, const int* mp_id, int* matched,
                                    float th, float max_dist, int* nmatches) {
    if (!f || n < 0 || !nmatches || (n > 0 && (!valid || !u || !v || !predicted_level || !mp_desc || !mp_id)) ||
        (f->N > 0 && !matched))
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(f->device));
    return search_projected(f, n, valid, u, v, nullptr, predicted_level, nullptr, mp_desc, mp_id, nullptr, 0, matched, th, 0, 0, 0,
                            max_dist, nmatches, true);
}

namespace {
// One host-fed window search without claims (Fuse, SearchBySim3, the loop / Sim3 projection forms, msorb_window_top4): queries,
// descriptors, the occupancy / skip flags (null = none set) and an optional gate array are staged side by side in the frame's pinned
// block and go up as ONE copy-kernel launch; the lists come back the same way into f->h_topk.  (Three to four hipMemcpyAsync from
// pageable vectors before: each one staged and synchronised by the runtime.)
int window_search_once(msorb_frame* f, int n, const WinQuery* q, const uint8_t* qdesc, const uint8_t* occ, const KpLite* gate,
                       const float* inv_level_sigma2, int n_levels, const uint8_t* qdesc_dev = nullptr) {   // qdesc_dev: the descriptors already lie on f's device
    int rc;
    const size_t qb = ((size_t)n * sizeof(WinQuery) + 15) & ~(size_t)15, db = qdesc_dev ? 0 : (size_t)n * 32, ob = ((size_t)std::max(f->N, 1) + 15) & ~(size_t)15,
                 gb = gate ? (size_t)f->N * sizeof(KpLite) : 0, total = qb + db + ob + gb;
    if ((rc = f->h_in.ensure(total + 64)) || (rc = f->d_win.ensure(total + 64)) || (rc = f->d_topk.ensure(n)) || (rc = f->h_topk.ensure(n))) return rc;
    if (f->N <= 0) {   // no train keypoints (also a handle that was never set, or whose set failed): every list is empty
        for (int i = 0; i < n; i++)
            for (int k = 0; k < kTopK; k++) { f->h_topk.p[i].idx[k] = -1; f->h_topk.p[i].dist[k] = 256; }
        return MSORB_OK;
    }
    uint8_t* h = f->h_in.p;
    std::memcpy(h, q, (size_t)n * sizeof(WinQuery));
    if (db) std::memcpy(h + qb, qdesc, db);
    if (occ && f->N) std::memcpy(h + qb + db, occ, (size_t)f->N); else std::memset(h + qb + db, 0, ob);
    if (gate) std::memcpy(h + qb + db + ob, gate, gb);
    hipStream_t s = f->stream;
    HIPCHK(small_copy(f->d_win.p, h, total, hipMemcpyHostToDevice, s));
    FrameView view = f->view();
    view.occupied = f->d_win.p + qb + db;
    if (gate) view.gate_kp = reinterpret_cast<const KpLite*>(f->d_win.p + qb + db + ob);
    if (inv_level_sigma2) for (int l = 0; l < n_levels; l++) view.inv_sigma2[l] = inv_level_sigma2[l];
    launch_window_topk(view, reinterpret_cast<const WinQuery*>(f->d_win.p), qdesc_dev ? qdesc_dev : f->d_win.p + qb, 0, n, f->d_topk.p, s);
    HIPCHK(hipGetLastError());
    HIPCHK(small_copy(f->h_topk.p, f->d_topk.p, (size_t)n * sizeof(TopK), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MSORB_OK;
}
}  // namespace

int msorb_window_top4(msorb_frame* f, int n_queries, const float* x, const float* y, const float* r, const float* ur,
                      const int* min_level, const int* max_level, const uint8_t* skip_occupied, const uint8_t* query_desc,
                      const uint8_t* occupied, int* best_idx, int* best_dist) {
    if (!f || n_queries < 0 || (n_queries > 0 && (!x || !y || !r || !min_level || !max_level || !query_desc || !best_idx ||
                                                  !best_dist)))
        return MSORB_E_INVALID;
    if (n_queries == 0) return MSORB_OK;
    HIPCHK(hipSetDevice(f->device));
    std::vector<WinQuery> q(n_queries);
    for (int i = 0; i < n_queries; i++) {
        WinQuery w{};
        w.x = x[i]; w.y = y[i]; w.r = r[i];
        w.ur = ur ? ur[i] : 0.f;
        w.min_level = (int16_t)min_level[i]; w.max_level = (int16_t)max_level[i];
        w.flags = kQValid | ((skip_occupied && skip_occupied[i]) ? kQSkipOccupied : 0);
        q[i] = w;
    }
    const int rc = window_search_once(f, n_queries, q.data(), query_desc, occupied, nullptr, nullptr, 0);
    if (rc) return rc;
    const TopK* const topk = f->h_topk.p;
    for (int i = 0; i < n_queries; i++)
        for (int k = 0; k < 4; k++) { best_idx[4 * i + k] = topk[i].idx[k]; best_dist[4 * i + k] = topk[i].dist[k]; }
    return MSORB_OK;
}

namespace {
int fuse_search_impl(msorb_frame* f, const msorb_keypoint* gate_kps, const float* gate_uright, const float* inv_level_sigma2, int n_levels,
                     int n, const uint8_t* valid, const float* u, const float* v, const float* ur, const int* predicted_level,
                     const float* radius, const uint8_t* mp_desc, int* best_idx, int* best_dist) {
    if (!f || n < 0 || !inv_level_sigma2 || n_levels < 1 || n_levels > MSORB_MAX_LEVELS ||
        (n > 0 && (!valid || !u || !v || !ur || !predicted_level || !radius || !mp_desc || !best_idx || !best_dist)))
        return MSORB_E_INVALID;
    if (n == 0) return MSORB_OK;
    const msorb_keypoint* gk = gate_kps ? gate_kps : f->kps.data();
    for (int i = 0; i < f->N; i++)
        if (gk[i].octave < 0 || gk[i].octave >= n_levels) { set_last_error("fuse_search: keypoint octave outside inv_level_sigma2"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(f->device));
    std::vector<WinQuery> q(n);
    for (int i = 0; i < n; i++) {
        WinQuery w{};
        if (valid[i]) {
            w.x = u[i]; w.y = v[i]; w.r = radius[i]; w.ur = ur[i];
            w.min_level = (int16_t)(predicted_level[i] - 1);  // :1513-1514
            w.max_level = (int16_t)predicted_level[i];
            w.flags = kQValid | kQFuseGate;
        }
        q[i] = w;
    }
    std::vector<KpLite> gate;
    if (gate_kps && f->N) {
        gate.resize(f->N);
        for (int i = 0; i < f->N; i++) gate[i] = KpLite{gate_kps[i].x, gate_kps[i].y, gate_uright ? gate_uright[i] : -1.0f, gate_kps[i].octave};
    }
    const int rc = window_search_once(f, n, q.data(), mp_desc, nullptr, gate.empty() ? nullptr : gate.data(), inv_level_sigma2, n_levels);
    if (rc) return rc;
    const TopK* const topk = f->h_topk.p;
    for (int i = 0; i < n; i++) { best_idx[i] = topk[i].idx[0]; best_dist[i] = topk[i].dist[0]; }
    return MSORB_OK;
}
}  // namespace

int msorb_fuse_search(msorb_frame* f, const float* inv_level_sigma2, int n_levels, int n, const uint8_t* valid,
                      const float* u, const float* v, const float* ur, const int* predicted_level, const float* radius,
                      const uint8_t* mp_desc, int* best_idx, int* best_dist) {
    return fuse_search_impl(f, nullptr, nullptr, inv_level_sigma2, n_levels, n, valid, u, v, ur, predicted_level, radius, mp_desc, best_idx, best_dist);
}

int msorb_fuse_search_gated(msorb_frame* f, const msorb_keypoint* gate_kps, const float* gate_uright, const float* inv_level_sigma2,
                            int n_levels, int n, const uint8_t* valid, const float* u, const float* v, const float* ur,
                            const int* predicted_level, const float* radius, const uint8_t* mp_desc, int* best_idx, int* best_dist) {
    if (f && f->N > 0 && !gate_kps) { set_last_error("fuse_search_gated: gate_kps is null"); return MSORB_E_INVALID; }
    return fuse_search_impl(f, gate_kps, gate_uright, inv_level_sigma2, n_levels, n, valid, u, v, ur, predicted_level, radius, mp_desc, best_idx, best_dist);
}

namespace {
// Claim-free window search on a KeyFrame: for every valid query the keypoint of f with the smallest distance inside
// GetFeaturesInArea(u, v, th * mvScaleFactors[level]) (KeyFrame.cc:796-845) at levels level-1 .. level, first strict
// minimum in scan order; best_idx -1 / best_dist INT_MAX when the band is empty.  The search shared by both passes of
// SearchBySim3 (ORBmatcher.cc:1796-1841, 1888-1913) and by Fuse(pKF, Scw, ...) (:1661-1696).
int plain_window_best(msorb_frame* f, int n, const uint8_t* valid, const float* u, const float* v, const int* level, float th,
                      const uint8_t* desc, int* best_idx, int* best_dist, int band_above = 0, const uint8_t* train_skip = nullptr) {
    for (int i = 0; i < n; i++) { best_idx[i] = -1; best_dist[i] = INT_MAX; }
    if (n == 0) return MSORB_OK;
    HIPCHK(hipSetDevice(f->device));
    std::vector<WinQuery> q(n);
    for (int i = 0; i < n; i++) {
        WinQuery w{};
        if (valid[i]) {
            if (level[i] < 0 || level[i] >= f->nlevels) { set_last_error("predicted level out of range"); return MSORB_E_INVALID; }
            w.x = u[i]; w.y = v[i];
            w.r = th * f->scale[level[i]];
            w.min_level = (int16_t)(level[i] - 1);
            w.max_level = (int16_t)(level[i] + band_above);
            w.flags = kQValid | kQNoUr | (train_skip ? kQSkipOccupied : 0);
        }
        q[i] = w;
    }
    const int rc = window_search_once(f, n, q.data(), desc, train_skip, nullptr, nullptr, 0);
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        if (f->h_topk.p[i].idx[0] >= 0) { best_idx[i] = f->h_topk.p[i].idx[0]; best_dist[i] = f->h_topk.p[i].dist[0]; }
    return MSORB_OK;
}
}  // namespace

int msorb_fuse_sim3_search(msorb_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                           const int* predicted_level, const uint8_t* mp_desc, float th, int* best_idx, int* best_dist) {
    if (!kf || n < 0 || (n > 0 && (!valid || !u || !v || !predicted_level || !mp_desc || !best_idx || !best_dist)))
        return MSORB_E_INVALID;
    return plain_window_best(kf, n, valid, u, v, predicted_level, th, mp_desc, best_idx, best_dist);
}

int msorb_search_by_projection_loop(msorb_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                                    const int* predicted_level, const uint8_t* mp_desc, const uint8_t* train_ok, float th,
                                    float max_dist, int* best_idx, int* nmatches) {
    if (!kf || n < 0 || !nmatches || (n > 0 && (!valid || !u || !v || !predicted_level || !mp_desc || !best_idx)) ||
        (kf->N > 0 && !train_ok))
        return MSORB_E_INVALID;
    *nmatches = 0;
    std::vector<uint8_t> skip(kf->N);
    for (int i = 0; i < kf->N; i++) skip[i] = !train_ok[i];             // :609-610
    std::vector<int> bd(n);
    const int rc = plain_window_best(kf, n, valid, u, v, predicted_level, th, mp_desc, best_idx, bd.data(), 1, skip.data());
    if (rc) return rc;
    int nm = 0;
    for (int i = 0; i < n; i++) {
        const int d = best_idx[i] >= 0 ? bd[i] : 256;                   // bestDist starts at 256 (:599)
        if ((float)d <= max_dist && best_idx[i] >= 0) nm++;             // :626-631
        else best_idx[i] = -1;
    }
    *nmatches = nm;
    return MSORB_OK;
}

int msorb_search_by_sim3(msorb_frame* kf1, msorb_frame* kf2, int n1, const uint8_t* valid1, const float* u1, const float* v1,
                         const int* level1, const uint8_t* desc1, int n2, const uint8_t* valid2, const float* u2,
                         const float* v2, const int* level2, const uint8_t* desc2, float th, int* match12, int* nfound) {
    if (!kf1 || !kf2 || n1 < 0 || n2 < 0 || !nfound || (n1 > 0 && (!valid1 || !u1 || !v1 || !level1 || !desc1 || !match12)) ||
        (n2 > 0 && (!valid2 || !u2 || !v2 || !level2 || !desc2)))
        return MSORB_E_INVALID;
    *nfound = 0;
    std::vector<int> m1(n1), d1(n1), m2(n2), d2(n2);
    int rc;
    if ((rc = plain_window_best(kf2, n1, valid1, u1, v1, level1, th, desc1, m1.data(), d1.data()))) return rc;  // :1758-1848
    if ((rc = plain_window_best(kf1, n2, valid2, u2, v2, level2, th, desc2, m2.data(), d2.data()))) return rc;  // :1850-1920
    int nf = 0;
    for (int i1 = 0; i1 < n1; i1++) {  // :1922-1937
        match12[i1] = -1;
        const int idx2 = d1[i1] <= kThHigh ? m1[i1] : -1;
        if (idx2 >= 0 && idx2 < n2) {
            const int idx1 = d2[idx2] <= kThHigh ? m2[idx2] : -1;
            if (idx1 == i1) { match12[i1] = idx2; nf++; }
        }
    }
    *nfound = nf;
    return MSORB_OK;
}

namespace {
// Every candidate of every window from query `from` on, with its distance, in the reference's scan order (window_list_kernel:
// count, then fill) — the complete form of the search below, taken when a top-8 list cannot settle a query.
int init_full_lists(msorb_frame* f1, msorb_frame* f2, std::vector<WinQuery> q, int from, std::vector<int>& cnt, std::vector<int>& beg,
                    std::vector<int2>& list) {
    const int N1 = (int)q.size();
    for (int i = 0; i < from; i++) q[i].flags = 0;
    std::vector<uint8_t> qdesc((size_t)N1 * 32);
    HIPCHK(hipSetDevice(f1->device));
    HIPCHK(hipMemcpy(qdesc.data(), f1->d_desc.p, (size_t)N1 * 32, hipMemcpyDeviceToHost));  // F1.mDescriptors as uploaded
    HIPCHK(hipSetDevice(f2->device));
    int rc;
    // grow-only scratch on the train frame's handle (a per-call hipMalloc / hipFree pair synchronises the whole device
    // while the other SLAM threads have kernels in flight, and leaked on the early returns)
    DBuf<int>&d_cnt = f2->d_init_cnt, &d_beg = f2->d_init_beg;
    DBuf<int2>& d_list = f2->d_init_list;
    if ((rc = f2->d_q.ensure(N1)) || (rc = f2->d_qdesc.ensure((size_t)N1 * 32)) || (rc = d_cnt.ensure(N1)) ||
        (rc = d_beg.ensure(N1 + 1)))
        return rc;
    hipStream_t s = f2->stream;
    cnt.assign(N1, 0); beg.assign(N1 + 1, 0);
    hipError_t e = hipMemcpyAsync(f2->d_q.p, q.data(), (size_t)N1 * sizeof(WinQuery), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(f2->d_qdesc.p, qdesc.data(), (size_t)N1 * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        launch_window_list(f2->view(), f2->d_q.p, f2->d_qdesc.p, N1, d_cnt.p, nullptr, nullptr, false, s);
        e = hipMemcpyAsync(cnt.data(), d_cnt.p, (size_t)N1 * sizeof(int), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
        for (int i = 0; i < N1; i++) beg[i + 1] = beg[i] + cnt[i];
        list.resize(std::max(beg[N1], 1));
        rc = d_list.ensure(list.size());
        if (rc == MSORB_OK && beg[N1] > 0) {
            e = hipMemcpyAsync(d_beg.p, beg.data(), (size_t)(N1 + 1) * sizeof(int), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) {
                launch_window_list(f2->view(), f2->d_q.p, f2->d_qdesc.p, N1, d_cnt.p, d_beg.p, d_list.p, true, s);
                e = hipMemcpyAsync(list.data(), d_list.p, (size_t)beg[N1] * sizeof(int2), hipMemcpyDeviceToHost, s);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return MSORB_E_HIP; }
    return rc;
}
}  // namespace

// ORBmatcher::SearchForInitialization (ORBmatcher.cc:755-870).  A query's scan keeps best / second best over the window's
// candidates that no earlier query holds at a smaller or equal distance (vMatchedDistance, :791-792): only the LOW end of a
// window's distances ever decides anything.  The device ranks each window's eight best candidates (window_topk_kernel: distance,
// then scan position — the first strict minimum of the scan is the first of the ranked list, and the second best VALUE is the next
// one's); the replay walks a list past the candidates the rule skips.  A list is conclusive when it ends before eight entries, when
// its first survivor is above TH_LOW, when two survivors are found, when the one survivor passes the ratio test against the
// EIGHTH distance (every unlisted candidate is at least that far), or when nothing survives and the eighth is already above TH_LOW.  The first query whose list is not switches the rest of the
// call to the complete lists (window_list_kernel: rounds 1-5 took them for every query — 4 MB and 4 ms at 10 000 features).
int msorb_search_for_initialization(msorb_frame* f1, msorb_frame* f2, float* prev_xy, int window_size, float nnratio,
                                    int check_orientation, int* matches12, int* nmatches) {
    if (!f1 || !f2 || !nmatches || (f1->N > 0 && (!prev_xy || !matches12))) return MSORB_E_INVALID;
    *nmatches = 0;
    const int N1 = f1->N, N2 = f2->N;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    if (N1 == 0) return MSORB_OK;
    HIPCHK(hipSetDevice(f2->device));
    std::vector<WinQuery> q(N1);
    for (int i = 0; i < N1; i++) {
        WinQuery w{};
        const int level1 = f1->kps[i].octave;
        if (level1 <= 0) {                                            // :769-771 (only level-0 keypoints are matched)
            w.x = prev_xy[2 * i]; w.y = prev_xy[2 * i + 1];
            w.r = (float)window_size;
            w.min_level = (int16_t)level1; w.max_level = (int16_t)level1;  // GetFeaturesInArea(x, y, windowSize, level1, level1)
            w.flags = kQValid | kQNoUr;
        }
        q[i] = w;
    }
    int rc;
    int full_from = N1;                 // queries from here on take the complete lists
    std::vector<int> cnt, beg;
    std::vector<int2> list;
    if (f1->device == f2->device && f1->d_desc.p) {
        if ((rc = window_search_once(f2, N1, q.data(), nullptr, nullptr, nullptr, nullptr, 0, f1->d_desc.p))) return rc;   // F1.mDescriptors as uploaded
    } else {
        full_from = 0;
        if ((rc = init_full_lists(f1, f2, q, 0, cnt, beg, list))) return rc;
    }
    const TopK* const topk = f2->h_topk.p;
    // the reference's loop (:767-835): every distance that can matter is known, the sequential part is compare / select
    int nm = 0;
    std::vector<int> rotHist[kHistoLength];
    const float factor = 1.0f / kHistoLength;
    std::vector<int> vMatchedDistance(N2, INT_MAX), vnMatches21(N2, -1);
    for (int i1 = 0; i1 < N1; i1++) {
        if (!(q[i1].flags & kQValid)) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        if (i1 < full_from) {
            const TopK& t = topk[i1];
            int n_listed = 0, survivors = 0;
            for (int k = 0; k < kTopK && t.idx[k] >= 0; k++) {
                n_listed++;
                if (vMatchedDistance[t.idx[k]] <= t.dist[k]) continue;                 // :791-792
                if (survivors == 0) { bestDist = t.dist[k]; bestIdx2 = t.idx[k]; }
                else if (survivors == 1) bestDist2 = t.dist[k];
                if (++survivors == 2) break;
            }
            bool settled = n_listed < kTopK || survivors == 2 || (survivors == 1 && bestDist > kThLow) ||
                           (survivors == 0 && t.dist[kTopK - 1] > kThLow);   // (nothing listed survives and every unlisted candidate is beyond TH_LOW: no match)
            if (!settled && survivors == 1 && (float)bestDist < (float)t.dist[kTopK - 1] * nnratio) {
                // one survivor, every unlisted candidate at least as far as the eighth: the ratio test passes whatever the second is
                settled = true;
                bestDist2 = t.dist[kTopK - 1];   // (a lower bound of it: only its product with nnratio is used, and that test is decided)
            }
            if (!settled) {
                full_from = i1;
                if ((rc = init_full_lists(f1, f2, q, full_from, cnt, beg, list))) return rc;
            }
        }
        if (i1 >= full_from) {
            bestDist = INT_MAX; bestDist2 = INT_MAX; bestIdx2 = -1;
            for (int k = beg[i1]; k < beg[i1 + 1]; k++) {
                const int i2 = list[k].x, dist = list[k].y;
                if (vMatchedDistance[i2] <= dist) continue;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
        }
        if (bestIdx2 >= 0 && bestDist <= kThLow) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nm--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nm++;
                if (check_orientation) {
                    float rot = f1->kps[i1].angle - f2->kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == kHistoLength) bin = 0;
                    if (bin >= 0 && bin < kHistoLength) rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {  // :837-861
        int sizes[kHistoLength], ind[3];
        for (int i = 0; i < kHistoLength; i++) sizes[i] = (int)rotHist[i].size();
        msorb_three_maxima(sizes, kHistoLength, ind);
        for (int i = 0; i < kHistoLength; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
        }
    }
    for (int i1 = 0; i1 < N1; i1++)  // :864-867
        if (matches12[i1] >= 0) { prev_xy[2 * i1] = f2->kps[matches12[i1]].x; prev_xy[2 * i1 + 1] = f2->kps[matches12[i1]].y; }
    *nmatches = nm;
    return MSORB_OK;
}

int msorb_three_maxima(const int* sizes, int L, int* ind) {
    if (!sizes || !ind || L < 0) return MSORB_E_INVALID;
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    ind[0] = ind1; ind[1] = ind2; ind[2] = ind3;
    return MSORB_OK;
}

int msorb_hamming_top2(int device, const uint8_t* qdesc, int nq, const uint8_t* tdesc, int nt, const int* cand_begin,
                       const int* cand_idx, int* best_idx, int* best_dist, int* second_idx, int* second_dist) {
    if (nq < 0 || nt < 0 || (nq > 0 && (!qdesc || !cand_begin || !best_idx || !best_dist || !second_idx || !second_dist)))
        return MSORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    if (nq == 0) return MSORB_OK;
    HIPCHK(hipSetDevice(device));
    const int total = cand_begin[nq];
    for (int i = 0; i < total; i++)
        if (cand_idx[i] < 0 || cand_idx[i] >= nt) { set_last_error("candidate index out of range"); return MSORB_E_INVALID; }
    // per-thread grow-only scratch, one staged block up ([queries | trains | list offsets | list]) and one down (rounds 1-5: five
    // hipMalloc / hipFree and eight synchronous copies on the null stream per call)
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        DBuf<uint8_t> d;
        HBuf<uint8_t> h;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            d.release(); h.release();
            if (s) (void)hipStreamDestroy(s);
            s = nullptr; device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch S;
    if (S.device != device) {
        S.release();
        S.device = device;
        HIPCHK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking));
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_t = up16((size_t)nq * 32), o_cb = o_t + up16((size_t)nt * 32), o_ci = o_cb + up16((size_t)(nq + 1) * 4),
                 in_bytes = o_ci + up16((size_t)total * 4), o_out = in_bytes, bytes = o_out + up16((size_t)4 * nq * 4);
    int rc;
    if ((rc = S.d.ensure(bytes)) || (rc = S.h.ensure(bytes))) return rc;
    std::memcpy(S.h.p, qdesc, (size_t)nq * 32);
    if (nt) std::memcpy(S.h.p + o_t, tdesc, (size_t)nt * 32);
    std::memcpy(S.h.p + o_cb, cand_begin, (size_t)(nq + 1) * sizeof(int));
    if (total) std::memcpy(S.h.p + o_ci, cand_idx, (size_t)total * sizeof(int));
    hipStream_t s = S.s;
    HIPCHK(small_copy(S.d.p, S.h.p, in_bytes, hipMemcpyHostToDevice, s));
    int* const dout = reinterpret_cast<int*>(S.d.p + o_out);
    launch_list_top2(S.d.p, S.d.p + o_t, reinterpret_cast<const int*>(S.d.p + o_cb), reinterpret_cast<const int*>(S.d.p + o_ci), nq, dout, dout + nq,
                     dout + 2 * nq, dout + 3 * nq, s);
    HIPCHK(hipGetLastError());
    HIPCHK(small_copy(S.h.p + o_out, S.d.p + o_out, (size_t)4 * nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int* ho = reinterpret_cast<const int*>(S.h.p + o_out);
    std::memcpy(best_idx, ho, (size_t)nq * sizeof(int));
    std::memcpy(best_dist, ho + nq, (size_t)nq * sizeof(int));
    std::memcpy(second_idx, ho + 2 * nq, (size_t)nq * sizeof(int));
    std::memcpy(second_dist, ho + 3 * nq, (size_t)nq * sizeof(int));
    return MSORB_OK;
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) as Frame::ComputeStereoFishEyeMatches uses it (Frame.cc:1076)
// on host arrays: per query the nearest and the second nearest train row (ties -> lower train index, the order batchDistance
// keeps).  The dense top-2 kernel holds 2048 train rows per launch; longer train sets go chunk by chunk and the per-chunk pairs
// are merged lexicographically (distance, index) on the host.
int msorb_knn_match2(int device, const uint8_t* query, int n_query, const uint8_t* train, int n_train, int* best_idx, int* best_dist,
                     int* second_idx, int* second_dist) {
    if (n_query < 0 || n_train < 0 || (n_query > 0 && (!query || !best_idx || !best_dist || !second_dist)) || (n_train > 0 && !train))
        return MSORB_E_INVALID;
    for (int i = 0; i < n_query; i++) { best_idx[i] = -1; best_dist[i] = 256; if (second_idx) second_idx[i] = -1; second_dist[i] = 256; }
    if (n_query == 0 || n_train == 0) return MSORB_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        DBuf<uint8_t> q, t;
        DBuf<int> out, n;
        HBuf<int> h;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            q.release(); t.release(); out.release(); n.release(); h.release();
            if (s) (void)hipStreamDestroy(s);
            s = nullptr; device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch S;
    if (S.device != device) {
        S.release();
        S.device = device;
        HIPCHK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking));
    }
    constexpr int kChunk = 2048;
    const int tc = std::min(n_train, kChunk);
    int rc;
    if ((rc = S.q.ensure((size_t)n_query * 32)) || (rc = S.t.ensure((size_t)tc * 32)) || (rc = S.out.ensure((size_t)3 * n_query)) ||
        (rc = S.n.ensure(2)) || (rc = S.h.ensure((size_t)3 * n_query + 2)))
        return rc;
    hipStream_t s = S.s;
    HIPCHK(hipMemcpyAsync(S.q.p, query, (size_t)n_query * 32, hipMemcpyHostToDevice, s));
    // the kernel reports best (index, distance) and the second DISTANCE; the second index is recovered on the host from the
    // train rows only when the caller asks for it
    for (int t0 = 0; t0 < n_train; t0 += kChunk) {
        const int nt = std::min(kChunk, n_train - t0);
        int* hn = S.h.p + 3 * (size_t)n_query;
        hn[0] = n_query; hn[1] = nt;
        HIPCHK(hipMemcpyAsync(S.n.p, hn, 2 * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(S.t.p, train + (size_t)t0 * 32, (size_t)nt * 32, hipMemcpyHostToDevice, s));
        launch_dense_top2(S.q.p, S.t.p, S.n.p, S.n.p + 1, 1, n_query, nt, n_query, nt, S.out.p, S.out.p + n_query, S.out.p + 2 * (size_t)n_query, s,
                          MSORB_DENSE_POPCOUNT);
        HIPCHK(hipMemcpyAsync(S.h.p, S.out.p, (size_t)3 * n_query * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        const int *bi = S.h.p, *bd = bi + n_query, *sd = bd + n_query;
        for (int i = 0; i < n_query; i++) {
            // merge {best, second} of this chunk into the running pair: all of a later chunk's indices are larger, so on equal
            // distances the earlier chunk's entries stay in front
            const int cb = bd[i], cs = sd[i], ci = bi[i] + t0;
            if (cb < best_dist[i]) {
                second_dist[i] = std::min(best_dist[i], cs);
                best_dist[i] = cb; best_idx[i] = ci;
            } else {
                second_dist[i] = std::min(second_dist[i], cb);
            }
        }
    }
    if (second_idx) {   // first train row (lowest index) at the second distance that is not the best itself
        for (int i = 0; i < n_query; i++) {
            if (second_dist[i] > 255) continue;
            uint64_t qa[4];
            std::memcpy(qa, query + (size_t)i * 32, 32);
            for (int j = 0; j < n_train; j++) {
                if (j == best_idx[i]) continue;
                uint64_t tb[4];
                std::memcpy(tb, train + (size_t)j * 32, 32);
                const int d = __builtin_popcountll(qa[0] ^ tb[0]) + __builtin_popcountll(qa[1] ^ tb[1]) + __builtin_popcountll(qa[2] ^ tb[2]) +
                              __builtin_popcountll(qa[3] ^ tb[3]);
                if (d == second_dist[i]) { second_idx[i] = j; break; }
            }
        }
    }
    return MSORB_OK;
}

int msorb_hamming_dense_top2_batch(int device, const uint8_t* d_query, const uint8_t* d_train, const int* d_n_query,
                                   const int* d_n_train, int n_frames, int query_stride, int train_stride, int max_query,
                                   int max_train, int* d_best_idx, int* d_best_dist, int* d_second_dist, int repeats,
                                   float* elapsed_ms) {
    return msorb_hamming_dense_top2_batch_ex(device, d_query, d_train, d_n_query, d_n_train, n_frames, query_stride, train_stride,
                                             max_query, max_train, d_best_idx, d_best_dist, d_second_dist, repeats, elapsed_ms,
                                             MSORB_DENSE_POPCOUNT);
}

int msorb_hamming_dense_top2_batch_ex(int device, const uint8_t* d_query, const uint8_t* d_train, const int* d_n_query,
                                      const int* d_n_train, int n_frames, int query_stride, int train_stride, int max_query,
                                      int max_train, int* d_best_idx, int* d_best_dist, int* d_second_dist, int repeats,
                                      float* elapsed_ms, int formulation) {
    if (formulation != MSORB_DENSE_MATRIX_CORES && formulation != MSORB_DENSE_POPCOUNT) return MSORB_E_INVALID;
    if (n_frames < 0 || query_stride < max_query || train_stride < max_train || max_train > 2048 || max_query < 0 ||
        (n_frames > 0 && (!d_query || !d_train || !d_n_query || !d_n_train || !d_best_idx || !d_best_dist || !d_second_dist)))
        return MSORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    // stream and timing events live with the calling thread (created once per device, released with the thread)
    struct Ctx {
        int device = -1;
        hipStream_t s = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        void release() {
            if (device >= 0 && hipSetDevice(device) == hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                if (s) (void)hipStreamDestroy(s);
            }
            s = nullptr; e0 = e1 = nullptr; device = -1;
        }
        ~Ctx() { release(); }
    };
    static thread_local Ctx ctx;
    if (ctx.device != device) {
        ctx.release();
        ctx.device = device;
        if (hipStreamCreateWithFlags(&ctx.s, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx.e0) != hipSuccess ||
            hipEventCreate(&ctx.e1) != hipSuccess) {
            ctx.release();
            set_last_error("stream / event creation failed");
            return MSORB_E_HIP;
        }
    }
    hipStream_t s = ctx.s;
    hipError_t e = hipEventRecord(ctx.e0, s);
    for (int r = 0; e == hipSuccess && r < (repeats > 0 ? repeats : 1); r++)
        launch_dense_top2(d_query, d_train, d_n_query, d_n_train, n_frames, query_stride, train_stride, max_query, max_train,
                          d_best_idx, d_best_dist, d_second_dist, s, formulation);
    if (e == hipSuccess) e = hipEventRecord(ctx.e1, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx.e0, ctx.e1);
    if (elapsed_ms) *elapsed_ms = ms;
    if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return MSORB_E_HIP; }
    return MSORB_OK;
}

int msorb_stereo_matches(msorb_extractor* left, msorb_extractor* right, const msorb_keypoint* kpsL, int nL,
                         const uint8_t* descL, const msorb_keypoint* kpsR, int nR, const uint8_t* descR, float mb,
                         float mbf, float* u_right, float* depth, int* n_oob) {
    if (!left || !right || nL < 0 || nR < 0 || (nL > 0 && (!kpsL || !descL || !u_right || !depth)) ||
        (nR > 0 && (!kpsR || !descR)))
        return MSORB_E_INVALID;
    PyramidView pl, pr;
    LevelScale sc;
    float inv_scale[MSORB_MAX_LEVELS];
    int devL = 0, devR = 0;
    hipStream_t s = nullptr, s2 = nullptr;
    int rc;
    if ((rc = extractor_last_view(left, &pl, &sc, inv_scale, &devL, &s, nullptr))) return rc;
    if ((rc = extractor_last_view(right, &pr, nullptr, nullptr, &devR, &s2, nullptr))) return rc;
    if (pl.nlevels != pr.nlevels || pl.lv[0].w != pr.lv[0].w || pl.lv[0].h != pr.lv[0].h) {
        set_last_error("left/right pyramids differ in geometry");
        return MSORB_E_INVALID;
    }
    if (n_oob) *n_oob = 0;
    if (nL == 0) return MSORB_OK;
    HIPCHK(hipSetDevice(devL));
    // One extractor object per GPU (MSORB_DEVICES=0,1: left eye on device A, right eye on device B): the right pyramid of the
    // frame — 1.5 MB for KITTI — is pulled to the left device, level by level, peer to peer (xGMI), after the right handle's
    // stream has drained; the association then runs on the left device as usual.  A left handle created under
    // MSORB_FORCE_PEER_PYRAMID takes this path with both handles on one device (test hook: a peer copy device -> same device is
    // an ordinary copy).
    if (devL != devR || extractor_force_peer_pyramid(left)) {
        struct PeerPyr {
            int device = -1;
            DBuf<uint8_t> buf;
            ~PeerPyr() { if (device >= 0 && hipSetDevice(device) == hipSuccess) buf.release(); }
        };
        static thread_local PeerPyr pp;
        if (pp.device != devL) { pp.buf.release(); pp.device = devL; }
        size_t off[MSORB_MAX_LEVELS], total = 0;
        for (int l = 0; l < pr.nlevels; l++) { off[l] = total; total += ((size_t)pr.lv[l].pitch * pr.lv[l].h + 255) & ~(size_t)255; }
        if ((rc = pp.buf.ensure(total + 256))) return rc;
        HIPCHK(hipSetDevice(devR));
        HIPCHK(hipStreamSynchronize(s2));          // the right eye's extraction (its own thread has returned, normally a no-op)
        HIPCHK(hipSetDevice(devL));
        for (int l = 0; l < pr.nlevels; l++) {
            HIPCHK(hipMemcpyPeerAsync(pp.buf.p + off[l], devL, pr.lv[l].base, devR, (size_t)pr.lv[l].pitch * pr.lv[l].h, s));
            pr.lv[l].base = pp.buf.p + off[l];
        }
    }
    for (int i = 0; i < nL; i++)
        if (kpsL[i].octave < 0 || kpsL[i].octave >= pl.nlevels) return MSORB_E_INVALID;
    for (int i = 0; i < nR; i++)
        if (kpsR[i].octave < 0 || kpsR[i].octave >= pl.nlevels) return MSORB_E_INVALID;
    // scratch kept per calling thread and device (this runs once per frame: allocating and freeing six device buffers per
    // call cost more than the kernel); inputs and outputs go through pinned staging
    struct Scratch {
        int device = -1;
        DBuf<uint8_t> d_in, d_out;
        HBuf<uint8_t> h_in, h_out;
        ~Scratch() {
            if (device >= 0 && hipSetDevice(device) == hipSuccess) { d_in.release(); d_out.release(); h_in.release(); h_out.release(); }
        }
    };
    static thread_local Scratch scr;
    if (scr.device != devL) { scr.d_in.release(); scr.d_out.release(); scr.h_in.release(); scr.h_out.release(); scr.device = devL; }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_kl = 0, o_dl = up16(o_kl + (size_t)nL * sizeof(msorb_keypoint)), o_kr = up16(o_dl + (size_t)nL * 32),
                 o_dr = up16(o_kr + (size_t)nR * sizeof(msorb_keypoint)), in_bytes = up16(o_dr + (size_t)nR * 32);
    const size_t out_bytes = (size_t)(3 * nL + 1) * 4;  // u_right, depth, sad[nL] + n_oob
    if ((rc = scr.d_in.ensure(in_bytes)) || (rc = scr.h_in.ensure(in_bytes)) || (rc = scr.d_out.ensure(out_bytes)) ||
        (rc = scr.h_out.ensure(out_bytes)))
        return rc;
    std::memcpy(scr.h_in.p + o_kl, kpsL, (size_t)nL * sizeof(msorb_keypoint));
    std::memcpy(scr.h_in.p + o_dl, descL, (size_t)nL * 32);
    if (nR) {
        std::memcpy(scr.h_in.p + o_kr, kpsR, (size_t)nR * sizeof(msorb_keypoint));
        std::memcpy(scr.h_in.p + o_dr, descR, (size_t)nR * 32);
    }
    float* d_ur = reinterpret_cast<float*>(scr.d_out.p);
    int* d_sad = reinterpret_cast<int*>(scr.d_out.p) + 2 * nL;
    hipError_t e = small_copy(scr.d_in.p, scr.h_in.p, in_bytes, hipMemcpyHostToDevice, s);   // (pinned blocks: the copy kernel, as for the frames)
    if (e == hipSuccess) e = hipMemsetAsync(d_sad + nL, 0, sizeof(int), s);
    if (e == hipSuccess) {
        StereoArgs a{};
        a.kpL = reinterpret_cast<const msorb_keypoint*>(scr.d_in.p + o_kl);
        a.kpR = reinterpret_cast<const msorb_keypoint*>(scr.d_in.p + o_kr);
        a.descL = scr.d_in.p + o_dl; a.descR = scr.d_in.p + o_dr;
        a.nL = nL; a.nR = nR; a.rows0 = pl.lv[0].h;
        for (int l = 0; l < pl.nlevels; l++) {
            a.pyrL[l] = pl.lv[l].base; a.pyrR[l] = pr.lv[l].base;
            a.pitchL[l] = pl.lv[l].pitch; a.pitchR[l] = pr.lv[l].pitch;
            a.rows[l] = pl.lv[l].h; a.cols[l] = pl.lv[l].w;
            a.scale[l] = sc.scale[l]; a.inv_scale[l] = inv_scale[l];
        }
        a.mb = mb; a.mbf = mbf;
        a.u_right = d_ur; a.depth = d_ur + nL; a.sad = d_sad; a.n_oob = d_sad + nL;
        launch_stereo_match(a, s);
        e = small_copy(scr.h_out.p, scr.d_out.p, out_bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return MSORB_E_HIP; }
    std::memcpy(u_right, scr.h_out.p, (size_t)nL * sizeof(float));
    std::memcpy(depth, scr.h_out.p + (size_t)nL * 4, (size_t)nL * sizeof(float));
    const int* sad = reinterpret_cast<const int*>(scr.h_out.p) + 2 * nL;
    if (n_oob) *n_oob = sad[nL];
    // median-based rejection, Frame.cc:899-912 (serial; vDistIdx is built in ascending iL order)
    std::vector<std::pair<int, int>> vDistIdx;
    for (int i = 0; i < nL; i++)
        if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
    if (vDistIdx.empty()) return MSORB_OK;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        u_right[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
    }
    return MSORB_OK;
}

// Shared body of the two batch forms: eyes interleaved in one extract batch (right == nullptr) or in two batches.
static int stereo_batch_impl(msorb_extractor* left, msorb_extractor* right, int n_pairs, const msorb_keypoint* d_kps_left,
                             const uint8_t* d_desc_left, const int* d_counts_left, const msorb_keypoint* d_kps_right,
                             const uint8_t* d_desc_right, const int* d_counts_right, int capacity, int max_left, float mb,
                             float mbf, float* d_u_right, float* d_depth, int* d_n_oob, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (!left || n_pairs < 0 || capacity <= 0 || max_left < 0 || max_left > capacity ||
        (n_pairs > 0 && (!d_kps_left || !d_desc_left || !d_counts_left || !d_u_right || !d_depth)) ||
        (right && n_pairs > 0 && (!d_kps_right || !d_desc_right || !d_counts_right)))
        return MSORB_E_INVALID;
    if (n_pairs == 0 || max_left == 0) return MSORB_OK;
    PyramidView pv, pr;
    LevelScale sc;
    float inv_scale[MSORB_MAX_LEVELS];
    int dev = 0, n_images = 0, dev_r = 0, n_images_r = 0;
    hipStream_t s = nullptr, s_r = nullptr;
    int rc;
    if ((rc = extractor_last_view(left, &pv, &sc, inv_scale, &dev, &s, &n_images))) return rc;
    const int step = right ? 1 : 2;
    if (right) {
        if ((rc = extractor_last_view(right, &pr, nullptr, nullptr, &dev_r, &s_r, &n_images_r))) return rc;
        if (dev_r != dev) { set_last_error("stereo matching needs both pyramids on one device"); return MSORB_E_INVALID; }
        if (pv.nlevels != pr.nlevels || pv.lv[0].w != pr.lv[0].w || pv.lv[0].h != pr.lv[0].h) {
            set_last_error("left/right pyramids differ in geometry");
            return MSORB_E_INVALID;
        }
    }
    if (step * n_pairs > n_images || (right && n_pairs > n_images_r)) {
        set_last_error("stereo_matches_batch: the last extract call of the handle(s) holds fewer images than n_pairs needs");
        return MSORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(dev));
    struct Scratch {
        int device = -1;
        DBuf<int> sad, oob, row_begin, row_list;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            sad.release(); oob.release(); row_begin.release(); row_list.release();
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            e0 = e1 = nullptr; device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch scr;
    if (scr.device != dev) {
        scr.release();
        scr.device = dev;
        HIPCHK(hipEventCreate(&scr.e0));
        HIPCHK(hipEventCreate(&scr.e1));
    }
    // a right keypoint enters rows floor(y - r) .. ceil(y + r), r = 2*scale[octave] (:762-768): at most 2r + 3 rows
    float smax = 0;
    for (int l = 0; l < pv.nlevels; l++) smax = std::max(smax, sc.scale[l]);
    const int row_cap = capacity * ((int)std::ceil(4.0f * smax) + 3);
    const int rows0 = pv.lv[0].h;
    if ((size_t)(2 * rows0 + 1) * sizeof(int) > 60000) {
        set_last_error("stereo_matches_batch: image too tall for the row table in LDS");
        return MSORB_E_INVALID;
    }
    if ((rc = scr.sad.ensure((size_t)n_pairs * capacity)) || (rc = scr.oob.ensure((size_t)n_pairs)) ||
        (rc = scr.row_begin.ensure((size_t)n_pairs * (rows0 + 1))) || (rc = scr.row_list.ensure((size_t)n_pairs * row_cap * 2)))
        return rc;
    int* oob = d_n_oob ? d_n_oob : scr.oob.p;
    StereoBatchArgs b{};
    b.pair_step = step;
    b.A.kpL = d_kps_left;
    b.A.descL = d_desc_left;
    b.A.kpR = right ? d_kps_right : d_kps_left + capacity;
    b.A.descR = right ? d_desc_right : d_desc_left + (size_t)capacity * 32;
    b.countsL = d_counts_left;
    b.countsR = right ? d_counts_right : d_counts_left + 1;
    b.A.rows0 = pv.lv[0].h;
    for (int l = 0; l < pv.nlevels; l++) {
        b.A.pyrL[l] = pv.lv[l].base;
        b.A.pyrR[l] = right ? pr.lv[l].base : pv.lv[l].base + pv.lv[l].img_stride;
        b.A.pitchL[l] = pv.lv[l].pitch;
        b.A.pitchR[l] = right ? pr.lv[l].pitch : pv.lv[l].pitch;
        b.A.rows[l] = pv.lv[l].h; b.A.cols[l] = pv.lv[l].w;
        b.A.scale[l] = sc.scale[l]; b.A.inv_scale[l] = inv_scale[l];
        b.img_strideL[l] = pv.lv[l].img_stride;
        b.img_strideR[l] = right ? pr.lv[l].img_stride : pv.lv[l].img_stride;
    }
    b.A.mb = mb; b.A.mbf = mbf;
    b.A.u_right = d_u_right; b.A.depth = d_depth; b.A.sad = scr.sad.p; b.A.n_oob = oob;
    b.capacity = capacity;
    b.row_begin = scr.row_begin.p; b.row_list = reinterpret_cast<int2*>(scr.row_list.p); b.row_cap = row_cap;
    if (right) HIPCHK(hipStreamSynchronize(s_r));  // the right handle's pyramid was built on its own stream
    HIPCHK(hipMemsetAsync(oob, 0, (size_t)n_pairs * sizeof(int), s));
    HIPCHK(hipEventRecord(scr.e0, s));
    launch_stereo_match_batch(b, n_pairs, max_left, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(scr.e1, s));
    HIPCHK(hipStreamSynchronize(s));
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1));
    return MSORB_OK;
}

int msorb_stereo_matches_batch(msorb_extractor* h, int n_pairs, const msorb_keypoint* d_keypoints,
                               const uint8_t* d_descriptors, int capacity, const int* d_counts, int max_left, float mb,
                               float mbf, float* d_u_right, float* d_depth, int* d_n_oob, float* elapsed_ms) {
    return stereo_batch_impl(h, nullptr, n_pairs, d_keypoints, d_descriptors, d_counts, nullptr, nullptr, nullptr, capacity,
                             max_left, mb, mbf, d_u_right, d_depth, d_n_oob, elapsed_ms);
}

int msorb_stereo_matches_split(msorb_extractor* left, msorb_extractor* right, int n_pairs, const msorb_keypoint* d_kps_left,
                               const uint8_t* d_desc_left, const int* d_counts_left, const msorb_keypoint* d_kps_right,
                               const uint8_t* d_desc_right, const int* d_counts_right, int capacity, int max_left, float mb,
                               float mbf, float* d_u_right, float* d_depth, int* d_n_oob, float* elapsed_ms) {
    if (!right) return MSORB_E_INVALID;
    return stereo_batch_impl(left, right, n_pairs, d_kps_left, d_desc_left, d_counts_left, d_kps_right, d_desc_right,
                             d_counts_right, capacity, max_left, mb, mbf, d_u_right, d_depth, d_n_oob, elapsed_ms);
}

}  // extern "C"
