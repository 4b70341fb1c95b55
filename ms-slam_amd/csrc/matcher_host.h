// Internals shared by the host-side matcher sources (matcher_host.hip, track.hip): the frame handle (features + 64x48
// grid on the device), grow-only device / pinned buffers, the error macro.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "matcher_device.h"
#include "orb_device.h"

namespace msorb {
void set_last_error(const std::string& s);
int extractor_last_view(msorb_extractor* h, PyramidView* pyr, LevelScale* sc, float* inv_scale, int* device,
                        hipStream_t* stream, int* n_images);
int extractor_device(const msorb_extractor* h);
int extractor_levels(const msorb_extractor* h);
bool extractor_force_peer_pyramid(const msorb_extractor* h);   // MSORB_FORCE_PEER_PYRAMID at the handle's creation (test hook)
}  // namespace msorb
struct msorb_frame_track;
namespace msorb {
void frame_track_release(msorb_frame* f);
int frame_host_grid(msorb_frame* f);
int frame_grid_max_keypoints();        // largest keypoint count the device grid takes on the current device (-1: query failed)
void frame_invalidate(msorb_frame* f);  // a failed set leaves the handle empty, never half new / half old
// sets the frame's device side (train arrays + grid) from device arrays: enqueue only, on stream s (track.hip)
// the motion-model projection that rides the frame's grid launch (track.hip frame_grid_kernel): `prepare` fills the kernel's
// LastFrameArgs (behind `args`) once the frame's fields are set, right before the launch
struct LastFrameProjector {
    int (*prepare)(void* ctx, hipStream_t s, void* args);
    void* ctx;
};
int enqueue_frame_from_device(msorb_frame* f, hipStream_t s, const msorb_keypoint* d_kps, const uint8_t* d_desc,
                              const float* d_u_right, const int* d_count, int n_fixed, int n_cap, float min_x, float max_x,
                              float min_y, float max_y, const float* scale_factors, int nlevels, const LastFrameProjector* proj = nullptr);
}  // namespace msorb

#define HIPCHK(expr)                                                               \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            msorb::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
            return MSORB_E_HIP;                                                    \
        }                                                                          \
    } while (0)

namespace msorb {
template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return MSORB_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T) + 16));   // + 16: small_copy moves whole 16-byte units
        n = std::max<size_t>(count, 1);
        return MSORB_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};
// pinned host staging: hipMemcpyAsync from / to pageable memory makes the driver stage and synchronise per call
template <typename T>
struct HBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return MSORB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipHostMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T) + 16, hipHostMallocDefault));
        n = std::max<size_t>(count, 1);
        return MSORB_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};
}  // namespace msorb

struct msorb_frame {
    int device = 0;
    hipStream_t stream = nullptr;
    int N = 0, nlevels = 0;
    float minX = 0, minY = 0, maxX = 0, maxY = 0, gridWInv = 0, gridHInv = 0;
    std::vector<msorb_keypoint> kps;
    std::vector<float> u_right, scale;
    std::vector<int> cell_begin, cell_idx;
    msorb::DBuf<msorb::KpLite> d_kp;
    msorb::DBuf<uint8_t> d_desc, d_occ, d_qdesc, d_stage, d_win;   // d_win: queries | descriptors | occupancy of a host-fed window search, one upload
    msorb::DBuf<int> d_cell_begin, d_cell_idx, d_n;
    msorb::DBuf<int> d_init_cnt, d_init_beg;   // msorb_search_for_initialization: candidate counts / list offsets / lists
    msorb::DBuf<int2> d_init_list;
    int last_rounds = 0;                // device rounds of the last claim-replaying search (msorb_frame_search_rounds)
    long long total_rounds = 0, total_searches = 0;
    bool host_grid_valid = false;       // cell_begin / cell_idx (host) mirror the device grid
    std::mutex grid_mu;                 // the lazy fetch of that mirror (msorb_frame_features_in_area is a const query)
    msorb_frame_track* track = nullptr;  // staging of the local-points chain (track.hip)
    msorb::DBuf<msorb::WinQuery> d_q;
    msorb::DBuf<msorb::TopK> d_topk;
    msorb::HBuf<uint8_t> h_in;    // queries + query descriptors + occupancy, staged
    msorb::HBuf<msorb::TopK> h_topk;
    msorb::FrameView view() const {
        msorb::FrameView v;
        v.kp = d_kp.p; v.desc = d_desc.p; v.cell_begin = d_cell_begin.p; v.cell_idx = d_cell_idx.p;
        v.occupied = d_occ.p; v.minX = minX; v.minY = minY; v.gridWInv = gridWInv; v.gridHInv = gridHInv; v.n = N;
        for (int l = 0; l < MSORB_MAX_LEVELS; l++) v.inv_sigma2[l] = 0.0f;
        v.gate_kp = nullptr;
        return v;
    }
};


namespace msorb {

// Shared replay driver of the claiming window searches (SearchByProjection forms, ORBmatcher.cc:88-90,129).
// The device computes, for every query at once, the kTopK best candidates against an occupancy SNAPSHOT; the host replays
// the accept rules in query order, dropping candidates claimed since the snapshot.  If a query's list is exhausted or a
// keypoint was freed (mbSparsified bypass), the snapshot is refreshed and the kernel re-run from that query on.
//   q / qdesc   host queries + descriptors to upload; nullptr: f->d_q / f->d_qdesc already hold them (built on the device)
//   flags       kQValid / kQSkipOccupied per query; nullptr: taken from q
//   ready       round 0 has been run by the caller: f->d_occ holds `occ`, f->h_topk[0, M) the lists (stream synchronised)
//   d_qdesc     device query descriptors when they do not live in f->d_qdesc (read by the re-runs)
//   lanes       lanes per query of the window kernel (0: chosen from the mean radius of the host queries)
// accept(q, idx, dist, n, &new_occ) is called in query order with the query's exact candidate prefix (>= need entries unless
// the true candidate set is smaller); it returns the keypoint index it assigned (or -1) and that keypoint's new occupancy.
template <typename Accept>
int run_window_search(msorb_frame* f, int M, const WinQuery* q, const uint8_t* flags, const uint8_t* qdesc,
                      std::vector<uint8_t>& occ, int need, Accept accept, bool ready = false, int* rounds_out = nullptr,
                      const uint8_t* d_qdesc = nullptr, int lanes = 0) {
    if (rounds_out) *rounds_out = 0;
    if (M <= 0 || f->N <= 0) return MSORB_OK;   // no queries / no train keypoints (also a handle whose set failed): no match
    int rc;
    if ((rc = f->d_q.ensure(M)) || (!d_qdesc && (rc = f->d_qdesc.ensure((size_t)M * 32))) || (rc = f->d_topk.ensure(M)) ||
        (rc = f->d_occ.ensure(f->N)))
        return rc;
    // host queries + host descriptors (the class paths: ORBmatcher::SearchByProjection with unchanged callers): queries, descriptors
    // and the occupancy snapshot are staged side by side and go up as ONE copy into one device block (three copy launches before:
    // each costs the host ~6 us to issue and the chain ~3 us)
    const bool one_block = q && qdesc && !d_qdesc;
    if (!d_qdesc) d_qdesc = f->d_qdesc.p;
    hipStream_t s = f->stream;
    const size_t qb = q ? (size_t)M * sizeof(WinQuery) : 0, db = qdesc ? (size_t)M * 32 : 0;
    const size_t qb16 = (qb + 15) & ~(size_t)15;
    if ((rc = f->h_in.ensure(qb16 + db + (size_t)f->N + 64)) || (rc = f->h_topk.ensure(M))) return rc;
    if (one_block && (rc = f->d_win.ensure(qb16 + db + (size_t)f->N + 64))) return rc;
    const WinQuery* const q_dev = one_block ? reinterpret_cast<const WinQuery*>(f->d_win.p) : f->d_q.p;
    if (one_block) d_qdesc = f->d_win.p + qb16;
    uint8_t* const occ_dev = one_block ? f->d_win.p + qb16 + db : f->d_occ.p;
    std::vector<uint8_t> flags_own;
    if (!flags) {
        if (!q) return MSORB_E_INVALID;
        flags_own.resize(M);
        for (int i = 0; i < M; i++) flags_own[i] = q[i].flags;
        flags = flags_own.data();
    }
    if (lanes == 0) {   // lanes per query from the mean window radius of the valid queries
        lanes = 16;
        if (q) {
            double sum = 0;
            int nv = 0;
            for (int i = 0; i < M; i++)
                if (q[i].flags & kQValid) { sum += q[i].r; nv++; }
            if (nv) lanes = window_lanes_for((float)(sum / nv), f->gridWInv, f->gridHInv);
        }
    }
    if (one_block) {
        std::memcpy(f->h_in.p, q, qb);
        std::memcpy(f->h_in.p + qb16, qdesc, db);
    } else {
        if (q) {
            std::memcpy(f->h_in.p, q, qb);
            HIPCHK(small_copy(f->d_q.p, f->h_in.p, qb, hipMemcpyHostToDevice, s));
        }
        if (qdesc) {
            std::memcpy(f->h_in.p + qb16, qdesc, db);
            HIPCHK(small_copy(f->d_qdesc.p, f->h_in.p + qb16, db, hipMemcpyHostToDevice, s));
        }
    }
    uint8_t* const h_occ = f->h_in.p + qb16 + db;
    TopK* const topk = f->h_topk.p;
    std::vector<int8_t> diff(f->N, 0);  // occupancy now vs snapshot: +1 claimed since, -1 freed since
    // a keypoint that was occupied at the snapshot and is free now is missing from the lists of exactly those queries whose window
    // (box and level band as window_topk_kernel tests them; its mvuRight test can only drop more) holds it: only they need a new round
    std::vector<int> freed;
    auto holds_freed = [&](const WinQuery& w) {
        for (int idx : freed) {
            if (diff[idx] >= 0) continue;
            const msorb_keypoint& kp = f->kps[idx];
            if (kp.octave < w.min_level || (w.max_level >= 0 && kp.octave > w.max_level)) continue;
            if (fabsf(kp.x - w.x) < w.r && fabsf(kp.y - w.y) < w.r) return true;
        }
        return false;
    };
    int q0 = 0, n_rounds = 0;
    while (q0 < M) {
        if (!(ready && n_rounds == 0)) {
            if (f->N) std::memcpy(h_occ, occ.data(), f->N);  // the previous round's copy has completed (stream synchronised below)
            if (one_block && n_rounds == 0)
                HIPCHK(small_copy(f->d_win.p, f->h_in.p, qb16 + db + (size_t)f->N, hipMemcpyHostToDevice, s));   // queries | descriptors | occupancy
            else if (f->N)
                HIPCHK(small_copy(occ_dev, h_occ, f->N, hipMemcpyHostToDevice, s));
            FrameView view = f->view();
            view.occupied = occ_dev;
            launch_window_topk(view, q_dev, d_qdesc, q0, M, f->d_topk.p, s, 1, 0, 0, nullptr, lanes);
            HIPCHK(small_copy(topk + q0, f->d_topk.p + q0, (size_t)(M - q0) * sizeof(TopK), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        std::vector<uint8_t> snap = occ;
        std::fill(diff.begin(), diff.end(), 0);
        int n_freed = 0;
        freed.clear();
        int qi = q0;
        bool resync = false;
        for (; qi < M; qi++) {
            if (!(flags[qi] & kQValid)) continue;
            const bool skip = flags[qi] & kQSkipOccupied;
            if (skip && n_freed > 0 && qi > q0 && (!q || holds_freed(q[qi]))) { resync = true; break; }   // (queries built on the device: windows unknown here)
            const TopK& t = topk[qi];
            int idx[kTopK], dist[kTopK], n = 0, n_dev = 0;
            for (int k = 0; k < kTopK; k++) {
                if (t.idx[k] < 0) break;
                n_dev++;
                if (skip && diff[t.idx[k]] > 0) continue;
                idx[n] = t.idx[k]; dist[n] = t.dist[k]; n++;
            }
            if (n < need && n < n_dev && n_dev == kTopK && qi > q0) { resync = true; break; }
            int new_occ = 0;
            const int assigned = accept(qi, idx, dist, n, &new_occ);
            if (assigned >= 0) {
                occ[assigned] = (uint8_t)new_occ;
                const int8_t d = (int8_t)((int)occ[assigned] - (int)snap[assigned]);
                if (diff[assigned] < 0) n_freed--;
                diff[assigned] = d;
                if (d < 0) { n_freed++; freed.push_back(assigned); }
            }
        }
        n_rounds++;
        if (!resync) break;
        q0 = qi;
    }
    if (rounds_out) *rounds_out = n_rounds;
    f->last_rounds = n_rounds; f->total_rounds += n_rounds; f->total_searches++;
    return MSORB_OK;
}

}  // namespace msorb
