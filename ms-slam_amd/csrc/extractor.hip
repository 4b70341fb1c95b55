// libmsorb.so — extractor handle, host orchestration and the C ABI of include/msorb.h.
//
// Per call (one stream, n_images same-sized frames):
//   pyramid (nlevels-1 launches) -> FAST cells -> candidate compaction -> [copy stream: D2H candidates]
//   -> Gaussian blur (overlaps the host stage) -> host quadtree selection (thread pool, one task per image)
//   -> H2D selection -> IC-angle + rBRIEF -> keypoints/descriptors in device memory.
// The quadtree (DistributeOctTree, ORBextractor.cc:555-779) is serial and order-defining; it stays on the
// host (orb_host.cc) in this version.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "orb_device.h"
#include "gauss7_stream_device.h"
#include "matcher_device.h"
#include "stereo_rowtable_device.h"

using namespace msorb;

namespace {

thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                             \
            return MSORB_E_HIP;                                                                       \
        }                                                                                             \
    } while (0)

const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

// Minimal persistent worker pool: parallel_for(n, fn) runs fn(i) for i in [0,n) on the workers + caller.
class Pool {
    struct Job {
        const std::function<void(int)>* fn;
        int n;
        std::atomic<int> next{0}, done{0};
    };

public:
    explicit Pool(int nthreads) {
        for (int i = 0; i < nthreads - 1; i++) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        {
            std::lock_guard<std::mutex> lk(m_);
            cur_ = job;
            gen_++;
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return job->done.load() >= n; });
        cur_.reset();
    }

private:
    void work(Job& job) {
        for (;;) {
            const int i = job.next.fetch_add(1);
            if (i >= job.n) break;
            (*job.fn)(i);
            if (job.done.fetch_add(1) + 1 == job.n) {
                std::lock_guard<std::mutex> lk(m_);
                done_cv_.notify_all();
            }
        }
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = cur_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::shared_ptr<Job> cur_;
    unsigned gen_ = 0;
    bool stop_ = false;
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return MSORB_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipMalloc((void**)&p, count * sizeof(T) + 16));   // + 16: small_copy moves whole 16-byte units
        n = count;
        return MSORB_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return MSORB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipHostMalloc((void**)&p, count * sizeof(T) + 16, hipHostMallocDefault));
        n = count;
        return MSORB_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

}  // namespace

constexpr int kMaxGroups = 4;
struct StreamGroup {  // one sub-batch pipeline: main stream (group 0 uses the handle's own), sync + timing events
    hipStream_t s = nullptr;
    bool own_stream = false;
    hipEvent_t ev_pyr = nullptr, ev_blur = nullptr, ev_fast = nullptr;
    hipEvent_t pe[10] = {};
    bool ready = false;
};

struct msorb_extractor {
    int device = 0;
    StreamGroup grp[kMaxGroups];
    int n_groups = 2;
    bool overlap_blur = true;
    OrbParams P;
    Semantics sem;   // msorb_extractor_set_semantics: variants of the [OpenCV-recall] primitives (defaults = SURVEY.md Appendix A)
    hipStream_t stream = nullptr, copy_stream = nullptr;
    // Environment switches, read ONCE when the handle is created (README.md lists them): MSORB_SERIAL_PIPELINE (one stream,
    // host-synchronised stages: debugging), MSORB_QUADTREE=host (DistributeOctTree on the host twin), MSORB_HOST_THREADS (its
    // worker threads), MSORB_SPLIT_NO_PEER (test hook: the two-device gather staged through the host), MSORB_FORCE_PEER_PYRAMID
    // (test hook: msorb_stereo_matches pulls the right pyramid over the peer path even on one device)
    const StereoRowJob* row_job = nullptr;   // set by the stereo-frame calls around run_pipeline: the right eye's band records leave the layout launch
    struct Knobs { bool serial_pipeline = false, quadtree_host = false, quadtree_global = false, split_no_peer = false, force_peer_pyramid = false, frame_compact = true; int host_threads = 0, frame_fuse = 2; } knobs;
    int lds_per_block = 64 * 1024;   // hipDeviceAttributeMaxSharedMemoryPerBlock of the handle's device
    static constexpr bool capturing = false;   // (no graph capture: plain launches; see tools/experiments/README.md)
    bool defer_sync = false;     // enqueue only, the caller appends more work and synchronises (msorb_extract[_stereo])
    bool last_prof = false;      // the stage events of the last run_pipeline_groups() call were recorded
    int pending_batch = 0;       // images of a batch enqueued by msorb_extract_batch_submit and not yet waited for
    bool skip_count_copies = false;  // with defer_sync: the caller fetches the counts from the device itself
    DevBuf<int> d_st_sad, d_st_rows, d_st_list;  // stereo association scratch of msorb_extract_stereo
    DevBuf<uint8_t> d_st_block, d_st_img;        // its output block and its two level-0 planes
    DevBuf<uint8_t> d_out1;                      // msorb_extract: keypoints + descriptors of one frame as one block
    DevBuf<uint8_t> d_gather_pyr;                // msorb_extract_stereo_split: the right eye's pyramid, gathered onto this (left) device
    DevBuf<int> d_gather_cnt;                    // ... and its keypoint count
    hipEvent_t ev_split = nullptr;               // ... recorded on the right handle's stream after the gather copies
    // mvImagePyramid for callers that read it on the host (unchanged Frame::ComputeStereoMatches, Frame.cc:840-855): levels
    // 1.. are copied to pinned memory on a stream of their own as soon as the pyramid kernels are done — the copy (1 MB
    // for KITTI) rides PCIe while FAST / quadtree / describe run; level 0 is the staged input image itself
    bool host_pyramid = false;
    hipStream_t pyr_stream = nullptr;
    hipEvent_t ev_pyr_done = nullptr;
    bool h_pyr_async = false;  // h_pyr holds levels 1.. of the last msorb_extract call, level 0 = h_img_pin
    const uint8_t* pair_l0[2] = {nullptr, nullptr};   // host level 0 of the two images of the last msorb_extract_pair call
    int pair_request = 0;      // one-shot: set by msorb_extract_pair around its run_pipeline call
    int pair_pyramids = 0;     // msorb_extract_pair: host pyramids of this many images are wanted / were copied (image i at h_pyr + i * pyramid_bytes)
    unsigned long long buffers_epoch = 0;  // bumped whenever a device / pinned buffer may have moved
    hipEvent_t ev_compact = nullptr, ev_pyramid = nullptr, ev_blur = nullptr;
    hipEvent_t pe[10] = {};
    bool profiling = false;
    float stage_ms[MSORB_N_STAGES] = {};

    FrameGeom G;
    bool geom_valid = false;
    std::vector<int> level_cell_begin;
    LevelScale scales;

    // device state
    DevBuf<uint8_t> d_pyr, d_blur, d_desc1;
    DevBuf<ResizeTap> d_taps;
    TowerPlan tower;           // the pyramid of a frame or two as one launch (orb_device.h); ntx == 0: not available for this geometry
    std::vector<size_t> tap_x_off, tap_y_off;
    DevBuf<CellDesc> d_cells;
    DevBuf<int> d_level_cell_begin, d_cell_count, d_cell_off, d_level_count, d_img_total, d_img_base, d_sel_count;
    DevBuf<Cand16> d_slots, d_compact;
    DevBuf<SelRec> d_sel;
    DevBuf<msorb_keypoint> d_kps1;
    DevBuf<uint16_t> d_label;
    DevBuf<int> d_sel_pt, d_sel_n, d_mono;
    QtLevels qt{};
    bool device_quadtree = true;
    bool qt_global = false;   // the selection's workspace lives in global memory (d_qt_ws): quotas beyond a workgroup's LDS, or MSORB_QT_GLOBAL=1
    DevBuf<char> d_qt_ws;
    DevBuf<uint32_t> d_label_wide;   // the global form's 32-bit candidate labels (one per candidate slot, like d_label)
    bool small_cells = false;  // every cell ROI <= 46 x 57: the FAST kernel's compact LDS geometry applies
    bool compact_on_host = false;  // h_compact / h_level_count / h_img_base hold the last call's candidates
    bool compact_fixed_stride = false;  // the last call left image i's candidates at i * slots_per_image on the device (device pipeline)
    // pinned host state
    PinBuf<int> h_level_count, h_img_base, h_sel_count, h_mono;
    PinBuf<Cand16> h_compact;
    PinBuf<SelRec> h_sel;
    PinBuf<uint8_t> h_pyr, h_img_pin, h_out_pin, h_gather;
    bool h_pyr_valid = false;

    // last call
    PyramidView last_pyr{}, last_blur{};
    int last_n_images = 0;
    int last_groups = 1;
    int sel_stride = 0;

    std::unique_ptr<Pool> pool;
    std::vector<std::vector<int>> kept_scratch;  // per worker-task scratch is allocated inside tasks
};

namespace {

// Upper bound of the keypoints one image returns.  DistributeOctTree overshoots a level's quota by at most 3 (the split that
// reaches it), but its FIRST pass divides every initial column unconditionally (ORBextractor.cc:610-681 runs before any quota
// check): a level returns up to max(quota + 3, 4 * nIni) keypoints, nIni = round(width / height) <= 4 for every camera the
// reference is configured for.  16 more rows per level cover that whatever the quota (tiny nfeatures on wide images).
// per-frame transfer between a pinned block and device memory on stream s (orb_kernels.hip small_copy)
int frame_copy(msorb_extractor*, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    HIPCHK(small_copy(dst, src, bytes, kind, s));
    return MSORB_OK;
}
int capacity_of(const msorb_extractor* h) { return h->P.nfeatures + (3 + 16) * h->P.nlevels; }

int ensure_geometry(msorb_extractor* h, int rows, int cols) {
    if (h->geom_valid && h->G.rows == rows && h->G.cols == cols) return MSORB_OK;
    h->geom_valid = false;
    FrameGeom g;
    if (!g.build(h->P, rows, cols)) {
        set_error("image too small for the reference's 35-px cell grid at some pyramid level");
        return MSORB_E_GEOMETRY;
    }
    for (const CellDesc& c : g.cells)
        if (c.rw > 76 || c.rh > 76) { set_error("cell ROI larger than 76 px"); return MSORB_E_GEOMETRY; }
    h->G = g;
    h->small_cells = true;
    for (const CellDesc& c : g.cells) h->small_cells = h->small_cells && c.rw <= 46 && c.rh <= 57;
    // resize taps for levels 1..n-1
    std::vector<ResizeTap> all;
    std::vector<std::vector<ResizeTap>> taps_x(g.nlevels), taps_y(g.nlevels);
    h->tap_x_off.assign(g.nlevels, 0);
    h->tap_y_off.assign(g.nlevels, 0);
    for (int l = 1; l < g.nlevels; l++) {
        auto tx = make_resize_taps(g.lv[l].w, g.lv[l - 1].w, true);
        auto ty = make_resize_taps(g.lv[l].h, g.lv[l - 1].h, false);
        taps_x[l] = tx; taps_y[l] = ty;
        while (all.size() & 3) all.push_back(ResizeTap{0, 0, 0, 0});  // 32-byte aligned x tables (uint4 loads)
        h->tap_x_off[l] = all.size();
        all.insert(all.end(), tx.begin(), tx.end());
        while (all.size() & 3) all.push_back(tx.back());             // a group of 4 taps may run past the last column
        h->tap_y_off[l] = all.size();
        all.insert(all.end(), ty.begin(), ty.end());
    }
    for (int k = 0; k < 4; k++) all.push_back(ResizeTap{0, 0, 0, 0});   // (the tower's tap fetch may read one entry past a y table)
    {   // single frames: the whole pyramid as one launch (TowerPlan)
        int lw[kMaxLevels], lh[kMaxLevels], lp[kMaxLevels];
        for (int l = 0; l < g.nlevels; l++) { lw[l] = g.lv[l].w; lh[l] = g.lv[l].h; lp[l] = g.lv[l].pitch; }
        if (!build_tower_plan(h->tower, g.nlevels, lw, lh, lp, taps_x, taps_y, 150 * 1024)) h->tower = TowerPlan{};
    }
    int rc;
    if ((rc = h->d_taps.ensure(std::max<size_t>(all.size(), 1)))) return rc;
    if (!all.empty()) HIPCHK(hipMemcpy(h->d_taps.p, all.data(), all.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    if ((rc = h->d_cells.ensure(g.cells.size()))) return rc;
    HIPCHK(hipMemcpy(h->d_cells.p, g.cells.data(), g.cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    h->level_cell_begin.assign(g.nlevels + 1, 0);
    for (int l = 0; l < g.nlevels; l++) h->level_cell_begin[l] = g.lv[l].cell_begin;
    h->level_cell_begin[g.nlevels] = (int)g.cells.size();
    if ((rc = h->d_level_cell_begin.ensure(g.nlevels + 1))) return rc;
    HIPCHK(hipMemcpy(h->d_level_cell_begin.p, h->level_cell_begin.data(), (g.nlevels + 1) * sizeof(int),
                     hipMemcpyHostToDevice));
    int sel_off = 0;
    h->qt.nlevels = g.nlevels;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& lg = g.lv[l];
        h->qt.W[l] = lg.max_x - lg.min_x;
        h->qt.H[l] = lg.max_y - lg.min_y;
        h->qt.quota[l] = lg.quota;
        h->qt.n_ini[l] = (int)std::round(static_cast<float>(h->qt.W[l]) / h->qt.H[l]);
        h->qt.sel_off[l] = sel_off;
        sel_off += std::max(lg.quota, h->qt.n_ini[l]) + 8 + 4 * h->qt.n_ini[l];
    }
    h->sel_stride = sel_off;
    // the device quadtree keeps a level's workspace in one workgroup's LDS (160 KB per CU on gfx950: nfeatures up to ~8000); beyond
    // that the same selection runs over a workspace in global memory (quadtree_global_kernels.hip) — up to the 16-bit rank tables
    // of that workspace (4 N + 16 slots <= 65535: a level quota N <= 16 379, nfeatures ~75 000 at 1.2 / 8 levels; Tracking.cc:601's
    // 5 x nFeatures initialisation extractor asks for 6 000 - 10 000).  The host twin (orb_host.cc) is a checker
    // (MSORB_QUADTREE=host); quotas beyond the global form are refused when the geometry is set (MSORB_E_CAPACITY).
    int max_quota = 1;
    for (int l = 0; l < g.nlevels; l++) max_quota = std::max(max_quota, std::max(h->qt.quota[l], h->qt.n_ini[l]));
    const bool fits_lds = (long long)quadtree_lds_bytes(h->qt) + 10 * 1024 <= (long long)h->lds_per_block;
    const bool fits_labels = 4 * max_quota + 16 <= 65535;   // (the 16-bit rank tables of the workspace; labels are 32 bits in the global form)
    if (!h->knobs.quadtree_host && !fits_lds && !fits_labels) {
        set_error("a level quota of " + std::to_string(max_quota) + " keypoints is beyond the device selection (16 379 per level); MSORB_QUADTREE=host runs it on the host twin");
        return MSORB_E_CAPACITY;
    }
    h->device_quadtree = !h->knobs.quadtree_host;
    h->qt_global = h->device_quadtree && (!fits_lds || h->knobs.quadtree_global) && fits_labels;
    h->geom_valid = true;
    h->last_n_images = 0;
    return MSORB_OK;
}

int ensure_batch(msorb_extractor* h, int n_images) {
    const FrameGeom& g = h->G;
    const size_t ncells = g.cells.size();
    int rc;
    // +256: row-coherent dword loads may run a few bytes past the last row of the last plane
    if ((rc = h->d_pyr.ensure((size_t)n_images * g.pyramid_bytes + 256))) return rc;
    if ((rc = h->d_blur.ensure((size_t)n_images * g.blur_bytes + 512))) return rc;
    if ((rc = h->d_slots.ensure((size_t)n_images * g.slots_per_image))) return rc;
    if ((rc = h->d_compact.ensure((size_t)n_images * g.slots_per_image))) return rc;
    if ((rc = h->d_cell_count.ensure((size_t)n_images * ncells))) return rc;
    if ((rc = h->d_cell_off.ensure((size_t)n_images * ncells))) return rc;
    if ((rc = h->d_level_count.ensure((size_t)n_images * g.nlevels))) return rc;
    if ((rc = h->d_img_total.ensure(n_images))) return rc;
    if ((rc = h->d_img_base.ensure(n_images + 1 + kMaxGroups))) return rc;
    if ((rc = h->d_sel_count.ensure(std::max(n_images, 4)))) return rc;
    if ((rc = h->d_sel.ensure((size_t)n_images * h->sel_stride))) return rc;
    if ((rc = h->h_level_count.ensure((size_t)n_images * g.nlevels))) return rc;
    if ((rc = h->h_img_base.ensure(n_images + 1))) return rc;
    if ((rc = h->h_sel_count.ensure(std::max(n_images, 4)))) return rc;
    if ((rc = h->h_sel.ensure((size_t)n_images * h->sel_stride))) return rc;
    if ((rc = h->h_mono.ensure(std::max(n_images, 4)))) return rc;
    if ((rc = h->d_label.ensure((size_t)n_images * g.slots_per_image))) return rc;
    if ((rc = h->d_sel_pt.ensure((size_t)n_images * h->sel_stride))) return rc;
    if ((rc = h->d_sel_n.ensure((size_t)n_images * g.nlevels))) return rc;
    if (h->qt_global && (rc = h->d_qt_ws.ensure((size_t)n_images * g.nlevels * quadtree_global_workspace_stride(h->qt)))) return rc;
    if (h->qt_global && (rc = h->d_label_wide.ensure((size_t)n_images * g.slots_per_image))) return rc;
    if ((rc = h->d_mono.ensure(std::max(n_images, 4)))) return rc;
    return MSORB_OK;
}

PyramidView make_view(const msorb_extractor* h, const uint8_t* base, const LevelView* level0) {
    PyramidView v{};
    const FrameGeom& g = h->G;
    v.nlevels = g.nlevels;
    for (int l = 0; l < g.nlevels; l++) {
        v.lv[l].base = base + g.lv[l].plane_off;
        v.lv[l].img_stride = g.pyramid_bytes;
        v.lv[l].pitch = g.lv[l].pitch;
        v.lv[l].w = g.lv[l].w;
        v.lv[l].h = g.lv[l].h;
    }
    if (level0) v.lv[0] = *level0;
    return v;
}

PyramidView make_blur_view(const msorb_extractor* h, const uint8_t* base) {   // tiled planes (orb_device.h blur_tile_off)
    PyramidView v{};
    const FrameGeom& g = h->G;
    v.nlevels = g.nlevels;
    for (int l = 0; l < g.nlevels; l++) {
        v.lv[l].base = base + g.lv[l].blur_off;
        v.lv[l].img_stride = g.blur_bytes;
        v.lv[l].pitch = g.lv[l].pitch;
        v.lv[l].w = g.lv[l].w;
        v.lv[l].h = g.lv[l].h;
    }
    return v;
}

// Copy the compacted FAST candidates of the last launch to pinned host memory (host-quadtree mode, debug hook).
int fetch_candidates(msorb_extractor* h, int n_images) {
    const int nl = h->G.nlevels;
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_compact, 0));
    HIPCHK(hipMemcpyAsync(h->h_level_count.p, h->d_level_count.p, (size_t)n_images * nl * sizeof(int),
                          hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipMemcpyAsync(h->h_img_base.p, h->d_img_base.p, (size_t)(n_images + 1) * sizeof(int),
                          hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    int rc;
    if (h->compact_fixed_stride) {
        // the device pipeline leaves image i's run at i * slots_per_image: the host copy is packed all the same
        std::vector<int> tot(n_images, 0);
        int total = 0;
        for (int i = 0; i < n_images; i++) {
            for (int l = 0; l < nl; l++) tot[i] += h->h_level_count.p[(size_t)i * nl + l];
            total += tot[i];
        }
        if ((rc = h->h_compact.ensure(std::max<size_t>((size_t)total + total / 4, 1024)))) return rc;
        int at = 0;
        for (int i = 0; i < n_images; i++) {
            if (tot[i] > 0)
                HIPCHK(hipMemcpyAsync(h->h_compact.p + at, h->d_compact.p + (size_t)h->h_img_base.p[i], (size_t)tot[i] * sizeof(Cand16),
                                      hipMemcpyDeviceToHost, h->copy_stream));
            h->h_img_base.p[i] = at;
            at += tot[i];
        }
        h->h_img_base.p[n_images] = at;
        HIPCHK(hipStreamSynchronize(h->copy_stream));
        h->compact_on_host = true;
        return MSORB_OK;
    }
    const int total = h->h_img_base.p[n_images];
    if ((rc = h->h_compact.ensure(std::max<size_t>((size_t)total + total / 4, 1024)))) return rc;
    if (total > 0) {
        HIPCHK(hipMemcpyAsync(h->h_compact.p, h->d_compact.p, (size_t)total * sizeof(Cand16), hipMemcpyDeviceToHost,
                              h->copy_stream));
        HIPCHK(hipStreamSynchronize(h->copy_stream));
    }
    h->compact_on_host = true;
    return MSORB_OK;
}

int ensure_group(msorb_extractor* h, int gi) {
    StreamGroup& G = h->grp[gi];
    if (G.ready) return MSORB_OK;
    // The GPU exposes 4 hardware queues per process and HIP deals streams onto them; streams that share a queue
    // serialise.  Keep the number of live streams minimal: group 0 runs on the handle's stream, every group's blur on
    // the one auxiliary stream, only groups >= 1 get a stream of their own (2 groups -> 3 streams + the null stream).
    if (gi == 0) G.s = h->stream;
    else { HIPCHK(hipStreamCreateWithFlags(&G.s, hipStreamNonBlocking)); G.own_stream = true; }
    HIPCHK(hipEventCreateWithFlags(&G.ev_pyr, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&G.ev_blur, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&G.ev_fast, hipEventDisableTiming));
    for (auto& e : G.pe) HIPCHK(hipEventCreate(&e));
    G.ready = true;
    return MSORB_OK;
}

// End of a run_pipeline_groups() call: waits for the sub-batch streams, hands the counts out, folds the stage events.
int finish_groups(msorb_extractor* h, int n_images, int* h_counts, int* h_mono) {
    const int ng = h->last_groups;
    for (int gi = 0; gi < ng; gi++) HIPCHK(hipStreamSynchronize(h->grp[gi].s));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n_images; i++) {
        if (h->h_sel_count.p[i] < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
        h_counts[i] = h->h_sel_count.p[i];
        if (h_mono) h_mono[i] = h->h_mono.p[i];
    }
    if (h->last_prof) {  // stage time = sum over the sub-batches of the stage's HIP-event interval on its own stream
        // (with the blur on the main stream its interval sits between pyramid and FAST: FAST = 8 -> 2)
        const int fast_from = h->overlap_blur ? 1 : 8;
        const int map[6][3] = {{MSORB_STAGE_PYRAMID, 0, 1}, {MSORB_STAGE_FAST, fast_from, 2}, {MSORB_STAGE_COMPACT, 2, 3},
                               {MSORB_STAGE_BLUR, 7, 8}, {MSORB_STAGE_SELECT, 3, 5}, {MSORB_STAGE_DESCRIBE, 5, 6}};
        for (auto& m : map) h->stage_ms[m[0]] = 0;
        for (int gi = 0; gi < ng; gi++)
            for (auto& m : map) {
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, h->grp[gi].pe[m[1]], h->grp[gi].pe[m[2]]));
                h->stage_ms[m[0]] += ms;
            }
    }
    return MSORB_OK;
}

// Device-only pipeline (device quadtree) over several sub-batches, each on its own pair of streams, so that the
// latency-bound quadtree of one sub-batch overlaps the FAST / pyramid kernels of the next; inside a sub-batch
// the blur runs on the second stream.  No host work between the stages; one read-back of the counts at the end.
int run_pipeline_groups(msorb_extractor* h, const LevelView& level0, int n_images, int lap0, int lap1,
                        msorb_keypoint* d_kps, uint8_t* d_desc, int capacity, int* h_counts, int* h_mono) {
    const FrameGeom& g = h->G;
    const int nl = g.nlevels;
    const int ncells = (int)g.cells.size();
    const bool prof = h->profiling;
    const int sel_stride = h->sel_stride;
    int ng = n_images >= 16 ? std::min(h->n_groups, kMaxGroups) : 1;
    ng = std::max(1, std::min(ng, n_images));
    const PyramidView pyr_all = make_view(h, h->d_pyr.p, &level0);
    const PyramidView blur_all = make_blur_view(h, h->d_blur.p);
    h->last_pyr = pyr_all; h->last_blur = blur_all; h->last_n_images = n_images;
    h->h_pyr_valid = false;
    h->h_pyr_async = false;
    h->compact_on_host = false;
    h->last_groups = ng;
    h->last_prof = prof;
    // sub-batches on streams of their own must not start before the handle's stream has drained (H2D of level 0 in
    // msorb_extract); a single group runs on that very stream, where the order is implicit — and the launches below are
    // then issued while the copy is still in flight instead of after it
    if (!h->capturing && ng > 1) HIPCHK(hipStreamSynchronize(h->stream));
    int first = 0;
    for (int gi = 0; gi < ng; gi++) {
        int rc;
        if ((rc = ensure_group(h, gi))) return rc;
        StreamGroup& G = h->grp[gi];
        const int n = (n_images - first) / (ng - gi);
        hipStream_t s = G.s;
        auto mark = [&](int i, hipStream_t st) { if (prof) (void)hipEventRecord(G.pe[i], st); };
        LevelView l0 = level0;
        l0.base = level0.base + (size_t)first * level0.img_stride;
        uint8_t* pyr_base = h->d_pyr.p + (size_t)first * g.pyramid_bytes;
        const PyramidView pyr = make_view(h, pyr_base, &l0);
        const PyramidView blur = make_blur_view(h, h->d_blur.p + (size_t)first * g.blur_bytes);
        int* img_base = h->d_img_base.p + first + gi;
        const size_t cslot = (size_t)first * g.slots_per_image;
        mark(0, s);
        if (!(n <= 4 && !h->sem.resize_single_stage && launch_pyramid_tower(pyr, h->tower, h->d_taps.p, h->tap_x_off.data(), h->tap_y_off.data(), n, s)))
            launch_pyramid(pyr, h->d_taps.p, h->tap_x_off.data(), h->tap_y_off.data(), n, s, h->sem);
        mark(1, s);
        // A frame (<= 4 images, one group): FAST and the blur leave as ONE launch on the main stream (frame_fast_blur_kernel) — no side
        // stream, no fork / join.  Not with stage timing on (the stage events want the two kernels apart), not for variants of the
        // table the streaming blur does not serve (launch_frame_fast_blur then returns false before launching anything).
        // frame_fuse (MSORB_FRAME_FUSE, read once per handle): 2 (default) the blur rides the SELECTION launch (quadtree_select_blur_kernel:
        // off the critical path — the selection leaves 240 CUs idle), 1 it rides FAST's launch (frame_fast_blur_kernel), 0 side stream.
        const bool fuse_ok = ng == 1 && n <= 4 && !prof && h->overlap_blur && h->sem.default_taps();
        FrameBlurJob blur_job;
        const bool fuse_qt = fuse_ok && h->knobs.frame_fuse == 2 && n_images <= 4 && make_frame_blur_job(pyr, blur, n, h->sem, &blur_job);
        const bool fuse_fb = fuse_ok && !fuse_qt && h->knobs.frame_fuse >= 1;
        hipStream_t sb = h->overlap_blur && !fuse_fb && !fuse_qt ? h->copy_stream : s;
        const bool side_blur = h->overlap_blur && !fuse_fb && !fuse_qt;
        if (side_blur || h->host_pyramid) HIPCHK(hipEventRecord(G.ev_pyr, s));   // the pyramid is complete: the side stream's blur and the host-pyramid copies wait for this
        if (side_blur) HIPCHK(hipStreamWaitEvent(sb, G.ev_pyr, 0));
        if (h->host_pyramid && h->pair_pyramids == 2 && n_images == 2 && ng == 1 && !h->capturing) {
            // msorb_extract_pair with the host pyramids requested: levels 1.. of both images leave on the pyramid stream
            int prc;
            if ((prc = h->h_pyr.ensure(2 * g.pyramid_bytes))) return prc;
            if (!h->pyr_stream) {
                HIPCHK(hipStreamCreateWithFlags(&h->pyr_stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->ev_pyr_done, hipEventDisableTiming));
            }
            HIPCHK(hipStreamWaitEvent(h->pyr_stream, G.ev_pyr, 0));
            if (nl > 1)
                for (int i = 0; i < 2; i++)
                    HIPCHK(hipMemcpyAsync(h->h_pyr.p + (size_t)i * g.pyramid_bytes + g.lv[1].plane_off,
                                          h->d_pyr.p + (size_t)i * g.pyramid_bytes + g.lv[1].plane_off, g.pyramid_bytes - g.lv[1].plane_off,
                                          hipMemcpyDeviceToHost, h->pyr_stream));
            h->h_pyr_async = true;
        }
        if (h->host_pyramid && n_images == 1 && !h->capturing && level0.base == h->d_pyr.p + g.lv[0].plane_off) {
            // per-frame call with the host pyramid requested: levels 1.. leave for pinned memory now, on their own stream
            int prc;
            if ((prc = h->h_pyr.ensure(g.pyramid_bytes))) return prc;
            if (!h->pyr_stream) {
                HIPCHK(hipStreamCreateWithFlags(&h->pyr_stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->ev_pyr_done, hipEventDisableTiming));
            }
            HIPCHK(hipStreamWaitEvent(h->pyr_stream, G.ev_pyr, 0));
            if (nl > 1)
                HIPCHK(hipMemcpyAsync(h->h_pyr.p + g.lv[1].plane_off, h->d_pyr.p + g.lv[1].plane_off,
                                      g.pyramid_bytes - g.lv[1].plane_off, hipMemcpyDeviceToHost, h->pyr_stream));
            h->h_pyr_async = true;
        }
        // with the blur on its own stream the critical chain goes first: FAST -> compaction -> quadtree is what describe waits
        // for; the blur (needed by describe only) fills in beside it
        bool blur_carried = false;
        auto blur_now = [&]() {
            mark(7, sb);
            (void)launch_gauss7(pyr, blur, n, sb, h->sem);
            mark(8, sb);
            if (side_blur) (void)hipEventRecord(G.ev_blur, sb);
        };
        if (!(fuse_fb && launch_frame_fast_blur(pyr, blur, h->d_cells.p, ncells, h->P.ini_th, h->P.min_th, g.slots_per_image, h->d_slots.p + cslot,
                                                h->d_cell_count.p + (size_t)first * ncells, n, h->small_cells, s, h->sem))) {
            if (!h->overlap_blur) blur_now();   // one stream: pyramid, blur, FAST (the stage events expect this order)
            launch_fast_cells(pyr, h->d_cells.p, ncells, h->P.ini_th, h->P.min_th, g.slots_per_image, h->d_slots.p + cslot,
                              h->d_cell_count.p + (size_t)first * ncells, n, h->small_cells, s);
            mark(2, s);
            if (h->overlap_blur && !fuse_qt) blur_now();
        }
        launch_cand_compact(h->d_cells.p, ncells, h->d_level_cell_begin.p, nl, g.slots_per_image, h->d_slots.p + cslot,
                            h->d_cell_count.p + (size_t)first * ncells, h->d_cell_off.p + (size_t)first * ncells,
                            h->d_level_count.p + (size_t)first * nl, h->d_img_total.p + first, img_base,
                            h->d_compact.p + cslot, n, s, /*packed=*/false, /*frame_form=*/h->knobs.frame_compact && ng == 1);
        h->compact_fixed_stride = true;
        mark(3, s);
        if ((rc = launch_quadtree(h->qt, h->d_compact.p + cslot, img_base, h->d_level_count.p + (size_t)first * nl,
                                  h->d_label.p + cslot, h->d_sel_pt.p + (size_t)first * sel_stride, h->d_sel_n.p + (size_t)first * nl,
                                  sel_stride, h->scales, lap0, lap1, capacity, h->d_sel.p + (size_t)first * sel_stride,
                                  h->d_sel_count.p + first, h->d_mono.p + first, n, s, ng == 1 ? h->row_job : nullptr,
                                  fuse_qt ? &blur_job : nullptr, &blur_carried,
                                  h->qt_global ? h->d_qt_ws.p + (size_t)first * nl * quadtree_global_workspace_stride(h->qt) : nullptr,   // (groups run side by side: a slice each)
                                  h->qt_global ? h->d_label_wide.p + cslot : nullptr)))
            return rc;
        if (fuse_qt && !blur_carried) blur_now();   // (the selection ran a form that does not carry the blur: on this stream, before describe)
        mark(5, s);
        if (side_blur) HIPCHK(hipStreamWaitEvent(s, G.ev_blur, 0));
        launch_describe(pyr, blur, h->d_sel.p + (size_t)first * sel_stride, h->d_sel_count.p + first, sel_stride, h->scales,
                        d_kps + (size_t)first * capacity, d_desc + (size_t)first * capacity * 32, capacity,
                        std::min(capacity, sel_stride), n, s, h->sem);
        mark(6, s);
        if (!h->skip_count_copies) {  // a fused caller takes the counts from the device itself
            if (n_images <= 4 && first == 0) {   // per-frame call: by copy kernel (the buffers hold >= 4 ints)
                int crc;
                if ((crc = frame_copy(h, h->h_sel_count.p, h->d_sel_count.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s)) ||
                    (crc = frame_copy(h, h->h_mono.p, h->d_mono.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s)))
                    return crc;
            } else {
                HIPCHK(hipMemcpyAsync(h->h_sel_count.p + first, h->d_sel_count.p + first, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(hipMemcpyAsync(h->h_mono.p + first, h->d_mono.p + first, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
            }
        }
        first += n;
    }
    if (h->capturing || h->defer_sync) return MSORB_OK;  // graph capture / fused / submitted call: the caller synchronises and reads back
    return finish_groups(h, n_images, h_counts, h_mono);
}

// The pipeline proper.  level0: where level 0 of every image lives (device memory).
int run_pipeline(msorb_extractor* h, const LevelView& level0, int n_images, int lap0, int lap1,
                 msorb_keypoint* d_kps, uint8_t* d_desc, int capacity, int* h_counts, int* h_mono) {
    h->pair_pyramids = h->pair_request;   // (any other call forgets the pair state of an earlier msorb_extract_pair)
    h->pair_request = 0;
    if (h->device_quadtree && !h->knobs.serial_pipeline)
        return run_pipeline_groups(h, level0, n_images, lap0, lap1, d_kps, d_desc, capacity, h_counts, h_mono);
    h->last_groups = 1;
    const FrameGeom& g = h->G;
    const int nl = g.nlevels;
    const int ncells = (int)g.cells.size();
    hipStream_t s = h->stream;
    const bool prof = h->profiling;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->pe[i], s); };

    const PyramidView pyr = make_view(h, h->d_pyr.p, &level0);
    const PyramidView blur = make_blur_view(h, h->d_blur.p);
    h->last_pyr = pyr; h->last_blur = blur; h->last_n_images = n_images;
    h->h_pyr_valid = false;

    mark(0);
    if (!(n_images <= 4 && !h->sem.resize_single_stage && launch_pyramid_tower(pyr, h->tower, h->d_taps.p, h->tap_x_off.data(), h->tap_y_off.data(), n_images, s)))
        launch_pyramid(pyr, h->d_taps.p, h->tap_x_off.data(), h->tap_y_off.data(), n_images, s, h->sem);
    mark(1);
    // the blur only feeds the descriptor stage: unless stage timing is on, it runs on the second stream, overlapping
    // the (VALU-bound) FAST kernel and the (latency-bound) quadtree with a bandwidth-bound kernel
    const bool overlap_blur = h->overlap_blur;
    if (overlap_blur) {
        HIPCHK(hipEventRecord(h->ev_pyramid, s));
        HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_pyramid, 0));
        if (prof) (void)hipEventRecord(h->pe[7], h->copy_stream);
        (void)launch_gauss7(pyr, blur, n_images, h->copy_stream, h->sem);
        if (prof) (void)hipEventRecord(h->pe[8], h->copy_stream);
        HIPCHK(hipEventRecord(h->ev_blur, h->copy_stream));
    }
    launch_fast_cells(pyr, h->d_cells.p, ncells, h->P.ini_th, h->P.min_th, g.slots_per_image, h->d_slots.p,
                      h->d_cell_count.p, n_images, h->small_cells, s);
    mark(2);
    launch_cand_compact(h->d_cells.p, ncells, h->d_level_cell_begin.p, nl, g.slots_per_image, h->d_slots.p,
                        h->d_cell_count.p, h->d_cell_off.p, h->d_level_count.p, h->d_img_total.p, h->d_img_base.p,
                        h->d_compact.p, n_images, s, /*packed=*/true);
    h->compact_fixed_stride = false;
    mark(3);
    HIPCHK(hipEventRecord(h->ev_compact, s));
    if (!overlap_blur) (void)launch_gauss7(pyr, blur, n_images, s, h->sem);
    mark(4);
    h->compact_on_host = false;
    const int sel_stride = h->sel_stride;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t1 = t0;

    if (h->device_quadtree) {
        // selection stays on the device: quadtree per (level, image), output layout per image
        const int qrc = launch_quadtree(h->qt, h->d_compact.p, h->d_img_base.p, h->d_level_count.p, h->d_label.p, h->d_sel_pt.p,
                                        h->d_sel_n.p, sel_stride, h->scales, lap0, lap1, capacity, h->d_sel.p, h->d_sel_count.p,
                                        h->d_mono.p, n_images, s, h->row_job, nullptr, nullptr, h->qt_global ? h->d_qt_ws.p : nullptr,
                                        h->qt_global ? h->d_label_wide.p : nullptr);
        if (qrc) return qrc;
        mark(5);
        if (overlap_blur) HIPCHK(hipStreamWaitEvent(s, h->ev_blur, 0));
        launch_describe(pyr, blur, h->d_sel.p, h->d_sel_count.p, sel_stride, h->scales, d_kps, d_desc, capacity,
                        std::min(capacity, sel_stride), n_images, s, h->sem);
        mark(6);
        HIPCHK(hipMemcpyAsync(h->h_sel_count.p, h->d_sel_count.p, (size_t)n_images * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_mono.p, h->d_mono.p, (size_t)n_images * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        for (int i = 0; i < n_images; i++) {
            if (h->h_sel_count.p[i] < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
            h_counts[i] = h->h_sel_count.p[i];
            if (h_mono) h_mono[i] = h->h_mono.p[i];
        }
    } else {
        int rc;
        if ((rc = fetch_candidates(h, n_images))) return rc;
        // host selection, one task per image
        std::atomic<int> overflow{0};
        std::function<void(int)> task = [&](int img) {
            const Cand16* c = h->h_compact.p + h->h_img_base.p[img];
            const int* lc = h->h_level_count.p + (size_t)img * nl;
            SelRec* out = h->h_sel.p + (size_t)img * sel_stride;
            std::vector<int> kept;
            int n = 0;
            // pass 1: quadtree per level, records in level-major / quadtree order
            for (int l = 0; l < nl; l++) {
                const LevelGeom& lg = g.lv[l];
                distribute_quadtree(c, lc[l], lg.min_x, lg.max_x, lg.min_y, lg.max_y, lg.quota, kept);
                for (int k : kept) {
                    if (n >= sel_stride || n >= capacity) { overflow.store(1); break; }
                    SelRec r;
                    r.x = (uint16_t)(c[k].x + kMinBorder);
                    r.y = (uint16_t)(c[k].y + kMinBorder);
                    r.score = c[k].score;
                    r.level = (uint8_t)l;
                    r.pad = 0;
                    r.dst = 0;
                    out[n++] = r;
                }
                c += lc[l];
            }
            // pass 2: output rows (ORBextractor.cc:1122-1163): inside [lap0,lap1] from the back, else from the front
            int mono = 0, stereo = n - 1;
            for (int i = 0; i < n; i++) {
                SelRec& r = out[i];
                const float fx = r.level ? (float)r.x * h->P.scale[r.level] : (float)r.x;
                if (fx >= (float)lap0 && fx <= (float)lap1) r.dst = stereo--;
                else r.dst = mono++;
            }
            h->h_sel_count.p[img] = n;
            h_counts[img] = n;
            if (h_mono) h_mono[img] = mono;
        };
        if (!h->pool) {   // the worker threads of the host-quadtree path (MSORB_QUADTREE=host) exist only once that path has run
            int nthreads = h->knobs.host_threads > 0 ? h->knobs.host_threads : (int)std::thread::hardware_concurrency();
            h->pool.reset(new Pool(std::max(1, std::min(nthreads, 64))));
        }
        h->pool->parallel_for(n_images, task);
        if (overflow.load()) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
        int max_sel = 0;
        for (int i = 0; i < n_images; i++) max_sel = std::max(max_sel, h->h_sel_count.p[i]);
        HIPCHK(hipMemcpyAsync(h->d_sel.p, h->h_sel.p, (size_t)n_images * sel_stride * sizeof(SelRec), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->d_sel_count.p, h->h_sel_count.p, (size_t)n_images * sizeof(int), hipMemcpyHostToDevice, s));
        t1 = std::chrono::steady_clock::now();
        mark(5);
        if (overlap_blur) HIPCHK(hipStreamWaitEvent(s, h->ev_blur, 0));
        launch_describe(pyr, blur, h->d_sel.p, h->d_sel_count.p, sel_stride, h->scales, d_kps, d_desc, capacity, max_sel,
                        n_images, s, h->sem);
        mark(6);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
    }
    if (prof) {
        float ms = 0;
        const int map[5][3] = {{MSORB_STAGE_PYRAMID, 0, 1}, {MSORB_STAGE_FAST, 1, 2}, {MSORB_STAGE_COMPACT, 2, 3},
                               {MSORB_STAGE_BLUR, overlap_blur ? 7 : 3, overlap_blur ? 8 : 4}, {MSORB_STAGE_DESCRIBE, 5, 6}};
        for (auto& m : map) {
            HIPCHK(hipEventElapsedTime(&ms, h->pe[m[1]], h->pe[m[2]]));
            h->stage_ms[m[0]] = ms;
        }
        if (h->device_quadtree) {
            HIPCHK(hipEventElapsedTime(&ms, h->pe[4], h->pe[5]));
            h->stage_ms[MSORB_STAGE_SELECT] = ms;
        } else {
            h->stage_ms[MSORB_STAGE_SELECT] = std::chrono::duration<float, std::milli>(t1 - t0).count();
        }
    }
    return MSORB_OK;
}

}  // namespace

namespace msorb {
void set_last_error(const std::string& s) { g_last_error = s; }
// Device views of the last call's pyramid (used by the stereo matcher, which reads both eyes' levels).
int extractor_last_view(msorb_extractor* h, PyramidView* pyr, LevelScale* sc, float* inv_scale, int* device,
                        hipStream_t* stream, int* n_images) {
    if (!h || !h->geom_valid || h->last_n_images < 1) {
        set_error("extractor has no pyramid yet (call msorb_extract first)");
        return MSORB_E_INVALID;
    }
    *pyr = h->last_pyr;
    if (sc) *sc = h->scales;
    if (inv_scale)
        for (int l = 0; l < h->P.nlevels; l++) inv_scale[l] = h->P.inv_scale[l];
    *device = h->device;
    *stream = h->stream;
    if (n_images) *n_images = h->last_n_images;
    return MSORB_OK;
}
int extractor_device(const msorb_extractor* h) { return h->device; }
bool extractor_force_peer_pyramid(const msorb_extractor* h) { return h->knobs.force_peer_pyramid; }
int extractor_levels(const msorb_extractor* h) { return h->P.nlevels; }
}  // namespace msorb

extern "C" {

const char* msorb_last_error(void) { return g_last_error.c_str(); }

int msorb_abi_version(void) { return MSORB_ABI_VERSION; }
int msorb_abi_compatible(int header_version) {
    return header_version / 1000 == MSORB_ABI_VERSION / 1000 && header_version % 1000 <= MSORB_ABI_VERSION % 1000;
}

namespace {
std::mutex g_fatal_mutex;
msorb_fatal_fn g_fatal_fn = nullptr;
void* g_fatal_user = nullptr;
}  // namespace
void msorb_set_fatal_callback(msorb_fatal_fn fn, void* user) {
    std::lock_guard<std::mutex> lk(g_fatal_mutex);
    g_fatal_fn = fn;
    g_fatal_user = user;
}
void msorb_notify_fatal(int code, const char* what) {
    msorb_fatal_fn fn;
    void* user;
    {
        std::lock_guard<std::mutex> lk(g_fatal_mutex);   // not held across the call: the callback may not return
        fn = g_fatal_fn;
        user = g_fatal_user;
    }
    if (fn) fn(code, what ? what : "", user);
}

int msorb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int msorb_device_memory(int device, size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return MSORB_OK;
}

int msorb_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int device,
                           msorb_extractor** out) {
    if (!out) return MSORB_E_INVALID;
    *out = nullptr;
    if (nfeatures <= 0 || nlevels < 1 || nlevels > MSORB_MAX_LEVELS || !(scale_factor > 1.0f) || ini_th < 0 ||
        min_th < 0 || ini_th > 255 || min_th > ini_th) {
        set_error("invalid extractor parameters");
        return MSORB_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    msorb_extractor* h = new msorb_extractor();
    h->device = device;
    h->P.init(nfeatures, scale_factor, nlevels, ini_th, min_th);
    for (int l = 0; l < nlevels; l++) {
        h->scales.scale[l] = h->P.scale[l];
        h->scales.patch[l] = (float)(int)(kPatchSize * h->P.scale[l]);
    }
    // The second stream carries the blur (VALU-bound, nobody waits for it before describe): at the lowest stream priority it takes
    // the workgroup slots the main chains leave free instead of competing with their FAST kernels — 1.345 -> 1.325 ms per step
    // with two batches in flight.  (Measured and dropped: the short dependent kernels — pyramid levels, scans, quadtree — on a
    // third, high-priority stream per handle: 1.54 ms, six streams on four hardware queues serialise; one handle above the other.)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio_blur = prio_least;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, prio_blur) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_compact, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_pyramid, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming) != hipSuccess) {
        set_error("stream/event creation failed");
        delete h;
        return MSORB_E_HIP;
    }
    for (auto& e : h->pe)
        if (hipEventCreate(&e) != hipSuccess) { delete h; return MSORB_E_HIP; }
    upload_patch_tables(kPattern, h->P.umax, h->stream);
    // the environment is read here and nowhere else in this file
    {
        const char* e;
        h->knobs.serial_pipeline = getenv("MSORB_SERIAL_PIPELINE") != nullptr;
        h->knobs.quadtree_host = (e = getenv("MSORB_QUADTREE")) && std::string(e) == "host";
        h->knobs.quadtree_global = (e = getenv("MSORB_QUADTREE")) && std::string(e) == "global";   // (test switch: the global-workspace selection at any quota)
        h->knobs.split_no_peer = getenv("MSORB_SPLIT_NO_PEER") != nullptr;
        h->knobs.force_peer_pyramid = getenv("MSORB_FORCE_PEER_PYRAMID") != nullptr;
        h->knobs.host_threads = (e = getenv("MSORB_HOST_THREADS")) ? atoi(e) : 0;
        // A/B switches of the frame chain's fused launches: MSORB_FRAME_FUSE = where a frame's blur runs (2 default: inside the selection
        // launch, 1: inside FAST's launch, 0: side stream); MSORB_FRAME_COMPACT=0: candidate scan + gather as two launches
        h->knobs.frame_fuse = (e = getenv("MSORB_FRAME_FUSE")) ? std::max(0, std::min(2, atoi(e))) : 2;
        h->knobs.frame_compact = !((e = getenv("MSORB_FRAME_COMPACT")) && e[0] == '0');
    }
    if (hipDeviceGetAttribute(&h->lds_per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || h->lds_per_block <= 0)
        h->lds_per_block = 64 * 1024;
    *out = h;
    return MSORB_OK;
}

void msorb_extractor_destroy(msorb_extractor* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->pool.reset();
    h->d_pyr.release(); h->d_blur.release(); h->d_desc1.release(); h->d_taps.release(); h->d_cells.release();
    h->d_level_cell_begin.release(); h->d_cell_count.release(); h->d_cell_off.release(); h->d_level_count.release();
    h->d_img_total.release(); h->d_img_base.release(); h->d_sel_count.release(); h->d_slots.release();
    h->d_compact.release(); h->d_sel.release(); h->d_kps1.release();
    h->d_st_sad.release(); h->d_st_rows.release(); h->d_st_list.release(); h->d_st_block.release(); h->d_st_img.release(); h->d_out1.release();
    h->d_gather_pyr.release(); h->d_gather_cnt.release();
    if (h->ev_split) (void)hipEventDestroy(h->ev_split);
    if (h->ev_pyr_done) (void)hipEventDestroy(h->ev_pyr_done);
    if (h->pyr_stream) { (void)hipStreamSynchronize(h->pyr_stream); (void)hipStreamDestroy(h->pyr_stream); }
    h->h_level_count.release(); h->h_img_base.release(); h->h_sel_count.release(); h->h_compact.release();
    h->h_sel.release(); h->h_pyr.release(); h->h_img_pin.release(); h->h_out_pin.release(); h->h_gather.release();
    for (auto& G : h->grp) {
        if (!G.ready) continue;
        (void)hipStreamSynchronize(G.s);
        for (auto& e : G.pe) if (e) (void)hipEventDestroy(e);
        (void)hipEventDestroy(G.ev_pyr); (void)hipEventDestroy(G.ev_blur); (void)hipEventDestroy(G.ev_fast);
        if (G.own_stream) (void)hipStreamDestroy(G.s);
    }
    for (auto& e : h->pe) if (e) (void)hipEventDestroy(e);
    if (h->ev_compact) (void)hipEventDestroy(h->ev_compact);
    if (h->ev_pyramid) (void)hipEventDestroy(h->ev_pyramid);
    if (h->ev_blur) (void)hipEventDestroy(h->ev_blur);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    delete h;
}

int msorb_extractor_tables(const msorb_extractor* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                           int* per_level) {
    if (!h) return MSORB_E_INVALID;
    for (int l = 0; l < h->P.nlevels; l++) {
        if (scale) scale[l] = h->P.scale[l];
        if (inv_scale) inv_scale[l] = h->P.inv_scale[l];
        if (sigma2) sigma2[l] = h->P.sigma2[l];
        if (inv_sigma2) inv_sigma2[l] = h->P.inv_sigma2[l];
        if (per_level) per_level[l] = h->P.per_level[l];
    }
    return MSORB_OK;
}

int msorb_extractor_capacity(const msorb_extractor* h) { return h ? capacity_of(h) : MSORB_E_INVALID; }

int msorb_extractor_set_profiling(msorb_extractor* h, int enable) {
    if (!h) return MSORB_E_INVALID;
    h->profiling = enable != 0;
    return MSORB_OK;
}
int msorb_extractor_set_overlap(msorb_extractor* h, int sub_batches, int blur_on_second_stream) {
    if (!h || sub_batches < 1 || sub_batches > kMaxGroups) return MSORB_E_INVALID;
    h->n_groups = sub_batches;
    h->overlap_blur = blur_on_second_stream != 0;
    return MSORB_OK;
}
int msorb_extractor_set_semantics(msorb_extractor* h, const msorb_semantics* sem) {
    if (!h) return MSORB_E_INVALID;
    if (h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    Semantics s;
    if (sem) {
        int sum = 0;
        for (int i = 0; i < 7; i++) {
            if (sem->gauss_taps[i] < 0 || sem->gauss_taps[i] > 255) { set_error("gauss tap outside 0..255"); return MSORB_E_INVALID; }
            s.gauss_taps[i] = sem->gauss_taps[i];
            sum += sem->gauss_taps[i];
        }
        if (sum < 1 || sum > 257) { set_error("gauss taps: the 16-bit horizontal sums need sum(taps) <= 257"); return MSORB_E_INVALID; }
        s.resize_single_stage = sem->resize_rounding != 0;
        s.atan2_fma = sem->atan2_fma != 0;
        if (sem->brief_tap < 0 || sem->brief_tap > 2) { set_error("brief_tap must be 0 (first product fused), 1 (second product fused) or 2 (no contraction)"); return MSORB_E_INVALID; }
        s.brief_tap = sem->brief_tap;
    }
    h->sem = s;
    return MSORB_OK;
}

int msorb_extractor_set_host_pyramid(msorb_extractor* h, int enable) {
    if (!h) return MSORB_E_INVALID;
    h->host_pyramid = enable != 0;
    return MSORB_OK;
}
int msorb_extractor_stage_ms(const msorb_extractor* h, float* ms) {
    if (!h || !ms) return MSORB_E_INVALID;
    for (int i = 0; i < MSORB_N_STAGES; i++) ms[i] = h->stage_ms[i];
    return MSORB_OK;
}

static int extract_batch_common(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols, size_t row_stride,
                                size_t image_stride, int lap0, int lap1, msorb_keypoint* d_kps, uint8_t* d_desc, int capacity,
                                int* h_counts, int* h_mono, bool submit_only) {
    if (!h || !d_kps || !d_desc || (!submit_only && !h_counts) || n_images < 0) { set_error("null argument"); return MSORB_E_INVALID; }
    if (h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    if (!d_images || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;
    if (n_images == 0) return MSORB_OK;
    if ((int)row_stride < cols || (n_images > 1 && image_stride < row_stride * (size_t)rows)) {
        set_error("bad strides");
        return MSORB_E_INVALID;
    }
    if (submit_only && (!h->device_quadtree || h->knobs.serial_pipeline)) {
        set_error("msorb_extract_batch_submit needs the device pipeline");
        return MSORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    if ((rc = ensure_batch(h, n_images))) return rc;
    LevelView l0{d_images, image_stride, (int)row_stride, cols, rows};
    // rows that are not 4-byte aligned (or leave no slack for the row-coherent dword reads) would push every kernel onto
    // its byte-granular variant: copy level 0 once into the handle's aligned planes instead
    const LevelGeom& g0 = h->G.lv[0];
    const bool misaligned = (reinterpret_cast<uintptr_t>(d_images) & 3) || (row_stride & 3) || (image_stride & 3) ||
                            row_stride < (size_t)(((cols + 3) & ~3) + 8);
    // measured on MI355X (KITTI rows of 1241 bytes): 256 images 1.88 vs 2.08 ms staged / in place, 64 images 0.65 vs 0.64 —
    // small batches keep the rows in place
    constexpr int stage_min = 128;
    if (misaligned && n_images >= stage_min) {
        launch_stage_level0(l0, h->d_pyr.p + g0.plane_off, g0.pitch, h->G.pyramid_bytes, n_images, h->stream);
        l0 = LevelView{h->d_pyr.p + g0.plane_off, h->G.pyramid_bytes, g0.pitch, cols, rows};
    }
    if (!submit_only) return run_pipeline(h, l0, n_images, lap0, lap1, d_kps, d_desc, capacity, h_counts, h_mono);
    h->defer_sync = true;
    rc = run_pipeline(h, l0, n_images, lap0, lap1, d_kps, d_desc, capacity, nullptr, nullptr);
    h->defer_sync = false;
    if (rc == MSORB_OK) h->pending_batch = n_images;
    return rc;
}

int msorb_extract_batch(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols,
                        size_t row_stride, size_t image_stride, int lap0, int lap1, msorb_keypoint* d_kps,
                        uint8_t* d_desc, int capacity, int* h_counts, int* h_mono) {
    return extract_batch_common(h, d_images, n_images, rows, cols, row_stride, image_stride, lap0, lap1, d_kps, d_desc, capacity,
                                h_counts, h_mono, false);
}

// msorb_extract_batch in two halves, so that a caller can keep several batches in flight (one handle each): submit enqueues the
// whole chain and returns; wait blocks until it has finished and hands out the counts.
int msorb_extract_batch_submit(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols, size_t row_stride,
                               size_t image_stride, int lap0, int lap1, msorb_keypoint* d_kps, uint8_t* d_desc, int capacity) {
    return extract_batch_common(h, d_images, n_images, rows, cols, row_stride, image_stride, lap0, lap1, d_kps, d_desc, capacity,
                                nullptr, nullptr, true);
}
int msorb_extract_batch_wait(msorb_extractor* h, int* h_counts, int* h_mono) {
    if (!h || !h_counts) { set_error("null argument"); return MSORB_E_INVALID; }
    if (!h->pending_batch) { set_error("no submitted batch to wait for"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->pending_batch;
    h->pending_batch = 0;
    return finish_groups(h, n, h_counts, h_mono);
}

int msorb_extract(msorb_extractor* h, const uint8_t* image, int rows, int cols, size_t stride, int lap0, int lap1,
                  msorb_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_keypoints, int* mono_index) {
    if (h && h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    if (!h || !n_keypoints || !mono_index) return MSORB_E_INVALID;
    *n_keypoints = 0;
    *mono_index = -1;
    if (!image || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;  // ORBextractor.cc:1090-1091
    if (!keypoints || !descriptors || (int)stride < cols) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    if ((rc = ensure_batch(h, 1))) return rc;
    const int cap = capacity_of(h);
    if ((rc = h->d_kps1.ensure(cap))) return rc;
    if ((rc = h->d_desc1.ensure((size_t)cap * 32))) return rc;
    const LevelGeom& g0 = h->G.lv[0];
    // level 0 = copy of the caller's image (ORBextractor.cc:1190: the input is never modified or aliased).  Pageable
    // memory is staged through pinned buffers owned by the handle: a plain hipMemcpy from pageable memory makes the
    // driver pin/unpin per call, which costs more than the whole kernel chain.
    const bool staged = h->h_img_pin.p && image == h->h_img_pin.p && stride == (size_t)g0.pitch;   // msorb_stage_image did the copy
    if ((rc = h->h_img_pin.ensure(staged ? 1 : (size_t)g0.pitch * rows))) return rc;
    if ((rc = h->h_out_pin.ensure((size_t)cap * (sizeof(msorb_keypoint) + 32)))) return rc;
    if (!staged)
        for (int y = 0; y < rows; y++) memcpy(h->h_img_pin.p + (size_t)y * g0.pitch, image + (size_t)y * stride, cols);
    LevelView l0{h->d_pyr.p + g0.plane_off, h->G.pyramid_bytes, g0.pitch, cols, rows};
    msorb_keypoint* pk = reinterpret_cast<msorb_keypoint*>(h->h_out_pin.p);
    uint8_t* pd = h->h_out_pin.p + (size_t)cap * sizeof(msorb_keypoint);
    int n = 0, mono = 0;

    if (h->device_quadtree && !h->knobs.serial_pipeline && !h->profiling) {
        // device pipeline: nothing in it needs the host, so the whole frame is enqueued, the full-capacity output block
        // (keypoints + descriptors, ~120 KB) follows in ONE copy and the call synchronises once
        const size_t o_desc = ((size_t)cap * sizeof(msorb_keypoint) + 15) & ~(size_t)15, blk_bytes = o_desc + (size_t)cap * 32;
        if ((rc = h->d_out1.ensure(blk_bytes))) return rc;
        if ((rc = h->h_out_pin.ensure(blk_bytes))) return rc;
        pk = reinterpret_cast<msorb_keypoint*>(h->h_out_pin.p);
        pd = h->h_out_pin.p + o_desc;
        if ((rc = frame_copy(h, h->d_pyr.p + g0.plane_off, h->h_img_pin.p, (size_t)g0.pitch * rows, hipMemcpyHostToDevice, h->stream))) return rc;
        h->defer_sync = true;
        rc = run_pipeline(h, l0, 1, lap0, lap1, reinterpret_cast<msorb_keypoint*>(h->d_out1.p), h->d_out1.p + o_desc, cap, &n, &mono);
        h->defer_sync = false;
        if (rc) return rc;
        if ((rc = frame_copy(h, h->h_out_pin.p, h->d_out1.p, blk_bytes, hipMemcpyDeviceToHost, h->stream))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->h_pyr_async) HIPCHK(hipStreamSynchronize(h->pyr_stream));
        HIPCHK(hipGetLastError());
        n = h->h_sel_count.p[0];
        mono = h->h_mono.p[0];
        if (n < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
        if (n > capacity) { set_error("caller capacity too small"); return MSORB_E_CAPACITY; }
    } else {
        HIPCHK(hipMemcpyAsync(h->d_pyr.p + g0.plane_off, h->h_img_pin.p, (size_t)g0.pitch * rows, hipMemcpyHostToDevice,
                              h->stream));
        if ((rc = run_pipeline(h, l0, 1, lap0, lap1, h->d_kps1.p, h->d_desc1.p, cap, &n, &mono))) return rc;
        if (n > capacity) { set_error("caller capacity too small"); return MSORB_E_CAPACITY; }
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(pk, h->d_kps1.p, (size_t)n * sizeof(msorb_keypoint), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(pd, h->d_desc1.p, (size_t)n * 32, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    if (n > 0) {
        memcpy(keypoints, pk, (size_t)n * sizeof(msorb_keypoint));
        memcpy(descriptors, pd, (size_t)n * 32);
    }
    *n_keypoints = n;
    *mono_index = mono;
    return MSORB_OK;
}

int msorb_extract_pair(msorb_extractor* h, const uint8_t* image_a, const uint8_t* image_b, int rows, int cols, size_t stride_a,
                       size_t stride_b, int lap0, int lap1, msorb_keypoint* kps_a, uint8_t* desc_a, int* n_a, int* mono_a,
                       msorb_keypoint* kps_b, uint8_t* desc_b, int* n_b, int* mono_b, int capacity, int staged) {
    if (h && h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    if (!h || !n_a || !n_b || !mono_a || !mono_b) return MSORB_E_INVALID;
    *n_a = *n_b = 0;
    *mono_a = *mono_b = -1;
    if (!image_a || !image_b || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;
    if (!kps_a || !desc_a || !kps_b || !desc_b || (int)stride_a < cols || (int)stride_b < cols) return MSORB_E_INVALID;
    if (!h->device_quadtree || h->knobs.serial_pipeline || h->profiling) { set_error("msorb_extract_pair needs the device pipeline"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    if ((rc = ensure_batch(h, 2))) return rc;
    const int cap = capacity_of(h);
    const FrameGeom& g = h->G;
    const LevelGeom& g0 = g.lv[0];
    const size_t kp_bytes = (size_t)cap * sizeof(msorb_keypoint), o_desc = 2 * kp_bytes, out_bytes = o_desc + (size_t)2 * cap * 32;
    const size_t plane = (size_t)g0.pitch * rows;
    if ((rc = h->d_st_block.ensure(out_bytes)) || (rc = h->d_st_img.ensure(2 * plane + 256)) || (rc = h->h_img_pin.ensure(2 * plane)) ||
        (rc = h->h_out_pin.ensure(out_bytes)))
        return rc;
    hipStream_t s = h->stream;
    const uint8_t* src[2] = {image_a, image_b};
    const size_t stride[2] = {stride_a, stride_b};
    // A staged image may lie in THIS handle's own staging block (msorb_stage_image stages into plane 0): the plane an un-staged
    // image is copied into must not be one a staged image of this call still has to be uploaded from.
    int own_plane[2] = {-1, -1};   // plane of h_img_pin a staged image occupies (overlaps), -1: memory of another handle
    for (int i = 0; i < 2; i++) {
        if (!(staged & (1 << i))) continue;
        if (stride[i] != (size_t)g0.pitch) { set_error("msorb_extract_pair: a staged image must have the staging pitch"); return MSORB_E_INVALID; }
        const uint8_t* lo = h->h_img_pin.p;
        if (src[i] + plane > lo && src[i] < lo + 2 * plane) {
            if (src[i] != lo && src[i] != lo + plane) { set_error("msorb_extract_pair: a staged pointer inside this handle's staging block must be a plane msorb_stage_image returned"); return MSORB_E_INVALID; }
            own_plane[i] = src[i] == lo ? 0 : 1;
        }
    }
    for (int i = 0; i < 2; i++) {
        const uint8_t* pin;
        if (staged & (1 << i)) {   // already in pinned memory at the library's pitch (msorb_stage_image)
            pin = src[i];
        } else {
            const int other = own_plane[1 - i];
            const int dst_plane = other == i ? 1 - i : i;   // the partner's staged image sits in this image's usual plane: take the other one
            uint8_t* d = h->h_img_pin.p + (size_t)dst_plane * plane;
            for (int y = 0; y < rows; y++) memcpy(d + (size_t)y * g0.pitch, src[i] + (size_t)y * stride[i], cols);
            pin = d;
        }
        h->pair_l0[i] = pin;
        if ((rc = frame_copy(h, h->d_st_img.p + (size_t)i * plane, pin, plane, hipMemcpyHostToDevice, s))) return rc;
    }
    LevelView l0{h->d_st_img.p, plane, g0.pitch, cols, rows};
    uint8_t* const blk = h->d_st_block.p;
    int counts[2] = {0, 0}, mono[2] = {0, 0};
    h->pair_request = 2;
    h->defer_sync = true;
    rc = run_pipeline(h, l0, 2, lap0, lap1, reinterpret_cast<msorb_keypoint*>(blk), blk + o_desc, cap, counts, mono);
    h->defer_sync = false;
    if (rc) { h->pair_pyramids = 0; return rc; }
    uint8_t* o = h->h_out_pin.p;
    if ((rc = frame_copy(h, o, blk, out_bytes, hipMemcpyDeviceToHost, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    if (h->h_pyr_async) HIPCHK(hipStreamSynchronize(h->pyr_stream));
    HIPCHK(hipGetLastError());
    const int na = h->h_sel_count.p[0], nb = h->h_sel_count.p[1];
    if (na < 0 || nb < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
    if (na > capacity || nb > capacity) { set_error("caller capacity too small"); return MSORB_E_CAPACITY; }
    memcpy(kps_a, o, (size_t)na * sizeof(msorb_keypoint));
    memcpy(kps_b, o + kp_bytes, (size_t)nb * sizeof(msorb_keypoint));
    memcpy(desc_a, o + o_desc, (size_t)na * 32);
    memcpy(desc_b, o + o_desc + (size_t)cap * 32, (size_t)nb * 32);
    *n_a = na; *n_b = nb;
    *mono_a = h->h_mono.p[0]; *mono_b = h->h_mono.p[1];
    return MSORB_OK;
}

int msorb_stage_image(msorb_extractor* h, const uint8_t* image, int rows, int cols, size_t stride, const uint8_t** pinned, size_t* pitch) {
    if (!h || !pinned || !image || rows <= 0 || cols <= 0 || (int)stride < cols) return MSORB_E_INVALID;
    if (h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    const LevelGeom& g0 = h->G.lv[0];
    if ((rc = h->h_img_pin.ensure(2 * (size_t)g0.pitch * rows))) return rc;   // (two planes: msorb_extract_pair stages here too)
    for (int y = 0; y < rows; y++) memcpy(h->h_img_pin.p + (size_t)y * g0.pitch, image + (size_t)y * stride, cols);
    *pinned = h->h_img_pin.p;
    if (pitch) *pitch = (size_t)g0.pitch;
    return MSORB_OK;
}

int msorb_pyramid_level_image(msorb_extractor* h, int image, int level, const uint8_t** data, int* rows, int* cols, size_t* stride) {
    if (image == 0 && !(h && h->pair_pyramids == 2)) return msorb_pyramid_level(h, level, data, rows, cols, stride);
    if (!h || !data || !h->geom_valid || h->pair_pyramids != 2 || h->last_n_images != 2 || image < 0 || image > 1 || level < 0 ||
        level >= h->G.nlevels)
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const FrameGeom& g = h->G;
    if (!h->h_pyr_async) {   // the pair call ran without msorb_extractor_set_host_pyramid: fetch both pyramids now, once
        int rc;
        if ((rc = h->h_pyr.ensure(2 * g.pyramid_bytes))) return rc;
        if (g.nlevels > 1)
            for (int i = 0; i < 2; i++)
                HIPCHK(hipMemcpy(h->h_pyr.p + (size_t)i * g.pyramid_bytes + g.lv[1].plane_off,
                                 h->d_pyr.p + (size_t)i * g.pyramid_bytes + g.lv[1].plane_off, g.pyramid_bytes - g.lv[1].plane_off, hipMemcpyDeviceToHost));
        if (!h->pyr_stream) {
            HIPCHK(hipStreamCreateWithFlags(&h->pyr_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_pyr_done, hipEventDisableTiming));
        }
        h->h_pyr_async = true;
    }
    HIPCHK(hipStreamSynchronize(h->pyr_stream));
    *data = level == 0 ? h->pair_l0[image] : h->h_pyr.p + (size_t)image * g.pyramid_bytes + g.lv[level].plane_off;
    if (rows) *rows = g.lv[level].h;
    if (cols) *cols = g.lv[level].w;
    if (stride) *stride = g.lv[level].pitch;
    return MSORB_OK;
}

// Both eyes of one stereo frame in one call: the two images go through the batch pipeline together (one chain of
// launches instead of two racing on two host threads), Frame::ComputeStereoMatches runs on the device outputs
// (band records from the layout launch, match + median kernels of msorb_stereo_matches_batch) and everything comes back with one synchronisation.
int msorb_extract_stereo(msorb_extractor* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t stride_left,
                         size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left, int* n_left,
                         msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right,
                         float* depth, int* n_oob) {
    return msorb::extract_stereo_sink(h, left, right, rows, cols, stride_left, stride_right, mb, mbf, kps_left, desc_left, n_left,
                                      kps_right, desc_right, n_right, capacity, u_right, depth, n_oob, nullptr, nullptr);
}
}  // extern "C"

// msorb_extract_stereo with a sink for the device outputs (track.hip chains the frame grid and the local-points search
// behind the stereo association, on the same stream, before the one read-back)
int msorb::extract_stereo_sink(msorb_extractor* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t stride_left,
                               size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left, int* n_left,
                               msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right,
                               float* depth, int* n_oob, StereoSinkFn sink, void* ctx) {
    if (h && h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    if (!h || !n_left || !n_right) return MSORB_E_INVALID;
    *n_left = *n_right = 0;
    if (n_oob) *n_oob = 0;
    if (!left || !right || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;
    if (!kps_left || !desc_left || !kps_right || !desc_right || !u_right || !depth || (int)stride_left < cols ||
        (int)stride_right < cols)
        return MSORB_E_INVALID;
    if (!h->device_quadtree || h->knobs.serial_pipeline) {
        set_error("msorb_extract_stereo needs the device pipeline");
        return MSORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    if ((rc = ensure_batch(h, 2))) return rc;
    const int cap = capacity_of(h);
    const FrameGeom& g = h->G;
    const LevelGeom& g0 = g.lv[0];
    float smax = 0;
    for (int l = 0; l < g.nlevels; l++) smax = std::max(smax, h->scales.scale[l]);
    const int row_cap = cap * ((int)std::ceil(4.0f * smax) + 3);
    const bool bands = rows <= 4095;   // (a band record holds 12-bit rows; taller images — none of the BASELINE configs — take the row table)
    if (!bands && (size_t)(2 * rows + 1) * sizeof(int) > 60000) { set_error("image too tall for the stereo row table"); return MSORB_E_INVALID; }
    // one device block for everything that travels back: [kps 2*cap][desc 2*cap*32][u_right cap][depth cap][n_oob], and
    // one device block for the two level-0 planes (read in place by the pipeline): one copy each way
    const size_t kp_bytes = (size_t)cap * sizeof(msorb_keypoint);
    const size_t o_desc = 2 * kp_bytes, o_ur = o_desc + (size_t)2 * cap * 32, o_dp = o_ur + (size_t)cap * 4,
                 o_oob = o_dp + (size_t)cap * 4, o_cnt = o_oob + 4, out_bytes = o_oob + 16;  // [n_oob][n_left][n_right]
    const size_t plane = (size_t)g0.pitch * rows;
    if ((rc = h->d_st_block.ensure(out_bytes)) || (rc = h->d_st_img.ensure(2 * plane + 256)) ||
        (rc = h->h_img_pin.ensure(2 * plane)) || (rc = h->h_out_pin.ensure(out_bytes)) || (rc = h->d_st_sad.ensure(cap)) ||
        (rc = h->d_st_rows.ensure((size_t)rows + 1)) || (rc = h->d_st_list.ensure((size_t)row_cap * 2)))
        return rc;
    // pageable rows -> pinned planes -> device, one eye at a time: the left plane rides PCIe while the right one is staged
    hipStream_t s = h->stream;
    for (int y = 0; y < rows; y++) memcpy(h->h_img_pin.p + (size_t)y * g0.pitch, left + (size_t)y * stride_left, cols);
    if ((rc = frame_copy(h, h->d_st_img.p, h->h_img_pin.p, plane, hipMemcpyHostToDevice, s))) return rc;
    for (int y = 0; y < rows; y++) memcpy(h->h_img_pin.p + plane + (size_t)y * g0.pitch, right + (size_t)y * stride_right, cols);
    if ((rc = frame_copy(h, h->d_st_img.p + plane, h->h_img_pin.p + plane, plane, hipMemcpyHostToDevice, s))) return rc;
    LevelView l0{h->d_st_img.p, plane, g0.pitch, cols, rows};
    uint8_t* const blk = h->d_st_block.p;
    msorb_keypoint* const d_kps = reinterpret_cast<msorb_keypoint*>(blk);
    uint8_t* const d_desc = blk + o_desc;
    int counts[2] = {0, 0}, mono[2] = {0, 0};
    // (n_oob is zeroed by the layout launch that writes the band records — or by the row-table kernel of launch_stereo_match_batch)
    // what vRowIndices (Frame.cc:757-776) would hold about the right keypoints leaves the selection-layout launch as band records
    const StereoRowJob row_job{1, rows, reinterpret_cast<int2*>(h->d_st_list.p), h->d_st_list.p + 2 * (size_t)cap, reinterpret_cast<int*>(blk + o_oob)};
    h->row_job = bands ? &row_job : nullptr;
    h->defer_sync = h->skip_count_copies = true;
    rc = run_pipeline(h, l0, 2, 0, 0, d_kps, d_desc, cap, counts, mono);
    h->defer_sync = h->skip_count_copies = false;
    h->row_job = nullptr;
    if (rc) return rc;
    // stereo association on the device outputs (pair 0 = images 0 / 1)
    StereoBatchArgs b{};
    b.A.kpL = d_kps;
    b.A.descL = d_desc;
    b.A.rows0 = rows;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelView& v = h->last_pyr.lv[l];
        b.A.pyrL[l] = v.base; b.A.pyrR[l] = v.base + v.img_stride;
        b.A.pitchL[l] = b.A.pitchR[l] = v.pitch;
        b.A.rows[l] = v.h; b.A.cols[l] = v.w;
        b.A.scale[l] = h->scales.scale[l]; b.A.inv_scale[l] = h->P.inv_scale[l];
        b.img_strideL[l] = b.img_strideR[l] = v.img_stride;
    }
    b.A.mb = mb; b.A.mbf = mbf;
    b.A.u_right = reinterpret_cast<float*>(blk + o_ur); b.A.depth = reinterpret_cast<float*>(blk + o_dp);
    b.A.sad = h->d_st_sad.p; b.A.n_oob = reinterpret_cast<int*>(blk + o_oob);
    b.capacity = cap;
    b.pair_step = 2;
    b.A.kpR = d_kps + cap; b.A.descR = d_desc + (size_t)cap * 32;
    b.countsL = h->d_sel_count.p; b.countsR = h->d_sel_count.p + 1;
    if (bands) {
        b.band = reinterpret_cast<const int2*>(h->d_st_list.p);
        b.band_level_begin = h->d_st_list.p + 2 * (size_t)cap;
    } else {
        b.row_begin = h->d_st_rows.p; b.row_list = reinterpret_cast<int2*>(h->d_st_list.p); b.row_cap = row_cap;
    }
    b.counts_out = reinterpret_cast<int*>(blk + o_cnt);
    uint8_t* o = h->h_out_pin.p;
    // Without a sink the median rule rides the read-back launch (stereo_median_readback_kernel; in-process A/B, alternating blocks
    // of 100 frames, three processes: msorb_extract_stereo through the Python mirror 0.2404 / 0.2450 / 0.2414 -> 0.2368 / 0.2407 /
    // 0.2376 ms) — unless the copies are SDMA's (MSORB_FRAME_COPIES=sdma) or the block's tail is not 16-byte aligned (odd capacity).
    static const bool sdma_copies = [] { const char* e = getenv("MSORB_FRAME_COPIES"); return e && std::string(e) == "sdma"; }();
    b.median_with_readback = !sink && !sdma_copies && (o_ur & 15) == 0 &&
                             ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(blk)) & 15) == 0;
    launch_stereo_match_batch(b, 1, cap, s, /*row_table_built=*/bands);
    if (sink) {
        // the frame's read-back (190 KB) leaves on the side stream while the sink's kernels (frame grid, local points, window
        // search) run on the main one: 25 us of copy that would otherwise sit between the stereo kernels and the grid
        if (!h->ev_split) HIPCHK(hipEventCreateWithFlags(&h->ev_split, hipEventDisableTiming));
        // The sink's launches go out first and the copy is the copy kernel (in-process A/B, twelve alternating blocks of 100
        // frames, three processes: msorb_extract_stereo_frame 0.1999 -> 0.1969 ms, the motion-model call 0.2341 -> 0.2335;
        // the same order with hipMemcpyAsync: no different from before).
        HIPCHK(hipEventRecord(h->ev_split, s));
        const StereoDeviceOutputs so{d_kps, d_desc, b.A.u_right, reinterpret_cast<const int*>(blk + o_cnt), cap, s};
        if ((rc = sink(ctx, so))) { (void)hipStreamSynchronize(s); return rc; }
        HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_split, 0));
        HIPCHK(small_copy(o, blk, out_bytes, hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipStreamSynchronize(h->copy_stream));
    } else {
        if (b.median_with_readback) launch_stereo_median_readback(b, o, blk, o_ur, out_bytes, s);
        else if ((rc = frame_copy(h, o, blk, out_bytes, hipMemcpyDeviceToHost, s))) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const int nl = reinterpret_cast<const int*>(o + o_cnt)[0], nr = reinterpret_cast<const int*>(o + o_cnt)[1];
    if (nl < 0 || nr < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
    if (nl > capacity || nr > capacity) { set_error("caller capacity too small"); return MSORB_E_CAPACITY; }
    memcpy(kps_left, o, (size_t)nl * sizeof(msorb_keypoint));
    memcpy(kps_right, o + kp_bytes, (size_t)nr * sizeof(msorb_keypoint));
    memcpy(desc_left, o + o_desc, (size_t)nl * 32);
    memcpy(desc_right, o + o_desc + (size_t)cap * 32, (size_t)nr * 32);
    memcpy(u_right, o + o_ur, (size_t)nl * sizeof(float));
    memcpy(depth, o + o_dp, (size_t)nl * sizeof(float));
    if (n_oob) *n_oob = *reinterpret_cast<const int*>(o + o_oob);
    *n_left = nl;
    *n_right = nr;
    return MSORB_OK;
}

extern "C" {

// The same frame with one eye per DEVICE (BASELINE config "left/right images on 2 MI355X, gather over xGMI"): what the two
// extractor threads of Frame.cc:122-125 and the join + ComputeStereoMatches of Frame.cc:126-137 do, with the left object on
// device A and the right object on device B.  Each eye runs its own kernel chain on its own device and stream; the right
// eye's keypoints, descriptors, count and pyramid (the SAD of Frame.cc:840-865 reads both pyramids) are then copied
// device-to-device onto A (hipMemcpyPeerAsync: xGMI when A != B), an event carries the order across the two streams, and the
// stereo association runs on A.  One synchronisation, one D2H block.  A == B (two handles on one device) takes the same path.
int msorb_extract_stereo_split(msorb_extractor* L, msorb_extractor* R, const uint8_t* left, const uint8_t* right, int rows,
                               int cols, size_t stride_left, size_t stride_right, float mb, float mbf,
                               msorb_keypoint* kps_left, uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right,
                               uint8_t* desc_right, int* n_right, int capacity, float* u_right, float* depth, int* n_oob) {
    if ((L && L->pending_batch) || (R && R->pending_batch)) { set_error("a submitted batch of a handle has not been waited for"); return MSORB_E_INVALID; }
    if (!L || !R || L == R || !n_left || !n_right) return MSORB_E_INVALID;
    *n_left = *n_right = 0;
    if (n_oob) *n_oob = 0;
    if (!left || !right || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;
    if (!kps_left || !desc_left || !kps_right || !desc_right || !u_right || !depth || (int)stride_left < cols ||
        (int)stride_right < cols)
        return MSORB_E_INVALID;
    if (L->P.nfeatures != R->P.nfeatures || L->P.nlevels != R->P.nlevels || L->P.scale_factor_f != R->P.scale_factor_f) {
        set_error("msorb_extract_stereo_split: the two extractors differ in their parameters");
        return MSORB_E_INVALID;
    }
    if (L->knobs.serial_pipeline || R->knobs.serial_pipeline) { set_error("msorb_extract_stereo_split needs the device pipeline"); return MSORB_E_INVALID; }
    const int cap = capacity_of(L);
    int rc;
    // geometry / buffers of both handles (each on its own device)
    for (msorb_extractor* h : {R, L}) {
        HIPCHK(hipSetDevice(h->device));
        if ((rc = ensure_geometry(h, rows, cols))) return rc;
        if (!h->device_quadtree) { set_error("msorb_extract_stereo_split needs the device pipeline"); return MSORB_E_INVALID; }
        if ((rc = ensure_batch(h, 1))) return rc;
        if ((rc = h->h_img_pin.ensure((size_t)h->G.lv[0].pitch * rows))) return rc;
    }
    const FrameGeom& g = L->G;
    const LevelGeom& g0 = g.lv[0];
    const size_t plane = (size_t)g0.pitch * rows;
    if ((size_t)(2 * rows + 1) * sizeof(int) > 60000) { set_error("image too tall for the stereo row table"); return MSORB_E_INVALID; }
    float smax = 0;
    for (int l = 0; l < g.nlevels; l++) smax = std::max(smax, L->scales.scale[l]);
    const int row_cap = cap * ((int)std::ceil(4.0f * smax) + 3);
    // left device: one block for everything that travels back: [kps 2*cap][desc 2*cap*32][u_right cap][depth cap][n_oob][n_left][n_right]
    const size_t kp_bytes = (size_t)cap * sizeof(msorb_keypoint);
    const size_t o_desc = 2 * kp_bytes, o_ur = o_desc + (size_t)2 * cap * 32, o_dp = o_ur + (size_t)cap * 4,
                 o_oob = o_dp + (size_t)cap * 4, o_cnt = o_oob + 4, out_bytes = o_oob + 16;
    HIPCHK(hipSetDevice(L->device));
    if ((rc = L->d_st_block.ensure(out_bytes)) || (rc = L->h_out_pin.ensure(out_bytes)) || (rc = L->d_st_sad.ensure(cap)) ||
        (rc = L->d_st_rows.ensure((size_t)rows + 1)) || (rc = L->d_st_list.ensure((size_t)row_cap * 2)) ||
        (rc = L->d_gather_pyr.ensure(g.pyramid_bytes + 256)) || (rc = L->d_gather_cnt.ensure(4)))
        return rc;
    // right device: its outputs as one block [kps cap][desc cap*32]
    HIPCHK(hipSetDevice(R->device));
    if ((rc = R->d_out1.ensure(kp_bytes + (size_t)cap * 32 + 16))) return rc;
    if (!R->ev_split) HIPCHK(hipEventCreateWithFlags(&R->ev_split, hipEventDisableTiming));
    // Direct xGMI copies when the devices can reach each other.  Without peer access (or with MSORB_SPLIT_NO_PEER=1, which
    // forces this path on any pair of handles so that it is testable on one GPU) the gather is staged through pinned host
    // memory explicitly: device B -> pinned block on B's stream, event, pinned block -> device A on A's stream.
    bool peer = true;
    if (L->device != R->device) {
        static std::mutex peer_mu;
        static std::vector<std::pair<std::pair<int, int>, bool>> known;
        std::lock_guard<std::mutex> lk(peer_mu);
        auto it = std::find_if(known.begin(), known.end(), [&](const auto& e) { return e.first == std::make_pair(L->device, R->device); });
        if (it == known.end()) {
            int can = 0;
            const bool ok = hipDeviceCanAccessPeer(&can, R->device, L->device) == hipSuccess && can;
            if (ok) {
                (void)hipSetDevice(R->device); (void)hipDeviceEnablePeerAccess(L->device, 0);
                (void)hipSetDevice(L->device); (void)hipDeviceEnablePeerAccess(R->device, 0);
                (void)hipGetLastError();  // "already enabled" is fine
            }
            known.emplace_back(std::make_pair(L->device, R->device), ok);
            peer = ok;
        } else {
            peer = it->second;
        }
    }
    if (L->knobs.split_no_peer || R->knobs.split_no_peer) peer = false;
    const size_t g_kp = 0, g_desc = kp_bytes, g_cnt = g_desc + (size_t)cap * 32, g_pyr = g_cnt + 16, g_total = g_pyr + g.pyramid_bytes;
    if (!peer) {
        HIPCHK(hipSetDevice(L->device));
        if ((rc = L->h_gather.ensure(g_total))) return rc;
    }
    int counts[1] = {0}, mono[1] = {0};
    // ---- right eye: upload + chain on device B (enqueued first: the join waits for it)
    HIPCHK(hipSetDevice(R->device));
    for (int y = 0; y < rows; y++) memcpy(R->h_img_pin.p + (size_t)y * g0.pitch, right + (size_t)y * stride_right, cols);
    HIPCHK(hipMemcpyAsync(R->d_pyr.p + R->G.lv[0].plane_off, R->h_img_pin.p, plane, hipMemcpyHostToDevice, R->stream));
    {
        LevelView l0{R->d_pyr.p + R->G.lv[0].plane_off, R->G.pyramid_bytes, g0.pitch, cols, rows};
        R->defer_sync = R->skip_count_copies = true;
        rc = run_pipeline(R, l0, 1, 0, 0, reinterpret_cast<msorb_keypoint*>(R->d_out1.p), R->d_out1.p + kp_bytes, cap, counts, mono);
        R->defer_sync = R->skip_count_copies = false;
        if (rc) return rc;
    }
    uint8_t* const blk = L->d_st_block.p;
    // gather onto device A, on B's stream behind its chain
    if (peer) {
        HIPCHK(hipMemcpyPeerAsync(blk + kp_bytes, L->device, R->d_out1.p, R->device, kp_bytes, R->stream));
        HIPCHK(hipMemcpyPeerAsync(blk + o_desc + (size_t)cap * 32, L->device, R->d_out1.p + kp_bytes, R->device, (size_t)cap * 32, R->stream));
        HIPCHK(hipMemcpyPeerAsync(L->d_gather_cnt.p, L->device, R->d_sel_count.p, R->device, sizeof(int), R->stream));
        HIPCHK(hipMemcpyPeerAsync(L->d_gather_pyr.p, L->device, R->d_pyr.p, R->device, g.pyramid_bytes, R->stream));
    } else {
        uint8_t* hg = L->h_gather.p;
        HIPCHK(hipMemcpyAsync(hg + g_kp, R->d_out1.p, kp_bytes + (size_t)cap * 32, hipMemcpyDeviceToHost, R->stream));   // keypoints + descriptors
        HIPCHK(hipMemcpyAsync(hg + g_cnt, R->d_sel_count.p, sizeof(int), hipMemcpyDeviceToHost, R->stream));
        HIPCHK(hipMemcpyAsync(hg + g_pyr, R->d_pyr.p, g.pyramid_bytes, hipMemcpyDeviceToHost, R->stream));
    }
    HIPCHK(hipEventRecord(R->ev_split, R->stream));
    // ---- left eye: upload + chain on device A
    HIPCHK(hipSetDevice(L->device));
    for (int y = 0; y < rows; y++) memcpy(L->h_img_pin.p + (size_t)y * g0.pitch, left + (size_t)y * stride_left, cols);
    hipStream_t s = L->stream;
    HIPCHK(hipMemcpyAsync(L->d_pyr.p + g0.plane_off, L->h_img_pin.p, plane, hipMemcpyHostToDevice, s));
    // (n_oob is zeroed by the row-table kernel of launch_stereo_match_batch)
    {
        LevelView l0{L->d_pyr.p + g0.plane_off, g.pyramid_bytes, g0.pitch, cols, rows};
        L->defer_sync = L->skip_count_copies = true;
        rc = run_pipeline(L, l0, 1, 0, 0, reinterpret_cast<msorb_keypoint*>(blk), blk + o_desc, cap, counts, mono);
        L->defer_sync = L->skip_count_copies = false;
        if (rc) return rc;
    }
    // ---- join (Frame.cc:126-127) + ComputeStereoMatches on device A
    HIPCHK(hipStreamWaitEvent(s, R->ev_split, 0));
    if (!peer) {   // second half of the staged gather: pinned block -> device A, behind the event
        const uint8_t* hg = L->h_gather.p;
        HIPCHK(hipMemcpyAsync(blk + kp_bytes, hg + g_kp, kp_bytes, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(blk + o_desc + (size_t)cap * 32, hg + g_desc, (size_t)cap * 32, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(L->d_gather_cnt.p, hg + g_cnt, sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(L->d_gather_pyr.p, hg + g_pyr, g.pyramid_bytes, hipMemcpyHostToDevice, s));
    }
    StereoBatchArgs b{};
    b.pair_step = 1;
    b.A.kpL = reinterpret_cast<msorb_keypoint*>(blk);
    b.A.kpR = b.A.kpL + cap;
    b.A.descL = blk + o_desc;
    b.A.descR = b.A.descL + (size_t)cap * 32;
    b.countsL = L->d_sel_count.p;
    b.countsR = L->d_gather_cnt.p;
    b.A.rows0 = rows;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelView& v = L->last_pyr.lv[l];
        b.A.pyrL[l] = v.base;
        b.A.pyrR[l] = L->d_gather_pyr.p + g.lv[l].plane_off;
        b.A.pitchL[l] = v.pitch; b.A.pitchR[l] = g.lv[l].pitch;
        b.A.rows[l] = v.h; b.A.cols[l] = v.w;
        b.A.scale[l] = L->scales.scale[l]; b.A.inv_scale[l] = L->P.inv_scale[l];
        b.img_strideL[l] = v.img_stride; b.img_strideR[l] = g.pyramid_bytes;
    }
    b.A.mb = mb; b.A.mbf = mbf;
    b.A.u_right = reinterpret_cast<float*>(blk + o_ur); b.A.depth = reinterpret_cast<float*>(blk + o_dp);
    b.A.sad = L->d_st_sad.p; b.A.n_oob = reinterpret_cast<int*>(blk + o_oob);
    b.capacity = cap;
    b.row_begin = L->d_st_rows.p; b.row_list = reinterpret_cast<int2*>(L->d_st_list.p); b.row_cap = row_cap;
    b.counts_out = reinterpret_cast<int*>(blk + o_cnt);
    launch_stereo_match_batch(b, 1, cap, s);
    uint8_t* o = L->h_out_pin.p;
    HIPCHK(hipMemcpyAsync(o, blk, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const int nl = reinterpret_cast<const int*>(o + o_cnt)[0], nr = reinterpret_cast<const int*>(o + o_cnt)[1];
    if (nl < 0 || nr < 0) { set_error("keypoint capacity exceeded"); return MSORB_E_CAPACITY; }
    if (nl > capacity || nr > capacity) { set_error("caller capacity too small"); return MSORB_E_CAPACITY; }
    memcpy(kps_left, o, (size_t)nl * sizeof(msorb_keypoint));
    memcpy(kps_right, o + kp_bytes, (size_t)nr * sizeof(msorb_keypoint));
    memcpy(desc_left, o + o_desc, (size_t)nl * 32);
    memcpy(desc_right, o + o_desc + (size_t)cap * 32, (size_t)nr * 32);
    memcpy(u_right, o + o_ur, (size_t)nl * sizeof(float));
    memcpy(depth, o + o_dp, (size_t)nl * sizeof(float));
    if (n_oob) *n_oob = *reinterpret_cast<const int*>(o + o_oob);
    *n_left = nl;
    *n_right = nr;
    return MSORB_OK;
}

// ComputePyramid (ORBextractor.cc:1170-1195) alone for a batch of device-resident images: fills the handle's pyramid (level 0
// read in place) so that msorb_stereo_matches_split can read it — the right eye's levels on the device that runs the stereo
// association, when only its keypoints / descriptors were gathered from another device.  Asynchronous on the handle's stream.
int msorb_pyramid_batch(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols, size_t row_stride,
                        size_t image_stride) {
    if (h && h->pending_batch) { set_error("a submitted batch of this handle has not been waited for"); return MSORB_E_INVALID; }
    if (!h || n_images < 0) return MSORB_E_INVALID;
    if (!d_images || rows <= 0 || cols <= 0) return MSORB_E_EMPTY;
    if (n_images == 0) return MSORB_OK;
    if ((int)row_stride < cols || (n_images > 1 && image_stride < row_stride * (size_t)rows)) { set_error("bad strides"); return MSORB_E_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_geometry(h, rows, cols))) return rc;
    if ((rc = h->d_pyr.ensure((size_t)n_images * h->G.pyramid_bytes + 256))) return rc;
    LevelView l0{d_images, image_stride, (int)row_stride, cols, rows};
    const PyramidView pyr = make_view(h, h->d_pyr.p, &l0);
    launch_pyramid(pyr, h->d_taps.p, h->tap_x_off.data(), h->tap_y_off.data(), n_images, h->stream, h->sem);
    HIPCHK(hipGetLastError());
    h->last_pyr = pyr; h->last_n_images = n_images;
    h->pair_pyramids = 0; h->h_pyr_valid = false; h->compact_on_host = false;
    return MSORB_OK;
}

int msorb_pyramid_level(msorb_extractor* h, int level, const uint8_t** data, int* rows, int* cols, size_t* stride) {
    if (!h || !data || !h->geom_valid || h->last_n_images < 1 || level < 0 || level >= h->G.nlevels)
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const FrameGeom& g = h->G;
    if (h->h_pyr_async) {  // filled by the last msorb_extract call itself (msorb_extractor_set_host_pyramid)
        HIPCHK(hipStreamSynchronize(h->pyr_stream));
        // level 0 is the caller's image where the last call left it in pinned memory: image 0 of a pair call may have been a staged
        // pointer or have been moved to the second plane (msorb_extract_pair)
        *data = level == 0 ? (h->pair_pyramids == 2 && h->pair_l0[0] ? h->pair_l0[0] : h->h_img_pin.p) : h->h_pyr.p + g.lv[level].plane_off;
        if (rows) *rows = g.lv[level].h;
        if (cols) *cols = g.lv[level].w;
        if (stride) *stride = g.lv[level].pitch;
        return MSORB_OK;
    }
    if (!h->h_pyr_valid) {
        int rc;
        if ((rc = h->h_pyr.ensure(g.pyramid_bytes))) return rc;
        for (int l = 0; l < g.nlevels; l++) {
            const LevelView& v = h->last_pyr.lv[l];
            HIPCHK(hipMemcpy2D(h->h_pyr.p + g.lv[l].plane_off, g.lv[l].pitch, v.base, v.pitch, v.w, v.h,
                               hipMemcpyDeviceToHost));
        }
        h->h_pyr_valid = true;
    }
    *data = h->h_pyr.p + g.lv[level].plane_off;
    if (rows) *rows = g.lv[level].h;
    if (cols) *cols = g.lv[level].w;
    if (stride) *stride = g.lv[level].pitch;
    return MSORB_OK;
}

int msorb_debug_std_sort(int device, const uint32_t* keys, int n, int frame_form, uint32_t* order, uint32_t* sorted_keys, float* sort_us) {
    if (!keys || !order || n < 0) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(device));
    const int rc = launch_debug_sort(keys, n, frame_form, order, sorted_keys, sort_us);
    if (rc == MSORB_E_HIP) set_error(std::string("msorb_debug_std_sort: ") + hipGetErrorString(hipGetLastError()));
    return rc;
}

int msorb_debug_patch_tables(msorb_extractor* h, int8_t* pattern, int8_t* umax) {
    if (!h || !pattern || !umax) return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(download_patch_tables(pattern, umax, h->stream));
    return MSORB_OK;
}

int msorb_debug_level_size(const msorb_extractor* h, int level, int* rows, int* cols) {
    if (!h || !h->geom_valid || level < 0 || level >= h->G.nlevels) return MSORB_E_INVALID;
    *rows = h->G.lv[level].h;
    *cols = h->G.lv[level].w;
    return MSORB_OK;
}

int msorb_debug_copy_level(msorb_extractor* h, int image, int level, int blurred, uint8_t* dst) {
    if (!h || !dst || !h->geom_valid || image < 0 || image >= h->last_n_images || level < 0 || level >= h->G.nlevels)
        return MSORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const LevelView& v = blurred ? h->last_blur.lv[level] : h->last_pyr.lv[level];
    if (blurred) {   // tiled on the device: fetch the plane, undo the tiling here
        std::vector<uint8_t> t((size_t)v.pitch * ((v.h + 7) & ~7));
        HIPCHK(hipMemcpy(t.data(), v.base + (size_t)image * v.img_stride, t.size(), hipMemcpyDeviceToHost));
        for (int y = 0; y < v.h; y++)
            for (int x = 0; x < v.w; x++) dst[(size_t)y * v.w + x] = t[blur_tile_off((uint32_t)x, (uint32_t)y, (uint32_t)v.pitch)];
        return MSORB_OK;
    }
    HIPCHK(hipMemcpy2D(dst, v.w, v.base + (size_t)image * v.img_stride, v.pitch, v.w, v.h, hipMemcpyDeviceToHost));
    return MSORB_OK;
}

int msorb_debug_candidates(msorb_extractor* h, int image, int level, int* xyscore, int capacity, int* n) {
    if (!h || !n || !h->geom_valid || image < 0 || image >= h->last_n_images || level < 0 || level >= h->G.nlevels)
        return MSORB_E_INVALID;
    const int nl = h->G.nlevels;
    if (h->last_groups != 1) { set_error("candidate inspection needs a single sub-batch (n_images < 16, or msorb_extractor_set_overlap(h, 1, ...) before the call)"); return MSORB_E_INVALID; }
    if (!h->compact_on_host) {
        HIPCHK(hipSetDevice(h->device));
        int rc;
        if ((rc = fetch_candidates(h, h->last_n_images))) return rc;
    }
    const int* lc = h->h_level_count.p + (size_t)image * nl;
    const Cand16* c = h->h_compact.p + h->h_img_base.p[image];
    for (int l = 0; l < level; l++) c += lc[l];
    *n = lc[level];
    for (int i = 0; i < lc[level] && i < capacity; i++) {
        xyscore[3 * i] = c[i].x; xyscore[3 * i + 1] = c[i].y; xyscore[3 * i + 2] = c[i].score;
    }
    return MSORB_OK;
}

int msorb_distribute_quadtree(const uint16_t* xs, const uint16_t* ys, const uint16_t* scores, int n, int min_x,
                              int max_x, int min_y, int max_y, int n_features, int* kept_idx, int capacity,
                              int* n_kept) {
    if (!n_kept || n < 0 || (n > 0 && (!xs || !ys || !scores)) || max_x <= min_x || max_y <= min_y ||
        (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y)) < 1)
        return MSORB_E_INVALID;
    std::vector<Cand16> c(n);
    for (int i = 0; i < n; i++) c[i] = Cand16{xs[i], ys[i], scores[i], 0};
    std::vector<int> kept;
    distribute_quadtree(c.data(), n, min_x, max_x, min_y, max_y, n_features, kept);
    *n_kept = (int)kept.size();
    if ((int)kept.size() > capacity) return MSORB_E_CAPACITY;
    for (size_t i = 0; i < kept.size(); i++) kept_idx[i] = kept[i];
    return MSORB_OK;
}

}  // extern "C"
