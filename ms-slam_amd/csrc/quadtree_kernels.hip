// Device-side keypoint selection: one workgroup per (pyramid level, image) runs the generation-synchronous
// DistributeOctTree of quadtree_device.h on the compacted FAST candidates, then one workgroup per image lays
// out the selected keypoints in ORBextractor::operator()'s output order (ORBextractor.cc:1122-1163).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <string>

#include "lds_limit.h"
#include "orb_device.h"
#include "quadtree_device.h"
#include "quadtree_paths_device.h"
#include "quadtree_devex_device.h"
#include "gauss7_stream_device.h"
#include "stereo_rowtable_device.h"

namespace msorb {

constexpr int kQtThreads = 256;  // 256 beats 512 (0.39 vs 0.47 ms per 256 images): fewer idle waves waiting at barriers
// register-resident candidates per thread: 0 for batches (with 16 the VGPR pressure halved the resident workgroups and
// measured slower); 8 for single frames, where a 1024-thread instance then holds a whole level (<= 8192 candidates) in
// registers and the point passes stop waiting on global memory
constexpr int kQtPointsPerThreadFrame = 8;
#ifndef QT_BATCH_PC
#define QT_BATCH_PC 24
#endif
constexpr int kQtPointsPerThreadBatch = QT_BATCH_PC;   // x 256 threads = 6144 candidates of a level in registers, the rest through global memory (select stage per 256 images: 12-20: 0.142-0.143, 24: 0.139-0.140, 28: 0.145, 32: 0.167 ms — spills)

#ifndef MSORB_QT_PATH_MIN_N_FACTOR
#define MSORB_QT_PATH_MIN_N_FACTOR 2
#endif
template <int PC, bool FRAME>
__device__ __forceinline__ void quadtree_select_body(const QtLevels& lv, const Cand16* __restrict__ compact,
                                                     const int* __restrict__ img_base,
                                                     const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                     int* __restrict__ sel_pt /* [img][sel_stride] candidate idx */,
                                                     int* __restrict__ sel_n /* [img][nlevels] */, int sel_stride,
                                                     int ws_N, int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    extern __shared__ __attribute__((aligned(16))) char qt_mem[];
    // grid = (image, level): consecutive workgroups (dealt round-robin to the 8 XCDs) are different images of one
    // level, so the heavy level-0 instances are spread over all XCDs instead of piling up on one
    const int level = blockIdx.y, img = blockIdx.x;
    const int* lc = level_count + (size_t)img * lv.nlevels;
    int off = img_base[img];
    for (int l = 0; l < level; l++) off += lc[l];
    const int n = lc[level];
    // mixed launch: the levels with many candidates get every wave of the workgroup, the others only the first small_nt threads
    // (their remaining waves leave at once; s_barrier counts the waves that are still there)
    const int nt_eff = level < big_levels ? (int)blockDim.x : min((int)blockDim.x, small_nt);
    if ((int)threadIdx.x >= nt_eff) return;
    qt::Workspace w;
    qt::workspace_carve(w, qt_mem, ws_N, ws_nini);
    DevExT<FRAME> ex;
    ex.kLaneSort = !(debug & 0x100);
    debug &= 0xff;
    ex.dbg = debug;
    ex.nt = nt_eff;
    int* out = sel_pt + (size_t)img * sel_stride + lv.sel_off[level];
    int kept = -1;
    // selection by quadrant path (quadtree_paths_device.h) when the tree fits the tables the LDS behind the workspace holds
    // A sparse level (n <= 2 N) takes the general form at once: its tree grows until (almost) every candidate is alone — 8-10
    // generations for neighbours two pixels apart, below any table the LDS holds, or tables of six generations whose zeroing and
    // summing cost more than the general form's passes over a few hundred candidates (low-texture stereo frame: 0.206 ms with an
    // attempt on every level, 0.196 with attempts on N < n <= 2 N only, general form everywhere 0.188-0.193).
    const int path_min_n = MSORB_QT_PATH_MIN_N_FACTOR * lv.quota[level];
    if (path_cap > 0 && n > path_min_n && (debug == 0 || debug == 3)) {
        const int N = lv.quota[level], n_ini = lv.n_ini[level];
        qt::PathTables pt;
        qt::path_tables_carve(pt, qt_mem + qt::workspace_bytes(ws_N, ws_nini), n_ini, qt::path_gmax(N, n_ini, path_cap), lv.W[level], lv.H[level]);
        kept = qt::select_paths<PC>(ex, reinterpret_cast<const qt::Pt*>(compact + off), n, lv.W[level], lv.H[level], N, w, pt, out);
    }
    if (kept < 0) {
        {
            // (opaque to the optimiser: with the candidate loads and the workspace pointers of the two forms merged, their bodies
            // shared one register allocation — 113 VGPRs + 104 bytes of scratch per lane against 82 / 95 and none on their own)
            asm volatile("" : "+s"(off) :: "memory");
            qt::workspace_carve(w, qt_mem, ws_N, ws_nini);
        }
        kept = qt::select<PC>(ex, reinterpret_cast<const qt::Pt*>(compact + off), n, label + off, lv.W[level], lv.H[level],
                              lv.quota[level], w, out, debug);
    }
    if (threadIdx.x == 0) sel_n[(size_t)img * lv.nlevels + level] = kept;
    ex.dump();
}
template <int PC>
__global__ __launch_bounds__(1024) void quadtree_select_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                              const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                              int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                              int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    quadtree_select_body<PC, PC == kQtPointsPerThreadFrame>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}
// The selection of a FRAME with the blur of its levels in the same launch: the selection keeps n_images x nlevels workgroups = 16
// of the chip's 256 CUs busy for 35-40 us (its level-0 instance is a chain of dependent steps), the blur — 9 us of bandwidth
// work that only the descriptor stage reads — runs on the others and leaves the frame's critical path (it sat beside FAST, which
// it lengthened by 3.5 us, and on a side stream before that: an event record, two stream waits, ~8 us of fork / join).  Blocks
// (x, y >= nlevels) are blur blocks: 1024 threads = four 256-thread blocks of gauss7_stream_kernel's numbering, waves independent.
template <int PC>
__global__ __launch_bounds__(1024) void quadtree_select_blur_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                   const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                                   int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                                   int ws_nini, int debug, int big_levels, int small_nt, int path_cap,
                                                                   FrameBlurJob blur) {
    if ((int)blockIdx.y >= lv.nlevels) {
        const int b = ((int)blockIdx.y - lv.nlevels) * (int)gridDim.x + (int)blockIdx.x;
        const int tile = 4 * b + (int)(threadIdx.x >> 8);
        if (tile < blur.blocks) gauss7_stream_body<kGaussRows>(blur.src, blur.dst, blur.plan, tile, (int)((threadIdx.x >> 6) & 3));
        return;
    }
    quadtree_select_body<PC, true>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}
// Batch form: 256-thread instances whose first PC x 256 candidates stay in registers for the whole selection (a level-0 instance of
// the BASELINE geometries has ~6 800): the per-generation point passes then touch no global memory at all.
template <int PC>
__global__ __launch_bounds__(256, 4) void quadtree_select_batch_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                   const int* __restrict__ level_count, uint16_t* __restrict__ label,
                                                                   int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                                   int ws_nini, int debug, int big_levels, int small_nt, int path_cap) {
    quadtree_select_body<PC, false>(lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, ws_N, ws_nini, debug, big_levels, small_nt, path_cap);
}

// One workgroup per image: records in level-major / quadtree order; output row = mono index from the front for
// keypoints outside [lap0, lap1], stereo index from the back for those inside (ORBextractor.cc:1153-1162).
// A stereo frame's launch also writes the band record of every right keypoint (StereoRowJob, stereo_rowtable_device.h): what
// Frame::ComputeStereoMatches' vRowIndices would hold about it, from the same selection records.
__global__ __launch_bounds__(256) void quadtree_layout_kernel(QtLevels lv, const Cand16* __restrict__ compact,
                                                              const int* __restrict__ img_base,
                                                              const int* __restrict__ level_count,
                                                              const int* __restrict__ sel_pt, const int* __restrict__ sel_n,
                                                              int sel_stride, LevelScale scales, int lap0, int lap1,
                                                              int capacity, SelRec* __restrict__ sel,
                                                              int* __restrict__ sel_count, int* __restrict__ mono_out, int n_images,
                                                              StereoRowJob job) {
    __shared__ int lvl_begin[kMaxLevels + 1], cand_begin[kMaxLevels + 1];
    __shared__ int part[256];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int a = 0, c = img_base[img];
        for (int l = 0; l < lv.nlevels; l++) {
            lvl_begin[l] = a; cand_begin[l] = c;
            a += sel_n[(size_t)img * lv.nlevels + l];
            c += level_count[(size_t)img * lv.nlevels + l];
        }
        lvl_begin[lv.nlevels] = a;
    }
    __syncthreads();
    const int n_all = lvl_begin[lv.nlevels];
    const int n = min(n_all, min(capacity, sel_stride));
    const int per = (n + 255) / 256;
    const int b = tid * per, e = min(b + per, n);
    SelRec* out = sel + (size_t)img * sel_stride;
    int2* const bands = job.band && img == job.right_img ? job.band : nullptr;
    if (bands && tid == 0 && job.n_oob) *job.n_oob = 0;
    if (lap1 < kMinBorder) {
        // no lapping area (mono / rectified stereo: every x is >= kMinBorder, so fx <= lap1 never holds): output row =
        // selection order, no second pass.  Items are taken 256 apart, four at a time, so that the two dependent loads
        // of an item (selected index -> candidate) are in flight for four items at once.
        for (int base = 0; base < n; base += 4 * 256) {
            int gg[4], ll[4], pt[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                gg[u] = base + u * 256 + tid;
                int l = 0;
                if (gg[u] < n) while (gg[u] >= lvl_begin[l + 1]) l++;
                ll[u] = l;
                pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (gg[u] - lvl_begin[l])] : 0;
            }
            Cand16 cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) cc[u] = gg[u] < n ? compact[cand_begin[ll[u]] + pt[u]] : Cand16{};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (gg[u] >= n) continue;
                SelRec r;
                r.x = (uint16_t)(cc[u].x + kMinBorder); r.y = (uint16_t)(cc[u].y + kMinBorder);
                r.score = cc[u].score; r.level = (uint8_t)ll[u]; r.pad = 0;
                r.dst = gg[u];
                out[gg[u]] = r;
                if (bands) {   // kp.pt = level coordinates times the level's scale factor, as the descriptor stage writes them (ORBextractor.cc:1149-1151)
                    const float fx = (float)r.x, fy = (float)r.y;
                    const int l = ll[u];
                    bands[gg[u]] = stereo_band_record(l ? __fmul_rn(fx, scales.scale[l]) : fx, l ? __fmul_rn(fy, scales.scale[l]) : fy, l, scales.scale, job.rows0);
                }
            }
        }
        if (tid == 0) {
            sel_count[img] = n_all > n ? -n_all : n;
            mono_out[img] = n;
        }
        return;
    }
    int lap_cnt = 0;
    for (int g = b; g < e; g++) {
        int l = 0;
        while (g >= lvl_begin[l + 1]) l++;
        const int pt = sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (g - lvl_begin[l])];
        const Cand16 c = compact[cand_begin[l] + pt];
        SelRec r;
        r.x = (uint16_t)(c.x + kMinBorder); r.y = (uint16_t)(c.y + kMinBorder);
        r.score = c.score; r.level = (uint8_t)l; r.pad = 0;
        const float fx = l ? __fmul_rn((float)r.x, scales.scale[l]) : (float)r.x;
        const bool lap = fx >= (float)lap0 && fx <= (float)lap1;
        r.dst = lap ? -1 : 0;  // provisional
        lap_cnt += lap;
        out[g] = r;
    }
    part[tid] = lap_cnt;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; }
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n - acc;
    }
    __syncthreads();
    int laps = part[tid];
    for (int g = b; g < e; g++) {
        if (out[g].dst < 0) { out[g].dst = n - 1 - laps; laps++; }
        else out[g].dst = g - laps;
    }
}

// Batches without a lapping area: the same records, but stored in the order the DESCRIPTOR stage should work through them —
// level-major as before, inside a level by tiles of 64 x 32 pixels, rows of tiles in serpentine order.  Output rows do not
// move (`dst` is the selection order, as in the plain form): only which keypoints the four waves of a descriptor workgroup,
// and the workgroups that share a CU, hold at the same time.  Keypoints in quadtree order are scattered over the level; in
// tile order their IC-angle patches and blurred neighbourhoods share cache lines, and describe_kernel — bound by the lines
// it pulls from L2 — is 8-9 % shorter (tools/experiments/README.md has the tile shapes that were measured).
// A counting sort in LDS: a pass that builds the records and takes a rank inside the record's tile (LDS atomic: the order
// inside a tile is whatever the atomics make it, and irrelevant), a scan over the tiles, a pass that stores.
struct TileOrder {
    int bin_begin[kMaxLevels + 1];   // first bin of each level; bins of a level = its tiles, row-major
    int ntx[kMaxLevels];             // tiles per row of tiles
    int cap;                         // records the LDS block holds (>= min(capacity, sel_stride))
};
// 256 threads: alone the launch takes 18 us where 1024 threads take 14.5 (every record's two dependent loads in flight at once),
// but a 16-wave workgroup has to wait for a CU with sixteen free wave slots beside the other batch's kernels: the pipelined step
// reads 1.048 ms with 256 threads, 1.079 with 1024 and 1.069 in selection order (five alternations on one box).
constexpr int kSortedLayoutThreads = 256;
__global__ __launch_bounds__(kSortedLayoutThreads) void quadtree_layout_sorted_kernel(QtLevels lv, const Cand16* __restrict__ compact,
                                                                     const int* __restrict__ img_base, const int* __restrict__ level_count,
                                                                     const int* __restrict__ sel_pt, const int* __restrict__ sel_n,
                                                                     int sel_stride, int capacity, SelRec* __restrict__ sel,
                                                                     int* __restrict__ sel_count, int* __restrict__ mono_out, TileOrder order) {
    extern __shared__ uint32_t tile_lds[];   // [bins] counters, then starts | [cap] records (2 dwords) | [cap] ranks (16 bit)
    __shared__ int lvl_begin[kMaxLevels + 1], cand_begin[kMaxLevels + 1], bin_begin[kMaxLevels + 1], ntx[kMaxLevels];
    __shared__ int part[256];
    constexpr int T = kSortedLayoutThreads;
    static_assert(T >= 256, "the scan over the tiles runs on the first 256 threads");
    const int img = blockIdx.x, tid = threadIdx.x;
    const int nb = order.bin_begin[lv.nlevels];
    uint32_t* const cnt = tile_lds;
    uint32_t* const rec = tile_lds + ((nb + 3) & ~3);
    uint16_t* const rank = reinterpret_cast<uint16_t*>(rec + 2 * (size_t)order.cap);
    if (tid == 0) {
        int a = 0, c = img_base[img];
        for (int l = 0; l < lv.nlevels; l++) {
            lvl_begin[l] = a; cand_begin[l] = c;
            a += sel_n[(size_t)img * lv.nlevels + l];
            c += level_count[(size_t)img * lv.nlevels + l];
        }
        lvl_begin[lv.nlevels] = a;
    }
    if (tid <= lv.nlevels) bin_begin[tid] = order.bin_begin[tid];
    if (tid < lv.nlevels) ntx[tid] = order.ntx[tid];
    for (int i = tid; i < nb; i += T) cnt[i] = 0;
    __syncthreads();
    const int n_all = lvl_begin[lv.nlevels];
    const int n = min(n_all, min(capacity, sel_stride));
    SelRec* out = sel + (size_t)img * sel_stride;
    auto bin_of = [&](uint32_t x, uint32_t y, int l) -> int {
        const int w = ntx[l], ty = (int)(y >> 5);
        int tx = min((int)(x >> 6), w - 1);
        if (ty & 1) tx = w - 1 - tx;
        return min(bin_begin[l] + ty * w + tx, bin_begin[l + 1] - 1);
    };
    // pass 1 (items T apart, two at a time: the two dependent loads of an item are in flight for all items at once)
    for (int base = 0; base < n; base += 2 * T) {
        int gg[2], ll[2], pt[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            gg[u] = base + u * T + tid;
            int l = 0;
            if (gg[u] < n) while (gg[u] >= lvl_begin[l + 1]) l++;
            ll[u] = l;
            pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + lv.sel_off[l] + (gg[u] - lvl_begin[l])] : 0;
        }
        Cand16 cc[2];
#pragma unroll
        for (int u = 0; u < 2; u++) cc[u] = gg[u] < n ? compact[cand_begin[ll[u]] + pt[u]] : Cand16{};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (gg[u] >= n) continue;
            const uint32_t x = (uint32_t)cc[u].x + kMinBorder, y = (uint32_t)cc[u].y + kMinBorder;
            rec[2 * gg[u] + 0] = x | (y << 16);
            rec[2 * gg[u] + 1] = (uint32_t)cc[u].score | ((uint32_t)ll[u] << 16);
            rank[gg[u]] = (uint16_t)atomicAdd(&cnt[bin_of(x, y, ll[u])], 1u);
        }
    }
    __syncthreads();
    // exclusive scan over the tiles: a run of consecutive bins per thread of the first four waves
    const int per = (nb + 255) / 256, b0 = min(tid * per, nb), b1 = tid < 256 ? min(b0 + per, nb) : b0;
    int mine = 0;
    for (int i = b0; i < b1; i++) mine += (int)cnt[i];
    if (tid < 256) part[tid] = mine;
    __syncthreads();
    if (tid < 256) {
        int run = 0;
        for (int i = 0; i < tid; i++) run += part[i];
        for (int i = b0; i < b1; i++) { const int c = (int)cnt[i]; cnt[i] = (uint32_t)run; run += c; }
    }
    __syncthreads();
    // pass 2
    for (int g = tid; g < n; g += T) {
        const uint32_t xy = rec[2 * g], sl = rec[2 * g + 1];
        SelRec r;
        r.x = (uint16_t)(xy & 0xffffu); r.y = (uint16_t)(xy >> 16);
        r.score = (uint16_t)(sl & 0xffffu); r.level = (uint8_t)(sl >> 16); r.pad = 0;
        r.dst = g;
        out[(int)cnt[bin_of(r.x, r.y, r.level)] + (int)rank[g]] = r;
    }
    if (tid == 0) {
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n;
    }
}

// The layout of a frame or two without a lapping area: 1024 threads (two records each, both dependent load pairs in flight), the
// per-level offsets from uniform loads in every thread (no LDS, no barrier) — the 256-thread form above spent 11 us of a stereo
// frame here, most of it one thread's chain of sixteen dependent loads and four sequential rounds of two.
__global__ __launch_bounds__(1024) void quadtree_layout_frame_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                    const int* __restrict__ level_count, const int* __restrict__ sel_pt,
                                                                    const int* __restrict__ sel_n, int sel_stride, LevelScale scales, int capacity,
                                                                    SelRec* __restrict__ sel, int* __restrict__ sel_count, int* __restrict__ mono_out,
                                                                    StereoRowJob job) {
    const int img = blockIdx.x, tid = threadIdx.x;
    int lb[kMaxLevels + 1], cb[kMaxLevels];
    {
        int sn[kMaxLevels], lc[kMaxLevels];
#pragma unroll
        for (int l = 0; l < kMaxLevels; l++) {
            sn[l] = l < lv.nlevels ? sel_n[(size_t)img * lv.nlevels + l] : 0;
            lc[l] = l < lv.nlevels ? level_count[(size_t)img * lv.nlevels + l] : 0;
        }
        int a = 0, c = img_base[img];
#pragma unroll
        for (int l = 0; l < kMaxLevels; l++) { lb[l] = a; cb[l] = c; a += sn[l]; c += lc[l]; }
        lb[kMaxLevels] = a;
    }
    const int n_all = lb[kMaxLevels];
    const int n = min(n_all, min(capacity, sel_stride));
    SelRec* out = sel + (size_t)img * sel_stride;
    int2* const bands = job.band && img == job.right_img ? job.band : nullptr;
    if (bands && tid == 0 && job.n_oob) *job.n_oob = 0;
    if (bands && job.level_begin && tid <= kMaxLevels) {
        int v = n;
#pragma unroll
        for (int k = 0; k <= kMaxLevels; k++) if (tid == k) v = min(lb[k], n);
        job.level_begin[tid] = v;
    }
    for (int base = 0; base < n; base += 2 * 1024) {
        int gg[2], ll[2], pt[2], cbeg[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            gg[u] = base + u * 1024 + tid;
            int l = 0, lbeg = 0, off = lv.sel_off[0];
            cbeg[u] = cb[0];
#pragma unroll
            for (int k = 1; k < kMaxLevels; k++)
                if (k < lv.nlevels && gg[u] >= lb[k]) { l = k; lbeg = lb[k]; off = lv.sel_off[k]; cbeg[u] = cb[k]; }
            ll[u] = l;
            pt[u] = gg[u] < n ? sel_pt[(size_t)img * sel_stride + off + (gg[u] - lbeg)] : 0;
        }
        Cand16 cc[2];
#pragma unroll
        for (int u = 0; u < 2; u++) cc[u] = gg[u] < n ? compact[cbeg[u] + pt[u]] : Cand16{};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (gg[u] >= n) continue;
            SelRec r;
            r.x = (uint16_t)(cc[u].x + kMinBorder); r.y = (uint16_t)(cc[u].y + kMinBorder);
            r.score = cc[u].score; r.level = (uint8_t)ll[u]; r.pad = 0;
            r.dst = gg[u];
            out[gg[u]] = r;
            if (bands) {   // kp.pt = level coordinates times the level's scale factor, as the descriptor stage writes them (ORBextractor.cc:1149-1151)
                const float fx = (float)r.x, fy = (float)r.y;
                const int l = ll[u];
                bands[gg[u]] = stereo_band_record(l ? __fmul_rn(fx, scales.scale[l]) : fx, l ? __fmul_rn(fy, scales.scale[l]) : fy, l, scales.scale, job.rows0);
            }
        }
    }
    if (tid == 0) {
        sel_count[img] = n_all > n ? -n_all : n;   // negative = capacity exceeded (host turns it into an error)
        mono_out[img] = n;
    }
}

int launch_quadtree(const QtLevels& lv, const Cand16* compact, const int* img_base, const int* level_count,
                     uint16_t* label, int* sel_pt, int* sel_n, int sel_stride, const LevelScale& scales, int lap0,
                     int lap1, int capacity, SelRec* sel, int* sel_count, int* mono, int n_images, hipStream_t s,
                     const StereoRowJob* row_job, const FrameBlurJob* blur_job, bool* blur_carried, char* global_ws, uint32_t* global_label) {
    if (blur_carried) *blur_carried = false;
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    const size_t lds = qt::workspace_bytes(maxN, max_ini);
#ifdef MSORB_QT_MARKS   // profiling build (tools/qt_marks.sh): MSORB_QT_DEBUG=3 prints the per-phase timestamps of instance (0, 0)
    static const int dbg0 = getenv("MSORB_QT_DEBUG") ? atoi(getenv("MSORB_QT_DEBUG")) : 0;
#else
    const int dbg0 = 0;
#endif
#ifdef MSORB_AB_BUILD    // in-process A/B builds (tools/ab_inprocess.py) read the switch per call; the library proper once
    const bool lane_sort_off = getenv("MSORB_QT_LANE_SORT") && atoi(getenv("MSORB_QT_LANE_SORT")) == 0;
#else
    static const bool lane_sort_off = getenv("MSORB_QT_LANE_SORT") && atoi(getenv("MSORB_QT_LANE_SORT")) == 0;
#endif
    const int dbg = dbg0 | (lane_sort_off ? 0x100 : 0);
    // Workgroup size by batch size: the generations are chains of dependent LDS round trips, hidden only by other
    // waves.  A big batch has other workgroups on the CU for that (256 threads: least barrier idling, best
    // throughput); a frame or two has nothing else, so the instance itself brings the waves (1024 threads).
    const int qt_threads = n_images <= 4 ? 1024 : n_images <= 16 ? 512 : kQtThreads;
    const int big_levels = kMaxLevels, small_nt = qt_threads;
    auto raise_lds = [&](const void* fn) -> bool {   // quotas beyond ~700 keypoints per level: past the default dynamic-LDS limit
        if (lds <= 64 * 1024) return true;
        if ((long long)lds <= dynamic_lds_room(fn)) return true;   // raised once per device to all the LDS there is, never lowered
        set_last_error("quadtree: the device refuses " + std::to_string(lds) + " bytes of LDS per workgroup");
        return false;
    };
    // path tables behind the workspace: the deepest generation (<= max_cap) whose tables the LDS still holds for every level
    static const bool paths_off = getenv("MSORB_QT_PATHS") && atoi(getenv("MSORB_QT_PATHS")) == 0;
    auto path_tables = [&](const void* fn, int max_cap, long long budget, int& path_cap, size_t& lds_paths) {
        path_cap = 0; lds_paths = 0;
        if (paths_off) return;
        const long long room = std::min<long long>(dynamic_lds_room(fn), budget);
        for (int cap = max_cap; cap >= 2 && !path_cap; cap--) {
            size_t need = 0;
            for (int l = 0; l < lv.nlevels; l++)
                need = max(need, qt::path_tables_bytes(lv.n_ini[l], qt::path_gmax(1 << 20, lv.n_ini[l], cap), lv.W[l], lv.H[l]));
            if ((long long)(lds + need) <= room) { path_cap = cap; lds_paths = need; }
        }
    };
    int path_cap = 0;
    size_t lds_paths = 0;
    if (global_ws && global_label) {   // the workspace of these quotas does not fit a workgroup's LDS: the selection over global memory
        launch_quadtree_select_global(lv, compact, img_base, level_count, global_label, sel_pt, sel_n, sel_stride, n_images, global_ws, s);
    } else if (qt_threads == 256) {
        const void* fn = reinterpret_cast<const void*>(quadtree_select_batch_kernel<kQtPointsPerThreadBatch>);
        // batches: tables of four generations keep three workgroups on a CU (workspace 39.7 KB + 11.7 KB at the KITTI quota; five
        // generations leave two: select stage alone 0.120 / 0.159 ms per 256 images against 0.139 without tables, pipelined step
        // 1.091 / 1.095 against 1.131 ms); a level whose tree grows deeper runs the general form
        static const int batch_cap = getenv("MSORB_QT_BATCH_PATHS") ? atoi(getenv("MSORB_QT_BATCH_PATHS")) : 4;
        if (batch_cap > 0) path_tables(fn, batch_cap, 80 * 1024, path_cap, lds_paths);
        if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
        hipLaunchKernelGGL(quadtree_select_batch_kernel<kQtPointsPerThreadBatch>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv,
                           compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
    } else if (qt_threads == 1024) {
        if (blur_job && blur_job->blocks > 0) {   // the frame's blur rides this launch (quadtree_select_blur_kernel)
            const void* fn = reinterpret_cast<const void*>(quadtree_select_blur_kernel<kQtPointsPerThreadFrame>);
            path_tables(fn, 6, 1 << 30, path_cap, lds_paths);
            if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
            const int wgs = (blur_job->blocks + 3) / 4, extra_rows = (wgs + n_images - 1) / n_images;
            hipLaunchKernelGGL(quadtree_select_blur_kernel<kQtPointsPerThreadFrame>, dim3(n_images, lv.nlevels + extra_rows), dim3(qt_threads),
                               lds + lds_paths, s, lv, compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg,
                               big_levels, small_nt, path_cap, *blur_job);
            if (blur_carried) *blur_carried = true;
        } else {
        const void* fn = reinterpret_cast<const void*>(quadtree_select_kernel<kQtPointsPerThreadFrame>);
        path_tables(fn, 6, 1 << 30, path_cap, lds_paths);
        if (!path_cap && !raise_lds(fn)) return MSORB_E_HIP;
        hipLaunchKernelGGL(quadtree_select_kernel<kQtPointsPerThreadFrame>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv,
                           compact, img_base, level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
        }
    } else {
        if (!raise_lds(reinterpret_cast<const void*>(quadtree_select_kernel<0>))) return MSORB_E_HIP;
        path_tables(reinterpret_cast<const void*>(quadtree_select_kernel<0>), 5, 80 * 1024, path_cap, lds_paths);
        hipLaunchKernelGGL(quadtree_select_kernel<0>, dim3(n_images, lv.nlevels), dim3(qt_threads), lds + lds_paths, s, lv, compact, img_base,
                           level_count, label, sel_pt, sel_n, sel_stride, maxN, max_ini, dbg, big_levels, small_nt, path_cap);
    }
    const StereoRowJob job = row_job && lap1 < kMinBorder && row_job->right_img < n_images ? *row_job : StereoRowJob{};
    if (n_images <= 4 && lap1 < kMinBorder)
        hipLaunchKernelGGL(quadtree_layout_frame_kernel, dim3(n_images), dim3(1024), 0, s, lv, compact, img_base, level_count, sel_pt, sel_n,
                           sel_stride, scales, capacity, sel, sel_count, mono, job);
    else {
        // batches without a lapping area: records in the descriptor stage's tile order (MSORB_DESC_ORDER=0: selection order)
        static const bool tile_order_off = getenv("MSORB_DESC_ORDER") && atoi(getenv("MSORB_DESC_ORDER")) == 0;
        TileOrder order{};
        size_t order_lds = 0;
        if (!tile_order_off && n_images > 4 && lap1 < kMinBorder && !job.band) {
            int nb = 0;
            for (int l = 0; l < lv.nlevels; l++) {
                order.bin_begin[l] = nb;
                order.ntx[l] = ((lv.W[l] + 2 * kMinBorder) >> 6) + 1;
                nb += order.ntx[l] * (((lv.H[l] + 2 * kMinBorder) >> 5) + 1);
            }
            order.bin_begin[lv.nlevels] = nb;
            order.cap = std::min(capacity, sel_stride);
            order_lds = ((size_t)((nb + 3) & ~3) + 2 * (size_t)order.cap) * 4 + 2 * (size_t)order.cap + 8;
            if (order.cap <= 0 || order.cap > 65535 || order_lds > 60 * 1024) order_lds = 0;   // (a geometry this form was not sized for: plain order)
        }
        if (order_lds)
            hipLaunchKernelGGL(quadtree_layout_sorted_kernel, dim3(n_images), dim3(kSortedLayoutThreads), order_lds, s, lv, compact, img_base, level_count, sel_pt,
                               sel_n, sel_stride, capacity, sel, sel_count, mono, order);
        else
            hipLaunchKernelGGL(quadtree_layout_kernel, dim3(n_images), dim3(256), 0, s, lv, compact, img_base, level_count, sel_pt,
                               sel_n, sel_stride, scales, lap0, lap1, capacity, sel, sel_count, mono, n_images, job);
    }
    return MSORB_OK;
}
// Test hook (msorb_debug_std_sort): the careful loop's std::sort restatement ALONE, as the selection kernels run it — FRAME: the
// 1024-thread queue form with wave_introsort64 below 65 items; otherwise the 256-thread level-synchronous form — on explicit keys;
// items = (key, position in the input).  The result must be the permutation libstdc++'s std::sort produces (tests/test_quadtree_sort_gpu.py).
template <bool FRAME>
__global__ __launch_bounds__(FRAME ? 1024 : 256) void debug_sort_kernel(const uint32_t* __restrict__ keys, int n, int m, uint32_t* __restrict__ out_key,
                                                                        uint32_t* __restrict__ out_node, int lane_sort, long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char mem[];
    char* p = mem;
    QT_LDS qt::SortItem* items = (QT_LDS qt::SortItem*)p; p += (size_t)(m + 4) * sizeof(qt::SortItem);
    qt::ParScratch ps;
    ps.tmp = (QT_LDS qt::SortItem*)p; p += (size_t)(m + 4) * sizeof(qt::SortItem);
    ps.gpos = (QT_LDS uint16_t*)p; p += (size_t)(m + 4) * sizeof(uint16_t);
    ps.lpos = (QT_LDS uint16_t*)p; p += (size_t)(m + 4) * sizeof(uint16_t);
    QT_LDS int* stack = (QT_LDS int*)p; p += 2 * 3 * (size_t)qt::stack_ranges(m) * sizeof(int);
    ps.stack_half = 3 * qt::stack_ranges(m);
    ps.scan_tmp = (QT_LDS int*)p; p += 16 * sizeof(int);
    ps.sc = (QT_LDS int*)p;
    for (int i = threadIdx.x; i < n; i += blockDim.x) items[i] = qt::SortItem{keys[i], (uint32_t)i};
    __syncthreads();
    DevExT<FRAME> ex;
    ex.nt = (int)blockDim.x;
    ex.kLaneSort = lane_sort != 0;
    const long long t0 = wall_clock64();
    ex.sort(items, n, stack, ps);
    __syncthreads();
    if (threadIdx.x == 0 && ticks) *ticks = wall_clock64() - t0;   // 100 MHz constant clock: 10 ns per tick
    for (int i = threadIdx.x; i < n; i += blockDim.x) { out_key[i] = items[i].key; out_node[i] = items[i].node; }
}
int launch_debug_sort(const uint32_t* h_keys, int n, int frame_form, uint32_t* h_nodes, uint32_t* h_keys_out, float* sort_us) {
    if (n < 0 || n > 4000) return MSORB_E_INVALID;
    if (n == 0) return MSORB_OK;
    const int m = ((n + 15) & ~1) | 0;   // (m + 4 items; gpos / lpos stay 4-byte aligned for an even m)
    const size_t lds = 2 * (size_t)(m + 4) * sizeof(qt::SortItem) + 2 * (size_t)(m + 4) * sizeof(uint16_t) +
                       2 * 3 * (size_t)qt::stack_ranges(m) * sizeof(int) + 20 * sizeof(int) + 64;
    uint32_t *d_in = nullptr, *d_k = nullptr, *d_n = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, 3 * (size_t)n * sizeof(uint32_t) + 16);
    if (e != hipSuccess) return MSORB_E_HIP;
    d_k = d_in + n; d_n = d_k + n;
    long long* d_ticks = reinterpret_cast<long long*>(d_in + ((3 * (size_t)n + 1) & ~size_t(1)));
    const int lane_sort = !(frame_form & 2);
    frame_form &= 1;
    e = hipMemcpy(d_in, h_keys, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (frame_form) {
            if ((long long)lds > dynamic_lds_room(reinterpret_cast<const void*>(debug_sort_kernel<true>))) { (void)hipFree(d_in); return MSORB_E_INVALID; }
            hipLaunchKernelGGL(debug_sort_kernel<true>, dim3(1), dim3(1024), lds, 0, d_in, n, m, d_k, d_n, lane_sort, d_ticks);
        } else {
            if ((long long)lds > dynamic_lds_room(reinterpret_cast<const void*>(debug_sort_kernel<false>))) { (void)hipFree(d_in); return MSORB_E_INVALID; }
            hipLaunchKernelGGL(debug_sort_kernel<false>, dim3(1), dim3(256), lds, 0, d_in, n, m, d_k, d_n, lane_sort, d_ticks);
        }
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(h_nodes, d_n, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_keys_out) e = hipMemcpy(h_keys_out, d_k, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && sort_us) { long long t = 0; e = hipMemcpy(&t, d_ticks, sizeof(t), hipMemcpyDeviceToHost); *sort_us = (float)t * 0.01f; }
    (void)hipFree(d_in);
    return e == hipSuccess ? MSORB_OK : MSORB_E_HIP;
}

size_t quadtree_lds_bytes(const QtLevels& lv) {
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    return qt::workspace_bytes(maxN, max_ini);
}

}  // namespace msorb
