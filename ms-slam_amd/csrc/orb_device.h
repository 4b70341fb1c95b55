// Structures shared by the host orchestration (extractor.hip) and the kernels (orb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/msorb.h"
#include "orb_host.h"

namespace msorb {
void set_last_error(const std::string& s);

struct LevelView {  // one pyramid level of a batch: image i's plane at base + i*img_stride
    const uint8_t* base;
    size_t img_stride;
    int pitch, w, h;
};
// The blurred planes (read by describe_kernel only) are stored in 4 x 4-pixel blocks of 16 bytes (4 rows of 4 pixels), the blocks
// of four image rows side by side: 128 bytes = 32 pixels x 4 rows.  The blur kernel writes a lane's block with one 16-byte
// store (8 lanes = one full line), and the 37 x 37 neighbourhood a descriptor samples lies in 10 x 10 blocks = ~23 cache lines
// instead of ~45 row segments of 40 bytes each.  In a blurred LevelView `pitch` is the row pitch of the raw plane (a multiple
// of 64); the plane holds ceil8(h) rows.
__host__ __device__ inline uint32_t blur_tile_off(uint32_t x, uint32_t y, uint32_t pitch) {
    return (y >> 2) * (pitch * 4u) + (x >> 2) * 16u + (y & 3u) * 4u + (x & 3u);
}
struct PyramidView {
    LevelView lv[kMaxLevels];
    int nlevels;
};
struct LevelScale {
    float scale[kMaxLevels];  // mvScaleFactor
    float patch[kMaxLevels];  // (float)(int)(31*scale), ORBextractor.cc:880,889
};
struct SelRec {  // one keypoint kept by the quadtree, level coordinates (border added back)
    uint16_t x, y, score;
    uint8_t level, pad;
    int32_t dst;  // output row (mono from the front / stereo from the back, ORBextractor.cc:1153-1162)
};

struct QtLevels {  // per-level geometry of the quadtree stage
    int W[kMaxLevels], H[kMaxLevels], quota[kMaxLevels], n_ini[kMaxLevels];
    int sel_off[kMaxLevels];  // first entry of the level inside one image's selection block
    int nlevels;
};

struct StereoRowJob;   // stereo_rowtable_device.h: the stereo row table of a frame, built beside the selection layout (nullptr: none)
struct FrameBlurJob;   // gauss7_stream_device.h: the blur of a frame's levels, carried by the selection launch of a frame (nullptr: none).
                       // *blur_carried (if given) tells whether it was: only the 1024-thread frame form carries it
int launch_quadtree(const QtLevels& lv, const Cand16* compact, const int* img_base, const int* level_count,
                     uint16_t* label, int* sel_pt, int* sel_n, int sel_stride, const LevelScale& scales, int lap0,
                     int lap1, int capacity, SelRec* sel, int* sel_count, int* mono, int n_images, hipStream_t s,
                     const StereoRowJob* row_job = nullptr, const FrameBlurJob* blur_job = nullptr, bool* blur_carried = nullptr,
                     char* global_ws = nullptr, uint32_t* global_label = nullptr);
                     // global_ws: n_images * nlevels * quadtree_global_workspace_stride(lv) bytes, global_label: one word per candidate slot (as
                     // `label`) -> the selection runs over a workspace in global memory (quotas beyond a workgroup's LDS, quadtree_global_kernels.hip)
size_t quadtree_global_workspace_stride(const QtLevels& lv);
void launch_quadtree_select_global(const QtLevels& lv, const Cand16* compact, const int* img_base, const int* level_count, uint32_t* label, int* sel_pt,
                                   int* sel_n, int sel_stride, int n_images, char* ws, hipStream_t s);
size_t quadtree_lds_bytes(const QtLevels& lv);
int launch_debug_sort(const uint32_t* h_keys, int n, int frame_form, uint32_t* h_nodes, uint32_t* h_keys_out, float* sort_us);   // msorb_debug_std_sort
void upload_patch_tables(const int8_t* pattern, const int* umax, hipStream_t stream);
hipError_t download_patch_tables(int8_t* pattern /* 1024 */, int8_t* umax /* 16 */, hipStream_t stream);   // back from the device's constant memory
// device <-> pinned-host copy by a kernel (per-frame calls; orb_kernels.hip blit16_kernel); 16-byte aligned pointers
void launch_blit(void* dst, const void* src, size_t bytes, hipStream_t s);
// pinned host <-> device on a stream: the copy kernel, or hipMemcpyAsync for unaligned pointers / MSORB_FRAME_COPIES=sdma
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
void launch_stage_level0(const LevelView& src, uint8_t* dst, int dst_pitch, size_t dst_image_stride, int n_images, hipStream_t s);
// The [OpenCV-recall] semantics that a real OpenCV build could turn out to differ in, as one table (msorb_semantics of the C
// ABI, oracle/cvprims.h Semantics): non-default entries route to the kernels that take them at run time.
struct Semantics {
    int gauss_taps[7] = {18, 34, 48, 56, 48, 34, 18};
    int resize_single_stage = 0;
    int atan2_fma = 0;
    int brief_tap = 0;   // rotated BRIEF tap: 0 fma(x, b, y*a) / fma(x, a, -(y*b)), 1 fma(y, a, x*b) / fma(-y, b, x*a), 2 no contraction (describe_kernel<kTap>)
    bool default_taps() const {
        static const int d[7] = {18, 34, 48, 56, 48, 34, 18};
        for (int i = 0; i < 7; i++) if (gauss_taps[i] != d[i]) return false;
        return true;
    }
};
// ComputePyramid of a frame or two as ONE launch ("tower"): the image is cut into ntx x nty tiles; a workgroup
// produces its tile of EVERY level, level l from its own copy of level l-1 in LDS — the part of level l-1 its tiles of the
// levels above need (`need`, a few pixels of halo more per level on the way down) — and stores the part it owns (`own`, the
// tiles partition each level).  No dependency between workgroups, so no launch boundary between levels: seven launches of
// ~5 us each (the levels of a frame are tiny: fixed cost) become one.
constexpr int kTowerTiles = 16;   // 16 x 8 tiles x the two images of a stereo frame = one workgroup per CU
struct TowerAxis {   // one axis of one level: tile k of n owns [tower_own(k, n, len), tower_own(k + 1, n, len)) and computes [need0[k], need1[k])
    int16_t need0[kTowerTiles], need1[kTowerTiles];
};
// x: boundaries at dword columns, the last one past the row's last dword; y: plain
__host__ __device__ inline int tower_own_x(int k, int n, int w) { return k >= n ? (w + 3) & ~3 : (int)((long long)k * w / n) & ~3; }
__host__ __device__ inline int tower_own_y(int k, int n, int h) { return k >= n ? h : (int)((long long)k * h / n); }
struct TowerPlan {
    TowerAxis x[kMaxLevels], y[kMaxLevels];   // x ranges are multiples of 4 (dword columns)
    int ntx = 0, nty = 0;                      // tiles used on each axis (0: no plan — per-level launches)
    int lds_even = 0, lds_odd = 0;             // bytes of the two ping-pong buffers (levels 0, 2, .. / 1, 3, ..)
    int lds_taps = 0;                          // bytes of the tile's taps of all levels
};
// taps_x / taps_y: HOST copies of the tap tables of level l (from level l - 1), as make_resize_taps returns them
bool build_tower_plan(TowerPlan& plan, int nlevels, const int* w, const int* h, const int* pitch, const std::vector<std::vector<ResizeTap>>& taps_x,
                      const std::vector<std::vector<ResizeTap>>& taps_y, size_t lds_limit);
// false: not applicable here (unaligned level 0, no plan, LDS) — the caller launches the levels one by one
bool launch_pyramid_tower(const PyramidView& pyr, const TowerPlan& plan, const ResizeTap* taps, const size_t* tap_x_off, const size_t* tap_y_off,
                          int n_images, hipStream_t s);
void launch_pyramid(const PyramidView& pyr, const ResizeTap* taps, const size_t* tap_x_off, const size_t* tap_y_off, int n_images,
                    hipStream_t s, const Semantics& sem = Semantics());
void launch_pyr_resize(const LevelView& src, const LevelView& dst, uint8_t* dst_base, const ResizeTap* tx,
                       const ResizeTap* ty, int n_images, hipStream_t s, int single_stage = 0);
void launch_fast_cells(const PyramidView& pyr, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                       int slots_per_image, Cand16* slots, int* cell_count, int n_images, bool small_cells, hipStream_t s);
void launch_cand_compact(const CellDesc* cells, int n_cells, const int* level_cell_begin, int nlevels,
                         int slots_per_image, const Cand16* slots, const int* cell_count, int* cell_off,
                         int* level_count, int* img_total, int* img_base, Cand16* compact, int n_images,
                         hipStream_t s, bool packed = true, bool frame_form = false);   // frame_form: <= 4 images, !packed: scan + gather as one launch
int launch_gauss7(const PyramidView& src, const PyramidView& dst, int n_images, hipStream_t s, const Semantics& sem = Semantics());
// FAST + blur of a frame (1-4 images) as one launch; false = not applicable (unaligned rows, non-default taps), nothing launched
bool launch_frame_fast_blur(const PyramidView& pyr, const PyramidView& blur, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                            int slots_per_image, Cand16* slots, int* cell_count, int n_images, bool small_cells, hipStream_t s,
                            const Semantics& sem = Semantics());
void launch_describe(const PyramidView& pyr, const PyramidView& blur, const SelRec* sel, const int* sel_count,
                     int sel_stride, const LevelScale& scales, msorb_keypoint* kps, uint8_t* desc, int out_stride,
                     int max_sel, int n_images, hipStream_t s, const Semantics& sem = Semantics());

}  // namespace msorb
