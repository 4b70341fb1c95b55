// Host-side ORB parameter tables, level/cell geometry and the quadtree keypoint selection.
// Plain C++ (no HIP): compiled into libmsorb.so and unit-tested on CPU through the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace msorb {

constexpr int kMaxLevels = 16;
constexpr int kEdgeThreshold = 19;  // ORBextractor.cc:73
constexpr int kHalfPatch = 15;      // ORBextractor.cc:72
constexpr int kPatchSize = 31;      // ORBextractor.cc:71
constexpr int kMinBorder = kEdgeThreshold - 3;  // ORBextractor.cc:789

// round-half-even, like cvRound (SSE cvtss2si) — ORBextractor.cc:441,1175
int round_half_even(float v);
int round_half_even(double v);

struct OrbParams {  // ORBextractor::ORBextractor, ORBextractor.cc:409-469
    int nfeatures = 0, nlevels = 0, ini_th = 0, min_th = 0;
    float scale_factor_f = 0;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> per_level;
    int umax[kHalfPatch + 1];
    void init(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
};

// One horizontal or vertical bilinear tap of cv::resize(INTER_LINEAR) on 8-bit data: two source
// indices (already clamped) and two 11-bit weights.
struct ResizeTap {
    int16_t i0, i1, c0, c1;
};
std::vector<ResizeTap> make_resize_taps(int dst_len, int src_len, bool horizontal);

struct CellDesc {  // one FAST cell of ComputeKeyPointsOctTree (ORBextractor.cc:805-872)
    int16_t level;
    int16_t x0, y0;  // ROI origin in level pixels (iniX, iniY)
    int16_t rw, rh;  // ROI size (maxX-iniX, maxY-iniY)
    // detection rows per thread of the FAST quick test: a thread owns one 4-pixel column group of R consecutive rows;
    // R = ceil(dh / (threads / G)) for 128- and 256-thread workgroups
    int8_t R128, R256;
    int32_t slot_off;  // first candidate slot of this cell inside one image's slot block
    int32_t slot_cap;  // worst-case number of NMS survivors
    // FAST kernel constants of the cell, precomputed here so that no workgroup spends instructions on integer division:
    // G = 4-pixel groups per detection row, ndw = dwords per staged ROI row, *_magic = ceil(2^20 / divisor)
    // (q = (n * magic) >> 20 is exact for the kernel's ranges: n < 2^12)
    uint32_t g_magic, ndw_magic, rw_magic;
    int16_t G, ndw;
    // Rows per wave (quick test): with spw = 64 / G strips inside every wave, wave w gives its threads rw[w] rows starting at
    // detection row yw[w] (one byte per wave) — e.g. 39 rows over 2 x 6 strips as 4 + 3 per thread instead of 4 + 4.
    // by_wave[i] != 0 (i = 0: 128 threads, 1: 256): this split needs fewer row iterations than R128 / R256 and fits the kernel's
    // register window; otherwise the kernel keeps the uniform split above.
    uint8_t by_wave[2], spw, pad_;
    uint32_t rw128, yw128, rw256, yw256;
};

struct LevelGeom {
    int w = 0, h = 0;    // level size (ComputePyramid, ORBextractor.cc:1174-1175)
    int pitch = 0;       // row pitch of the device planes (multiple of 64 bytes)
    size_t plane_off = 0;  // byte offset of the plane inside one image's pyramid block
    size_t blur_off = 0;   // ... and of its blurred twin, stored in 4 x 4-pixel blocks (orb_device.h blur_tile_off): pitch x ceil8(h) bytes
    int n_cols = 0, n_rows = 0, w_cell = 0, h_cell = 0;
    int cell_begin = 0, cell_count = 0;  // slice of the per-image cell table
    int min_x = 0, max_x = 0, min_y = 0, max_y = 0;  // minBorderX.. maxBorderY
    int quota = 0;
};

struct FrameGeom {
    int rows = 0, cols = 0, nlevels = 0;
    LevelGeom lv[kMaxLevels];
    std::vector<CellDesc> cells;
    size_t pyramid_bytes = 0;  // per image, levels 1.. (level 0 may live in caller memory)
    size_t blur_bytes = 0;     // per image, the tiled blurred planes of all levels
    size_t plane0_bytes = 0;
    int slots_per_image = 0;
    // false if the reference's arithmetic would divide by zero for this size
    bool build(const OrbParams& p, int rows, int cols);
};

struct Cand16 {  // one FAST survivor as the device writes it: coordinates relative to (16,16)
    uint16_t x, y, score, pad;
};

// DistributeOctTree (ORBextractor.cc:555-779): picks <= ~N candidates, returns indices into `c`
// in the reference's result order (list order of the surviving nodes).
void distribute_quadtree(const Cand16* c, int n, int min_x, int max_x, int min_y, int max_y, int N,
                         std::vector<int>& kept);

}  // namespace msorb
