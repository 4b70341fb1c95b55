// Structures shared by matcher.hip (kernels) and matcher_host.hip (frame handle, replay logic, C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msorb.h"

namespace msorb {

constexpr int kGridCols = 64, kGridRows = 48;  // FRAME_GRID_COLS / FRAME_GRID_ROWS, Frame.h:44-45
constexpr int kThHigh = 100, kThLow = 50, kHistoLength = 30;  // ORBmatcher.cc:35-37
constexpr int kTopK = 8;  // candidates kept per query: with 4 a busy frame needed 4-5 device rounds (exhausted lists), with 8 fewer

struct KpLite {  // what the window search needs of one train keypoint (16 B, one load)
    float x, y, u_right;
    int octave;
};
struct FrameView {
    const KpLite* kp;
    const uint8_t* desc;
    const int* cell_begin;  // kGridCols*kGridRows + 1, cell = ix*kGridRows + iy (mGrid[ix][iy])
    const int* cell_idx;    // keypoint indices, insertion (ascending index) order inside a cell
    const uint8_t* occupied;  // F.mvpMapPoints[idx] && Observations() > 0
    float minX, minY, gridWInv, gridHInv;
    int n;
    float inv_sigma2[MSORB_MAX_LEVELS];  // mvInvLevelSigma2, read by the kQFuseGate queries only
    const KpLite* gate_kp;  // nullptr, or n keypoints the level band and the kQFuseGate error test read INSTEAD of kp (the window test
                            // stays on kp): ORBmatcher::Fuse(..., bRight = true) walks the right camera's grid but reads
                            // pKF->GetKeyPoint(idx) / GetuRight(idx) with that right-camera index (ORBmatcher.cc:1502, 1509-1545)
};
constexpr uint8_t kQValid = 1, kQSkipOccupied = 2, kQFuseGate = 4, kQNoUr = 8;
struct WinQuery {
    float x, y, r, ur;
    int16_t min_level, max_level;
    uint8_t flags, pad[3];
};
struct TopK {
    int idx[kTopK];
    int dist[kTopK];
};
struct StereoArgs {
    const msorb_keypoint *kpL, *kpR;
    const uint8_t *descL, *descR;
    int nL, nR, rows0;
    const uint8_t* pyrL[MSORB_MAX_LEVELS];
    const uint8_t* pyrR[MSORB_MAX_LEVELS];
    int pitchL[MSORB_MAX_LEVELS], pitchR[MSORB_MAX_LEVELS], rows[MSORB_MAX_LEVELS], cols[MSORB_MAX_LEVELS];
    float scale[MSORB_MAX_LEVELS], inv_scale[MSORB_MAX_LEVELS];
    float mb, mbf;
    float *u_right, *depth;
    int *sad, *n_oob;
    const int* row_begin;      // vRowIndices as CSR over the rows of level 0 (nullptr: band test per candidate)
    const int2* row_list;      // entry = {iR | octave << 24, bits of kpR[iR].x}: the candidate filters need no keypoint load
};

// A: the pointers of pair 0 (A.kpL / A.descL / A.pyrL = left eye, A.kpR / A.descR / A.pyrR = right eye).  Pair p lives
// pair_step images further in every array: pair_step = 2 when both eyes sit interleaved in ONE extract batch (images
// 2p / 2p+1: the right pointers are the left ones advanced by one image), 1 when the eyes come from two batches (two
// extractor handles, possibly filled on two devices and gathered).
struct StereoBatchArgs {
    StereoArgs A;
    size_t img_strideL[MSORB_MAX_LEVELS], img_strideR[MSORB_MAX_LEVELS];
    int capacity;
    int pair_step;
    const int *countsL, *countsR;  // device: n_keypoints of the left / right image of pair 0 (pair p at [p * pair_step])
    int* row_begin;             // per pair: [rows0 + 1]
    int2* row_list;             // per pair: [row_cap] entries {iR | octave << 24, bits of x}
    int row_cap;
    const int2* band;           // stereo frames (row_begin == nullptr): per right keypoint {minr | maxr << 12 | octave << 24, bits of x}, pair p at [p * capacity]
    const int* band_level_begin;   // with band: [MSORB_MAX_LEVELS + 1] first record of each octave (the records are in level-major order)
    int* counts_out;            // optional: the median kernel copies the counts of pair p to [2p], [2p+1] (fused per-frame calls)
    bool median_with_readback;  // one pair, no sink: launch_stereo_match_batch leaves the median rule to launch_stereo_median_readback
};

// n_frames > 1: a batch, frame b's arrays frame_stride keypoints / q_stride queries behind frame b-1's (grid: 3073 ints);
// n_eval != nullptr: the kernel adds the number of Hamming distances it evaluated (measurement runs)
void launch_window_topk(const FrameView& F, const WinQuery* q, const uint8_t* qdesc, int q_begin, int q_end,
                        TopK* out, hipStream_t s, int n_frames = 1, int frame_stride = 0, int q_stride = 0,
                        unsigned long long* n_eval = nullptr, int lanes = 16);
// lanes per query for a search whose windows have the radius r (pixels) on a frame with these grid constants
inline int window_lanes_for(float r, float gridWInv, float gridHInv) {
    const float cells = (2.0f * r * gridWInv + 1.5f) * (2.0f * r * gridHInv + 1.5f);
    return cells <= 14.0f ? 4 : 16;
}
// msorb_extract_stereo hands its DEVICE outputs to a sink after the last kernel of the frame has been enqueued and before
// the read-back + synchronisation: whatever the sink enqueues on `stream` completes with the same synchronisation.
struct StereoDeviceOutputs {
    const msorb_keypoint* kps_left;  // `capacity` entries
    const uint8_t* desc_left;
    const float* u_right;            // mvuRight of the left keypoints
    const int* n_left;               // device count (negative: capacity exceeded)
    int capacity;
    hipStream_t stream;
};
using StereoSinkFn = int (*)(void* ctx, const StereoDeviceOutputs& o);
int extract_stereo_sink(msorb_extractor* h, const uint8_t* left, const uint8_t* right, int rows, int cols, size_t stride_left,
                        size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left, int* n_left,
                        msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right, float* depth,
                        int* n_oob, StereoSinkFn sink, void* ctx);

void launch_window_list(const FrameView& F, const WinQuery* q, const uint8_t* qdesc, int n, int* count, const int* list_begin,
                        int2* list, bool fill, hipStream_t s);
void launch_list_top2(const uint8_t* qdesc, const uint8_t* tdesc, const int* cand_begin, const int* cand_idx, int nq,
                      int* bi, int* bd, int* si, int* sd, hipStream_t s);
void launch_stereo_match(const StereoArgs& a, hipStream_t s);
void launch_stereo_match_batch(const StereoBatchArgs& b, int n_pairs, int max_left, hipStream_t s, bool row_table_built = false);
// the median rule of pair 0 + the copy of the frame's output block [0, all_bytes) to pinned memory, head_bytes (a multiple of 16)
// of it untouched by the rule; both pointers 16-byte aligned
void launch_stereo_median_readback(const StereoBatchArgs& b, void* dst, const void* src, size_t head_bytes, size_t all_bytes, hipStream_t s);
void launch_dense_top2(const uint8_t* q, const uint8_t* t, const int* n_q, const int* n_t, int n_frames, int q_stride,
                       int t_stride, int max_q, int max_t, int* bi, int* bd, int* sd, hipStream_t s, int formulation = 0);

}  // namespace msorb
