/* msorb — MI355X-native ORB front-end for MS-SLAM: C ABI of libmsorb.so.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * pointers and sizes, never throws, never allocates across the boundary, and returns MSORB_OK (0)
 * or a negative MSORB_E_* code (msorb_last_error() gives the thread-local detail string).
 * Each function names the reference interface it replaces (paths are into fishmarch/MS-SLAM).
 *
 * The kernels are hand-written HIP for gfx950; there is NO CPU fallback: on a machine without a
 * usable GPU every compute entry returns MSORB_E_NO_DEVICE.
 */
#ifndef MSORB_H
#define MSORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSORB_OK 0
#define MSORB_E_INVALID -1    /* bad argument */
#define MSORB_E_NO_DEVICE -2  /* no HIP device / HIP runtime error at init */
#define MSORB_E_HIP -3        /* HIP runtime error during the call: outputs and in/out arrays (frame_mp, cur_mp, matched, ...)
                                 are unspecified — the claim-replaying searches may have applied the accepts of earlier rounds */
#define MSORB_E_CAPACITY -4   /* caller-provided buffer too small */
#define MSORB_E_GEOMETRY -5   /* image too small for the reference's cell arithmetic (it would divide by zero) */
#define MSORB_E_EMPTY -6      /* empty input image: ORBextractor::operator() returns -1 (ORBextractor.cc:1090-1091) */

#define MSORB_MAX_LEVELS 16
#define MSORB_DESC_BYTES 32

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct msorb_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} msorb_keypoint;

const char* msorb_last_error(void);
int msorb_device_count(void);
/* Free / total memory of a device in bytes (hipMemGetInfo): what a long-running integration watches to see that the library's
 * resident state — extractor handles, matcher frames, the KeyFrame store — stays bounded (tests/soak_main.cc).  Since ABI 6000. */
int msorb_device_memory(int device, size_t* free_bytes, size_t* total_bytes);

/* ABI version of the library that was loaded: MSORB_ABI_VERSION of the header it was BUILT from.  major * 1000 + minor; a new
 * minor only appends entry points (or appends `_ex` forms with more parameters), a new major changes or removes one.  The host
 * classes (host/ORBextractor.cc, host/ORBmatcher_device.h) and the Python mirror compare msorb_abi_version() with the header
 * they were compiled against and refuse a library with another major or an older minor (msorb_abi_compatible). */
#define MSORB_ABI_VERSION 6001
int msorb_abi_version(void);
/* 1 if a caller compiled against `header_version` may use this library (same major, library minor >= header minor). */
int msorb_abi_compatible(int header_version);

/* Process-wide fatal-error callback of the host layer.  The reference's ORBextractor / ORBmatcher cannot fail and their callers
 * (Tracking.cc, LocalMapping.cc, LoopClosing.cc) have no handler around them, so the drop-in classes end the process on a lost
 * GPU the way the reference ends it on its own fatal conditions (message + exit(-1), System.cc:117-120).  An embedding
 * application registers a callback to get control FIRST (flush the map / atlas, log, raise its own flag): the host classes call
 * msorb_notify_fatal(code, what) before their default action (MSORB_THROW=1: throw std::runtime_error, else message + exit(-1));
 * the callback may itself not return (exit, longjmp, throw through C++ frames).  fn == NULL unregisters.  Thread safe; the
 * callback runs on the thread that hit the error.  The C entry points themselves never call it: they return MSORB_E_*. */
typedef void (*msorb_fatal_fn)(int code, const char* what, void* user);
void msorb_set_fatal_callback(msorb_fatal_fn fn, void* user);
void msorb_notify_fatal(int code, const char* what);

/* ------------------------------------------------------------------------------------------------
 * Extractor — replaces ORB_SLAM3::ORBextractor (include/ORBextractor.h:43-109, src/ORBextractor.cc)
 * ---------------------------------------------------------------------------------------------- */
typedef struct msorb_extractor msorb_extractor;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.cc:409-469).  `device` is the HIP device ordinal the handle lives on.  One handle per
 * extractor object; a handle owns its stream, device pyramid and pinned staging and must not be
 * entered concurrently (the reference never does: Frame.cc:122-125 uses one object per eye). */
int msorb_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                           int device, msorb_extractor** out);
void msorb_extractor_destroy(msorb_extractor* h);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (ORBextractor.h:61-81) and mnFeaturesPerLevel; each array has nlevels entries; NULL = skip. */
int msorb_extractor_tables(const msorb_extractor* h, float* scale, float* inv_scale, float* sigma2,
                           float* inv_sigma2, int* features_per_level);

/* Upper bound on keypoints one image can return: nfeatures + 19*nlevels — a level overshoots its quota by at most 3 (SURVEY.md §8a
 * a5), or returns the 4 * nIni children of the quadtree's unconditional first pass when its quota is smaller than that (nIni =
 * round(width / height) <= 4). */
int msorb_extractor_capacity(const msorb_extractor* h);

/* ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea) (ORBextractor.cc:1086-1168)
 * on one HOST image (8-bit, 1 channel, `stride` bytes per row).  Writes *n_keypoints keypoints
 * (reference order: keypoints outside [lap0,lap1] from the front, inside from the back) and
 * *n_keypoints x 32 descriptor bytes into caller buffers of `capacity` rows; *mono_index is the
 * operator()'s return value.  Returns MSORB_E_EMPTY for rows==0||cols==0||image==NULL. */
int msorb_extract(msorb_extractor* h, const uint8_t* image, int rows, int cols, size_t stride, int lap0, int lap1,
                  msorb_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_keypoints,
                  int* mono_index);

/* One stereo frame in one call — what Frame::Frame(imLeft, imRight, ...) does with two extractor threads and
 * ComputeStereoMatches (Frame.cc:119-137): both images go through the batch pipeline together, the stereo association
 * (Frame.cc:743-913, median rejection included) runs on the device-resident outputs, and keypoints, descriptors,
 * mvuRight and mvDepth come back with a single synchronisation.  Lapping areas are 0 (rectified stereo).  Results are
 * identical to two msorb_extract calls followed by msorb_stereo_matches.  capacity = entries available in each of the
 * caller's arrays (>= msorb_extractor_capacity). */
int msorb_extract_stereo(msorb_extractor* h, const uint8_t* left, const uint8_t* right, int rows, int cols,
                         size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left,
                         uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right,
                         int capacity, float* u_right, float* depth, int* n_oob);

/* The same stereo frame with one extractor object per DEVICE — BASELINE configs[3], "left/right images on 2 MI355X, gather of
 * keypoints/descriptors over xGMI".  Replaces the two extractor threads + join of Frame::Frame (Frame.cc:122-127) followed by
 * ComputeStereoMatches (Frame.cc:743-913) when mpORBextractorLeft and mpORBextractorRight (Tracking.cc:595-596) live on
 * different GPUs: `left` and `right` are two distinct handles (same parameters) created on devices A and B.  Each eye runs
 * its kernel chain on its own device; the right eye's keypoints, descriptors, count and pyramid are copied device to device
 * onto A (peer copy over xGMI when A != B), a HIP event orders the two streams, the stereo association runs on A and one
 * block comes back after one synchronisation.  A == B (two handles on one device) is allowed and takes the same path.
 * Results are identical to msorb_extract_stereo / to two msorb_extract calls + msorb_stereo_matches. */
int msorb_extract_stereo_split(msorb_extractor* left, msorb_extractor* right, const uint8_t* img_left,
                               const uint8_t* img_right, int rows, int cols, size_t stride_left, size_t stride_right,
                               float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left, int* n_left,
                               msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right,
                               float* depth, int* n_oob);

/* ComputePyramid (ORBextractor.cc:1170-1195) alone over n_images DEVICE-resident images (arguments as msorb_extract_batch):
 * fills the handle's pyramid, asynchronously on its stream, so that msorb_stereo_matches_split can read it.  Used on the
 * device that runs the stereo association when only the other eye's keypoints / descriptors were gathered (60 B per
 * keypoint) and its levels (1.5 MB per KITTI image) are rebuilt locally instead of being moved. */
int msorb_pyramid_batch(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols, size_t row_stride,
                        size_t image_stride);

/* mvImagePyramid[level] (ORBextractor.h:83) of the last msorb_extract() call as a host-visible plane
 * (interior pixels; the 19-px border of ORBextractor.cc:1185-1191 is not materialised).  The memory
 * is owned by the handle and valid until the next extract call.  Without msorb_extractor_set_host_pyramid the first
 * call after an extraction copies the whole pyramid synchronously (8 x hipMemcpy2D). */
int msorb_pyramid_level(msorb_extractor* h, int level, const uint8_t** data, int* rows, int* cols, size_t* stride);

/* enable != 0: every msorb_extract() call also brings levels 1.. of the pyramid to pinned host memory, with ONE asynchronous
 * copy on a stream of its own that starts when the pyramid kernels are done and overlaps FAST / quadtree / describe
 * (level 0 is the staged input image); msorb_pyramid_level then only hands out pointers.  For callers that keep reading
 * mvImagePyramid on the host (an unchanged Frame::ComputeStereoMatches, Frame.cc:840-855).  Default off. */
int msorb_extractor_set_host_pyramid(msorb_extractor* h, int enable);

/* TWO images of the same size through ONE kernel chain (the batch pipeline with two images) — what the two extractor threads of
 * Frame::Frame (Frame.cc:122-125) compute with two concurrent operator() calls, for callers that can hand both images over at
 * once WITHOUT the stereo match (msorb_extract_stereo is the call that also matches).  0.159 ms per pair against 2 x 0.13 for
 * two calls.  (A rendezvous of the two eye threads inside the drop-in class onto this call was measured and retired:
 * tools/experiments/README.md.)  Both images share the lapping area [lap0, lap1] (rectified stereo: 0, 0).  Results identical to two
 * msorb_extract calls on two handles.  With msorb_extractor_set_host_pyramid the levels 1.. of BOTH pyramids come back to pinned
 * memory as in msorb_extract (msorb_pyramid_level_image).  capacity = entries available in each of the caller's arrays.
 * staged: bit 0 / bit 1 = image_a / image_b is a pointer msorb_stage_image returned (from ANY handle on this device, with the
 * same geometry): the image already lies in pinned memory at the library's row pitch and is uploaded from there. */
int msorb_extract_pair(msorb_extractor* h, const uint8_t* image_a, const uint8_t* image_b, int rows, int cols, size_t stride_a,
                       size_t stride_b, int lap0, int lap1, msorb_keypoint* kps_a, uint8_t* desc_a, int* n_a, int* mono_a,
                       msorb_keypoint* kps_b, uint8_t* desc_b, int* n_b, int* mono_b, int capacity, int staged);
/* Copies a host image into the handle's pinned staging plane (what msorb_extract does first) and returns that plane: *pinned
 * (valid until the handle's next stage / extract call), *pitch its row pitch.  msorb_extract on the SAME handle recognises
 * the pointer and skips its own copy; msorb_extract_pair takes it with the `staged` bits. */
int msorb_stage_image(msorb_extractor* h, const uint8_t* image, int rows, int cols, size_t stride, const uint8_t** pinned, size_t* pitch);
/* msorb_pyramid_level for image 0 / 1 of the last msorb_extract_pair call (image 0 only after any other extract call). */
int msorb_pyramid_level_image(msorb_extractor* h, int image, int level, const uint8_t** data, int* rows, int* cols, size_t* stride);

/* The OpenCV primitives the extractor restates (resize, GaussianBlur, fastAtan2) are un-vendored dependencies of the reference
 * (CMakeLists.txt:35: OpenCV >= 4.4, no pinned version).  Their semantics follow SURVEY.md Appendix A; the three places where
 * a real OpenCV build could differ — and the one float expression of the reference itself whose rounding its COMPILER decides
 * (brief_tap) — are ONE runtime-selectable table, in the kernels (here) and in the oracle
 * (oracle/cvprims.h Semantics) alike — if a pin run (tools/pin_opencv.py) disagrees with a default, the fix is this call:
 *   gauss_taps       Q8 taps of GaussianBlur(7x7, sigma 2) (ORBextractor.cc:1133).  Default {18,34,48,56,48,34,18}: the
 *                    bit-exact fixed-point path of OpenCV >= 4.2; sum(taps) <= 257.  Other taps run the generic blur kernels.
 *   resize_rounding  vertical pass of resize(INTER_LINEAR, 8U) (ORBextractor.cc:1183).  0 (default): VResizeLinear<uchar>,
 *                    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2;  1: the generic FixedPtCast form (S0*b0 + S1*b1 + (1<<21)) >> 22
 *                    (generic resize kernel).
 *   atan2_fma        polynomial of fastAtan2 (ORBextractor.cc:102).  0 (default): separate multiply / add (x86-64 baseline
 *                    build); 1: contracted Horner steps (aarch64, -ffp-contract=fast builds).
 *   brief_tap        the rotated rBRIEF tap cvRound(x*b + y*a), cvRound(x*a - y*b) (ORBextractor.cc:117-119), built with -O3
 *                    -march=native (CMakeLists.txt:10-13), i.e. with whatever contraction that compiler and target choose.
 *                    0 (default): the FIRST product fused, fma(x, b, y*a) / fma(x, a, -(y*b)) — g++ and clang on an FMA target
 *                    (tools/probe_brief_tap.cc, compiled with the reference's flags, prints which one a given build has);
 *                    1: the SECOND product fused, fma(y, a, x*b) / fma(-y, b, x*a);  2: no contraction (a target without FMA,
 *                    -ffp-contract=off).  The conventions differ on about 3 of 10^7 (pattern point, angle) pairs
 *                    (tests/test_semantics_variants.py counts them).
 * sem == NULL restores the defaults.  Applies to every later call on the handle. */
typedef struct msorb_semantics {
    int gauss_taps[7];
    int resize_rounding;
    int atan2_fma;
    int brief_tap;   /* since ABI 6000 */
} msorb_semantics;
int msorb_extractor_set_semantics(msorb_extractor* h, const msorb_semantics* sem);

/* Batched operator() over n_images same-sized DEVICE-resident images (image i at d_images +
 * i*image_stride, rows of row_stride bytes).  Outputs stay on the device: image i's keypoints at
 * d_keypoints + i*capacity, descriptors at d_descriptors + i*capacity*32.  h_counts[i] / h_mono[i]
 * (host arrays, n_images entries, h_mono may be NULL) receive n_keypoints / mono_index.
 * Level 0 is read in place from d_images (it must stay valid until the call returns); rows that are not 4-byte
 * aligned work too — big batches of them (>= 128 images) are first copied into aligned planes owned by the handle,
 * which is faster than running the byte-granular kernel variants (1.88 vs 2.08 ms per 256 KITTI images; 1.72 ms with a
 * 64-byte row pitch). */
int msorb_extract_batch(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols,
                        size_t row_stride, size_t image_stride, int lap0, int lap1, msorb_keypoint* d_keypoints,
                        uint8_t* d_descriptors, int capacity, int* h_counts, int* h_mono);

/* msorb_extract_batch in two halves: _submit enqueues the whole chain on the handle's streams and returns, _wait blocks until
 * it has finished and hands out the counts (same arrays and error codes as msorb_extract_batch).  One batch per handle may be
 * pending; a caller that alternates two handles keeps the GPU busy across batch boundaries (bench.py does: the next batch's
 * pyramid / FAST run under the previous batch's quadtree / descriptor tail).  The output arrays and the input images must
 * stay untouched until _wait returns.  No reference counterpart: the reference processes one frame at a time. */
int msorb_extract_batch_submit(msorb_extractor* h, const uint8_t* d_images, int n_images, int rows, int cols,
                               size_t row_stride, size_t image_stride, int lap0, int lap1, msorb_keypoint* d_keypoints,
                               uint8_t* d_descriptors, int capacity);
int msorb_extract_batch_wait(msorb_extractor* h, int* h_counts, int* h_mono_index);

/* Stage timing of the last batch call, measured with HIP events on the handle's stream.
 * enable!=0 switches recording on.  Stage order: MSORB_STAGE_* below; ms[] has MSORB_N_STAGES slots. */
#define MSORB_STAGE_PYRAMID 0
#define MSORB_STAGE_FAST 1
#define MSORB_STAGE_COMPACT 2
#define MSORB_STAGE_BLUR 3
#define MSORB_STAGE_SELECT 4   /* device quadtree + output layout (MSORB_QUADTREE=host: D2H candidates + host quadtree + H2D, wall time) */
#define MSORB_STAGE_DESCRIBE 5 /* IC-angle + rBRIEF */
#define MSORB_N_STAGES 6
int msorb_extractor_set_profiling(msorb_extractor* h, int enable);
/* Execution shape of msorb_extract_batch: number of concurrently scheduled sub-batches (1..4, default 2; batches
 * of fewer than 16 images always use 1) and whether the blur runs on a second stream (default yes).  (1, 0) runs
 * every kernel alone on the GPU — the setting used for per-kernel roofline measurements. */
int msorb_extractor_set_overlap(msorb_extractor* h, int sub_batches, int blur_on_second_stream);
int msorb_extractor_stage_ms(const msorb_extractor* h, float* ms);

/* Test/inspection hooks (device -> host copies of intermediate products of the last call). */
int msorb_debug_level_size(const msorb_extractor* h, int level, int* rows, int* cols);
int msorb_debug_copy_level(msorb_extractor* h, int image, int level, int blurred, uint8_t* dst /* rows*cols */);
/* FAST candidates handed to the quadtree (vToDistributeKeys, ORBextractor.cc:795-869) of one image
 * and level, reference order; coordinates relative to (16,16); xyscore[3*i..3*i+2]. */
int msorb_debug_candidates(msorb_extractor* h, int image, int level, int* xyscore, int capacity, int* n);
/* The tables describe_kernel reads, copied back FROM THE DEVICE's constant memory: the 256 x 4 rBRIEF pattern (bit_pattern_31_,
 * ORBextractor.cc:149-406) and umax[16] (:453-468).  For tests that hold what the chip holds to constants recorded independently
 * of this library's sources (tests/golden/reference_constants.json).  Since ABI 6000. */
int msorb_debug_patch_tables(msorb_extractor* h, int8_t* pattern /* 1024 */, int8_t* umax /* 16 */);
/* The std::sort restatement of DistributeOctTree's careful loop (ORBextractor.cc:700: std::sort(vPrevSizeAndPointerToNode, compareNodes),
 * unstable — libstdc++'s introsort decides the order of equal keys, and with it which nodes are split before the quota is reached)
 * ALONE, as the selection kernels run it, on n <= 4000 explicit keys: frame_form != 0 the 1024-thread form of single frames
 * (ranges of <= 64 items sorted in the lanes of one wave; | 2: without that, round 5's form), 0 the 256-thread form of batches.
 * order[i] = input position of the item that ends at position i; sorted_keys and sort_us (the sort alone, timed on the device's
 * constant clock) may be NULL.  Since ABI 6000. */
int msorb_debug_std_sort(int device, const uint32_t* keys, int n, int frame_form, uint32_t* order, uint32_t* sorted_keys, float* sort_us);
/* Host-only: DistributeOctTree (ORBextractor.cc:555-779) on explicit candidates; writes the indices of
 * the kept candidates in result order.  Needs no GPU. */
int msorb_distribute_quadtree(const uint16_t* xs, const uint16_t* ys, const uint16_t* scores, int n, int min_x,
                              int max_x, int min_y, int max_y, int n_features, int* kept_idx, int capacity,
                              int* n_kept);

/* ------------------------------------------------------------------------------------------------
 * Matcher — the data-parallel core of ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:36-112,
 * src/ORBmatcher.cc) and Frame::ComputeStereoMatches (src/Frame.cc:743-913) on flat arrays.
 * The graph/mutex side of the reference (shared_ptr<MapPoint>, Observations(), pose projection with
 * Sophus/Eigen) stays in the caller; these entries start from projected coordinates and descriptor
 * arrays and return exactly the assignments the reference's loops would make (same candidate sets,
 * same scan order, same tie-breaks, same sequential side effects).
 * ---------------------------------------------------------------------------------------------- */
typedef struct msorb_frame msorb_frame;

/* A frame's features on the device plus its 64x48 feature grid.  One handle per thread of use. */
int msorb_frame_create(int device, msorb_frame** out);
void msorb_frame_destroy(msorb_frame* f);

/* Frame members the matcher reads: mvKeysUn, mDescriptors, mvuRight (NULL = all -1), image bounds
 * mnMinX..mnMaxY, mvScaleFactors.  Builds mGrid like Frame::AssignFeaturesToGrid (Frame.cc:385-416,
 * PosInGrid :657-667) — on the device, from the uploaded features (frame_grid_kernel).  Host arrays.  At most 32768 keypoints
 * per frame on gfx950 (what one workgroup's 160 KB of LDS sorts; MSORB_E_CAPACITY beyond, decided before the handle is
 * touched).  A call that fails later (allocation, HIP error) leaves the handle EMPTY: no keypoints, an empty grid. */
int msorb_frame_set(msorb_frame* f, const msorb_keypoint* keypoints, int n, const uint8_t* descriptors,
                    const float* u_right, float min_x, float max_x, float min_y, float max_y,
                    const float* scale_factors, int nlevels);

/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (Frame.cc:589-655): indices in the reference's
 * order (cells ix outer / iy inner, insertion order inside a cell).  Host-side walk of the same grid
 * the kernels use. */
int msorb_frame_features_in_area(const msorb_frame* f, float x, float y, float r, int min_level, int max_level,
                                 int* out_idx, int capacity, int* n);

/* Telemetry of the searches that replay sequential claims on the host (all SearchByProjection forms on a frame / KeyFrame
 * handle): returns the number of device rounds the LAST such search on `f` needed — 1 unless a query's candidate list (8
 * entries) was exhausted by earlier claims, in which case the kernel is re-run from that query on (at most one round per
 * query) —, and through the optional pointers the totals since msorb_frame_create. */
int msorb_frame_search_rounds(const msorb_frame* f, long long* total_rounds, long long* total_searches);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint>&, th, bFarPoints, thFarPoints)
 * (ORBmatcher.cc:43-142, rectified branch).  Map-point table of m entries visited in index order:
 *   track_in_view=mbTrackInView  bad=isBad()  sparsified=mbSparsified  proj_x/proj_y/proj_xr=mTrackProjX/Y/XR
 *   track_depth=mTrackDepth  level=mnTrackScaleLevel  view_cos=mTrackViewCos  mp_desc=GetDescriptor() (m x 32)
 *   obs=Observations().
 * frame_mp[n]: F.mvpMapPoints as table indices (-1 = none, every entry < m), updated in place.  *nmatches = return value. */
int msorb_search_by_projection_mps(msorb_frame* f, int m, const uint8_t* track_in_view, const uint8_t* bad,
                                   const uint8_t* sparsified, const float* proj_x, const float* proj_y,
                                   const float* proj_xr, const float* track_depth, const int* level,
                                   const float* view_cos, const uint8_t* mp_desc, const int* obs, int* frame_mp,
                                   float th, int far_points, float th_far_points, float nnratio, int* nmatches);

/* The same search on a TWO-CAMERA frame (F.Nleft != -1, the KannalaBrandt8 stereo rig): ORBmatcher.cc:43-213 with both arms —
 * per map point a left pass (:61-142; no mvuRight test for such a frame, :92) and a right pass (:144-210; radius not scaled by th,
 * no mbSparsified bypass), coupled through mvLeftToRightMatch / mvRightToLeftMatch (a match on one side also claims the stereo
 * partner on the other, :130-134 / :196-200) and through the `continue` of a failed left ratio test (:125-126: no right pass for
 * that point).  `left` / `right`: two msorb_frame handles on one device, set from F.mvKeys[0, Nleft) / F.mvKeysRight with their
 * descriptor rows and WITHOUT mvuRight (what Frame::GetFeaturesInArea(..., bRight) walks, Frame.cc:589-655).  The table has, beside
 * the entries of msorb_search_by_projection_mps, the right camera's scratch: track_in_view_r=mbTrackInViewR,
 * proj_xr / proj_yr=mTrackProjXR / mTrackProjYR, level_r=mnTrackScaleLevelR (-1: no right pass), view_cos_r=mTrackViewCosR.
 * left_to_right[n_left] / right_to_left[n_right] = F.mvLeftToRightMatch / F.mvRightToLeftMatch (-1 none).
 * frame_mp[n_left + n_right] = F.mvpMapPoints as table indices, updated in place.  Since ABI 6000. */
int msorb_search_by_projection_mps_rig(msorb_frame* left, msorb_frame* right, int m, const uint8_t* track_in_view,
                                       const uint8_t* track_in_view_r, const uint8_t* bad, const uint8_t* sparsified, const float* proj_x,
                                       const float* proj_y, const float* proj_xr, const float* proj_yr, const float* track_depth,
                                       const int* level, const int* level_r, const float* view_cos, const float* view_cos_r,
                                       const uint8_t* mp_desc, const int* obs, const int* left_to_right, const int* right_to_left,
                                       int* frame_mp, float th, int far_points, float th_far_points, float nnratio, int* nmatches);

/* The window search on its own, for the SearchByProjection variants that keep their accept rules in the caller
 * (KeyFrame / Sim3 / relocalisation forms, ORBmatcher.cc:423-753, 2154-2275): for each query the 4 nearest
 * descriptors among GetFeaturesInArea(x, y, r, min_level, max_level) in the reference's scan order (ties ->
 * earlier in the scan), skipping keypoints flagged in occupied[n] when skip_occupied[i] != 0 and keypoints whose
 * mvuRight differs from ur[i] by more than r[i] (ur == NULL disables nothing: pass a frame without mvuRight).
 * best_idx / best_dist have 4 entries per query (-1 / 256 = none). */
int msorb_window_top4(msorb_frame* f, int n_queries, const float* x, const float* y, const float* r, const float* ur,
                      const int* min_level, const int* max_level, const uint8_t* skip_occupied, const uint8_t* query_desc,
                      const uint8_t* occupied, int* best_idx, int* best_dist);

/* The search of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false) (ORBmatcher.cc:1404-1597; LocalMapping::
 * SearchInNeighbors, LocalMapping.cc:793-826): for every map point the best keypoint of the KeyFrame inside
 * GetFeaturesInArea(u, v, radius) (KeyFrame.cc:796-845) at level predicted-1 .. predicted (:1513-1514) that passes the
 * reprojection-error gate (:1517-1545: e2 * mvInvLevelSigma2[level] <= 7.8 with the stereo term when mvuRight >= 0,
 * <= 5.99 otherwise), first strict minimum in scan order (:1555-1559).  `f` = the KeyFrame loaded with msorb_frame_set: mvKeysUn,
 * mDescriptors, mvuRight, image bounds, scale factors.  Per point: valid (it survived :1436-1497), u, v (projection),
 * ur (u - bf*invz), predicted_level (PredictScale), radius (th * mvScaleFactors[level]), mp_desc.  best_idx / best_dist
 * (-1 / 256 = none).  What happens with a match (Replace / AddObservation, :1563-1588) mutates the map and stays with
 * the caller, which also re-checks isBad() / IsInKeyFrame() at that time like the reference's loop does. */
int msorb_fuse_search(msorb_frame* f, const float* inv_level_sigma2, int n_levels, int n, const uint8_t* valid,
                      const float* u, const float* v, const float* ur, const int* predicted_level, const float* radius,
                      const uint8_t* mp_desc, int* best_idx, int* best_dist);

/* The same search for ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = true) on a two-camera KeyFrame (NLeft != -1;
 * LocalMapping.cc:794-795, 825-826).  There the window is KeyFrame::GetFeaturesInArea(u, v, radius, true): the RIGHT camera's
 * grid and keypoints (KeyFrame.cc:826-836) — `f` = that camera loaded with msorb_frame_set — while the level band and the
 * reprojection-error gate read pKF->GetKeyPoint(idx) and pKF->GetuRight(idx) with the right-camera index idx as it comes out
 * of the grid (ORBmatcher.cc:1509-1545; `idx += NLeft` follows at :1547), i.e. mvKeys[idx] for idx < NLeft and
 * mvKeysRight[idx - NLeft] beyond (KeyFrame.h:377-385).  gate_kps[f->N] / gate_uright[f->N] are those values per right-camera
 * index (gate_uright NULL = -1 everywhere); everything else as msorb_fuse_search, best_idx in right-camera indices (the caller adds
 * NLeft).  With gate_kps = the frame's own keypoints and gate_uright = its mvuRight this is msorb_fuse_search.  Since ABI 6001. */
int msorb_fuse_search_gated(msorb_frame* f, const msorb_keypoint* gate_kps, const float* gate_uright, const float* inv_level_sigma2,
                            int n_levels, int n, const uint8_t* valid, const float* u, const float* v, const float* ur,
                            const int* predicted_level, const float* radius, const uint8_t* mp_desc, int* best_idx, int* best_dist);

/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono) (ORBmatcher.cc:1941-2057,
 * 2129-2152) from the projected coordinates on.  Per last-frame keypoint i: valid (map point present, not
 * outlier, positive depth, inside the image), u,v (projection), ur (u - mbf/z), last_octave, last_angle,
 * mp_desc (n_last x 32), last_mp (id stored into cur_mp), obs[id] = Observations() of map point id, n_obs entries: every
 * id in last_mp (of a valid entry) and in cur_mp must be < n_obs (MSORB_E_INVALID otherwise).  cur_mp[n] in/out. */
int msorb_search_by_projection_frames(msorb_frame* cur, int n_last, const uint8_t* valid, const float* u,
                                      const float* v, const float* ur, const int* last_octave,
                                      const float* last_angle, const uint8_t* mp_desc, const int* last_mp,
                                      const int* obs, int n_obs, int* cur_mp, float th, int forward, int backward,
                                      int check_orientation, int* nmatches);

/* The same search on a two-camera CurrentFrame (Nleft != -1; ORBmatcher.cc:1941-2152 with the right-camera arm :2059-2124), from
 * the projected coordinates on.  left / right: the current frame's two cameras as in msorb_search_by_projection_mps_rig.  Per
 * last-frame keypoint: valid (the tests of :1962-1983, on the LEFT projection), u, v (left camera), u_r, v_r
 * (mpCamera->project(GetRelativePoseTrl() * x3Dc), :2060-2061), last_octave / last_angle (the Nleft-aware keypoint of LastFrame).
 * cur_mp[n_left + n_right] in / out.  The right window of a keypoint is searched only when its left window held a candidate
 * (:2003-2004); one rotation histogram over both arms.  Since ABI 6000. */
int msorb_search_by_projection_frames_rig(msorb_frame* left, msorb_frame* right, int n_last, const uint8_t* valid, const float* u,
                                          const float* v, const float* u_r, const float* v_r, const int* last_octave,
                                          const float* last_angle, const uint8_t* mp_desc, const int* last_mp, const int* obs, int n_obs,
                                          int* cur_mp, float th, int forward, int backward, int check_orientation, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& Current, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:2154-2275,
 * Tracking::Relocalization :3672, :3695) from the projected coordinates on.  Per map point of the KeyFrame that the
 * loop reaches (not bad, not in sAlreadyFound, inside the image, distance inside the scale pyramid, :2173-2196):
 * valid, u, v, predicted_level (PredictScale, :2198), kf_angle (pKF->GetKeyUn(i).angle, :2237), mp_desc, mp_id (stored
 * into cur_mp).  A keypoint of the frame that holds ANY map point is never taken (:2214-2215); accept bestDist <=
 * orb_dist (:2229); cur_mp[n] in/out (-1 = none). */
int msorb_search_by_projection_kf(msorb_frame* cur, int n, const uint8_t* valid, const float* u, const float* v,
                                  const int* predicted_level, const float* kf_angle, const uint8_t* mp_desc,
                                  const int* mp_id, int* cur_mp, float th, int orb_dist, int check_orientation,
                                  int* nmatches);

/* The Sim3 / loop-closing window searches that claim keypoints: SearchByProjection(pKF, Scw, vpPoints, vpMatched, th,
 * ratioHamming) (ORBmatcher.cc:423-530) and the (pKF, Scw, vpPoints, vpPointsKFs, ...) form (:639-753), from the projected
 * coordinates on (SearchByProjectionLoop, :532-637, has different rules: msorb_search_by_projection_loop).  `kf` = the KeyFrame loaded with msorb_frame_set.  Per candidate point
 * that passed :446-480: valid, u, v, predicted_level, mp_desc, mp_id.  Keypoints with matched[idx] >= 0 are skipped
 * (:499-500), level band predicted-1 .. predicted (:505-507), first strict minimum, accepted when
 * (float)bestDist <= max_dist (= TH_LOW * ratioHamming, :521) and then claimed: matched[bestIdx] = mp_id (in/out).
 * The claim-free searches of Fuse(pKF, Scw, ...) (:1599-1716) and SearchBySim3 (:1718-1939) have entries of their own:
 * msorb_fuse_sim3_search and msorb_search_by_sim3. */
int msorb_search_by_projection_sim3(msorb_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                                    const int* predicted_level, const uint8_t* mp_desc, const int* mp_id, int* matched,
                                    float th, float max_dist, int* nmatches);

/* ORBmatcher::SearchByProjectionLoop(pKF, Scw, vpPoints, vpMatched, vpMatchedKF, th, ratioHamming) (ORBmatcher.cc:532-637;
 * LoopClosing::DetectCommonRegionsFromBoW :744,:753) from the projected coordinates on.  Per candidate point that passed
 * :555-588 (not bad, vpMatched[iMP] still empty, positive depth, inside the image, distance, viewing angle): valid, u, v,
 * predicted_level, mp_desc.  A keypoint is a candidate only when it holds a good map point (train_ok[idx] =
 * vpMapPointsToMatch[idx] && !isBad(), :609-610), level band predicted-1 .. predicted+1 (:613), first strict minimum,
 * accepted when bestDist <= max_dist (= TH_LOW * ratioHamming, :626).  Results are per POINT and independent of each other:
 * best_idx[i] = the keypoint whose map point becomes vpMatched[iMP], or -1; *nmatches = the return value. */
int msorb_search_by_projection_loop(msorb_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                                    const int* predicted_level, const uint8_t* mp_desc, const uint8_t* train_ok, float th,
                                    float max_dist, int* best_idx, int* nmatches);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (ORBmatcher.cc:1718-1939; LoopClosing) from the projected
 * coordinates on.  kf1 / kf2 = the two KeyFrames loaded with msorb_frame_set (n1 / n2 = their feature counts).  Per map
 * point i1 of pKF1 that pass 1 reaches (:1760-1794: present, not already matched, not bad, positive depth after S21 * T1w,
 * inside pKF2's image, distance inside the scale pyramid): valid1, u1, v1 (projection into pKF2), level1 =
 * PredictScale(dist3D, pKF2), desc1 = GetDescriptor(); symmetric arrays for pass 2 (:1850-1885).  Each pass searches
 * GetFeaturesInArea(u, v, th * mvScaleFactors[level]) of the other KeyFrame at levels level-1 .. level, first strict
 * minimum, accepted when bestDist <= TH_HIGH (:1843-1846, :1915-1918); match12[i1] = idx2 when both passes agree
 * (:1922-1937), else -1; *nfound = the return value.  vpMatches12[i1] = vpMapPoints2[match12[i1]] stays with the caller. */
int msorb_search_by_sim3(msorb_frame* kf1, msorb_frame* kf2, int n1, const uint8_t* valid1, const float* u1, const float* v1,
                         const int* level1, const uint8_t* desc1, int n2, const uint8_t* valid2, const float* u2,
                         const float* v2, const int* level2, const uint8_t* desc2, float th, int* match12, int* nfound);

/* The search of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:1599-1716; LoopClosing::
 * SearchAndFuse): for every candidate point that passed :1622-1656 (valid, u, v, predicted_level, mp_desc) the best keypoint
 * of the KeyFrame inside GetFeaturesInArea(u, v, th * mvScaleFactors[level]) at levels level-1 .. level, first strict
 * minimum, NO reprojection-error gate (unlike msorb_fuse_search).  best_idx -1 / best_dist INT_MAX = none.  The accept rule
 * bestDist <= TH_LOW and what follows (vpReplacePoint / AddObservation / AddMapPoint, :1698-1713) stay with the caller:
 * they read and mutate the map sequentially. */
int msorb_fuse_sim3_search(msorb_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                           const int* predicted_level, const uint8_t* mp_desc, float th, int* best_idx, int* best_dist);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:755-870; monocular
 * initialisation).  f1 / f2 = the two frames loaded with msorb_frame_set (both on one device).  prev_xy = vbPrevMatched
 * (x, y per F1 keypoint; updated in place like :864-867), matches12[N1] = vnMatches12, *nmatches = the return value.  The
 * device evaluates every (level-0 keypoint of F1, candidate of F2 in the window) Hamming distance in the reference's scan
 * order; the sequential rule "a train matched at a smaller or equal distance is skipped" (:791-792), the ratio test, the
 * re-assignment of trains and the rotation histogram are replayed on the host over those lists. */
int msorb_search_for_initialization(msorb_frame* f1, msorb_frame* f2, float* prev_xy, int window_size, float nnratio,
                                    int check_orientation, int* matches12, int* nmatches);

/* Best / second-best Hamming match of each query over an explicit candidate list (CSR: candidates of
 * query i are cand_idx[cand_begin[i] .. cand_begin[i+1])), scanned in list order with strict '<' — the
 * inner loop of SearchByBoW / SearchForTriangulation / Fuse (e.g. ORBmatcher.cc:288-330).  Host arrays. */
int msorb_hamming_top2(int device, const uint8_t* query_desc, int n_queries, const uint8_t* train_desc, int n_train,
                       const int* cand_begin, const int* cand_idx, int* best_idx, int* best_dist, int* second_idx,
                       int* second_dist);

/* Dense brute-force top-2 Hamming match, batched and device resident (the "brute-force Hamming match" of BASELINE.json's
 * metric): for each of n_frames frames, every query row against every train row of the same frame with
 * ORBmatcher::DescriptorDistance (ORBmatcher.cc:2323-2339); candidates scanned in index order with strict '<' (ties -> lowest
 * index).  d_* are DEVICE pointers: descriptors [n_frames][stride][32], counts [n_frames], outputs [n_frames][query_stride]
 * (rows past a frame's query count are not written).  Two formulations with identical results:
 *   MSORB_DENSE_POPCOUNT      v_xor + v_bcnt per dword: BASELINE north_star's formulation ("per-wavefront popcount ... no MFMA"),
 *                             the DEFAULT of every entry that does not take the argument (since ABI 6000)
 *   MSORB_DENSE_MATRIX_CORES  an opt-in variant: distances as int8 dot products (v_mfma_i32_32x32x32_i8; the accumulator is the
 *                             (distance, index) key) — 2.9x the rate of the popcount form, outside the north_star's design rule
 * The launch is repeated `repeats` times on a private stream between two HIP events; *elapsed_ms (may be NULL)
 * receives the total.  max_train <= 2048. */
#define MSORB_DENSE_MATRIX_CORES 0
#define MSORB_DENSE_POPCOUNT 1
int msorb_hamming_dense_top2_batch_ex(int device, const uint8_t* d_query, const uint8_t* d_train, const int* d_n_query,
                                      const int* d_n_train, int n_frames, int query_stride, int train_stride, int max_query,
                                      int max_train, int* d_best_idx, int* d_best_dist, int* d_second_dist, int repeats,
                                      float* elapsed_ms, int formulation);
/* The entry without the formulation argument: MSORB_DENSE_POPCOUNT since ABI 6000 (rounds 1-5 ran the matrix-core variant
 * here; results are identical, only the rate differs). */
int msorb_hamming_dense_top2_batch(int device, const uint8_t* d_query, const uint8_t* d_train, const int* d_n_query,
                                   const int* d_n_train, int n_frames, int query_stride, int train_stride, int max_query,
                                   int max_train, int* d_best_idx, int* d_best_dist, int* d_second_dist, int repeats,
                                   float* elapsed_ms);

/* cv::BFMatcher(cv::NORM_HAMMING).knnMatch(query, train, matches, 2) — the brute-force step of
 * Frame::ComputeStereoFishEyeMatches (Frame.cc:1057-1076: left vs right descriptors of the lapping area) — on HOST arrays of
 * 32-byte rows: best_idx / best_dist = matches[i][0] (trainIdx, distance), second_dist (and second_idx, may be NULL) =
 * matches[i][1]; ties go to the lower train index.  -1 / 256 where the train set has fewer than one / two rows.  The Lowe
 * ratio test and KannalaBrandt8::TriangulateMatches of :1082-1098 stay with the caller (host mirror:
 * msorb_host::ComputeStereoFishEyeMatches).  Runs the popcount formulation of the dense kernel. */
int msorb_knn_match2(int device, const uint8_t* query, int n_query, const uint8_t* train, int n_train, int* best_idx,
                     int* best_dist, int* second_idx, int* second_dist);

/* Frame::ComputeStereoMatches (Frame.cc:743-913).  left/right are the two extractor handles whose last
 * msorb_extract() call produced the images' pyramids (mpORBextractorLeft/Right->mvImagePyramid stay on
 * the device).  Keypoint/descriptor arrays are host arrays as returned by msorb_extract.  Writes
 * mvuRight / mvDepth (n_left entries, -1 = none).  *n_oob counts keypoints whose SAD window would leave
 * the pyramid plane (the reference would hit a CV_Assert there; they are skipped).  The two handles may live on
 * different devices (one extractor object per GPU): the right pyramid is then copied to the left handle's device peer to
 * peer before the kernel runs there. */
int msorb_stereo_matches(msorb_extractor* left, msorb_extractor* right, const msorb_keypoint* kps_left, int n_left,
                         const uint8_t* desc_left, const msorb_keypoint* kps_right, int n_right,
                         const uint8_t* desc_right, float mb, float mbf, float* u_right, float* depth, int* n_oob);

/* ORBmatcher::SearchByBoW — the three forms: (pKF, F, vpMapPointMatches) ORBmatcher.cc:223-421 (pinhole branch,
 * F.Nleft == -1; TrackReferenceKeyFrame Tracking.cc:2727, Relocalization :3585) and the two KeyFrame-KeyFrame forms
 * :872-1016, :1018-1166 (LoopClosing).  One msorb_bow_pair per call of the reference; a batch of pairs (all
 * relocalisation / loop candidates of one frame) is matched by ONE launch, one wavefront per (pair, common node).
 * Set 1 = the queries (pKF / pKF1), set 2 = the trains (F / pKF2):
 *   desc1/desc2     n x 32 descriptor rows
 *   valid1[n1]      1 = the query is visited (map point present and not bad, descriptor not empty, :253-263 /
 *                   :910-920 / :1066-1078)
 *   avail2[n2]      1 = the train may be chosen (:934-944 / :1090-1103); NULL = all (the Frame form, :227)
 *   fvK_*           DBoW2::FeatureVector as CSR: node ids ascending (std::map order), features of node r are
 *                   fvK_feat[fvK_begin[r] .. fvK_begin[r+1]); a feature index appears at most once per vector
 *   angle1/angle2   keypoint angles for the rotation histogram (may be NULL when check_orientation == 0)
 * Outputs: match12[n1] = index of the matched train or -1, match21[n2] (may be NULL) the inverse, nmatches — all
 * after the histogram filter (:396-418).  th_low / inclusive: bestDist1 <= TH_LOW (:332) or bestDist1 < TH_LOW
 * (:959, :1118).  Host arrays; *elapsed_ms (may be NULL) = device time of the launch. */
typedef struct msorb_bow_pair {
    int n1, n2;
    const uint8_t *desc1, *desc2;
    const uint8_t *valid1, *avail2;
    int fv1_nodes;
    const int *fv1_node, *fv1_begin, *fv1_feat;
    int fv2_nodes;
    const int *fv2_node, *fv2_begin, *fv2_feat;
    const float *angle1, *angle2;
    int *match12, *match21;
    int nmatches;
} msorb_bow_pair;
int msorb_search_by_bow(int device, msorb_bow_pair* pairs, int n_pairs, int th_low, int inclusive, float nnratio,
                        int check_orientation, float* elapsed_ms);
/* ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) on a two-camera frame (F.Nleft != -1; ORBmatcher.cc:223-421 with the arms of
 * :276-309 / :357-382): inside a BoW node every KeyFrame feature keeps a best / second over the frame's LEFT features (rows
 * < n_left) and a best over its RIGHT features, each among the features no earlier KeyFrame feature has claimed; the left match as on
 * a one-camera frame (<= th_low, ratio test), the right match at <= th_low WITHOUT a ratio test and only when the left best distance
 * was <= th_low (:330 encloses :357).  pair: set 1 = the KeyFrame (valid1 = map point present and good), set 2 = the frame's
 * n2 = N features, left camera first, angle2 = [mvKeys angles | mvKeysRight angles]; avail2 is not read.  match21[n2] (required):
 * the KeyFrame feature whose map point frame feature j receives, -1 none; match12 (optional): the LEFT partner of each KeyFrame
 * feature; nmatches counts both cameras.  One rotation histogram over both (:338-353, :361-378, :396-418).  Since ABI 6000. */
int msorb_search_by_bow_rig(int device, msorb_bow_pair* pair, int n_left, int th_low, float nnratio, int check_orientation);

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1168-1402) with the geometric test of :1332 —
 * pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, sigma1, sigma2) — left to the CALLER: the form for the two-camera
 * KeyFrames of a fisheye rig (:1294-1330 pick one of four relative poses and two camera models per candidate pair;
 * KannalaBrandt8::epipolarConstrain triangulates, KannalaBrandt8.cpp:216-220) and for any camera model this library does not
 * restate.  pair: as msorb_search_by_bow — set 1 = pKF1's N features (valid1 = no map point, stereo when bOnlyStereo, :1237-1247),
 * set 2 = pKF2's (avail2 = the same for pKF2, :1259-1271; NULL = all), the two FeatureVectors, angle1 / angle2 = GetKeyPoint(idx).angle.
 * Per common node and visited query, in the reference's order, the library finds the trains within th_low (TH_LOW) on the device
 * and calls accept(ctx, idx1, idx2) on them — smallest distance first, among equal distances the LATER train first (the scan's
 * `dist > bestDist` rule, :1277, lets a later equal candidate replace an earlier one) — skipping trains an earlier query took,
 * until one passes: that train is the query's match.  accept must be a pure predicate of (idx1, idx2) (the reference's is); it is
 * called on the calling thread, between the device passes and the return.  Not applied: the epipole-distance test of :1283-1291
 * (`!pKF1->mpCamera2` guards it: a caller that wants it puts it into accept).  match12 / match21 / nmatches after the rotation
 * histogram (:1360-1381).  Since ABI 6001. */
typedef int (*msorb_pair_accept)(void* ctx, int idx1, int idx2);
int msorb_search_for_triangulation_cb(int device, msorb_bow_pair* pair, int th_low, int check_orientation, msorb_pair_accept accept,
                                      void* ctx);

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1168-1402; LocalMapping::CreateNewMapPoints, LocalMapping.cc:492)
 * for pinhole KeyFrames without a second camera.  One msorb_triangulation_pair per (mpCurrentKeyFrame, neighbour)
 * call; all neighbours of one CreateNewMapPoints pass are matched by ONE launch.
 *   valid1[n1]   1 = the query is visited: no map point (:1237-1241), stereo when bOnlyStereo (:1245-1247), descriptor
 *                not empty;   avail2[n2]  1 = the train is eligible: no map point, stereo when bOnlyStereo (:1264-1271)
 *   stereo1/2    GetuRight(idx) >= 0 (:1243, :1267)
 *   kp1 / kp2    GetKeyPoint(idx) for every feature (cv::KeyPoint layout: pt, angle and octave are read)
 *   scale_factors2 / level_sigma2_2 [n_levels2]   pKF2->mvScaleFactors / mvLevelSigma2 (:1287, :1332)
 *   F12          K1^-T [t12]x R12 K2^-1, row major, exactly the Eigen expression of Pinhole::epipolarConstrain
 *                (Pinhole.cpp:109-112) — constant per pair, so the caller evaluates it once instead of per candidate
 *   ep           pKF2->mpCamera->project(T2w * pKF1->GetCameraCenter()) (:1177-1181)
 * coarse = bCoarse (:1332 skips the epipolar test).  match12[n1] = vMatches12 after the orientation filter
 * (:1360-1381); nmatches = the return value; vMatchedPairs = the (i, match12[i]) with match12[i] >= 0 in index order. */
typedef struct msorb_triangulation_pair {
    int n1, n2;
    const uint8_t *desc1, *desc2;
    const uint8_t *valid1, *avail2, *stereo1, *stereo2;
    int fv1_nodes;
    const int *fv1_node, *fv1_begin, *fv1_feat;
    int fv2_nodes;
    const int *fv2_node, *fv2_begin, *fv2_feat;
    const msorb_keypoint *kp1, *kp2;
    const float *scale_factors2, *level_sigma2_2;
    int n_levels2;
    float F12[9];
    float ep[2];
    int* match12;
    int nmatches;
} msorb_triangulation_pair;
int msorb_search_for_triangulation(int device, msorb_triangulation_pair* pairs, int n_pairs, int coarse,
                                   int check_orientation, float* elapsed_ms);

/* Resident KeyFrames for the BoW-node searches.  A KeyFrame's descriptors, keypoints and FeatureVector are fixed from
 * KeyFrame::ComputeBoW on (until map sparsification compacts them: remove + add), while it is matched many times — by
 * LocalMapping::CreateNewMapPoints against every neighbour (LocalMapping.cc:430-492), as a relocalisation / loop candidate
 * (Tracking.cc:3577-3600, LoopClosing.cc).  msorb_search_by_bow / msorb_search_for_triangulation stage both sides of every
 * pair on every call; a store keeps them on the device, and a search moves only one flag byte per feature, the (pair,
 * common node) work items and, for the KeyFrame-vs-Frame form, the frame.  Same kernels, same results.  Thread-safe:
 * searches may run concurrently with each other; add / remove wait for running searches. */
typedef struct msorb_kf_store msorb_kf_store;
int msorb_kf_store_create(int device, msorb_kf_store** out);
void msorb_kf_store_destroy(msorb_kf_store* s);
int msorb_kf_store_count(const msorb_kf_store* s);
/* kps = GetAllKeyUn() (pt, angle, octave are used), desc = GetDescriptor(i) rows, fv_* = GetFeatureVector() as CSR (node ids
 * ascending), scale_factors / level_sigma2 = mvScaleFactors / mvLevelSigma2.  *kf_id identifies the KeyFrame in later calls. */
int msorb_kf_store_add(msorb_kf_store* s, int n, const msorb_keypoint* kps, const uint8_t* desc, int fv_nodes, const int* fv_node,
                       const int* fv_begin, const int* fv_feat, const float* scale_factors, const float* level_sigma2, int n_levels,
                       int* kf_id);
/* The KeyFrame's rows and its id go back to the store: the next add takes them (first fit over the freed ranges), so that the
 * store's footprint follows the LIVE KeyFrames of a sequence, not its length (tests/soak_main.cc).  kf_id is invalid afterwards. */
int msorb_kf_store_remove(msorb_kf_store* s, int kf_id);
/* Feature rows in use / reserved on the device (one row = 92 bytes over four arrays).  Since ABI 6000. */
int msorb_kf_store_rows(const msorb_kf_store* s, size_t* rows_in_use, size_t* rows_reserved);

/* msorb_search_by_bow with resident KeyFrames: kf1 = the query KeyFrame, kf2 = the train KeyFrame (KeyFrame-KeyFrame forms,
 * ORBmatcher.cc:872-1166) or -1 = the frame passed to the call (SearchByBoW(pKF, F, ...), :223-421; then every pair of the
 * call has kf2 == -1).  valid1 / avail2 / match12 / match21 / nmatches as in msorb_bow_pair. */
typedef struct msorb_bow_kf_pair {
    int kf1, kf2;
    const uint8_t *valid1, *avail2;
    int *match12, *match21;
    int nmatches;
} msorb_bow_kf_pair;
typedef struct msorb_bow_frame {  /* F.mDescriptors, F.mFeatVec as CSR, F.mvKeysUn[i].angle */
    int n;
    const uint8_t* desc;
    int fv_nodes;
    const int *fv_node, *fv_begin, *fv_feat;
    const float* angle;
} msorb_bow_frame;
int msorb_search_by_bow_kf(msorb_kf_store* s, msorb_bow_kf_pair* pairs, int n_pairs, const msorb_bow_frame* frame, int th_low,
                           int inclusive, float nnratio, int check_orientation, float* elapsed_ms);

/* msorb_search_for_triangulation with resident KeyFrames (fields as in msorb_triangulation_pair). */
typedef struct msorb_triangulation_kf_pair {
    int kf1, kf2;
    const uint8_t *valid1, *avail2, *stereo1, *stereo2;
    float F12[9];
    float ep[2];
    int* match12;
    int nmatches;
} msorb_triangulation_kf_pair;
int msorb_search_for_triangulation_kf(msorb_kf_store* s, msorb_triangulation_kf_pair* pairs, int n_pairs, int coarse,
                                      int check_orientation, float* elapsed_ms);

/* Frame::ComputeStereoMatches (Frame.cc:743-913, median rejection :899-912 included) for every stereo pair of the last
 * msorb_extract_batch() call of `h`, all on the device: pair p = images 2p (left) and 2p+1 (right) of that batch.
 * d_keypoints / d_descriptors / capacity are the arrays that call filled, d_counts[2*n_pairs] the keypoint counts as a
 * DEVICE array.  Outputs (device): d_u_right / d_depth [n_pairs][capacity] (mvuRight / mvDepth of the left image, -1 =
 * none; entries past the left count are not written), d_n_oob[n_pairs] (may be NULL; see msorb_stereo_matches).
 * The pyramids of the batch must still be alive (no other extract call on `h` in between).  *elapsed_ms (may be NULL) =
 * device time of the two kernels. */
int msorb_stereo_matches_batch(msorb_extractor* h, int n_pairs, const msorb_keypoint* d_keypoints,
                               const uint8_t* d_descriptors, int capacity, const int* d_counts, int max_left, float mb,
                               float mbf, float* d_u_right, float* d_depth, int* d_n_oob, float* elapsed_ms);

/* The same with the two eyes in separate batches: left images = the last msorb_extract_batch() of `left`, right images = the
 * last msorb_extract_batch() or msorb_pyramid_batch() of `right` (both handles on ONE device; pair p = image p of each).
 * The right keypoints / descriptors / counts may have been extracted on another device and gathered (RCCL / peer copy). */
int msorb_stereo_matches_split(msorb_extractor* left, msorb_extractor* right, int n_pairs, const msorb_keypoint* d_kps_left,
                               const uint8_t* d_desc_left, const int* d_counts_left, const msorb_keypoint* d_kps_right,
                               const uint8_t* d_desc_right, const int* d_counts_right, int capacity, int max_left, float mb,
                               float mbf, float* d_u_right, float* d_depth, int* d_n_oob, float* elapsed_ms);

/* ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:2277-2318) on histogram bin sizes; ind[3]. Host only. */
int msorb_three_maxima(const int* bin_sizes, int n_bins, int* ind);

/* ------------------------------------------------------------------------------------------------
 * Map sparsification — constraint-matrix assembly of MapSparsification::Sparsifying
 * (src/MapSparsification.cc:58-151) as CSR, built on the device.  The caller flattens, under the
 * reference's locks, what the loop reads:
 *   window keyframe k (vpKFs order) owns slots [kf_slot_begin[k], kf_slot_begin[k+1]) in the grid walk
 *   order of :82-84 (grid column, grid row, cell index list): slot_point = map point id or -1 (null or
 *   isBad()), slot_cell = column*rows+row;  point_nobs = Observations(); obs_begin/obs_kf = GetObservations()
 *   keyframe ids per point; kf_in_window = (mnMapSaprsificationId == mnId); kf_num_mps = GetNumberMPs().
 * Outputs (host arrays): columns = map points in first-encounter order (col_point, obj_coef = nMaxObs -
 * Observations()); rows in the order the reference adds constraints: per window keyframe its valid cells
 * (row_kind 0, rhs 1) then the keyframe row (kind 1, rhs N), then one row per outside keyframe observing a
 * column point (kind 2, rhs count/GetNumberMPs()*N) in ascending keyframe id (the reference walks a
 * std::map keyed by shared_ptr there, i.e. run-dependent order).  row_begin has *n_rows+1 entries; col_idx
 * lists column indices in the order the reference accumulates the terms.  Every row owns one implicit slack
 * variable (th_grid: binary, cost GridLambda; th: integer 0..1000, cost Lambda) — see INTEGRATION.md for the
 * GUROBI C-API mapping.  n_max_obs_floor: max Observations() over map points of window keyframes that are in
 * no grid cell (normally 0).
 * ---------------------------------------------------------------------------------------------- */
int msorb_visibility_csr(int device, int n_window_kf, const int* kf_slot_begin, const int* slot_point,
                         const int* slot_cell, int n_points, const int* point_nobs, const int* obs_begin,
                         const int* obs_kf, int n_kf_total, const uint8_t* kf_in_window, const int* kf_num_mps, int n,
                         int n_max_obs_floor, int* n_cols, int* col_point, int cap_cols, int* n_rows, int* row_begin,
                         int* row_kind, int* row_owner, float* row_rhs, int cap_rows, int* col_idx, int cap_nnz,
                         int* nnz, float* obj_coef, int* n_max_obs);

/* ------------------------------------------------------------------------------------------------
 * Bag of words — DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&,
 * FeatureVector&, levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1123-1191, descent :1218-1259) as
 * called by Frame::ComputeBoW (src/Frame.cc:670-677) and KeyFrame::ComputeBoW with levelsup = 4.
 * The vocabulary tree lives on the device (children of a node contiguous); the descent is one 16/32-lane
 * group per descriptor, the two std::map containers are assembled per frame by a sort + run-length pass
 * that repeats the reference's accumulation order (sequential double adds), so values are bit-identical.
 * ---------------------------------------------------------------------------------------------- */
typedef struct msorb_vocabulary msorb_vocabulary;

/* Tree as loadFromTextFile (TemplatedVocabulary.h:1338-1423) builds it: n_nodes nodes, node 0 = root;
 * for i >= 1: parent[i] (< i is not required), is_leaf[i] = the file's leaf flag (word ids are handed out to
 * flagged nodes in node order), descriptors[i*32..], weights[i].  scoring / weighting = the file header's
 * ScoringType / WeightingType (BowVector.h:39-56); ORBvoc.txt is k=10, L=6, L1_NORM(0), TF_IDF(0). */
int msorb_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int* parent,
                            const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights,
                            msorb_vocabulary** out);
/* Reads the ORBvoc.txt text format (header "k L scoring weighting", then one line per node:
 * "parent is_leaf b0 .. b31 weight").  Empty lines are skipped (the reference's `while(!f.eof())` loop turns a
 * trailing newline into a phantom child of the root with an indeterminate descriptor; not reproduced). */
int msorb_vocabulary_load_text(int device, const char* path, msorb_vocabulary** out);
void msorb_vocabulary_destroy(msorb_vocabulary* v);
int msorb_vocabulary_info(const msorb_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words);

/* transform() for n_frames frames whose descriptors are DEVICE resident (frame i: d_descriptors +
 * i*desc_stride*32, h_counts[i] rows — exactly msorb_extract_batch's outputs).  Device outputs, `stride` entries
 * per frame (stride >= max count, <= 8192): BowVector as ascending (d_bow_word, d_bow_value) with d_n_bow[i]
 * entries; FeatureVector as ascending node ids d_fv_node with CSR d_fv_begin (stride+1 per frame) into d_fv_feat
 * (feature indices, ascending inside a node) and d_n_fv[i] nodes.  elapsed_ms (may be NULL): kernel time. */
int msorb_bow_transform_batch(msorb_vocabulary* v, const uint8_t* d_descriptors, const int* h_counts, int n_frames,
                              int desc_stride, int levelsup, int stride, int* d_bow_word, double* d_bow_value,
                              int* d_n_bow, int* d_fv_node, int* d_fv_begin, int* d_fv_feat, int* d_n_fv,
                              float* elapsed_ms);
/* One frame, HOST arrays in and out (capacity n entries each, fv_begin n+1).  feat_word / feat_node /
 * feat_weight (may be NULL): per-feature result of the descent (:1218-1259). */
int msorb_bow_transform(msorb_vocabulary* v, const uint8_t* descriptors, int n, int levelsup, int* bow_word,
                        double* bow_value, int* n_bow, int* fv_node, int* fv_begin, int* fv_feat, int* n_fv,
                        int* feat_word, int* feat_node, double* feat_weight);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:351-429) batched over map points: point p owns the
 * descriptors [obs_begin[p], obs_begin[p+1]) of `descriptors` (32 B rows, in the order the reference collects
 * vDescriptors).  best_idx[p] = position inside the point's list of the descriptor with the least median
 * distance to the others (first minimum; -1 for a point without descriptors), best_median[p] (may be NULL) that
 * median.  Host arrays; elapsed_ms (may be NULL): kernel time. */
int msorb_distinctive_descriptors(int device, const uint8_t* descriptors, const int* obs_begin, int n_points,
                                  int* best_idx, int* best_median, float* elapsed_ms);

/* ------------------------------------------------------------------------------------------------
 * Frame::isInFrustum (src/Frame.cc:512-571, pinhole / Nleft == -1) over n map points — the pre-pass of
 * Tracking::SearchLocalPoints (src/Tracking.cc:3343-3361); its outputs are exactly the per-point arrays
 * msorb_search_by_projection_mps takes.  The caller skips nothing: points the reference does not visit
 * (mnLastFrameSeen == frame id, isBad()) are simply not passed.
 * ---------------------------------------------------------------------------------------------- */
typedef struct msorb_frustum {
    float Rcw[9];                       /* mRcw, row major */
    float tcw[3];                       /* mtcw */
    float Ow[3];                        /* mOw (camera centre) */
    float fx, fy, cx, cy;               /* Pinhole mvParameters[0..3] */
    float min_x, max_x, min_y, max_y;   /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    float mbf;                          /* baseline * fx */
    float log_scale_factor;             /* mfLogScaleFactor */
    int n_scale_levels;                 /* mnScaleLevels */
} msorb_frustum;

/* Inputs: pos_w / normal = GetWorldPos() / GetNormal() (3 floats per point), max_distance / min_distance =
 * mfMaxDistance / mfMinDistance (the 1.2 / 0.8 invariance factors are applied inside, MapPoint.cc:528-538).
 * Outputs (host, n entries): track_in_view = the return value / mbTrackInView; proj_x, proj_y = mTrackProjX/Y
 * (-1 when the point is behind the camera or outside the image, the projection otherwise, like :515-540);
 * proj_xr, track_depth, scale_level (PredictScale, MapPoint.cc:557-572), view_cos — written when in view, 0
 * otherwise (the reference leaves stale values there that no caller reads).  elapsed_ms may be NULL. */
int msorb_is_in_frustum(int device, const msorb_frustum* f, float viewing_cos_limit, int n, const float* pos_w,
                        const float* normal, const float* max_distance, const float* min_distance,
                        uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr, float* track_depth,
                        int* scale_level, float* view_cos, float* elapsed_ms);

/* ------------------------------------------------------------------------------------------------
 * The tracking front-end of one frame as ONE device-resident chain — BASELINE configs[2] ("extract +
 * SearchByProjection inside the full Tracking loop"), SURVEY.md 8f-2.  The extractor's outputs do not travel to the host and
 * back on their way into the matcher: Frame::AssignFeaturesToGrid (src/Frame.cc:385-416, PosInGrid :657-667) runs on the
 * device keypoints, Frame::isInFrustum writes the window queries of ORBmatcher::SearchByProjection on the device, and one
 * block comes back for the sequential claim replay.
 * ---------------------------------------------------------------------------------------------- */

/* msorb_frame_set from DEVICE arrays (e.g. the outputs of msorb_extract_batch): keypoints (cv::KeyPoint layout), 32-byte
 * descriptors, mvuRight (NULL = all -1), on the frame's device.  The grid is built by a device counting sort that keeps the
 * ascending keypoint index inside a cell (the reference's push_back order, Frame.cc:405-414). */
int msorb_frame_set_device(msorb_frame* f, const msorb_keypoint* d_keypoints, int n, const uint8_t* d_descriptors,
                           const float* d_u_right, float min_x, float max_x, float min_y, float max_y,
                           const float* scale_factors, int nlevels);

/* Inspection: mGrid of the frame as CSR — cell_begin[64*48 + 1] (cell = ix*48 + iy = mGrid[ix][iy]), cell_idx[*n_assigned]
 * keypoint indices in the reference's insertion order. */
int msorb_frame_grid(msorb_frame* f, int* cell_begin, int* cell_idx, int capacity, int* n_assigned);

/* Frame::Frame(imLeft, imRight, ...) up to and including AssignFeaturesToGrid (Frame.cc:119-137, :385-416): msorb_extract_stereo
 * (same arguments, same host outputs) and, on the same stream before the one synchronisation, the frame handle `f` is filled
 * from the device-resident left keypoints / descriptors / mvuRight.  min_x .. max_y = mnMinX .. mnMaxY (rectified: 0, cols, 0,
 * rows).  `h` and `f` must live on one device.  Afterwards `f` serves every msorb_search_* entry like a frame loaded with
 * msorb_frame_set. */
int msorb_extract_stereo_frame(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                               size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left,
                               uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right,
                               int capacity, float* u_right, float* depth, int* n_oob, float min_x, float max_x, float min_y,
                               float max_y);

/* Tracking::SearchLocalPoints from its second loop on (src/Tracking.cc:3343-3388): Frame::isInFrustum(pMP, viewing_cos_limit)
 * for the m local map points (Frame.cc:512-571; inputs as msorb_is_in_frustum; visit[i] = the loop reaches the point: not
 * mnLastFrameSeen == mnId, not isBad(); NULL = all) and ORBmatcher(nnratio).SearchByProjection(F, mvpLocalMapPoints, th,
 * bFarPoints, thFarPoints) (ORBmatcher.cc:43-142; bad / sparsified / mp_desc / obs / frame_mp as msorb_search_by_projection_mps)
 * as one chain on the device: one upload, frustum + window queries, window search, one read-back, then the claim replay.
 * Outputs: the per-point scratch of isInFrustum (any may be NULL) and frame_mp / *nmatches.  Results are identical to
 * msorb_is_in_frustum followed by msorb_search_by_projection_mps. */
int msorb_search_local_points(msorb_frame* f, const msorb_frustum* frustum, float viewing_cos_limit, int m, const float* pos_w,
                              const float* normal, const float* max_distance, const float* min_distance, const uint8_t* visit,
                              const uint8_t* bad, const uint8_t* sparsified, const uint8_t* mp_desc, const int* obs, int* frame_mp,
                              float th, int far_points, float th_far_points, float nnratio, uint8_t* track_in_view, float* proj_x,
                              float* proj_y, float* proj_xr, float* track_depth, int* scale_level, float* view_cos, int* nmatches);

/* msorb_extract_stereo_frame + msorb_search_local_points in ONE call with ONE synchronisation, for a caller that knows the
 * pose before the images arrive (motion-model or IMU prediction, Tracking.cc:2800-2835 / PredictStateIMU): H2D of the images
 * and of the map points, extraction of both eyes, ComputeStereoMatches, AssignFeaturesToGrid, isInFrustum, window search and
 * the read-back are one stream of work; the host only replays the claims.  A new frame holds no map points
 * (Frame.cc:139), so frame_mp[capacity] is an OUTPUT here (-1 = none).  *rounds (may be NULL) = device rounds of the window
 * search (1 unless a candidate list was exhausted by earlier claims). */
int msorb_track_frontend(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                         size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left, uint8_t* desc_left,
                         int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right, int capacity, float* u_right,
                         float* depth, int* n_oob, float min_x, float max_x, float min_y, float max_y, const msorb_frustum* frustum,
                         float viewing_cos_limit, int m, const float* pos_w, const float* normal, const float* max_distance,
                         const float* min_distance, const uint8_t* visit, const uint8_t* bad, const uint8_t* sparsified,
                         const uint8_t* mp_desc, const int* obs, int* frame_mp, float th, int far_points, float th_far_points,
                         float nnratio, uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr, float* track_depth,
                         int* scale_level, float* view_cos, int* nmatches, int* rounds);

/* ------------------------------------------------------------------------------------------------
 * TrackWithMotionModel's search (src/Tracking.cc:2833-2870): ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1941-2152, rectified rig / Nleft == -1) with the projection of :1962-1990 ON THE DEVICE.  The last
 * frame's points stay resident on the current frame's handle between calls (the retry at 2 * th, Tracking.cc:2861-2868, is a
 * second msorb_search_last_frame without another upload).
 * ---------------------------------------------------------------------------------------------- */
typedef struct msorb_motion_model {
    float q[4];                         /* Tcw.unit_quaternion() as Sophus stores it: x, y, z, w (:1951) */
    float t[3];                         /* Tcw.translation() */
    float fx, fy, cx, cy;               /* Pinhole mvParameters[0..3] (Pinhole.cpp:43-49) */
    float mbf;                          /* CurrentFrame.mbf (:2019) */
    int forward, backward;              /* bForward / bBackward (:1957-1958): two scalar pose operations of the caller */
} msorb_motion_model;

/* The last frame's side of the search, n entries = LastFrame.N: has_point[i] = LastFrame.mvpMapPoints[i] && !mvbOutlier[i]
 * (:1962-1965), pos_w = pMP->GetWorldPos() (3 floats), octave = LastFrame.mvKeys[i].octave (:1986), angle =
 * LastFrame.mvKeysUn[i].angle (:2044), mp_desc = pMP->GetDescriptor() (32 bytes).  Entries without a point are not read.
 * Copies the arrays (the caller's may be reused at once) and starts the upload on the frame's stream; the table then lives on
 * the handle until the next msorb_frame_set_last_points.  May be called before the frame itself is set. */
int msorb_frame_set_last_points(msorb_frame* cur, int n, const uint8_t* has_point, const float* pos_w, const int* octave,
                                const float* angle, const uint8_t* mp_desc);
/* n of the table resident on the handle (the length the proj_* arrays of msorb_search_last_frame need); -1: no table set. */
int msorb_frame_last_points_count(const msorb_frame* cur);

/* The search against the resident table: per point x3Dc = Tcw * x3Dw (Sophus' quaternion action, so3.hpp:358-367, in the
 * float convention stated in DESIGN.md), invzc < 0 / image-bounds rejections (:1973-1983), radius = th * mvScaleFactors[octave],
 * the forward / backward / default level band (:1993-1998), the window search with the occupancy and mvuRight filters
 * (:2011-2022), then — on the host, in last-frame order — the sequential claims (:2035-2038) and the rotation histogram
 * (:2040-2057, :2129-2149).  Map-point ids: last-frame point i = id i; obs[n_obs] = Observations() per id (n_obs >= n of the
 * table; ids >= n are points the current frame already holds); cur_mp[N] in / out (-1 = none).  proj_valid / proj_u / proj_v /
 * proj_ur (may be NULL, n entries): what :1962-1983 and :2019 computed — the inputs msorb_search_by_projection_frames takes, for
 * inspection.  Results equal msorb_search_by_projection_frames on those inputs. */
int msorb_search_last_frame(msorb_frame* cur, const msorb_motion_model* mm, const int* obs, int n_obs, int* cur_mp, float th,
                            int check_orientation, int* nmatches, uint8_t* proj_valid, float* proj_u, float* proj_v, float* proj_ur);

/* Frame::Frame(imLeft, imRight, ...) (Frame.cc:119-137, as msorb_extract_stereo_frame) AND the motion-model search in ONE
 * call with ONE synchronisation: the pose guess mVelocity * mLastFrame.GetPose() (Tracking.cc:2854) does not depend on the new
 * images, so images + last-frame table go up together and extraction, ComputeStereoMatches, AssignFeaturesToGrid, projection,
 * window search and the read-back are one stream of work.  A new frame holds no map points (Frame.cc:139; Tracking.cc:2857
 * clears them anyway): cur_mp[capacity] is an OUTPUT, obs has one entry per last-frame point (n of the table). */
int msorb_track_frontend_motion(msorb_extractor* h, msorb_frame* f, const uint8_t* left, const uint8_t* right, int rows, int cols,
                                size_t stride_left, size_t stride_right, float mb, float mbf, msorb_keypoint* kps_left,
                                uint8_t* desc_left, int* n_left, msorb_keypoint* kps_right, uint8_t* desc_right, int* n_right,
                                int capacity, float* u_right, float* depth, int* n_oob, float min_x, float max_x, float min_y,
                                float max_y, const msorb_motion_model* mm, const int* obs, int* cur_mp, float th,
                                int check_orientation, int* nmatches);

/* The device part of the same chain for a BATCH of frames whose features are device resident (offline throughput and the
 * measurement of the windowed Hamming rate): frame b's keypoints / descriptors / count are image b*frame_step of an
 * msorb_extract_batch output (frame_step = 2: the left images of interleaved stereo pairs), d_u_right[b*capacity ..] its
 * mvuRight (msorb_stereo_matches_batch output; NULL = none).  Per frame: frusta[b] (HOST array) and m map points in the device
 * arrays d_pos_w [n_frames][3m], d_normal, d_max_distance, d_min_distance [n_frames][m], d_flags (bit 0 visit, bit 1 isBad,
 * bit 2 mbSparsified), d_mp_desc [n_frames][m][32].  Three launches: grids, frustum + queries, window search against frames
 * that hold no map points yet.  d_topk[n_frames][m][16]: per map point the 8 best candidates in the reference's scan order —
 * 8 keypoint indices (-1 = none) then 8 distances; d_track_in_view (may be NULL) [n_frames][m]; d_cell_begin / d_cell_idx (may
 * be NULL) receive the grids ([n_frames][3073] / [n_frames][capacity]).  elapsed_ms[3] (may be NULL) = device time of the three
 * launches; *n_pairs (may be NULL; HOST) = Hamming distances evaluated by the window search. */
int msorb_track_batch(int device, int n_frames, const msorb_keypoint* d_keypoints, const uint8_t* d_descriptors,
                      const float* d_u_right, const int* d_counts, int frame_step, int capacity, float min_x, float max_x,
                      float min_y, float max_y, const float* scale_factors, int nlevels, const msorb_frustum* frusta,
                      float viewing_cos_limit, int m, const float* d_pos_w, const float* d_normal, const float* d_max_distance,
                      const float* d_min_distance, const uint8_t* d_flags, const uint8_t* d_mp_desc, float th, int far_points,
                      float th_far_points, int* d_topk, uint8_t* d_track_in_view, int* d_cell_begin, int* d_cell_idx,
                      float* elapsed_ms, unsigned long long* n_pairs);

#ifdef __cplusplus
}
#endif
#endif /* MSORB_H */
