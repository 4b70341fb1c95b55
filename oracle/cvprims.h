// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be linked, imported or executed
// by the product (ms-slam_amd/, bench.py's GPU leg).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
//
// PARITY UNPINNED: the reference (fishmarch/MS-SLAM) ships no tests, fixtures or golden vectors,
// and the arithmetic of the primitives below lives in OpenCV (un-vendored, un-pinned ">= 4.4",
// /root/reference/CMakeLists.txt:35), which is absent from this image, so neither the reference
// nor OpenCV can be run here.  Each primitive restates the published OpenCV 4.x algorithm for
// CV_8UC1 data as recorded in SURVEY.md Appendix A; "bit-exact" in this repo means bit-exact
// against THIS restatement until one golden run on real OpenCV confirms it (tools/pin_opencv.py is the kit for that run).
// What IS checked against independent third-party code present in the image (scikit-image 0.18.3 / scipy 1.7.1 under
// /opt/conda, fixtures in tests/golden/skimage_pins, tests/test_oracle_pins.py): the FAST-9/16 detection set and the
// cornerScore semantics (exact, over a ladder of thresholds), the sampling geometry / border rules of resize and blur (to within
// the fixed-point rounding), fastAtan2 (to OpenCV's documented 0.3 degrees).  Still resting on Appendix A alone: the exact
// fixed-point rounding of resize, the 8-bit Gaussian taps, FAST's 3x3 NMS rule and emission order, fastAtan2's bit patterns.
//
// Scalar CPU restatements of the OpenCV primitives that ORBextractor.cc calls:
//   cvRound/cvFloor/cvCeil        (call sites ORBextractor.cc:80,114,118-119,441,455-456,459,1175)
//   resize(INTER_LINEAR) 8UC1     (ORBextractor.cc:1183)
//   FAST(img, kps, th, true)      (ORBextractor.cc:826,845)       TYPE_9_16 + cornerScore + 3x3 NMS
//   GaussianBlur 7x7 s=2 8UC1     (ORBextractor.cc:1133)          fixed-point Q8.8 path
//   fastAtan2                     (ORBextractor.cc:102)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

// --- A.1 rounding helpers -------------------------------------------------------------------
// cvRound = SSE cvtss2si / cvtsd2si => round-half-to-even in the default rounding mode.
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
static inline short sat_short(float v) {
    int i = cv_round(v);
    return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

// A plane of 8-bit pixels (row-major, stride == cols).
struct Plane {
    int rows = 0, cols = 0;
    std::vector<uint8_t> px;
    Plane() {}
    Plane(int r, int c) : rows(r), cols(c), px((size_t)r * c) {}
    uint8_t* row(int y) { return px.data() + (size_t)y * cols; }
    const uint8_t* row(int y) const { return px.data() + (size_t)y * cols; }
};

static inline int reflect101(int p, int len) {
    if (p < 0) return -p;
    if (p >= len) return 2 * (len - 1) - p;
    return p;
}

// --- A.3 resize, INTER_LINEAR, 8UC1 ---------------------------------------------------------
// 11-bit fixed-point bilinear: horizontal pass to int32, vertical pass
//   dst = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.
struct ResizeTab {
    std::vector<int> ofs;       // source index per destination index
    std::vector<short> coef;    // 2 weights per destination index (x2048, independently rounded)
};
static inline ResizeTab resize_tab(int dlen, int slen, bool clamp_x) {
    ResizeTab t;
    t.ofs.resize(dlen);
    t.coef.resize((size_t)dlen * 2);
    const double inv_scale = (double)dlen / slen;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dlen; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (clamp_x) {  // horizontal table: OpenCV clamps the tap AND zeroes the fraction
            if (s < 0) { f = 0; s = 0; }
            if (s >= slen - 1) { f = 0; s = slen - 1; }
        }
        t.ofs[d] = s;
        t.coef[2 * d] = sat_short((1.f - f) * 2048);
        t.coef[2 * d + 1] = sat_short(f * 2048);
    }
    return t;
}
// The [OpenCV-recall] semantics that could differ in a real OpenCV build, as ONE runtime-selectable table shared by the oracle
// and (through msorb_extractor_set_semantics) the kernels: if a pin run on real OpenCV (tools/pin_opencv.py) disagrees with a
// default, switching the variant is a setting, not a rewrite.  Defaults = SURVEY.md Appendix A.
struct Semantics {
    int gauss_taps[7] = {18, 34, 48, 56, 48, 34, 18};  // Q8 taps of GaussianBlur(7x7, sigma 2): bit-exact path of OpenCV >= 4.2 (sum 256);
                                                       // e.g. {18,34,49,55,49,34,18} (sum 257) for a float-kernel build
    int resize_single_stage = 0;  // 0: VResizeLinear<uchar>: ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2
                                  // 1: generic FixedPtCast: (S0*b0 + S1*b1 + (1 << 21)) >> 22
    int atan2_fma = 0;            // 0: separate multiply / add (x86-64 baseline build); 1: contracted Horner steps (aarch64, -ffp-contract=fast)
    int brief_tap = 0;            // rounding of the rotated BRIEF tap cvRound(x*b + y*a), cvRound(x*a - y*b) (ORBextractor.cc:117-119), which the
                                  // reference leaves to its compiler (-O3 -march=native, CMakeLists.txt:10-13):
                                  // 0: FIRST product fused   fma(x, b, y*a), fma(x, a, -(y*b))   (g++ / clang on an FMA target: probed, tools/probe_brief_tap.cc)
                                  // 1: SECOND product fused  fma(y, a, x*b), fma(-y, b, x*a)
                                  // 2: no contraction        (x*b) + (y*a), (x*a) - (y*b)        (no-FMA target, -ffp-contract=off, MSVC /fp:precise)
};
static inline Semantics& semantics() { static Semantics s; return s; }

static inline void resize_linear_u8(const Plane& src, Plane& dst) {
    const ResizeTab tx = resize_tab(dst.cols, src.cols, true);
    const ResizeTab ty = resize_tab(dst.rows, src.rows, false);
    std::vector<int> h0(dst.cols), h1(dst.cols);
    for (int dy = 0; dy < dst.rows; dy++) {
        // vertical taps are clipped to [0, rows-1] with the weights kept
        int sy0 = ty.ofs[dy], sy1 = sy0 + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 < src.rows ? sy0 : src.rows - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < src.rows ? sy1 : src.rows - 1);
        const uint8_t* s0 = src.row(sy0);
        const uint8_t* s1 = src.row(sy1);
        for (int dx = 0; dx < dst.cols; dx++) {
            int sx = tx.ofs[dx];
            if (sx + 1 >= src.cols) {  // right edge: single tap x ONE (2048)
                h0[dx] = s0[sx] * 2048;
                h1[dx] = s1[sx] * 2048;
            } else {
                int a0 = tx.coef[2 * dx], a1 = tx.coef[2 * dx + 1];
                h0[dx] = s0[sx] * a0 + s0[sx + 1] * a1;
                h1[dx] = s1[sx] * a0 + s1[sx + 1] * a1;
            }
        }
        int b0 = ty.coef[2 * dy], b1 = ty.coef[2 * dy + 1];
        uint8_t* d = dst.row(dy);
        if (semantics().resize_single_stage)
            for (int dx = 0; dx < dst.cols; dx++) d[dx] = (uint8_t)((h0[dx] * b0 + h1[dx] * b1 + (1 << 21)) >> 22);
        else
            for (int dx = 0; dx < dst.cols; dx++)
                d[dx] = (uint8_t)((((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// --- A.4 FAST-9/16 + score + NMS on one ROI -------------------------------------------------
struct FastPt { int x, y, score; };

static const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: largest threshold at which the pixel is still a FAST-9 corner.
static inline int fast_corner_score(const uint8_t* p, int stride, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) {
        const int* o = kCircle[k & 15];
        d[k] = v - p[o[1] * stride + o[0]];
    }
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int m = 4; m <= 8; m++) a = std::min(a, d[k + m]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        for (int m = 6; m <= 8; m++) b = std::max(b, d[k + m]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// Segment test: >= 9 contiguous circle pixels all darker than v-t or all brighter than v+t.
static inline bool fast_is_corner(const uint8_t* p, int stride, int t) {
    const int v = p[0];
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 25; k++) {
        const int* o = kCircle[k & 15];
        const int x = p[o[1] * stride + o[0]];
        if (x < v - t) { if (++run_d > 8) return true; } else run_d = 0;
        if (x > v + t) { if (++run_b > 8) return true; } else run_b = 0;
    }
    return false;
}

// FAST(roi, kps, threshold, nonmaxSuppression=true).  `img` points at the ROI's (0,0);
// neighbours outside the ROI's detection area count as score 0 (rolling 3-row buffer of the call).
static inline void fast9_nms(const uint8_t* img, int stride, int rows, int cols, int threshold,
                             std::vector<FastPt>& out) {
    out.clear();
    if (rows < 7 || cols < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> score((size_t)rows * cols, 0), corner((size_t)rows * cols, 0);
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, stride, threshold)) {
                corner[(size_t)y * cols + x] = 1;
                score[(size_t)y * cols + x] = (uint8_t)fast_corner_score(p, stride, threshold);
            }
        }
    // emit in ascending y, then ascending x; all comparisons strict
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            if (!corner[(size_t)y * cols + x]) continue;
            const int s = score[(size_t)y * cols + x];
            const uint8_t* r0 = &score[(size_t)(y - 1) * cols + x];
            const uint8_t* r1 = &score[(size_t)y * cols + x];
            const uint8_t* r2 = &score[(size_t)(y + 1) * cols + x];
            if (s > r1[1] && s > r1[-1] && s > r0[-1] && s > r0[0] && s > r0[1] && s > r2[-1] && s > r2[0] &&
                s > r2[1])
                out.push_back({x, y, s});
        }
}

// --- A.5 GaussianBlur 7x7 sigma 2, 8UC1, BORDER_REFLECT_101, fixed-point path ----------------
static inline void gaussian7_q88(const Plane& src, Plane& dst) {
    const int* kGauss7 = semantics().gauss_taps;  // default {18,34,48,56,48,34,18}: Q8.8, error-diffusion rounding, sum 256
    dst = Plane(src.rows, src.cols);
    std::vector<uint16_t> h((size_t)src.rows * src.cols);
    for (int y = 0; y < src.rows; y++) {
        const uint8_t* s = src.row(y);
        for (int x = 0; x < src.cols; x++) {
            unsigned acc = 0;
            for (int i = 0; i < 7; i++) acc += kGauss7[i] * s[reflect101(x + i - 3, src.cols)];
            h[(size_t)y * src.cols + x] = (uint16_t)acc;  // <= 255*256
        }
    }
    for (int y = 0; y < src.rows; y++) {
        uint8_t* d = dst.row(y);
        for (int x = 0; x < src.cols; x++) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; j++)
                acc += (uint32_t)kGauss7[j] * h[(size_t)reflect101(y + j - 3, src.rows) * src.cols + x];
            const uint32_t v = (acc + 32768u) >> 16;   // taps summing to 257 can reach 256 / 257 on saturated pixels: saturate_cast
            d[x] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

// --- A.6 fastAtan2 (degrees, [0,360]) -------------------------------------------------------
// Separate float multiply/add in Horner order (OpenCV's baseline x86-64 build has no FMA).
// Compile this translation unit with -ffp-contract=off.
static inline float fast_atan2(float y, float x) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    const bool fma = semantics().atan2_fma != 0;
    auto poly = [&](float cc, float cc2) {
        if (fma) return std::fmaf(std::fmaf(std::fmaf(p7, cc2, p5), cc2, p3), cc2, p1) * cc;
        return (((p7 * cc2 + p5) * cc2 + p3) * cc2 + p1) * cc;
    };
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = poly(c, c2);
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - poly(c, c2);
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace orc
