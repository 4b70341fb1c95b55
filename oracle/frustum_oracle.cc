// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h header).  PARITY UNPINNED (no reference build / vectors here).
//
// Frame::isInFrustum, pinhole / Nleft == -1 branch (/root/reference/src/Frame.cc:512-571), run over the local map
// points as Tracking::SearchLocalPoints does (src/Tracking.cc:3343-3361): Pinhole::project (src/CameraModels/
// Pinhole.cpp:43-49), MapPoint::GetMin/MaxDistanceInvariance (src/MapPoint.cc:528-538), MapPoint::PredictScale
// (src/MapPoint.cc:557-572).  log() is the platform's logf (this file calls libm; the device restates glibc's).
//
// Float association convention (Eigen is not installed, so this cannot be probed here — stated, not pinned):
// Eigen's fixed-size 3-element reductions (row*vector of `mRcw * P`, `squaredNorm`, `dot`) evaluate
// x0 + (x1 + x2) (redux_novec_unroller splits [0,3) into [0,1) and [1,3)); with the reference's
// `-O3 -march=native` (CMakeLists.txt:10-13, GCC contracts a*b+c) that is fma(a0,b0, fma(a1,b1, a2*b2)), and
// `uv(0) - mbf*invz` is fma(-mbf, invz, uv(0)).  The same expressions are spelled out in csrc/frustum.hip.
#include <cmath>
#include <cstdint>

namespace {
inline float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return std::fmaf(a0, b0, std::fmaf(a1, b1, a2 * b2));
}
// (int)ceil(x) as x86 cvttss2si converts it: out-of-range / NaN -> INT_MIN ("integer indefinite")
inline int x86_float_to_int(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int)v;
}
}  // namespace

struct orc_frustum {
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy;
    float min_x, max_x, min_y, max_y;
    float mbf, log_scale_factor;
    int n_scale_levels;
};

extern "C" void orc_is_in_frustum(const orc_frustum* F, float viewingCosLimit, int n, const float* pos_w,
                                  const float* normal, const float* max_distance, const float* min_distance,
                                  uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr,
                                  float* track_depth, int* scale_level, float* view_cos) {
    for (int i = 0; i < n; i++) {
        track_in_view[i] = 0;  // :515-517
        proj_x[i] = -1;
        proj_y[i] = -1;
        proj_xr[i] = 0; track_depth[i] = 0; scale_level[i] = 0; view_cos[i] = 0;  // not written by a failing call
        const float P0 = pos_w[3 * i], P1 = pos_w[3 * i + 1], P2 = pos_w[3 * i + 2];
        // Pc = mRcw * P + mtcw (:523)
        const float Pc0 = dot3(F->Rcw[0], P0, F->Rcw[1], P1, F->Rcw[2], P2) + F->tcw[0];
        const float Pc1 = dot3(F->Rcw[3], P0, F->Rcw[4], P1, F->Rcw[5], P2) + F->tcw[1];
        const float PcZ = dot3(F->Rcw[6], P0, F->Rcw[7], P1, F->Rcw[8], P2) + F->tcw[2];
        const float Pc_dist = std::sqrt(dot3(Pc0, Pc0, Pc1, Pc1, PcZ, PcZ));  // :524
        const float invz = 1.0f / PcZ;                                          // :528
        if (PcZ < 0.0f) continue;                                               // :529-530
        const float u = F->fx * Pc0 / PcZ + F->cx;                              // Pinhole.cpp:45-46
        const float v = F->fy * Pc1 / PcZ + F->cy;
        if (u < F->min_x || u > F->max_x) continue;                             // :534-537
        if (v < F->min_y || v > F->max_y) continue;
        proj_x[i] = u;                                                          // :539-540
        proj_y[i] = v;
        const float maxDistance = 1.2f * max_distance[i];                       // MapPoint.cc:534-538
        const float minDistance = 0.8f * min_distance[i];
        const float PO0 = P0 - F->Ow[0], PO1 = P1 - F->Ow[1], PO2 = P2 - F->Ow[2];
        const float dist = std::sqrt(dot3(PO0, PO0, PO1, PO1, PO2, PO2));       // :546
        if (dist < minDistance || dist > maxDistance) continue;                 // :548-549
        const float viewCos = dot3(PO0, normal[3 * i], PO1, normal[3 * i + 1], PO2, normal[3 * i + 2]) / dist;  // :554
        if (viewCos < viewingCosLimit) continue;                                // :556-557
        // PredictScale (MapPoint.cc:557-572)
        const float ratio = max_distance[i] / dist;
        int nScale = x86_float_to_int(std::ceil(std::log(ratio) / F->log_scale_factor));
        if (nScale < 0) nScale = 0;
        else if (nScale >= F->n_scale_levels) nScale = F->n_scale_levels - 1;
        track_in_view[i] = 1;                                                   // :563-571
        proj_xr[i] = std::fmaf(-F->mbf, invz, u);
        track_depth[i] = Pc_dist;
        scale_level[i] = nScale;
        view_cos[i] = viewCos;
    }
}

// The per-keypoint projection of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
// (/root/reference/src/ORBmatcher.cc:1951-1990, rectified / Nleft == -1): x3Dc = Tcw * x3Dw, invzc, Pinhole::project, the
// image-bounds rejections, ur = uv(0) - mbf*invzc (:2019).  Tcw * x3Dw is Sophus' SE3 action,
//   Thirdparty/Sophus/sophus/se3.hpp:321-324   so3() * p + translation()
//   Thirdparty/Sophus/sophus/so3.hpp:358-367   uv = q.vec().cross(p); uv += uv; return p + q.w() * uv + q.vec().cross(uv);
// with Eigen's 3-vector cross product (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0).
// Float convention — STATED, NOT PINNED (Eigen / Sophus cannot be compiled here): of a difference of two products the first is
// fused, a*b - c*d -> fma(a, b, -(c*d)); p + w*uv -> fma(w, uv, p); sums stay sums.  A probe of this expression tree with this
// image's g++ 11 -O3 -march=x86-64-v3 shows the SLP vectoriser choosing fmsub for some components and fnmadd (the SECOND product
// fused) for others, so no single convention reproduces every build of the reference; the last ulp of u / v decides a match only
// when a keypoint sits exactly on a window or cell border.  The product keeps msorb_search_by_projection_frames (coordinates
// projected by the caller's own build) beside the device projection for that reason.
struct orc_motion_model {
    float q[4], t[3];
    float fx, fy, cx, cy, mbf;
    int forward, backward;
};
namespace {
inline float diff_of_products(float a, float b, float c, float d) { return std::fmaf(a, b, -(c * d)); }
}
extern "C" void orc_project_last_frame(const orc_motion_model* M, float min_x, float max_x, float min_y, float max_y, int n,
                                       const uint8_t* has_point, const float* pos_w, uint8_t* valid, float* u, float* v,
                                       float* ur) {
    const float qx = M->q[0], qy = M->q[1], qz = M->q[2], qw = M->q[3];
    for (int i = 0; i < n; i++) {
        valid[i] = 0; u[i] = 0; v[i] = 0; ur[i] = 0;
        if (!has_point[i]) continue;                                           // :1962-1965
        const float px = pos_w[3 * i], py = pos_w[3 * i + 1], pz = pos_w[3 * i + 2];
        float uvx = diff_of_products(qy, pz, qz, py), uvy = diff_of_products(qz, px, qx, pz), uvz = diff_of_products(qx, py, qy, px);
        uvx += uvx; uvy += uvy; uvz += uvz;
        const float c0 = diff_of_products(qy, uvz, qz, uvy), c1 = diff_of_products(qz, uvx, qx, uvz), c2 = diff_of_products(qx, uvy, qy, uvx);
        const float xc = (std::fmaf(qw, uvx, px) + c0) + M->t[0];
        const float yc = (std::fmaf(qw, uvy, py) + c1) + M->t[1];
        const float zc = (std::fmaf(qw, uvz, pz) + c2) + M->t[2];
        const float invzc = 1.0 / zc;                                          // :1973 (double quotient, rounded to float)
        if (invzc < 0) continue;                                               // :1975-1976
        const float uu = M->fx * xc / zc + M->cx;                              // Pinhole.cpp:45-46
        const float vv = M->fy * yc / zc + M->cy;
        if (uu < min_x || uu > max_x) continue;                                // :1980-1983
        if (vv < min_y || vv > max_y) continue;
        valid[i] = 1; u[i] = uu; v[i] = vv;
        ur[i] = std::fmaf(-M->mbf, invzc, uu);                                 // :2019
    }
}
