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
