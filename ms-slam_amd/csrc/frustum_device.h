// Frame::isInFrustum (src/Frame.cc:512-571, pinhole / Nleft == -1 branch) for ONE map point: the device function shared by
// frustum_kernel (frustum.hip: SoA in / SoA out) and local_points_kernel (track.hip: the same test fused with the query
// set-up of ORBmatcher::SearchByProjection).  Float expressions are written with explicit fmaf / __fdiv_rn and sqrtf in the
// association the reference compiles to (oracle/frustum_oracle.cc header); log() is glibc's logf restated.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>

#include "../../include/msorb.h"
#include "logf_restated.h"

namespace msorb {

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return __fmaf_rn(a0, b0, __fmaf_rn(a1, b1, __fmul_rn(a2, b2)));
}
__device__ __forceinline__ int x86_float_to_int(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

struct FrustumOut {
    uint8_t in_view;
    float px, py, pxr, depth, vc;
    int level;
};

__device__ __forceinline__ FrustumOut frustum_point(const msorb_frustum& F, float cos_limit, float P0, float P1, float P2,
                                                    float n0, float n1, float n2, float max_d, float min_d) {
    FrustumOut o;
    o.in_view = 0; o.px = -1.0f; o.py = -1.0f; o.pxr = 0.0f; o.depth = 0.0f; o.vc = 0.0f; o.level = 0;
    const float Pc0 = __fadd_rn(dot3(F.Rcw[0], P0, F.Rcw[1], P1, F.Rcw[2], P2), F.tcw[0]);
    const float Pc1 = __fadd_rn(dot3(F.Rcw[3], P0, F.Rcw[4], P1, F.Rcw[5], P2), F.tcw[1]);
    const float PcZ = __fadd_rn(dot3(F.Rcw[6], P0, F.Rcw[7], P1, F.Rcw[8], P2), F.tcw[2]);
    do {
        if (PcZ < 0.0f) break;
        const float u = __fadd_rn(__fdiv_rn(__fmul_rn(F.fx, Pc0), PcZ), F.cx);
        const float v = __fadd_rn(__fdiv_rn(__fmul_rn(F.fy, Pc1), PcZ), F.cy);
        if (u < F.min_x || u > F.max_x) break;
        if (v < F.min_y || v > F.max_y) break;
        o.px = u;
        o.py = v;
        const float maxD = __fmul_rn(1.2f, max_d);
        const float minD = __fmul_rn(0.8f, min_d);
        const float PO0 = __fsub_rn(P0, F.Ow[0]), PO1 = __fsub_rn(P1, F.Ow[1]), PO2 = __fsub_rn(P2, F.Ow[2]);
        const float dist = sqrtf(dot3(PO0, PO0, PO1, PO1, PO2, PO2));
        if (dist < minD || dist > maxD) break;
        const float viewCos = __fdiv_rn(dot3(PO0, n0, PO1, n1, PO2, n2), dist);
        if (viewCos < cos_limit) break;
        const float ratio = __fdiv_rn(max_d, dist);
        int nScale = x86_float_to_int(ceilf(__fdiv_rn(glibc_logf(ratio), F.log_scale_factor)));
        if (nScale < 0) nScale = 0;
        else if (nScale >= F.n_scale_levels) nScale = F.n_scale_levels - 1;
        o.in_view = 1;
        o.pxr = __fmaf_rn(-F.mbf, __fdiv_rn(1.0f, PcZ), u);
        o.depth = sqrtf(dot3(Pc0, Pc0, Pc1, Pc1, PcZ, PcZ));
        o.level = nScale;
        o.vc = viewCos;
    } while (0);
    return o;
}

}  // namespace msorb
