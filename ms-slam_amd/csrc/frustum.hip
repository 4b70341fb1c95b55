// Frame::isInFrustum (src/Frame.cc:512-571, pinhole / Nleft == -1 branch) over a batch of map points — the
// pre-pass of Tracking::SearchLocalPoints (src/Tracking.cc:3343-3361) that fills the per-point scratch
// (mbTrackInView, mTrackProjX/Y/XR, mTrackDepth, mnTrackScaleLevel, mTrackViewCos) msorb_search_by_projection_mps
// consumes.  One thread per point, SoA in / SoA out: a pure stream (32 B in, 25 B out per point).
// Float expressions are written with explicit fmaf / __fdiv_rn and sqrtf (correctly rounded in hipcc's default mode;
// __fsqrt_rn is the bare 1-ulp v_sqrt_f32 and is NOT used) in the association
// the reference compiles to (see oracle/frustum_oracle.cc header); log() is glibc's logf restated
// (logf_restated.h).  The library is built with -ffp-contract=off, nothing else is fused.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/msorb.h"
#include "frustum_device.h"

namespace msorb {
// pinned host <-> device on a stream by the copy kernel (orb_kernels.hip; hipMemcpyAsync for unaligned pointers / MSORB_FRAME_COPIES=sdma)
hipError_t small_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
}

namespace msorb {
void set_last_error(const std::string& s);
}
using msorb::set_last_error;

namespace {

__global__ __launch_bounds__(256) void frustum_kernel(msorb_frustum F, float cos_limit, int n,
                                                      const float* __restrict__ pos_w, const float* __restrict__ normal,
                                                      const float* __restrict__ max_distance,
                                                      const float* __restrict__ min_distance,
                                                      uint8_t* __restrict__ track_in_view, float* __restrict__ proj_x,
                                                      float* __restrict__ proj_y, float* __restrict__ proj_xr,
                                                      float* __restrict__ track_depth, int* __restrict__ scale_level,
                                                      float* __restrict__ view_cos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const msorb::FrustumOut o = msorb::frustum_point(F, cos_limit, pos_w[3 * i], pos_w[3 * i + 1], pos_w[3 * i + 2], normal[3 * i],
                                                     normal[3 * i + 1], normal[3 * i + 2], max_distance[i], min_distance[i]);
    track_in_view[i] = o.in_view;
    proj_x[i] = o.px;
    proj_y[i] = o.py;
    proj_xr[i] = o.pxr;
    track_depth[i] = o.depth;
    scale_level[i] = o.level;
    view_cos[i] = o.vc;
}

}  // namespace

extern "C" int msorb_is_in_frustum(int device, const msorb_frustum* f, float viewing_cos_limit, int n, const float* pos_w,
                                   const float* normal, const float* max_distance, const float* min_distance,
                                   uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr,
                                   float* track_depth, int* scale_level, float* view_cos, float* elapsed_ms) {
    if (elapsed_ms) *elapsed_ms = 0;
    if (!f || n < 0 || f->n_scale_levels < 1 ||
        (n > 0 && (!pos_w || !normal || !max_distance || !min_distance || !track_in_view || !proj_x || !proj_y ||
                   !proj_xr || !track_depth || !scale_level || !view_cos)))
        return MSORB_E_INVALID;
    if (n == 0) return MSORB_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libmsorb has no CPU fallback)");
        return MSORB_E_NO_DEVICE;
    }
    // Called once per frame (Tracking::SearchLocalPoints): stream, events, device buffer and pinned staging are kept per
    // calling thread and device.  One buffer: inputs (8 floats / point) then outputs (6 x 4 B + 1 B / point).
    struct Scratch {
        int device = -1;
        hipStream_t s = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        char *d = nullptr, *h = nullptr;
        size_t cap = 0;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (s) (void)hipStreamDestroy(s);
            d = h = nullptr; s = nullptr; e0 = e1 = nullptr; cap = 0; device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch scr;
    const size_t N = (size_t)n;
    const size_t in_bytes = N * 8 * sizeof(float), out_bytes = N * 6 * 4 + N, total = in_bytes + out_bytes;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess && scr.device != device) {
        scr.release();
        scr.device = device;
        e = hipStreamCreateWithFlags(&scr.s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&scr.e0);
        if (e == hipSuccess) e = hipEventCreate(&scr.e1);
    }
    if (e == hipSuccess && total > scr.cap) {
        if (scr.d) (void)hipFree(scr.d);
        if (scr.h) (void)hipHostFree(scr.h);
        scr.d = scr.h = nullptr; scr.cap = 0;
        e = hipMalloc((void**)&scr.d, total + total / 2);
        if (e == hipSuccess) e = hipHostMalloc((void**)&scr.h, total + total / 2, hipHostMallocDefault);
        if (e == hipSuccess) scr.cap = total + total / 2;
    }
    if (e == hipSuccess) {
        hipStream_t s = scr.s;
        float* d_pos = (float*)scr.d;
        float* d_nrm = d_pos + 3 * N;
        float* d_max = d_nrm + 3 * N;
        float* d_min = d_max + N;
        float* d_px = d_min + N;
        float* d_py = d_px + N;
        float* d_pxr = d_py + N;
        float* d_depth = d_pxr + N;
        int* d_level = (int*)(d_depth + N);
        float* d_vc = (float*)(d_level + N);
        uint8_t* d_in = (uint8_t*)(d_vc + N);
        float* h_f = (float*)scr.h;
        std::memcpy(h_f, pos_w, N * 12);
        std::memcpy(h_f + 3 * N, normal, N * 12);
        std::memcpy(h_f + 6 * N, max_distance, N * 4);
        std::memcpy(h_f + 7 * N, min_distance, N * 4);
        e = msorb::small_copy(scr.d, scr.h, in_bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(scr.e0, s);
        if (e == hipSuccess)
            hipLaunchKernelGGL(frustum_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, *f, viewing_cos_limit, n, d_pos,
                               d_nrm, d_max, d_min, d_in, d_px, d_py, d_pxr, d_depth, d_level, d_vc);
        if (e == hipSuccess) e = hipEventRecord(scr.e1, s);
        if (e == hipSuccess) e = msorb::small_copy(scr.h + in_bytes, scr.d + in_bytes, out_bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, scr.e0, scr.e1);
        if (e == hipSuccess) {
            const char* ho = scr.h + in_bytes;
            std::memcpy(proj_x, ho, N * 4);
            std::memcpy(proj_y, ho + N * 4, N * 4);
            std::memcpy(proj_xr, ho + N * 8, N * 4);
            std::memcpy(track_depth, ho + N * 12, N * 4);
            std::memcpy(scale_level, ho + N * 16, N * 4);
            std::memcpy(view_cos, ho + N * 20, N * 4);
            std::memcpy(track_in_view, ho + N * 24, N);
        }
    }
    if (e != hipSuccess) {
        set_last_error(std::string("is_in_frustum: ") + hipGetErrorString(e));
        return MSORB_E_HIP;
    }
    return MSORB_OK;
}
