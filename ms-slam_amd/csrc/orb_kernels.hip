// Hand-written gfx950 (CDNA4, wave64) kernels of the ORB extractor.  Integer / bit work only — no MFMA.
//
//   pyr_resize_kernel   ComputePyramid            ORBextractor.cc:1170-1195  (cv::resize INTER_LINEAR, 8-bit)
//   fast_cells_kernel   cell loop + cv::FAST      ORBextractor.cc:805-872    (FAST-9/16 score, 3x3 NMS, th fallback)
//   cand_* kernels      vToDistributeKeys order   ORBextractor.cc:863-867    (row-major cell / scan order compaction)
//   gauss7_kernel       GaussianBlur 7x7 s=2      ORBextractor.cc:1132-1133  (Q8.8 separable, reflect-101)
//   describe_kernel     IC_Angle + rBRIEF         ORBextractor.cc:76-146,894-895,1138
//
// Arithmetic follows SURVEY.md Appendix A / oracle/cvprims.h exactly (bit-exact contract).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"
#include "sincosf_restated.h"

namespace msorb {

// ------------------------------------------------------------------------------------------------
// Pyramid level l from level l-1.  One thread = 4 horizontally adjacent destination pixels of one
// row, written as one aligned 32-bit store (pitch is a multiple of 64).  block = 64 x 4.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyr_resize_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base,
                                                         const ResizeTap* __restrict__ tx,
                                                         const ResizeTap* __restrict__ ty) {
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= dst.h || dx0 >= dst.w) return;
    const ResizeTap vy = ty[dy];
    const uint8_t* s0 = src.base + (size_t)img * src.img_stride + (size_t)vy.i0 * src.pitch;
    const uint8_t* s1 = src.base + (size_t)img * src.img_stride + (size_t)vy.i1 * src.pitch;
    const int b0 = vy.c0, b1 = vy.c1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        if (dx < dst.w) {
            const ResizeTap vx = tx[dx];
            const int h0 = s0[vx.i0] * vx.c0 + s0[vx.i1] * vx.c1;
            const int h1 = s1[vx.i0] * vx.c0 + s1[vx.i1] * vx.c1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    uint8_t* d = dst_base + (size_t)img * dst.img_stride + (size_t)dy * dst.pitch + dx0;
    *reinterpret_cast<uint32_t*>(d) = packed;
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 on one reference cell ROI per workgroup.
//   phase 0  stage the ROI (<= 76x76 bytes) in LDS
//   phase 1  cheap necessary test on the two antipodal compass pairs at minTh; survivors -> LDS list
//   phase 2  full score S = max(A,-B)-1 for survivors only (all lanes busy); S>=minTh -> score plane
//   phase 3  strict 3x3 NMS inside the ROI's detection area; iniTh set if non-empty, else minTh set;
//            survivors written in scan order (ascending y, then x) to the cell's fixed slot run
// The FAST score is threshold independent for detected corners (cornerScore returns
// max(th, A, -B) - 1 and a corner has max(A,-B) > th), so one score plane serves both thresholds:
// corner at th  <=>  S >= th, and NMS at iniTh keeps exactly the minTh survivors with S >= iniTh.
// ------------------------------------------------------------------------------------------------
constexpr int kTileMax = 80;                 // max ROI edge supported (w_cell, h_cell <= 70 + 6)
constexpr int kTilePitch = kTileMax;         // bytes
constexpr int kScorePitch = kTileMax;        // (rw-6)+2 <= 72

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

__device__ __forceinline__ int fast_score16(const uint8_t* p /* LDS, centre */) {
    // circle offsets (x,y): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * kTilePitch];
    d[1] = v - p[3 * kTilePitch + 1];
    d[2] = v - p[2 * kTilePitch + 2];
    d[3] = v - p[1 * kTilePitch + 3];
    d[4] = v - p[3];
    d[5] = v - p[-1 * kTilePitch + 3];
    d[6] = v - p[-2 * kTilePitch + 2];
    d[7] = v - p[-3 * kTilePitch + 1];
    d[8] = v - p[-3 * kTilePitch];
    d[9] = v - p[-3 * kTilePitch - 1];
    d[10] = v - p[-2 * kTilePitch - 2];
    d[11] = v - p[-1 * kTilePitch - 3];
    d[12] = v - p[-3];
    d[13] = v - p[1 * kTilePitch - 3];
    d[14] = v - p[2 * kTilePitch - 2];
    d[15] = v - p[3 * kTilePitch - 1];
    int mn3[16], mx3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        mn3[i] = min3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
        mx3[i] = max3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    }
    int A = -512, B = 512;
#pragma unroll
    for (int i = 0; i < 16; i++) {  // arc i..i+8
        A = max(A, min3i(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]));
        B = min(B, max3i(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]));
    }
    return max(A, -B) - 1;
}

__global__ __launch_bounds__(256) void fast_cells_kernel(PyramidView pyr, const CellDesc* __restrict__ cells,
                                                         int ini_th, int min_th, int slots_per_image,
                                                         Cand16* __restrict__ slots, int* __restrict__ cell_count,
                                                         int n_cells) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileMax * kTilePitch];
    __shared__ uint8_t score[kTileMax * kScorePitch];
    __shared__ uint16_t surv[(kTileMax - 6) * (kTileMax - 6)];
    __shared__ int n_surv;
    __shared__ int wave_cnt[2][4];

    const int cell_id = blockIdx.x;
    const int img = blockIdx.y;
    const CellDesc cd = cells[cell_id];
    const LevelView lv = pyr.lv[cd.level];
    const int rw = cd.rw, rh = cd.rh;
    const int dw = rw - 6, dh = rh - 6;  // detection area (FAST skips 3 px on every side of its input)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // phase 0
    const uint8_t* src = lv.base + (size_t)img * lv.img_stride + (size_t)cd.y0 * lv.pitch + cd.x0;
    for (int i = tid; i < rw * rh; i += 256) {
        const int y = i / rw, x = i - y * rw;
        tile[y * kTilePitch + x] = src[(size_t)y * lv.pitch + x];
    }
    for (int i = tid; i < (dh + 2) * kScorePitch; i += 256) score[i] = 0;
    if (tid == 0) n_surv = 0;
    __syncthreads();

    const int n_det = dw * dh;
    // phase 1
    for (int p0 = 0; p0 < n_det; p0 += 256) {
        const int p = p0 + tid;
        bool pass = false;
        int y = 0, x = 0;
        if (p < n_det) {
            y = p / dw; x = p - y * dw;
            const uint8_t* c = &tile[(y + 3) * kTilePitch + x + 3];
            const int v = c[0];
            const int d0 = v - c[3 * kTilePitch], d8 = v - c[-3 * kTilePitch];
            const int d4 = v - c[3], d12 = v - c[-3];
            // every 9-arc contains one pixel of each antipodal pair
            const bool dark = max(d0, d8) > min_th && max(d4, d12) > min_th;
            const bool bright = min(d0, d8) < -min_th && min(d4, d12) < -min_th;
            pass = dark || bright;
        }
        const unsigned long long m = __ballot(pass);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&n_surv, __popcll(m));
        base = __shfl(base, 0);
        if (pass) surv[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)((y << 8) | x);
    }
    __syncthreads();

    // phase 2
    const int ns = n_surv;
    for (int i = tid; i < ns; i += 256) {
        const int yx = surv[i];
        const int y = yx >> 8, x = yx & 255;
        const int s = fast_score16(&tile[(y + 3) * kTilePitch + x + 3]);
        if (s >= min_th) score[(y + 1) * kScorePitch + x + 1] = (uint8_t)s;
    }
    __syncthreads();

    // phase 3a: NMS flags per owned pixel (bit k = chunk k), counts of the iniTh set
    uint32_t keep = 0, keep_ini = 0;
    int chunk = 0;
    for (int p0 = 0; p0 < n_det; p0 += 256, chunk++) {
        const int p = p0 + tid;
        if (p < n_det) {
            const int y = p / dw, x = p - y * dw;
            const uint8_t* sp = &score[(y + 1) * kScorePitch + x + 1];
            const int s = sp[0];
            if (s) {
                int m = max3i(sp[-kScorePitch - 1], sp[-kScorePitch], sp[-kScorePitch + 1]);
                m = max3i(m, sp[-1], sp[1]);
                m = max(m, max3i(sp[kScorePitch - 1], sp[kScorePitch], sp[kScorePitch + 1]));
                if (s > m) {
                    keep |= 1u << chunk;
                    if (s >= ini_th) keep_ini |= 1u << chunk;
                }
            }
        }
    }
    const int any_ini = __syncthreads_or(keep_ini != 0);
    const uint32_t sel = any_ini ? keep_ini : keep;

    // phase 3b: ordered emission — pixel index p = chunk*256 + tid ascends with (chunk, wave, lane)
    Cand16* out = slots + (size_t)img * slots_per_image + cd.slot_off;
    int running = 0;
    chunk = 0;
    for (int p0 = 0; p0 < n_det; p0 += 256, chunk++) {
        const bool f = (sel >> chunk) & 1u;
        const unsigned long long m = __ballot(f);
        const int buf = chunk & 1;
        if (lane == 0) wave_cnt[buf][wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int cw = wave_cnt[buf][w];
            if (w < wave) before += cw;
            total += cw;
        }
        if (f) {
            const int p = p0 + tid;
            const int y = p / dw, x = p - y * dw;
            Cand16 c;
            c.x = (uint16_t)(cd.x0 + 3 + x - kMinBorder);
            c.y = (uint16_t)(cd.y0 + 3 + y - kMinBorder);
            c.score = score[(y + 1) * kScorePitch + x + 1];
            c.pad = 0;
            out[running + before + __popcll(m & ((1ull << lane) - 1))] = c;
        }
        running += total;
    }
    if (tid == 0) cell_count[(size_t)img * n_cells + cell_id] = running;
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: per image exclusive scan of the cell counts (cells are already in the
// reference's row-major cell order), a one-block scan over images, then a gather.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_scan_cells_kernel(const int* __restrict__ cell_count, int n_cells,
                                                              const int* __restrict__ level_cell_begin, int nlevels,
                                                              int* __restrict__ cell_off, int* __restrict__ level_count,
                                                              int* __restrict__ img_total) {
    __shared__ int part[256];
    __shared__ int lvl[kMaxLevels];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int* cnt = cell_count + (size_t)img * n_cells;
    int* off = cell_off + (size_t)img * n_cells;
    const int per = (n_cells + 255) / 256;
    const int b = tid * per, e = min(b + per, n_cells);
    int s = 0;
    for (int i = b; i < e; i++) s += cnt[i];
    part[tid] = s;
    if (tid < kMaxLevels) lvl[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; }
        img_total[img] = acc;
    }
    __syncthreads();
    int acc = part[tid];
    for (int i = b; i < e; i++) { off[i] = acc; acc += cnt[i]; }
    // per-level totals
    for (int l = 0; l < nlevels; l++) {
        const int lb = level_cell_begin[l], le = level_cell_begin[l + 1];
        int t = 0;
        for (int i = max(b, lb); i < min(e, le); i++) t += cnt[i];
        if (t) atomicAdd(&lvl[l], t);
    }
    __syncthreads();
    if (tid < nlevels) level_count[(size_t)img * nlevels + tid] = lvl[tid];
}

__global__ __launch_bounds__(256) void cand_scan_images_kernel(const int* __restrict__ img_total, int n_images,
                                                               int* __restrict__ img_base /* n_images+1 */) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (n_images + 255) / 256;
    const int b = tid * per, e = min(b + per, n_images);
    int s = 0;
    for (int i = b; i < e; i++) s += img_total[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; }
        img_base[n_images] = acc;
    }
    __syncthreads();
    int acc = part[tid];
    for (int i = b; i < e; i++) { img_base[i] = acc; acc += img_total[i]; }
}

__global__ __launch_bounds__(64) void cand_gather_kernel(const CellDesc* __restrict__ cells, int n_cells,
                                                         int slots_per_image, const Cand16* __restrict__ slots,
                                                         const int* __restrict__ cell_count,
                                                         const int* __restrict__ cell_off,
                                                         const int* __restrict__ img_base, Cand16* __restrict__ compact) {
    const int cell_id = blockIdx.x, img = blockIdx.y;
    const int n = cell_count[(size_t)img * n_cells + cell_id];
    if (n == 0) return;
    const Cand16* s = slots + (size_t)img * slots_per_image + cells[cell_id].slot_off;
    Cand16* d = compact + img_base[img] + cell_off[(size_t)img * n_cells + cell_id];
    for (int i = threadIdx.x; i < n; i += 64) d[i] = s[i];
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, Q8.8 fixed point [18,34,48,56,48,34,18], BORDER_REFLECT_101, per level.
// Tile = 64 x 16 output pixels per 256-thread block; 70 x 22 input pixels staged in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int refl101(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); }

__global__ __launch_bounds__(256) void gauss7_kernel(LevelView src, LevelView dst, uint8_t* __restrict__ dst_base) {
    constexpr int TW = 64, TH = 16, IW = TW + 6, IH = TH + 6;
    __shared__ uint8_t in[IH][IW + 2];
    __shared__ uint16_t hz[IH][TW];
    const int img = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const uint8_t* s = src.base + (size_t)img * src.img_stride;
    const int tid = threadIdx.x;
    for (int i = tid; i < IW * IH; i += 256) {
        const int ty = i / IW, tx = i - ty * IW;
        const int gy = refl101(min(y0 + ty - 3, src.h + 2), src.h);  // rows past the image are never used
        const int gx = refl101(min(x0 + tx - 3, src.w + 2), src.w);
        in[ty][tx] = s[(size_t)gy * src.pitch + gx];
    }
    __syncthreads();
    for (int i = tid; i < TW * IH; i += 256) {
        const int ty = i >> 6, tx = i & 63;
        const uint8_t* r = &in[ty][tx];
        hz[ty][tx] = (uint16_t)(18 * (r[0] + r[6]) + 34 * (r[1] + r[5]) + 48 * (r[2] + r[4]) + 56 * r[3]);
    }
    __syncthreads();
    // each thread: 4 adjacent output pixels of one row
    const int ty = tid >> 4, tx = (tid & 15) * 4;
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < src.h && gx < dst.pitch) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t acc = 18u * (hz[ty][tx + i] + hz[ty + 6][tx + i]) + 34u * (hz[ty + 1][tx + i] + hz[ty + 5][tx + i]) +
                                 48u * (hz[ty + 2][tx + i] + hz[ty + 4][tx + i]) + 56u * hz[ty + 3][tx + i];
            packed |= (((acc + 32768u) >> 16) & 255u) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(dst_base + (size_t)img * dst.img_stride + (size_t)gy * dst.pitch + gx) = packed;
    }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: one wave64 per selected keypoint, 4 keypoints per block.
// ------------------------------------------------------------------------------------------------
struct PatchTables {
    int8_t pattern[256 * 4];  // (x0,y0,x1,y1) per pair
    int8_t pu[768], pv[768];  // the 749 (u,v) offsets of the circular patch, padded
};
__constant__ PatchTables c_tab;

void upload_patch_tables(const int8_t* pattern, const int* umax, hipStream_t stream) {
    PatchTables t;
    for (int i = 0; i < 1024; i++) t.pattern[i] = pattern[i];
    int n = 0;
    for (int v = -kHalfPatch; v <= kHalfPatch; v++) {
        const int d = umax[v < 0 ? -v : v];
        for (int u = -d; u <= d; u++) { t.pu[n] = (int8_t)u; t.pv[n] = (int8_t)v; n++; }
    }
    for (; n < 768; n++) { t.pu[n] = 0; t.pv[n] = 0; }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tab), &t, sizeof(t), 0, hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);
}
constexpr int kPatchPixels = 749;

// cv::fastAtan2 — separate multiply/add, no contraction (oracle/cvprims.h fast_atan2).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__global__ __launch_bounds__(256) void describe_kernel(PyramidView pyr, PyramidView blur, const SelRec* __restrict__ sel,
                                                       const int* __restrict__ sel_count, int sel_stride,
                                                       LevelScale scales, msorb_keypoint* __restrict__ kps,
                                                       uint8_t* __restrict__ desc, int out_stride) {
    const int img = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= sel_count[img]) return;
    const SelRec r = sel[(size_t)img * sel_stride + k];
    const LevelView lv = pyr.lv[r.level];
    const uint8_t* center = lv.base + (size_t)img * lv.img_stride + (size_t)r.y * lv.pitch + r.x;

    // IC_Angle: integer moments over the 749-pixel circular patch (un-blurred level)
    int m10 = 0, m01 = 0;
    for (int i = lane; i < kPatchPixels; i += 64) {
        const int u = c_tab.pu[i], v = c_tab.pv[i];
        const int val = center[v * lv.pitch + u];
        m10 += u * val;
        m01 += v * val;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // steered BRIEF on the blurred level
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.0);
    float a, b;
    glibc_sincosf<true>(__fmul_rn(angle, factor_pi), &b, &a);  // a = cos, b = sin (ORBextractor.cc:112)
    const LevelView bv = blur.lv[r.level];
    const uint8_t* bc = bv.base + (size_t)img * bv.img_stride + (size_t)r.y * bv.pitch + r.x;
    unsigned long long word[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int8_t* pt = &c_tab.pattern[(w * 64 + lane) * 4];
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        // cvRound(x*b + y*a), cvRound(x*a - y*b) with the contraction order of oracle/orb_extractor_oracle.cc
        const int r0 = __float2int_rn(__fmaf_rn(x0, b, __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fmaf_rn(x0, a, -__fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fmaf_rn(x1, b, __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fmaf_rn(x1, a, -__fmul_rn(y1, b)));
        const int t0 = bc[r0 * bv.pitch + q0];
        const int t1 = bc[r1 * bv.pitch + q1];
        word[w] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)img * out_stride + r.dst) * 32);
        d[lane] = lane == 0 ? word[0] : lane == 1 ? word[1] : lane == 2 ? word[2] : word[3];
    }
    if (lane == 0) {
        msorb_keypoint kp;
        const float sc = scales.scale[r.level];
        kp.x = r.level ? __fmul_rn((float)r.x, sc) : (float)r.x;   // keypoint->pt *= scale (ORBextractor.cc:1149-1151)
        kp.y = r.level ? __fmul_rn((float)r.y, sc) : (float)r.y;
        kp.size = scales.patch[r.level];
        kp.angle = angle;
        kp.response = (float)r.score;
        kp.octave = r.level;
        kp.class_id = -1;
        kps[(size_t)img * out_stride + r.dst] = kp;
    }
}

// ---- launch wrappers (called from extractor.hip) --------------------------------------------------
void launch_pyr_resize(const LevelView& src, const LevelView& dst, uint8_t* dst_base, const ResizeTap* tx,
                       const ResizeTap* ty, int n_images, hipStream_t s) {
    dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, n_images);
    hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, s, src, dst, dst_base, tx, ty);
}
void launch_fast_cells(const PyramidView& pyr, const CellDesc* cells, int n_cells, int ini_th, int min_th,
                       int slots_per_image, Cand16* slots, int* cell_count, int n_images, hipStream_t s) {
    hipLaunchKernelGGL(fast_cells_kernel, dim3(n_cells, n_images), dim3(256), 0, s, pyr, cells, ini_th, min_th,
                       slots_per_image, slots, cell_count, n_cells);
}
void launch_cand_compact(const CellDesc* cells, int n_cells, const int* level_cell_begin, int nlevels,
                         int slots_per_image, const Cand16* slots, const int* cell_count, int* cell_off,
                         int* level_count, int* img_total, int* img_base, Cand16* compact, int n_images,
                         hipStream_t s) {
    hipLaunchKernelGGL(cand_scan_cells_kernel, dim3(n_images), dim3(256), 0, s, cell_count, n_cells, level_cell_begin,
                       nlevels, cell_off, level_count, img_total);
    hipLaunchKernelGGL(cand_scan_images_kernel, dim3(1), dim3(256), 0, s, img_total, n_images, img_base);
    hipLaunchKernelGGL(cand_gather_kernel, dim3(n_cells, n_images), dim3(64), 0, s, cells, n_cells, slots_per_image,
                       slots, cell_count, cell_off, img_base, compact);
}
void launch_gauss7(const LevelView& src, const LevelView& dst, uint8_t* dst_base, int n_images, hipStream_t s) {
    dim3 grid((src.w + 63) / 64, (src.h + 15) / 16, n_images);
    hipLaunchKernelGGL(gauss7_kernel, grid, dim3(256), 0, s, src, dst, dst_base);
}
void launch_describe(const PyramidView& pyr, const PyramidView& blur, const SelRec* sel, const int* sel_count,
                     int sel_stride, const LevelScale& scales, msorb_keypoint* kps, uint8_t* desc, int out_stride,
                     int max_sel, int n_images, hipStream_t s) {
    if (max_sel <= 0) return;
    hipLaunchKernelGGL(describe_kernel, dim3((max_sel + 3) / 4, n_images), dim3(256), 0, s, pyr, blur, sel, sel_count,
                       sel_stride, scales, kps, desc, out_stride);
}

}  // namespace msorb
