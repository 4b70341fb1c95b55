// Keypoint selection for quotas whose workspace no workgroup's LDS holds (a level quota above ~1 800 keypoints: nfeatures above
// ~8 000 at 1.2 / 8 levels — the monocular initialisation extractor of Tracking.cc:601, 5 * nFeatures): the SAME selection code
// (quadtree_device.h's generation-synchronous DistributeOctTree, the executor of quadtree_devex_device.h) compiled over plain
// pointers, its workspace a slice of a global-memory buffer per (image, level) instance.  Every access of the workspace is a
// global load / store / atomic (L2-resident: 200 KB per instance) instead of a ds_* instruction — several times slower per
// generation than the LDS form, and still on the device: rounds 1-5 handed such quotas to the host twin (orb_host.cc), a CPU
// stage in the middle of the product path.  Same results by construction: one source, two address spaces (and 32-bit candidate
// labels, qt::Label, where the LDS build packs 16: their 14-bit node slots end at a level quota of 4 092).
#define MSORB_QT_GLOBAL_WORKSPACE 1
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"
#include "quadtree_device.h"
#include "quadtree_devex_device.h"

namespace msorb {

constexpr int kQtGlobalThreads = 256;
constexpr int kQtGlobalPC = 0;   // no register-resident candidates: that form packs a 16-bit label beside the response; here labels are 32 bits
                                 // (qt::Label: node slots up to 4 N + 16 <= 65535, a level quota of 16 379) and live in global memory

__global__ __launch_bounds__(kQtGlobalThreads) void quadtree_select_global_kernel(QtLevels lv, const Cand16* __restrict__ compact, const int* __restrict__ img_base,
                                                                                 const int* __restrict__ level_count, qt::Label* __restrict__ label,
                                                                                 int* __restrict__ sel_pt, int* __restrict__ sel_n, int sel_stride, int ws_N,
                                                                                 int ws_nini, char* __restrict__ ws, size_t ws_stride) {
    const int level = blockIdx.y, img = blockIdx.x;
    const int* lc = level_count + (size_t)img * lv.nlevels;
    int off = img_base[img];
    for (int l = 0; l < level; l++) off += lc[l];
    const int n = lc[level];
    char* const mem = ws + ((size_t)img * lv.nlevels + level) * ws_stride;
    qt::Workspace w;
    qt::workspace_carve(w, mem, ws_N, ws_nini);
    DevExT<false> ex;
    ex.kLaneSort = false;   // (wave_introsort64 keeps its items in lanes whatever the address space; the round form is the one measured in batches)
    ex.dbg = 0;
    ex.nt = kQtGlobalThreads;
    int* out = sel_pt + (size_t)img * sel_stride + lv.sel_off[level];
    const int kept = qt::select<kQtGlobalPC>(ex, reinterpret_cast<const qt::Pt*>(compact + off), n, label + off, lv.W[level], lv.H[level], lv.quota[level], w, out, 0);
    if (threadIdx.x == 0) sel_n[(size_t)img * lv.nlevels + level] = kept;
}

size_t quadtree_global_workspace_stride(const QtLevels& lv) {
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    return (qt::workspace_bytes(maxN, max_ini) + 255) & ~(size_t)255;
}

void launch_quadtree_select_global(const QtLevels& lv, const Cand16* compact, const int* img_base, const int* level_count, uint32_t* label, int* sel_pt,
                                   int* sel_n, int sel_stride, int n_images, char* ws, hipStream_t s) {
    int maxN = 1, max_ini = 1;
    for (int l = 0; l < lv.nlevels; l++) { maxN = max(maxN, lv.quota[l]); max_ini = max(max_ini, lv.n_ini[l]); }
    hipLaunchKernelGGL(quadtree_select_global_kernel, dim3(n_images, lv.nlevels), dim3(kQtGlobalThreads), 0, s, lv, compact, img_base, level_count, label,
                       sel_pt, sel_n, sel_stride, maxN, max_ini, ws, quadtree_global_workspace_stride(lv));
}

}  // namespace msorb
