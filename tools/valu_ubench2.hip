// Second VALU issue-cost micro-benchmark for gfx950 (instructions the first one left out or measured oddly):
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench2.hip -o /tmp/valu_ubench2 && /tmp/valu_ubench2
// 8 waves per SIMD, 8 independent chains per lane; prints cycles per wave-instruction per SIMD at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, NI, ASM)                                                                     \
    __global__ void NAME(uint32_t* out, int iters) {                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = blockIdx.x * 3 + 1, c = 0x0c020c00u;                                         \
        for (int it = 0; it < iters; it++) {                                                      \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)             \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                              : "v"(b), "v"(c) : "vcc", "s10", "s11", "s12", "s13");)             \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
    }                                                                                             \
    static const int NAME##_ni = NI;
#define CMPCND(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define CMPCND64(i) "v_cmp_lt_i32 s[10:11], %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define CMPONLY(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_add_u32 %" #i ", %" #i ", %8\n"
#define CNDVCC64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define CNDVCC32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define MAD64(i) "v_mad_u64_u32 v[40:41], s[12:13], %" #i ", 44, v[42:43]\nv_add_u32 %" #i ", %" #i ", v40\n"
#define MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define MBCNTH(i) "v_mbcnt_hi_u32_b32 %" #i ", %8, %" #i "\n"
#define FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define READL(i) "v_readfirstlane_b32 s10, %" #i "\nv_add_u32 %" #i ", s10, %" #i "\n"
#define LSHL64(i) "v_lshlrev_b64 v[40:41], %8, v[42:43]\nv_add_u32 %" #i ", %" #i ", v40\n"
#define MADI24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define SUBREV(i) "v_subrev_u32 %" #i ", %8, %" #i "\n"
#define BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define PKFMA(i) "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[40:41]\n"
#define ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define SDWAADD(i) "v_add_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define CMPCND2(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define CMPXCND(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_add_u32 %" #i ", %" #i ", %9\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define CMPXXCND(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_add_u32 %" #i ", %" #i ", %9\nv_xor_b32 %" #i ", %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define SMOVCND(i) "s_mov_b64 vcc, s[12:13]\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define CMP64CND2(i) "v_cmp_lt_i32 s[10:11], %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define CMPCND2E64(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, vcc\n"
#define ADDCOCI(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\nv_addc_co_u32 %" #i ", vcc, %" #i ", %9, vcc\n"
#define MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define MIN3F(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define MED3F(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define MINU16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define MIN3I16(i) "v_min3_i16 %" #i ", %" #i ", %8, %9\n"
#define SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define MADF(i) "v_mad_f32 %" #i ", %" #i ", %8, %9\n"
#define FMAMIX(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9\n"
#define PKMINF16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define PKADDF32(i) "v_pk_add_f32 v[40:41], v[42:43], v[40:41]\n"
#define CVTU0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define XOR3(i) "v_xor3_b32 %" #i ", %" #i ", %8, %9\n"
#define ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define SADU16(i) "v_sad_u16 %" #i ", %8, %9, %" #i "\n"
#define MSAD(i) "v_msad_u8 %" #i ", %8, %9, %" #i "\n"
KERNEL(k_minf, 1, MINF) KERNEL(k_min3f, 1, MIN3F) KERNEL(k_max3f, 1, MAX3F) KERNEL(k_med3f, 1, MED3F) KERNEL(k_maxf, 1, MAXF)
KERNEL(k_minu16, 1, MINU16) KERNEL(k_min3i16, 1, MIN3I16) KERNEL(k_subf, 1, SUBF) KERNEL(k_pkminf16, 1, PKMINF16)
KERNEL(k_pkaddf32, 1, PKADDF32) KERNEL(k_cvtu0, 1, CVTU0) KERNEL(k_addlshl, 1, ADDLSHL) KERNEL(k_lshladd, 1, LSHLADD)
KERNEL(k_sadu16, 1, SADU16) KERNEL(k_msad, 1, MSAD)
KERNEL(k_cmpcnd2, 3, CMPCND2) KERNEL(k_cmpxcnd, 3, CMPXCND) KERNEL(k_cmpxxcnd, 4, CMPXXCND) KERNEL(k_smovcnd, 1, SMOVCND)
KERNEL(k_cmp64cnd2, 3, CMP64CND2) KERNEL(k_cmpcnd2e64, 3, CMPCND2E64) KERNEL(k_addcoci, 2, ADDCOCI)
KERNEL(k_cmpcnd, 2, CMPCND) KERNEL(k_cmpcnd64, 2, CMPCND64) KERNEL(k_cmponly, 2, CMPONLY) KERNEL(k_cndvcc64, 1, CNDVCC64)
KERNEL(k_cndvcc32, 1, CNDVCC32) KERNEL(k_mad64, 2, MAD64) KERNEL(k_mbcnt, 1, MBCNT) KERNEL(k_mbcnth, 1, MBCNTH) KERNEL(k_ffbl, 1, FFBL)
KERNEL(k_cvtub, 1, CVTUB) KERNEL(k_addco, 1, ADDCO) KERNEL(k_readl, 2, READL) KERNEL(k_lshl64, 2, LSHL64) KERNEL(k_madi24, 1, MADI24)
KERNEL(k_subrev, 1, SUBREV) KERNEL(k_bfi, 1, BFI) KERNEL(k_or3, 1, OR3) KERNEL(k_pkfma, 1, PKFMA) KERNEL(k_addf, 1, ADDF)
KERNEL(k_sdwaadd, 1, SDWAADD) KERNEL(k_ashr, 1, ASHR) KERNEL(k_not, 1, NOT) KERNEL(k_or, 1, OR)
// fast + slow class mixes: alternating (xor, bcnt per chain) against runs (8 xor, then 8 bcnt), and runs of 16
#define XB_ALT(i) "v_xor_b32 %" #i ", %" #i ", %8\nv_bcnt_u32_b32 %" #i ", %9, %" #i "\n"
#define XONLY(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define BONLY(i) "v_bcnt_u32_b32 %" #i ", %9, %" #i "\n"
#define XB_RUN8(i) XONLY(i)
KERNEL(k_xb_alt, 2, XB_ALT)
#define KERNEL_RUNS(NAME, NI, REPS)                                                               \
    __global__ void NAME(uint32_t* out, int iters) {                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = blockIdx.x * 3 + 1, c = 0x0c020c00u;                                         \
        for (int it = 0; it < iters; it++) {                                                      \
            REP8(asm volatile(REPS(XONLY(0) XONLY(1) XONLY(2) XONLY(3) XONLY(4) XONLY(5) XONLY(6) XONLY(7)) \
                              REPS(BONLY(0) BONLY(1) BONLY(2) BONLY(3) BONLY(4) BONLY(5) BONLY(6) BONLY(7)) \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                              : "v"(b), "v"(c));)                                                 \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
    }                                                                                             \
    static const int NAME##_ni = NI;
#define ONCE(X) X
#define TWICE(X) X X
KERNEL_RUNS(k_xb_run8, 2, ONCE) KERNEL_RUNS(k_xb_run16, 4, TWICE)
typedef void (*kern_t)(uint32_t*, int);
static void run(const char* name, kern_t k, int ni) {
    static uint32_t* d = nullptr;
    if (!d) (void)hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 1000, blocks = 2048;  // 8 waves per SIMD
    k<<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(e0); k<<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)blocks * 4 * iters * 64;   // instruction groups per SIMD-wave
    printf("%-44s %7.3f ms  %6.2f cycles@2.4GHz per group of %d instruction(s) per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / groups, ni);
}
#define RUN(label, k) run(label, k, k##_ni)
int main() {
    RUN("v_xor + v_bcnt alternating", k_xb_alt); RUN("8 x v_xor then 8 x v_bcnt (per chain: 1 + 1)", k_xb_run8);
    RUN("16 x v_xor then 16 x v_bcnt (per chain: 2 + 2)", k_xb_run16);
    RUN("v_cmp (vcc) + v_cndmask_e32 (vcc)", k_cmpcnd); RUN("v_cmp (sgpr pair) + v_cndmask_e64 (sgpr pair)", k_cmpcnd64);
    RUN("v_cmp (vcc) + 2 x v_cndmask_e32 (vcc)", k_cmpcnd2); RUN("v_cmp (vcc) + v_add + v_cndmask_e32 (vcc)", k_cmpxcnd);
    RUN("v_cmp (vcc) + v_add + v_xor + v_cndmask_e32", k_cmpxxcnd); RUN("s_mov vcc + v_cndmask_e32 (1 VALU)", k_smovcnd);
    RUN("v_cmp (sgpr) + 2 x v_cndmask_e64 (sgpr)", k_cmp64cnd2); RUN("v_cmp (vcc) + 2 x v_cndmask_e64 (vcc)", k_cmpcnd2e64);
    RUN("v_add_co + v_addc_co (vcc chain)", k_addcoci);
    RUN("v_cmp (vcc) + v_add_u32", k_cmponly); RUN("v_cndmask_e64 ..., vcc (vcc constant)", k_cndvcc64); RUN("v_cndmask_e32 ..., vcc (vcc constant)", k_cndvcc32);
    RUN("v_mad_u64_u32 + v_add_u32", k_mad64); RUN("v_mbcnt_lo_u32_b32", k_mbcnt); RUN("v_mbcnt_hi_u32_b32", k_mbcnth); RUN("v_ffbl_b32", k_ffbl);
    RUN("v_cvt_f32_ubyte1", k_cvtub); RUN("v_add_co_u32 (writes vcc)", k_addco); RUN("v_readfirstlane + v_add_u32 (sgpr)", k_readl);
    RUN("v_lshlrev_b64 + v_add_u32", k_lshl64); RUN("v_mad_i32_i24", k_madi24); RUN("v_subrev_u32", k_subrev); RUN("v_bfi_b32", k_bfi);
    RUN("v_or3_b32", k_or3); RUN("v_pk_fma_f32 (x8, one chain)", k_pkfma); RUN("v_add_f32", k_addf); RUN("v_add_u32_sdwa", k_sdwaadd);
    RUN("v_ashrrev_i32 (imm)", k_ashr); RUN("v_not_b32", k_not); RUN("v_or_b32", k_or);
    RUN("v_min_f32", k_minf); RUN("v_max_f32", k_maxf); RUN("v_min3_f32", k_min3f); RUN("v_max3_f32", k_max3f); RUN("v_med3_f32", k_med3f);
    RUN("v_min_u16", k_minu16); RUN("v_min3_i16", k_min3i16); RUN("v_sub_f32", k_subf); RUN("v_pk_min_f16", k_pkminf16);
    RUN("v_pk_add_f32 (x8, one chain)", k_pkaddf32); RUN("v_cvt_f32_ubyte0", k_cvtu0); RUN("v_add_lshl_u32", k_addlshl);
    RUN("v_lshl_add_u32", k_lshladd); RUN("v_sad_u16", k_sadu16); RUN("v_msad_u8", k_msad);
    return 0;
}
