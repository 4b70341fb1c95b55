// VALU issue-cost micro-benchmark for gfx950: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
// (8 waves per SIMD, 8 independent chains per lane; prints cycles per wave-instruction per SIMD at a nominal 2.4 GHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, ASM)                                                                         \
    __global__ void NAME(uint32_t* out, int iters) {                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = blockIdx.x * 3 + 1, c = 0x0c020c00u;                                         \
        for (int it = 0; it < iters; it++) {                                                      \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)             \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                              : "v"(b), "v"(c));)                                                 \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
    }
#define XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define BCNT(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define ALIGNB(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define DOT2(i) "v_dot2_u32_u16 %" #i ", %8, %9, %" #i "\n"
#define MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define PKMIN(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define MIN3(i) "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
#define BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
KERNEL(k_xor, XOR) KERNEL(k_add, ADD) KERNEL(k_bcnt, BCNT) KERNEL(k_perm, PERM) KERNEL(k_alignb, ALIGNB) KERNEL(k_dot4, DOT4)
KERNEL(k_dot2, DOT2) KERNEL(k_mad24, MAD24) KERNEL(k_mullo, MULLO) KERNEL(k_pksub, PKSUB) KERNEL(k_pkmin, PKMIN) KERNEL(k_min3, MIN3)
KERNEL(k_bfe, BFE) KERNEL(k_lshlor, LSHLOR) KERNEL(k_andor, ANDOR) KERNEL(k_cnd, CNDMASK) KERNEL(k_fma, FMA32) KERNEL(k_sad, SAD)
#define AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define LSHRV(i) "v_lshrrev_b32 %" #i ", %8, %" #i "\n"
#define LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define MAXU(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define MOV(i) "v_mov_b32 %" #i ", %8\n"
#define ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define CMP(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
#define CNDE(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define DPP(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define MAX3(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define CVTF(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define FMA64(i) ""
#define PKMUL24(i) "v_mul_u32_u24_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define LERP(i) "v_lerp_u8 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_and, AND) KERNEL(k_lshr, LSHR) KERNEL(k_lshrv, LSHRV) KERNEL(k_lshl, LSHL) KERNEL(k_mini, MINI) KERNEL(k_maxu, MAXU) KERNEL(k_sub, SUB)
KERNEL(k_mul24, MUL24) KERNEL(k_mov, MOV) KERNEL(k_add3, ADD3) KERNEL(k_cmp, CMP) KERNEL(k_cnds, CNDS) KERNEL(k_cnde, CNDE) KERNEL(k_med3, MED3)
KERNEL(k_dpp, DPP) KERNEL(k_adddpp, ADDDPP) KERNEL(k_pkadd, PKADD) KERNEL(k_max3, MAX3) KERNEL(k_xad, XAD) KERNEL(k_cvtf, CVTF)
KERNEL(k_rndne, RNDNE) KERNEL(k_mulf, MULF) KERNEL(k_mul24sdwa, PKMUL24) KERNEL(k_lerp, LERP)
typedef void (*kern_t)(uint32_t*, int);
static void run(const char* name, kern_t k) {
    static uint32_t* d = nullptr;
    if (!d) (void)hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 1000, blocks = 2048;  // 8 waves per SIMD
    k<<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(e0); k<<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 64;
    printf("%-16s %7.3f ms  %5.2f cycles@2.4GHz per wave-instruction per SIMD  (%.1f T lane-ops/s)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / winstr, winstr * 64 / (ms * 1e-3) / 1e12);
}
int main() {
    printf("-- fast class --\n");
    run("v_xor_b32", k_xor); run("v_and_b32", k_and); run("v_add_u32", k_add); run("v_sub_u32", k_sub); run("v_lshrrev(imm)", k_lshr);
    run("v_lshrrev(vgpr)", k_lshrv); run("v_mov_b32", k_mov); run("v_mul_f32", k_mulf); run("v_fma_f32", k_fma);
    printf("-- slow class --\n");
    run("v_lshlrev(imm)", k_lshl); run("v_min_i32", k_mini); run("v_max_u32", k_maxu); run("v_mul_u32_u24", k_mul24); run("v_mad_u32_u24", k_mad24);
    run("v_mul_lo_u32", k_mullo); run("v_add3_u32", k_add3); run("v_xad_u32", k_xad); run("v_bfe_u32", k_bfe); run("v_lshl_or_b32", k_lshlor);
    run("v_and_or_b32", k_andor); run("v_min3_i32", k_min3); run("v_max3_i32", k_max3); run("v_med3_i32", k_med3); run("v_bcnt_u32_b32", k_bcnt);
    run("v_perm_b32", k_perm); run("v_alignbyte_b32", k_alignb); run("v_sad_u8", k_sad); run("v_dot4_u32_u8", k_dot4); run("v_dot2_u32_u16", k_dot2);
    run("v_pk_add_u16", k_pkadd); run("v_pk_sub_i16", k_pksub); run("v_pk_min_i16", k_pkmin); run("v_cvt_f32_i32", k_cvtf); run("v_rndne_f32", k_rndne);
    run("v_cmp_lt_i32", k_cmp); run("v_cndmask sgpr", k_cnds); run("v_mov_dpp", k_dpp); run("v_add_dpp", k_adddpp);
    run("v_mul_u32_u24_sdwa", k_mul24sdwa); run("v_lerp_u8", k_lerp);
    printf("-- outlier (unexplained; vcc not written in the loop) --\n");
    run("v_cndmask_e32 vcc", k_cnde);
    return 0;
}
