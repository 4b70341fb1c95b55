// sinf/cosf exactly as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h},
// the ARM optimized-routines algorithm): double-precision reduction by pi/2 with one multiply-subtract and
// two short double polynomials, the result rounded once to float.  The reference calls them through
// std::cos/std::sin(float) (ORBextractor.cc:112) and the descriptor taps depend on every bit of a,b
// (cvRound(x*b + y*a) flips on a 1-ulp change), so the device restates the algorithm operation by
// operation instead of using an approximate device cosf.
//
// Shared between the HIP kernels and a host-side exhaustive check against the installed glibc
// (tests/sincosf_check.cc).  FUSED selects the contraction GCC applies in glibc's *_fma ifunc
// variants (a + b*c -> fma(b,c,a)).  Domain: 0 <= y < 120.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MSORB_HD __host__ __device__ __forceinline__
#else
#define MSORB_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MSORB_DMUL(a, b) __dmul_rn((a), (b))
#define MSORB_DADD(a, b) __dadd_rn((a), (b))
#define MSORB_DFMA(a, b, c) __fma_rn((a), (b), (c))
#else
#include <math.h>
#define MSORB_DMUL(a, b) ((a) * (b))
#define MSORB_DADD(a, b) ((a) + (b))
#define MSORB_DFMA(a, b, c) fma((a), (b), (c))
#endif

namespace msorb {

template <bool FUSED>
MSORB_HD double sc_muladd(double a, double b, double c) {  // c + a*b
    if (FUSED) return MSORB_DFMA(a, b, c);
    return MSORB_DADD(c, MSORB_DMUL(a, b));
}

template <bool FUSED>
MSORB_HD void glibc_sincosf(float y, float* sinp, float* cosp) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    uint32_t bits;
#if defined(__HIP_DEVICE_COMPILE__)
    bits = __float_as_uint(y);
#else
    memcpy(&bits, &y, 4);
#endif
    const uint32_t top = (bits >> 20) & 0x7ff;  // abstop12
    double x = (double)y;
    int n = 0;
    if (top < 0x3f4u) {        // abstop12(pi/4 = 0x1.921FB6p-1f)
        if (top < 0x398u) {    // abstop12(0x1p-12f)
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
    } else {  // reduce_fast
        const double r = MSORB_DMUL(x, HPI_INV);
        n = ((int32_t)r + 0x800000) >> 24;
        x = sc_muladd<FUSED>(-(double)n, HPI, x);  // x - n*hpi
    }
    const double x2 = MSORB_DMUL(x, x);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[n & 3]
    const double neg = (n & 2) ? -1.0 : 1.0;                         // table[1] negates c0..c4
    const double xs = MSORB_DMUL(x, sgn);
    // sine polynomial of xs (sinf_poly, even quadrant)
    double sp;
    {
        const double x3 = MSORB_DMUL(xs, x2);
        const double s1 = sc_muladd<FUSED>(x2, S3, S2);
        const double x7 = MSORB_DMUL(x3, x2);
        const double s = sc_muladd<FUSED>(x3, S1, xs);
        sp = sc_muladd<FUSED>(x7, s1, s);
    }
    // cosine polynomial (sinf_poly, odd quadrant), coefficients of table[(n>>1)&1]
    double cp;
    {
        const double x4 = MSORB_DMUL(x2, x2);
        const double c2 = sc_muladd<FUSED>(x2, neg * C4, neg * C3);
        const double c1 = sc_muladd<FUSED>(x2, neg * C1, neg * C0);
        const double x6 = MSORB_DMUL(x4, x2);
        const double c = sc_muladd<FUSED>(x4, neg * C2, c1);
        cp = sc_muladd<FUSED>(x6, c2, c);
    }
    // sinf -> sinf_poly(.., n): even = sine poly, odd = cosine poly;  cosf -> sinf_poly(.., n ^ 1)
    *sinp = (float)((n & 1) ? cp : sp);
    *cosp = (float)((n & 1) ? sp : cp);
}

}  // namespace msorb
