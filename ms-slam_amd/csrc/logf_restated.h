// glibc >= 2.27 logf (sysdeps/ieee754/flt-32/e_logf.c, the ARM optimized-routines algorithm): what
// `log(ratio)` in MapPoint::PredictScale (src/MapPoint.cc:565, float argument -> std::log(float) -> logf) resolves
// to on the reference's platform.  Restated in double exactly as published (16-entry table, degree-3 polynomial),
// fp contraction off.  Checked against the installed glibc over EVERY positive finite float (2 139 095 039 values,
// 0 mismatches; tests/logf_check.cc, strided in CI, exhaustive with MSORB_EXHAUSTIVE=1).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define MSORB_HD __host__ __device__ __forceinline__
#else
#define MSORB_HD inline
#endif

namespace msorb {

MSORB_HD float glibc_logf(float x) {
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const double Ln2 = 0x1.62e42fefa39efp-1;
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        const float y = x * 0x1p23f;  // subnormal: normalise
        memcpy(&ix, &y, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    float zf;
    memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * invc[i] - 1;
    const double y0 = logc[i] + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace msorb
