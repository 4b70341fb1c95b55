// Exhaustive host check of ms-slam_amd/csrc/sincosf_restated.h against the installed glibc sinf/cosf
// over every float in [lo, hi].  Prints the mismatch counts of the un-fused and fused variants.
// usage: sincosf_check <lo> <hi> [stride]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../ms-slam_amd/csrc/sincosf_restated.h"

int main(int argc, char** argv) {
    const float lo = argc > 1 ? (float)atof(argv[1]) : 0.f;
    const float hi = argc > 2 ? (float)atof(argv[2]) : 6.2831855f;
    const unsigned stride = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
    uint32_t a, b;
    memcpy(&a, &lo, 4);
    memcpy(&b, &hi, 4);
    unsigned long long n = 0, bad_plain = 0, bad_fused = 0;
    for (uint64_t u = a; u <= b; u += stride) {
        uint32_t uu = (uint32_t)u;
        float x;
        memcpy(&x, &uu, 4);
        const float s = sinf(x), c = cosf(x);
        float s0, c0, s1, c1;
        msorb::glibc_sincosf<false>(x, &s0, &c0);
        msorb::glibc_sincosf<true>(x, &s1, &c1);
        if (memcmp(&s, &s0, 4) || memcmp(&c, &c0, 4)) {
            if (bad_plain < 5) printf("plain mismatch x=%a sin %a vs %a cos %a vs %a\n", x, s, s0, c, c0);
            bad_plain++;
        }
        if (memcmp(&s, &s1, 4) || memcmp(&c, &c1, 4)) {
            if (bad_fused < 5) printf("fused mismatch x=%a sin %a vs %a cos %a vs %a\n", x, s, s1, c, c1);
            bad_fused++;
        }
        n++;
    }
    printf("checked=%llu bad_plain=%llu bad_fused=%llu\n", n, bad_plain, bad_fused);
    return 0;
}
