// Compares csrc/logf_restated.h with the installed glibc logf.  argv[1] = stride over float bit patterns
// (1 = every positive finite float).  Prints "checked N mismatches M".
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../ms-slam_amd/csrc/logf_restated.h"

int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    long bad = 0, n = 0;
    for (uint64_t u = 1; u < 0x7f800000u; u += stride) {
        const uint32_t v = (uint32_t)u;
        float x;
        memcpy(&x, &v, 4);
        const float a = logf(x), b = msorb::glibc_logf(x);
        uint32_t ua, ub;
        memcpy(&ua, &a, 4);
        memcpy(&ub, &b, 4);
        n++;
        if (ua != ub) {
            if (bad < 5) printf("x=%a glibc=%a restated=%a\n", x, a, b);
            bad++;
        }
    }
    // specials
    const float sp[] = {0.0f, -0.0f, -1.0f, INFINITY, 1.0f};
    for (float x : sp) {
        const float a = logf(x), b = msorb::glibc_logf(x);
        if (!((isnan(a) && isnan(b)) || a == b)) { bad++; printf("special %a\n", x); }
    }
    printf("checked %ld mismatches %ld\n", n, bad);
    return bad != 0;
}
