// Exhaustive check: for which constants c is   q = x*r; e = fma(-q, c, x); q' = fma(e, r, q)   (r = RN(1/c))
// equal to the IEEE quotient x/c for EVERY finite binary32 x?   gcc -O2 -ffp-contract=off -fopenmp tools/check_divconst.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float fb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void) {
    const float cs[] = {255.0f, 1000.0f, 3.1415926535897932f, 20.0f, 80.0f, 25.0f, 65535.0f, 64.0f, 3.0f, 120.0f, 6.2831853071795864f};
    for (unsigned k = 0; k < sizeof cs/sizeof cs[0]; k++) {
        const float c = cs[k], r = 1.0f/c;
        long bad = 0; uint32_t first = 0;
        #pragma omp parallel for reduction(+:bad)
        for (long i = 0; i < (1L << 32); i++) {
            const uint32_t u = (uint32_t)i;
            const float x = fb(u);
            if (!isfinite(x)) continue;
            const float want = x/c;
            const float q = x*r;
            const float e = fmaf(-q, c, x);
            const float got = fmaf(e, r, q);
            if (memcmp(&want, &got, 4)) { bad++; if (!first) first = u; }
        }
        printf("c = %-12.9g r = %-14.9g mismatches = %ld%s\n", c, r, bad, bad ? "  (e.g. x bits" : "");
        if (bad) printf("   first x = %g\n", fb(first));
    }
    return 0;
}
