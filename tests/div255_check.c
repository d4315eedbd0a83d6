/* Exhaustive check of the 3-operation x/255.0f used by the augmentation kernels
 * (augcl_amd/csrc/aug_kernels.hip: div255): q = x*r; q' = fma(fma(-q,255,x), r, q), r = RN(1/255),
 * against the IEEE-754 single-precision quotient for EVERY non-negative finite float (2^31 - 2^23 bit
 * patterns, denormals included).  Negative inputs follow by symmetry (every operation is odd in x).
 * Prints the number of mismatches; exit status 0 iff none.  Built and run by tests/test_div255.py (gcc, host). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float div255_3op(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    const float e = fmaf(-q, 255.0f, x);
    return fmaf(e, r, q);
}

int main(void)
{
    unsigned long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 20) /* the denormal range is ~100x slower per input */
    for (int64_t bits = 0; bits < 0x7f800000LL; ++bits) { /* [+0, +inf) */
        uint32_t u = (uint32_t)bits;
        float x, a, b;
        memcpy(&x, &u, 4);
        volatile float d = 255.0f; /* keeps the compiler from turning the division into a multiplication */
        a = x / d;
        b = div255_3op(x);
        uint32_t ua, ub;
        memcpy(&ua, &a, 4);
        memcpy(&ub, &b, 4);
        bad += (ua != ub);
    }
    printf("%llu\n", bad);
    return bad != 0;
}
