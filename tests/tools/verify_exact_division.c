/* verify_exact_division.c -- test tool (not shipped).  Exhaustively checks that the reciprocal form the HIP
 * kernels use for division by a plan constant,
 *      q0 = x * r;  e = fma(-q0, d, x);  q = fma(e, r, q0)        with r = RN(1/d),
 * returns exactly RN(x / d) (IEEE-754 binary32 division) for every x of the stated domain.
 *   kg   <hex bits of d>           x over three full binades (all 2^23 mantissas, both signs).  The identity is
 *                                  invariant under scaling x by powers of two (no overflow/underflow occurs in the
 *                                  kernels' operand range), so a binade covers every exponent.
 *   norm <range> <bias> <maxcp>    x = (float)cp - bias for every code point cp in [0, maxcp], d = (float)range.
 * Prints one line per request: "<kind> <d> tested=<n> mismatches=<m>".  Build with -O2 -mfma -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float asFloat(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float recipForm(float x, float d, float r)
{
    const float q0 = x * r;
    const float e = fmaf(-q0, d, x);
    return fmaf(e, r, q0);
}

int main(int argc, char ** argv)
{
    int i = 1, failures = 0;
    while (i < argc) {
        if (!strcmp(argv[i], "kg") && i + 1 < argc) {
            const float d = asFloat((uint32_t)strtoul(argv[i + 1], NULL, 16));
            const float r = 1.0f / d;
            unsigned long long n = 0, bad = 0;
            for (uint32_t e = 126; e <= 128; ++e)
                for (uint32_t m = 0; m < (1u << 23); ++m)
                    for (uint32_t s = 0; s < 2; ++s) {
                        const float x = asFloat((s << 31) | (e << 23) | m);
                        bad += (recipForm(x, d, r) != x / d);
                        ++n;
                    }
            printf("kg %.9g tested=%llu mismatches=%llu\n", d, n, bad);
            failures += bad != 0;
            i += 2;
        } else if (!strcmp(argv[i], "norm") && i + 3 < argc) {
            const float d = (float)atoi(argv[i + 1]), bias = (float)atoi(argv[i + 2]);
            const long maxcp = atol(argv[i + 3]);
            const float r = 1.0f / d;
            unsigned long long n = 0, bad = 0;
            for (long cp = 0; cp <= maxcp; ++cp) {
                const float x = (float)cp - bias;
                bad += (recipForm(x, d, r) != x / d);
                ++n;
            }
            printf("norm %.9g bias=%.9g tested=%llu mismatches=%llu\n", d, bias, n, bad);
            failures += bad != 0;
            i += 4;
        } else {
            fprintf(stderr, "usage: %s (kg <hexbits> | norm <range> <bias> <maxcp>)...\n", argv[0]);
            return 2;
        }
    }
    return failures ? 1 : 0;
}
