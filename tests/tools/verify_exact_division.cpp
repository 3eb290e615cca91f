/* verify_exact_division.cpp -- test tool (not shipped).  Exhaustively checks that the reciprocal form the tiled HIP
 * kernels use for division by a plan constant,
 *      q = fma(x, hi, x * lo)        with (hi, lo) = splitReciprocal(d, scale)   (libavif_amd/csrc/exactdiv.h),
 * returns exactly scale * RN(x / d) (IEEE-754 binary32 division) for every x of the stated domain, for EVERY divisor
 * on exactdiv.h's verified lists:
 *   kg list       x over three full binades (all 2^23 mantissas, both signs), scale 2 (the kernels fold the
 *                 reference's "2 *" into the constant).  The identity is invariant under scaling x by powers of two
 *                 while x * lo stays normal, so a binade covers every exponent the kernels produce.
 *   integer list  the same mantissa sweep with scale 1, plus every integer x in [-65536, 65536].
 *   chroma denominators (2*(1-kb), 2*(1-kr), encode direction)  the mantissa sweep with scale 1.
 * Prints one line per divisor; exit status 1 if any mismatch.  Build: g++ -O2 -mfma -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "exactdiv.h"

using namespace avifhip;

static float asFloat(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline float form(float x, RcpSplit r)
{
    return fmaf(x, r.hi, x * r.lo);
}
static unsigned long long sweepMantissas(float d, float scale, RcpSplit r, unsigned long long * tested)
{
    unsigned long long bad = 0;
    for (uint32_t e = 126; e <= 128; ++e)
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            const float x = asFloat((e << 23) | m);
            bad += (form(x, r) != scale * (x / d));
            bad += (form(-x, r) != scale * ((-x) / d));
            *tested += 2;
        }
    return bad;
}

int main()
{
    int failures = 0;
    for (uint32_t bits : kVerifiedKgBits) {
        const float d = asFloat(bits);
        const RcpSplit r = splitReciprocal(d, 2.0f);
        unsigned long long n = 0;
        const unsigned long long bad = sweepMantissas(d, 2.0f, r, &n);
        printf("kg %.9g (0x%08x) hi=%.9g lo=%.9g tested=%llu mismatches=%llu\n", d, bits, r.hi, r.lo, n, bad);
        failures += bad != 0;
    }
    for (float d : kVerifiedIntegerDivisors) {
        const RcpSplit r = splitReciprocal(d, 1.0f);
        unsigned long long n = 0;
        unsigned long long bad = sweepMantissas(d, 1.0f, r, &n);
        for (int x = -65536; x <= 65536; ++x) {
            bad += (form((float)x, r) != (float)x / d);
            ++n;
        }
        printf("int %.9g hi=%.9g lo=%.9g tested=%llu mismatches=%llu\n", d, r.hi, r.lo, n, bad);
        failures += bad != 0;
    }
    for (uint32_t bits : kVerifiedChromaDenBits) {
        const float d = asFloat(bits);
        const RcpSplit r = splitReciprocal(d, 1.0f);
        unsigned long long n = 0;
        const unsigned long long bad = sweepMantissas(d, 1.0f, r, &n);
        printf("den %.9g (0x%08x) hi=%.9g lo=%.9g tested=%llu mismatches=%llu\n", d, bits, r.hi, r.lo, n, bad);
        failures += bad != 0;
    }
    return failures ? 1 : 0;
}
