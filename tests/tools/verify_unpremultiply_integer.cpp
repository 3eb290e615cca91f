/* verify_unpremultiply_integer.cpp -- test tool (not shipped).  Exhaustively checks the integer un-premultiply of
 * libavif_amd/csrc/exactdiv.h (unpremultiplyByEstimate) against the reference's float expression (src/alpha.c:367-381)
 *      min(floorf((float)c * maxF / (float)a + 0.5f), maxF)
 * for every 16-bit code c and every 0 < a < max, max in {255, 1023, 4095}, with the reciprocal estimate 1 / (2a) taken
 * correctly rounded and perturbed by -2 .. +2 ulp (the kernels' v_rcp_f32 is good to 1 ulp).  Likewise quotient65536ByEstimate
 * (floor(65536 / a), 0 < a < 256) against the integer division, and unpremultiplyByLowEstimate (estimate biased low, one-sided correction).
 * Prints one line per (max, perturbation); exit status 1 on any mismatch.  Build: g++ -O2 -ffp-contract=off (no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "exactdiv.h"

using namespace avifhip;

static float nudge(float f, int ulps)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (uint32_t)((int32_t)u + ulps);
    memcpy(&f, &u, 4);
    return f;
}

int main()
{
    int failures = 0;
    const unsigned maxima[3] = { 255u, 1023u, 4095u };
    for (unsigned maxv : maxima) {
        const float maxF = (float)maxv;
        const unsigned cEnd = (maxv == 255u) ? 256u : 65536u; // 8-bit channels cannot hold more; 16-bit containers can hold any code
        for (int ulps = -2; ulps <= 2; ++ulps) {
            unsigned long long bad = 0, badLow = 0, tested = 0;
            for (unsigned a = 1; a < maxv; ++a) {
                const float r = nudge(1.0f / (float)(2u * a), ulps);
                const volatile float rLowV = r * kUnpremultiplyBias;
                const float rLow = rLowV;
                for (unsigned c = 0; c < cEnd; ++c) {
                    const volatile float t = (float)c * maxF; // (volatile: one rounding per operation, as written)
                    const volatile float q = t / (float)a;
                    const float rounded = floorf(q + 0.5f);
                    const unsigned want = (unsigned)(rounded < maxF ? rounded : maxF);
                    bad += (unpremultiplyByEstimate(c, a, maxv, r) != want);
                    badLow += (unpremultiplyByLowEstimate(c, a, maxv, rLow) != want);
                    ++tested;
                }
            }
            // the operands the kernels form for the two ends of the alpha range: a == 0 (the reference answers 0) divides 0 by 1, a == max (the
            // reference leaves the channel alone) takes the general form and must return c for every c <= max (the kernels restore larger ones)
            {
                const volatile float rOneV = nudge(1.0f, ulps) * kUnpremultiplyBias, rMaxV = nudge(1.0f / (float)(2u * maxv), ulps) * kUnpremultiplyBias;
                const float rOne = rOneV, rMax = rMaxV;
                for (unsigned c = 0; c < cEnd; ++c) {
                    badLow += (unpremultiplyByLowEstimateOperands(c, 0u, 1u, 0u, maxv, rOne) != 0u);
                    if (c <= maxv)
                        badLow += (unpremultiplyByLowEstimateOperands(c, maxv, 2u * maxv, 2u * maxv, maxv, rMax) != c);
                }
            }
            printf("max=%u ulps=%+d tested=%llu mismatches=%llu\n", maxv, ulps, tested, bad);
            printf("max=%u ulps=%+d low-estimate form tested=%llu mismatches=%llu\n", maxv, ulps, tested, badLow);
            failures += bad != 0 || badLow != 0;
        }
    }
    for (int ulps = -2; ulps <= 2; ++ulps) { // ARGBUnattenuate's reciprocals
        unsigned long long bad = 0, tested = 0;
        for (unsigned a = 1; a < 256; ++a, ++tested)
            bad += (quotient65536ByEstimate(a, nudge(1.0f / (float)a, ulps)) != 65536u / a);
        printf("quotient65536 ulps=%+d tested=%llu mismatches=%llu\n", ulps, tested, bad);
        failures += bad != 0;
    }
    return failures ? 1 : 0;
}
