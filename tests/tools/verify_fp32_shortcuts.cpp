/* verify_fp32_shortcuts.cpp -- test tool (not shipped).  Enumerates the instruction-saving identities the fp32 tiles
 * (libavif_amd/csrc/tile_impl.h) and the alpha passes (kernels_generic.hip, pixel_math.h, pixel_fixed.h) rely on, with the host's IEEE
 * arithmetic (hardware FMA: build with -mfma -ffp-contract=off -I libavif_amd/csrc):
 *
 * 1. quantizeArg: the reference quantises a channel as (T)(0.5f + (c * max)) -- a multiply, an add, a truncation
 *    (src/reformat.c:952-961).  The kernels compute fmaf(c, max, 0.5f): ONE rounding.  Checked: both truncate (and saturate to
 *    [0, max]) to the same integer for EVERY binary32 c (all 2^32 bit patterns, NaNs skipped) and max in {255, 1023, 4095, 65535}.
 *
 * 2. inLoopChannel<UNMUL>: the slow path's un-premultiply divides the three clamped colours of a pixel by the same Ac = a / max
 *    (src/reformat.c:927-934).  The kernels form r = RN(1 / Ac) once -- v_rcp_f32 (1 ulp) and one Newton step,
 *    r = fma(fma(-Ac, r0, 1), r0, r0) -- and then q = fma(fma(-q0, Ac, c), r, q0) with q0 = c * r (Markstein's correction step).
 *    Checked, for every alpha code 0 < a < max of 8-, 10- and 12-bit planes: the Newton step returns the correctly rounded reciprocal from
 *    every estimate within 2 ulp of it, and q equals the IEEE quotient c / Ac for every c in [0.5, 1] (argument "full": every c in
 *    {0} U [2^-40, 1], 1.8e12 quotients, ~25 minutes on 8 cores; profiles/r03_shared_reciprocal_check.txt).  The sequence is homogeneous
 *    in c -- scaling c by a power of two scales q0, the remainder and q by the same power as long as nothing leaves the normal range --
 *    so one binade of c stands for all of them down to 2^-100, far below anything the matrix can produce (its terms are multiples of 2^-70).
 *
 * 3. unpremulRcp / unpremulRcpArg: the integer post-pass min(floorf((float)c * maxF / (float)a + 0.5f), maxF) (src/alpha.c:367-381) with the
 *    same shared-reciprocal division, x = RN(c * maxF) over the integer a.  Checked for every 16-bit code c (8-bit codes for max 255) and
 *    every 0 < a < max, max in {255, 1023, 4095, 65535} -- 4.6e9 pairs -- reciprocal estimates off by up to 2 ulp.
 *
 * 4. quotient65536ByEstimate (exactdiv.h): floor(65536 / a), ARGBUnattenuate's 8.8 reciprocal, from an estimate of 1 / a off by up to
 *    2 ulp, against the integer division, every 0 < a < 256.
 *
 * Prints one line per case ending in "mismatches=0"; exit status 1 otherwise.
 */
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "exactdiv.h"

static inline float fromBits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t toBits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
// v_cvt_u32_f32 followed by the kernels' unsigned min: negatives and NaN give 0, the rest truncates, the min saturates
static inline uint32_t truncSat(float t, uint32_t maxv)
{
    if (!(t > 0.0f))
        return 0;
    if (t >= 4294967296.0f)
        return maxv;
    const uint32_t q = (uint32_t)t;
    return q < maxv ? q : maxv;
}

template <class F>
static void parallel(unsigned n, F f)
{
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n; ++t)
        th.emplace_back(f, t);
    for (auto & x : th)
        x.join();
}

int main(int argc, char ** argv)
{
    const bool full = argc > 1 && !strcmp(argv[1], "full");
    const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
    int failures = 0;

    const float maxima[4] = { 255.0f, 1023.0f, 4095.0f, 65535.0f };
    for (float mx : maxima) {
        std::atomic<uint64_t> bad { 0 }, tested { 0 };
        parallel(nt, [&](unsigned t) {
            uint64_t b = 0, n = 0;
            for (uint64_t i = t; i < (1ull << 32); i += nt) {
                const float c = fromBits((uint32_t)i);
                if (c != c)
                    continue;
                const volatile float p = c * mx; // (volatile: one rounding per operation, as the reference is written)
                const float two = 0.5f + p;
                const float one = fmaf(c, mx, 0.5f);
                b += truncSat(two, (uint32_t)mx) != truncSat(one, (uint32_t)mx);
                ++n;
            }
            bad += b, tested += n;
        });
        printf("quantise by fma max=%u tested=%llu mismatches=%llu\n", (unsigned)mx, (unsigned long long)tested.load(), (unsigned long long)bad.load());
        failures += bad != 0;
    }

    const unsigned alphaMaxima[3] = { 255u, 1023u, 4095u };
    const uint32_t cFirst = full ? toBits(ldexpf(1.0f, -40)) : toBits(0.5f), cLast = toBits(1.0f);
    for (unsigned maxv : alphaMaxima) {
        std::atomic<uint64_t> bad { 0 }, badRcp { 0 }, tested { 0 };
        parallel(nt, [&](unsigned t) {
            uint64_t b = 0, br = 0, n = 0;
            for (unsigned a = 1 + t; a < maxv; a += nt) {
                const volatile float AcV = (float)a / (float)maxv;
                const float Ac = AcV;
                // the correctly rounded reciprocal: among the neighbours of the double-precision quotient's rounding, the one nearest to 1 / Ac
                float r = (float)(1.0 / (double)Ac);
                for (int d = -1; d <= 1; ++d) {
                    const float cand = fromBits((uint32_t)((int32_t)toBits((float)(1.0 / (double)Ac)) + d));
                    if (fabsl((long double)cand - 1.0L / (long double)Ac) < fabsl((long double)r - 1.0L / (long double)Ac))
                        r = cand;
                }
                for (int ulps = -2; ulps <= 2; ++ulps) {
                    const float r0 = fromBits((uint32_t)((int32_t)toBits(r) + ulps));
                    br += fmaf(fmaf(-Ac, r0, 1.0f), r0, r0) != r;
                }
                for (uint32_t bits = cFirst;; ++bits) {
                    const float c = (bits == cFirst && full) ? 0.0f : fromBits(bits); // (the full sweep starts with c = 0)
                    const float q0 = c * r;
                    const float q = fmaf(fmaf(-q0, Ac, c), r, q0);
                    const volatile float want = c / Ac;
                    b += q != want;
                    ++n;
                    if (bits == cLast)
                        break;
                }
            }
            bad += b, badRcp += br, tested += n;
        });
        printf("shared reciprocal max=%u reciprocal mismatches=%llu\n", maxv, (unsigned long long)badRcp.load());
        printf("shared reciprocal max=%u quotients tested=%llu mismatches=%llu\n", maxv, (unsigned long long)tested.load(), (unsigned long long)bad.load());
        failures += bad != 0 || badRcp != 0;
    }
    const unsigned pixelMaxima[4] = { 255u, 1023u, 4095u, 65535u };
    for (unsigned maxv : pixelMaxima) {
        const float maxF = (float)maxv;
        const unsigned cEnd = (maxv == 255u) ? 256u : 65536u; // 8-bit channels cannot hold more; 16-bit containers can hold any code
        std::atomic<uint64_t> bad { 0 }, badRcp { 0 }, tested { 0 };
        parallel(nt, [&](unsigned t) {
            uint64_t b = 0, br = 0, n = 0;
            for (unsigned a = 1 + t; a < maxv; a += nt) {
                const float af = (float)a;
                const float r = (float)(1.0 / (double)a); // a < 2^16: the double quotient rounds to the correctly rounded binary32 reciprocal
                for (int ulps = -2; ulps <= 2; ++ulps) {
                    const float r0 = fromBits((uint32_t)((int32_t)toBits(r) + ulps));
                    br += fmaf(fmaf(-af, r0, 1.0f), r0, r0) != r;
                }
                for (unsigned c = 0; c < cEnd; ++c, ++n) {
                    const volatile float xv = (float)c * maxF; // (volatile: one rounding per operation, as the reference is written)
                    const float x = xv;
                    const float q0 = x * r;
                    const float mine = floorf(fmaf(fmaf(-q0, af, x), r, q0) + 0.5f);
                    const volatile float quotient = x / af;
                    const float want = floorf(quotient + 0.5f);
                    b += (mine < maxF ? mine : maxF) != (want < maxF ? want : maxF);
                }
            }
            bad += b, badRcp += br, tested += n;
        });
        printf("integer unpremultiply max=%u reciprocal mismatches=%llu\n", maxv, (unsigned long long)badRcp.load());
        printf("integer unpremultiply max=%u tested=%llu mismatches=%llu\n", maxv, (unsigned long long)tested.load(), (unsigned long long)bad.load());
        failures += bad != 0 || badRcp != 0;
    }

    for (int ulps = -2; ulps <= 2; ++ulps) { // ARGBUnattenuate's reciprocals
        unsigned long long bad = 0, tested = 0;
        for (unsigned a = 1; a < 256; ++a, ++tested) {
            const float r = fromBits((uint32_t)((int32_t)toBits(1.0f / (float)a) + ulps));
            bad += avifhip::quotient65536ByEstimate(a, r) != 65536u / a;
        }
        printf("quotient65536 ulps=%+d tested=%llu mismatches=%llu\n", ulps, tested, bad);
        failures += bad != 0;
    }
    return failures ? 1 : 0;
}
