// exactdiv.h -- division by plan constants without the IEEE divide sequence.
//
// For a constant d let  hi = RN(1/d)  and  lo = RN(1/d - hi)  (splitReciprocal).  Then
//        q = fma(x, hi, x * lo)
// differs from the exact quotient by a relative 2^-47 before its single final rounding, so it equals the
// correctly rounded IEEE-754 binary32 quotient x / d unless x / d lies that close to a rounding boundary.
// That is NOT guaranteed in general; it is established per divisor by exhaustive enumeration
// (tests/tools/verify_exact_division.cpp, run by tests/test_exact_division.py):
//   * verifiedKgDivisor       every mantissa and both signs of x (the identity is invariant under scaling x by
//                             powers of two as long as x * lo stays normal, which holds for every operand the
//                             kernels form);
//   * verifiedIntegerDivisor  the same full-mantissa sweep, plus every integer x in [-65536, 65536]
//                             (code points minus bias).
//   * verifiedChromaDenominator  2*(1-kb) and 2*(1-kr) of the encode direction (src/reformat.c:384-385): the
//                             full-mantissa sweep (the dividends B-Y, R-Y are arbitrary fp32 values in [-1, 1]).
// Divisors that are not listed keep the IEEE divide (the universal kernels).
#pragma once

#include <stdint.h>
#include <string.h>

namespace avifhip {

struct RcpSplit
{
    float hi, lo;
};

// scale / d (scale a power of two) as RN(scale / d) and the rounded remainder
inline RcpSplit splitReciprocal(float d, float scale)
{
    RcpSplit r = { 0.0f, 0.0f };
    if (d != 0.0f) {
        r.hi = scale * (1.0f / d);
        r.lo = (float)((double)scale / (double)d - (double)r.hi);
    }
    return r;
}

inline uint32_t floatBits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// kg = 1 - kr - kb for every matrixCoefficients / colorPrimaries combination libavif accepts
// (src/colr.c:123-135 table, :517-542 primaries-derived).
static const uint32_t kVerifiedKgBits[] = {
    0x3f161fb4u, 0x3f1645a1u, 0x3f170a3du, 0x3f2c18a0u, 0x3f2d9147u, 0x3f2d9169u, 0x3f2da76au, 0x3f3115c6u,
    0x3f3374bcu, 0x3f3378a8u, 0x3f34e753u, 0x3f37154au, 0x3f371759u, 0x3f38ba77u, 0x3f800000u,
};
inline bool verifiedKgDivisor(float kg)
{
    const uint32_t b = floatBits(kg);
    for (uint32_t v : kVerifiedKgBits)
        if (v == b)
            return true;
    return false;
}

// rangeY / rangeUV: 219<<(d-8), 224<<(d-8); channel maxima (1<<d)-1; d in {8, 10, 12, 16}
// (src/reformat.c:153-156, :49, src/alpha.c:93,189)
static const float kVerifiedIntegerDivisors[] = { 219.0f, 224.0f, 255.0f, 876.0f, 896.0f, 1023.0f, 3504.0f, 3584.0f, 4095.0f, 56064.0f, 57344.0f, 65535.0f };
inline bool verifiedIntegerDivisor(float d)
{
    for (float v : kVerifiedIntegerDivisors)
        if (v == d)
            return true;
    return false;
}

// 2*(1-kb), 2*(1-kr) for every matrixCoefficients / colorPrimaries combination libavif accepts (same enumeration as kg),
// EXCEPT 0x3fee5bb7 (1.86217391, a primaries-derived value): the enumeration finds 6 operands for which the reciprocal
// form is off by one ulp, so plans with that divisor keep the IEEE divide.
static const uint32_t kVerifiedChromaDenBits[] = {
    0x3fb33333u, 0x3fb374bcu, 0x3fb376ecu, 0x3fbcbfadu, 0x3fbcbfb2u, 0x3fbf1507u, 0x3fc4abfeu, 0x3fc561ebu, 0x3fc72ab9u, 0x3fc9907cu,
    0x3fc9930cu, 0x3fc9a1b3u, 0x3fc9ba5eu, 0x3fca5ec0u, 0x3fe2a8c8u, 0x3fe2d0e5u, 0x3fe3d70au, 0x3fe76ca2u, 0x3fe9ba5eu, 0x3fe9d6f5u,
    0x3febb3dbu, 0x3fed844du, 0x3fed84ceu, 0x3fedbc9au, 0x3fee9263u, 0x3ff0d19au, 0x3ff0d1b7u, 0x40000000u,
};
inline bool verifiedChromaDenominator(float d)
{
    const uint32_t b = floatBits(d);
    for (uint32_t v : kVerifiedChromaDenBits)
        if (v == b)
            return true;
    return false;
}

// floor(65536 / a) for 0 < a < 256 (ARGBUnattenuate's table of reciprocals) from an estimate r of 1 / a good to a
// few ulp and one correction step with the exact remainder -- instead of the 32-bit integer division sequence.  Enumerated for every a with
// the estimate off by up to 2 ulp (tests/tools/verify_unpremultiply_integer.cpp).
constexpr unsigned quotient65536ByEstimate(unsigned a, float r)
{
    const unsigned q0 = (unsigned)(65536.0f * r);
    const int rem = (int)(65536u - (q0 & 0xffffffu) * (a & 0xffu));
    return rem < 0 ? q0 - 1u : ((unsigned)rem >= a ? q0 + 1u : q0);
}

} // namespace avifhip
