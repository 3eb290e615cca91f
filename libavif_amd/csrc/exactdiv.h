// exactdiv.h -- divisors for which the reciprocal form  q0 = x*r; q = fma(fma(-q0, d, x), r, q0), r = RN(1/d)
// has been verified to equal IEEE-754 binary32 x / d over the kernels' whole input domain
// (tests/tools/verify_exact_division.c, run by tests/test_exact_division.py: every mantissa and sign for the
// kg divisors, every code point for the range divisors).  Anything not listed keeps the IEEE divide.
#pragma once

#include <stdint.h>
#include <string.h>

namespace avifhip {

inline uint32_t floatBits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// kg = 1 - kr - kb for every matrixCoefficients / colorPrimaries combination libavif accepts
// (src/colr.c:123-135 table, :517-542 primaries-derived).
inline bool verifiedKgDivisor(float kg)
{
    static const uint32_t kVerified[] = {
        0x3f161fb4u, 0x3f1645a1u, 0x3f170a3du, 0x3f2c18a0u, 0x3f2d9147u, 0x3f2d9169u, 0x3f2da76au, 0x3f3115c6u,
        0x3f3374bcu, 0x3f3378a8u, 0x3f34e753u, 0x3f37154au, 0x3f371759u, 0x3f38ba77u, 0x3f800000u,
    };
    const uint32_t b = floatBits(kg);
    for (uint32_t v : kVerified)
        if (v == b)
            return true;
    return false;
}

// rangeY / rangeUV: 219<<(d-8), 224<<(d-8) and (1<<d)-1 for d in {8, 10, 12, 16} (src/reformat.c:153-156)
inline bool verifiedRangeDivisor(float range)
{
    static const float kVerified[] = { 219.0f, 224.0f, 255.0f, 876.0f, 896.0f, 1023.0f, 3504.0f, 3584.0f, 4095.0f, 56064.0f, 57344.0f, 65535.0f };
    for (float v : kVerified)
        if (v == range)
            return true;
    return false;
}

} // namespace avifhip
