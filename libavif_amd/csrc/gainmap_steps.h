// The step search of the gain-map computation's histogram and quantiser kernels, from a guessed index: one function for the device
// (kernels_gainmap.hip) and for the host test that compares it with the plain walks on random tables (tests/tools/hostlogic.cpp
// hostCheckStepSearch, tests/test_host_plans.py) -- the decision is integer and comparison logic, identical wherever it is compiled.
#ifndef AVIFHIP_GAINMAP_STEPS_H
#define AVIFHIP_GAINMAP_STEPS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define AVIFHIP_STEPS_FN __host__ __device__ __forceinline__
#else
#define AVIFHIP_STEPS_FN inline
#endif

namespace avifhip {

// the walks of rounds 1-5: from any m to the largest k in [0, last] with steps[k] <= x (0 if there is none)
AVIFHIP_STEPS_FN uint32_t stepIndexWalk(const float * steps, uint32_t last, uint32_t m, float x)
{
    while (m < last && steps[m + 1] <= x)
        ++m;
    while (m > 0 && !(steps[m] <= x))
        --m;
    return m;
}

// The answer is the largest k in [0, last] with steps[k] <= x (0 if there is none; the steps are monotone).  Four steps around the guess m
// (0 <= m <= last), read side by side -- no loop whose trip count differs from lane to lane (round 6: the two correction walks, each
// iteration an LDS round trip inside divergent control flow, were most of the histogram and quantiser kernels' time) -- decide every guess
// that lies within one step below / two above the answer; the walks serve the rest (a bad guess, a NaN, an infinity).
AVIFHIP_STEPS_FN uint32_t stepIndexFromGuess(const float * steps, uint32_t last, uint32_t m, float x)
{
    const uint32_t i0 = m - (m != 0u ? 1u : 0u), i2 = (m + 1u < last) ? m + 1u : last, i3 = (m + 2u < last) ? m + 2u : last;
    const float s0 = steps[i0], s1 = steps[m], s2 = steps[i2], s3 = steps[i3];
    const bool c0 = m == 0u || s0 <= x, c1 = s1 <= x, c2 = m + 1u <= last && s2 <= x, c3 = m + 2u <= last && s3 <= x;
    if (__builtin_expect((c3 && m + 2u < last) || (!c0 && m >= 2u), 0))
        return stepIndexWalk(steps, last, m, x);
    const int k = (int)m - 1 + (c1 ? 1 : 0) + (c2 ? 1 : 0) + (c3 ? 1 : 0);
    return (uint32_t)(k < 0 ? 0 : k);
}

} // namespace avifhip

#endif
