// gainmap_plan.h -- host side of gain-map application (avifRGBImageApplyGainMap, reference src/gainmap.c:73-315).
//
// The reference evaluates, per pixel and channel, two or three libm transcendentals on each side of a handful of fp32
// multiply-adds.  libm's powf / exp2f / logf are not correctly rounded, so no GPU implementation of them can promise the
// reference's bits.  What makes byte-exact output possible anyway:
//   * every transcendental on the INPUT side is a function of one integer sample (base sample -> linear light; gain-map
//     sample -> exp2f(lerp(min, max, powf(v, 1/gamma)) * weight)): tabulated here, on the host, with the host's libm --
//     the very values a libavif on this machine computes -- in O(2^depth) work per call;
//   * the OUTPUT side, code = quantise(clamp(linearToGamma(x))), is a monotone step function of the fp32 value x: its
//     steps (for each output code k the smallest x that quantises to >= k) are found here by bisection over the ordered
//     fp32 values, again with the host's libm, cached per (transfer function, depth, float?); the kernel locates x among
//     them by binary search.
// What is left for the GPU is IEEE fp32 / fp64 multiply-add arithmetic (exact, contraction off) and table lookups.
#pragma once

#include <stdint.h>

#include <vector>

namespace avifhip {

// gammaToLinear / linearToGamma of a transfer characteristic (src/colr.c:214-515), evaluated with the host's libm
float gainMapToLinear(int transferCharacteristics, float gamma);
float gainMapToGamma(int transferCharacteristics, float linear);

// avifColorPrimariesComputeRGBToRGBMatrix, src/colrconvert.c:163-179; false when a matrix is singular
bool gainMapPrimariesMatrix(int srcPrimaries, int dstPrimaries, double coeffs[9]);

// linear light of every sample code of an image: codes 0 .. 2^depth - 1 (v / max), or the 65536 half-float codes
std::vector<float> gainMapLinearLut(int transferCharacteristics, uint32_t depth, bool isFloat);

// exp2f(lerp(minLog2, maxLog2, powf(v / max, gammaInv)) * weight) for every sample code of the gain map, src/gainmap.c:253-254
std::vector<float> gainMapGainLut(uint32_t depth, float gammaInv, float minLog2, float maxLog2, float weight);

// Output steps of a transfer function, in two pieces of pieceEntries (a power of two >= maxCode + 1) entries (x < 0 first, then
// x >= 0; within each the quantised function is monotone; entries past maxCode are NaN): T[k], k = 1 .. maxCode, is the smallest fp32 x of the piece whose code
// quantise(nanSafeClamp(linearToGamma(x))) is >= k; +inf when no x of the piece reaches k.  T[0] = -inf.  Integer outputs: code =
// (uint)(0.5f + v * (2^depth - 1)), maxCode = 2^depth - 1; half-float outputs: code = bits(v * 2^-112) >> 13, maxCode =
// 0x3c00 (1.0).  Cached; the returned pointer stays valid for the life of the process.
struct GainMapSteps
{
    std::vector<float> steps;
    uint32_t maxCode = 0, pieceEntries = 0;
    // Where to start looking: for the x >= 0 piece, guide[b] is the code of the first fp32 value of bucket b, buckets being
    // runs of 2^kGainMapGuideShift consecutive fp32 bit patterns from 2^kGainMapGuideMinExp up (64 per octave); the code of an x
    // in bucket b lies in [guide[b], guide[b + 1]].  kGainMapGuideBuckets + 1 entries.
    std::vector<uint16_t> guide;
};
constexpr int kGainMapGuideShift = 17, kGainMapGuideMinExp = -24, kGainMapGuideOctaves = 40;
constexpr uint32_t kGainMapGuideBuckets = (uint32_t)kGainMapGuideOctaves << (23 - kGainMapGuideShift);
constexpr uint32_t kGainMapGuideFirstBits = (uint32_t)(127 + kGainMapGuideMinExp) << 23;
const GainMapSteps & gainMapOutputSteps(int transferCharacteristics, uint32_t depth, bool isFloat);

} // namespace avifhip
