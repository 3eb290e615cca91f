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

#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace avifhip {

// gammaToLinear / linearToGamma of a transfer characteristic (src/colr.c:214-515), evaluated with the host's libm
float gainMapToLinear(int transferCharacteristics, float gamma);
float gainMapToGamma(int transferCharacteristics, float linear);

// avifColorPrimariesComputeRGBToRGBMatrix, src/colrconvert.c:163-179; false when a matrix is singular
bool gainMapPrimariesMatrix(int srcPrimaries, int dstPrimaries, double coeffs[9]);

// linear light of every sample code of an image: codes 0 .. 2^depth - 1 (v / max), or the 65536 half-float codes
const std::vector<float> & gainMapLinearLut(int transferCharacteristics, uint32_t depth, bool isFloat); // (cached per process)

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
    // One-read locator over the x >= 0 piece (integer outputs up to 12 bits whose x < 0 piece is all code 0 -- every curve but
    // BT.1361 / IEC 61966-2-4): buckets of 2^locShift consecutive fp32 bit patterns from locFirstBits up, narrow enough that
    // no bucket holds two steps.  With t = clamp((int)bits(x), (int)locFirstBits, (int)locFirstBits + (locBuckets << locShift) - 1) - locFirstBits and
    // e = locator[t >> locShift], the code of x is (e & 0xfff) + ((t << (32 - locShift)) > e): the low 12 bits of e are the code
    // at the start of the bucket, the high locShift bits the offset of the bucket's step minus one (all ones: no step); locShift <= 16
    // for codes of more than 8 bits, so that bits 12-15 of e are clear (the kernel packs such codes by their low 16 bits).
    // Negative x (sign bit: a negative integer) and x below the first step clamp into bucket 0 ahead of its step: code 0; x past the last
    // step, +inf and NaN clamp into the last bucket, which holds no step.  Empty when the curve / depth does not qualify.
    std::vector<uint32_t> locator;
    uint32_t locFirstBits = 0, locShift = 0, locBuckets = 0;
};
constexpr uint32_t kGainMapLocatorMaxBuckets = 12288; // 48 KB of LDS
// the code the locator gives for x (what the fast kernel computes); S.locator must not be empty
uint32_t gainMapLocate(const GainMapSteps & S, float x);
constexpr int kGainMapGuideShift = 17, kGainMapGuideMinExp = -24, kGainMapGuideOctaves = 40;
constexpr uint32_t kGainMapGuideBuckets = (uint32_t)kGainMapGuideOctaves << (23 - kGainMapGuideShift);
constexpr uint32_t kGainMapGuideFirstBits = (uint32_t)(127 + kGainMapGuideMinExp) << 23;
const GainMapSteps & gainMapOutputSteps(int transferCharacteristics, uint32_t depth, bool isFloat);

} // namespace avifhip

// ---- gain-map computation (avifRGBImageComputeGainMap, reference src/gainmap.c:535-843), host side ------------------------------
// The same idea as for application: the GPU never evaluates log2f / powf.  The kernels carry the exact fp32 ratio r of every
// sample ((alt + offset) / (base + offset), floored at 1e-10); the gain-map value sign * log2f(r), its histogram bucket and its
// final quantised code are monotone step functions of r, whose steps the host finds by bisection with its own libm.
namespace avifhip {

// avifChooseColorSpaceForGainMapMath, src/gainmap.c:496-533; false: a primaries matrix is singular
bool gainMapChooseMathPrimaries(int basePrimaries, int altPrimaries, int * mathPrimaries);
// avifColorPrimariesComputeYCoeffs, src/colr.c:517-542
void gainMapYCoefficients(int primaries, float coeffs[3]);
// avifDoubleToSignedFraction / avifDoubleToUnsignedFraction, src/utils.c:238-299
bool gainMapDoubleToFraction(double v, int32_t * n, uint32_t * d);
bool gainMapDoubleToUnsignedFraction(double v, uint32_t * n, uint32_t * d);

// One channel's gain-map values as a function of the ratio: value(r) = sign * log2f(r) (sign -1 when the alternate image has the
// smaller headroom, src/gainmap.c:728-739).
struct GainMapChannelRange
{
    float sign = 1.0f;
    float minRatio = 0.0f, maxRatio = 0.0f; // extreme ratios over the image (from the GPU reduction)
    float lo = 0.0f, hi = 0.0f;             // extreme gain-map values: value(minRatio), value(maxRatio) in value order
    int numBuckets = 0;                     // 0: no histogram needed (avifFindMinMaxWithoutOutliers returns [lo, hi], :392-394)
    int maxOutliersOnEachSide = 0;
};
// the first half of avifFindMinMaxWithoutOutliers (:375-399): extremes, outlier budget, bucket count
GainMapChannelRange gainMapChannelRange(float sign, float minRatio, float maxRatio, size_t numPixels);
// Steps of the MONOTONE bucket index m(r) over r in [minRatio, maxRatio]: m = bucket for sign > 0, numBuckets - 1 - bucket for sign
// < 0.  entries (a power of two >= numBuckets) floats: T[0] = -inf, T[i] = smallest r with m(r) >= i (+inf if none), NaN padding.
std::vector<float> gainMapBucketSteps(const GainMapChannelRange & range, uint32_t * entries);
// the second half of avifFindMinMaxWithoutOutliers (:403-425) on a histogram indexed by BUCKET (not m)
void gainMapRangeWithoutOutliers(const GainMapChannelRange & range, const uint32_t * histogram, float * rangeMin, float * rangeMax);
// Steps of the monotone final code m(r): code for sign > 0, maxCode - code for sign < 0, where code = (uint)(0.5f + clamp01(powf((clamp(
// value(r), minLog2, maxLog2) - minLog2) / range, gamma)) * maxCode) (src/gainmap.c:762-787, src/reformat.c:1920-1937); entries =
// 2^depth.  rangeIsZero: every code is 0 (:766-773).
std::vector<float> gainMapCodeSteps(const GainMapChannelRange & range, float minLog2, float maxLog2, float gamma, uint32_t depth);

} // namespace avifhip
