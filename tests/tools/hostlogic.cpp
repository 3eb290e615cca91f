// hostlogic.cpp -- test tool: C wrappers over the PRODUCT's host-only logic (libavif_amd/csrc/scale_plan.cpp, gainmap_plan.cpp:
// no HIP, no GPU), compiled with g++ into tests/tools/libhostlogic.so by tests/test_host_plans.py and compared with the oracle on
// the CPU.  What the kernels do with these tables is the GPU tests' business; that the tables are right is checked here.
#include <string.h>

#include "gainmap_plan.h"
#include "gainmap_steps.h"
#include "plan.h"
#include "scale_plan.h"

using namespace avifhip;

extern "C" {

// the rectangle a fused crop / rotate / mirror converts (plan.h coverOfCrop): out = {x, y, width, height}
void hostCoverOfCrop(uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, int quarterTurns, int mirrorAxis, uint64_t pixelsAddress, uint32_t pixelBytes, uint32_t out[4])
{
    const avifCropRect r = { cx, cy, cw, ch };
    const PixelMap map = makePixelMap(cx, cy, cw, ch, quarterTurns, mirrorAxis);
    const avifCropRect c = coverOfCrop(r, map, (uintptr_t)pixelsAddress, pixelBytes);
    out[0] = c.x, out[1] = c.y, out[2] = c.width, out[3] = c.height;
}

int hostScaleSchedule(int srcW, int srcH, int dstW, int dstH, int wide, int * colA, int * colB, int * rowA, int * rowB, int * rowF)
{
    const ScaleSchedule S = makeScaleSchedule(srcW, srcH, dstW, dstH, wide != 0);
    memcpy(colA, S.colA.data(), S.colA.size() * sizeof(int)), memcpy(colB, S.colB.data(), S.colB.size() * sizeof(int));
    memcpy(rowA, S.rowA.data(), S.rowA.size() * sizeof(int)), memcpy(rowB, S.rowB.data(), S.rowB.size() * sizeof(int));
    memcpy(rowF, S.rowF.data(), S.rowF.size() * sizeof(int));
    return S.mode;
}

// which specialised kernel the schedule qualifies for: bit 0 = doubling kernel (2x on both axes), bits 8.. = exact-box size N (0, 4, 8)
int hostScaleSpecialisation(int srcW, int srcH, int dstW, int dstH, int wide)
{
    const ScaleSchedule S = makeScaleSchedule(srcW, srcH, dstW, dstH, wide != 0);
    return (S.doubling ? 1 : 0) | (S.exactBox << 8);
}

float hostTransferFunction(int tc, int direction, float v)
{
    return direction ? gainMapToGamma(tc, v) : gainMapToLinear(tc, v);
}

int hostPrimariesMatrix(int src, int dst, double coeffs[9])
{
    return gainMapPrimariesMatrix(src, dst, coeffs) ? 1 : 0;
}

int hostDoubleToSignedFraction(double v, int32_t * n, uint32_t * d)
{
    return gainMapDoubleToFraction(v, n, d) ? 1 : 0;
}
int hostDoubleToUnsignedFraction(double v, uint32_t * n, uint32_t * d)
{
    return gainMapDoubleToUnsignedFraction(v, n, d) ? 1 : 0;
}

// output steps of a transfer function: copies up to `capacity` floats, returns the entries per piece; *maxCode receives the last code
uint32_t hostOutputSteps(int tc, uint32_t depth, int isFloat, float * steps, uint32_t capacity, uint32_t * maxCode)
{
    const GainMapSteps & S = gainMapOutputSteps(tc, depth, isFloat != 0);
    const size_t n = S.steps.size() < capacity ? S.steps.size() : capacity;
    memcpy(steps, S.steps.data(), n * sizeof(float));
    *maxCode = S.maxCode;
    return S.pieceEntries;
}

// the one-read locator of the fast apply kernel: codes of `count` values of x; returns the bucket count (0: this curve / depth has
// no locator and the general kernel serves it), *shift receives the bucket width in bits
uint32_t hostLocatorCodes(int tc, uint32_t depth, const float * x, uint32_t count, uint32_t * codes, uint32_t * shift)
{
    const GainMapSteps & S = gainMapOutputSteps(tc, depth, false);
    if (S.locator.empty())
        return 0;
    for (uint32_t k = 0; k < count; ++k)
        codes[k] = gainMapLocate(S, x[k]);
    *shift = S.locShift;
    return S.locBuckets;
}

int hostChooseMathPrimaries(int basePrimaries, int altPrimaries)
{
    int out = -1;
    return gainMapChooseMathPrimaries(basePrimaries, altPrimaries, &out) ? out : -1;
}

} // extern "C"

// ---- gain-map computation: the monotone-step tables against the formulas they tabulate (both evaluated here, on the host) ----
#include <math.h>

#include <random>

namespace {

uint32_t searchSteps(const std::vector<float> & T, uint32_t entries, float x) // what the kernels do: largest k with T[k] <= x
{
    uint32_t pos = 0;
    for (uint32_t s = entries >> 1; s; s >>= 1)
        pos += (T[pos + s] <= x) ? s : 0;
    return pos;
}
float clampRef(float v, float lo, float hi)
{
    return (v < lo) ? lo : ((hi < v) ? hi : v);
}
// a ratio between minR and maxR, log-uniform, snapped to nearby step boundaries now and then
float drawRatio(std::mt19937 & rng, float minR, float maxR, const std::vector<float> & T)
{
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    if (u(rng) < 0.3f) {
        const float t = T[1 + rng() % (T.size() - 1)];
        if (t == t && t >= minR && t <= maxR)
            return (u(rng) < 0.5f) ? t : nextafterf(t, (u(rng) < 0.5f) ? 0.0f : INFINITY);
    }
    return minR * powf(maxR / minR, u(rng));
}

} // namespace

extern "C" {

// mismatches between the bucket found through gainMapBucketSteps and avifValueToBucketIdx evaluated directly, over `samples` ratios
int hostCheckBucketSteps(float sign, float minRatio, float maxRatio, uint64_t numPixels, int samples, uint32_t seed, int * numBuckets)
{
    const GainMapChannelRange R = gainMapChannelRange(sign, minRatio, maxRatio, (size_t)numPixels);
    *numBuckets = R.numBuckets;
    if (R.numBuckets == 0)
        return 0;
    uint32_t entries = 0;
    const std::vector<float> T = gainMapBucketSteps(R, &entries);
    std::mt19937 rng(seed);
    int bad = 0;
    for (int k = 0; k < samples; ++k) {
        float r = drawRatio(rng, minRatio, maxRatio, T);
        r = clampRef(r, minRatio, maxRatio);
        const uint32_t m = searchSteps(T, entries, r);
        const int got = sign > 0 ? (int)m : R.numBuckets - 1 - (int)m;
        float v = clampRef(sign * log2f(r), R.lo, R.hi);
        int want = (int)floorf((v - R.lo) / (R.hi - R.lo) * R.numBuckets + 0.5f);
        want = want < R.numBuckets - 1 ? want : R.numBuckets - 1;
        bad += got != want;
    }
    return bad;
}

// the same for the final codes (gainMapCodeSteps vs src/gainmap.c:776-782 + avifSetRGBAPixel)
int hostCheckCodeSteps(float sign, float minRatio, float maxRatio, float minLog2, float maxLog2, float gamma, uint32_t depth, int samples, uint32_t seed)
{
    GainMapChannelRange R = gainMapChannelRange(sign, minRatio, maxRatio, 1000000);
    const std::vector<float> T = gainMapCodeSteps(R, minLog2, maxLog2, gamma, depth);
    const uint32_t entries = 1u << depth, maxCode = entries - 1;
    const float range = maxLog2 - minLog2;
    std::mt19937 rng(seed);
    int bad = 0;
    for (int k = 0; k < samples; ++k) {
        float r = clampRef(drawRatio(rng, minRatio, maxRatio, T), minRatio, maxRatio);
        const uint32_t m = searchSteps(T, entries, r);
        const uint32_t got = sign > 0 ? m : maxCode - m;
        float v = clampRef(sign * log2f(r), minLog2, maxLog2);
        v = powf((v - minLog2) / range, gamma);
        v = fminf(1.0f, fmaxf(0.0f, v));
        const uint32_t want = (uint32_t)(0.5f + v * (float)maxCode);
        bad += got != want;
    }
    return bad;
}

// The kernels' step search from a guessed index (gainmap_steps.h stepIndexFromGuess: four steps around the guess decide, the walks serve the
// rest) against the plain walks, on random monotone tables (equal steps, NaN padding behind `last`) with guesses near and far, NaN and
// infinite samples: mismatches over `trials` tables x 200 samples.
int hostCheckStepSearch(int trials, uint32_t seed)
{
    std::mt19937 rng(seed);
    auto draw = [&](uint32_t n) { return (uint32_t)(rng() % n); };
    int bad = 0;
    for (int t = 0; t < trials; ++t) {
        const uint32_t last = draw(40);
        std::vector<float> st(last + 4);
        st[0] = -INFINITY;
        float v = (float)draw(100) / 10.0f - 3.0f;
        for (uint32_t k = 1; k <= last; ++k) {
            v += draw(4) == 0 ? 0.0f : (float)draw(100) / 50.0f;
            st[k] = v;
        }
        for (uint32_t k = last + 1; k < last + 4; ++k)
            st[k] = NAN;
        for (int q = 0; q < 200; ++q) {
            const uint32_t kind = draw(20);
            const float x = kind == 0 ? NAN : kind == 1 ? INFINITY : kind == 2 ? -INFINITY : (kind < 8 && last > 0) ? st[1 + draw(last)] : (float)draw(10000) / 100.0f - 10.0f;
            const uint32_t m = draw(4) == 0 ? draw(last + 1) : (uint32_t)std::min<int64_t>(last, std::max<int64_t>(0, (int64_t)avifhip::stepIndexWalk(st.data(), last, 0, x) + (int)draw(7) - 3));
            bad += avifhip::stepIndexFromGuess(st.data(), last, m, x) != avifhip::stepIndexWalk(st.data(), last, m, x);
        }
    }
    return bad;
}

} // extern "C"


// ---- rebindYuvToRgbPlan against makeYuvToRgbPlan (plan.cpp): the tiles 1 .. N-1 of a batch get their plans by rebinding tile 0's; whatever
// plan derivation reads, the rebound plan must be the plan made from scratch, byte for byte (the batch tables are compared with memcmp) ----
namespace {

struct Job
{
    avifImage image;
    avifRGBImage rgb;
};

uint32_t pick(std::mt19937 & rng, std::initializer_list<uint32_t> v)
{
    return *(v.begin() + rng() % v.size());
}

// a random configuration; `other` = the same configuration over different buffers, pitches (same alignment class) and, when asked, one changed field
void drawJobs(std::mt19937 & rng, Job & a, Job & b, int mutate)
{
    memset(&a, 0, sizeof(a));
    avifImage & im = a.image;
    avifRGBImage & rgb = a.rgb;
    im.width = 64 + 8 * (rng() % 64), im.height = 16 + 2 * (rng() % 64);
    im.depth = pick(rng, { 8, 10, 12 });
    im.yuvFormat = (avifPixelFormat)pick(rng, { AVIF_PIXEL_FORMAT_YUV444, AVIF_PIXEL_FORMAT_YUV422, AVIF_PIXEL_FORMAT_YUV420, AVIF_PIXEL_FORMAT_YUV400 });
    im.yuvRange = (avifRange)pick(rng, { AVIF_RANGE_LIMITED, AVIF_RANGE_FULL });
    im.matrixCoefficients = (avifMatrixCoefficients)pick(rng, { 1, 5, 6, 9, 2, 7 });
    im.colorPrimaries = 1, im.transferCharacteristics = 13;
    im.alphaPremultiplied = rng() % 4 == 0;
    const bool alpha = rng() % 2;
    const uint32_t bps = im.depth > 8 ? 2 : 1, cw = (im.yuvFormat == AVIF_PIXEL_FORMAT_YUV444) ? im.width : (im.width + 1) / 2;
    const uint32_t pitchY = (im.width * bps + 255) & ~255u, pitchC = (cw * bps + 255) & ~255u;
    uintptr_t at = 0x10000000;
    for (int p = 0; p < 3; ++p) {
        if (p && im.yuvFormat == AVIF_PIXEL_FORMAT_YUV400)
            break;
        im.yuvPlanes[p] = (uint8_t *)at, im.yuvRowBytes[p] = p ? pitchC : pitchY;
        at += 0x1000000;
    }
    if (alpha)
        im.alphaPlane = (uint8_t *)at, im.alphaRowBytes = pitchY;
    rgb.width = im.width, rgb.height = im.height;
    rgb.depth = pick(rng, { 8, 8, 10, 12, 16 });
    rgb.format = (avifRGBFormat)pick(rng, { AVIF_RGB_FORMAT_RGB, AVIF_RGB_FORMAT_RGBA, AVIF_RGB_FORMAT_ARGB, AVIF_RGB_FORMAT_BGR, AVIF_RGB_FORMAT_BGRA, AVIF_RGB_FORMAT_ABGR });
    rgb.chromaUpsampling = (avifChromaUpsampling)pick(rng, { 0, 1, 2, 3, 4 });
    rgb.avoidLibYUV = rng() % 2, rgb.ignoreAlpha = rng() % 4 == 0, rgb.alphaPremultiplied = rng() % 3 == 0, rgb.isFloat = (rgb.depth == 16) && rng() % 2;
    rgb.maxThreads = 1;
    const uint32_t px = ((rgb.format == AVIF_RGB_FORMAT_RGB || rgb.format == AVIF_RGB_FORMAT_BGR) ? 3 : 4) * (rgb.depth > 8 ? 2 : 1);
    rgb.rowBytes = (im.width * px + 255) & ~255u;
    rgb.pixels = (uint8_t *)0x40000000;
    b = a;
    // the other tile: other addresses (same alignment), larger pitches
    for (int p = 0; p < 3; ++p)
        if (b.image.yuvPlanes[p])
            b.image.yuvPlanes[p] += 0x4000 * (1 + rng() % 7), b.image.yuvRowBytes[p] += 256 * (rng() % 3);
    if (b.image.alphaPlane)
        b.image.alphaPlane += 0x8000, b.image.alphaRowBytes += 256;
    b.rgb.pixels += 0x100000 * (1 + rng() % 5), b.rgb.rowBytes += 256 * (rng() % 2);
    switch (mutate) { // a difference a plan depends on: rebinding must refuse
        case 1: b.image.matrixCoefficients = (avifMatrixCoefficients)(im.matrixCoefficients == 1 ? 6 : 1); break;
        case 2: b.image.yuvRange = (avifRange)(im.yuvRange == AVIF_RANGE_FULL ? AVIF_RANGE_LIMITED : AVIF_RANGE_FULL); break;
        case 3: b.rgb.format = (avifRGBFormat)(rgb.format == AVIF_RGB_FORMAT_RGBA ? AVIF_RGB_FORMAT_BGRA : AVIF_RGB_FORMAT_RGBA); break;
        case 4: b.rgb.alphaPremultiplied = !rgb.alphaPremultiplied; break;
        case 5: b.image.depth = (im.depth == 8) ? 10 : 8; break;
        case 6: b.rgb.avoidLibYUV = !rgb.avoidLibYUV; break;
        case 7: b.image.height += 2, b.rgb.height += 2; break;
        case 8: b.rgb.chromaUpsampling = (avifChromaUpsampling)(rgb.chromaUpsampling == 3 ? 4 : 3); break;
        default: break;
    }
}

} // namespace

extern "C" {

// `count` random jobs: returns how many rebound plans differ from the plan made from scratch (must be 0); *refused counts the mutated jobs
// rebinding turned down (must equal *mutated: every one of those differences changes or may change the plan)
int hostCheckRebind(uint32_t seed, int count, int * refused, int * mutated, int * rebound)
{
    std::mt19937 rng(seed);
    int bad = 0;
    *refused = *mutated = *rebound = 0;
    for (int k = 0; k < count; ++k) {
        const int mutate = (rng() % 4 == 0) ? 1 + (int)(rng() % 8) : 0;
        Job a, b;
        drawJobs(rng, a, b, mutate);
        const int arith = (int)(rng() % 3);
        const uint32_t tuning = 0;
        avifCropRect rect = { 0, 0, b.image.width, b.image.height };
        const bool useRect = rng() % 2;
        if (useRect)
            rect.x = 8 * (rng() % 4), rect.y = 2 * (rng() % 4), rect.width -= rect.x + 8 * (rng() % 3), rect.height -= rect.y + 2 * (rng() % 3);
        YuvToRgbPlan proto, scratch, reboundPlan;
        if (makeYuvToRgbPlan(&a.image, &a.rgb, nullptr, arith, tuning, &proto) != AVIF_RESULT_OK)
            continue; // (an invalid combination: nothing to rebind)
        avifResult rr = AVIF_RESULT_OK;
        memset(&reboundPlan, 0xAB, sizeof(reboundPlan)); // padding bytes must come out of the rebind as the maker leaves them
        const bool did = rebindYuvToRgbPlan(proto, &a.image, &a.rgb, &b.image, &b.rgb, useRect ? &rect : nullptr, &reboundPlan, &rr);
        if (mutate) {
            ++*mutated;
            *refused += did ? 0 : 1;
            continue;
        }
        if (!did) {
            ++bad;
            continue;
        }
        ++*rebound;
        const avifResult mr = makeYuvToRgbPlan(&b.image, &b.rgb, useRect ? &rect : nullptr, arith, tuning, &scratch);
        if (mr != rr || (mr == AVIF_RESULT_OK && memcmp(&scratch, &reboundPlan, sizeof(scratch)) != 0))
            ++bad;
    }
    return bad;
}

} // extern "C"
