// api_gainmap.cpp -- C ABI entry points for gain maps (include/avifhip.h): avifhipRGBImageApplyGainMap[Async],
// avifhipImageApplyGainMap, avifhipRGBImageComputeGainMap.  Host logic mirrors reference src/gainmap.c; tables come from
// gainmap_plan.cpp, kernels from kernels_gainmap.hip.
#include "api_internal.h"

#if defined(__GLIBC__)
#include <malloc.h>
#endif

#include <chrono>

using namespace avifhip;
using namespace avifhip::api;

// =================================================================================================
// gain-map application, reference src/gainmap.c:73-355
// =================================================================================================

namespace {
// ---- light levels ----
std::atomic<int> gExactLightLevels { -1 }; // -1: AVIFHIP_EXACT_LIGHT_LEVELS decides (read once)
bool exactLightLevels()
{
    int v = gExactLightLevels.load(std::memory_order_relaxed);
    if (v < 0) {
        const char * e = getenv("AVIFHIP_EXACT_LIGHT_LEVELS");
        v = (e && e[0] == '1' && !e[1]) ? 1 : 0;
        gExactLightLevels.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}
uint16_t lightLevelNits(float v) // src/gainmap.c:303,305
{
    const float r = floorf(v * 203.0f + 0.5f);
    return (uint16_t)((r < 0.0f) ? 0.0f : ((65535.0f < r) ? 65535.0f : r));
}
} // namespace

namespace {

// AVIFHIP_GAINMAP_KERNEL=general keeps calls off the fast apply kernel (tests run both kernels over the same cases)
bool fastKernelDisabled()
{
    const char * e = getenv("AVIFHIP_GAINMAP_KERNEL");
    return e && strcmp(e, "general") == 0;
}

// AVIFHIP_GAINMAP_TRACE: the phases of the host-resident gain-map calls on stderr, milliseconds since the call began (where a call's time goes:
// tests/tools/gm_call_bench.py)
struct PhaseTrace
{
    bool on;
    std::chrono::steady_clock::time_point start;
    explicit PhaseTrace(const char * what) : on(getenv("AVIFHIP_GAINMAP_TRACE") != nullptr), start(std::chrono::steady_clock::now())
    {
        if (on)
            fprintf(stderr, "avifhip %s:", what);
    }
    void mark(const char * phase)
    {
        if (on)
            fprintf(stderr, " %s %.3f", phase, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count());
    }
    ~PhaseTrace()
    {
        if (on)
            fprintf(stderr, "\n");
    }
};

// AVIFHIP_GAINMAP_PLANES=0: the gain map is converted by a launch of its own into an RGBA copy (rounds 2-4) even where the fast apply kernel
// could read its planes (tests and A/B measurements run both)

// tls.gainMapPartials, pinned: the apply kernel's statistics (kGainMapMaxGroups partials of 16 bytes) and what the computation fetches between
// its passes -- kGainMapMaxGroups x 8 floats of partials, at most 3 x 10 000 histogram counters (round 6: those downloads went into pageable
// vectors, through the runtime's staging copy)
static constexpr size_t kGainMapPinnedBytes = (size_t)kGainMapMaxGroups * 16 * sizeof(float); // (pass 1's partials and pass 0's behind them)
static_assert(kGainMapPinnedBytes >= (size_t)kGainMapMaxGroups * sizeof(GainMapPartial) && kGainMapPinnedBytes >= (size_t)3 * 10240 * sizeof(uint32_t), "pinned buffer size");
bool gainPlanesDisabled()
{
    const char * e = getenv("AVIFHIP_GAINMAP_PLANES");
    return e && strcmp(e, "0") == 0;
}

// The fast apply kernel converts the gain map's pixels itself when the map's own avifImageYUVToRGB (src/gainmap.c:185-212) is a function of
// one pixel's samples that it carries (kernels.h: GainMapPlaneConversion): 8-bit 4:4:4 or 4:0:0, no pending alpha arithmetic, libyuv's 8-bit
// entries or the fp32 loops with the matrix-coefficient / identity / YCgCo transforms and divisors on the verified list.
bool gainPlaneConversionOf(const YuvToRgbPlan & p, GainMapPlaneConversion * K)
{
    const YuvSide & s = p.yuv;
    memset(K, 0, sizeof(*K));
    if (s.depth != 8 || s.chanBytes != 1 || (s.format != AVIF_PIXEL_FORMAT_YUV444 && s.format != AVIF_PIXEL_FORMAT_YUV400))
        return false;
    if (p.inLoopMul != MUL_NONE || p.postMul != MUL_NONE || p.rgb.isFloat || p.rgb.is565 || p.rgb.isGray || p.rgb.depth != 8 || p.rgb.map.on)
        return false;
    if (s.format == AVIF_PIXEL_FORMAT_YUV444 && (!s.hasColor || !s.plane[1] || !s.plane[2] || s.rowBytes[1] != s.rowBytes[2]))
        return false;
    K->hasColor = s.hasColor ? 1 : 0;
    if (p.identityCopy) {
        K->identityCopy = 1;
        return s.hasColor != 0;
    }
    if (p.arith == ARITH_LIBYUV) {
        if (p.fxNative != 8 || p.fxDownshift != 0 || (p.fxMono != 0) != (s.hasColor == 0))
            return false;
        K->fixedPoint = 1, K->fx = p.fx;
        return true;
    }
    if (!s.exactDiv || (s.hasColor && s.mode != MODE_COEFF && s.mode != MODE_IDENTITY && s.mode != MODE_YCGCO))
        return false;
    K->mode = s.mode;
    K->biasY = s.biasY, K->rangeY = s.rangeY, K->biasUV = s.biasUV, K->rangeUV = s.rangeUV;
    K->twoOneMinusKr = s.twoOneMinusKr, K->twoOneMinusKb = s.twoOneMinusKb, K->krOneMinusKr = s.krOneMinusKr, K->kbOneMinusKb = s.kbOneMinusKb;
    K->rcpKgTimes2 = s.rcpKgTimes2;
    return true;
}

void diagClear(avifDiagnostics * diag)
{
    if (diag)
        diag->error[0] = '\0';
}
void diagPrintf(avifDiagnostics * diag, const char * fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    char text[AVIF_DIAGNOSTICS_ERROR_BUFFER_SIZE];
    vsnprintf(text, sizeof(text), fmt, ap);
    va_end(ap);
    if (diag)
        memcpy(diag->error, text, sizeof(text));
    setError("%s", text);
}

inline float fractionToFloat(avifSignedFraction f) // src/gainmap.c:32-38
{
    return f.d == 0 ? 0.0f : (float)f.n / f.d;
}
inline float fractionToFloat(avifUnsignedFraction f) // :40-46
{
    return f.d == 0 ? 0.0f : (float)f.n / f.d;
}

avifResult gainMapValidateMetadata(const avifGainMap * gainMap, avifDiagnostics * diag) // :430-457
{
    for (int i = 0; i < 3; ++i) {
        if (gainMap->gainMapMin[i].d == 0 || gainMap->gainMapMax[i].d == 0 || gainMap->gainMapGamma[i].d == 0 || gainMap->baseOffset[i].d == 0 ||
            gainMap->alternateOffset[i].d == 0) {
            diagPrintf(diag, "Per-channel denominator is 0 in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        if ((int64_t)gainMap->gainMapMax[i].n * gainMap->gainMapMin[i].d < (int64_t)gainMap->gainMapMin[i].n * gainMap->gainMapMax[i].d) {
            diagPrintf(diag, "Per-channel max is less than per-channel min in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        if (gainMap->gainMapGamma[i].n == 0) {
            diagPrintf(diag, "Per-channel gamma is 0 in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
    }
    if (gainMap->baseHdrHeadroom.d == 0 || gainMap->alternateHdrHeadroom.d == 0) {
        diagPrintf(diag, "Headroom denominator is 0 in gain map metadata");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    if (gainMap->useBaseColorSpace != 0 && gainMap->useBaseColorSpace != 1) {
        diagPrintf(diag, "useBaseColorSpace is %d in gain map metadata", gainMap->useBaseColorSpace);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    return AVIF_RESULT_OK;
}

float gainMapWeight(float hdrHeadroom, const avifGainMap * gainMap) // avifGetGainMapWeight, :52-63
{
    const float base = fractionToFloat(gainMap->baseHdrHeadroom), alternate = fractionToFloat(gainMap->alternateHdrHeadroom);
    if (base == alternate)
        return 0.0f;
    const float r = (hdrHeadroom - base) / (alternate - base);
    const float w = (r < 0.0f) ? 0.0f : ((1.0f < r) ? 1.0f : r);
    return (alternate < base) ? -w : w;
}

bool gainMapLayout(const avifRGBImage * rgb, GainMapPixelLayout * L) // avifGetRGBColorSpaceInfo, src/reformat.c:32-117
{
    if (rgb->depth != 8 && rgb->depth != 10 && rgb->depth != 12 && rgb->depth != 16)
        return false;
    if ((rgb->isFloat && rgb->depth != 16) || (rgb->format == AVIF_RGB_FORMAT_RGB_565 && rgb->depth != 8))
        return false;
    memset(L, 0, sizeof(*L));
    const uint32_t cb = (rgb->depth > 8) ? 2 : 1;
    L->channelBytes = cb;
    uint32_t n = 0;
    switch (rgb->format) {
        case AVIF_RGB_FORMAT_RGB: L->offR = 0, L->offG = cb, L->offB = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_RGBA: L->offR = 0, L->offG = cb, L->offB = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ARGB: L->offA = 0, L->offR = cb, L->offG = 2 * cb, L->offB = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_BGR: L->offB = 0, L->offG = cb, L->offR = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_BGRA: L->offB = 0, L->offG = cb, L->offR = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ABGR: L->offA = 0, L->offB = cb, L->offG = 2 * cb, L->offR = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_RGB_565: L->is565 = 1, n = 2; break;
        default: return false; // gray layouts have no R, G, B offsets for the tone-mapping loop to index
    }
    L->pixelBytes = L->is565 ? 2 : n * cb;
    L->hasAlpha = (n == 4 && !L->is565) ? 1 : 0;
    L->isFloat = rgb->isFloat ? 1 : 0;
    L->depth = rgb->depth;
    L->maxF = (float)((1u << rgb->depth) - 1);
    return true;
}

// The tone-mapping of device-resident images.  `gainImage`: gainMap->image with device plane pointers.  The tone-mapped
// image must already own device pixels of the base image's size.  Waits for the stream: the result code and the CLLI
// values depend on the pixels.
avifResult applyGainMapOnDevice(const avifRGBImage * base, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC, const avifGainMap * gainMap,
                                const avifImage * gainImage, float weight, avifColorPrimaries outPrimaries, avifTransferCharacteristics outTC,
                                avifRGBImage * out, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag, hipStream_t stream, bool mayReturnEarly = false)
{
    const uint32_t width = base->width, height = base->height;
    GainMapArgs A;
    memset(&A, 0, sizeof(A));
    if (!gainMapLayout(base, &A.baseL) || !gainMapLayout(out, &A.outL)) {
        diagPrintf(diag, "Unsupported RGB color space");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    A.base = base->pixels, A.basePitch = base->rowBytes, A.out = out->pixels, A.outPitch = out->rowBytes;
    A.width = width, A.height = height;

    const avifColorPrimaries mathPrimaries =
        (gainMap->useBaseColorSpace || (gainMap->altColorPrimaries == AVIF_COLOR_PRIMARIES_UNSPECIFIED)) ? basePrimaries : gainMap->altColorPrimaries;
    const bool applyGain = weight != 0.0f;
    if (!applyGain) { // "Just convert from one rgb format to another", src/gainmap.c:142-170
        const bool primariesDiffer = basePrimaries != outPrimaries;
        if (primariesDiffer && !gainMapPrimariesMatrix(basePrimaries, outPrimaries, A.inM)) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
        A.inConv = primariesDiffer ? 1 : 0;
        A.convert = (outTC != baseTC || primariesDiffer) ? 1 : 0;
    } else {
        A.convert = 1;
        A.inConv = (basePrimaries != mathPrimaries) ? 1 : 0, A.outConv = (mathPrimaries != outPrimaries) ? 1 : 0;
        if ((A.inConv && !gainMapPrimariesMatrix(basePrimaries, mathPrimaries, A.inM)) ||
            (A.outConv && !gainMapPrimariesMatrix(mathPrimaries, outPrimaries, A.outM))) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
    }

    // ---- the gain map as RGB at the base image's size, :185-212 ----
    uint32_t gainDepth = 8;
    avifImage gm;
    memset(&gm, 0, sizeof(gm));
    if (applyGain) {
        memcpy(&gm, gainImage, sizeof(avifImage));
        if (gm.width != width || gm.height != height) {
            avifImage scaled;
            memcpy(&scaled, &gm, sizeof(avifImage));
            scaled.width = width, scaled.height = height;
            const PlaneDims dd = planeDims(width, height, (int)gm.yuvFormat);
            const size_t bps = (gm.depth > 8) ? 2 : 1;
            size_t offset[4] = { 0, 0, 0, 0 }, total = 0;
            uint32_t pitch[4] = { 0, 0, 0, 0 };
            for (int p = 0; p < 4; ++p) {
                const uint8_t * sp = (p < 3) ? gm.yuvPlanes[p] : gm.alphaPlane;
                if (!sp || ((p == 1 || p == 2) && gm.yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
                    continue;
                pitch[p] = alignUp((uint32_t)(dd.w[p] * bps), 256);
                offset[p] = total, total += (size_t)pitch[p] * dd.h[p];
            }
            const avifResult rr = reserve(tls.gainMap[4], total ? total : 1);
            if (rr != AVIF_RESULT_OK)
                return rr;
            for (int p = 0; p < 4; ++p) {
                uint8_t * dp = pitch[p] ? (uint8_t *)tls.gainMap[4].ptr + offset[p] : nullptr;
                if (p < 3)
                    scaled.yuvPlanes[p] = dp, scaled.yuvRowBytes[p] = pitch[p];
                else
                    scaled.alphaPlane = dp, scaled.alphaRowBytes = pitch[p];
            }
            const avifResult sr = avifhipImageScaleAsync(&gm, &scaled, stream);
            if (sr != AVIF_RESULT_OK)
                return sr;
            memcpy(&gm, &scaled, sizeof(avifImage));
        }
        A.gainDepth = gainDepth = gm.depth; // (where the samples come from is decided below, once the kernel is known)
        for (int c = 0; c < 3; ++c)
            A.baseOffset[c] = fractionToFloat(gainMap->baseOffset[c]), A.altOffset[c] = fractionToFloat(gainMap->alternateOffset[c]);
    }

    // ---- tables (kept while the parameters stay the same: sequences of frames, tiles) ----
    if (A.convert) {
        GainMapTableCache & cache = tls.gainMapCache;
        GainMapTableCache::Key key;
        memset(&key, 0, sizeof(key));
        key.baseTC = baseTC, key.baseDepth = base->depth, key.baseFloat = base->isFloat ? 1 : 0;
        key.outTC = outTC, key.outDepth = A.outL.is565 ? 8 : out->depth, key.outFloat = out->isFloat ? 1 : 0;
        key.gainDepth = gainDepth, key.applyGain = applyGain ? 1 : 0, key.stream = (uint64_t)(uintptr_t)stream;
        if (applyGain) {
            for (int c = 0; c < 3; ++c) {
                key.gammaInv[c] = 1.0f / fractionToFloat(gainMap->gainMapGamma[c]);
                key.minLog2[c] = fractionToFloat(gainMap->gainMapMin[c]), key.maxLog2[c] = fractionToFloat(gainMap->gainMapMax[c]);
            }
            key.weight = weight;
        }
        if (!cache.valid || memcmp(&cache.key, &key, sizeof(key)) != 0) {
            cache.valid = false;
            std::vector<float> tables = gainMapLinearLut(baseTC, base->depth, base->isFloat != 0);
            cache.baseLutOffset = 0, cache.gainLutOffset = tables.size();
            if (applyGain) {
                for (int c = 0; c < 3; ++c) {
                    const std::vector<float> g = gainMapGainLut(gainDepth, key.gammaInv[c], key.minLog2[c], key.maxLog2[c], weight);
                    tables.insert(tables.end(), g.begin(), g.end());
                }
            }
            // largest magnitudes of the tables (NaN / inf: infinity), for the bound that lets the fast kernel skip its NaN test
            auto largest = [](const float * v, size_t n) {
                float m = 0.0f;
                for (size_t k = 0; k < n; ++k)
                    m = (fabsf(v[k]) <= m) ? m : ((v[k] == v[k]) ? fabsf(v[k]) : INFINITY);
                return m;
            };
            cache.baseMax = largest(tables.data(), cache.gainLutOffset);
            for (int c = 0; c < 3; ++c)
                cache.gainMax[c] = applyGain ? largest(tables.data() + cache.gainLutOffset + (size_t)c * ((size_t)1 << gainDepth), (size_t)1 << gainDepth) : 0.0f;
            cache.stepsOffset = tables.size();
            const GainMapSteps & S = gainMapOutputSteps(outTC, key.outDepth, out->isFloat != 0);
            tables.insert(tables.end(), S.steps.begin(), S.steps.end());
            cache.guideOffset = tables.size();
            tables.resize(tables.size() + (S.guide.size() + 1) / 2, 0.0f); // the 16-bit guide entries ride in float slots
            memcpy(tables.data() + cache.guideOffset, S.guide.data(), S.guide.size() * sizeof(uint16_t));
            cache.maxCode = S.maxCode, cache.stepEntries = S.pieceEntries;
            // the fast kernel's tables: the one-read locator (when the curve and depth have one) and the output alpha code of
            // every base alpha code, (T)(0.5f + (a / max) * max') of avifGetRGBAPixel + avifSetRGBAPixel (src/reformat.c:1856-1937)
            // (both 16-byte aligned and padded: the kernel moves them 16 bytes at a time)
            tables.resize((tables.size() + 3) & ~(size_t)3, 0.0f);
            cache.locOffset = tables.size(), cache.locBuckets = (uint32_t)S.locator.size();
            cache.locFirstBits = S.locFirstBits, cache.locShift = S.locShift;
            tables.resize(tables.size() + S.locator.size(), 0.0f);
            if (!S.locator.empty())
                memcpy(tables.data() + cache.locOffset, S.locator.data(), S.locator.size() * sizeof(uint32_t));
            tables.resize((tables.size() + 3) & ~(size_t)3, 0.0f);
            cache.alphaOffset = tables.size();
            if (!base->isFloat && !out->isFloat && base->depth <= 12) {
                const uint32_t n = 1u << base->depth;
                std::vector<uint16_t> alpha(n);
                const float baseMaxF = (float)(n - 1), outMaxF = (float)((1u << key.outDepth) - 1);
                for (uint32_t a = 0; a < n; ++a)
                    alpha[a] = (uint16_t)((uint32_t)(int32_t)(0.5f + (((float)a / baseMaxF) * outMaxF)) & ((key.outDepth > 8) ? 0xffffu : 0xffu));
                tables.resize(tables.size() + n / 2, 0.0f);
                memcpy(tables.data() + cache.alphaOffset, alpha.data(), n * sizeof(uint16_t));
            }
            tables.resize((tables.size() + 3) & ~(size_t)3, 0.0f);
            // what the reference computes for a NaN input (the weight-0 path can meet one in a half-float base image)
            const float nanGamma = fminf(1.0f, fmaxf(0.0f, gainMapToGamma(outTC, NAN)));
            cache.nanCode = out->isFloat ? ((uint32_t)0) : (uint32_t)(0.5f + nanGamma * (float)((1u << key.outDepth) - 1));
            if (out->isFloat) {
                const float f = nanGamma * 1.9259299444e-34f;
                uint32_t u;
                memcpy(&u, &f, 4);
                cache.nanCode = (u >> 13) & 0xffffu;
            }
            const avifResult rr = reserve(tls.gainMap[2], tables.size() * sizeof(float));
            if (rr != AVIF_RESULT_OK)
                return rr;
            const avifResult ur = uploadTableAsync(tls.gainMap[2].ptr, tables.data(), tables.size() * sizeof(float), stream);
            if (ur != AVIF_RESULT_OK)
                return ur;
            cache.key = key;
            cache.valid = true;
        }
        const float * t = (const float *)tls.gainMap[2].ptr;
        A.baseLut = t + cache.baseLutOffset, A.gainLut = t + cache.gainLutOffset, A.steps = t + cache.stepsOffset;
        A.maxCode = cache.maxCode, A.nanCode = cache.nanCode, A.stepEntries = cache.stepEntries;
        A.guide = (const uint16_t *)(t + cache.guideOffset);
        A.guideFirstBits = kGainMapGuideFirstBits, A.guideShift = kGainMapGuideShift, A.guideBuckets = kGainMapGuideBuckets;
        // the kernel keeps the tables in LDS when all of them fit (up to 12-bit images; 16-bit and half-float tables stay in
        // global memory)
        const size_t stepsEntries = 2 * (size_t)cache.stepEntries, baseEntries = cache.gainLutOffset - cache.baseLutOffset,
                     gainEntries = cache.stepsOffset - cache.gainLutOffset;
        if ((stepsEntries + baseEntries + gainEntries) * sizeof(float) + (kGainMapGuideBuckets + 2) * sizeof(uint16_t) <= 64 * 1024)
            A.ldsSteps = (uint32_t)stepsEntries, A.ldsBaseLut = (uint32_t)baseEntries, A.ldsGainLut = (uint32_t)gainEntries;
        // ... or, where the output curve has a locator, the base and gain tables and the locator: no search at all (the general kernel's
        // own use of what the fast kernel below is built on; the weight-0 paths have no gain table)
        if (cache.locBuckets && !A.baseL.isFloat && (baseEntries + gainEntries + cache.locBuckets) * sizeof(float) <= 64 * 1024 - 256 && !fastKernelDisabled()) {
            A.ldsLocator = 1, A.ldsSteps = 0;
            A.ldsBaseLut = (uint32_t)baseEntries, A.ldsGainLut = (uint32_t)gainEntries;
        }
        // the fast kernel: 4-channel integer pixels at naturally aligned addresses on both sides, a gain map, a locator, and
        // tables that fit the LDS (kernels_gainmap.hip)
        auto plain4 = [](const GainMapPixelLayout & L, const void * pixels, uint32_t pitch) {
            return !L.isFloat && !L.is565 && L.hasAlpha && L.depth <= 12 && (L.pixelBytes == 4 || L.pixelBytes == 8) &&
                   (((uintptr_t)pixels | pitch) & (L.pixelBytes - 1)) == 0;
        };
        A.locator = (const uint32_t *)(t + cache.locOffset), A.alphaLut = (const uint16_t *)(t + cache.alphaOffset);
        A.locFirstBits = cache.locFirstBits, A.locShift = cache.locShift, A.locBuckets = cache.locBuckets;
        A.fast = applyGain && cache.locBuckets && width >= 4 && plain4(A.baseL, A.base, A.basePitch) && plain4(A.outL, A.out, A.outPitch) && gainDepth <= 12 &&
                 gainMapFastLdsBytes(A.baseL.pixelBytes, gainDepth, cache.locBuckets) <= kGainMapFastLdsBytes && !fastKernelDisabled();
        if (A.fast) {
            // in place (the device entry point with the tone-mapped pixels on top of the base pixels, same layout): the fast kernel's lanes take runs
            // of 4 pixels and the last run of a row is shifted left to stay inside it -- over pixels a neighbouring lane, possibly of another
            // workgroup, has already replaced.  The general kernel, one lane per pixel, reads every pixel exactly once before it writes it.
            const uint8_t * b0 = A.base, * b1 = b0 + (size_t)A.basePitch * height;
            const uint8_t * o0 = A.out, * o1 = o0 + (size_t)A.outPitch * height;
            if (b0 < o1 && o0 < b1)
                A.fast = 0;
        }
        if (A.fast) {
            // no intermediate value can reach FLT_MAX (so none is infinite, so none is a NaN) when the tables are finite and the
            // products of their largest entries with the coefficients stay far below it: only then may the fast kernel, which does
            // not look for NaNs, serve the call
            auto rowSum = [](const double M[9]) {
                double m = 0.0;
                for (int r = 0; r < 3; ++r)
                    m = fmax(m, fabs(M[3 * r]) + fabs(M[3 * r + 1]) + fabs(M[3 * r + 2]));
                return m;
            };
            double bound = cache.baseMax;
            if (A.inConv)
                bound *= rowSum(A.inM) * 1.001;
            double tone = 0.0;
            for (int c = 0; c < 3; ++c)
                tone = fmax(tone, (bound + fabs((double)A.baseOffset[c])) * cache.gainMax[c] * 1.001 + fabs((double)A.altOffset[c]));
            if (A.outConv)
                tone *= rowSum(A.outM) * 1.001;
            if (!(bound < 1e37 && tone < 1e37)) // (a NaN or an infinity on the way fails the comparisons too)
                A.fast = 0;                     // the fast kernel has no NaN test: the general one serves this call
        }
        if (A.fast) {
            // selectors of the byte permutations between the pixels' layouts and R, G, B, A order (v_perm_b32: selector byte k names the
            // source byte that lands in byte k of the result; 0-3 = second operand, 4-7 = first)
            const uint32_t bo[4] = { A.baseL.offR, A.baseL.offG, A.baseL.offB, A.baseL.offA }, oo[4] = { A.outL.offR, A.outL.offG, A.outL.offB, A.outL.offA };
            if (A.baseL.pixelBytes == 4) {
                A.selBase[0] = bo[0] | (bo[1] << 8) | (bo[2] << 16) | (bo[3] << 24), A.selBase[1] = 0;
            } else {
                A.selBase[0] = bo[0] | ((bo[0] + 1) << 8) | (bo[1] << 16) | ((bo[1] + 1) << 24);
                A.selBase[1] = bo[2] | ((bo[2] + 1) << 8) | (bo[3] << 16) | ((bo[3] + 1) << 24);
            }
            A.selOut[0] = A.selOut[1] = 0;
            if (A.outL.pixelBytes == 4) {
                const uint32_t from[4] = { 0, 1, 4, 5 }; // (c0 | c1 << 8) is the second operand, (c2 | alpha << 8) the first
                for (int ch = 0; ch < 4; ++ch)
                    A.selOut[0] |= from[ch] << (8 * oo[ch]);
            } else {
                for (int ch = 0; ch < 4; ++ch) // (c0 | c1 << 16) second operand, (c2 | alpha << 16) first: channel ch sits at bytes 2 ch, 2 ch + 1
                    for (uint32_t b = 0; b < 2; ++b) {
                        const uint32_t place = oo[ch] + b;
                        A.selOut[place >> 2] |= (2 * (uint32_t)ch + b) << (8 * (place & 3));
                    }
            }
        }
    }

    // ---- the gain map's samples: its planes, converted by the fast kernel pixel by pixel, or an RGBA copy made by the conversion kernels ----
    bool gainFromPlanes = false;
    if (applyGain) {
        avifRGBImage rgbGain; // avifRGBImageSetDefaults, src/avif.c:700-717
        memset(&rgbGain, 0, sizeof(rgbGain));
        rgbGain.width = width, rgbGain.height = height, rgbGain.depth = gm.depth, rgbGain.format = AVIF_RGB_FORMAT_RGBA;
        rgbGain.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgbGain.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
        rgbGain.maxThreads = 1;
        rgbGain.rowBytes = alignUp(width * 4 * ((gm.depth > 8) ? 2 : 1), 256);
        if (A.fast && gm.depth == 8 && (gm.yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || gm.yuvFormat == AVIF_PIXEL_FORMAT_YUV400) && gm.yuvPlanes[0] &&
            gainMapFastLdsBytes(A.baseL.pixelBytes, gainDepth, A.locBuckets, true) <= kGainMapFastLdsBytes && !gainPlanesDisabled()) {
            YuvToRgbPlan plan;
            rgbGain.pixels = gm.yuvPlanes[0]; // (a plan wants a destination; nothing is converted into it)
            if (makeYuvToRgbPlan(&gm, &rgbGain, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan) == AVIF_RESULT_OK &&
                gainPlaneConversionOf(plan, &A.gainConv)) {
                gainFromPlanes = true;
                A.gainPlanes = 1;
                A.gain = gm.yuvPlanes[0], A.gainPitch = gm.yuvRowBytes[0];
                A.gainU = A.gainConv.hasColor ? gm.yuvPlanes[1] : nullptr, A.gainV = A.gainConv.hasColor ? gm.yuvPlanes[2] : nullptr;
                A.gainPitchUV = A.gainConv.hasColor ? gm.yuvRowBytes[1] : 0;
            }
            rgbGain.pixels = nullptr;
        }
        if (!gainFromPlanes) {
            const avifResult rr = reserve(tls.gainMap[1], (size_t)rgbGain.rowBytes * height);
            if (rr != AVIF_RESULT_OK)
                return rr;
            rgbGain.pixels = (uint8_t *)tls.gainMap[1].ptr;
            const avifResult cr = avifhipImageYUVToRGBAsync(&gm, &rgbGain, stream);
            if (cr != AVIF_RESULT_OK)
                return cr;
            A.gain = rgbGain.pixels, A.gainPitch = rgbGain.rowBytes;
        }
    }

    // statistics: every workgroup stores its partial into device scratch; a copy behind the kernel brings them into pinned host memory, where they
    // are added up, in index order, once the stream has drained.  (Until round 6 the kernel stored them straight into the pinned buffer;
    // AVIFHIP_GM_PARTIALS=host keeps that route for experiments.)
    static const bool partialsOnHost = [] {
        const char * e = getenv("AVIFHIP_GM_PARTIALS");
        return e && !strcmp(e, "host");
    }();
    if (!tls.gainMapPartials) {
        HIP_TRY(hipHostMalloc(&tls.gainMapPartials, kGainMapPinnedBytes, hipHostMallocDefault));
        AVIFHIP_NEW_HOST_MEMORY(tls.gainMapPartials, kGainMapPinnedBytes);
    }
    if (partialsOnHost) {
        A.partials = (GainMapPartial *)tls.gainMapPartials;
    } else {
        const avifResult pr = reserve(tls.gainMap[3], (size_t)kGainMapMaxGroups * 8 * sizeof(float)); // (the computation's partials share the buffer)
        if (pr != AVIF_RESULT_OK)
            return pr;
        A.partials = (GainMapPartial *)tls.gainMap[3].ptr;
    }
    // exact light levels (opt-in): the kernel also leaves every pixel's max(0, r, g, b); the host adds them up like the reference does
    const bool exactLevels = applyGain && clli && exactLightLevels() && tls.gainMapTimeIters <= 0;
    if (exactLevels) {
        const avifResult xr = reserve(tls.gainMap[11], (size_t)width * height * sizeof(float));
        if (xr != AVIF_RESULT_OK)
            return xr;
        A.pixelMax = (float *)tls.gainMap[11].ptr;
    }
    uint32_t partials = 0;
    if (tls.gainMapTimeIters > 0) { // avifhipTimeRGBImageApplyGainMap: the apply kernel alone, back to back, between two events
        for (int k = 0; k < tls.gainMapTimeWarmup; ++k)
            HIP_TRY(launchGainMapApply(A, stream, &partials));
        hipEvent_t t0 = nullptr, t1 = nullptr;
        HIP_TRY(hipEventCreate(&t0));
        if (hipEventCreate(&t1) != hipSuccess) {
            (void)hipEventDestroy(t0);
            return hipFailed(hipGetLastError(), "hipEventCreate");
        }
        (void)hipEventRecord(t0, stream);
        hipError_t timed = hipSuccess;
        for (int k = 0; k < tls.gainMapTimeIters - 1 && timed == hipSuccess; ++k)
            timed = launchGainMapApply(A, stream, &partials);
        (void)hipEventRecord(t1, stream); // (the last of the iters launches is the call's own, below)
        float ms = -1.0f;
        if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
            ms = -1.0f;
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        if (timed != hipSuccess)
            return hipFailed(timed, "gain map kernel launch (timed loop)");
        tls.gainMapTimedMs = (ms < 0 || tls.gainMapTimeIters < 2) ? -1.0 : (double)ms / (tls.gainMapTimeIters - 1);
    }
    const hipError_t e = launchGainMapApply(A, stream, &partials);
    if (e != hipSuccess)
        return hipFailed(e, "gain map kernel launch");
    tls.lastKernel = applyGain ? (A.fast ? (gainFromPlanes ? "gainmap_apply_fast<planes>" : "gainmap_apply_fast") : "gainmap_apply") : (A.convert ? "gainmap_convert" : "gainmap_requantise");
    ++tls.launches;
    // Nothing of the answer depends on the pixels when the caller wants no light levels and the fast kernel serves the call: its precondition
    // (above) is that no NaN can arise, and the result code is then AVIF_RESULT_OK whatever the pixels hold.  The asynchronous entry point
    // returns with its work enqueued, like every other Async call.
    if (mayReturnEarly && applyGain && A.fast && !clli && tls.gainMapTimeIters <= 0)
        return AVIF_RESULT_OK;
    // ... and with light levels asked for, the asynchronous entry point still returns with its work enqueued: the partials travel into a pinned
    // slot behind the kernel, and the thread's next avifhipSynchronize on that stream turns them into *clli (settleLightLevels; the fast
    // kernel's precondition rules the NaN result out, so the result code is known now)
    if (mayReturnEarly && applyGain && A.fast && clli && !exactLevels && !partialsOnHost && partials && tls.gainMapTimeIters <= 0) {
        const uint32_t slot = tls.lightSlot++ % (uint32_t)Context::kLightSlots;
        if (!tls.lightPinned[slot]) {
            HIP_TRY(hipHostMalloc(&tls.lightPinned[slot], (size_t)kGainMapMaxGroups * sizeof(GainMapPartial), hipHostMallocDefault));
            AVIFHIP_NEW_HOST_MEMORY(tls.lightPinned[slot], (size_t)kGainMapMaxGroups * sizeof(GainMapPartial));
        }
        if (!tls.lightCopied[slot])
            HIP_TRY(hipEventCreateWithFlags(&tls.lightCopied[slot], hipEventDisableTiming));
        if (tls.lightPending[slot].pending) { // the call of kLightSlots calls ago has not been settled: its clli is filled now (waits for its copy)
            const avifResult sr = settleLightLevels(nullptr, true, true);
            if (sr != AVIF_RESULT_OK)
                return sr;
        }
        HIP_TRY(hipMemcpyAsync(tls.lightPinned[slot], A.partials, (size_t)partials * sizeof(GainMapPartial), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(tls.lightCopied[slot], stream));
        Context::PendingLight & P = tls.lightPending[slot];
        P.pending = true, P.count = partials, P.pixels = (size_t)width * height, P.clli = clli, P.stream = stream;
        return AVIF_RESULT_OK;
    }
    const GainMapPartial * hostPartials = (const GainMapPartial *)tls.gainMapPartials;
    std::vector<float> pixelMaxima;
    if (exactLevels) {
        pixelMaxima.resize((size_t)width * height);
        HIP_TRY(hipMemcpyAsync(pixelMaxima.data(), A.pixelMax, pixelMaxima.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    }
    if (!partialsOnHost && partials)
        HIP_TRY(hipMemcpyAsync(tls.gainMapPartials, A.partials, (size_t)partials * sizeof(GainMapPartial), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    GainMapStats stats = { 0, 0, 0.0 };
    {
        float rgbMax = 0.0f;
        for (uint32_t k = 0; k < partials; ++k) {
            const GainMapPartial & part = hostPartials[k];
            rgbMax = (part.max > rgbMax) ? part.max : rgbMax;
            stats.sum += part.sum;
            stats.nan |= (int32_t)part.nan;
        }
        memcpy(&stats.maxBits, &rgbMax, 4);
    }
    if (applyGain && stats.nan) {
        diagPrintf(diag, "Degenerate gain map parameters produce NaN");
        return AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE;
    }
    if (applyGain && clli) { // src/gainmap.c:292-302 (the reference sums in fp32 pixel by pixel; here fp64 partial sums)
        float rgbMaxLinear;
        memcpy(&rgbMaxLinear, &stats.maxBits, 4);
        clli->maxCLL = lightLevelNits(rgbMaxLinear);
        if (exactLevels) {
            // src/gainmap.c:223,293,304: ONE fp32 accumulator over the pixels in raster order -- at 8 megapixels the sum is past 2^23 and every
            // addition rounds, so only the same additions in the same order give the same average (8 ms of host time for a 4K image)
            float rgbSumLinear = 0.0f;
            for (const float m : pixelMaxima)
                rgbSumLinear += m;
            clli->maxPALL = lightLevelNits(rgbSumLinear / (float)((size_t)width * height));
        } else {
            clli->maxPALL = lightLevelNits((float)stats.sum / (float)((size_t)width * height));
        }
    }
    return AVIF_RESULT_OK;
}

// argument checks shared by the entry points, src/gainmap.c:86-94
avifResult gainMapCheckArguments(const avifRGBImage * base, const avifGainMap * gainMap, float hdrHeadroom, const avifRGBImage * out, avifDiagnostics * diag)
{
    diagClear(diag);
    if (hdrHeadroom < 0.0f) {
        diagPrintf(diag, "hdrHeadroom should be >= 0, got %f", hdrHeadroom);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    if (base == NULL || gainMap == NULL || out == NULL) {
        diagPrintf(diag, "NULL input image");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    return gainMapValidateMetadata(gainMap, diag);
}

bool gainMapIsPlainCopy(const avifRGBImage * base, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC, float weight,
                        avifColorPrimaries outPrimaries, avifTransferCharacteristics outTC, const avifRGBImage * out) // :120-128
{
    return weight == 0.0f && outTC == baseTC && outPrimaries == basePrimaries && base->format == out->format && base->depth == out->depth &&
           base->isFloat == out->isFloat && base->rowBytes == out->rowBytes;
}

} // namespace

extern "C" avifResult avifhipRGBImageApplyGainMapAsync(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                                       avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap,
                                                       float hdrHeadroom, avifColorPrimaries outputColorPrimaries,
                                                       avifTransferCharacteristics outputTransferCharacteristics, avifRGBImage * toneMappedImage,
                                                       avifContentLightLevelInformationBox * clli, avifDiagnostics * diag, void * hipStream)
{
    const avifResult ar = gainMapCheckArguments(baseImage, gainMap, hdrHeadroom, toneMappedImage, diag);
    if (ar != AVIF_RESULT_OK)
        return ar;
    if (!baseImage->pixels || !toneMappedImage->pixels || !toneMappedImage->rowBytes || !gainMap->image) {
        diagPrintf(diag, "avifhipRGBImageApplyGainMapAsync: device-resident base, gain map and tone-mapped pixels are required");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    ScratchScope scratch(stream); // tables and work buffers are per thread, not per stream
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    toneMappedImage->width = baseImage->width, toneMappedImage->height = baseImage->height;
    const float weight = gainMapWeight(hdrHeadroom, gainMap);
    if (gainMapIsPlainCopy(baseImage, baseColorPrimaries, baseTransferCharacteristics, weight, outputColorPrimaries, outputTransferCharacteristics,
                           toneMappedImage)) {
        HIP_TRY(hipMemcpyAsync(toneMappedImage->pixels, baseImage->pixels, (size_t)baseImage->rowBytes * baseImage->height, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return AVIF_RESULT_OK;
    }
    return applyGainMapOnDevice(baseImage, baseColorPrimaries, baseTransferCharacteristics, gainMap, gainMap->image, weight, outputColorPrimaries,
                                outputTransferCharacteristics, toneMappedImage, clli, diag, stream, /*mayReturnEarly=*/true);
}

// the apply kernel of that call alone: milliseconds per launch over back-to-back launches (HIP events on the launch stream)
extern "C" double avifhipTimeRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                                  avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                                  avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                                  avifRGBImage * toneMappedImage, int warmup, int iters, void * hipStream)
{
    if (iters < 2 || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    tls.gainMapTimeWarmup = warmup < 0 ? 0 : warmup, tls.gainMapTimeIters = iters, tls.gainMapTimedMs = -1.0;
    const avifResult r = avifhipRGBImageApplyGainMapAsync(baseImage, baseColorPrimaries, baseTransferCharacteristics, gainMap, hdrHeadroom, outputColorPrimaries,
                                                          outputTransferCharacteristics, toneMappedImage, nullptr, nullptr, hipStream);
    tls.gainMapTimeIters = 0;
    return (r == AVIF_RESULT_OK) ? tls.gainMapTimedMs : -1.0;
}

// How many bytes a block the CALLER's image owns can take, for the "a buffer of the right size stays" shortcuts below: known only where the C
// library can be asked (glibc, whose allocator libavif's avifAlloc and this library's malloc share in an ordinary build); 0 -- "unknown: release
// and allocate, as the reference always does" -- anywhere else, and under AVIFHIP_KEEP_BUFFERS=0 for processes whose avifAlloc is not malloc
// (malloc_usable_size on a pointer of another allocator is undefined).
static size_t ownedBlockCapacity(const void * block)
{
#if defined(__GLIBC__)
    static const bool keep = [] {
        const char * e = getenv("AVIFHIP_KEEP_BUFFERS");
        return !(e && !strcmp(e, "0"));
    }();
    return (block && keep) ? malloc_usable_size(const_cast<void *>(block)) : 0;
#else
    (void)block;
    return 0;
#endif
}

// host-resident images, like the reference: the tone-mapped image's pixels are (re)allocated with malloc (src/gainmap.c:112-114).
// baseOnDevice: the base pixels are a device buffer of this call's own (avifhipImageApplyGainMap below: the YUV base image converted straight
// into HBM, rows padded to 256 bytes) standing for the tightly packed host image the reference allocates there.
static avifResult applyGainMapToHostImage(const avifRGBImage * baseImage, bool baseOnDevice, avifColorPrimaries baseColorPrimaries,
                                          avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                          avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                          avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag)
{
    const avifResult ar = gainMapCheckArguments(baseImage, gainMap, hdrHeadroom, toneMappedImage, diag);
    if (ar != AVIF_RESULT_OK)
        return ar;
    const uint32_t width = baseImage->width, height = baseImage->height;
    toneMappedImage->width = width, toneMappedImage->height = height;
    // avifRGBImageAllocatePixels, src/avif.c:719-737 (frees what the image holds, allocates width x height pixels).  A buffer that already has
    // that size stays (round 5): the caller cannot tell, and a buffer the runtime pinned for the last call's download costs 3.8 ms to release
    // and another 2.6 ms of page faults under this call's download when it is 66 MB (a 4K RGBA10 image: 8.7 -> 2.5 ms per call for callers
    // that tone-map a sequence into one avifRGBImage).
    const uint32_t outPixelBytes = rgbPixelBytes(toneMappedImage);
    if (!width || !height || width > UINT32_MAX / outPixelBytes) {
        free(toneMappedImage->pixels);
        toneMappedImage->pixels = NULL, toneMappedImage->rowBytes = 0;
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const uint32_t outRowBytes = width * outPixelBytes;
    const size_t outBytes = (size_t)outRowBytes * height;
    const size_t have = ownedBlockCapacity(toneMappedImage->pixels);
    if (have < outBytes || have > outBytes + outBytes / 8 + 4096) {
        free(toneMappedImage->pixels);
        toneMappedImage->pixels = NULL, toneMappedImage->rowBytes = 0;
        toneMappedImage->pixels = (uint8_t *)malloc(outBytes);
        if (!toneMappedImage->pixels)
            return AVIF_RESULT_OUT_OF_MEMORY;
    }
    toneMappedImage->rowBytes = outRowBytes;

    PhaseTrace trace("apply gain map");
    trace.mark("pixels allocated");
    const float weight = gainMapWeight(hdrHeadroom, gainMap);
    const uint32_t baseWidthBytes = width * rgbPixelBytes(baseImage);
    avifRGBImage asTheReferenceSeesIt; // (the row pitch decides whether the call is a plain copy, :120-128)
    memcpy(&asTheReferenceSeesIt, baseImage, sizeof(avifRGBImage));
    if (baseOnDevice)
        asTheReferenceSeesIt.rowBytes = baseWidthBytes;
    if (gainMapIsPlainCopy(&asTheReferenceSeesIt, baseColorPrimaries, baseTransferCharacteristics, weight, outputColorPrimaries, outputTransferCharacteristics,
                           toneMappedImage)) {
        if (baseOnDevice) {
            HIP_TRY(hipMemcpy2DAsync(toneMappedImage->pixels, outRowBytes, baseImage->pixels, baseImage->rowBytes, baseWidthBytes, height, hipMemcpyDeviceToHost, tls.stream));
            HIP_TRY(hipStreamSynchronize(tls.stream));
        } else {
            memcpy(toneMappedImage->pixels, baseImage->pixels, (size_t)baseImage->rowBytes * baseImage->height); // "Copy the base image", :124-127
        }
        return AVIF_RESULT_OK;
    }
    if (!baseImage->pixels || (weight != 0.0f && !gainMap->image))
        return AVIF_RESULT_INVALID_ARGUMENT;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    // the work buffers and tables below are the thread's, not a stream's: an asynchronous application this thread enqueued on ANOTHER stream
    // (which since round 5 may return with its kernel pending) must have finished reading them before this call rewrites them
    ScratchScope scratch(tls.stream);
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    QuiesceOnExit quiesceOnExit; // (an unsupported colour space, a NaN ... is found after the uploads were enqueued)
    // device copies: base pixels, gain map planes, tone-mapped pixels
    avifRGBImage baseView, outView;
    memcpy(&baseView, baseImage, sizeof(avifRGBImage));
    memcpy(&outView, toneMappedImage, sizeof(avifRGBImage));
    avifResult r = AVIF_RESULT_OK;
    if (!baseOnDevice) {
        baseView.rowBytes = alignUp(baseWidthBytes, 256);
        r = reserve(tls.gainMap[5], (size_t)baseView.rowBytes * height);
        if (r != AVIF_RESULT_OK)
            return r;
        baseView.pixels = (uint8_t *)tls.gainMap[5].ptr;
        HIP_TRY(hipMemcpy2DAsync(baseView.pixels, baseView.rowBytes, baseImage->pixels, baseImage->rowBytes, baseWidthBytes, height, hipMemcpyHostToDevice, tls.stream));
    }
    outView.rowBytes = alignUp(outRowBytes, 256);
    r = reserve(tls.gainMap[0], (size_t)outView.rowBytes * height);
    if (r != AVIF_RESULT_OK)
        return r;
    outView.pixels = (uint8_t *)tls.gainMap[0].ptr;
    avifImage gainView;
    memset(&gainView, 0, sizeof(gainView));
    if (weight != 0.0f) {
        memcpy(&gainView, gainMap->image, sizeof(avifImage));
        r = stagePlanes(&gainView, true, false);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    trace.mark("uploads");
    r = applyGainMapOnDevice(&baseView, baseColorPrimaries, baseTransferCharacteristics, gainMap, &gainView, weight, outputColorPrimaries,
                             outputTransferCharacteristics, &outView, clli, diag, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;
    trace.mark("applied");
    HIP_TRY(hipMemcpy2DAsync(toneMappedImage->pixels, outRowBytes, outView.pixels, outView.rowBytes, outRowBytes, height, hipMemcpyDeviceToHost, tls.stream));
    HIP_TRY(hipStreamSynchronize(tls.stream));
    trace.mark("downloaded");
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                                  avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                                  avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                                  avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag)
{
    return applyGainMapToHostImage(baseImage, false, baseColorPrimaries, baseTransferCharacteristics, gainMap, hdrHeadroom, outputColorPrimaries,
                                   outputTransferCharacteristics, toneMappedImage, clli, diag);
}

// avifImageApplyGainMap, src/gainmap.c:317-355: the base image arrives as YUV.  The reference converts it into an RGB image of the API's defaults
// on the heap and hands that to avifRGBImageApplyGainMap; here that image lives in HBM only (round 5: it used to travel to the host and back --
// 8 bytes per pixel over the link for nothing): planes up, conversion, application, tone-mapped pixels down.
extern "C" avifResult avifhipImageApplyGainMap(const avifImage * baseImage, const avifGainMap * gainMap, float hdrHeadroom,
                                               avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                               avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag)
{
    diagClear(diag);
    if (!baseImage || !gainMap)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseImage->avifhipOpaqueIcc_[1] > 0 || gainMap->altICC.size > 0) { // icc.size, include/avifhip/avif_abi.h; :328-331
        diagPrintf(diag, "Tone mapping for images with ICC profiles is not supported");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    avifRGBImage baseRgb; // avifRGBImageSetDefaults + avifRGBImageAllocatePixels, :333-335
    memset(&baseRgb, 0, sizeof(baseRgb));
    baseRgb.width = baseImage->width, baseRgb.height = baseImage->height, baseRgb.depth = baseImage->depth, baseRgb.format = AVIF_RGB_FORMAT_RGBA;
    baseRgb.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, baseRgb.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    baseRgb.maxThreads = 1;
    const uint32_t pixelBytes = rgbPixelBytes(&baseRgb);
    if (!baseRgb.width || !baseRgb.height || baseRgb.width > UINT32_MAX / pixelBytes)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    ScratchScope scratch(tls.stream); // (the base pixels go into the thread's gain-map scratch: see applyGainMapToHostImage)
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    baseRgb.rowBytes = alignUp(baseRgb.width * pixelBytes, 256);
    avifResult r = reserve(tls.gainMap[5], (size_t)baseRgb.rowBytes * baseRgb.height);
    if (r != AVIF_RESULT_OK)
        return r;
    baseRgb.pixels = (uint8_t *)tls.gainMap[5].ptr;
    r = avifhipImageYUVToRGB(baseImage, &baseRgb); // host planes (or device ones) into the device buffer
    if (r != AVIF_RESULT_OK)
        return r;
    return applyGainMapToHostImage(&baseRgb, true, baseImage->colorPrimaries, baseImage->transferCharacteristics, gainMap, hdrHeadroom, outputColorPrimaries,
                                   outputTransferCharacteristics, toneMappedImage, clli, diag);
}

// ---- gain-map computation (the encode side), reference src/gainmap.c:535-843 ----

namespace {

// device planes (Y, U, V, A) of a wxh image in one scratch buffer, 256-byte row pitch
avifResult deviceGainMapPlanes(avifImage * view, uint32_t width, uint32_t height, Scratch & scratch)
{
    view->width = width, view->height = height;
    const PlaneDims d = planeDims(width, height, (int)view->yuvFormat);
    const size_t bps = (view->depth > 8) ? 2 : 1;
    size_t offset[4], total = 0;
    uint32_t pitch[4];
    for (int p = 0; p < 4; ++p) {
        const bool present = !((p == 1 || p == 2) && view->yuvFormat == AVIF_PIXEL_FORMAT_YUV400);
        pitch[p] = present ? alignUp((uint32_t)(d.w[p] * bps), 256) : 0;
        offset[p] = total, total += (size_t)pitch[p] * d.h[p];
    }
    const avifResult r = reserve(scratch, total);
    if (r != AVIF_RESULT_OK)
        return r;
    for (int p = 0; p < 4; ++p) {
        uint8_t * ptr = pitch[p] ? (uint8_t *)scratch.ptr + offset[p] : nullptr;
        if (p < 3)
            view->yuvPlanes[p] = ptr, view->yuvRowBytes[p] = pitch[p];
        else
            view->alphaPlane = ptr, view->alphaRowBytes = pitch[p];
    }
    return AVIF_RESULT_OK;
}

void freeHostPlanes(avifImage * image) // avifImageFreePlanes(AVIF_PLANES_ALL), src/avif.c:492-517
{
    if (image->imageOwnsYUVPlanes)
        for (int p = 0; p < 3; ++p)
            free(image->yuvPlanes[p]);
    for (int p = 0; p < 3; ++p)
        image->yuvPlanes[p] = NULL, image->yuvRowBytes[p] = 0;
    image->imageOwnsYUVPlanes = AVIF_FALSE;
    if (image->imageOwnsAlphaPlane)
        free(image->alphaPlane);
    image->alphaPlane = NULL, image->alphaRowBytes = 0, image->imageOwnsAlphaPlane = AVIF_FALSE;
}

} // namespace

// Host images in, gain-map metadata and (malloc'ed) gain-map planes out, like the reference; or (deviceResident) everything in device memory:
// the two renditions' pixels, and the planes of gainMap->image, which the caller allocated at the requested size.
static avifResult computeGainMapImpl(const avifRGBImage * baseRgbImage, avifColorPrimaries baseColorPrimaries,
                                     avifTransferCharacteristics baseTransferCharacteristics, const avifRGBImage * altRgbImage,
                                     avifColorPrimaries altColorPrimaries, avifTransferCharacteristics altTransferCharacteristics,
                                     avifGainMap * gainMap, avifDiagnostics * diag, bool deviceResident, void * hipStream)
{
    diagClear(diag);
    if (baseRgbImage == NULL || altRgbImage == NULL || gainMap == NULL || gainMap->image == NULL)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseRgbImage->width != altRgbImage->width || baseRgbImage->height != altRgbImage->height) {
        diagPrintf(diag, "Both images should have the same dimensions");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    avifImage * gmImage = gainMap->image;
    if (gmImage->width == 0 || gmImage->height == 0 || gmImage->depth == 0 || (int)gmImage->yuvFormat <= (int)AVIF_PIXEL_FORMAT_NONE ||
        (int)gmImage->yuvFormat > (int)AVIF_PIXEL_FORMAT_YUV400) {
        diagPrintf(diag, "gainMap->image should be non null with desired width, height, depth and yuvFormat set");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const bool colorSpacesDiffer = baseColorPrimaries != altColorPrimaries;
    int mathPrimaries = 0;
    if (!gainMapChooseMathPrimaries(baseColorPrimaries, altColorPrimaries, &mathPrimaries))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    const uint32_t width = baseRgbImage->width, height = baseRgbImage->height;
    GainMapComputeArgs A;
    memset(&A, 0, sizeof(A));
    if (!gainMapLayout(baseRgbImage, &A.baseL) || !gainMapLayout(altRgbImage, &A.altL)) {
        diagPrintf(diag, "Unsupported RGB color space");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    if (!width || !height || !baseRgbImage->pixels || !altRgbImage->pixels || gmImage->depth > 16)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const size_t numPixels = (size_t)width * height;
    const bool singleChannel = gmImage->yuvFormat == AVIF_PIXEL_FORMAT_YUV400;
    const int channels = singleChannel ? 1 : 3;
    avifResult r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    hipStream_t stream = deviceResident ? pickStream(hipStream) : tls.stream;
    ScratchScope scratch(stream); // (shares work buffers with the application's asynchronous entry point: see applyGainMapToHostImage)
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    QuiesceOnExit quiesceOnExit;
    PhaseTrace trace("compute gain map");
    if (deviceResident) {
        const PlaneGeometry want = planeGeometry(gmImage);
        for (int p = 0; p < 3; ++p) {
            const bool needed = p == 0 || gmImage->yuvFormat != AVIF_PIXEL_FORMAT_YUV400;
            if (needed && (!gmImage->yuvPlanes[p] || gmImage->yuvRowBytes[p] < want.widthBytes[p])) {
                setError("avifhipRGBImageComputeGainMapAsync: gainMap->image's planes must be allocated by the caller, in device memory, at the requested size");
                return AVIF_RESULT_INVALID_ARGUMENT;
            }
        }
    }
    tls.gainMapCache.valid = false; // (the apply path's tables are not touched, but keep the two paths independent of call order)

    // avifGainMapSetEncodingDefaults, :18-30
    for (int i = 0; i < 3; ++i) {
        gainMap->gainMapMin[i].n = 1, gainMap->gainMapMin[i].d = 1, gainMap->gainMapMax[i].n = 1, gainMap->gainMapMax[i].d = 1;
        gainMap->baseOffset[i].n = 1, gainMap->baseOffset[i].d = 64, gainMap->alternateOffset[i].n = 1, gainMap->alternateOffset[i].d = 64;
        gainMap->gainMapGamma[i].n = 1, gainMap->gainMapGamma[i].d = 1;
    }
    gainMap->baseHdrHeadroom.n = 0, gainMap->baseHdrHeadroom.d = 1, gainMap->alternateHdrHeadroom.n = 1, gainMap->alternateHdrHeadroom.d = 1;
    gainMap->useBaseColorSpace = (mathPrimaries == (int)baseColorPrimaries) ? AVIF_TRUE : AVIF_FALSE;

    if (colorSpacesDiffer) {
        const bool ok = gainMap->useBaseColorSpace ? gainMapPrimariesMatrix(altColorPrimaries, baseColorPrimaries, A.M)
                                                   : gainMapPrimariesMatrix(baseColorPrimaries, altColorPrimaries, A.M);
        if (!ok) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
        A.convertAlt = gainMap->useBaseColorSpace ? 1 : 0, A.convertBase = gainMap->useBaseColorSpace ? 0 : 1;
    }
    A.singleChannel = singleChannel ? 1 : 0;
    gainMapYCoefficients(mathPrimaries, A.yCoeffs);
    float baseOffset[3], altOffset[3];
    for (int c = 0; c < 3; ++c)
        baseOffset[c] = fractionToFloat(gainMap->baseOffset[c]), altOffset[c] = fractionToFloat(gainMap->alternateOffset[c]);

    // ---- device copies of the two images, lookup tables ----
    A.width = width, A.height = height;
    const uint32_t baseWidthBytes = width * rgbPixelBytes(baseRgbImage), altWidthBytes = width * rgbPixelBytes(altRgbImage);
    if (deviceResident) {
        if (baseRgbImage->rowBytes < baseWidthBytes || altRgbImage->rowBytes < altWidthBytes)
            return AVIF_RESULT_INVALID_ARGUMENT;
        A.base = baseRgbImage->pixels, A.basePitch = baseRgbImage->rowBytes, A.alt = altRgbImage->pixels, A.altPitch = altRgbImage->rowBytes;
    } else {
        A.basePitch = alignUp(baseWidthBytes, 256), A.altPitch = alignUp(altWidthBytes, 256);
        if ((r = reserve(tls.gainMap[5], (size_t)A.basePitch * height)) != AVIF_RESULT_OK || (r = reserve(tls.gainMap[9], (size_t)A.altPitch * height)) != AVIF_RESULT_OK)
            return r;
        A.base = (const uint8_t *)tls.gainMap[5].ptr, A.alt = (const uint8_t *)tls.gainMap[9].ptr;
        HIP_TRY(hipMemcpy2DAsync(tls.gainMap[5].ptr, A.basePitch, baseRgbImage->pixels, baseRgbImage->rowBytes, baseWidthBytes, height, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpy2DAsync(tls.gainMap[9].ptr, A.altPitch, altRgbImage->pixels, altRgbImage->rowBytes, altWidthBytes, height, hipMemcpyHostToDevice, stream));
        trace.mark("uploads");
    }
    std::vector<float> tables = gainMapLinearLut(baseTransferCharacteristics, baseRgbImage->depth, baseRgbImage->isFloat != 0);
    const size_t altLutOffset = tables.size();
    {
        const std::vector<float> & alt = gainMapLinearLut(altTransferCharacteristics, altRgbImage->depth, altRgbImage->isFloat != 0);
        tables.insert(tables.end(), alt.begin(), alt.end());
    }
    // Pass 0 looks for negative channels on the converted side (:618-660).  Where every coefficient of the conversion and every entry of that
    // side's linear-light table is >= 0, every product and every sum is, the minimum the pass would find is the 0 it starts from, and the
    // offsets stay as they are: the pass is not run (BT.709 or P3 into BT.2020 -- the math space is the wider gamut -- is such a conversion).
    bool minimaAreZero = colorSpacesDiffer;
    if (colorSpacesDiffer) {
        for (int k = 0; k < 9; ++k)
            minimaAreZero = minimaAreZero && A.M[k] >= 0.0;
        const size_t first = A.convertAlt ? altLutOffset : 0, last = A.convertAlt ? tables.size() : altLutOffset;
        for (size_t k = first; k < last && minimaAreZero; ++k)
            minimaAreZero = tables[k] >= 0.0f; // (a NaN entry fails the comparison too)
    }
    // room for the step tables that follow (3 channels x at most 65536 entries)
    const size_t stepsOffset = (tables.size() + 3) & ~(size_t)3, stepsCapacity = (size_t)3 * 65536;
    if ((r = reserve(tls.gainMap[6], (stepsOffset + stepsCapacity) * sizeof(float))) != AVIF_RESULT_OK)
        return r;
    if ((r = uploadTableAsync(tls.gainMap[6].ptr, tables.data(), tables.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
        return r;
    float * deviceTables = (float *)tls.gainMap[6].ptr;
    A.baseLut = deviceTables, A.altLut = deviceTables + altLutOffset;
    A.baseLutEntries = (uint32_t)altLutOffset, A.altLutEntries = (uint32_t)(tables.size() - altLutOffset);
    if ((r = reserve(tls.gainMap[7], (size_t)channels * numPixels * sizeof(float))) != AVIF_RESULT_OK ||
        (r = reserve(tls.gainMap[3], ((size_t)kGainMapMaxGroups * 16 + 8) * sizeof(float))) != AVIF_RESULT_OK)
        return r;
    A.ratios = (float *)tls.gainMap[7].ptr, A.partials = (float *)tls.gainMap[3].ptr;
    const uint32_t groups = gainMapComputeGroups(width, height);
    if (!tls.gainMapPartials) {
        HIP_TRY(hipHostMalloc(&tls.gainMapPartials, kGainMapPinnedBytes, hipHostMallocDefault));
        AVIFHIP_NEW_HOST_MEMORY(tls.gainMapPartials, kGainMapPinnedBytes);
    }
    float * const partials = (float *)tls.gainMapPartials; // (groups x 8, fetched after passes 0 and 1; the histograms after pass 2)
    const size_t partialsBytes = (size_t)groups * 8 * sizeof(float);
    if (2 * partialsBytes > kGainMapPinnedBytes)
        return AVIF_RESULT_UNKNOWN_ERROR;

    // ---- pass 0: offsets that keep the converted side's channels positive, :618-660; pass 1: ratios, maxima, extreme ratios, :662-715 ----
    // One wait for both (round 6): pass 0 leaves its minima behind pass 1's partials, one workgroup folds them into the offsets pass 1 reads
    // (kernels_gainmap.hip gainMapOffsetsKernel); the host repeats the fold -- in the reference's order, with its AVIF_MIN -- for the metadata, and runs pass 1 again
    // with its own offsets in the one case the two folds can differ (a NaN among the minima).
    A.offsets = nullptr;
    for (int c = 0; c < 3; ++c)
        A.baseOffset[c] = baseOffset[c], A.altOffset[c] = altOffset[c];
    auto foldMinima = [&](const float * minima, bool referenceOrder, float base[3], float alt[3]) {
        float channelMin[3] = { 0.0f, 0.0f, 0.0f };
        for (uint32_t g = 0; g < groups; ++g)
            for (int c = 0; c < 3; ++c) {
                const float p = minima[(size_t)g * 8 + c];
                channelMin[c] = referenceOrder ? ((channelMin[c] < p) ? channelMin[c] : p) : fminf(channelMin[c], p);
            }
        for (int c = 0; c < 3; ++c) {
            const float maxOffset = 0.1f;
            if (channelMin[c] < -1e-10f) {
                if (gainMap->useBaseColorSpace) {
                    const float o = alt[c] - channelMin[c];
                    alt[c] = (o < maxOffset) ? o : maxOffset;
                } else {
                    const float o = base[c] - channelMin[c];
                    base[c] = (o < maxOffset) ? o : maxOffset;
                }
            }
        }
    };
    {
        float * const deviceMinima = A.partials + (size_t)groups * 8;
        const bool runPass0 = colorSpacesDiffer && !minimaAreZero;
        if (runPass0) {
            GainMapComputeArgs A0 = A;
            A0.partials = deviceMinima;
            const hipError_t e = launchGainMapChannelMin(A0, stream);
            if (e != hipSuccess)
                return hipFailed(e, "gain map channel-minimum kernel launch");
            float * const deviceOffsets = deviceMinima + (size_t)groups * 8;
            const hipError_t f = launchGainMapOffsets(deviceMinima, groups, gainMap->useBaseColorSpace != 0, baseOffset, altOffset, deviceOffsets, stream);
            if (f != hipSuccess)
                return hipFailed(f, "gain map offsets kernel launch");
            A.offsets = deviceOffsets;
        }
        hipError_t e = launchGainMapRatios(A, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map ratio kernel launch");
        HIP_TRY(hipMemcpyAsync(partials, A.partials, partialsBytes * (runPass0 ? 2 : 1), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        trace.mark("minima");
        if (runPass0) {
            float deviceBase[3] = { baseOffset[0], baseOffset[1], baseOffset[2] }, deviceAlt[3] = { altOffset[0], altOffset[1], altOffset[2] };
            foldMinima(partials + (size_t)groups * 8, false, deviceBase, deviceAlt);
            foldMinima(partials + (size_t)groups * 8, true, baseOffset, altOffset);
            A.offsets = nullptr;
            for (int c = 0; c < 3; ++c)
                A.baseOffset[c] = baseOffset[c], A.altOffset[c] = altOffset[c];
            if (memcmp(deviceBase, baseOffset, sizeof(deviceBase)) != 0 || memcmp(deviceAlt, altOffset, sizeof(deviceAlt)) != 0) {
                e = launchGainMapRatios(A, stream); // (the kernel's fold dropped a NaN the reference's lets through)
                if (e != hipSuccess)
                    return hipFailed(e, "gain map ratio kernel launch");
                HIP_TRY(hipMemcpyAsync(partials, A.partials, partialsBytes, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
        }
        trace.mark("ratios");
    }
    float baseMax = 1.0f, altMax = 1.0f, minRatio[3] = { INFINITY, INFINITY, INFINITY }, maxRatio[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t g = 0; g < groups; ++g) {
        const float * p = &partials[(size_t)g * 8];
        baseMax = fmaxf(baseMax, p[0]), altMax = fmaxf(altMax, p[1]);
        for (int c = 0; c < channels; ++c)
            minRatio[c] = fminf(minRatio[c], p[2 + c]), maxRatio[c] = fmaxf(maxRatio[c], p[5 + c]);
    }
    const float kEps = 1e-10f;
    const double baseHeadroom = log2f(baseMax > kEps ? baseMax : kEps), alternateHeadroom = log2f(altMax > kEps ? altMax : kEps);
    if (!gainMapDoubleToUnsignedFraction(baseHeadroom, &gainMap->baseHdrHeadroom.n, &gainMap->baseHdrHeadroom.d) ||
        !gainMapDoubleToUnsignedFraction(alternateHeadroom, &gainMap->alternateHdrHeadroom.n, &gainMap->alternateHdrHeadroom.d))
        return AVIF_RESULT_INVALID_ARGUMENT;
    const float sign = (alternateHeadroom < baseHeadroom) ? -1.0f : 1.0f; // :728-739

    // ---- pass 2: range without outliers, :741-749 ----
    GainMapChannelRange ranges[3];
    GainMapStepTable stepTables[3];
    memset(stepTables, 0, sizeof(stepTables));
    float minLog2[3] = { 0.0f, 0.0f, 0.0f }, maxLog2[3] = { 0.0f, 0.0f, 0.0f };
    bool anyHistogram = false;
    std::vector<float> hostSteps;
    size_t histogramOffset[3] = { 0, 0, 0 }, histogramTotal = 0;
    for (int c = 0; c < channels; ++c) {
        ranges[c] = gainMapChannelRange(sign, minRatio[c], maxRatio[c], numPixels);
        minLog2[c] = ranges[c].lo, maxLog2[c] = ranges[c].hi;
        if (ranges[c].numBuckets > 0) {
            uint32_t entries = 0;
            const std::vector<float> steps = gainMapBucketSteps(ranges[c], &entries);
            stepTables[c].steps = deviceTables + stepsOffset + hostSteps.size();
            stepTables[c].entries = entries, stepTables[c].flipped = sign < 0 ? 1 : 0, stepTables[c].flip = (uint32_t)ranges[c].numBuckets - 1;
            // the buckets are equally wide in log2 of the ratio: a line through the first and the last finite step gives the kernel its first
            // guess of a sample's bucket (it corrects the guess against the steps themselves: kernels_gainmap.hip stepIndexGuessed)
            const uint32_t lastStep = (uint32_t)ranges[c].numBuckets - 1;
            if (lastStep >= 2 && steps[1] > 0.0f && steps[lastStep] > steps[1] && std::isfinite(steps[lastStep])) {
                const double l1 = std::log2((double)steps[1]), l2 = std::log2((double)steps[lastStep]);
                const double a = (double)(lastStep - 1) / (l2 - l1);
                stepTables[c].guessA = (float)a, stepTables[c].guessB = (float)(1.0 - a * l1 + 0.5);
            }
            hostSteps.insert(hostSteps.end(), steps.begin(), steps.end());
            histogramOffset[c] = histogramTotal, histogramTotal += (size_t)ranges[c].numBuckets;
            anyHistogram = true;
        }
    }
    if (anyHistogram) {
        if ((r = reserve(tls.gainMap[8], histogramTotal * sizeof(uint32_t))) != AVIF_RESULT_OK)
            return r;
        if ((r = uploadTableAsync(deviceTables + stepsOffset, hostSteps.data(), hostSteps.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
            return r;
        HIP_TRY(hipMemsetAsync(tls.gainMap[8].ptr, 0, histogramTotal * sizeof(uint32_t), stream));
        uint32_t * histograms[3];
        for (int c = 0; c < 3; ++c)
            histograms[c] = (uint32_t *)tls.gainMap[8].ptr + histogramOffset[c];
        const hipError_t e = launchGainMapHistogram(A.ratios, numPixels, channels, stepTables, histograms, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map histogram kernel launch");
        if (histogramTotal * sizeof(uint32_t) > kGainMapPinnedBytes)
            return AVIF_RESULT_UNKNOWN_ERROR; // (the reference caps a histogram at 10 000 buckets, src/gainmap.c:393)
        const uint32_t * const hostHistograms = (const uint32_t *)tls.gainMapPartials;
        HIP_TRY(hipMemcpyAsync(tls.gainMapPartials, tls.gainMap[8].ptr, histogramTotal * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        trace.mark("histograms");
        for (int c = 0; c < channels; ++c)
            if (ranges[c].numBuckets > 0)
                gainMapRangeWithoutOutliers(ranges[c], hostHistograms + histogramOffset[c], &minLog2[c], &maxLog2[c]);
    }
    for (int c = 0; c < 3; ++c) { // metadata, :751-760
        const int k = singleChannel ? 0 : c;
        if (!gainMapDoubleToFraction(minLog2[k], &gainMap->gainMapMin[c].n, &gainMap->gainMapMin[c].d) ||
            !gainMapDoubleToFraction(maxLog2[k], &gainMap->gainMapMax[c].n, &gainMap->gainMapMax[c].d) ||
            !gainMapDoubleToFraction(altOffset[c], &gainMap->alternateOffset[c].n, &gainMap->alternateOffset[c].d) ||
            !gainMapDoubleToFraction(baseOffset[c], &gainMap->baseOffset[c].n, &gainMap->baseOffset[c].d))
            return AVIF_RESULT_INVALID_ARGUMENT;
    }

    // ---- pass 3: [min, max] -> codes -> RGBA -> YUV (-> requested size), :762-829 ----
    hostSteps.clear();
    memset(stepTables, 0, sizeof(stepTables));
    const uint32_t depth = gmImage->depth;
    for (int c = 0; c < channels; ++c) {
        const float range = (maxLog2[c] - minLog2[c] > 0.0f) ? maxLog2[c] - minLog2[c] : 0.0f;
        if (range == 0.0f)
            continue; // every value becomes 0, :766-773
        const std::vector<float> steps = gainMapCodeSteps(ranges[c], minLog2[c], maxLog2[c], fractionToFloat(gainMap->gainMapGamma[c]), depth);
        stepTables[c].steps = deviceTables + stepsOffset + hostSteps.size();
        stepTables[c].entries = (uint32_t)steps.size(), stepTables[c].flipped = sign < 0 ? 1 : 0, stepTables[c].flip = (1u << depth) - 1;
        // gamma 1 (the encoding default): the codes are equally spaced in log2 of the ratio between the range's ends -- the kernel starts from
        // the line through the first and the last step and corrects against the steps (any other gamma: bisection)
        const uint32_t lastCode = (1u << depth) - 1;
        if (fractionToFloat(gainMap->gainMapGamma[c]) == 1.0f && lastCode >= 2 && steps.size() > lastCode && steps[1] > 0.0f && steps[lastCode] > steps[1] &&
            std::isfinite(steps[lastCode])) {
            const double l1 = std::log2((double)steps[1]), l2 = std::log2((double)steps[lastCode]);
            const double a = (double)(lastCode - 1) / (l2 - l1);
            stepTables[c].guessA = (float)a, stepTables[c].guessB = (float)(1.0 - a * l1 + 0.5);
        }
        hostSteps.insert(hostSteps.end(), steps.begin(), steps.end());
    }
    if (!hostSteps.empty() && (r = uploadTableAsync(deviceTables + stepsOffset, hostSteps.data(), hostSteps.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
        return r;
    avifRGBImage rgbGain; // avifRGBImageSetDefaults, src/avif.c:700-717
    memset(&rgbGain, 0, sizeof(rgbGain));
    rgbGain.width = width, rgbGain.height = height, rgbGain.depth = depth, rgbGain.format = AVIF_RGB_FORMAT_RGBA;
    rgbGain.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgbGain.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    rgbGain.maxThreads = 1;
    rgbGain.rowBytes = alignUp(width * 4 * ((depth > 8) ? 2 : 1), 256);
    if ((r = reserve(tls.gainMap[1], (size_t)rgbGain.rowBytes * height)) != AVIF_RESULT_OK)
        return r;
    rgbGain.pixels = (uint8_t *)tls.gainMap[1].ptr;
    {
        const hipError_t e = launchGainMapQuantise(A.ratios, width, height, channels, stepTables, rgbGain.pixels, rgbGain.rowBytes, depth, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map quantisation kernel launch");
    }
    const uint32_t requestedWidth = gmImage->width, requestedHeight = gmImage->height;
    if (deviceResident) {
        // the codes become planes where the caller put them: RGBA -> YUV at the images' size (into the caller's planes when that is the
        // requested size, else into scratch and through the plane scaler); nothing is downloaded, the planes are final when `stream` has drained
        avifImage deviceGain;
        memcpy(&deviceGain, gmImage, sizeof(avifImage));
        const bool scaled = requestedWidth != width || requestedHeight != height;
        if (scaled && (r = deviceGainMapPlanes(&deviceGain, width, height, tls.gainMap[10])) != AVIF_RESULT_OK)
            return r;
        if (scaled && !gmImage->alphaPlane)
            deviceGain.alphaPlane = nullptr, deviceGain.alphaRowBytes = 0; // (the scratch image mirrors the caller's planes)
        avifRGBImage codes = rgbGain; // (the gain map has no alpha plane: the opaque alpha of its RGBA codes goes nowhere, src/gainmap.c:792-800)
        codes.ignoreAlpha = deviceGain.alphaPlane ? AVIF_FALSE : AVIF_TRUE;
        if ((r = avifhipImageRGBToYUVAsync(&deviceGain, &codes, stream)) != AVIF_RESULT_OK)
            return r;
        if (scaled && (r = avifhipImageScaleAsync(&deviceGain, gmImage, stream)) != AVIF_RESULT_OK)
            return r;
        tls.lastKernel = "gainmap_compute";
        return AVIF_RESULT_OK;
    }
    // The planes a previous call left in gainMap->image are released only after the stream has drained: free() of memory the runtime pinned
    // for that call's downloads unmaps it from the GPU as well, and that stalls whatever kernel is running (the quantisation kernel above:
    // 168 us in a process's first call, 15-25 ms in every later one, when the release sat here).
    struct DeferredPlanes
    {
        uint8_t * yuv[3] = { nullptr, nullptr, nullptr };
        uint8_t * alpha = nullptr;
        ~DeferredPlanes()
        {
            for (uint8_t * p : yuv)
                free(p);
            free(alpha);
        }
    } stale;
    // ... and a plane that already has the size and pitch the allocation below would give it stays where it is (the reference frees and
    // allocates, src/gainmap.c:792-793: the same bytes at an address the caller cannot tell apart).  A caller that computes gain maps for a
    // sequence of frames into one avifGainMap spares every call the release of 25 MB the runtime had pinned and the faults of 6 000 fresh pages
    // under the downloads (3.5 of 9.3 ms for a 4K 4:4:4 map, and 15-25 ms stalls of the NEXT call's kernels while the unpinning ran).
    {
        const PlaneGeometry want = planeGeometry(gmImage);
        auto fits = [&](const uint8_t * plane, uint32_t rowBytes, int p) {
            return plane && want.rows[p] && rowBytes == want.widthBytes[p] && ownedBlockCapacity(plane) >= (size_t)rowBytes * want.rows[p];
        };
        if (gmImage->imageOwnsYUVPlanes)
            for (int p = 0; p < 3; ++p)
                if (!fits(gmImage->yuvPlanes[p], gmImage->yuvRowBytes[p], p))
                    stale.yuv[p] = gmImage->yuvPlanes[p], gmImage->yuvPlanes[p] = NULL;
        if (gmImage->imageOwnsAlphaPlane && !fits(gmImage->alphaPlane, gmImage->alphaRowBytes, 3))
            stale.alpha = gmImage->alphaPlane, gmImage->alphaPlane = NULL;
    }
    DeferredPlanes kept; // (out of the image while the device works -- an early return releases them -- and back in before the downloads)
    uint32_t keptRowBytes[4] = { gmImage->yuvRowBytes[0], gmImage->yuvRowBytes[1], gmImage->yuvRowBytes[2], gmImage->alphaRowBytes };
    if (gmImage->imageOwnsYUVPlanes)
        for (int p = 0; p < 3; ++p)
            kept.yuv[p] = gmImage->yuvPlanes[p], gmImage->yuvPlanes[p] = NULL;
    if (gmImage->imageOwnsAlphaPlane)
        kept.alpha = gmImage->alphaPlane, gmImage->alphaPlane = NULL;
    freeHostPlanes(gmImage); // (what is left: pointers the image does not own)
    avifImage deviceGain;
    memcpy(&deviceGain, gmImage, sizeof(avifImage));
    if ((r = deviceGainMapPlanes(&deviceGain, width, height, tls.gainMap[10])) != AVIF_RESULT_OK)
        return r;
    if ((r = avifhipImageRGBToYUVAsync(&deviceGain, &rgbGain, stream)) != AVIF_RESULT_OK)
        return r;
    avifImage deviceFinal;
    memcpy(&deviceFinal, &deviceGain, sizeof(avifImage));
    if (requestedWidth != width || requestedHeight != height) {
        if ((r = deviceGainMapPlanes(&deviceFinal, requestedWidth, requestedHeight, tls.gainMap[4])) != AVIF_RESULT_OK)
            return r;
        if ((r = avifhipImageScaleAsync(&deviceGain, &deviceFinal, stream)) != AVIF_RESULT_OK)
            return r;
    }
    gmImage->width = deviceFinal.width, gmImage->height = deviceFinal.height;
    trace.mark("enqueued");
    for (int p = 0; p < 3; ++p)
        if (kept.yuv[p])
            gmImage->yuvPlanes[p] = kept.yuv[p], gmImage->yuvRowBytes[p] = keptRowBytes[p], kept.yuv[p] = nullptr;
    if (kept.alpha)
        gmImage->alphaPlane = kept.alpha, gmImage->alphaRowBytes = keptRowBytes[3], kept.alpha = nullptr;
    if ((r = allocateHostPlanes(gmImage, true)) != AVIF_RESULT_OK) {
        freeHostPlanes(gmImage);
        return r;
    }
    trace.mark("planes allocated");
    const PlaneGeometry g = planeGeometry(gmImage);
    for (int p = 0; p < 4; ++p) {
        uint8_t * host = (p < 3) ? gmImage->yuvPlanes[p] : gmImage->alphaPlane;
        const uint8_t * dev = (p < 3) ? deviceFinal.yuvPlanes[p] : deviceFinal.alphaPlane;
        if (!host || !dev)
            continue;
        HIP_TRY(hipMemcpy2DAsync(host, (p < 3) ? gmImage->yuvRowBytes[p] : gmImage->alphaRowBytes, dev, (p < 3) ? deviceFinal.yuvRowBytes[p] : deviceFinal.alphaRowBytes,
                                 g.widthBytes[p], g.rows[p], hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    trace.mark("downloaded");
    tls.lastKernel = "gainmap_compute";
    return AVIF_RESULT_OK;
}


extern "C" avifResult avifhipRGBImageComputeGainMap(const avifRGBImage * baseRgbImage, avifColorPrimaries baseColorPrimaries,
                                                    avifTransferCharacteristics baseTransferCharacteristics, const avifRGBImage * altRgbImage,
                                                    avifColorPrimaries altColorPrimaries, avifTransferCharacteristics altTransferCharacteristics,
                                                    avifGainMap * gainMap, avifDiagnostics * diag)
{
    return computeGainMapImpl(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries, altTransferCharacteristics, gainMap, diag,
                              false, nullptr);
}

// ... with both renditions and gainMap->image's planes in device memory (include/avifhip.h)
extern "C" avifResult avifhipRGBImageComputeGainMapAsync(const avifRGBImage * baseRgbImage, avifColorPrimaries baseColorPrimaries,
                                                         avifTransferCharacteristics baseTransferCharacteristics, const avifRGBImage * altRgbImage,
                                                         avifColorPrimaries altColorPrimaries, avifTransferCharacteristics altTransferCharacteristics,
                                                         avifGainMap * gainMap, avifDiagnostics * diag, void * hipStream)
{
    return computeGainMapImpl(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries, altTransferCharacteristics, gainMap, diag,
                              true, hipStream);
}

// the kernels of avifhipRGBImageComputeGainMapAsync between two events on the stream (a tool for bench.py / rocprof rows): milliseconds per call
extern "C" double avifhipTimeRGBImageComputeGainMap(const avifRGBImage * baseRgbImage, avifColorPrimaries baseColorPrimaries,
                                                    avifTransferCharacteristics baseTransferCharacteristics, const avifRGBImage * altRgbImage,
                                                    avifColorPrimaries altColorPrimaries, avifTransferCharacteristics altTransferCharacteristics,
                                                    avifGainMap * gainMap, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (avifhipRGBImageComputeGainMapAsync(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries, altTransferCharacteristics,
                                               gainMap, nullptr, stream) != AVIF_RESULT_OK)
            return -1.0;
    if (hipStreamSynchronize(stream) != hipSuccess)
        return -1.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < iters; ++k)
        if (avifhipRGBImageComputeGainMapAsync(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries, altTransferCharacteristics,
                                               gainMap, nullptr, stream) != AVIF_RESULT_OK)
            return -1.0;
    if (hipStreamSynchronize(stream) != hipSuccess)
        return -1.0;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
}

// avifImageComputeGainMap, src/gainmap.c:843-912: both renditions arrive as YUV
extern "C" avifResult avifhipImageComputeGainMap(const avifImage * baseImage, const avifImage * altImage, avifGainMap * gainMap, avifDiagnostics * diag)
{
    diagClear(diag);
    if (baseImage == NULL || altImage == NULL || gainMap == NULL)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseImage->avifhipOpaqueIcc_[1] > 0 || altImage->avifhipOpaqueIcc_[1] > 0) {
        diagPrintf(diag, "Computing gain maps for images with ICC profiles is not supported");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    if (baseImage->width != altImage->width || baseImage->height != altImage->height) {
        diagPrintf(diag, "Image dimensions don't match, got %dx%d and %dx%d", baseImage->width, baseImage->height, altImage->width, altImage->height);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    avifRGBImage rgb[2]; // avifRGBImageSetDefaults + avifRGBImageAllocatePixels for each rendition
    const avifImage * src[2] = { baseImage, altImage };
    memset(rgb, 0, sizeof(rgb));
    avifResult r = AVIF_RESULT_OK;
    for (int k = 0; k < 2 && r == AVIF_RESULT_OK; ++k) {
        rgb[k].width = src[k]->width, rgb[k].height = src[k]->height, rgb[k].depth = src[k]->depth, rgb[k].format = AVIF_RGB_FORMAT_RGBA;
        rgb[k].chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgb[k].chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
        rgb[k].maxThreads = 1;
        const uint32_t pixelBytes = rgbPixelBytes(&rgb[k]);
        if (!rgb[k].width || !rgb[k].height || rgb[k].width > UINT32_MAX / pixelBytes) {
            r = AVIF_RESULT_INVALID_ARGUMENT;
            break;
        }
        rgb[k].rowBytes = rgb[k].width * pixelBytes;
        rgb[k].pixels = (uint8_t *)malloc((size_t)rgb[k].rowBytes * rgb[k].height);
        r = rgb[k].pixels ? avifhipImageYUVToRGB(src[k], &rgb[k]) : AVIF_RESULT_OUT_OF_MEMORY;
    }
    if (r == AVIF_RESULT_OK)
        r = avifhipRGBImageComputeGainMap(&rgb[0], baseImage->colorPrimaries, baseImage->transferCharacteristics, &rgb[1], altImage->colorPrimaries,
                                          altImage->transferCharacteristics, gainMap, diag);
    if (r == AVIF_RESULT_OK) { // :900-906 (the alternate image has no ICC profile here: avifRWDataSet(.., NULL, 0) empties altICC)
        free(gainMap->altICC.data);
        gainMap->altICC.data = NULL, gainMap->altICC.size = 0;
        gainMap->altColorPrimaries = altImage->colorPrimaries;
        gainMap->altTransferCharacteristics = altImage->transferCharacteristics;
        gainMap->altMatrixCoefficients = altImage->matrixCoefficients;
        gainMap->altDepth = altImage->depth;
        gainMap->altPlaneCount = (altImage->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 1 : 3;
        memcpy(&gainMap->altCLLI, altImage->avifhipOpaqueTail_, sizeof(gainMap->altCLLI)); // avifImage.clli @110
    }
    free(rgb[0].pixels);
    free(rgb[1].pixels);
    return r;
}

// Light levels exactly like the reference accumulates them (include/avifhip.h)
extern "C" void avifhipSetExactLightLevels(int on)
{
    gExactLightLevels.store(on ? 1 : 0, std::memory_order_relaxed);
}

avifResult avifhip::api::settleLightLevels(hipStream_t stream, bool everyStream, bool wait)
{
    for (int k = 0; k < Context::kLightSlots; ++k) {
        Context::PendingLight & P = tls.lightPending[k];
        if (!P.pending || (!everyStream && P.stream != stream))
            continue;
        if (wait) {
            HIP_TRY(hipEventSynchronize(tls.lightCopied[k]));
        } else if (hipEventQuery(tls.lightCopied[k]) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        const GainMapPartial * partials = (const GainMapPartial *)tls.lightPinned[k];
        float rgbMax = 0.0f;
        double sum = 0.0;
        for (uint32_t g = 0; g < P.count; ++g) { // (index order, like the synchronous path)
            rgbMax = (partials[g].max > rgbMax) ? partials[g].max : rgbMax;
            sum += partials[g].sum;
        }
        avifContentLightLevelInformationBox * clli = static_cast<avifContentLightLevelInformationBox *>(P.clli);
        clli->maxCLL = lightLevelNits(rgbMax);
        clli->maxPALL = lightLevelNits((float)sum / (float)P.pixels);
        P.pending = false;
    }
    return AVIF_RESULT_OK;
}
