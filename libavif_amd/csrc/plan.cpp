// plan.cpp -- derives a kernel plan from (avifImage, avifRGBImage): the host-side parameter layer.
// Mirrors the state preparation and dispatch decisions of libavif's reformat.c so that the fused
// kernels reproduce exactly what the reference's sequence of passes computes.
// Compiled with -ffp-contract=off: kr/kg/kb products must round like the reference's fp32 code.
#include "plan.h"

#include "exactdiv.h"

#include <string.h>

namespace avifhip {

bool rgbFormatHasAlpha(int f) // reference src/avif.c:675-679
{
    return f == AVIF_RGB_FORMAT_RGBA || f == AVIF_RGB_FORMAT_ARGB || f == AVIF_RGB_FORMAT_BGRA || f == AVIF_RGB_FORMAT_ABGR ||
           f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY;
}
bool rgbFormatIsGray(int f) // reference src/avif.c:670-673
{
    return f == AVIF_RGB_FORMAT_GRAY || f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY;
}
int rgbFormatChannelCount(int f) // reference src/avif.c:681-690
{
    if (f == AVIF_RGB_FORMAT_GRAY)
        return 1;
    if (f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY)
        return 2;
    return rgbFormatHasAlpha(f) ? 4 : 3;
}

static RcpHL reciprocalOf(float d, float scale)
{
    const RcpSplit r = splitReciprocal(d, scale); // exactdiv.h: the form the verifier enumerates
    return RcpHL { r.hi, r.lo };
}

// reference src/reformat.c:32-117 (avifGetRGBColorSpaceInfo)
static bool fillRgbSide(const avifRGBImage * rgb, RgbSide * s)
{
    const uint32_t d = rgb->depth;
    if (!(d == 8 || d == 10 || d == 12 || d == 16))
        return false;
    if (rgb->isFloat && d != 16)
        return false;
    const int f = (int)rgb->format;
    if (f == AVIF_RGB_FORMAT_RGB_565 && d != 8)
        return false;
    if (f < AVIF_RGB_FORMAT_RGB || f >= AVIF_RGB_FORMAT_COUNT)
        return false;
    memset(s, 0, sizeof(*s));
    s->pixels = rgb->pixels;
    s->rowBytes = rgb->rowBytes;
    s->depth = d;
    s->format = f;
    s->chanBytes = (d > 8) ? 2 : 1;
    s->hasAlpha = rgbFormatHasAlpha(f);
    s->isGray = rgbFormatIsGray(f);
    s->is565 = (f == AVIF_RGB_FORMAT_RGB_565);
    s->isFloat = rgb->isFloat ? 1 : 0;
    s->pixBytes = s->is565 ? 2 : rgbFormatChannelCount(f) * s->chanBytes;
    int r = 0, g = 0, b = 0, a = 0, y = 0; // channel slots
    switch (f) {
        case AVIF_RGB_FORMAT_RGB:   r = 0, g = 1, b = 2; break;
        case AVIF_RGB_FORMAT_RGBA:  r = 0, g = 1, b = 2, a = 3; break;
        case AVIF_RGB_FORMAT_ARGB:  a = 0, r = 1, g = 2, b = 3; break;
        case AVIF_RGB_FORMAT_BGR:   b = 0, g = 1, r = 2; break;
        case AVIF_RGB_FORMAT_BGRA:  b = 0, g = 1, r = 2, a = 3; break;
        case AVIF_RGB_FORMAT_ABGR:  a = 0, b = 1, g = 2, r = 3; break;
        case AVIF_RGB_FORMAT_GRAYA: y = 0, a = 1; break;
        case AVIF_RGB_FORMAT_AGRAY: a = 0, y = 1; break;
        default: break; // 565 and GRAY: everything at offset 0
    }
    s->offR = r * s->chanBytes;
    s->offG = g * s->chanBytes;
    s->offB = b * s->chanBytes;
    s->offA = a * s->chanBytes;
    s->offGray = y * s->chanBytes;
    s->maxv = (1 << d) - 1;
    s->maxf = (float)s->maxv;
    const float scale = 1.0f / (float)((1 << d) - 1); // reference src/reformat.c:1429-1430
    s->f16Multiplier = 1.9259299444e-34f * scale;
    s->rcpMax = reciprocalOf(s->maxf, 1.0f);
    return true;
}

// reference src/colr.c:16-29 and :517-542 (CHROMA_DERIVED_NCL: kr/kb from the colour primaries)
static void coefficientsFromPrimaries(unsigned cp, float * kr, float * kb)
{
    struct Row { unsigned cp; float v[8]; };
    static const Row rows[] = {
        { 1, { 0.64f, 0.33f, 0.3f, 0.6f, 0.15f, 0.06f, 0.3127f, 0.329f } },
        { 4, { 0.67f, 0.33f, 0.21f, 0.71f, 0.14f, 0.08f, 0.310f, 0.316f } },
        { 5, { 0.64f, 0.33f, 0.29f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f } },
        { 6, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
        { 7, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
        { 8, { 0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f, 0.316f } },
        { 9, { 0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f } },
        { 10, { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.3333f, 0.3333f } },
        { 11, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f, 0.351f } },
        { 12, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f } },
        { 22, { 0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f } },
    };
    const float * v = rows[0].v;
    for (const Row & row : rows) {
        if (row.cp == cp) {
            v = row.v;
            break;
        }
    }
    const float rX = v[0], rY = v[1], gX = v[2], gY = v[3], bX = v[4], bY = v[5], wX = v[6], wY = v[7];
    const float rZ = 1.0f - (rX + rY), gZ = 1.0f - (gX + gY), bZ = 1.0f - (bX + bY), wZ = 1.0f - (wX + wY);
    const float den = (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    *kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) / den;
    *kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) / den;
}

// reference src/colr.c:123-189 (matrixCoefficientsTables, avifCalcYUVCoefficients)
void calcYuvCoefficients(const avifImage * image, float * krOut, float * kgOut, float * kbOut)
{
    float kr = 0.299f, kb = 0.114f; // unspecified => BT.601, src/colr.c:173-176
    bool known = true;
    switch (image->matrixCoefficients) {
        case AVIF_MATRIX_COEFFICIENTS_BT709: kr = 0.2126f, kb = 0.0722f; break;
        case AVIF_MATRIX_COEFFICIENTS_FCC: kr = 0.30f, kb = 0.11f; break;
        case AVIF_MATRIX_COEFFICIENTS_BT470BG:
        case AVIF_MATRIX_COEFFICIENTS_BT601: kr = 0.299f, kb = 0.114f; break;
        case AVIF_MATRIX_COEFFICIENTS_SMPTE240: kr = 0.212f, kb = 0.087f; break;
        case AVIF_MATRIX_COEFFICIENTS_BT2020_NCL: kr = 0.2627f, kb = 0.0593f; break;
        case AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL: coefficientsFromPrimaries(image->colorPrimaries, &kr, &kb); break;
        default: known = false; break;
    }
    float kg = 1.0f - 0.299f - 0.114f;
    if (known)
        kg = 1.0f - kr - kb;
    *krOut = kr;
    *kgOut = kg;
    *kbOut = kb;
}

// reference src/reformat.c:119-159 (avifGetYUVColorSpaceInfo), src/avif.c:39-72
static bool fillYuvSide(const avifImage * image, YuvSide * s)
{
    const uint32_t d = image->depth;
    if (!(d == 8 || d == 10 || d == 12 || d == 16))
        return false;
    const int fmt = (int)image->yuvFormat;
    if (fmt < AVIF_PIXEL_FORMAT_YUV444 || fmt >= AVIF_PIXEL_FORMAT_COUNT)
        return false;
    if (image->yuvRange != AVIF_RANGE_LIMITED && image->yuvRange != AVIF_RANGE_FULL)
        return false;
    const unsigned mc = image->matrixCoefficients;
    const bool ycgco = (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO || mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE ||
                        mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO);
    if (mc == 3 || (ycgco && image->yuvRange == AVIF_RANGE_LIMITED) || mc == AVIF_MATRIX_COEFFICIENTS_BT2020_CL ||
        mc == AVIF_MATRIX_COEFFICIENTS_SMPTE2085 || mc == AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_CL ||
        mc == AVIF_MATRIX_COEFFICIENTS_ICTCP || mc >= AVIF_MATRIX_COEFFICIENTS_LAST)
        return false;
    if (mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY && fmt != AVIF_PIXEL_FORMAT_YUV444 && fmt != AVIF_PIXEL_FORMAT_YUV400)
        return false;

    memset(s, 0, sizeof(*s));
    for (int p = 0; p < 3; ++p) {
        s->plane[p] = image->yuvPlanes[p];
        s->rowBytes[p] = image->yuvRowBytes[p];
    }
    s->alpha = image->alphaPlane;
    s->alphaRowBytes = image->alphaRowBytes;
    s->depth = d;
    s->format = fmt;
    s->chanBytes = (d > 8) ? 2 : 1;
    s->shiftX = (fmt == AVIF_PIXEL_FORMAT_YUV444) ? 0 : 1;
    s->shiftY = (fmt == AVIF_PIXEL_FORMAT_YUV420 || fmt == AVIF_PIXEL_FORMAT_YUV400) ? 1 : 0;
    s->hasColor = (image->yuvPlanes[1] && image->yuvPlanes[2] && image->yuvRowBytes[1] && image->yuvRowBytes[2] &&
                   fmt != AVIF_PIXEL_FORMAT_YUV400)
                      ? 1
                      : 0;
    s->limited = (image->yuvRange == AVIF_RANGE_LIMITED) ? 1 : 0;
    s->maxv = (1 << d) - 1;

    calcYuvCoefficients(image, &s->kr, &s->kg, &s->kb);
    s->biasY = s->limited ? (float)(16 << (d - 8)) : 0.0f;
    s->biasUV = (float)(1 << (d - 1));
    s->rangeY = (float)(s->limited ? (219 << (d - 8)) : s->maxv);
    s->rangeUV = (float)(s->limited ? (224 << (d - 8)) : s->maxv);
    return true;
}

// reference src/reformat.c:161-194 (avifPrepareReformatState)
static bool prepareState(const avifImage * image, const avifRGBImage * rgb, YuvSide * y, RgbSide * r)
{
    const unsigned mc = image->matrixCoefficients;
    if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE || mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO) {
        const int bitOffset = (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE) ? 2 : 1;
        if ((int)image->depth - bitOffset != (int)rgb->depth)
            return false;
    }
    if (!fillRgbSide(rgb, r) || !fillYuvSide(image, y))
        return false;
    y->mode = MODE_COEFF;
    if (mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY)
        y->mode = MODE_IDENTITY;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO)
        y->mode = MODE_YCGCO;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE)
        y->mode = MODE_YCGCO_RE;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO)
        y->mode = MODE_YCGCO_RO;
    if (y->mode != MODE_COEFF)
        y->kr = y->kg = y->kb = 0.0f;
    // the constant sub-expressions of src/reformat.c:874-876, rounded like the reference's fp32 code
    y->twoOneMinusKr = 2 * (1 - y->kr);
    y->twoOneMinusKb = 2 * (1 - y->kb);
    y->krOneMinusKr = y->kr * (1 - y->kr);
    y->kbOneMinusKb = y->kb * (1 - y->kb);
    // reciprocal forms, usable only when every divisor is on the verified list (exactdiv.h)
    y->rcpRangeY = reciprocalOf(y->rangeY, 1.0f);
    y->rcpRangeUV = reciprocalOf(y->rangeUV, 1.0f);
    y->rcpKgTimes2 = reciprocalOf(y->kg, 2.0f);
    y->rcpMax = reciprocalOf((float)y->maxv, 1.0f);
    y->rcpCbDen = reciprocalOf(y->twoOneMinusKb, 1.0f);
    y->rcpCrDen = reciprocalOf(y->twoOneMinusKr, 1.0f);
    y->exactDivEncode = (y->mode == MODE_COEFF && verifiedChromaDenominator(y->twoOneMinusKb) && verifiedChromaDenominator(y->twoOneMinusKr) &&
                         verifiedIntegerDivisor(r->maxf))
                            ? 1
                            : 0;
    y->exactDiv = (verifiedIntegerDivisor(y->rangeY) && verifiedIntegerDivisor(y->rangeUV) && verifiedIntegerDivisor((float)y->maxv) &&
                   verifiedIntegerDivisor(r->maxf) && (y->mode != MODE_COEFF || verifiedKgDivisor(y->kg)))
                      ? 1
                      : 0;
    return true;
}


// ---- the integer path: what libavif hands to libyuv (src/reformat_libyuv.c) ---------------------------------------

namespace {

// YuvConstants as libavif selects them (src/reformat_libyuv.c:775-904), numbers per SURVEY.md appendix D.1
bool selectFixedPointMatrix(const avifImage * image, FixedPointMatrix * out)
{
    unsigned mc = image->matrixCoefficients;
    if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400 && mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY)
        mc = AVIF_MATRIX_COEFFICIENTS_BT601; // :777-781
    enum { NONE, BT709, BT601, BT2020 } family = NONE;
    if (mc == AVIF_MATRIX_COEFFICIENTS_BT709) {
        family = BT709;
    } else if (mc == AVIF_MATRIX_COEFFICIENTS_BT470BG || mc == AVIF_MATRIX_COEFFICIENTS_BT601 || mc == AVIF_MATRIX_COEFFICIENTS_UNSPECIFIED) {
        family = BT601;
    } else if (mc == AVIF_MATRIX_COEFFICIENTS_BT2020_NCL) {
        family = BT2020;
    } else if (mc == AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL) {
        const unsigned cp = image->colorPrimaries;
        family = (cp == 1 || cp == 2) ? BT709 : (cp == 5 || cp == 6) ? BT601 : (cp == 9) ? BT2020 : NONE;
    }
    const bool full = image->yuvRange == AVIF_RANGE_FULL;
    static const FixedPointMatrix limited[3] = { { 18997, -1160, 128, 14, 34, 115 }, { 18997, -1160, 128, 25, 52, 102 }, { 19003, -1160, 128, 12, 42, 107 } };
    static const FixedPointMatrix fullRange[3] = { { 16320, 32, 119, 12, 30, 101 }, { 16320, 32, 113, 22, 46, 90 }, { 16320, 32, 120, 11, 37, 94 } };
    if (family == NONE)
        return false;
    *out = (full ? fullRange : limited)[(int)family - 1];
    return true;
}

// libyuv entry points per RGB layout as bit sets over avifPixelFormat (the lookup tables of src/reformat_libyuv.c:551-712)
constexpr uint8_t Y444 = 1u << AVIF_PIXEL_FORMAT_YUV444, Y422 = 1u << AVIF_PIXEL_FORMAT_YUV422, Y420 = 1u << AVIF_PIXEL_FORMAT_YUV420;
struct FxEntries
{
    uint8_t filter8, alphaFilter8, matrix8, alphaMatrix8; // 8-bit planes: *MatrixFilter, *AlphaTo*MatrixFilter, *Matrix, *AlphaTo*Matrix
    uint8_t filter10, matrix10;                           // 10-bit planes (each has its alpha twin)
    uint8_t matrix12;                                     // 12-bit planes (no alpha twin)
    uint8_t mono;                                         // I400ToARGBMatrix
};
constexpr FxEntries kRgb24 = { Y422 | Y420, 0, Y444 | Y420, 0, 0, 0, 0, 0 };
constexpr FxEntries kArgbWord = { Y422 | Y420, Y422 | Y420, Y444 | Y422 | Y420, Y444 | Y422 | Y420, Y422 | Y420, Y444 | Y422 | Y420, Y420, 1 };
constexpr FxEntries kRgbaWord = { 0, 0, Y422 | Y420, 0, 0, 0, 0, 0 };
const FxEntries * fxEntriesFor(int format)
{
    switch (format) {
        case AVIF_RGB_FORMAT_RGB:
        case AVIF_RGB_FORMAT_BGR: return &kRgb24;      // libyuv "RGB24"/"RAW"
        case AVIF_RGB_FORMAT_RGBA:
        case AVIF_RGB_FORMAT_BGRA: return &kArgbWord;  // libyuv "ARGB"/"ABGR" (word order)
        case AVIF_RGB_FORMAT_ARGB:
        case AVIF_RGB_FORMAT_ABGR:
        case AVIF_RGB_FORMAT_RGB_565: return &kRgbaWord; // libyuv "RGBA"/"BGRA" and RGB565: nearest-only entries
        default: return nullptr;                        // gray layouts: no entry
    }
}

struct FxRoute
{
    bool mono, filter, alpha;
    int native; // 8, 10, 12
};

// getLibYUVConversionFunction, src/reformat_libyuv.c:714-772
bool selectFixedPointRoute(int yuvFormat, int depth, const avifRGBImage * rgb, bool alphaPreferred, FxRoute * r)
{
    const FxEntries * e = fxEntriesFor((int)rgb->format);
    if (!e)
        return false;
    const uint8_t bit = (uint8_t)(1u << yuvFormat);
    const bool nearestOk = rgb->chromaUpsampling != AVIF_CHROMA_UPSAMPLING_BILINEAR && rgb->chromaUpsampling != AVIF_CHROMA_UPSAMPLING_BEST_QUALITY;
    *r = FxRoute { false, false, false, 8 };
    if (depth > 8) {
        if (yuvFormat != AVIF_PIXEL_FORMAT_YUV444 && depth == 10 && (e->filter10 & bit)) {
            *r = FxRoute { false, true, alphaPreferred, 10 };
            return true;
        }
        if (yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || nearestOk) {
            if (depth == 10 && (e->matrix10 & bit)) {
                *r = FxRoute { false, false, alphaPreferred, 10 };
                return true;
            }
            if (depth == 12 && (e->matrix12 & bit)) {
                *r = FxRoute { false, false, false, 12 };
                return true;
            }
        }
        // no high-bit-depth entry: an 8-bit one after a downshift, :743-745
    }
    if (yuvFormat == AVIF_PIXEL_FORMAT_YUV400) {
        r->mono = true;
        return e->mono != 0;
    }
    if (yuvFormat != AVIF_PIXEL_FORMAT_YUV444) {
        if (alphaPreferred && (e->alphaFilter8 & bit)) {
            r->filter = r->alpha = true;
            return true;
        }
        if (e->filter8 & bit) {
            r->filter = true;
            return true;
        }
        if (!nearestOk)
            return false;
    }
    if (alphaPreferred && (e->alphaMatrix8 & bit)) {
        r->alpha = true;
        return true;
    }
    return (e->matrix8 & bit) != 0;
}

// which (range, RGB layout, YUV layout) libyuv converts to YUV for libavif, src/reformat_libyuv.c:293-377
bool fixedPointRgbToYuvCovered(const avifImage * image, const avifRGBImage * rgb)
{
    if (image->depth != 8 || rgb->depth != 8)
        return false;
    if (image->matrixCoefficients != AVIF_MATRIX_COEFFICIENTS_BT470BG && image->matrixCoefficients != AVIF_MATRIX_COEFFICIENTS_BT601)
        return false;
    const int f = (int)rgb->format, yf = (int)image->yuvFormat;
    if (f < AVIF_RGB_FORMAT_RGB || f > AVIF_RGB_FORMAT_ABGR)
        return false;
    const bool full = image->yuvRange == AVIF_RANGE_FULL;
    if (yf == AVIF_PIXEL_FORMAT_YUV400)
        return full ? (f != AVIF_RGB_FORMAT_ARGB) : (f == AVIF_RGB_FORMAT_BGRA);
    if (yf != AVIF_PIXEL_FORMAT_YUV444 && yf != AVIF_PIXEL_FORMAT_YUV422 && yf != AVIF_PIXEL_FORMAT_YUV420)
        return false;
    if (!full || f == AVIF_RGB_FORMAT_RGB)
        return true;
    return yf != AVIF_PIXEL_FORMAT_YUV444; // no full-range 4:4:4 entry for the other layouts, :352-358
}

bool attenuateCovered(const avifRGBImage * rgb) // src/reformat_libyuv.c:1120-1133
{
    return rgb->depth == 8 && (rgb->format == AVIF_RGB_FORMAT_RGBA || rgb->format == AVIF_RGB_FORMAT_BGRA);
}

} // namespace

// NOTE for whoever adds an input here (or to prepareState): rebindYuvToRgbPlan below lists, by hand, every field of the two images a plan is
// derived from -- a field missing there makes tiles 1 .. N-1 of a batch inherit tile 0's plan.  tests/test_host_plans.py
// (test_rebound_plans_equal_plans_made_from_scratch) compares the two byte for byte over random jobs; extend its generator as well.
avifResult makeYuvToRgbPlan(const avifImage * image, const avifRGBImage * rgb, const avifCropRect * rect, int arithMode, uint32_t tuning, YuvToRgbPlan * out,
                            bool colorOnly, bool reformatAlphaHook)
{
    if (!image->yuvPlanes[AVIF_CHAN_Y] || rgb->maxThreads < 0)
        return AVIF_RESULT_REFORMAT_FAILED; // src/reformat.c:1653-1655
    memset(out, 0, sizeof(*out));
    if (!prepareState(image, rgb, &out->yuv, &out->rgb))
        return AVIF_RESULT_REFORMAT_FAILED; // :1657-1660
    out->canvasW = image->width;
    out->canvasH = image->height;
    if (rect) {
        if (rect->width > image->width || rect->height > image->height || rect->x > image->width - rect->width ||
            rect->y > image->height - rect->height)
            return AVIF_RESULT_INVALID_ARGUMENT;
        if (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400 && ((rect->x & out->yuv.shiftX) || (rect->y & out->yuv.shiftY)))
            return AVIF_RESULT_INVALID_ARGUMENT; // src/avif.c:335-337
        out->x0 = rect->x, out->y0 = rect->y, out->w = rect->width, out->h = rect->height;
    } else {
        out->x0 = 0, out->y0 = 0, out->w = image->width, out->h = image->height;
    }

    out->cwinX0 = 0, out->cwinY0 = 0;
    out->cwinX1 = (int32_t)(((uint64_t)image->width + out->yuv.shiftX) >> out->yuv.shiftX) - 1;
    out->cwinY1 = (int32_t)(((uint64_t)image->height + ((image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0)) >> ((image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0)) - 1;

    // alpha multiply mode, src/reformat.c:1662-1677
    const bool rgbHasAlpha = out->rgb.hasAlpha != 0;
    int mul = MUL_NONE;
    if (image->alphaPlane) {
        if (!rgbHasAlpha || rgb->ignoreAlpha) {
            if (!image->alphaPremultiplied)
                mul = MUL_MULTIPLY;
        } else if (!image->alphaPremultiplied && rgb->alphaPremultiplied) {
            mul = MUL_MULTIPLY;
        } else if (image->alphaPremultiplied && !rgb->alphaPremultiplied) {
            mul = MUL_UNMULTIPLY;
        }
    }
    const int mulOfTheCall = mul; // what avifImageYUVToRGB derived before it reached the hook
    out->mulOfTheCall = mul;
    if (colorOnly) {
        mul = MUL_NONE;        // src/reformat.c:1574-1585 stays with the caller
        out->rgb.isFloat = 0;  // and so does src/reformat.c:1588-1590
    }
    // alpha channel, src/reformat.c:1449-1486
    const bool reformatAlpha = colorOnly ? (rgbHasAlpha && reformatAlphaHook) : (rgbHasAlpha && (!rgb->ignoreAlpha || mul != MUL_NONE));
    out->alphaSource = ALPHA_KEEP;
    if (reformatAlpha)
        out->alphaSource = (image->alphaPlane && image->alphaRowBytes) ? ALPHA_PLANE : ALPHA_FILL;

    // which of the reference's loops runs decides where (un)premultiply rounds, src/reformat.c:1494-1567
    const bool nearest =
        (rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_FASTEST || rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_NEAREST);
    out->bilinear = nearest ? 0 : 1;
    bool fast = false;
    if (!out->rgb.isGray && (!out->yuv.hasColor || image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || nearest) &&
        (mul == MUL_NONE || rgbHasAlpha)) {
        if (out->yuv.mode == MODE_IDENTITY) {
            if (image->depth == 8 && rgb->depth == 8 && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 &&
                image->yuvRange == AVIF_RANGE_FULL) {
                fast = true;
                out->identityCopy = 1;
            }
        } else if (out->yuv.mode == MODE_COEFF) {
            fast = true;
        }
    }
    out->inLoopMul = fast ? MUL_NONE : mul;
    out->postMul = fast ? mul : MUL_NONE;
    out->arith = ARITH_FLOAT;
    out->tuning = tuning;

    // A libavif built with libyuv asks libyuv first, src/reformat.c:1453-1462 (the hook is that very call).
    const bool askLibyuv = arithMode != AVIFHIP_ARITHMETIC_FLOAT && (arithMode == AVIFHIP_ARITHMETIC_LIBYUV || !rgb->avoidLibYUV || colorOnly) &&
                           (mul == MUL_NONE || rgbHasAlpha);
    FxRoute route;
    const bool hasAlphaPlane = image->alphaPlane && image->alphaRowBytes;
    if (askLibyuv && rgb->depth == 8 && (image->depth == 8 || image->depth == 10 || image->depth == 12) && // src/reformat_libyuv.c:939
        selectFixedPointMatrix(image, &out->fx) &&
        selectFixedPointRoute((int)image->yuvFormat, (int)image->depth, rgb, reformatAlpha && hasAlphaPlane, &route)) {
        out->arith = ARITH_LIBYUV;
        out->identityCopy = 0;
        out->inLoopMul = MUL_NONE; // libyuv never (un)multiplies (attenuate = 0): always the post-pass, src/reformat.c:1574-1585
        out->postMul = mul;
        out->bilinear = (route.filter && !nearest) ? 1 : 0; // src/reformat_libyuv.c:965-968; plain *Matrix entries are nearest
        out->fxNative = route.native;
        out->fxDownshift = (image->depth > 8 && route.native == 8) ? (int)image->depth - 8 : 0; // :906-930
        out->fxMono = route.mono ? 1 : 0;
        if (!rgbHasAlpha) {
            out->alphaSource = ALPHA_KEEP; // RGB / BGR / 565: there is no A byte
        } else if (route.alpha) {
            out->alphaSource = ALPHA_PLANE;
            out->fxAlpha = FXA_SHIFT;
            out->fxAlphaShift = (route.native == 10) ? 2 : out->fxDownshift;
        } else if (hasAlphaPlane && reformatAlpha) {
            out->alphaSource = ALPHA_PLANE; // libyuv's 255, then avifReformatAlpha over it, src/reformat.c:1464-1486
            out->fxAlpha = FXA_FLOAT;
        } else {
            out->alphaSource = ALPHA_FILL; // libyuv's 255, even under rgb->ignoreAlpha
            out->fxAlpha = FXA_OPAQUE;
        }
    }
    if (colorOnly && arithMode != AVIFHIP_ARITHMETIC_FLOAT && out->arith != ARITH_LIBYUV && mulOfTheCall != MUL_NONE) {
        // libyuv has no entry here, so a stock libavif converts with its own loops -- and if that is the slow loop, the
        // pending (un)multiply happens INSIDE it in fp32 (src/reformat.c:894-947,1563-1566), which a hook cannot reproduce
        // (after AVIF_RESULT_OK libavif runs the integer post-pass).  Decline: libavif's CPU code keeps the result identical.
        bool fastOfTheCall = false;
        if (!out->rgb.isGray && (!out->yuv.hasColor || image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || nearest) && rgbHasAlpha)
            fastOfTheCall = (out->yuv.mode == MODE_COEFF) || out->identityCopy;
        if (!fastOfTheCall)
            return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    // avifRGBImagePremultiplyAlpha / Unpremultiply ask libyuv whatever avoidLibYUV says, src/alpha.c:163,350
    out->postMulFx = (arithMode != AVIFHIP_ARITHMETIC_FLOAT && out->postMul != MUL_NONE && attenuateCovered(rgb)) ? 1 : 0;
    return AVIF_RESULT_OK;
}

bool rebindYuvToRgbPlan(const YuvToRgbPlan & proto, const avifImage * pi, const avifRGBImage * pr, const avifImage * image, const avifRGBImage * rgb,
                        const avifCropRect * rect, YuvToRgbPlan * out, avifResult * result)
{
    // everything makeYuvToRgbPlan reads, except buffer addresses, pitches and the rectangle
    if (image->width != pi->width || image->height != pi->height || image->depth != pi->depth || image->yuvFormat != pi->yuvFormat ||
        image->yuvRange != pi->yuvRange || image->colorPrimaries != pi->colorPrimaries || image->transferCharacteristics != pi->transferCharacteristics ||
        image->matrixCoefficients != pi->matrixCoefficients || image->alphaPremultiplied != pi->alphaPremultiplied)
        return false;
    for (int p = 0; p < 3; ++p)
        if ((image->yuvPlanes[p] != nullptr) != (pi->yuvPlanes[p] != nullptr) || (image->yuvRowBytes[p] != 0) != (pi->yuvRowBytes[p] != 0))
            return false;
    if ((image->alphaPlane != nullptr) != (pi->alphaPlane != nullptr) || (image->alphaRowBytes != 0) != (pi->alphaRowBytes != 0))
        return false;
    if (rgb != pr &&
        (rgb->width != pr->width || rgb->height != pr->height || rgb->depth != pr->depth || rgb->format != pr->format || rgb->chromaUpsampling != pr->chromaUpsampling ||
         rgb->chromaDownsampling != pr->chromaDownsampling || rgb->avoidLibYUV != pr->avoidLibYUV || rgb->ignoreAlpha != pr->ignoreAlpha ||
         rgb->alphaPremultiplied != pr->alphaPremultiplied || rgb->isFloat != pr->isFloat || rgb->maxThreads != pr->maxThreads || (rgb->pixels != nullptr) != (pr->pixels != nullptr)))
        return false;
    *out = proto;
    for (int p = 0; p < 3; ++p) {
        out->yuv.plane[p] = image->yuvPlanes[p];
        out->yuv.rowBytes[p] = image->yuvRowBytes[p];
    }
    out->yuv.alpha = image->alphaPlane;
    out->yuv.alphaRowBytes = image->alphaRowBytes;
    out->rgb.pixels = rgb->pixels;
    out->rgb.rowBytes = rgb->rowBytes;
    *result = AVIF_RESULT_OK;
    if (rect) {
        if (rect->width > image->width || rect->height > image->height || rect->x > image->width - rect->width || rect->y > image->height - rect->height ||
            (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400 && ((rect->x & out->yuv.shiftX) || (rect->y & out->yuv.shiftY))))
            *result = AVIF_RESULT_INVALID_ARGUMENT;
        out->x0 = rect->x, out->y0 = rect->y, out->w = rect->width, out->h = rect->height;
    } else {
        out->x0 = 0, out->y0 = 0, out->w = image->width, out->h = image->height;
    }
    return true;
}

avifResult makeRgbToYuvPlan(const avifImage * image, const avifRGBImage * rgb, int arithMode, RgbToYuvPlan * out)
{
    if (!rgb->pixels || rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return AVIF_RESULT_REFORMAT_FAILED; // src/reformat.c:223-225
    memset(out, 0, sizeof(*out));
    if (!prepareState(image, rgb, &out->yuv, &out->rgb))
        return AVIF_RESULT_REFORMAT_FAILED;
    if (rgb->isFloat)
        return AVIF_RESULT_NOT_IMPLEMENTED; // :232-234
    out->width = image->width;
    out->height = image->height;
    out->rx0 = 0, out->ry0 = 0, out->rw = image->width, out->rh = image->height;
    const bool hasAlpha = out->rgb.hasAlpha && !rgb->ignoreAlpha;
    out->mul = MUL_NONE; // :242-249
    if (hasAlpha) {
        if (!rgb->alphaPremultiplied && image->alphaPremultiplied)
            out->mul = MUL_MULTIPLY;
        else if (rgb->alphaPremultiplied && !image->alphaPremultiplied)
            out->mul = MUL_UNMULTIPLY;
    }
    out->arith = ARITH_FLOAT;
    // src/reformat.c:264-272: libyuv is asked unless the source is gray, an alpha (un)multiply is pending or avoidLibYUV
    if (arithMode != AVIFHIP_ARITHMETIC_FLOAT && (arithMode == AVIFHIP_ARITHMETIC_LIBYUV || !rgb->avoidLibYUV) && !out->rgb.isGray &&
        out->mul == MUL_NONE && fixedPointRgbToYuvCovered(image, rgb)) {
        out->arith = ARITH_LIBYUV;
        out->fxFullRange = (image->yuvRange == AVIF_RANGE_FULL) ? 1 : 0;
    }
    return AVIF_RESULT_OK;
}

avifResult makeAlphaMulPlan(const avifRGBImage * rgb, bool unmultiply, int arithMode, AlphaMulPlan * out)
{
    if (!rgb->pixels || !rgb->rowBytes)
        return AVIF_RESULT_REFORMAT_FAILED; // src/alpha.c:154-156, :341-343
    if (!rgbFormatHasAlpha((int)rgb->format))
        return unmultiply ? AVIF_RESULT_REFORMAT_FAILED : AVIF_RESULT_INVALID_ARGUMENT; // :346-348 / :159-161
    memset(out, 0, sizeof(*out));
    if (!fillRgbSide(rgb, &out->rgb))
        return AVIF_RESULT_NOT_IMPLEMENTED; // the reference asserts depth in [8,16]
    out->width = rgb->width;
    out->height = rgb->height;
    out->unmultiply = unmultiply ? 1 : 0;
    out->arith = (arithMode != AVIFHIP_ARITHMETIC_FLOAT && attenuateCovered(rgb)) ? ARITH_LIBYUV : ARITH_FLOAT; // src/alpha.c:163,350
    out->exactDiv = verifiedIntegerDivisor(out->rgb.maxf) ? 1 : 0;
    return AVIF_RESULT_OK;
}

} // namespace avifhip
