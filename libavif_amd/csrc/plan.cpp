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
    y->exactDiv = (verifiedIntegerDivisor(y->rangeY) && verifiedIntegerDivisor(y->rangeUV) && verifiedIntegerDivisor((float)y->maxv) &&
                   verifiedIntegerDivisor(r->maxf) && (y->mode != MODE_COEFF || verifiedKgDivisor(y->kg)))
                      ? 1
                      : 0;
    return true;
}

avifResult makeYuvToRgbPlan(const avifImage * image, const avifRGBImage * rgb, const avifCropRect * rect, int arithMode, uint32_t tuning, YuvToRgbPlan * out,
                            bool colorOnly, bool reformatAlphaHook)
{
    (void)arithMode;
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
    return AVIF_RESULT_OK;
}

avifResult makeRgbToYuvPlan(const avifImage * image, const avifRGBImage * rgb, int arithMode, RgbToYuvPlan * out)
{
    (void)arithMode;
    if (!rgb->pixels || rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return AVIF_RESULT_REFORMAT_FAILED; // src/reformat.c:223-225
    memset(out, 0, sizeof(*out));
    if (!prepareState(image, rgb, &out->yuv, &out->rgb))
        return AVIF_RESULT_REFORMAT_FAILED;
    if (rgb->isFloat)
        return AVIF_RESULT_NOT_IMPLEMENTED; // :232-234
    out->width = image->width;
    out->height = image->height;
    const bool hasAlpha = out->rgb.hasAlpha && !rgb->ignoreAlpha;
    out->mul = MUL_NONE; // :242-249
    if (hasAlpha) {
        if (!rgb->alphaPremultiplied && image->alphaPremultiplied)
            out->mul = MUL_MULTIPLY;
        else if (rgb->alphaPremultiplied && !image->alphaPremultiplied)
            out->mul = MUL_UNMULTIPLY;
    }
    out->arith = ARITH_FLOAT;
    return AVIF_RESULT_OK;
}

avifResult makeAlphaMulPlan(const avifRGBImage * rgb, bool unmultiply, int arithMode, AlphaMulPlan * out)
{
    (void)arithMode;
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
    out->arith = ARITH_FLOAT;
    return AVIF_RESULT_OK;
}

} // namespace avifhip
