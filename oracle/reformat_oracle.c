/*
 * reformat_oracle.c -- TEST INFRASTRUCTURE (the checker), NOT PRODUCT CODE.
 *
 * Plain-C restatement of the floating-point pixel-reformat path of libavif
 * v1.4.2-devel (/root/reference/src/reformat.c, src/alpha.c, src/colr.c).
 * It is organised differently from the reference (one generic per-pixel
 * routine + an explicit "plan" instead of nine specialised loops) but performs
 * every fp32 operation in the reference's order so results are byte-identical.
 * Each function cites the reference lines it restates.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares every entry
 * point with oracle/_ref/libavif_ref.so (the reference compiled from its own
 * sources by oracle/Makefile) over the configuration sweep, and against the
 * known-answer facts of the reference's own tests.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off (no -mfma / -ffast-math): binary32
 * arithmetic, round-to-nearest-even, no fused multiply-add.
 */
#include "reformat_oracle.h"
#include "oracle_backend.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */

static float clampf01(float x) /* AVIF_CLAMP(x,0,1): include/avif/internal.h:18 */
{
    return (x < 0.0f) ? 0.0f : ((1.0f < x) ? 1.0f : x);
}
static int clampi(int x, int lo, int hi)
{
    return (x < lo) ? lo : ((hi < x) ? hi : x);
}
static float roundHalfUp(float v) /* avifRoundf, src/utils.c:11-14 */
{
    return floorf(v + 0.5f);
}
static unsigned load16(const uint8_t * p)
{
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
static void store16(uint8_t * p, unsigned v)
{
    uint16_t w = (uint16_t)v;
    memcpy(p, &w, 2);
}

static int fmtHasAlpha(avifRGBFormat f) /* src/avif.c:675-679 */
{
    return f == AVIF_RGB_FORMAT_RGBA || f == AVIF_RGB_FORMAT_ARGB || f == AVIF_RGB_FORMAT_BGRA || f == AVIF_RGB_FORMAT_ABGR ||
           f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY;
}
static int fmtIsGray(avifRGBFormat f) /* src/avif.c:670-673 */
{
    return f == AVIF_RGB_FORMAT_GRAY || f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY;
}
static int fmtChannels(avifRGBFormat f) /* src/avif.c:681-690 */
{
    if (f == AVIF_RGB_FORMAT_GRAY)
        return 1;
    if (f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY)
        return 2;
    return fmtHasAlpha(f) ? 4 : 3;
}

/* ------------------------------------------------------------------------- */
/* plan: everything derived from (avifImage, avifRGBImage) before the loops   */

enum { MODE_COEFF = 0, MODE_IDENTITY, MODE_YCGCO, MODE_YCGCO_RE, MODE_YCGCO_RO };
enum { MUL_NONE = 0, MUL_MULTIPLY, MUL_UNMULTIPLY };

typedef struct RgbLayout
{
    int chanBytes, pixBytes;
    int offR, offG, offB, offA, offGray;
    int maxv;
    float maxf;
} RgbLayout;

typedef struct YuvSpace
{
    float kr, kg, kb;
    int chanBytes, depth, maxv;
    int limited;
    float biasY, biasUV, rangeY, rangeUV;
    int shiftX, shiftY, mono;
    int mode;
} YuvSpace;

/* src/reformat.c:32-117 (avifGetRGBColorSpaceInfo) */
static int describeRgb(const avifRGBImage * rgb, RgbLayout * L)
{
    const uint32_t d = rgb->depth;
    if (!(d == 8 || d == 10 || d == 12 || d == 16))
        return 0;
    if (rgb->isFloat && d != 16)
        return 0;
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565 && d != 8)
        return 0;
    if ((int)rgb->format < AVIF_RGB_FORMAT_RGB || rgb->format >= AVIF_RGB_FORMAT_COUNT)
        return 0;
    const int cb = (d > 8) ? 2 : 1;
    memset(L, 0, sizeof(*L));
    L->chanBytes = cb;
    L->pixBytes = (rgb->format == AVIF_RGB_FORMAT_RGB_565) ? 2 : fmtChannels(rgb->format) * cb; /* src/avif.c:692-698 */
    /* channel order per format, in units of one channel */
    static const signed char order[AVIF_RGB_FORMAT_COUNT][5] = {
        /*            R  G  B  A  Gray */
        /* RGB   */ { 0, 1, 2, 0, 0 },
        /* RGBA  */ { 0, 1, 2, 3, 0 },
        /* ARGB  */ { 1, 2, 3, 0, 0 },
        /* BGR   */ { 2, 1, 0, 0, 0 },
        /* BGRA  */ { 2, 1, 0, 3, 0 },
        /* ABGR  */ { 3, 2, 1, 0, 0 },
        /* 565   */ { 0, 0, 0, 0, 0 },
        /* GRAY  */ { 0, 0, 0, 0, 0 },
        /* GRAYA */ { 0, 0, 0, 1, 0 },
        /* AGRAY */ { 0, 0, 0, 0, 1 },
    };
    const signed char * o = order[rgb->format];
    L->offR = o[0] * cb;
    L->offG = o[1] * cb;
    L->offB = o[2] * cb;
    L->offA = o[3] * cb;
    L->offGray = o[4] * cb;
    L->maxv = (1 << d) - 1;
    L->maxf = (float)L->maxv;
    return 1;
}

/* src/colr.c:16-29 primaries table, :517-542 avifColorPrimariesComputeYCoeffs */
static void primariesYCoeffs(unsigned cp, float * krOut, float * kbOut)
{
    static const struct
    {
        unsigned cp;
        float p[8];
    } T[] = {
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
    const float * p = T[0].p; /* unknown primaries fall back to BT.709: src/colr.c:43-44 */
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); ++i) {
        if (T[i].cp == cp) {
            p = T[i].p;
            break;
        }
    }
    const float rX = p[0], rY = p[1], gX = p[2], gY = p[3], bX = p[4], bY = p[5], wX = p[6], wY = p[7];
    const float rZ = 1.0f - (rX + rY);
    const float gZ = 1.0f - (gX + gY);
    const float bZ = 1.0f - (bX + bY);
    const float wZ = 1.0f - (wX + wY);
    *krOut = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) /
             (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    *kbOut = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) /
             (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
}

/* src/colr.c:123-189 (matrixCoefficientsTables, avifCalcYUVCoefficients) */
static void lumaCoefficients(const avifImage * image, float * kr, float * kg, float * kb)
{
    float r = 0.299f, b = 0.114f; /* default: BT.601 */
    float g = 1.0f - r - b;
    int found = 1;
    switch (image->matrixCoefficients) {
        case AVIF_MATRIX_COEFFICIENTS_BT709:
            r = 0.2126f, b = 0.0722f;
            break;
        case AVIF_MATRIX_COEFFICIENTS_FCC:
            r = 0.30f, b = 0.11f;
            break;
        case AVIF_MATRIX_COEFFICIENTS_BT470BG:
        case AVIF_MATRIX_COEFFICIENTS_BT601:
            r = 0.299f, b = 0.114f;
            break;
        case AVIF_MATRIX_COEFFICIENTS_SMPTE240:
            r = 0.212f, b = 0.087f;
            break;
        case AVIF_MATRIX_COEFFICIENTS_BT2020_NCL:
            r = 0.2627f, b = 0.0593f;
            break;
        case AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL:
            primariesYCoeffs(image->colorPrimaries, &r, &b);
            break;
        default:
            found = 0;
            break;
    }
    if (found)
        g = 1.0f - r - b;
    *kr = r;
    *kg = g;
    *kb = b;
}

/* src/reformat.c:119-159 (avifGetYUVColorSpaceInfo) + src/avif.c:39-72 */
static int describeYuv(const avifImage * image, YuvSpace * S)
{
    const uint32_t d = image->depth;
    if (!(d == 8 || d == 10 || d == 12 || d == 16))
        return 0;
    if (image->yuvFormat < AVIF_PIXEL_FORMAT_YUV444 || image->yuvFormat >= AVIF_PIXEL_FORMAT_COUNT)
        return 0;
    if (image->yuvRange != AVIF_RANGE_LIMITED && image->yuvRange != AVIF_RANGE_FULL)
        return 0;
    const unsigned mc = image->matrixCoefficients;
    const int ycgcoFamily = (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO || mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE ||
                             mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO);
    if (mc == 3 || (ycgcoFamily && image->yuvRange == AVIF_RANGE_LIMITED) || mc == AVIF_MATRIX_COEFFICIENTS_BT2020_CL ||
        mc == AVIF_MATRIX_COEFFICIENTS_SMPTE2085 || mc == AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_CL ||
        mc == AVIF_MATRIX_COEFFICIENTS_ICTCP || mc >= AVIF_MATRIX_COEFFICIENTS_LAST)
        return 0;
    if (mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY && image->yuvFormat != AVIF_PIXEL_FORMAT_YUV444 &&
        image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400)
        return 0;
    memset(S, 0, sizeof(*S));
    switch (image->yuvFormat) {
        case AVIF_PIXEL_FORMAT_YUV422:
            S->shiftX = 1;
            break;
        case AVIF_PIXEL_FORMAT_YUV420:
            S->shiftX = 1, S->shiftY = 1;
            break;
        case AVIF_PIXEL_FORMAT_YUV400:
            S->shiftX = 1, S->shiftY = 1, S->mono = 1;
            break;
        default:
            break;
    }
    lumaCoefficients(image, &S->kr, &S->kg, &S->kb);
    S->chanBytes = (d > 8) ? 2 : 1;
    S->depth = (int)d;
    S->maxv = (1 << d) - 1;
    S->limited = (image->yuvRange == AVIF_RANGE_LIMITED);
    S->biasY = S->limited ? (float)(16 << (d - 8)) : 0.0f;
    S->biasUV = (float)(1 << (d - 1));
    S->rangeY = (float)(S->limited ? (219 << (d - 8)) : S->maxv);
    S->rangeUV = (float)(S->limited ? (224 << (d - 8)) : S->maxv);
    return 1;
}

/* src/reformat.c:161-194 (avifPrepareReformatState) */
static int makePlan(const avifImage * image, const avifRGBImage * rgb, RgbLayout * L, YuvSpace * S)
{
    const unsigned mc = image->matrixCoefficients;
    if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE || mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO) {
        const int bitOffset = (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE) ? 2 : 1;
        if ((int)image->depth - bitOffset != (int)rgb->depth)
            return 0;
    }
    if (!describeRgb(rgb, L) || !describeYuv(image, S))
        return 0;
    S->mode = MODE_COEFF;
    if (mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY)
        S->mode = MODE_IDENTITY;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO)
        S->mode = MODE_YCGCO;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RE)
        S->mode = MODE_YCGCO_RE;
    else if (mc == AVIF_MATRIX_COEFFICIENTS_YCGCO_RO)
        S->mode = MODE_YCGCO_RO;
    if (S->mode != MODE_COEFF)
        S->kr = S->kg = S->kb = 0.0f;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* alpha channel: fill / copy / rescale      (src/alpha.c:9-149)              */

typedef struct AlphaJob
{
    uint32_t width, height;
    int srcDepth, srcPixBytes;
    const uint8_t * src; /* NULL => fill with opaque */
    size_t srcRowBytes;
    int dstDepth, dstPixBytes;
    uint8_t * dst;
    size_t dstRowBytes;
} AlphaJob;

static void runAlphaJob(const AlphaJob * J)
{
    const int dstMax = (1 << J->dstDepth) - 1;
    const float srcMaxF = (float)((1 << J->srcDepth) - 1);
    const float dstMaxF = (float)dstMax;
    for (uint32_t j = 0; j < J->height; ++j) {
        const uint8_t * s = J->src ? J->src + (size_t)j * J->srcRowBytes : NULL;
        uint8_t * d = J->dst + (size_t)j * J->dstRowBytes;
        for (uint32_t i = 0; i < J->width; ++i, d += J->dstPixBytes) {
            int a;
            if (!s) {
                a = dstMax; /* avifFillAlpha, :9-35 */
            } else {
                const int sa = (J->srcDepth > 8) ? (int)load16(s) : (int)*s;
                s += J->srcPixBytes;
                if (J->srcDepth == J->dstDepth) {
                    a = sa; /* plain strided copy (no clamp), :44-79 */
                } else {
                    const float alphaF = (float)sa / srcMaxF; /* :93-96 */
                    a = (int)(0.5f + (alphaF * dstMaxF));
                    a = clampi(a, 0, dstMax);
                }
            }
            if (J->dstDepth > 8)
                store16(d, (unsigned)a);
            else
                *d = (uint8_t)a;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* premultiply / unpremultiply on stored integers   (src/alpha.c:151-535)      */

static avifResult integerAlphaPass(avifRGBImage * rgb, int unmultiply)
{
    if (!rgb->pixels || !rgb->rowBytes)
        return AVIF_RESULT_REFORMAT_FAILED; /* :154-156, :341-343 */
    if (!fmtHasAlpha(rgb->format))
        return unmultiply ? AVIF_RESULT_REFORMAT_FAILED : AVIF_RESULT_INVALID_ARGUMENT; /* :346-348 vs :159-161 */
    const uint32_t maxv = (1u << rgb->depth) - 1;
    const float maxF = (float)maxv;
    const int wide = rgb->depth > 8;
    const int nch = fmtChannels(rgb->format);
    /* alpha first for ARGB/ABGR/AGRAY, last for RGBA/BGRA/GRAYA */
    const int alphaFirst =
        (rgb->format == AVIF_RGB_FORMAT_ARGB || rgb->format == AVIF_RGB_FORMAT_ABGR || rgb->format == AVIF_RGB_FORMAT_AGRAY);
    const int aIdx = alphaFirst ? 0 : nch - 1;
    const int c0 = alphaFirst ? 1 : 0;
    const int cb = wide ? 2 : 1;
    for (uint32_t j = 0; j < rgb->height; ++j) {
        uint8_t * px = rgb->pixels + (size_t)j * rgb->rowBytes;
        for (uint32_t i = 0; i < rgb->width; ++i, px += nch * cb) {
            const uint32_t a = wide ? load16(px + aIdx * cb) : px[aIdx];
            if (a >= maxv)
                continue; /* opaque: untouched */
            for (int c = c0; c < c0 + nch - 1; ++c) {
                uint8_t * p = px + c * cb;
                unsigned out;
                if (a == 0) {
                    out = 0;
                } else {
                    const float v = (float)(wide ? load16(p) : *p);
                    if (!unmultiply) {
                        out = (unsigned)roundHalfUp(v * (float)a / maxF); /* :189 */
                    } else {
                        const float q = roundHalfUp(v * maxF / (float)a); /* :375-380 */
                        out = (unsigned)((q < maxF) ? q : maxF);
                    }
                }
                if (wide)
                    store16(p, out);
                else
                    *p = (uint8_t)out;
            }
        }
    }
    return AVIF_RESULT_OK;
}

/* src/alpha.c:163-166 / :350-353: the backend is asked first, whatever rgb->avoidLibYUV says */
static avifResult alphaPassWithBackend(avifRGBImage * rgb, int unmultiply, const OracleBackend * backend)
{
    if (!rgb->pixels || !rgb->rowBytes)
        return AVIF_RESULT_REFORMAT_FAILED;
    if (!fmtHasAlpha(rgb->format))
        return unmultiply ? AVIF_RESULT_REFORMAT_FAILED : AVIF_RESULT_INVALID_ARGUMENT;
    if (backend) {
        avifResult (*fn)(avifRGBImage *) = unmultiply ? backend->unpremultiply : backend->premultiply;
        if (fn) {
            const avifResult br = fn(rgb);
            if (br != AVIF_RESULT_NOT_IMPLEMENTED)
                return br;
        }
    }
    return integerAlphaPass(rgb, unmultiply);
}

avifResult oracleAlphaPassWithBackend(avifRGBImage * rgb, int unmultiply, const OracleBackend * backend)
{
    return alphaPassWithBackend(rgb, unmultiply, backend);
}

avifResult oracleRGBImagePremultiplyAlpha(avifRGBImage * rgb)
{
    return integerAlphaPass(rgb, 0);
}
avifResult oracleRGBImageUnpremultiplyAlpha(avifRGBImage * rgb)
{
    return integerAlphaPass(rgb, 1);
}

/* ------------------------------------------------------------------------- */
/* YUV -> RGB                                                                */

typedef struct YuvToRgbJob
{
    const avifImage * image; /* canvas */
    avifRGBImage * rgb;      /* canvas */
    RgbLayout L;
    YuvSpace S;
    int hasColor;   /* chroma planes present and format != 400 */
    int bilinear;   /* 4-tap chroma filter (only meaningful for 420/422) */
    int inLoopMul;  /* MUL_* applied in float inside the loop (slow path, :894-947) */
    int identityCopy; /* src/reformat.c:1278-1309 */
    const float * lutY;
    const float * lutUV;
} YuvToRgbJob;

static unsigned readSample(const uint8_t * plane, size_t rowBytes, uint32_t x, uint32_t y, int chanBytes, unsigned maxv)
{
    const uint8_t * p = plane + (size_t)y * rowBytes + (size_t)x * chanBytes;
    if (chanBytes == 1)
        return *p;
    const unsigned v = load16(p);
    return (v < maxv) ? v : maxv; /* "clamp incoming data to protect against bad LUT lookups", :712 */
}

/* One output pixel at canvas coordinates (i,j).  src/reformat.c:703-974, and the
 * specialised loops :980-1407 which perform the same arithmetic. */
static void convertPixel(const YuvToRgbJob * J, uint32_t i, uint32_t j)
{
    const avifImage * im = J->image;
    const YuvSpace * S = &J->S;
    const RgbLayout * L = &J->L;
    uint8_t * dst = J->rgb->pixels + (size_t)j * J->rgb->rowBytes + (size_t)i * L->pixBytes;

    if (J->identityCopy) {
        const uint8_t g = im->yuvPlanes[0][(size_t)j * im->yuvRowBytes[0] + i];
        const uint8_t b = im->yuvPlanes[1][(size_t)j * im->yuvRowBytes[1] + i];
        const uint8_t r = im->yuvPlanes[2][(size_t)j * im->yuvRowBytes[2] + i];
        if (J->rgb->format == AVIF_RGB_FORMAT_RGB_565) {
            store16(dst, (unsigned)((b >> 3) | ((g >> 2) << 5) | ((r >> 3) << 11)));
        } else {
            dst[L->offR] = r;
            dst[L->offG] = g;
            dst[L->offB] = b;
        }
        return;
    }

    const unsigned unormY = readSample(im->yuvPlanes[0], im->yuvRowBytes[0], i, j, S->chanBytes, (unsigned)S->maxv);
    const float Y = J->lutY[unormY];
    float Cb = 0.5f, Cr = 0.5f;

    if (J->hasColor) {
        const uint32_t uvI = i >> S->shiftX;
        const uint32_t uvJ = j >> S->shiftY;
        const uint8_t * planeU = im->yuvPlanes[1];
        const uint8_t * planeV = im->yuvPlanes[2];
        const size_t rbU = im->yuvRowBytes[1], rbV = im->yuvRowBytes[2];
        const unsigned mx = (unsigned)S->maxv;
        if (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || !J->bilinear) {
            Cb = J->lutUV[readSample(planeU, rbU, uvI, uvJ, S->chanBytes, mx)];
            Cr = J->lutUV[readSample(planeV, rbV, uvI, uvJ, S->chanBytes, mx)];
        } else {
            /* neighbour selection, :766-795: the horizontally adjacent chroma sample is the
             * one on the side the luma pixel leans to; duplicated at the image border. */
            int dx, dy;
            if (i == 0 || (i == im->width - 1 && (i & 1)))
                dx = 0;
            else
                dx = (i & 1) ? 1 : -1;
            if (j == 0 || (j == im->height - 1 && (j & 1)) || im->yuvFormat == AVIF_PIXEL_FORMAT_YUV422)
                dy = 0;
            else
                dy = (j & 1) ? 1 : -1;
            const uint32_t xn = (uint32_t)((int)uvI + dx), yn = (uint32_t)((int)uvJ + dy);
            const float u00 = J->lutUV[readSample(planeU, rbU, uvI, uvJ, S->chanBytes, mx)];
            const float u10 = J->lutUV[readSample(planeU, rbU, xn, uvJ, S->chanBytes, mx)];
            const float u01 = J->lutUV[readSample(planeU, rbU, uvI, yn, S->chanBytes, mx)];
            const float u11 = J->lutUV[readSample(planeU, rbU, xn, yn, S->chanBytes, mx)];
            const float v00 = J->lutUV[readSample(planeV, rbV, uvI, uvJ, S->chanBytes, mx)];
            const float v10 = J->lutUV[readSample(planeV, rbV, xn, uvJ, S->chanBytes, mx)];
            const float v01 = J->lutUV[readSample(planeV, rbV, uvI, yn, S->chanBytes, mx)];
            const float v11 = J->lutUV[readSample(planeV, rbV, xn, yn, S->chanBytes, mx)];
            /* :834-837 -- four products, summed left to right */
            Cb = (u00 * (9.0f / 16.0f)) + (u10 * (3.0f / 16.0f)) + (u01 * (3.0f / 16.0f)) + (u11 * (1.0f / 16.0f));
            Cr = (v00 * (9.0f / 16.0f)) + (v10 * (3.0f / 16.0f)) + (v01 * (3.0f / 16.0f)) + (v11 * (1.0f / 16.0f));
        }
    }

    const int rgbHasColor = !fmtIsGray(J->rgb->format);
    float Rc = 0.0f, Gc = 0.0f, Bc = 0.0f, grayc = 0.0f;
    if (rgbHasColor) {
        float R, G, B;
        if (!J->hasColor) {
            R = G = B = Y; /* :878-883 */
        } else if (S->mode == MODE_IDENTITY) {
            G = Y, B = Cb, R = Cr; /* :846-851 */
        } else if (S->mode == MODE_YCGCO) {
            const float t = Y - Cb; /* :855-858 */
            G = Y + Cb;
            B = t - Cr;
            R = t + Cr;
        } else if (S->mode == MODE_YCGCO_RE || S->mode == MODE_YCGCO_RO) {
            const int Cg = (int)roundHalfUp(Cb * (float)S->maxv); /* :862-871 */
            const int Co = (int)roundHalfUp(Cr * (float)S->maxv);
            const int t = (int)unormY - (Cg >> 1);
            G = (float)clampi(t + Cg, 0, L->maxv);
            B = (float)clampi(t - (Co >> 1), 0, L->maxv);
            R = (float)clampi((int)B + Co, 0, L->maxv); /* B is an exact small integer here */
            G /= L->maxf;
            B /= L->maxf;
            R /= L->maxf;
        } else {
            const float kr = S->kr, kg = S->kg, kb = S->kb; /* :874-876 */
            R = Y + (2 * (1 - kr)) * Cr;
            B = Y + (2 * (1 - kb)) * Cb;
            G = Y - ((2 * ((kr * (1 - kr) * Cr) + (kb * (1 - kb) * Cb))) / kg);
        }
        Rc = clampf01(R);
        Gc = clampf01(G);
        Bc = clampf01(B);
    } else {
        grayc = clampf01(Y);
    }

    if (J->inLoopMul != MUL_NONE) { /* :894-947 */
        const unsigned unormA = readSample(im->alphaPlane, im->alphaRowBytes, i, j, S->chanBytes, (unsigned)S->maxv);
        const float Ac = clampf01((float)unormA / ((float)S->maxv));
        if (Ac == 0.0f) {
            Rc = Gc = Bc = grayc = 0.0f;
        } else if (Ac < 1.0f) {
            if (J->inLoopMul == MUL_MULTIPLY) {
                Rc *= Ac, Gc *= Ac, Bc *= Ac, grayc *= Ac;
            } else {
                Rc /= Ac, Gc /= Ac, Bc /= Ac, grayc /= Ac;
                Rc = (Rc < 1.0f) ? Rc : 1.0f;
                Gc = (Gc < 1.0f) ? Gc : 1.0f;
                Bc = (Bc < 1.0f) ? Bc : 1.0f;
                grayc = (grayc < 1.0f) ? grayc : 1.0f;
            }
        }
    }

    /* store, :949-973 / :619-633 */
    if (rgbHasColor) {
        const unsigned r = (unsigned)(0.5f + (Rc * L->maxf));
        const unsigned g = (unsigned)(0.5f + (Gc * L->maxf));
        const unsigned b = (unsigned)(0.5f + (Bc * L->maxf));
        if (J->rgb->format == AVIF_RGB_FORMAT_RGB_565) {
            store16(dst, ((b & 0xff) >> 3) | (((g & 0xff) >> 2) << 5) | (((r & 0xff) >> 3) << 11));
        } else if (L->chanBytes == 1) {
            dst[L->offR] = (uint8_t)r;
            dst[L->offG] = (uint8_t)g;
            dst[L->offB] = (uint8_t)b;
        } else {
            store16(dst + L->offR, r);
            store16(dst + L->offG, g);
            store16(dst + L->offB, b);
        }
    } else {
        const unsigned g = (unsigned)(0.5f + (grayc * L->maxf));
        if (L->chanBytes == 1)
            dst[L->offGray] = (uint8_t)g;
        else
            store16(dst + L->offGray, g);
    }
}

/* src/reformat.c:1411-1443 (avifRGBImageToF16) restricted to a rectangle */
static void toHalfFloat(avifRGBImage * rgb, const avifCropRect * r)
{
    const int nch = fmtChannels(rgb->format);
    const float scale = 1.0f / ((1 << rgb->depth) - 1);
    const float multiplier = 1.9259299444e-34f * scale;
    for (uint32_t j = r->y; j < r->y + r->height; ++j) {
        uint8_t * p = rgb->pixels + (size_t)j * rgb->rowBytes + (size_t)r->x * nch * 2;
        for (uint32_t k = 0; k < r->width * (uint32_t)nch; ++k, p += 2) {
            const float f = (float)load16(p) * multiplier;
            uint32_t bits;
            memcpy(&bits, &f, 4);
            store16(p, bits >> 13);
        }
    }
}

static avifResult yuvToRgbRect(const avifImage * image, avifRGBImage * rgb, const avifCropRect * rect, const OracleBackend * backend)
{
    if (!image->yuvPlanes[AVIF_CHAN_Y] || rgb->maxThreads < 0)
        return AVIF_RESULT_REFORMAT_FAILED; /* :1653-1655 */
    YuvToRgbJob J;
    memset(&J, 0, sizeof(J));
    J.image = image;
    J.rgb = rgb;
    if (!makePlan(image, rgb, &J.L, &J.S))
        return AVIF_RESULT_REFORMAT_FAILED;

    const int rgbHasAlpha = fmtHasAlpha(rgb->format);
    /* alpha multiply mode, :1662-1677 */
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

    /* the accelerated backend (libyuv in the reference) gets the first shot at the colour conversion, :1449-1462 */
    const int reformatAlpha = rgbHasAlpha && (!rgb->ignoreAlpha || mul != MUL_NONE);
    int convertedByBackend = 0;
    avifBool alphaDoneByBackend = AVIF_FALSE;
    if (backend && backend->yuvToRgb && !rgb->avoidLibYUV && (mul == MUL_NONE || rgbHasAlpha) && rect->x == 0 && rect->y == 0 &&
        rect->width == image->width && rect->height == image->height) {
        const avifResult br = backend->yuvToRgb(image, rgb, reformatAlpha ? AVIF_TRUE : AVIF_FALSE, &alphaDoneByBackend);
        if (br == AVIF_RESULT_OK)
            convertedByBackend = 1;
        else if (br != AVIF_RESULT_NOT_IMPLEMENTED)
            return br;
    }

    /* alpha channel, :1464-1486 */
    if (reformatAlpha && !alphaDoneByBackend) {
        AlphaJob A;
        memset(&A, 0, sizeof(A));
        A.width = rect->width;
        A.height = rect->height;
        A.dstDepth = (int)rgb->depth;
        A.dstPixBytes = J.L.pixBytes;
        A.dstRowBytes = rgb->rowBytes;
        A.dst = rgb->pixels + (size_t)rect->y * rgb->rowBytes + (size_t)rect->x * J.L.pixBytes + J.L.offA;
        if (image->alphaPlane && image->alphaRowBytes) {
            A.srcDepth = (int)image->depth;
            A.srcPixBytes = J.S.chanBytes;
            A.srcRowBytes = image->alphaRowBytes;
            A.src = image->alphaPlane + (size_t)rect->y * image->alphaRowBytes + (size_t)rect->x * J.S.chanBytes;
        }
        runAlphaJob(&A);
    }

    /* which loop would the reference run?  :1494-1567 */
    const int nearest =
        (rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_FASTEST || rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_NEAREST);
    J.hasColor = image->yuvPlanes[1] && image->yuvPlanes[2] && image->yuvRowBytes[1] && image->yuvRowBytes[2] &&
                 image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400;
    J.bilinear = !nearest;
    int fast = 0;
    if (!fmtIsGray(rgb->format) && (!J.hasColor || image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || nearest) &&
        (mul == MUL_NONE || rgbHasAlpha)) {
        if (J.S.mode == MODE_IDENTITY) {
            if (image->depth == 8 && rgb->depth == 8 && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 &&
                image->yuvRange == AVIF_RANGE_FULL) {
                fast = 1;
                J.identityCopy = 1;
            }
        } else if (J.S.mode == MODE_COEFF) {
            fast = 1;
        }
    }
    J.inLoopMul = fast ? MUL_NONE : mul;
    if (convertedByBackend)
        fast = 1; /* the backend never (un)multiplies: the integer post-pass below does, :1574-1585 */

    /* look-up tables, :575-603 */
    const size_t n = convertedByBackend ? 0 : ((size_t)1 << image->depth);
    float * lutY = (float *)malloc(n * sizeof(float));
    float * lutUV = (float *)malloc(n * sizeof(float));
    if (n && (!lutY || !lutUV)) {
        free(lutY);
        free(lutUV);
        return AVIF_RESULT_OUT_OF_MEMORY;
    }
    for (size_t cp = 0; cp < n; ++cp) {
        lutY[cp] = ((float)cp - J.S.biasY) / J.S.rangeY;
        lutUV[cp] = (J.S.mode == MODE_IDENTITY) ? lutY[cp] : (((float)cp - J.S.biasUV) / J.S.rangeUV);
    }
    J.lutY = lutY;
    J.lutUV = lutUV;

    for (uint32_t j = rect->y; !convertedByBackend && j < rect->y + rect->height; ++j)
        for (uint32_t i = rect->x; i < rect->x + rect->width; ++i)
            convertPixel(&J, i, j);
    free(lutY);
    free(lutUV);

    /* integer post-pass after the fast loops, :1574-1585 */
    if (fast && mul != MUL_NONE) {
        avifRGBImage view = *rgb;
        view.pixels = rgb->pixels + (size_t)rect->y * rgb->rowBytes + (size_t)rect->x * J.L.pixBytes;
        view.width = rect->width;
        view.height = rect->height;
        const avifResult r = alphaPassWithBackend(&view, mul == MUL_UNMULTIPLY, backend);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    if (rgb->isFloat)
        toHalfFloat(rgb, rect); /* :1587-1590 */
    return AVIF_RESULT_OK;
}

avifResult oracleImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    /* loops run over image->width x image->height (:686,:703); alpha over rgb->width x rgb->height (:1467-1468).
     * The API requires both to match (include/avif/avif.h:934-936); the oracle uses the image's. */
    const avifCropRect whole = { 0, 0, image->width, image->height };
    return yuvToRgbRect(image, rgb, &whole, NULL);
}

avifResult oracleImageYUVToRGBWithBackend(const avifImage * image, avifRGBImage * rgb, const OracleBackend * backend)
{
    const avifCropRect whole = { 0, 0, image->width, image->height };
    return yuvToRgbRect(image, rgb, &whole, backend);
}

avifResult oracleImageYUVToRGBRect(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * rect)
{
    if (rect->width > canvas->width || rect->height > canvas->height || rect->x > canvas->width - rect->width ||
        rect->y > canvas->height - rect->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    return yuvToRgbRect(canvas, rgbCanvas, rect, NULL);
}

/* ------------------------------------------------------------------------- */
/* RGB -> YUV                                     (src/reformat.c:221-571)     */

static int quantY(const YuvSpace * S, float v) /* :197-201 */
{
    return clampi((int)roundHalfUp(v * S->rangeY + S->biasY), 0, S->maxv);
}
static int quantUV(const YuvSpace * S, float v) /* :203-219 */
{
    const int q = (S->mode == MODE_IDENTITY) ? (int)roundHalfUp(v * S->rangeY + S->biasY) : (int)roundHalfUp(v * S->rangeUV + S->biasUV);
    return clampi(q, 0, S->maxv);
}
static void putSample(uint8_t * plane, size_t rowBytes, uint32_t x, uint32_t y, int chanBytes, int v)
{
    uint8_t * p = plane + (size_t)y * rowBytes + (size_t)x * chanBytes;
    if (chanBytes == 1)
        *p = (uint8_t)v;
    else
        store16(p, (unsigned)v);
}

static avifResult allocatePlanes(avifImage * image, int withAlpha) /* src/avif.c:431-490 */
{
    if (image->width == 0 || image->height == 0 || image->depth == 0 || image->depth > 16)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const size_t cs = (image->depth > 8) ? 2 : 1;
    const size_t fullRow = cs * image->width;
    if (image->yuvFormat != AVIF_PIXEL_FORMAT_NONE) {
        image->imageOwnsYUVPlanes = AVIF_TRUE;
        if (!image->yuvPlanes[0]) {
            image->yuvPlanes[0] = (uint8_t *)malloc(fullRow * image->height);
            if (!image->yuvPlanes[0])
                return AVIF_RESULT_OUT_OF_MEMORY;
            image->yuvRowBytes[0] = (uint32_t)fullRow;
        }
        if (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400) {
            const int sx = (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV444);
            const int sy = (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420);
            const size_t cw = ((size_t)image->width + sx) >> sx, ch = ((size_t)image->height + sy) >> sy;
            for (int p = 1; p <= 2; ++p) {
                if (!image->yuvPlanes[p]) {
                    image->yuvPlanes[p] = (uint8_t *)malloc(cs * cw * ch);
                    if (!image->yuvPlanes[p])
                        return AVIF_RESULT_OUT_OF_MEMORY;
                    image->yuvRowBytes[p] = (uint32_t)(cs * cw);
                }
            }
        }
    }
    if (withAlpha) {
        image->imageOwnsAlphaPlane = AVIF_TRUE;
        if (!image->alphaPlane) {
            image->alphaPlane = (uint8_t *)malloc(fullRow * image->height);
            if (!image->alphaPlane)
                return AVIF_RESULT_OUT_OF_MEMORY;
            image->alphaRowBytes = (uint32_t)fullRow;
        }
    }
    return AVIF_RESULT_OK;
}

static avifResult rgbToYuv(avifImage * image, const avifRGBImage * rgb, const OracleBackend * backend);

avifResult oracleImageRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    return rgbToYuv(image, rgb, NULL);
}
avifResult oracleImageRGBToYUVWithBackend(avifImage * image, const avifRGBImage * rgb, const OracleBackend * backend)
{
    return rgbToYuv(image, rgb, backend);
}

static avifResult rgbToYuv(avifImage * image, const avifRGBImage * rgb, const OracleBackend * backend)
{
    if (!rgb->pixels || rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return AVIF_RESULT_REFORMAT_FAILED; /* :223-225 */
    RgbLayout L;
    YuvSpace S;
    if (!makePlan(image, rgb, &L, &S))
        return AVIF_RESULT_REFORMAT_FAILED;
    if (rgb->isFloat)
        return AVIF_RESULT_NOT_IMPLEMENTED; /* :232-234 */
    const int hasAlpha = fmtHasAlpha(rgb->format) && !rgb->ignoreAlpha;
    const avifResult ar = allocatePlanes(image, hasAlpha); /* :236-240 */
    if (ar != AVIF_RESULT_OK)
        return ar;

    int mul = MUL_NONE; /* :242-249 */
    if (hasAlpha) {
        if (!rgb->alphaPremultiplied && image->alphaPremultiplied)
            mul = MUL_MULTIPLY;
        else if (rgb->alphaPremultiplied && !image->alphaPremultiplied)
            mul = MUL_UNMULTIPLY;
    }

    const uint32_t W = image->width, H = image->height;
    const int gray = fmtIsGray(rgb->format);
    if (!gray && rgb->chromaDownsampling == AVIF_CHROMA_DOWNSAMPLING_SHARP_YUV && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420)
        return AVIF_RESULT_NOT_IMPLEMENTED; /* libsharpyuv absent: src/reformat_libsharpyuv.c:77-84 via :255-263 */

    int convertedByBackend = 0; /* :264-272 */
    if (!gray && backend && backend->rgbToYuv && !rgb->avoidLibYUV && mul == MUL_NONE) {
        const avifResult br = backend->rgbToYuv(image, rgb);
        if (br == AVIF_RESULT_OK)
            convertedByBackend = 1;
        else if (br != AVIF_RESULT_NOT_IMPLEMENTED)
            return br;
    }

    if (convertedByBackend) {
        /* colour planes are done */
    } else if (!gray) {
        const float kr = S.kr, kg = S.kg, kb = S.kb;
        for (uint32_t oj = 0; oj < H; oj += 2) {
            for (uint32_t oi = 0; oi < W; oi += 2) {
                const uint32_t bw = (oi + 1 >= W) ? 1 : 2, bh = (oj + 1 >= H) ? 1 : 2;
                float blkU[2][2], blkV[2][2]; /* [bI][bJ] like :280 */
                for (uint32_t bJ = 0; bJ < bh; ++bJ) {
                    for (uint32_t bI = 0; bI < bw; ++bI) {
                        const uint32_t i = oi + bI, j = oj + bJ;
                        const uint8_t * px = rgb->pixels + (size_t)j * rgb->rowBytes + (size_t)i * L.pixBytes;
                        float c[3];
                        const int offs[3] = { L.offR, L.offG, L.offB };
                        for (int k = 0; k < 3; ++k) /* :312-323 */
                            c[k] = (float)((L.chanBytes > 1) ? load16(px + offs[k]) : px[offs[k]]) / L.maxf;
                        if (mul != MUL_NONE) { /* :325-358 */
                            const float a = (float)((L.chanBytes > 1) ? load16(px + L.offA) : px[L.offA]) / L.maxf;
                            if (a == 0) {
                                c[0] = c[1] = c[2] = 0;
                            } else if (a < 1.0f) {
                                for (int k = 0; k < 3; ++k) {
                                    if (mul == MUL_MULTIPLY) {
                                        c[k] *= a;
                                    } else {
                                        c[k] /= a;
                                        c[k] = (c[k] < 1.0f) ? c[k] : 1.0f;
                                    }
                                }
                            }
                        }
                        float y, u, v;
                        if (S.mode == MODE_IDENTITY) { /* :361-365 */
                            y = c[1], u = c[2], v = c[0];
                        } else if (S.mode == MODE_YCGCO) { /* :366-370 */
                            y = 0.5f * c[1] + 0.25f * (c[0] + c[2]);
                            u = 0.5f * c[1] - 0.25f * (c[0] + c[2]);
                            v = 0.5f * (c[0] - c[2]);
                        } else if (S.mode == MODE_YCGCO_RE || S.mode == MODE_YCGCO_RO) { /* :371-381 */
                            int q[3];
                            for (int k = 0; k < 3; ++k) {
                                float t = c[k] * L.maxf;
                                t = (t < 0.0f) ? 0.0f : ((L.maxf < t) ? L.maxf : t);
                                q[k] = (int)roundHalfUp(t);
                            }
                            const int Co = q[0] - q[2];
                            const int t = q[2] + (Co >> 1);
                            const int Cg = q[1] - t;
                            y = (float)(t + (Cg >> 1)) / S.rangeY;
                            u = (float)Cg / S.rangeUV;
                            v = (float)Co / S.rangeUV;
                        } else { /* :383-386 */
                            y = (kr * c[0]) + (kg * c[1]) + (kb * c[2]);
                            u = (c[2] - y) / (2 * (1 - kb));
                            v = (c[0] - y) / (2 * (1 - kr));
                        }
                        blkU[bI][bJ] = u;
                        blkV[bI][bJ] = v;
                        putSample(image->yuvPlanes[0], image->yuvRowBytes[0], i, j, S.chanBytes, quantY(&S, y));
                        if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444) {
                            putSample(image->yuvPlanes[1], image->yuvRowBytes[1], i, j, S.chanBytes, quantUV(&S, u));
                            putSample(image->yuvPlanes[2], image->yuvRowBytes[2], i, j, S.chanBytes, quantUV(&S, v));
                        }
                    }
                }
                if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) { /* :413-440 */
                    float su = 0.0f, sv = 0.0f;
                    for (uint32_t bJ = 0; bJ < bh; ++bJ)
                        for (uint32_t bI = 0; bI < bw; ++bI) {
                            su += blkU[bI][bJ];
                            sv += blkV[bI][bJ];
                        }
                    const float cnt = (float)(bw * bh);
                    putSample(image->yuvPlanes[1], image->yuvRowBytes[1], oi >> 1, oj >> 1, S.chanBytes, quantUV(&S, su / cnt));
                    putSample(image->yuvPlanes[2], image->yuvRowBytes[2], oi >> 1, oj >> 1, S.chanBytes, quantUV(&S, sv / cnt));
                } else if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV422) { /* :441-467 */
                    for (uint32_t bJ = 0; bJ < bh; ++bJ) {
                        float su = 0.0f, sv = 0.0f;
                        for (uint32_t bI = 0; bI < bw; ++bI) {
                            su += blkU[bI][bJ];
                            sv += blkV[bI][bJ];
                        }
                        const float cnt = (float)bw;
                        putSample(image->yuvPlanes[1], image->yuvRowBytes[1], oi >> 1, oj + bJ, S.chanBytes, quantUV(&S, su / cnt));
                        putSample(image->yuvPlanes[2], image->yuvRowBytes[2], oi >> 1, oj + bJ, S.chanBytes, quantUV(&S, sv / cnt));
                    }
                }
            }
        }
    } else { /* gray source, :471-543 */
        for (uint32_t j = 0; j < H; ++j) {
            for (uint32_t i = 0; i < W; ++i) {
                const uint8_t * px = rgb->pixels + (size_t)j * rgb->rowBytes + (size_t)i * L.pixBytes;
                float g = (float)((L.chanBytes > 1) ? load16(px + L.offGray) : px[L.offGray]) / L.maxf;
                if (mul != MUL_NONE) {
                    const float a = (float)((L.chanBytes > 1) ? load16(px + L.offA) : px[L.offA]) / L.maxf;
                    if (a == 0) {
                        g = 0;
                    } else if (a < 1.0f) {
                        if (mul == MUL_MULTIPLY) {
                            g *= a;
                        } else {
                            g /= a;
                            g = (g < 1.0f) ? g : 1.0f;
                        }
                    }
                }
                putSample(image->yuvPlanes[0], image->yuvRowBytes[0], i, j, S.chanBytes, quantY(&S, g));
            }
        }
        /* chroma planes set to half over their full rowBytes extent, :520-542 */
        const uint32_t shiftedH = (uint32_t)(((uint64_t)H + S.shiftY) >> S.shiftY);
        const int half = 1 << (image->depth - 1);
        for (int p = 1; p <= 2; ++p) {
            if (!image->yuvPlanes[p])
                continue;
            const size_t bytes = (size_t)shiftedH * image->yuvRowBytes[p];
            if (S.chanBytes > 1) {
                for (size_t k = 0; k < bytes / 2; ++k)
                    store16(image->yuvPlanes[p] + 2 * k, (unsigned)half);
            } else {
                memset(image->yuvPlanes[p], half, bytes);
            }
        }
    }

    if (image->alphaPlane && image->alphaRowBytes) { /* :545-569 */
        AlphaJob A;
        memset(&A, 0, sizeof(A));
        A.width = W;
        A.height = H;
        A.dstDepth = (int)image->depth;
        A.dstPixBytes = S.chanBytes;
        A.dstRowBytes = image->alphaRowBytes;
        A.dst = image->alphaPlane;
        if (fmtHasAlpha(rgb->format) && !rgb->ignoreAlpha) {
            A.srcDepth = (int)rgb->depth;
            A.srcPixBytes = L.pixBytes;
            A.srcRowBytes = rgb->rowBytes;
            A.src = rgb->pixels + L.offA;
        }
        runAlphaJob(&A);
    }
    return AVIF_RESULT_OK;
}

/* ------------------------------------------------------------------------- */
/* grid images: tile -> canvas, alpha range, conversion  (src/read.c:1823-1877, :6724-6764, :6818-6828)   */

static void copyPlaneRect(uint8_t * dst, size_t dstRowBytes, const uint8_t * src, size_t srcRowBytes, size_t bytesPerRow, uint32_t rows)
{
    for (uint32_t j = 0; j < rows; ++j) /* avifImageCopySamples, src/avif.c:187-225 */
        memcpy(dst + (size_t)j * dstRowBytes, src + (size_t)j * srcRowBytes, bytesPerRow);
}

avifResult oracleGridYUVToRGB(const oracleGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                              avifBool alphaIsLimitedRange, avifRGBImage * rgb, int libyuvBuild)
{
    const avifImage * first = colorTiles[0];
    const uint32_t tw = first->width, th = first->height;
    const size_t bps = (first->depth > 8) ? 2 : 1;
    const int sx = (first->yuvFormat == AVIF_PIXEL_FORMAT_YUV444) ? 0 : 1;
    const int sy = (first->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    avifImage canvas;
    memcpy(&canvas, first, sizeof(canvas));
    canvas.width = grid->outputWidth, canvas.height = grid->outputHeight;
    canvas.yuvPlanes[0] = canvas.yuvPlanes[1] = canvas.yuvPlanes[2] = canvas.alphaPlane = NULL;
    canvas.imageOwnsYUVPlanes = canvas.imageOwnsAlphaPlane = AVIF_FALSE;
    avifResult r = allocatePlanes(&canvas, alphaTiles != NULL);
    if (r != AVIF_RESULT_OK)
        return r;
    for (uint32_t t = 0; t < grid->rows * grid->columns; ++t) {
        const uint32_t X0 = (t % grid->columns) * tw, Y0 = (t / grid->columns) * th;
        const uint32_t w = (X0 + tw > grid->outputWidth) ? grid->outputWidth - X0 : tw; /* :1863-1868 */
        const uint32_t h = (Y0 + th > grid->outputHeight) ? grid->outputHeight - Y0 : th;
        const avifImage * tile = colorTiles[t];
        for (int p = 0; p < 3; ++p) {
            if (!canvas.yuvPlanes[p] || !tile->yuvPlanes[p])
                continue;
            const int px = p ? sx : 0, py = p ? sy : 0;
            copyPlaneRect(canvas.yuvPlanes[p] + (size_t)(Y0 >> py) * canvas.yuvRowBytes[p] + (size_t)(X0 >> px) * bps, canvas.yuvRowBytes[p],
                          tile->yuvPlanes[p], tile->yuvRowBytes[p], (((size_t)w + px) >> px) * bps, (h + py) >> py);
        }
        if (alphaTiles) {
            const avifImage * atile = alphaTiles[t];
            for (uint32_t j = 0; j < h; ++j) {
                const uint8_t * src = atile->alphaPlane + (size_t)j * atile->alphaRowBytes;
                uint8_t * dst = canvas.alphaPlane + (size_t)(Y0 + j) * canvas.alphaRowBytes + (size_t)X0 * bps;
                for (uint32_t i = 0; i < w; ++i) {
                    int a = (bps == 2) ? (int)load16(src + 2 * (size_t)i) : src[i];
                    if (alphaIsLimitedRange)
                        a = oracleLimitedToFullY(first->depth, a); /* :6748,6758 */
                    if (bps == 2)
                        store16(dst + 2 * (size_t)i, (unsigned)a);
                    else
                        dst[i] = (uint8_t)a;
                }
            }
        }
    }
    r = libyuvBuild ? oracleLibyuvImageYUVToRGB(&canvas, rgb) : oracleImageYUVToRGB(&canvas, rgb);
    for (int p = 0; p < 3; ++p)
        free(canvas.yuvPlanes[p]);
    free(canvas.alphaPlane);
    return r;
}

/* ------------------------------------------------------------------------- */
/* Sample Transform expressions                       (src/sampletransform.c)                                        */

static int32_t sat32(int64_t v) /* avifSampleTransformClamp32b, :194-197 */
{
    return v <= INT32_MIN ? INT32_MIN : (v >= INT32_MAX ? INT32_MAX : (int32_t)v);
}
static int32_t satoUnary(int32_t a, int op) /* :199-224 */
{
    if (op == AVIF_SAMPLE_TRANSFORM_NEGATION)
        return sat32(-(int64_t)a);
    if (op == AVIF_SAMPLE_TRANSFORM_ABSOLUTE)
        return a >= 0 ? a : sat32(-(int64_t)a);
    if (op == AVIF_SAMPLE_TRANSFORM_NOT)
        return ~a;
    int32_t log2v = 0; /* BSR */
    if (a <= 0)
        return 0;
    for (a >>= 1; a != 0; a >>= 1)
        ++log2v;
    return log2v;
}
static int32_t satoBinary(int32_t l, int32_t r, int op) /* :226-277 */
{
    switch (op) {
        case AVIF_SAMPLE_TRANSFORM_SUM:
            return sat32((int64_t)l + r);
        case AVIF_SAMPLE_TRANSFORM_DIFFERENCE:
            return sat32((int64_t)l - r);
        case AVIF_SAMPLE_TRANSFORM_PRODUCT:
            return sat32((int64_t)l * r);
        case AVIF_SAMPLE_TRANSFORM_QUOTIENT:
            return r == 0 ? l : sat32((int64_t)l / r);
        case AVIF_SAMPLE_TRANSFORM_AND:
            return l & r;
        case AVIF_SAMPLE_TRANSFORM_OR:
            return l | r;
        case AVIF_SAMPLE_TRANSFORM_XOR:
            return l ^ r;
        case AVIF_SAMPLE_TRANSFORM_POW: {
            if (l == 0 || l == 1)
                return l;
            if (l == -1)
                return (r % 2 == 0) ? 1 : -1;
            if (r == 0)
                return 1;
            if (r == 1)
                return l;
            if (r < 0)
                return 0;
            int64_t acc = l;
            for (int32_t i = 1; i < r; ++i) {
                acc *= l;
                if (acc < INT32_MIN || acc > INT32_MAX)
                    return (l > 0 || r % 2 == 0) ? INT32_MAX : INT32_MIN;
            }
            return (int32_t)acc;
        }
        case AVIF_SAMPLE_TRANSFORM_MIN:
            return l <= r ? l : r;
        default:
            return l <= r ? r : l;
    }
}

static void planeGeometry(const avifImage * im, int c, uint32_t * w, uint32_t * h) /* avifImagePlaneWidth / Height, src/avif.c:351-400 */
{
    const int sx = (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 0 : 1;
    const int sy = (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    const int present = (c < 3) ? (im->yuvPlanes[c] != NULL && !((c == 1 || c == 2) && im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)) : (im->alphaPlane != NULL);
    *w = *h = 0;
    if (!present)
        return;
    *w = (c == 1 || c == 2) ? (im->width + sx) >> sx : im->width;
    *h = (c == 1 || c == 2) ? (im->height + sy) >> sy : im->height;
}

avifResult oracleImageApplyOperations(avifImage * dstImage, avifSampleTransformBitDepth bitDepth, uint32_t numTokens,
                                      const avifSampleTransformToken * tokens, uint8_t numInputImageItems, const avifImage * const * inputImageItems,
                                      avifPlanesFlags planes)
{
    /* avifSampleTransformExpressionIsValid, :13-40 */
    uint32_t depth = 0;
    for (uint32_t t = 0; t < numTokens; ++t) {
        const int type = (int)tokens[t].type;
        if (type >= AVIF_SAMPLE_TRANSFORM_RESERVED)
            return AVIF_RESULT_INTERNAL_ERROR;
        if (type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX && (tokens[t].inputImageItemIndex == 0 || tokens[t].inputImageItemIndex > numInputImageItems))
            return AVIF_RESULT_INTERNAL_ERROR;
        if (type < AVIF_SAMPLE_TRANSFORM_FIRST_UNARY_OPERATOR) {
            ++depth;
        } else if (type < AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR) {
            if (depth < 1)
                return AVIF_RESULT_INTERNAL_ERROR;
        } else {
            if (depth < 2)
                return AVIF_RESULT_INTERNAL_ERROR;
            --depth;
        }
    }
    if (depth != 1)
        return AVIF_RESULT_INTERNAL_ERROR;
    const int skipColor = !(planes & AVIF_PLANES_YUV), skipAlpha = !(planes & AVIF_PLANES_A);
    for (int c = 0; c < 4; ++c) { /* :371-384 */
        if ((skipColor && c < 3) || (skipAlpha && c == 3))
            continue;
        uint32_t w, h;
        planeGeometry(dstImage, c, &w, &h);
        for (uint32_t i = 0; i < numInputImageItems; ++i) {
            uint32_t wi, hi;
            planeGeometry(inputImageItems[i], c, &wi, &hi);
            if (wi != w || hi != h)
                return AVIF_RESULT_BMFF_PARSE_FAILED;
        }
    }
    if (bitDepth != AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_32)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    int32_t * stack = (int32_t *)malloc((numTokens / 2 + 1) * sizeof(int32_t));
    if (!stack)
        return AVIF_RESULT_OUT_OF_MEMORY;
    const int32_t maxValue = (1 << dstImage->depth) - 1;
    for (int c = 0; c < 4; ++c) { /* :296-350 */
        if ((skipColor && c < 3) || (skipAlpha && c == 3))
            continue;
        uint32_t w, h;
        planeGeometry(dstImage, c, &w, &h);
        for (uint32_t y = 0; y < h; ++y) {
            for (uint32_t x = 0; x < w; ++x) {
                uint32_t n = 0;
                for (uint32_t t = 0; t < numTokens; ++t) {
                    const int type = (int)tokens[t].type;
                    if (type == AVIF_SAMPLE_TRANSFORM_CONSTANT) {
                        stack[n++] = tokens[t].constant;
                    } else if (type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX) {
                        const avifImage * im = inputImageItems[tokens[t].inputImageItemIndex - 1];
                        const uint8_t * row = ((c < 3) ? im->yuvPlanes[c] : im->alphaPlane) + (size_t)((c < 3) ? im->yuvRowBytes[c] : im->alphaRowBytes) * y;
                        stack[n++] = (im->depth > 8) ? (int32_t)load16(row + 2 * (size_t)x) : row[x];
                    } else if (type < AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR) {
                        stack[n - 1] = satoUnary(stack[n - 1], type);
                    } else {
                        stack[n - 2] = satoBinary(stack[n - 2], stack[n - 1], type);
                        --n;
                    }
                }
                const int32_t v = clampi(stack[0], 0, maxValue);
                uint8_t * row = ((c < 3) ? dstImage->yuvPlanes[c] : dstImage->alphaPlane) + (size_t)((c < 3) ? dstImage->yuvRowBytes[c] : dstImage->alphaRowBytes) * y;
                if (dstImage->depth > 8)
                    store16(row + 2 * (size_t)x, (unsigned)v);
                else
                    row[x] = (uint8_t)v;
            }
        }
    }
    free(stack);
    return AVIF_RESULT_OK;
}

/* ------------------------------------------------------------------------- */
/* crop / rotate / mirror of an RGB image            (apps/shared/avifutil.c:667-825)                              */

static uint32_t rgbPixelSize(const avifRGBImage * rgb) /* avifRGBImagePixelSize, src/avif.c:692-698 */
{
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return 2;
    return (uint32_t)fmtChannels(rgb->format) * ((rgb->depth > 8) ? 2 : 1);
}

avifResult oracleRGBImageTransform(avifRGBImage * dst, const avifRGBImage * src, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                   avifBool mirror, uint8_t axis)
{
    const uint32_t px = rgbPixelSize(src);
    /* the clean-aperture view, :667-682 */
    const uint8_t * base = src->pixels;
    uint32_t w = src->width, h = src->height;
    if (crop) {
        base += (size_t)crop->y * src->rowBytes + (size_t)crop->x * px;
        w = crop->width, h = crop->height;
    }
    /* rotation into a new image, :687-743 (angle 0 or an absent box leave the view as it is, :805) */
    const uint8_t a = rotate ? angle : 0;
    if (a > 3)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t nw = (a == 0 || a == 2) ? w : h, nh = (a == 0 || a == 2) ? h : w;
    const size_t tmpBytes = (size_t)nw * nh * px;
    uint8_t * tmp = (uint8_t *)malloc(tmpBytes > 0 ? tmpBytes : 1);
    if (!tmp)
        return AVIF_RESULT_OUT_OF_MEMORY;
    const size_t tmpRow = (size_t)nw * px;
    for (uint32_t j = 0; j < h; ++j) {
        for (uint32_t i = 0; i < w; ++i) {
            const uint8_t * s = base + (size_t)j * src->rowBytes + (size_t)i * px;
            uint32_t x, y;
            switch (a) {
                case 1: x = j, y = w - 1 - i; break;          /* 90 degrees anti-clockwise, :711-718 */
                case 2: x = w - 1 - i, y = h - 1 - j; break;  /* 180 degrees, :721-729 */
                case 3: x = h - 1 - j, y = i; break;          /* 90 degrees clockwise, :732-739 */
                default: x = i, y = j; break;
            }
            memcpy(tmp + (size_t)y * tmpRow + (size_t)x * px, s, px);
        }
    }
    /* mirror in place, :745-785 */
    if (mirror) {
        if (axis == 0) {
            for (uint32_t y = 0; y < nh / 2; ++y)
                for (size_t k = 0; k < tmpRow; ++k) {
                    const uint8_t t = tmp[(size_t)y * tmpRow + k];
                    tmp[(size_t)y * tmpRow + k] = tmp[(size_t)(nh - 1 - y) * tmpRow + k];
                    tmp[(size_t)(nh - 1 - y) * tmpRow + k] = t;
                }
        } else if (axis == 1) {
            for (uint32_t y = 0; y < nh; ++y)
                for (uint32_t x = 0; x < nw / 2; ++x)
                    for (uint32_t k = 0; k < px; ++k) {
                        uint8_t * p1 = tmp + (size_t)y * tmpRow + (size_t)x * px + k;
                        uint8_t * p2 = tmp + (size_t)y * tmpRow + (size_t)(nw - 1 - x) * px + k;
                        const uint8_t t = *p1;
                        *p1 = *p2;
                        *p2 = t;
                    }
        } else {
            free(tmp);
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
    }
    if (dst->width != nw || dst->height != nh) {
        free(tmp);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    for (uint32_t y = 0; y < nh; ++y)
        memcpy(dst->pixels + (size_t)y * dst->rowBytes, tmp + (size_t)y * tmpRow, tmpRow);
    free(tmp);
    return AVIF_RESULT_OK;
}

/* ------------------------------------------------------------------------- */
/* limited <-> full integer helpers              (src/reformat.c:1750-1840)    */

static int limitedToFull(int v, int lo, int hi, int full)
{
    v = (((v - lo) * full) + ((hi - lo) / 2)) / (hi - lo);
    return clampi(v, 0, full);
}
static int fullToLimited(int v, int lo, int hi, int full)
{
    v = (((v * (hi - lo)) + (full / 2)) / full) + lo;
    return clampi(v, lo, hi);
}
static int depthIndex(uint32_t depth)
{
    return depth == 8 ? 0 : depth == 10 ? 1 : depth == 12 ? 2 : -1;
}
static const int kLo[3] = { 16, 64, 256 };
static const int kHiY[3] = { 235, 940, 3760 };
static const int kHiUV[3] = { 240, 960, 3840 };
static const int kFull[3] = { 255, 1023, 4095 };

int oracleLimitedToFullY(uint32_t depth, int v)
{
    const int k = depthIndex(depth);
    return k < 0 ? v : limitedToFull(v, kLo[k], kHiY[k], kFull[k]);
}
int oracleLimitedToFullUV(uint32_t depth, int v)
{
    const int k = depthIndex(depth);
    return k < 0 ? v : limitedToFull(v, kLo[k], kHiUV[k], kFull[k]);
}
int oracleFullToLimitedY(uint32_t depth, int v)
{
    const int k = depthIndex(depth);
    return k < 0 ? v : fullToLimited(v, kLo[k], kHiY[k], kFull[k]);
}
int oracleFullToLimitedUV(uint32_t depth, int v)
{
    const int k = depthIndex(depth);
    return k < 0 ? v : fullToLimited(v, kLo[k], kHiUV[k], kFull[k]);
}
