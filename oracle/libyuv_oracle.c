/*
 * libyuv_oracle.c -- TEST INFRASTRUCTURE (the checker), NOT PRODUCT CODE.
 *
 * The "integer path" of the reference: what a libavif built WITH libyuv computes.  libavif hands part of its
 * reformat work to libyuv through four hooks (/root/reference/include/avif/internal.h:346-378, implemented in
 * src/reformat_libyuv.c); this file restates
 *   (1) libavif's dispatch: which (format, depth, range, matrix, RGB layout, upsampling) combinations go to which
 *       libyuv function and what happens around the call (src/reformat_libyuv.c:270-381, :544-1108, :1112-1161),
 *       from the reference source, and
 *   (2) the fixed-point arithmetic of those libyuv functions.  libyuv is a THIRD-PARTY dependency whose source is
 *       not under /root/reference (chromium.googlesource.com/libyuv/libyuv, pinned 5d03bf9 = LIBYUV_VERSION 1949,
 *       cmake/Modules/LocalLibyuv.cmake:4); its arithmetic is restated from the closed forms of SURVEY.md
 *       Appendix D, which were recovered black-box from the only libyuv-enabled libavif binary available offline
 *       (Pillow's bundled libavif 1.4.1 + libyuv 1922).
 *
 * Parity status: PINNED against that binary (tests/test_libyuv_oracle.py: the public entry points of the Pillow
 * .so, default avoidLibYUV=0, over the configuration sweep, byte-identical) and by golden fixtures generated from it
 * (tests/golden/yuvlib_*.npz, tests/tools/make_golden_libyuv.py).  Version skew 1922 -> 1949 cannot be
 * quantified offline; the libavif-side dispatch restated here is the reference's (1.4.2-devel).
 *
 * Domain: samples are expected inside their nominal bit depth (libyuv reads 10/12-bit samples without clamping).
 */
#include "oracle_backend.h"
#include "reformat_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* helpers                                                                   */

static int clamp255(int v)
{
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
static int minInt(int a, int b)
{
    return a < b ? a : b;
}
static unsigned rd16(const uint8_t * p)
{
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
static int rgbHasAlpha(avifRGBFormat f)
{
    return f == AVIF_RGB_FORMAT_RGBA || f == AVIF_RGB_FORMAT_ARGB || f == AVIF_RGB_FORMAT_BGRA || f == AVIF_RGB_FORMAT_ABGR ||
           f == AVIF_RGB_FORMAT_GRAYA || f == AVIF_RGB_FORMAT_AGRAY;
}

/* ------------------------------------------------------------------------- */
/* YuvConstants as seen through libavif (Appendix D.1)                        */

typedef struct FxMatrix
{
    int yg, yb, ub, ug, vg, vr;
} FxMatrix;

static const FxMatrix kI601 = { 18997, -1160, 128, 25, 52, 102 };  /* kYuvI601Constants  */
static const FxMatrix kJPEG = { 16320, 32, 113, 22, 46, 90 };      /* kYuvJPEGConstants  */
static const FxMatrix kH709 = { 18997, -1160, 128, 14, 34, 115 };  /* kYuvH709Constants  */
static const FxMatrix kF709 = { 16320, 32, 119, 12, 30, 101 };     /* kYuvF709Constants  */
static const FxMatrix k2020 = { 19003, -1160, 128, 12, 42, 107 };  /* kYuv2020Constants  */
static const FxMatrix kV2020 = { 16320, 32, 120, 11, 37, 94 };     /* kYuvV2020Constants */

/* src/reformat_libyuv.c:775-904 (getLibYUVConstants) */
static const FxMatrix * selectMatrix(const avifImage * image)
{
    unsigned mc = image->matrixCoefficients;
    if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400 && mc == AVIF_MATRIX_COEFFICIENTS_IDENTITY)
        mc = AVIF_MATRIX_COEFFICIENTS_BT601; /* :777-781 */
    const int full = (image->yuvRange == AVIF_RANGE_FULL);
    int family = 0; /* 1 = 709, 2 = 601, 3 = 2020 */
    switch (mc) {
        case AVIF_MATRIX_COEFFICIENTS_BT709:
            family = 1;
            break;
        case AVIF_MATRIX_COEFFICIENTS_BT470BG:
        case AVIF_MATRIX_COEFFICIENTS_BT601:
        case AVIF_MATRIX_COEFFICIENTS_UNSPECIFIED:
            family = 2;
            break;
        case AVIF_MATRIX_COEFFICIENTS_BT2020_NCL:
            family = 3;
            break;
        case AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL:
            switch (image->colorPrimaries) {
                case 1: /* BT709 */
                case 2: /* UNSPECIFIED */
                    family = 1;
                    break;
                case 5: /* BT470BG */
                case 6: /* BT601 */
                    family = 2;
                    break;
                case 9: /* BT2020 */
                    family = 3;
                    break;
                default:
                    break;
            }
            break;
        default:
            break;
    }
    switch (family) {
        case 1:
            return full ? &kF709 : &kH709;
        case 2:
            return full ? &kJPEG : &kI601;
        case 3:
            return full ? &kV2020 : &k2020;
        default:
            return NULL;
    }
}

/* ------------------------------------------------------------------------- */
/* which libyuv entry libavif picks (src/reformat_libyuv.c:544-773)            */

typedef struct Route
{
    int mono;        /* I400ToARGBMatrix */
    int withFilter;  /* a *MatrixFilter entry: upsampling follows rgb->chromaUpsampling; otherwise nearest */
    int withAlpha;   /* an *Alpha* entry: the A channel is taken from the alpha plane */
    int nativeDepth; /* 8 (inputs deeper than 8 bits are downshifted first, :906-930), 10 (I010/I210/I410), 12 (I012) */
} Route;

static int fmtIsRGBAorBGRA(int f)
{
    return f == AVIF_RGB_FORMAT_RGBA || f == AVIF_RGB_FORMAT_BGRA;
}
static int fmtIsRGBorBGR(int f)
{
    return f == AVIF_RGB_FORMAT_RGB || f == AVIF_RGB_FORMAT_BGR;
}
static int fmtIsARGBorABGR(int f)
{
    return f == AVIF_RGB_FORMAT_ARGB || f == AVIF_RGB_FORMAT_ABGR;
}
static int sub2(int yf) /* 4:2:2 or 4:2:0 */
{
    return yf == AVIF_PIXEL_FORMAT_YUV422 || yf == AVIF_PIXEL_FORMAT_YUV420;
}
static int colour(int yf) /* 4:4:4, 4:2:2 or 4:2:0 */
{
    return yf == AVIF_PIXEL_FORMAT_YUV444 || sub2(yf);
}
/* the lookup tables of :551-712 as predicates */
static int has8Filter(int f, int yf)
{
    return (fmtIsRGBAorBGRA(f) || fmtIsRGBorBGR(f)) && sub2(yf);
}
static int has8FilterAlpha(int f, int yf)
{
    return fmtIsRGBAorBGRA(f) && sub2(yf);
}
static int has8Matrix(int f, int yf)
{
    if (fmtIsRGBorBGR(f))
        return yf == AVIF_PIXEL_FORMAT_YUV444 || yf == AVIF_PIXEL_FORMAT_YUV420;
    if (fmtIsRGBAorBGRA(f))
        return colour(yf);
    if (fmtIsARGBorABGR(f) || f == AVIF_RGB_FORMAT_RGB_565)
        return sub2(yf);
    return 0;
}
static int has8MatrixAlpha(int f, int yf)
{
    return fmtIsRGBAorBGRA(f) && colour(yf);
}
static int has10Filter(int f, int yf)
{
    return fmtIsRGBAorBGRA(f) && sub2(yf);
}
static int has10Matrix(int f, int yf)
{
    return fmtIsRGBAorBGRA(f) && colour(yf);
}
static int has12Matrix(int f, int yf)
{
    return fmtIsRGBAorBGRA(f) && yf == AVIF_PIXEL_FORMAT_YUV420;
}
static int nearestAllowed(int up) /* :536-539 */
{
    return up != AVIF_CHROMA_UPSAMPLING_BILINEAR && up != AVIF_CHROMA_UPSAMPLING_BEST_QUALITY;
}

static int selectRoute(int yf, int depth, const avifRGBImage * rgb, int alphaPreferred, Route * r)
{
    const int f = (int)rgb->format;
    memset(r, 0, sizeof(*r));
    r->nativeDepth = 8;
    if (depth > 8) { /* :716-745 */
        const int ten = (depth == 10);
        if (yf != AVIF_PIXEL_FORMAT_YUV444) {
            if (ten && has10Filter(f, yf)) { /* the alpha twin exists wherever the plain one does */
                r->withFilter = 1, r->withAlpha = alphaPreferred, r->nativeDepth = 10;
                return 1;
            }
        }
        if (yf == AVIF_PIXEL_FORMAT_YUV444 || nearestAllowed((int)rgb->chromaUpsampling)) {
            if (ten && has10Matrix(f, yf)) {
                r->withAlpha = alphaPreferred, r->nativeDepth = 10;
                return 1;
            }
            if (!ten && has12Matrix(f, yf)) { /* no alpha twin at 12 bits, :701-711 */
                r->nativeDepth = 12;
                return 1;
            }
        }
    }
    if (yf == AVIF_PIXEL_FORMAT_YUV400) { /* :746-749 */
        r->mono = 1;
        return fmtIsRGBAorBGRA(f);
    }
    if (yf != AVIF_PIXEL_FORMAT_YUV444) { /* :750-764 */
        if (alphaPreferred && has8FilterAlpha(f, yf)) {
            r->withFilter = 1, r->withAlpha = 1;
            return 1;
        }
        if (has8Filter(f, yf)) {
            r->withFilter = 1;
            return 1;
        }
        if (!nearestAllowed((int)rgb->chromaUpsampling))
            return 0;
    }
    if (alphaPreferred && has8MatrixAlpha(f, yf)) { /* :765-772 */
        r->withAlpha = 1;
        return 1;
    }
    return has8Matrix(f, yf);
}

/* ------------------------------------------------------------------------- */
/* YUV -> RGB (Appendix D.1-D.3)                                              */

typedef struct PlaneReader
{
    const uint8_t * base;
    size_t rowBytes;
    int wide;  /* 16-bit container */
    int shift; /* Convert16To8Plane: sample >> shift, saturated to 255 (:906-930) */
} PlaneReader;

static int sampleAt(const PlaneReader * p, uint32_t x, uint32_t y)
{
    const uint8_t * s = p->base + (size_t)y * p->rowBytes;
    if (!p->wide)
        return s[x];
    const int v = (int)rd16(s + 2 * (size_t)x);
    return p->shift ? minInt(v >> p->shift, 255) : v;
}

/* One luma row's worth of chroma for plane `p`, width w, at luma row j (D.2). */
static void chromaRow(const PlaneReader * p, int yf, int bilinear, uint32_t w, uint32_t h, uint32_t j, int * d)
{
    if (yf == AVIF_PIXEL_FORMAT_YUV444) {
        for (uint32_t i = 0; i < w; ++i)
            d[i] = sampleAt(p, i, j);
        return;
    }
    const uint32_t cj = (yf == AVIF_PIXEL_FORMAT_YUV420) ? (j >> 1) : j;
    if (!bilinear) { /* kFilterNone and the plain *Matrix functions */
        for (uint32_t i = 0; i < w; ++i)
            d[i] = sampleAt(p, i >> 1, cj);
        return;
    }
    const uint32_t pairs = ((w - 1) & ~1u) / 2;
    const uint32_t last = (w - 1) >> 1;
    const int oneRow = (yf == AVIF_PIXEL_FORMAT_YUV422) || j == 0 || (j == h - 1 && !(h & 1));
    if (oneRow) { /* ScaleRowUp2_Linear */
        d[0] = sampleAt(p, 0, cj);
        for (uint32_t k = 0; k < pairs; ++k) {
            const int a = sampleAt(p, k, cj), b = sampleAt(p, k + 1, cj);
            d[2 * k + 1] = (3 * a + b + 2) >> 2;
            d[2 * k + 2] = (a + 3 * b + 2) >> 2;
        }
        d[w - 1] = sampleAt(p, last, cj);
        return;
    }
    /* Scale2RowUp_Bilinear: near row cj, far row towards the side luma row j leans to */
    const uint32_t fj = (j & 1) ? cj + 1 : cj - 1;
    d[0] = (3 * sampleAt(p, 0, cj) + sampleAt(p, 0, fj) + 2) >> 2;
    for (uint32_t k = 0; k < pairs; ++k) {
        const int a0 = sampleAt(p, k, cj), a1 = sampleAt(p, k + 1, cj);
        const int b0 = sampleAt(p, k, fj), b1 = sampleAt(p, k + 1, fj);
        d[2 * k + 1] = (9 * a0 + 3 * a1 + 3 * b0 + b1 + 8) >> 4;
        d[2 * k + 2] = (3 * a0 + 9 * a1 + b0 + 3 * b1 + 8) >> 4;
    }
    d[w - 1] = (3 * sampleAt(p, last, cj) + sampleAt(p, last, fj) + 2) >> 2;
}

/* avifImageYUVToRGBLibYUV, src/reformat_libyuv.c:932-1108 */
avifResult oracleLibyuvHookYUVToRGB(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, avifBool * alphaReformatted)
{
    *alphaReformatted = AVIF_FALSE;
    if (image->width > INT_MAX || image->height > INT_MAX || image->yuvRowBytes[0] > INT_MAX || rgb->rowBytes > INT_MAX)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (rgb->depth != 8 || (image->depth != 8 && image->depth != 10 && image->depth != 12))
        return AVIF_RESULT_NOT_IMPLEMENTED; /* :939-941 */
    const FxMatrix * M = selectMatrix(image);
    if (!M)
        return AVIF_RESULT_NOT_IMPLEMENTED; /* :946-949 */
    const int hasAlphaPlane = image->alphaPlane && image->alphaRowBytes;
    const int alphaPreferred = reformatAlpha && hasAlphaPlane;
    Route R;
    if (!selectRoute((int)image->yuvFormat, (int)image->depth, rgb, alphaPreferred, &R))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (!hasAlphaPlane || R.withAlpha)
        *alphaReformatted = AVIF_TRUE; /* :956-959 and the *Alpha* branches */

    const uint32_t W = image->width, H = image->height;
    const int depth = (int)image->depth;
    const int wide = depth > 8;
    const int shift = (wide && R.nativeDepth == 8) ? depth - 8 : 0;
    const int yf = (int)image->yuvFormat;
    const int bilinear = R.withFilter && !(rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_FASTEST ||
                                           rgb->chromaUpsampling == AVIF_CHROMA_UPSAMPLING_NEAREST); /* :965-968 */
    const PlaneReader PY = { image->yuvPlanes[0], image->yuvRowBytes[0], wide, shift };
    const PlaneReader PU = { image->yuvPlanes[1], image->yuvRowBytes[1], wide, shift };
    const PlaneReader PV = { image->yuvPlanes[2], image->yuvRowBytes[2], wide, shift };
    const PlaneReader PA = { image->alphaPlane, image->alphaRowBytes, wide, shift };

    int * u = (int *)malloc(sizeof(int) * 2 * (size_t)W);
    if (!u)
        return AVIF_RESULT_REFORMAT_FAILED;
    int * v = u + W;
    const int f = (int)rgb->format;
    for (uint32_t j = 0; j < H; ++j) {
        if (!R.mono) {
            chromaRow(&PU, yf, bilinear, W, H, j, u);
            chromaRow(&PV, yf, bilinear, W, H, j, v);
        }
        uint8_t * dst = rgb->pixels + (size_t)j * rgb->rowBytes;
        for (uint32_t i = 0; i < W; ++i) {
            const int y = sampleAt(&PY, i, j);
            uint32_t y32;
            int u8 = 128, v8 = 128;
            if (R.nativeDepth == 10) { /* D.3 */
                y32 = (uint32_t)((y << 6) | (y >> 4));
                if (!R.mono)
                    u8 = clamp255(u[i] >> 2), v8 = clamp255(v[i] >> 2);
            } else if (R.nativeDepth == 12) {
                y32 = (uint32_t)((y << 4) | (y >> 8));
                if (!R.mono)
                    u8 = clamp255(u[i] >> 4), v8 = clamp255(v[i] >> 4);
            } else {
                y32 = (uint32_t)y * 0x0101u;
                if (!R.mono)
                    u8 = u[i], v8 = v[i];
            }
            const int y1 = (int)((y32 * (uint32_t)M->yg) >> 16) + M->yb; /* D.1 */
            const int b = clamp255((y1 + M->ub * (u8 - 128)) >> 6);
            const int g = clamp255((y1 - (M->ug * (u8 - 128) + M->vg * (v8 - 128))) >> 6);
            const int r = clamp255((y1 + M->vr * (v8 - 128)) >> 6);
            int a = 255;
            if (R.withAlpha) {
                a = sampleAt(&PA, i, j);
                if (R.nativeDepth == 10)
                    a = clamp255(a >> 2);
            }
            switch (f) {
                case AVIF_RGB_FORMAT_RGB:
                    dst[3 * i + 0] = (uint8_t)r, dst[3 * i + 1] = (uint8_t)g, dst[3 * i + 2] = (uint8_t)b;
                    break;
                case AVIF_RGB_FORMAT_BGR:
                    dst[3 * i + 0] = (uint8_t)b, dst[3 * i + 1] = (uint8_t)g, dst[3 * i + 2] = (uint8_t)r;
                    break;
                case AVIF_RGB_FORMAT_RGBA:
                    dst[4 * i + 0] = (uint8_t)r, dst[4 * i + 1] = (uint8_t)g, dst[4 * i + 2] = (uint8_t)b, dst[4 * i + 3] = (uint8_t)a;
                    break;
                case AVIF_RGB_FORMAT_BGRA:
                    dst[4 * i + 0] = (uint8_t)b, dst[4 * i + 1] = (uint8_t)g, dst[4 * i + 2] = (uint8_t)r, dst[4 * i + 3] = (uint8_t)a;
                    break;
                case AVIF_RGB_FORMAT_ARGB:
                    dst[4 * i + 0] = (uint8_t)a, dst[4 * i + 1] = (uint8_t)r, dst[4 * i + 2] = (uint8_t)g, dst[4 * i + 3] = (uint8_t)b;
                    break;
                case AVIF_RGB_FORMAT_ABGR:
                    dst[4 * i + 0] = (uint8_t)a, dst[4 * i + 1] = (uint8_t)b, dst[4 * i + 2] = (uint8_t)g, dst[4 * i + 3] = (uint8_t)r;
                    break;
                default: { /* AVIF_RGB_FORMAT_RGB_565 */
                    const uint16_t px = (uint16_t)((b >> 3) | ((g >> 2) << 5) | ((r >> 3) << 11));
                    memcpy(dst + 2 * (size_t)i, &px, 2);
                    break;
                }
            }
        }
    }
    free(u);
    return AVIF_RESULT_OK;
}

/* ------------------------------------------------------------------------- */
/* RGB -> YUV (Appendix D.5)                                                  */

/* src/reformat_libyuv.c:285-381: which (range, RGB layout, YUV layout) have an entry */
static int rgbToYuvCovered(int full, int f, int yf)
{
    if (f < AVIF_RGB_FORMAT_RGB || f > AVIF_RGB_FORMAT_ABGR)
        return 0; /* 565 and the gray layouts have no entry */
    if (yf == AVIF_PIXEL_FORMAT_YUV400) /* :298-320 */
        return full ? (f != AVIF_RGB_FORMAT_ARGB) : (f == AVIF_RGB_FORMAT_BGRA);
    if (!colour(yf))
        return 0;
    if (!full)
        return 1; /* :339-348 */
    if (f == AVIF_RGB_FORMAT_RGB)
        return 1; /* RAWToJ444 / avifRAWToJ422 / RAWToJ420 */
    return sub2(yf); /* :352-358: no full-range 4:4:4 entry for the other layouts */
}

typedef struct Rgb8
{
    int r, g, b;
} Rgb8;

static Rgb8 pixelAt(const avifRGBImage * rgb, int f, uint32_t x, uint32_t y)
{
    const int alphaFirst = (f == AVIF_RGB_FORMAT_ARGB || f == AVIF_RGB_FORMAT_ABGR);
    const int n = (f == AVIF_RGB_FORMAT_RGB || f == AVIF_RGB_FORMAT_BGR) ? 3 : 4;
    const uint8_t * p = rgb->pixels + (size_t)y * rgb->rowBytes + (size_t)x * n + (alphaFirst ? 1 : 0);
    const int bgr = (f == AVIF_RGB_FORMAT_BGR || f == AVIF_RGB_FORMAT_BGRA || f == AVIF_RGB_FORMAT_ABGR);
    Rgb8 o;
    o.r = bgr ? p[2] : p[0];
    o.g = p[1];
    o.b = bgr ? p[0] : p[2];
    return o;
}

static int lumaOf(int full, Rgb8 c)
{
    return full ? ((77 * c.r + 150 * c.g + 29 * c.b + 128) >> 8) : ((66 * c.r + 129 * c.g + 25 * c.b + 0x1080) >> 8);
}
static int cbOf(int full, Rgb8 c)
{
    return full ? ((128 * c.b - 85 * c.g - 43 * c.r + 0x8000) >> 8) : ((112 * c.b - 74 * c.g - 38 * c.r + 0x8000) >> 8);
}
static int crOf(int full, Rgb8 c)
{
    return full ? ((128 * c.r - 107 * c.g - 21 * c.b + 0x8000) >> 8) : ((112 * c.r - 94 * c.g - 18 * c.b + 0x8000) >> 8);
}

/* avifImageRGBToYUVLibYUV, src/reformat_libyuv.c:270-381 */
avifResult oracleLibyuvHookRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    if (image->width > INT_MAX || image->height > INT_MAX || image->yuvRowBytes[0] > INT_MAX || rgb->rowBytes > INT_MAX)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (image->depth != 8 || rgb->depth != 8)
        return AVIF_RESULT_NOT_IMPLEMENTED; /* :277-282 */
    if (image->matrixCoefficients != AVIF_MATRIX_COEFFICIENTS_BT470BG && image->matrixCoefficients != AVIF_MATRIX_COEFFICIENTS_BT601)
        return AVIF_RESULT_NOT_IMPLEMENTED; /* :293 */
    const int full = (image->yuvRange == AVIF_RANGE_FULL);
    const int f = (int)rgb->format, yf = (int)image->yuvFormat;
    if (!rgbToYuvCovered(full, f, yf))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    const uint32_t W = image->width, H = image->height;
    for (uint32_t j = 0; j < H; ++j)
        for (uint32_t i = 0; i < W; ++i)
            image->yuvPlanes[0][(size_t)j * image->yuvRowBytes[0] + i] = (uint8_t)lumaOf(full, pixelAt(rgb, f, i, j));
    if (yf == AVIF_PIXEL_FORMAT_YUV400)
        return AVIF_RESULT_OK;
    const int sx = (yf != AVIF_PIXEL_FORMAT_YUV444), sy = (yf == AVIF_PIXEL_FORMAT_YUV420);
    const uint32_t CW = (W + sx) >> sx, CH = (H + sy) >> sy;
    for (uint32_t cj = 0; cj < CH; ++cj) {
        for (uint32_t ci = 0; ci < CW; ++ci) {
            /* the RGB of the covered pixels is averaged first, per channel, with the last column / row replicated */
            const uint32_t x0 = ci << sx, y0 = cj << sy;
            const uint32_t x1 = (sx && x0 + 1 < W) ? x0 + 1 : x0, y1 = (sy && y0 + 1 < H) ? y0 + 1 : y0;
            Rgb8 c;
            if (!sx) {
                c = pixelAt(rgb, f, x0, y0);
            } else if (!sy) {
                const Rgb8 a = pixelAt(rgb, f, x0, y0), b = pixelAt(rgb, f, x1, y0);
                c.r = (a.r + b.r + 1) >> 1, c.g = (a.g + b.g + 1) >> 1, c.b = (a.b + b.b + 1) >> 1;
            } else {
                const Rgb8 a = pixelAt(rgb, f, x0, y0), b = pixelAt(rgb, f, x1, y0), d = pixelAt(rgb, f, x0, y1), e = pixelAt(rgb, f, x1, y1);
                c.r = (a.r + b.r + d.r + e.r + 2) >> 2, c.g = (a.g + b.g + d.g + e.g + 2) >> 2, c.b = (a.b + b.b + d.b + e.b + 2) >> 2;
            }
            image->yuvPlanes[1][(size_t)cj * image->yuvRowBytes[1] + ci] = (uint8_t)cbOf(full, c);
            image->yuvPlanes[2][(size_t)cj * image->yuvRowBytes[2] + ci] = (uint8_t)crOf(full, c);
        }
    }
    return AVIF_RESULT_OK;
}

/* ------------------------------------------------------------------------- */
/* ARGBAttenuate / ARGBUnattenuate (Appendix D.4)                              */

static avifResult attenuatePass(avifRGBImage * rgb, int unattenuate) /* src/reformat_libyuv.c:1112-1161 */
{
    if (rgb->width > INT_MAX || rgb->height > INT_MAX || rgb->rowBytes > INT_MAX)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (rgb->depth != 8)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (rgb->format != AVIF_RGB_FORMAT_RGBA && rgb->format != AVIF_RGB_FORMAT_BGRA)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    for (uint32_t j = 0; j < rgb->height; ++j) {
        uint8_t * p = rgb->pixels + (size_t)j * rgb->rowBytes;
        for (uint32_t i = 0; i < rgb->width; ++i, p += 4) {
            const unsigned a = p[3];
            for (int c = 0; c < 3; ++c) {
                const unsigned fch = p[c];
                if (!unattenuate) {
                    p[c] = (uint8_t)((fch * a + 255) >> 8);
                } else {
                    const unsigned ia = (a == 0) ? 0u : (a == 1) ? 0xffffu : (a == 255) ? 0x100u : (0x10000u / a);
                    const unsigned t = ((fch * 0x101u) * ia) >> 16;
                    p[c] = (uint8_t)((t >= 0x8000u) ? 0u : (t > 255u ? 255u : t));
                }
            }
        }
    }
    return AVIF_RESULT_OK;
}

avifResult oracleLibyuvHookPremultiplyAlpha(avifRGBImage * rgb)
{
    return attenuatePass(rgb, 0);
}
avifResult oracleLibyuvHookUnpremultiplyAlpha(avifRGBImage * rgb)
{
    return attenuatePass(rgb, 1);
}

/* ------------------------------------------------------------------------- */
/* a libavif built with libyuv, end to end                                    */

static const OracleBackend kLibyuvBackend = { oracleLibyuvHookYUVToRGB, oracleLibyuvHookRGBToYUV, oracleLibyuvHookPremultiplyAlpha,
                                              oracleLibyuvHookUnpremultiplyAlpha };

avifResult oracleLibyuvImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    (void)rgbHasAlpha;
    return oracleImageYUVToRGBWithBackend(image, rgb, &kLibyuvBackend);
}
avifResult oracleLibyuvImageRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    return oracleImageRGBToYUVWithBackend(image, rgb, &kLibyuvBackend);
}
avifResult oracleLibyuvRGBImagePremultiplyAlpha(avifRGBImage * rgb)
{
    return oracleAlphaPassWithBackend(rgb, 0, &kLibyuvBackend);
}
avifResult oracleLibyuvRGBImageUnpremultiplyAlpha(avifRGBImage * rgb)
{
    return oracleAlphaPassWithBackend(rgb, 1, &kLibyuvBackend);
}
