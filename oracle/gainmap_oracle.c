/*
 * gainmap_oracle.c -- TEST INFRASTRUCTURE ONLY (see reformat_oracle.h): plain-C restatement of the reference's gain-map
 * application, avifRGBImageApplyGainMap / avifImageApplyGainMap (reference src/gainmap.c:73-355), with what it calls:
 * the transfer functions of src/colr.c:214-515, the primaries tables of src/colr.c:16-43, the RGB -> RGB matrices of
 * src/colrconvert.c:11-195 and the float pixel accessors of src/reformat.c:1842-1939.
 *
 * Every transcendental goes through this machine's libm (powf / exp2f / expf / logf / log10f / sqrtf), like the
 * reference's: compiled with the same compiler family and flags (no FMA contraction) the restatement is pinned
 * BIT-EXACT against avifRGBImageApplyGainMap of the reference built from its sources (tests/test_gainmap.py).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "reformat_oracle.h"

/* ---- transfer characteristics, src/colr.c:214-515 ---- */

typedef float (*TransferFn)(float);

#define CLAMPF(x, lo, hi) (((x) < (lo)) ? (lo) : (((hi) < (x)) ? (hi) : (x)))
#define MINF(a, b) (((a) < (b)) ? (a) : (b))
#define MAXF(a, b) (((a) > (b)) ? (a) : (b))

static float toLinear709(float g) /* :214-225 */
{
    if (g < 0.0f)
        return 0.0f;
    if (g < 4.5f * 0.018053968510807f)
        return g / 4.5f;
    if (g < 1.0f)
        return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
    return 1.0f;
}
static float toGamma709(float l) /* :227-238 */
{
    if (l < 0.0f)
        return 0.0f;
    if (l < 0.018053968510807f)
        return l * 4.5f;
    if (l < 1.0f)
        return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
    return 1.0f;
}
static float toLinear470M(float g) { return powf(CLAMPF(g, 0.0f, 1.0f), 2.2f); }         /* :240-243 */
static float toGamma470M(float l) { return powf(CLAMPF(l, 0.0f, 1.0f), 1.0f / 2.2f); }    /* :245-248 */
static float toLinear470BG(float g) { return powf(CLAMPF(g, 0.0f, 1.0f), 2.8f); }        /* :250-253 */
static float toGamma470BG(float l) { return powf(CLAMPF(l, 0.0f, 1.0f), 1.0f / 2.8f); }   /* :255-258 */
static float toLinearSMPTE240(float g) /* :260-271 */
{
    if (g < 0.0f)
        return 0.0f;
    if (g < 4.0f * 0.022821585529445f)
        return g / 4.0f;
    if (g < 1.0f)
        return powf((g + 0.111572195921731f) / 1.111572195921731f, 1.0f / 0.45f);
    return 1.0f;
}
static float toGammaSMPTE240(float l) /* :273-284 */
{
    if (l < 0.0f)
        return 0.0f;
    if (l < 0.022821585529445f)
        return l * 4.0f;
    if (l < 1.0f)
        return 1.111572195921731f * powf(l, 0.45f) - 0.111572195921731f;
    return 1.0f;
}
static float toGammaLinear(float g) { return CLAMPF(g, 0.0f, 1.0f); } /* :286-289, both directions */
static float toLinearLog100(float g)                                   /* :291-296 */
{
    const float mid = 0.01f / 2.f;
    return (g <= 0.0f) ? mid : powf(10.0f, 2.f * (MINF(g, 1.f) - 1.0f));
}
static float toGammaLog100(float l) { return l <= 0.01f ? 0.0f : 1.0f + log10f(MINF(l, 1.0f)) / 2.0f; } /* :298-301 */
static float toLinearLog100Sqrt10(float g)                                                                /* :303-308 */
{
    const float mid = 0.00316227766f / 2.f;
    return (g <= 0.0f) ? mid : powf(10.0f, 2.5f * (MINF(g, 1.f) - 1.0f));
}
static float toGammaLog100Sqrt10(float l) { return l <= 0.00316227766f ? 0.0f : 1.0f + log10f(MINF(l, 1.0f)) / 2.5f; } /* :310-313 */
static float toLinearIEC61966(float g)                                                                                 /* :315-324 */
{
    if (g < -4.5f * 0.018053968510807f)
        return -powf((g - 0.09929682680944f) / -1.09929682680944f, 1.0f / 0.45f);
    if (g < 4.5f * 0.018053968510807f)
        return g / 4.5f;
    return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
}
static float toGammaIEC61966(float l) /* :326-335 */
{
    if (l < -0.018053968510807f)
        return -1.09929682680944f * powf(-l, 0.45f) + 0.09929682680944f;
    if (l < 0.018053968510807f)
        return l * 4.5f;
    return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
}
static float toLinearBT1361(float g) /* :337-350 */
{
    if (g < -0.25f)
        return -0.25f;
    if (g < 0.0f)
        return powf((g - 0.02482420670236f) / -0.27482420670236f, 1.0f / 0.45f) / -4.0f;
    if (g < 4.5f * 0.018053968510807f)
        return g / 4.5f;
    if (g < 1.0f)
        return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
    return 1.0f;
}
static float toGammaBT1361(float l) /* :352-365 */
{
    if (l < -0.25f)
        return -0.25f;
    if (l < 0.0f)
        return -0.27482420670236f * powf(-4.0f * l, 0.45f) + 0.02482420670236f;
    if (l < 0.018053968510807f)
        return l * 4.5f;
    if (l < 1.0f)
        return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
    return 1.0f;
}
static float toLinearSRGB(float g) /* :367-378 */
{
    if (g < 0.0f)
        return 0.0f;
    if (g < 12.92f * 0.0030412825601275209f)
        return g / 12.92f;
    if (g < 1.0f)
        return powf((g + 0.0550107189475866f) / 1.0550107189475866f, 2.4f);
    return 1.0f;
}
static float toGammaSRGB(float l) /* :380-391 */
{
    if (l < 0.0f)
        return 0.0f;
    if (l < 0.0030412825601275209f)
        return l * 12.92f;
    if (l < 1.0f)
        return 1.0550107189475866f * powf(l, 1.0f / 2.4f) - 0.0550107189475866f;
    return 1.0f;
}
#define PQ_MAX_NITS 10000.0f
#define HLG_PEAK_NITS 1000.0f
#define SDR_WHITE_NITS 203.0f
static float toLinearPQ(float g) /* :397-409 */
{
    if (g > 0.0f) {
        const float powGamma = powf(g, 1.0f / 78.84375f);
        const float num = MAXF(powGamma - 0.8359375f, 0.0f);
        const float den = MAXF(18.8515625f - 18.6875f * powGamma, FLT_MIN);
        const float linear = powf(num / den, 1.0f / 0.1593017578125f);
        return linear * PQ_MAX_NITS / SDR_WHITE_NITS;
    }
    return 0.0f;
}
static float toGammaPQ(float l) /* :411-423 */
{
    if (l > 0.0f) {
        l = CLAMPF(l * SDR_WHITE_NITS / PQ_MAX_NITS, 0.0f, 1.0f);
        const float powLinear = powf(l, 0.1593017578125f);
        const float num = 0.1640625f * powLinear - 0.1640625f;
        const float den = 1.0f + 18.6875f * powLinear;
        return powf(1.0f + num / den, 78.84375f);
    }
    return 0.0f;
}
static float toLinearSMPTE428(float g) { return powf(MAXF(g, 0.0f), 2.6f) / 0.91655527974030934f; }            /* :425-428 */
static float toGammaSMPTE428(float l) { return powf(0.91655527974030934f * MAXF(l, 0.0f), 1.0f / 2.6f); }       /* :430-433 */
static float toLinearHLG(float g)                                                                              /* :439-455 */
{
    if (g < 0.0f)
        return 0.0f;
    float linear;
    if (g <= 0.5f)
        linear = powf((g * g) * (1.0f / 3.0f), 1.2f);
    else
        linear = powf((expf((g - 0.55991073f) / 0.17883277f) + 0.28466892f) / 12.0f, 1.2f);
    return linear * HLG_PEAK_NITS / SDR_WHITE_NITS;
}
static float toGammaHLG(float l) /* :457-470 */
{
    l = CLAMPF(l * SDR_WHITE_NITS / HLG_PEAK_NITS, 0.0f, 1.0f);
    l = powf(l, 1.0f / 1.2f);
    if (l < 0.0f)
        return 0.0f;
    if (l <= (1.0f / 12.0f))
        return sqrtf(3.0f * l);
    return 0.17883277f * logf(12.0f * l - 0.28466892f) + 0.55991073f;
}

static void transferFunctions(int tc, TransferFn * toLinear, TransferFn * toGamma) /* table :472-489, lookups :494-515 */
{
    *toLinear = toLinear709, *toGamma = toGamma709; /* BT.709, BT.601, BT.2020 10/12-bit, and "a reasonable default" */
    switch (tc) {
        case 4: *toLinear = toLinear470M, *toGamma = toGamma470M; break;
        case 5: *toLinear = toLinear470BG, *toGamma = toGamma470BG; break;
        case 7: *toLinear = toLinearSMPTE240, *toGamma = toGammaSMPTE240; break;
        case 8: *toLinear = toGammaLinear, *toGamma = toGammaLinear; break;
        case 9: *toLinear = toLinearLog100, *toGamma = toGammaLog100; break;
        case 10: *toLinear = toLinearLog100Sqrt10, *toGamma = toGammaLog100Sqrt10; break;
        case 11: *toLinear = toLinearIEC61966, *toGamma = toGammaIEC61966; break;
        case 12: *toLinear = toLinearBT1361, *toGamma = toGammaBT1361; break;
        case 13: *toLinear = toLinearSRGB, *toGamma = toGammaSRGB; break;
        case 16: *toLinear = toLinearPQ, *toGamma = toGammaPQ; break;
        case 17: *toLinear = toLinearSMPTE428, *toGamma = toGammaSMPTE428; break;
        case 18: *toLinear = toLinearHLG, *toGamma = toGammaHLG; break;
        default: break;
    }
}

/* for tests: one transfer function value (direction 0 = gamma -> linear, 1 = linear -> gamma) */
float oracleTransferFunction(int tc, int direction, float v)
{
    TransferFn a, b;
    transferFunctions(tc, &a, &b);
    return direction ? b(v) : a(v);
}

/* ---- colour primaries and RGB -> RGB matrices, src/colr.c:16-43, src/colrconvert.c:11-195 ---- */

static void primariesValues(int cp, float out[8])
{
    static const struct
    {
        int cp;
        float v[8];
    } table[] = { { 1, { 0.64f, 0.33f, 0.3f, 0.6f, 0.15f, 0.06f, 0.3127f, 0.329f } },
                  { 4, { 0.67f, 0.33f, 0.21f, 0.71f, 0.14f, 0.08f, 0.310f, 0.316f } },
                  { 5, { 0.64f, 0.33f, 0.29f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f } },
                  { 6, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
                  { 7, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
                  { 8, { 0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f, 0.316f } },
                  { 9, { 0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f } },
                  { 10, { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.3333f, 0.3333f } },
                  { 11, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f, 0.351f } },
                  { 12, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f } },
                  { 22, { 0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f } } };
    for (size_t i = 0; i < sizeof(table) / sizeof(table[0]); ++i) {
        if (table[i].cp == cp) {
            memcpy(out, table[i].v, sizeof(table[i].v));
            return;
        }
    }
    memcpy(out, table[0].v, sizeof(table[0].v)); /* unknown: "a reasonable default", :40-42 */
}

static const double kEpsilon = 1e-12;

static int matInv(double M[3][3], double I[3][3]) /* colrconvert.c:26-47 */
{
    double det = M[0][0] * (M[1][1] * M[2][2] - M[2][1] * M[1][2]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                 M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    if (fabs(det) < kEpsilon)
        return 0;
    det = 1.0 / det;
    I[0][0] = (M[1][1] * M[2][2] - M[2][1] * M[1][2]) * det;
    I[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * det;
    I[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * det;
    I[1][0] = (M[1][2] * M[2][0] - M[1][0] * M[2][2]) * det;
    I[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * det;
    I[1][2] = (M[1][0] * M[0][2] - M[0][0] * M[1][2]) * det;
    I[2][0] = (M[1][0] * M[2][1] - M[2][0] * M[1][1]) * det;
    I[2][1] = (M[2][0] * M[0][1] - M[0][0] * M[2][1]) * det;
    I[2][2] = (M[0][0] * M[1][1] - M[1][0] * M[0][1]) * det;
    return 1;
}
static void matMul(double A[3][3], double B[3][3], double C[3][3]) /* :50-61 */
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[r][c] = A[r][0] * B[0][c] + A[r][1] * B[1][c] + A[r][2] * B[2][c];
}
static void matDiag(const double d[3], double M[3][3]) /* :64-75 */
{
    memset(M, 0, 9 * sizeof(double));
    M[0][0] = d[0], M[1][1] = d[1], M[2][2] = d[2];
}
static void vecMul(double M[3][3], const double x[3], double y[3]) /* :78-83 */
{
    for (int r = 0; r < 3; ++r)
        y[r] = M[r][0] * x[0] + M[r][1] * x[1] + M[r][2] * x[2];
}
static int rgbToXyzD50(int cp, double coeffs[3][3]) /* :97-153 */
{
    static double bradford[3][3] = { { 0.8951, 0.2664, -0.1614 }, { -0.7502, 1.7135, 0.0367 }, { 0.0389, -0.0685, 1.0296 } };
    static const double lmsD50[3] = { 0.996284, 1.02043, 0.818644 };
    float p[8];
    primariesValues(cp, p);
    if (fabsf(p[7]) < kEpsilon) /* avifXyToXYZ, :11-23 */
        return 0;
    double white[3];
    const double factor = 1.0 / p[7];
    white[0] = p[6] * factor, white[1] = 1, white[2] = (1 - p[6] - p[7]) * factor;
    double prim[3][3] = { { p[0], p[2], p[4] }, { p[1], p[3], p[5] }, { 1.0 - p[0] - p[1], 1.0 - p[2] - p[3], 1.0 - p[4] - p[5] } };
    double primInv[3][3];
    if (!matInv(prim, primInv))
        return 0;
    double rgbCoefficients[3], rgbCoefficientsMat[3][3], rgbXYZ[3][3];
    vecMul(primInv, white, rgbCoefficients);
    matDiag(rgbCoefficients, rgbCoefficientsMat);
    matMul(prim, rgbCoefficientsMat, rgbXYZ);
    double lms[3];
    vecMul(bradford, white, lms);
    for (int i = 0; i < 3; ++i) {
        if (fabs(lms[i]) < kEpsilon)
            return 0;
        lms[i] = lmsD50[i] / lms[i];
    }
    double adaptation[3][3], tmp[3][3], bradfordInv[3][3];
    matDiag(lms, adaptation);
    matMul(adaptation, bradford, tmp);
    if (!matInv(bradford, bradfordInv))
        return 0;
    matMul(bradfordInv, tmp, adaptation);
    matMul(adaptation, rgbXYZ, coeffs);
    return 1;
}
/* avifColorPrimariesComputeRGBToRGBMatrix, :163-179 */
int oracleColorPrimariesComputeRGBToRGBMatrix(int src, int dst, double coeffs[3][3])
{
    double srcToXyz[3][3], dstToXyz[3][3], xyzToDst[3][3];
    if (!rgbToXyzD50(src, srcToXyz) || !rgbToXyzD50(dst, dstToXyz) || !matInv(dstToXyz, xyzToDst))
        return 0;
    matMul(xyzToDst, srcToXyz, coeffs);
    return 1;
}
static void convertColorSpace(float rgb[4], double coeffs[3][3]) /* avifLinearRGBConvertColorSpace, :186-195 */
{
    const double in[3] = { rgb[0], rgb[1], rgb[2] };
    double out[3];
    vecMul(coeffs, in, out);
    rgb[0] = (float)out[0], rgb[1] = (float)out[1], rgb[2] = (float)out[2];
}

/* ---- float pixel accessors, src/reformat.c:32-117 (layout), :1842-1939 ---- */

typedef struct PixelLayout
{
    uint32_t channelBytes, pixelBytes, offR, offG, offB, offA;
    int hasAlpha, is565;
    float maxF;
} PixelLayout;

static int pixelLayout(const avifRGBImage * rgb, PixelLayout * L)
{
    if (rgb->depth != 8 && rgb->depth != 10 && rgb->depth != 12 && rgb->depth != 16)
        return 0;
    if (rgb->isFloat && rgb->depth != 16)
        return 0;
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565 && rgb->depth != 8)
        return 0;
    memset(L, 0, sizeof(*L));
    L->channelBytes = (rgb->depth > 8) ? 2 : 1;
    const uint32_t cb = L->channelBytes;
    int n = 0;
    switch (rgb->format) {
        case AVIF_RGB_FORMAT_RGB: L->offR = 0, L->offG = cb, L->offB = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_RGBA: L->offR = 0, L->offG = cb, L->offB = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ARGB: L->offA = 0, L->offR = cb, L->offG = 2 * cb, L->offB = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_BGR: L->offB = 0, L->offG = cb, L->offR = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_BGRA: L->offB = 0, L->offG = cb, L->offR = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ABGR: L->offA = 0, L->offB = cb, L->offG = 2 * cb, L->offR = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_RGB_565: L->is565 = 1, n = 2; break;
        default: return 0; /* gray formats: the gain-map code indexes R, G, B; not exercised by the reference's tests */
    }
    L->pixelBytes = L->is565 ? 2 : n * cb;
    L->hasAlpha = (n == 4);
    L->maxF = (float)((1 << rgb->depth) - 1);
    return 1;
}

#define F16_MULTIPLIER 1.9259299444e-34f
static float f16ToFloat(uint16_t v)
{
    union
    {
        float f;
        uint32_t u;
    } x;
    x.u = (uint32_t)v << 13;
    return x.f / F16_MULTIPLIER;
}
static uint16_t floatToF16(float v)
{
    union
    {
        float f;
        uint32_t u;
    } x;
    x.f = v * F16_MULTIPLIER;
    return (uint16_t)(x.u >> 13);
}

static void getPixel(const avifRGBImage * src, uint32_t x, uint32_t y, const PixelLayout * L, float out[4])
{
    const uint8_t * p = &src->pixels[(size_t)y * src->rowBytes + (size_t)x * L->pixelBytes];
    if (L->channelBytes > 1) {
        const uint16_t r = *(const uint16_t *)&p[L->offR], g = *(const uint16_t *)&p[L->offG], b = *(const uint16_t *)&p[L->offB];
        const uint16_t a = L->hasAlpha ? *(const uint16_t *)&p[L->offA] : (uint16_t)((1 << src->depth) - 1);
        if (src->isFloat) {
            out[0] = f16ToFloat(r), out[1] = f16ToFloat(g), out[2] = f16ToFloat(b);
            out[3] = L->hasAlpha ? f16ToFloat(a) : 1.0f;
        } else {
            out[0] = r / L->maxF, out[1] = g / L->maxF, out[2] = b / L->maxF, out[3] = a / L->maxF;
        }
    } else if (L->is565) {
        const uint16_t v = *(const uint16_t *)p; /* avifGetRGB565, :635-647 */
        const uint16_t r5 = (v >> 11) & 0x1f, g6 = (v >> 5) & 0x3f, b5 = v & 0x1f;
        out[0] = (uint8_t)((r5 << 3) | (r5 >> 2)) / L->maxF;
        out[1] = (uint8_t)((g6 << 2) | (g6 >> 4)) / L->maxF;
        out[2] = (uint8_t)((b5 << 3) | (b5 >> 2)) / L->maxF;
        out[3] = 1.0f;
    } else {
        out[0] = p[L->offR] / L->maxF, out[1] = p[L->offG] / L->maxF, out[2] = p[L->offB] / L->maxF;
        out[3] = L->hasAlpha ? (p[L->offA] / L->maxF) : 1.0f;
    }
}

static void setPixel(const avifRGBImage * dst, uint32_t x, uint32_t y, const PixelLayout * L, const float in[4])
{
    uint8_t * p = &dst->pixels[(size_t)y * dst->rowBytes + (size_t)x * L->pixelBytes];
    if (dst->depth > 8) {
        if (dst->isFloat) {
            *(uint16_t *)&p[L->offR] = floatToF16(in[0]), *(uint16_t *)&p[L->offG] = floatToF16(in[1]), *(uint16_t *)&p[L->offB] = floatToF16(in[2]);
            if (L->hasAlpha)
                *(uint16_t *)&p[L->offA] = floatToF16(in[3]);
        } else {
            *(uint16_t *)&p[L->offR] = (uint16_t)(0.5f + (in[0] * L->maxF));
            *(uint16_t *)&p[L->offG] = (uint16_t)(0.5f + (in[1] * L->maxF));
            *(uint16_t *)&p[L->offB] = (uint16_t)(0.5f + (in[2] * L->maxF));
            if (L->hasAlpha)
                *(uint16_t *)&p[L->offA] = (uint16_t)(0.5f + (in[3] * L->maxF));
        }
    } else {
        const uint8_t r = (uint8_t)(0.5f + (in[0] * L->maxF)), g = (uint8_t)(0.5f + (in[1] * L->maxF)), b = (uint8_t)(0.5f + (in[2] * L->maxF));
        if (L->is565) {
            *(uint16_t *)p = (uint16_t)((b >> 3) | ((g >> 2) << 5) | ((r >> 3) << 11)); /* avifStoreRGB8Pixel, :619-633 */
        } else {
            p[L->offR] = r, p[L->offG] = g, p[L->offB] = b;
            if (L->hasAlpha)
                p[L->offA] = (uint8_t)(0.5f + (in[3] * L->maxF));
        }
    }
}

/* ---- gain map application, src/gainmap.c ---- */

static float nanSafeClamp(float v) { return fminf(1.0f, fmaxf(0.0f, v)); } /* :13-16 */
static float sFrac(avifSignedFraction f) { return f.d == 0 ? 0.0f : (float)f.n / f.d; }
static float uFrac(avifUnsignedFraction f) { return f.d == 0 ? 0.0f : (float)f.n / f.d; }

/* avifGetGainMapWeight, :52-63 */
float oracleGainMapWeight(float hdrHeadroom, const avifGainMap * gm)
{
    const float base = uFrac(gm->baseHdrHeadroom), alt = uFrac(gm->alternateHdrHeadroom);
    if (base == alt)
        return 0.0f;
    const float w = CLAMPF((hdrHeadroom - base) / (alt - base), 0.0f, 1.0f);
    return (alt < base) ? -w : w;
}
static float lerpf(float a, float b, float w) { return (1.0f - w) * a + w * b; } /* :66-69 */

/* avifGainMapValidateMetadata, :430-457 */
avifResult oracleGainMapValidateMetadata(const avifGainMap * gm)
{
    for (int i = 0; i < 3; ++i) {
        if (gm->gainMapMin[i].d == 0 || gm->gainMapMax[i].d == 0 || gm->gainMapGamma[i].d == 0 || gm->baseOffset[i].d == 0 ||
            gm->alternateOffset[i].d == 0)
            return AVIF_RESULT_INVALID_ARGUMENT;
        if ((int64_t)gm->gainMapMax[i].n * gm->gainMapMin[i].d < (int64_t)gm->gainMapMin[i].n * gm->gainMapMax[i].d)
            return AVIF_RESULT_INVALID_ARGUMENT;
        if (gm->gainMapGamma[i].n == 0)
            return AVIF_RESULT_INVALID_ARGUMENT;
    }
    if (gm->baseHdrHeadroom.d == 0 || gm->alternateHdrHeadroom.d == 0)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (gm->useBaseColorSpace != 0 && gm->useBaseColorSpace != 1)
        return AVIF_RESULT_INVALID_ARGUMENT;
    return AVIF_RESULT_OK;
}

static uint32_t rgbPixelSize(const avifRGBImage * rgb) /* src/avif.c:692-698 */
{
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return 2;
    const uint32_t n = (rgb->format == AVIF_RGB_FORMAT_GRAY) ? 1
                       : (rgb->format == AVIF_RGB_FORMAT_GRAYA || rgb->format == AVIF_RGB_FORMAT_AGRAY) ? 2
                       : (rgb->format == AVIF_RGB_FORMAT_RGB || rgb->format == AVIF_RGB_FORMAT_BGR)     ? 3
                                                                                                        : 4;
    return n * ((rgb->depth > 8) ? 2 : 1);
}
static avifResult allocatePixels(avifRGBImage * rgb) /* avifRGBImageAllocatePixels, src/avif.c:719-737 */
{
    free(rgb->pixels);
    rgb->pixels = NULL, rgb->rowBytes = 0;
    const uint32_t px = rgbPixelSize(rgb);
    if (rgb->width == 0 || rgb->height == 0 || rgb->width > UINT32_MAX / px)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t rowBytes = rgb->width * px;
    rgb->pixels = (uint8_t *)malloc((size_t)rowBytes * rgb->height);
    if (!rgb->pixels)
        return AVIF_RESULT_OUT_OF_MEMORY;
    rgb->rowBytes = rowBytes;
    return AVIF_RESULT_OK;
}

/*
 * avifRGBImageApplyGainMap, src/gainmap.c:73-315.  `libyuvBuild`: how the gain map's own YUV -> RGB conversion (:211) is
 * computed -- by a libavif built with libyuv (the default build, integer path where libyuv serves) or without (fp32).
 * toneMappedImage->pixels must be NULL or malloc'ed: it is replaced (:112-114).
 */
avifResult oracleRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                      avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                      avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                      avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, int libyuvBuild)
{
    if (hdrHeadroom < 0.0f)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseImage == NULL || gainMap == NULL || toneMappedImage == NULL)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifResult res = oracleGainMapValidateMetadata(gainMap);
    if (res != AVIF_RESULT_OK)
        return res;

    const uint32_t width = baseImage->width, height = baseImage->height;
    const avifColorPrimaries mathPrimaries =
        (gainMap->useBaseColorSpace || (gainMap->altColorPrimaries == AVIF_COLOR_PRIMARIES_UNSPECIFIED)) ? baseColorPrimaries : gainMap->altColorPrimaries;
    const int needsInputConversion = (baseColorPrimaries != mathPrimaries);
    const int needsOutputConversion = (mathPrimaries != outputColorPrimaries);

    toneMappedImage->width = width, toneMappedImage->height = height;
    res = allocatePixels(toneMappedImage);
    if (res != AVIF_RESULT_OK)
        return res;

    const float weight = oracleGainMapWeight(hdrHeadroom, gainMap);
    if (weight == 0.0f && outputTransferCharacteristics == baseTransferCharacteristics && outputColorPrimaries == baseColorPrimaries &&
        baseImage->format == toneMappedImage->format && baseImage->depth == toneMappedImage->depth && baseImage->isFloat == toneMappedImage->isFloat &&
        baseImage->rowBytes == toneMappedImage->rowBytes) {
        memcpy(toneMappedImage->pixels, baseImage->pixels, (size_t)baseImage->rowBytes * baseImage->height);
        return AVIF_RESULT_OK;
    }

    PixelLayout baseL, outL;
    if (!pixelLayout(baseImage, &baseL) || !pixelLayout(toneMappedImage, &outL))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    TransferFn gammaToLinear, linearToGamma, unused;
    transferFunctions(baseTransferCharacteristics, &gammaToLinear, &unused);
    transferFunctions(outputTransferCharacteristics, &unused, &linearToGamma);

    if (weight == 0.0f) { /* :142-170 */
        const int primariesDiffer = (baseColorPrimaries != outputColorPrimaries);
        double coeffs[3][3];
        if (primariesDiffer && !oracleColorPrimariesComputeRGBToRGBMatrix(baseColorPrimaries, outputColorPrimaries, coeffs))
            return AVIF_RESULT_NOT_IMPLEMENTED;
        for (uint32_t j = 0; j < height; ++j) {
            for (uint32_t i = 0; i < width; ++i) {
                float px[4];
                getPixel(baseImage, i, j, &baseL, px);
                if (outputTransferCharacteristics != baseTransferCharacteristics || primariesDiffer) {
                    for (int c = 0; c < 3; ++c)
                        px[c] = gammaToLinear(px[c]);
                    if (primariesDiffer)
                        convertColorSpace(px, coeffs);
                    for (int c = 0; c < 3; ++c)
                        px[c] = nanSafeClamp(linearToGamma(px[c]));
                }
                setPixel(toneMappedImage, i, j, &outL, px);
            }
        }
        return AVIF_RESULT_OK;
    }

    double inputCoeffs[3][3], outputCoeffs[3][3];
    if (needsInputConversion && !oracleColorPrimariesComputeRGBToRGBMatrix(baseColorPrimaries, mathPrimaries, inputCoeffs))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    if (needsOutputConversion && !oracleColorPrimariesComputeRGBToRGBMatrix(mathPrimaries, outputColorPrimaries, outputCoeffs))
        return AVIF_RESULT_NOT_IMPLEMENTED;

    /* gain map pixels as RGB at the base image's size, :185-212 */
    avifImage gm = *gainMap->image; /* a view: planes are not owned */
    gm.imageOwnsYUVPlanes = AVIF_FALSE, gm.imageOwnsAlphaPlane = AVIF_FALSE;
    int scaled = 0;
    if (gm.width != width || gm.height != height) {
        res = oracleImageScale(&gm, width, height);
        if (res != AVIF_RESULT_OK)
            return res;
        scaled = 1;
    }
    avifRGBImage rgbGainMap;
    memset(&rgbGainMap, 0, sizeof(rgbGainMap));
    rgbGainMap.width = gm.width, rgbGainMap.height = gm.height, rgbGainMap.depth = gm.depth; /* avifRGBImageSetDefaults, src/avif.c:700-717 */
    rgbGainMap.format = AVIF_RGB_FORMAT_RGBA;
    rgbGainMap.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgbGainMap.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    rgbGainMap.maxThreads = 1;
    res = allocatePixels(&rgbGainMap);
    if (res == AVIF_RESULT_OK)
        res = libyuvBuild ? oracleLibyuvImageYUVToRGB(&gm, &rgbGainMap) : oracleImageYUVToRGB(&gm, &rgbGainMap);
    PixelLayout gmL;
    if (res == AVIF_RESULT_OK && !pixelLayout(&rgbGainMap, &gmL))
        res = AVIF_RESULT_NOT_IMPLEMENTED;

    if (res == AVIF_RESULT_OK) {
        float rgbMaxLinear = 0, rgbSumLinear = 0;
        float gammaInv[3], gmMin[3], gmMax[3], baseOffset[3], altOffset[3];
        for (int c = 0; c < 3; ++c) {
            gammaInv[c] = 1.0f / uFrac(gainMap->gainMapGamma[c]);
            gmMin[c] = sFrac(gainMap->gainMapMin[c]), gmMax[c] = sFrac(gainMap->gainMapMax[c]);
            baseOffset[c] = sFrac(gainMap->baseOffset[c]), altOffset[c] = sFrac(gainMap->alternateOffset[c]);
        }
        for (uint32_t j = 0; j < height && res == AVIF_RESULT_OK; ++j) {
            for (uint32_t i = 0; i < width; ++i) {
                float base[4], g[4], out[4];
                getPixel(baseImage, i, j, &baseL, base);
                getPixel(&rgbGainMap, i, j, &gmL, g);
                float pixelMax = 0.0f;
                for (int c = 0; c < 3; ++c)
                    base[c] = gammaToLinear(base[c]);
                if (needsInputConversion)
                    convertColorSpace(base, inputCoeffs);
                for (int c = 0; c < 3; ++c) {
                    const float gainMapLog2 = lerpf(gmMin[c], gmMax[c], powf(g[c], gammaInv[c]));
                    const float tone = (base[c] + baseOffset[c]) * exp2f(gainMapLog2 * weight) - altOffset[c];
                    if (tone > rgbMaxLinear)
                        rgbMaxLinear = tone;
                    if (tone > pixelMax)
                        pixelMax = tone;
                    out[c] = tone;
                }
                if (needsOutputConversion)
                    convertColorSpace(out, outputCoeffs);
                int bad = 0;
                for (int c = 0; c < 3; ++c) {
                    if (isnan(out[c])) {
                        bad = 1;
                        break;
                    }
                    out[c] = nanSafeClamp(linearToGamma(out[c]));
                }
                if (bad) {
                    res = AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE;
                    break;
                }
                out[3] = base[3];
                rgbSumLinear += pixelMax;
                setPixel(toneMappedImage, i, j, &outL, out);
            }
        }
        if (res == AVIF_RESULT_OK && clli != NULL) { /* :292-302 */
            clli->maxCLL = (uint16_t)CLAMPF(floorf(rgbMaxLinear * SDR_WHITE_NITS + 0.5f), 0.0f, (float)UINT16_MAX);
            const float average = rgbSumLinear / ((size_t)width * height);
            clli->maxPALL = (uint16_t)CLAMPF(floorf(average * SDR_WHITE_NITS + 0.5f), 0.0f, (float)UINT16_MAX);
        }
    }
    free(rgbGainMap.pixels);
    if (scaled) {
        for (int p = 0; p < 3; ++p)
            free(gm.yuvPlanes[p]);
        free(gm.alphaPlane);
    }
    return res;
}

/* ---- gain-map computation (the encode side), src/gainmap.c:357-428, :493-843 ---- */

static const float kGainEpsilon = 1e-10f; /* :487 */

static float roundfHalfUp(float v) { return floorf(v + 0.5f); } /* avifRoundf, src/utils.c:11-14 */

/* avifDoubleToUnsignedFractionImpl, src/utils.c:238-281: best continued-fraction approximation */
static int doubleToFraction(double v, uint32_t maxNumerator, uint32_t * numerator, uint32_t * denominator)
{
    if (isnan(v) || v < 0 || v > maxNumerator)
        return 0;
    const uint32_t maxD = (v <= 1) ? UINT32_MAX : (uint32_t)floor(maxNumerator / v);
    *denominator = 1;
    uint32_t previousD = 0;
    double currentV = v - floor(v);
    for (int iter = 0; iter < 39; ++iter) {
        const double numeratorDouble = (double)(*denominator) * v;
        *numerator = (uint32_t)round(numeratorDouble);
        if (fabs(numeratorDouble - (*numerator)) == 0.0)
            return 1;
        currentV = 1.0 / currentV;
        const double newD = previousD + floor(currentV) * (*denominator);
        if (newD > (double)maxD)
            return 1;
        previousD = *denominator;
        *denominator = (uint32_t)newD;
        currentV -= floor(currentV);
    }
    *numerator = (uint32_t)round((double)(*denominator) * v);
    return 1;
}
int oracleDoubleToSignedFraction(double v, avifSignedFraction * f) /* :283-294 */
{
    uint32_t n;
    if (!doubleToFraction(fabs(v), INT32_MAX, &n, &f->d))
        return 0;
    f->n = (int32_t)n;
    if (v < 0)
        f->n *= -1;
    return 1;
}
int oracleDoubleToUnsignedFraction(double v, avifUnsignedFraction * f) /* :296-299 */
{
    return doubleToFraction(v, UINT32_MAX, &f->n, &f->d);
}

/* avifColorPrimariesComputeYCoeffs, src/colr.c:517-542 */
static void yCoefficients(int cp, float coeffs[3])
{
    float p[8];
    primariesValues(cp, p);
    const float rX = p[0], rY = p[1], gX = p[2], gY = p[3], bX = p[4], bY = p[5], wX = p[6], wY = p[7];
    const float rZ = 1.0f - (rX + rY), gZ = 1.0f - (gX + gY), bZ = 1.0f - (bX + bY), wZ = 1.0f - (wX + wY);
    const float kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) /
                     (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    const float kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) /
                     (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    coeffs[0] = kr, coeffs[2] = kb, coeffs[1] = 1.0f - coeffs[0] - coeffs[2];
}

/* avifChooseColorSpaceForGainMapMath, src/gainmap.c:496-533 */
static avifResult chooseMathPrimaries(int basePrimaries, int altPrimaries, int * out)
{
    if (basePrimaries == altPrimaries) {
        *out = basePrimaries;
        return AVIF_RESULT_OK;
    }
    double baseToAlt[3][3], altToBase[3][3];
    if (!oracleColorPrimariesComputeRGBToRGBMatrix(basePrimaries, altPrimaries, baseToAlt) ||
        !oracleColorPrimariesComputeRGBToRGBMatrix(altPrimaries, basePrimaries, altToBase))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    float baseMin = 0, altMin = 0;
    for (int c = 0; c < 3; ++c) {
        float rgba[4] = { 0, 0, 0, 0 };
        rgba[c] = 1.0f;
        convertColorSpace(rgba, altToBase);
        for (int i = 0; i < 3; ++i)
            baseMin = MINF(baseMin, rgba[i]);
        rgba[0] = rgba[1] = rgba[2] = 0;
        rgba[c] = 1.0f;
        convertColorSpace(rgba, baseToAlt);
        for (int i = 0; i < 3; ++i)
            altMin = MINF(altMin, rgba[i]);
    }
    *out = (altMin <= baseMin) ? basePrimaries : altPrimaries;
    return AVIF_RESULT_OK;
}

/* avifFindMinMaxWithoutOutliers, :375-428 */
static int valueToBucket(float v, float lo, float hi, int n) /* :363-367 */
{
    v = CLAMPF(v, lo, hi);
    const int idx = (int)roundfHalfUp((v - lo) / (hi - lo) * n);
    return idx < n - 1 ? idx : n - 1;
}
static float bucketToValue(int idx, float lo, float hi, int n) { return idx * (hi - lo) / n + lo; } /* :369-372 */
avifResult oracleFindMinMaxWithoutOutliers(const float * g, size_t numPixels, float * rangeMin, float * rangeMax)
{
    const float bucketSize = 0.01f, maxOutliersRatio = 0.001f;
    const int maxOutliersOnEachSide = (int)roundfHalfUp(numPixels * maxOutliersRatio / 2.0f);
    float lo = g[0], hi = g[0];
    for (size_t i = 1; i < numPixels; ++i) {
        lo = MINF(lo, g[i]);
        hi = MAXF(hi, g[i]);
    }
    *rangeMin = lo, *rangeMax = hi;
    if ((hi - lo) <= (bucketSize * 2) || maxOutliersOnEachSide == 0)
        return AVIF_RESULT_OK;
    const int maxNumBuckets = 10000;
    const int byWidth = (int)ceilf((hi - lo) / bucketSize);
    const int numBuckets = byWidth < maxNumBuckets ? byWidth : maxNumBuckets;
    int * histogram = (int *)calloc((size_t)numBuckets, sizeof(int));
    if (!histogram)
        return AVIF_RESULT_OUT_OF_MEMORY;
    for (size_t i = 0; i < numPixels; ++i)
        ++histogram[valueToBucket(g[i], lo, hi, numBuckets)];
    int leftOutliers = 0;
    for (int i = 0; i < numBuckets; ++i) {
        leftOutliers += histogram[i];
        if (leftOutliers > maxOutliersOnEachSide)
            break;
        if (histogram[i] == 0)
            *rangeMin = bucketToValue(i + 1, lo, hi, numBuckets);
    }
    int rightOutliers = 0;
    for (int i = numBuckets - 1; i >= 0; --i) {
        rightOutliers += histogram[i];
        if (rightOutliers > maxOutliersOnEachSide)
            break;
        if (histogram[i] == 0)
            *rangeMax = bucketToValue(i, lo, hi, numBuckets);
    }
    free(histogram);
    return AVIF_RESULT_OK;
}

static void freePlanes(avifImage * image) /* avifImageFreePlanes(ALL), src/avif.c:492-517 */
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

/*
 * avifRGBImageComputeGainMap, src/gainmap.c:535-843.  gainMap->image carries the requested width / height / depth / format
 * (and range / matrix) on entry and owns malloc'ed planes on success; `libyuvBuild` as in oracleRGBImageApplyGainMap (the
 * gain map's RGB -> YUV conversion, :814).
 */
avifResult oracleRGBImageComputeGainMap(const avifRGBImage * baseRgb, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC,
                                        const avifRGBImage * altRgb, avifColorPrimaries altPrimaries, avifTransferCharacteristics altTC,
                                        avifGainMap * gainMap, int libyuvBuild)
{
    if (baseRgb == NULL || altRgb == NULL || gainMap == NULL || gainMap->image == NULL)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseRgb->width != altRgb->width || baseRgb->height != altRgb->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifImage * gmImage = gainMap->image;
    if (gmImage->width == 0 || gmImage->height == 0 || gmImage->depth == 0 || (int)gmImage->yuvFormat <= 0 || (int)gmImage->yuvFormat >= 5)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const int colorSpacesDiffer = (basePrimaries != altPrimaries);
    int mathPrimaries;
    avifResult res = chooseMathPrimaries(basePrimaries, altPrimaries, &mathPrimaries);
    if (res != AVIF_RESULT_OK)
        return res;
    const uint32_t width = baseRgb->width, height = baseRgb->height;
    PixelLayout baseL, altL;
    if (!pixelLayout(baseRgb, &baseL) || !pixelLayout(altRgb, &altL))
        return AVIF_RESULT_NOT_IMPLEMENTED;

    const size_t numPixels = (size_t)width * height;
    const int singleChannel = (gmImage->yuvFormat == AVIF_PIXEL_FORMAT_YUV400);
    const int numChannels = singleChannel ? 1 : 3;
    float * gainMapF[3] = { NULL, NULL, NULL };
    avifRGBImage gainMapRGB;
    memset(&gainMapRGB, 0, sizeof(gainMapRGB));
    for (int c = 0; c < numChannels; ++c) {
        gainMapF[c] = (float *)malloc(numPixels * sizeof(float));
        if (!gainMapF[c]) {
            res = AVIF_RESULT_OUT_OF_MEMORY;
            goto cleanup;
        }
    }

    for (int i = 0; i < 3; ++i) { /* avifGainMapSetEncodingDefaults, :18-30 */
        gainMap->gainMapMin[i] = (avifSignedFraction) { 1, 1 }, gainMap->gainMapMax[i] = (avifSignedFraction) { 1, 1 };
        gainMap->baseOffset[i] = (avifSignedFraction) { 1, 64 }, gainMap->alternateOffset[i] = (avifSignedFraction) { 1, 64 };
        gainMap->gainMapGamma[i] = (avifUnsignedFraction) { 1, 1 };
    }
    gainMap->baseHdrHeadroom = (avifUnsignedFraction) { 0, 1 }, gainMap->alternateHdrHeadroom = (avifUnsignedFraction) { 1, 1 };
    gainMap->useBaseColorSpace = (mathPrimaries == basePrimaries);

    TransferFn baseToLinear, altToLinear, unused;
    transferFunctions(baseTC, &baseToLinear, &unused);
    transferFunctions(altTC, &altToLinear, &unused);
    float yCoeffs[3];
    yCoefficients(mathPrimaries, yCoeffs);
    double coeffs[3][3];
    if (colorSpacesDiffer) {
        const int ok = gainMap->useBaseColorSpace ? oracleColorPrimariesComputeRGBToRGBMatrix(altPrimaries, basePrimaries, coeffs)
                                                  : oracleColorPrimariesComputeRGBToRGBMatrix(basePrimaries, altPrimaries, coeffs);
        if (!ok) {
            res = AVIF_RESULT_NOT_IMPLEMENTED;
            goto cleanup;
        }
    }
    float baseOffset[3], altOffset[3];
    for (int c = 0; c < 3; ++c)
        baseOffset[c] = sFrac(gainMap->baseOffset[c]), altOffset[c] = sFrac(gainMap->alternateOffset[c]);

    if (colorSpacesDiffer) { /* offsets that keep converted channels positive, :618-660 */
        float rgba[4] = { 0 }, channelMin[3] = { 0 };
        for (uint32_t j = 0; j < height; ++j) {
            for (uint32_t i = 0; i < width; ++i) {
                getPixel(gainMap->useBaseColorSpace ? altRgb : baseRgb, i, j, gainMap->useBaseColorSpace ? &altL : &baseL, rgba);
                for (int c = 0; c < 3; ++c)
                    rgba[c] = gainMap->useBaseColorSpace ? altToLinear(rgba[c]) : baseToLinear(rgba[c]);
                convertColorSpace(rgba, coeffs);
                for (int c = 0; c < 3; ++c)
                    channelMin[c] = MINF(channelMin[c], rgba[c]);
            }
        }
        for (int c = 0; c < 3; ++c) {
            const float maxOffset = 0.1f;
            if (channelMin[c] < -kGainEpsilon) {
                if (gainMap->useBaseColorSpace)
                    altOffset[c] = MINF(altOffset[c] - channelMin[c], maxOffset);
                else
                    baseOffset[c] = MINF(baseOffset[c] - channelMin[c], maxOffset);
            }
        }
    }

    float baseMax = 1.0f, altMax = 1.0f; /* raw log2 ratios, :663-715 */
    for (uint32_t j = 0; j < height; ++j) {
        for (uint32_t i = 0; i < width; ++i) {
            float b[4], a[4];
            getPixel(baseRgb, i, j, &baseL, b);
            getPixel(altRgb, i, j, &altL, a);
            for (int c = 0; c < 3; ++c)
                b[c] = baseToLinear(b[c]), a[c] = altToLinear(a[c]);
            if (colorSpacesDiffer)
                convertColorSpace(gainMap->useBaseColorSpace ? a : b, coeffs);
            for (int c = 0; c < numChannels; ++c) {
                float base = b[c], alt = a[c];
                if (singleChannel) {
                    base = yCoeffs[0] * b[0] + yCoeffs[1] * b[1] + yCoeffs[2] * b[2];
                    alt = yCoeffs[0] * a[0] + yCoeffs[1] * a[1] + yCoeffs[2] * a[2];
                }
                if (base > baseMax)
                    baseMax = base;
                if (alt > altMax)
                    altMax = alt;
                const float ratio = (alt + altOffset[c]) / (base + baseOffset[c]);
                gainMapF[c][(size_t)j * width + i] = log2f(MAXF(ratio, kGainEpsilon));
            }
        }
    }

    const double baseHeadroom = log2f(MAXF(baseMax, kGainEpsilon)), altHeadroom = log2f(MAXF(altMax, kGainEpsilon));
    if (!oracleDoubleToUnsignedFraction(baseHeadroom, &gainMap->baseHdrHeadroom) || !oracleDoubleToUnsignedFraction(altHeadroom, &gainMap->alternateHdrHeadroom)) {
        res = AVIF_RESULT_INVALID_ARGUMENT;
        goto cleanup;
    }
    if (altHeadroom < baseHeadroom)
        for (int c = 0; c < numChannels; ++c)
            for (size_t k = 0; k < numPixels; ++k)
                gainMapF[c][k] *= -1.f;

    float minLog2[3] = { 0, 0, 0 }, maxLog2[3] = { 0, 0, 0 };
    for (int c = 0; c < numChannels; ++c) {
        res = oracleFindMinMaxWithoutOutliers(gainMapF[c], numPixels, &minLog2[c], &maxLog2[c]);
        if (res != AVIF_RESULT_OK)
            goto cleanup;
    }
    for (int c = 0; c < 3; ++c) {
        if (!oracleDoubleToSignedFraction(minLog2[singleChannel ? 0 : c], &gainMap->gainMapMin[c]) ||
            !oracleDoubleToSignedFraction(maxLog2[singleChannel ? 0 : c], &gainMap->gainMapMax[c]) ||
            !oracleDoubleToSignedFraction(altOffset[c], &gainMap->alternateOffset[c]) || !oracleDoubleToSignedFraction(baseOffset[c], &gainMap->baseOffset[c])) {
            res = AVIF_RESULT_INVALID_ARGUMENT;
            goto cleanup;
        }
    }

    for (int c = 0; c < numChannels; ++c) { /* [min, max] -> [0, 1], :762-787 */
        const float range = MAXF(maxLog2[c] - minLog2[c], 0.0f);
        if (range == 0.0f) {
            for (size_t k = 0; k < numPixels; ++k)
                gainMapF[c][k] = 0.0f;
        } else {
            const float gamma = uFrac(gainMap->gainMapGamma[c]);
            for (size_t k = 0; k < numPixels; ++k) {
                float v = gainMapF[c][k];
                v = CLAMPF(v, minLog2[c], maxLog2[c]);
                v = powf((v - minLog2[c]) / range, gamma);
                gainMapF[c][k] = nanSafeClamp(v);
            }
        }
    }

    const uint32_t requestedWidth = gmImage->width, requestedHeight = gmImage->height; /* to YUV, :789-823 */
    gmImage->width = width, gmImage->height = height;
    freePlanes(gmImage);
    gainMapRGB.width = width, gainMapRGB.height = height, gainMapRGB.depth = gmImage->depth, gainMapRGB.format = AVIF_RGB_FORMAT_RGBA;
    gainMapRGB.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, gainMapRGB.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    gainMapRGB.maxThreads = 1;
    res = allocatePixels(&gainMapRGB);
    if (res != AVIF_RESULT_OK)
        goto cleanup;
    PixelLayout gmL;
    if (!pixelLayout(&gainMapRGB, &gmL)) {
        res = AVIF_RESULT_NOT_IMPLEMENTED;
        goto cleanup;
    }
    for (uint32_t j = 0; j < height; ++j) {
        for (uint32_t i = 0; i < width; ++i) {
            const size_t k = (size_t)j * width + i;
            const float r = gainMapF[0][k], g = singleChannel ? r : gainMapF[1][k], b = singleChannel ? r : gainMapF[2][k];
            const float px[4] = { r, g, b, 1.0f };
            setPixel(&gainMapRGB, i, j, &gmL, px);
        }
    }
    res = libyuvBuild ? oracleLibyuvImageRGBToYUV(gmImage, &gainMapRGB) : oracleImageRGBToYUV(gmImage, &gainMapRGB);
    if (res != AVIF_RESULT_OK)
        goto cleanup;
    if (requestedWidth != gmImage->width || requestedHeight != gmImage->height)
        res = oracleImageScale(gmImage, requestedWidth, requestedHeight);

cleanup:
    for (int c = 0; c < 3; ++c)
        free(gainMapF[c]);
    free(gainMapRGB.pixels);
    if (res != AVIF_RESULT_OK)
        freePlanes(gmImage);
    return res;
}
