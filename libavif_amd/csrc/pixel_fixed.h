// pixel_fixed.h -- the reference's INTEGER path: the fixed-point arithmetic of the libyuv functions libavif dispatches
// to (src/reformat_libyuv.c), per SURVEY.md appendix D.  Integer-only, bit-exact; shared by the universal kernels
// (kernels_generic.hip) and the tiled integer kernels.  All intermediate values fit in 32-bit integers.
#pragma once

#include <hip/hip_runtime.h>

#include "exactdiv.h"
#include "pixel_math.h"

namespace avifhip {

__device__ __forceinline__ int fxClamp255(int v)
{
    return min(max(v, 0), 255);
}

// A plane sample as the selected libyuv entry sees it: Convert16To8Plane (src/reformat_libyuv.c:906-930) first when the
// entry is an 8-bit one fed from deeper planes.
__device__ __forceinline__ int fxReduceSample(unsigned v, int downshift)
{
    return downshift ? min((int)(v >> downshift), 255) : (int)v;
}

// Upsampled chroma at luma position (i, j) of a canvasW x canvasH image (appendix D.2: ScaleRowUp2_Linear /
// Scale2RowUp_Bilinear as libyuv's I4xxToARGBMatrixFilter applies them).  Taps are 3:1 per axis towards the side the
// luma sample leans to; the first and the LAST column are horizontally unfiltered (whatever the width's parity), the
// first row and the last row of an even height are vertically unfiltered; each stage rounds to an integer.  `load(x, y)`
// returns the plane's raw sample.  Samples of 16-bit containers are held to 12 bits here, like the packed filter of the
// tiled kernels (tile_fx_impl.h): a no-op for samples inside their nominal depth.
template <class Load>
__device__ __forceinline__ int fxChromaBilinear(const Load & load, const YuvToRgbPlan & p, bool vertical, uint32_t i, uint32_t j)
{
    const int downshift = p.fxDownshift;
    const bool wide = p.yuv.chanBytes == 2;
    auto sample = [&](uint32_t x, uint32_t y) {
        const int v = fxReduceSample(load(x, y), downshift);
        return (wide && downshift == 0) ? min(v, 4095) : v;
    };
    const uint32_t ci = i >> 1;
    const uint32_t cj = vertical ? (j >> 1) : j;
    const bool edgeX = (i == 0) || (i == p.canvasW - 1);
    const bool edgeY = !vertical || (j == 0) || (j == p.canvasH - 1 && !(p.canvasH & 1));
    // neighbours towards the side the luma sample leans to, held inside the job's chroma window
    const uint32_t fi = (uint32_t)clampInt((i & 1) ? (int)ci + 1 : (int)ci - 1, p.cwinX0, p.cwinX1);
    const uint32_t fj = (uint32_t)clampInt((j & 1) ? (int)cj + 1 : (int)cj - 1, p.cwinY0, p.cwinY1);
    const int a0 = sample(ci, cj);
    if (edgeX && edgeY)
        return a0;
    if (edgeY) {
        const int a1 = sample(fi, cj);
        return (3 * a0 + a1 + 2) >> 2;
    }
    const int b0 = sample(ci, fj);
    if (edgeX)
        return (3 * a0 + b0 + 2) >> 2;
    const int a1 = sample(fi, cj);
    const int b1 = sample(fi, fj);
    return (9 * a0 + 3 * a1 + 3 * b0 + b1 + 8) >> 4;
}

struct Rgb8
{
    int r, g, b;
};

// appendix D.1 (8-bit entries) and D.3 (I010 / I012 families): y, u, v are the samples the entry reads (u, v already
// upsampled at the plane's depth).
__device__ __forceinline__ Rgb8 fxMatrix(int y, int u, int v, const FixedPointMatrix & m, int native)
{
    uint32_t y32;
    if (native == 10) {
        y32 = (uint32_t)((y << 6) | (y >> 4));
        u = fxClamp255(u >> 2), v = fxClamp255(v >> 2);
    } else if (native == 12) {
        y32 = (uint32_t)((y << 4) | (y >> 8));
        u = fxClamp255(u >> 4), v = fxClamp255(v >> 4);
    } else {
        y32 = (uint32_t)y * 0x0101u;
    }
    const int y1 = (int)((y32 * (uint32_t)m.yg) >> 16) + m.yb;
    const int ub = u - 128, vb = v - 128;
    Rgb8 c;
    c.b = fxClamp255((y1 + m.ub * ub) >> 6);
    c.g = fxClamp255((y1 - (m.ug * ub + m.vg * vb)) >> 6);
    c.r = fxClamp255((y1 + m.vr * vb) >> 6);
    return c;
}

// ARGBAttenuate / ARGBUnattenuate on one channel (appendix D.4)
__device__ __forceinline__ unsigned fxAttenuate(unsigned c, unsigned a)
{
    return (c * a + 255u) >> 8;
}
// ARGBUnattenuate's 8.8 reciprocal of a pixel's alpha, formed once per pixel: floor(65536 / a) from v_rcp_f32 and one exact correction step
// instead of the 32-bit integer division sequence (exactdiv.h: quotient65536ByEstimate, enumerated for every a)
__device__ __forceinline__ unsigned fxUnattenuateReciprocal(unsigned a)
{
    const unsigned d = (a & 0xffu) ? (a & 0xffu) : 1u;
    const unsigned q = quotient65536ByEstimate(d, __builtin_amdgcn_rcpf((float)d));
    return (a == 0u) ? 0u : (a == 1u) ? 0xffffu : (a == 255u) ? 0x100u : q;
}
__device__ __forceinline__ unsigned fxUnattenuateBy(unsigned c, unsigned ia)
{
    const unsigned t = (((c & 0xffu) * 0x101u) * (ia & 0xffffu)) >> 16;
    return (t >= 0x8000u) ? 0u : min(t, 255u);
}
__device__ __forceinline__ unsigned fxUnattenuate(unsigned c, unsigned a)
{
    return fxUnattenuateBy(c, fxUnattenuateReciprocal(a));
}
__device__ __forceinline__ unsigned fxAlphaMul(unsigned c, unsigned a, int mulMode)
{
    return (mulMode == MUL_MULTIPLY) ? fxAttenuate(c, a) : fxUnattenuate(c, a);
}

// One output pixel of a fixed-point YUV -> 8-bit RGB conversion, with what libavif runs after the libyuv call fused in:
// the alpha channel (src/reformat.c:1464-1486) and the (un)premultiply post-pass (:1574-1585).
template <class Reader>
__device__ inline void yuvToRgbPixelFixedT(const YuvToRgbPlan & p, const Reader & rd, uint32_t i, uint32_t j)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    uint8_t * dst = rgbPixelAddress(o, i, j);
    if (!dst)
        return; // outside the fused crop

    const int y = fxReduceSample(rd.y(i, j), p.fxDownshift);
    int u = 128, v = 128;
    if (!p.fxMono) {
        if (s.format == AVIF_PIXEL_FORMAT_YUV444) {
            u = fxReduceSample(rd.u(i, j), p.fxDownshift);
            v = fxReduceSample(rd.v(i, j), p.fxDownshift);
        } else if (!p.bilinear) {
            const uint32_t cj = j >> s.shiftY;
            u = fxReduceSample(rd.u(i >> 1, cj), p.fxDownshift);
            v = fxReduceSample(rd.v(i >> 1, cj), p.fxDownshift);
        } else {
            const bool vertical = s.format == AVIF_PIXEL_FORMAT_YUV420;
            u = fxChromaBilinear([&](uint32_t x, uint32_t yy) { return rd.u(x, yy); }, p, vertical, i, j);
            v = fxChromaBilinear([&](uint32_t x, uint32_t yy) { return rd.v(x, yy); }, p, vertical, i, j);
        }
    } else if (p.fxNative != 8) {
        u = v = 128 << (p.fxNative - 8); // unreachable today (no mono entry above 8 bits); keeps fxMatrix's reduction neutral
    }
    const Rgb8 c = fxMatrix(y, u, v, p.fx, p.fxNative);
    unsigned r = (unsigned)c.r, g = (unsigned)c.g, b = (unsigned)c.b;

    unsigned a = 255;
    if (o.hasAlpha && p.alphaSource == ALPHA_PLANE) {
        const unsigned sa = rd.a(i, j);
        if (p.fxAlpha == FXA_SHIFT)
            a = min(sa >> p.fxAlphaShift, 255u);
        else
            a = (s.depth == o.depth) ? sa : rescaleAlpha(sa, (float)s.maxv, o.maxf, o.maxv);
    }
    if (p.postMul != MUL_NONE) {
        if (p.postMulFx) {
            r = fxAlphaMul(r, a, p.postMul), g = fxAlphaMul(g, a, p.postMul), b = fxAlphaMul(b, a, p.postMul);
        } else {
            r = alphaMulInt(r, a, 255u, 255.0f, p.postMul), g = alphaMulInt(g, a, 255u, 255.0f, p.postMul), b = alphaMulInt(b, a, 255u, 255.0f, p.postMul);
        }
    }
    if (o.is565) {
        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)pack565(r, g, b);
        return;
    }
    dst[o.offR] = (uint8_t)r;
    dst[o.offG] = (uint8_t)g;
    dst[o.offB] = (uint8_t)b;
    if (o.hasAlpha)
        dst[o.offA] = (uint8_t)a;
}

__device__ inline void yuvToRgbPixelFixed(const YuvToRgbPlan & p, uint32_t i, uint32_t j)
{
    yuvToRgbPixelFixedT(p, PlanReader { p.yuv }, i, j);
}

// ---- RGB -> YUV, 8-bit BT.601 (appendix D.5) ----------------------------------------------------------------

__device__ __forceinline__ Rgb8 fxLoadRgb(const RgbSide & o, uint32_t x, uint32_t y)
{
    const uint8_t * px = o.pixels + (size_t)y * o.rowBytes + (size_t)x * o.pixBytes;
    Rgb8 c;
    c.r = px[o.offR], c.g = px[o.offG], c.b = px[o.offB];
    return c;
}
__device__ __forceinline__ int fxLuma(Rgb8 c, bool full)
{
    return full ? ((77 * c.r + 150 * c.g + 29 * c.b + 128) >> 8) : ((66 * c.r + 129 * c.g + 25 * c.b + 0x1080) >> 8);
}
__device__ __forceinline__ int fxCb(Rgb8 c, bool full)
{
    return full ? ((128 * c.b - 85 * c.g - 43 * c.r + 0x8000) >> 8) : ((112 * c.b - 74 * c.g - 38 * c.r + 0x8000) >> 8);
}
__device__ __forceinline__ int fxCr(Rgb8 c, bool full)
{
    return full ? ((128 * c.r - 107 * c.g - 21 * c.b + 0x8000) >> 8) : ((112 * c.r - 94 * c.g - 18 * c.b + 0x8000) >> 8);
}

} // namespace avifhip
