// pixel_math.h -- device-side fp32 arithmetic shared by all reformat kernels.
//
// Every function performs the reference's operations in the reference's order with separate
// (un-fused) fp32 multiplies and adds and IEEE-754 division; the translation unit is compiled
// with -ffp-contract=off and correctly-rounded division.  The reference reads normalised samples
// from per-call look-up tables (src/reformat.c:575-603); the table entry is a pure function of
// the code point, so it is recomputed here with the same expression instead of being fetched.
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace avifhip {

__device__ __forceinline__ float clamp01(float x) // AVIF_CLAMP(x, 0.0f, 1.0f), include/avif/internal.h:18
{
    return (x < 0.0f) ? 0.0f : ((1.0f < x) ? 1.0f : x);
}
__device__ __forceinline__ int clampInt(int x, int lo, int hi)
{
    return (x < lo) ? lo : ((hi < x) ? hi : x);
}
__device__ __forceinline__ float roundHalfUp(float v) // avifRoundf, src/utils.c:11-14
{
    return floorf(v + 0.5f);
}

// unormFloatTableY[cp] / unormFloatTableUV[cp], src/reformat.c:583,598
__device__ __forceinline__ float normY(unsigned cp, const YuvSide & s)
{
    return ((float)cp - s.biasY) / s.rangeY;
}
__device__ __forceinline__ float normUV(unsigned cp, const YuvSide & s)
{
    // identity mode reuses the luma table for chroma, src/reformat.c:587-589
    return (s.mode == MODE_IDENTITY) ? (((float)cp - s.biasY) / s.rangeY) : (((float)cp - s.biasUV) / s.rangeUV);
}

// x / d for a plan constant d on the verified list (exactdiv.h): bit-identical to the IEEE quotient
__device__ __forceinline__ float divExact(float x, RcpHL r)
{
    return __builtin_fmaf(x, r.hi, x * r.lo);
}

__device__ __forceinline__ unsigned loadSample(const uint8_t * plane, uint32_t rowBytes, uint32_t x, uint32_t y, int chanBytes)
{
    const uint8_t * p = plane + (size_t)y * rowBytes + (size_t)x * chanBytes;
    return (chanBytes == 1) ? (unsigned)*p : (unsigned)*reinterpret_cast<const uint16_t *>(p);
}
// "clamp incoming data to protect against bad LUT lookups", src/reformat.c:712,727,821
__device__ __forceinline__ unsigned loadSampleClamped(const uint8_t * plane, uint32_t rowBytes, uint32_t x, uint32_t y, int chanBytes, unsigned maxv)
{
    const unsigned v = loadSample(plane, rowBytes, x, y, chanBytes);
    return (v < maxv) ? v : maxv;
}

// avifLimitedToFullY (src/reformat.c:1778-1791) on one alpha sample: ((v - lo) * full + (hi - lo) / 2) / (hi - lo), clamped
__device__ __forceinline__ unsigned limitedToFullAlpha(unsigned v, int depth)
{
    const int lo = 16 << (depth - 8), range = 219 << (depth - 8), full = (1 << depth) - 1;
    const int n = ((int)v - lo) * full + range / 2;
    const int q = n / range; // C division truncates toward zero, like the reference
    return (unsigned)clampInt(q, 0, full);
}

// Where a pixel routine gets its samples from.  PlanReader: the planes of the plan (one contiguous image).
struct PlanReader
{
    const YuvSide & s;
    __device__ __forceinline__ unsigned y(uint32_t x, uint32_t yy) const { return loadSample(s.plane[0], s.rowBytes[0], x, yy, s.chanBytes); }
    __device__ __forceinline__ unsigned u(uint32_t x, uint32_t yy) const { return loadSample(s.plane[1], s.rowBytes[1], x, yy, s.chanBytes); }
    __device__ __forceinline__ unsigned v(uint32_t x, uint32_t yy) const { return loadSample(s.plane[2], s.rowBytes[2], x, yy, s.chanBytes); }
    __device__ __forceinline__ unsigned a(uint32_t x, uint32_t yy) const
    {
        const unsigned sa = loadSample(s.alpha, s.alphaRowBytes, x, yy, s.chanBytes);
        return s.alphaLimited ? limitedToFullAlpha(sa, (int)s.depth) : sa;
    }
};

// 4-tap chroma filter on normalised samples, src/reformat.c:834-837: four products, three adds, left to right.
__device__ __forceinline__ float bilinear4(float closest, float horiz, float vert, float diag)
{
    return (closest * (9.0f / 16.0f)) + (horiz * (3.0f / 16.0f)) + (vert * (3.0f / 16.0f)) + (diag * (1.0f / 16.0f));
}

struct Rgbf
{
    float r, g, b;
};

// Y,Cb,Cr (normalised) -> R,G,B (unclamped), src/reformat.c:845-884.  unormY feeds the YCgCo-R lifting.
__device__ __forceinline__ Rgbf yuvToRgbCore(float Y, float Cb, float Cr, unsigned unormY, const YuvSide & s, const RgbSide & o)
{
    Rgbf c;
    if (!s.hasColor) {
        c.r = c.g = c.b = Y;
    } else if (s.mode == MODE_COEFF) {
        c.r = Y + s.twoOneMinusKr * Cr;
        c.b = Y + s.twoOneMinusKb * Cb;
        c.g = Y - ((2 * ((s.krOneMinusKr * Cr) + (s.kbOneMinusKb * Cb))) / s.kg);
    } else if (s.mode == MODE_IDENTITY) {
        c.g = Y, c.b = Cb, c.r = Cr;
    } else if (s.mode == MODE_YCGCO) {
        const float t = Y - Cb;
        c.g = Y + Cb;
        c.b = t - Cr;
        c.r = t + Cr;
    } else { // YCgCo-Re / YCgCo-Ro, :859-871
        const int Cg = (int)roundHalfUp(Cb * (float)s.maxv);
        const int Co = (int)roundHalfUp(Cr * (float)s.maxv);
        const int t = (int)unormY - (Cg >> 1);
        const int gi = clampInt(t + Cg, 0, o.maxv);
        const int bi = clampInt(t - (Co >> 1), 0, o.maxv);
        const int ri = clampInt(bi + Co, 0, o.maxv);
        c.g = (float)gi / o.maxf;
        c.b = (float)bi / o.maxf;
        c.r = (float)ri / o.maxf;
    }
    return c;
}

// in-loop alpha (un)multiply on a clamped channel, src/reformat.c:905-946
__device__ __forceinline__ float applyAlphaF(float c, float Ac, int mulMode)
{
    if (Ac == 0.0f)
        return 0.0f;
    if (Ac < 1.0f) {
        if (mulMode == MUL_MULTIPLY)
            return c * Ac;
        const float q = c / Ac;
        return (q < 1.0f) ? q : 1.0f;
    }
    return c;
}

// quantise a clamped channel: (T)(0.5f + (c * max)), src/reformat.c:952-961
__device__ __forceinline__ unsigned quantize(float c, float maxf)
{
    return (unsigned)(0.5f + (c * maxf));
}

// (un)premultiply on stored integers, src/alpha.c:180-192 and :367-381 (caller handles a>=max / a==0)
__device__ __forceinline__ unsigned premultiplyInt(unsigned c, unsigned a, float maxf)
{
    return (unsigned)roundHalfUp((float)c * (float)a / maxf);
}
__device__ __forceinline__ unsigned unpremultiplyInt(unsigned c, unsigned a, float maxf)
{
    const float q = roundHalfUp((float)c * maxf / (float)a);
    return (unsigned)((q < maxf) ? q : maxf);
}
__device__ __forceinline__ unsigned alphaMulInt(unsigned c, unsigned a, unsigned maxv, float maxf, int mulMode)
{
    if (a >= maxv)
        return c;
    if (a == 0)
        return 0;
    return (mulMode == MUL_MULTIPLY) ? premultiplyInt(c, a, maxf) : unpremultiplyInt(c, a, maxf);
}

// Un-premultiply on stored integers WITHOUT the IEEE division sequence (tiled kernels and the 16-byte in-place pass).  The reference's
// min(floorf((float)c * maxF / (float)a + 0.5f), maxF) (src/alpha.c:367-381) divides the three colours of a pixel by the same alpha, so the
// reciprocal is formed once: v_rcp_f32 (1 ulp) and one Newton step give r = RN(1 / a) for every alpha code, and then
//      q = fma(fma(-q0, a, x), r, q0),  q0 = x * r,  x = RN(c * maxF)
// IS the correctly rounded x / a (Markstein's correction step) -- enumerated for every 16-bit code c and every 0 < a < max, max in
// {255, 1023, 4095, 65535}, reciprocal estimates off by up to 2 ulp (tests/tools/verify_fp32_shortcuts.cpp).  Five full-rate fp32
// instructions per channel: on gfx950 fp32 add / mul / fma issue in ~2.8 cycles per wave, conversions and integer operations in ~4.3
// (profiles/r03_valu_rate.txt), which is why this beats the integer form with its remainder test.
struct UnpremulRcp
{
    float af, r;
};
__device__ __forceinline__ UnpremulRcp unpremulRcp(float af) // af: the alpha code as a float, not 0
{
    const float r0 = __builtin_amdgcn_rcpf(af);
    return { af, __builtin_fmaf(__builtin_fmaf(-af, r0, 1.0f), r0, r0) };
}
// ... the argument of the final truncation: x / a + 0.5f (the caller truncates and holds the result to max)
__device__ __forceinline__ float unpremulRcpArg(float cf, const UnpremulRcp & R, float maxf)
{
    const float x = cf * maxf;
    const float q0 = x * R.r;
    return __builtin_fmaf(__builtin_fmaf(-q0, R.af, x), R.r, q0) + 0.5f;
}

// alpha depth rescale, src/alpha.c:93-96
__device__ __forceinline__ unsigned rescaleAlpha(unsigned srcAlpha, float srcMaxF, float dstMaxF, int dstMax)
{
    const float alphaF = (float)srcAlpha / srcMaxF;
    const int dstAlpha = (int)(0.5f + (alphaF * dstMaxF));
    return (unsigned)clampInt(dstAlpha, 0, dstMax);
}

// integer -> IEEE half via the subnormal-multiply trick, src/reformat.c:1411-1438
__device__ __forceinline__ unsigned toHalfBits(unsigned v, float multiplier)
{
    const float f = (float)v * multiplier;
    return __float_as_uint(f) >> 13;
}

// RGB565 packing, src/reformat.c:619
__device__ __forceinline__ unsigned pack565(unsigned r, unsigned g, unsigned b)
{
    return ((b & 0xff) >> 3) | (((g & 0xff) >> 2) << 5) | (((r & 0xff) >> 3) << 11);
}

// Y/U/V quantisation for RGB->YUV, src/reformat.c:197-219
__device__ __forceinline__ int quantizeY(float v, const YuvSide & s)
{
    return clampInt((int)roundHalfUp(v * s.rangeY + s.biasY), 0, s.maxv);
}
__device__ __forceinline__ int quantizeUV(float v, const YuvSide & s)
{
    const int q = (s.mode == MODE_IDENTITY) ? (int)roundHalfUp(v * s.rangeY + s.biasY) : (int)roundHalfUp(v * s.rangeUV + s.biasUV);
    return clampInt(q, 0, s.maxv);
}

// address of canvas pixel (i, j) in the destination buffer, nullptr when the fused crop drops it (plan.h PixelMap)
__device__ __forceinline__ uint8_t * rgbPixelAddress(const RgbSide & o, uint32_t i, uint32_t j)
{
    if (!o.map.on)
        return o.pixels + (size_t)j * o.rowBytes + (size_t)i * o.pixBytes;
    const uint32_t ii = i - o.map.cx, jj = j - o.map.cy;
    if (ii >= o.map.cw || jj >= o.map.ch)
        return nullptr;
    const int32_t a = (int32_t)(o.map.transposed ? jj : ii), b = (int32_t)(o.map.transposed ? ii : jj);
    const uint32_t x = (uint32_t)(o.map.sx * a + o.map.kx), y = (uint32_t)(o.map.sy * b + o.map.ky);
    return o.pixels + (size_t)y * o.rowBytes + (size_t)x * o.pixBytes;
}

} // namespace avifhip
