// pixel_generic.h -- the universal per-pixel YUV->RGB device routine, shared by the generic kernel
// (kernels_generic.hip) and by the partial-tile path of the tiled kernels (kernels_tile.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "pixel_fixed.h"
#include "pixel_math.h"

namespace avifhip {

// --------------------------------------------------------------------------------------------
// YUV -> RGB, one lane per pixel.  Fuses what the reference runs as separate passes:
// alpha fill/copy/rescale (src/alpha.c:9-149), colour conversion with chroma upsampling and
// in-loop alpha multiply (src/reformat.c:650-978) or the specialised loops (:980-1407) followed by
// integer (un)premultiply (src/alpha.c:151-535), and the half-float pass (src/reformat.c:1419-1443).
template <class Reader>
__device__ inline void yuvToRgbPixelT(const YuvToRgbPlan & p, const Reader & rd, uint32_t i, uint32_t j)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    uint8_t * dst = rgbPixelAddress(o, i, j);
    if (!dst)
        return; // outside the fused crop

    unsigned r = 0, g = 0, b = 0, gray = 0;

    if (p.identityCopy) { // src/reformat.c:1278-1309
        g = rd.y(i, j);
        b = rd.u(i, j);
        r = rd.v(i, j);
    } else {
        const unsigned mxv = (unsigned)s.maxv; // "clamp incoming data to protect against bad LUT lookups", src/reformat.c:712,727,821
        const unsigned unormY = min(rd.y(i, j), mxv);
        const float Y = normY(unormY, s);
        float Cb = 0.5f, Cr = 0.5f;
        if (s.hasColor) {
            const uint32_t uvI = i >> s.shiftX;
            const uint32_t uvJ = j >> s.shiftY;
            if (s.format == AVIF_PIXEL_FORMAT_YUV444 || !p.bilinear) {
                Cb = normUV(min(rd.u(uvI, uvJ), mxv), s);
                Cr = normUV(min(rd.v(uvI, uvJ), mxv), s);
            } else {
                // neighbour selection against the CANVAS borders, src/reformat.c:766-795
                int dx, dy;
                if (i == 0 || (i == p.canvasW - 1 && (i & 1)))
                    dx = 0;
                else
                    dx = (i & 1) ? 1 : -1;
                if (j == 0 || (j == p.canvasH - 1 && (j & 1)) || s.format == AVIF_PIXEL_FORMAT_YUV422)
                    dy = 0;
                else
                    dy = (j & 1) ? 1 : -1;
                // ... and against the job's chroma window (the whole plane unless the canvas is a grid of separate tiles)
                const uint32_t xn = (uint32_t)clampInt((int)uvI + dx, p.cwinX0, p.cwinX1), yn = (uint32_t)clampInt((int)uvJ + dy, p.cwinY0, p.cwinY1);
                const float u00 = normUV(min(rd.u(uvI, uvJ), mxv), s);
                const float u10 = normUV(min(rd.u(xn, uvJ), mxv), s);
                const float u01 = normUV(min(rd.u(uvI, yn), mxv), s);
                const float u11 = normUV(min(rd.u(xn, yn), mxv), s);
                const float v00 = normUV(min(rd.v(uvI, uvJ), mxv), s);
                const float v10 = normUV(min(rd.v(xn, uvJ), mxv), s);
                const float v01 = normUV(min(rd.v(uvI, yn), mxv), s);
                const float v11 = normUV(min(rd.v(xn, yn), mxv), s);
                Cb = bilinear4(u00, u10, u01, u11);
                Cr = bilinear4(v00, v10, v01, v11);
            }
        }

        float Rc = 0.0f, Gc = 0.0f, Bc = 0.0f, grayc = 0.0f;
        if (!o.isGray) {
            const Rgbf c = yuvToRgbCore(Y, Cb, Cr, unormY, s, o);
            Rc = clamp01(c.r);
            Gc = clamp01(c.g);
            Bc = clamp01(c.b);
        } else {
            grayc = clamp01(Y);
        }
        if (p.inLoopMul != MUL_NONE) { // src/reformat.c:894-947
            const unsigned unormA = min(rd.a(i, j), mxv);
            const float Ac = clamp01((float)unormA / ((float)s.maxv));
            Rc = applyAlphaF(Rc, Ac, p.inLoopMul);
            Gc = applyAlphaF(Gc, Ac, p.inLoopMul);
            Bc = applyAlphaF(Bc, Ac, p.inLoopMul);
            grayc = applyAlphaF(grayc, Ac, p.inLoopMul);
        }
        r = quantize(Rc, o.maxf);
        g = quantize(Gc, o.maxf);
        b = quantize(Bc, o.maxf);
        gray = quantize(grayc, o.maxf);
    }

    // alpha channel value at the destination depth
    unsigned a = 0;
    bool writeAlpha = false;
    if (o.hasAlpha) {
        if (p.alphaSource == ALPHA_FILL) {
            a = (unsigned)o.maxv;
            writeAlpha = true;
        } else if (p.alphaSource == ALPHA_PLANE) {
            const unsigned sa = rd.a(i, j);
            a = (s.depth == o.depth) ? sa : rescaleAlpha(sa, (float)s.maxv, o.maxf, o.maxv);
            writeAlpha = true;
        } else if (o.isFloat) {
            // alpha bytes are not ours to define, but the half-float pass still runs over them
            a = (o.chanBytes == 1) ? dst[o.offA] : *reinterpret_cast<const uint16_t *>(dst + o.offA);
            writeAlpha = true;
        }
    }

    if (p.postMul != MUL_NONE) { // src/reformat.c:1574-1585 on the stored integers
        if (p.postMulFx) { // a libyuv build attenuates 8-bit RGBA / BGRA with libyuv, src/alpha.c:163,350
            r = fxAlphaMul(r, a, p.postMul), g = fxAlphaMul(g, a, p.postMul), b = fxAlphaMul(b, a, p.postMul);
        } else {
            r = alphaMulInt(r, a, (unsigned)o.maxv, o.maxf, p.postMul);
            g = alphaMulInt(g, a, (unsigned)o.maxv, o.maxf, p.postMul);
            b = alphaMulInt(b, a, (unsigned)o.maxv, o.maxf, p.postMul);
        }
    }

    if (o.isFloat) { // depth 16 only
        r = toHalfBits(r, o.f16Multiplier);
        g = toHalfBits(g, o.f16Multiplier);
        b = toHalfBits(b, o.f16Multiplier);
        gray = toHalfBits(gray, o.f16Multiplier);
        a = toHalfBits(a, o.f16Multiplier);
    }

    if (o.is565) {
        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)pack565(r, g, b);
        return;
    }
    if (o.chanBytes == 1) {
        if (o.isGray) {
            dst[o.offGray] = (uint8_t)gray;
        } else {
            dst[o.offR] = (uint8_t)r;
            dst[o.offG] = (uint8_t)g;
            dst[o.offB] = (uint8_t)b;
        }
        if (writeAlpha)
            dst[o.offA] = (uint8_t)a;
    } else {
        if (o.isGray) {
            *reinterpret_cast<uint16_t *>(dst + o.offGray) = (uint16_t)gray;
        } else {
            *reinterpret_cast<uint16_t *>(dst + o.offR) = (uint16_t)r;
            *reinterpret_cast<uint16_t *>(dst + o.offG) = (uint16_t)g;
            *reinterpret_cast<uint16_t *>(dst + o.offB) = (uint16_t)b;
        }
        if (writeAlpha)
            *reinterpret_cast<uint16_t *>(dst + o.offA) = (uint16_t)a;
    }
}

__device__ inline void yuvToRgbPixel(const YuvToRgbPlan & p, uint32_t i, uint32_t j)
{
    yuvToRgbPixelT(p, PlanReader { p.yuv }, i, j);
}

} // namespace avifhip
