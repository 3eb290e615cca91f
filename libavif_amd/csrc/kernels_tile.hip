// kernels_tile.hip -- bandwidth-tuned YUV->RGB kernels for gfx950 (MI355X).
//
// Scope: matrix-coefficient ("normal YUV") conversions into interleaved 3- or 4-channel RGB at 8-bit
// or 16-bit containers, from 8-bit or 16-bit-container 4:4:4 / 4:2:2 / 4:2:0 / 4:0:0 planes, with
// nearest or bilinear chroma upsampling, alpha fill / copy / rescale and both flavours of alpha
// (un)premultiply -- i.e. every BASELINE configuration and what avifdec asks for.  Anything else
// (gray outputs, RGB565, half float, identity / YCgCo matrices, unaligned user buffers) is served by
// kernels_generic.hip, whose per-pixel routine is also the edge-tile fallback here.
//
// Work decomposition (wave = 64 lanes, block = 4 waves):
//   block tile  = 256 x 8 luma samples; lane (tx,ty) owns 4 consecutive pixels of rows 2ty, 2ty+1;
//   luma/alpha  : one 4-sample vector load per lane and row (row-coalesced, 256 B or 512 B per wave);
//   chroma      : the tile's chroma neighbourhood (4:2:0: 6 rows x 130 samples per plane) is loaded
//                 once with 4-sample vector loads, normalised to fp32 once per sample and staged
//                 in LDS; every lane then reads its 4x3 neighbourhood from LDS (the reference
//                 re-reads and re-normalises up to 4 chroma samples per output pixel);
//   stores      : 4 pixels per lane and row as one 16-byte store (RGBA8: fully coalesced 1 KiB per
//                 wave instruction), 12 bytes for RGB8, 2 x 16 bytes for 16-bit RGBA.
// The arithmetic is that of pixel_math.h (reference order, no contraction, IEEE division).
#include <hip/hip_runtime.h>

#include <stdio.h>

#include "kernels.h"
#include "pixel_generic.h"
#include "pixel_math.h"

namespace avifhip {

namespace {

constexpr int kTileW = 256;
constexpr int kTileH = 8;
constexpr int kLanesX = 64; // 4 pixels each
constexpr int kLanesY = 4;  // 2 rows each
constexpr int kChromaPitch = 132; // 2 (left pad incl. halo) + 128 + 2 floats: interior starts 8-byte aligned
constexpr int kChromaRowsMax = 8;

enum Subsampling : int { SUB_444 = 0, SUB_422 = 1, SUB_420 = 2, SUB_400 = 3 };

template <typename T>
__device__ __forceinline__ void load4(const uint8_t * p, unsigned v[4]);
template <>
__device__ __forceinline__ void load4<uint8_t>(const uint8_t * p, unsigned v[4])
{
    const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
    v[0] = w & 0xffu;
    v[1] = (w >> 8) & 0xffu;
    v[2] = (w >> 16) & 0xffu;
    v[3] = w >> 24;
}
template <>
__device__ __forceinline__ void load4<uint16_t>(const uint8_t * p, unsigned v[4])
{
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    v[0] = w.x & 0xffffu;
    v[1] = w.x >> 16;
    v[2] = w.y & 0xffffu;
    v[3] = w.y >> 16;
}

template <typename T>
__device__ __forceinline__ unsigned load1(const uint8_t * plane, uint32_t rowBytes, uint32_t x, uint32_t y)
{
    return (unsigned)*reinterpret_cast<const T *>(plane + (size_t)y * rowBytes + (size_t)x * sizeof(T));
}

__device__ __forceinline__ unsigned minU(unsigned a, unsigned b)
{
    return a < b ? a : b;
}

// One pixel from fully prepared inputs: matrix, clamp, optional fp32 alpha multiply, quantise,
// optional integer alpha multiply.  a = alpha at the RGB depth (valid when the plan has one).
struct PixelOut
{
    unsigned r, g, b;
};

// Y,Cb,Cr -> unclamped R,G,B, src/reformat.c:874-876.  kFast: the plan's divisors are on the verified list
// (reciprocal form, exactdiv.h).
template <bool kFast>
__device__ __forceinline__ void matrixRgb(const YuvSide & s, float Y, float Cb, float Cr, float & R, float & G, float & B)
{
    if (s.hasColor) {
        R = Y + s.twoOneMinusKr * Cr;
        B = Y + s.twoOneMinusKb * Cb;
        const float num = 2 * ((s.krOneMinusKr * Cr) + (s.kbOneMinusKb * Cb));
        G = Y - (kFast ? divByVerifiedConstant(num, s.kg, s.rcpKg) : (num / s.kg));
    } else {
        R = G = B = Y;
    }
}

// (uint8_t)(0.5f + clamp01(c) * 255) packed into byte `slot` of `word`.  v_cvt_pk_u8_f32 rounds to nearest even
// and saturates to [0, 255] (probed on gfx950, tests/tools/probe_cvt.hip); fed with floor(0.5f + c * 255) it is
// exact, and because 0.5f + c * 255 is monotonic in c the saturation selects the same byte as clamping c first.
__device__ __forceinline__ unsigned packByte(float c, float maxf, unsigned slot, unsigned word)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(floorf(0.5f + (c * maxf)), slot, word);
}

template <bool kHasMul, bool kFast>
__device__ __forceinline__ PixelOut finishPixel(const YuvToRgbPlan & p, float Y, float Cb, float Cr, unsigned unormA, unsigned a)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    float R, G, B;
    matrixRgb<kFast>(s, Y, Cb, Cr, R, G, B);
    float Rc = clamp01(R), Gc = clamp01(G), Bc = clamp01(B);
    if (kHasMul && p.inLoopMul != MUL_NONE) {
        const float Ac = clamp01((float)minU(unormA, (unsigned)s.maxv) / ((float)s.maxv));
        Rc = applyAlphaF(Rc, Ac, p.inLoopMul);
        Gc = applyAlphaF(Gc, Ac, p.inLoopMul);
        Bc = applyAlphaF(Bc, Ac, p.inLoopMul);
    }
    PixelOut q;
    q.r = quantize(Rc, o.maxf);
    q.g = quantize(Gc, o.maxf);
    q.b = quantize(Bc, o.maxf);
    if (kHasMul && p.postMul != MUL_NONE) {
        q.r = alphaMulInt(q.r, a, (unsigned)o.maxv, o.maxf, p.postMul);
        q.g = alphaMulInt(q.g, a, (unsigned)o.maxv, o.maxf, p.postMul);
        q.b = alphaMulInt(q.b, a, (unsigned)o.maxv, o.maxf, p.postMul);
    }
    return q;
}

// Store 4 consecutive pixels.  swapRB: B is the first colour channel; alphaFirst: A precedes colour.
template <typename RT, int NCH>
__device__ __forceinline__ void store4(uint8_t * dst, const PixelOut q[4], const unsigned a[4], bool swapRB, bool alphaFirst)
{
    unsigned x[4], z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = swapRB ? q[k].b : q[k].r;
        z[k] = swapRB ? q[k].r : q[k].b;
    }
    if constexpr (sizeof(RT) == 1 && NCH == 4) {
        uint4 w;
        unsigned * wv = reinterpret_cast<unsigned *>(&w);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            wv[k] = alphaFirst ? (a[k] | (x[k] << 8) | (q[k].g << 16) | (z[k] << 24)) : (x[k] | (q[k].g << 8) | (z[k] << 16) | (a[k] << 24));
        *reinterpret_cast<uint4 *>(dst) = w;
    } else if constexpr (sizeof(RT) == 1 && NCH == 3) {
        // 12 bytes: x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3
        struct __attribute__((packed, aligned(4))) Rgb12
        {
            unsigned w0, w1, w2;
        } w;
        w.w0 = x[0] | (q[0].g << 8) | (z[0] << 16) | (x[1] << 24);
        w.w1 = q[1].g | (z[1] << 8) | (x[2] << 16) | (q[2].g << 24);
        w.w2 = z[2] | (x[3] << 8) | (q[3].g << 16) | (z[3] << 24);
        *reinterpret_cast<Rgb12 *>(dst) = w;
    } else if constexpr (sizeof(RT) == 2 && NCH == 4) {
        uint4 w0, w1;
        unsigned * v0 = reinterpret_cast<unsigned *>(&w0);
        unsigned * v1 = reinterpret_cast<unsigned *>(&w1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = alphaFirst ? (a[k] | (x[k] << 16)) : (x[k] | (q[k].g << 16));
            const unsigned hi = alphaFirst ? (q[k].g | (z[k] << 16)) : (z[k] | (a[k] << 16));
            unsigned * v = (k < 2) ? v0 : v1;
            v[(k & 1) * 2 + 0] = lo;
            v[(k & 1) * 2 + 1] = hi;
        }
        reinterpret_cast<uint4 *>(dst)[0] = w0;
        reinterpret_cast<uint4 *>(dst)[1] = w1;
    } else { // 16-bit, 3 channels: 24 bytes = 3 x 8
        uint2 w0, w1, w2;
        w0.x = x[0] | (q[0].g << 16);
        w0.y = z[0] | (x[1] << 16);
        w1.x = q[1].g | (z[1] << 16);
        w1.y = x[2] | (q[2].g << 16);
        w2.x = z[2] | (x[3] << 16);
        w2.y = q[3].g | (z[3] << 16);
        reinterpret_cast<uint2 *>(dst)[0] = w0;
        reinterpret_cast<uint2 *>(dst)[1] = w1;
        reinterpret_cast<uint2 *>(dst)[2] = w2;
    }
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL, bool FAST>
__device__ __forceinline__ void tileBody(const YuvToRgbPlan & p, float (*sU)[kChromaPitch], float (*sV)[kChromaPitch])
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const uint32_t tileX = blockIdx.x * kTileW; // relative to the rectangle
    const uint32_t tileY = blockIdx.y * kTileH;
    if (tileX >= p.w || tileY >= p.h)
        return; // batch launches are sized for the largest job

    // Partial tiles (right / bottom border of odd-sized jobs) take the per-pixel routine.
    if (tileX + kTileW > p.w || tileY + kTileH > p.h) {
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const uint32_t j = tileY + 2 * ty + r;
            if (j >= p.h)
                continue;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = tileX + 4 * tx + k;
                if (i < p.w)
                    yuvToRgbPixel(p, p.x0 + i, p.y0 + j);
            }
        }
        return;
    }

    const uint32_t X = p.x0 + tileX + 4 * tx; // canvas coordinates of this lane's first pixel
    const uint32_t Y0 = p.y0 + tileY + 2 * ty;
    const unsigned yuvMax = (unsigned)s.maxv;
    constexpr bool kWide = sizeof(YT) == 2;

    // ---- issue the streaming loads first: luma (and alpha) for both rows ----
    // alpha samples feed the A channel (copy / rescale) and the fp32 in-loop multiply (3-channel outputs included)
    const bool needAlpha = (NCH == 4 && p.alphaSource == ALPHA_PLANE) || (HASMUL && p.inLoopMul != MUL_NONE);
    unsigned yv[2][4], av[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        load4<YT>(s.plane[0] + (size_t)(Y0 + r) * s.rowBytes[0] + (size_t)X * sizeof(YT), yv[r]);
        if (needAlpha)
            load4<YT>(s.alpha + (size_t)(Y0 + r) * s.alphaRowBytes + (size_t)X * sizeof(YT), av[r]);
    }

    // ---- chroma ----
    float cb[2][4], cr[2][4];
    if constexpr (SUB == SUB_400) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                cb[r][k] = cr[r][k] = 0.5f;
    } else if constexpr (SUB == SUB_444) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            unsigned uq[4], vq[4];
            load4<YT>(s.plane[1] + (size_t)(Y0 + r) * s.rowBytes[1] + (size_t)X * sizeof(YT), uq);
            load4<YT>(s.plane[2] + (size_t)(Y0 + r) * s.rowBytes[2] + (size_t)X * sizeof(YT), vq);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cb[r][k] = normUVT<FAST>(kWide ? minU(uq[k], yuvMax) : uq[k], s);
                cr[r][k] = normUVT<FAST>(kWide ? minU(vq[k], yuvMax) : vq[k], s);
            }
        }
    } else if constexpr (!BILINEAR) {
        // nearest: chroma sample (i>>1, j>>shiftY), two per lane and chroma row
        const uint32_t cx = X >> 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (SUB == SUB_420 && r == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cb[1][k] = cb[0][k];
                    cr[1][k] = cr[0][k];
                }
                break;
            }
            const uint32_t cy = (SUB == SUB_420) ? (Y0 >> 1) : (Y0 + r);
            unsigned u0 = load1<YT>(s.plane[1], s.rowBytes[1], cx, cy), u1 = load1<YT>(s.plane[1], s.rowBytes[1], cx + 1, cy);
            unsigned v0 = load1<YT>(s.plane[2], s.rowBytes[2], cx, cy), v1 = load1<YT>(s.plane[2], s.rowBytes[2], cx + 1, cy);
            if (kWide) {
                u0 = minU(u0, yuvMax), u1 = minU(u1, yuvMax), v0 = minU(v0, yuvMax), v1 = minU(v1, yuvMax);
            }
            const float fu0 = normUVT<FAST>(u0, s), fu1 = normUVT<FAST>(u1, s), fv0 = normUVT<FAST>(v0, s), fv1 = normUVT<FAST>(v1, s);
            cb[r][0] = cb[r][1] = fu0;
            cb[r][2] = cb[r][3] = fu1;
            cr[r][0] = cr[r][1] = fv0;
            cr[r][2] = cr[r][3] = fv1;
        }
    } else {
        // bilinear: stage the normalised chroma neighbourhood of the tile in LDS.
        // LDS row q holds canvas chroma row clamp(cy0 - 1 + q) for 4:2:0 (6 rows) or cy0 + q for 4:2:2 (8 rows);
        // LDS column c+2 holds canvas chroma column cx0 + c, c in [-1, 128].
        const uint32_t cw = (p.canvasW + 1) >> 1;
        const uint32_t ch = (SUB == SUB_420) ? ((p.canvasH + 1) >> 1) : p.canvasH;
        const uint32_t cx0 = (p.x0 + tileX) >> 1;
        const uint32_t cy0 = (SUB == SUB_420) ? ((p.y0 + tileY) >> 1) : (p.y0 + tileY);
        constexpr int kRows = (SUB == SUB_420) ? (kTileH / 2 + 2) : kTileH;
        const int t = ty * kLanesX + tx;
        {
            const int row = t >> 5, grp = t & 31; // 32 groups of 4 samples per row
            if (row < kRows) {
                int cy = (SUB == SUB_420) ? ((int)cy0 - 1 + row) : ((int)cy0 + row);
                cy = cy < 0 ? 0 : (cy > (int)ch - 1 ? (int)ch - 1 : cy);
                unsigned uq[4], vq[4];
                load4<YT>(s.plane[1] + (size_t)cy * s.rowBytes[1] + (size_t)(cx0 + 4 * grp) * sizeof(YT), uq);
                load4<YT>(s.plane[2] + (size_t)cy * s.rowBytes[2] + (size_t)(cx0 + 4 * grp) * sizeof(YT), vq);
                float2 f0, f1;
                f0.x = normUVT<FAST>(kWide ? minU(uq[0], yuvMax) : uq[0], s);
                f0.y = normUVT<FAST>(kWide ? minU(uq[1], yuvMax) : uq[1], s);
                f1.x = normUVT<FAST>(kWide ? minU(uq[2], yuvMax) : uq[2], s);
                f1.y = normUVT<FAST>(kWide ? minU(uq[3], yuvMax) : uq[3], s);
                *reinterpret_cast<float2 *>(&sU[row][2 + 4 * grp]) = f0;
                *reinterpret_cast<float2 *>(&sU[row][4 + 4 * grp]) = f1;
                f0.x = normUVT<FAST>(kWide ? minU(vq[0], yuvMax) : vq[0], s);
                f0.y = normUVT<FAST>(kWide ? minU(vq[1], yuvMax) : vq[1], s);
                f1.x = normUVT<FAST>(kWide ? minU(vq[2], yuvMax) : vq[2], s);
                f1.y = normUVT<FAST>(kWide ? minU(vq[3], yuvMax) : vq[3], s);
                *reinterpret_cast<float2 *>(&sV[row][2 + 4 * grp]) = f0;
                *reinterpret_cast<float2 *>(&sV[row][4 + 4 * grp]) = f1;
            }
        }
        if (t < 4 * kRows) {
            // halo columns: lane -> (row, side, plane); sample coordinates clamp to the canvas, which is
            // exactly the reference's border rule (src/reformat.c:768,784): the neighbour of an edge
            // sample is the sample itself.
            const int row = t >> 2, side = (t >> 1) & 1, plane = t & 1;
            int cy = (SUB == SUB_420) ? ((int)cy0 - 1 + row) : ((int)cy0 + row);
            cy = cy < 0 ? 0 : (cy > (int)ch - 1 ? (int)ch - 1 : cy);
            int cx = side ? (int)cx0 + 128 : (int)cx0 - 1;
            cx = cx < 0 ? 0 : (cx > (int)cw - 1 ? (int)cw - 1 : cx);
            unsigned v = load1<YT>(s.plane[1 + plane], s.rowBytes[1 + plane], (uint32_t)cx, (uint32_t)cy);
            if (kWide)
                v = minU(v, yuvMax);
            float (*dstPlane)[kChromaPitch] = plane ? sV : sU;
            dstPlane[row][side ? (2 + 128) : 1] = normUVT<FAST>(v, s);
        }
        __syncthreads();

        // lane's neighbourhood: LDS columns 2tx+1 .. 2tx+4  (canvas chroma columns cxL-1 .. cxL+2)
        const int colA = 2 * tx + 1;
        const int rowM0 = (SUB == SUB_420) ? (ty + 1) : (2 * ty);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            float (*src)[kChromaPitch] = pl ? sV : sU;
            float(*out)[4] = pl ? cr : cb;
            if constexpr (SUB == SUB_420) {
                float m[4], tp[4], bt[4];
                m[0] = src[rowM0][colA];
                const float2 mm = *reinterpret_cast<const float2 *>(&src[rowM0][colA + 1]);
                m[1] = mm.x, m[2] = mm.y;
                m[3] = src[rowM0][colA + 3];
                tp[0] = src[rowM0 - 1][colA];
                const float2 tt = *reinterpret_cast<const float2 *>(&src[rowM0 - 1][colA + 1]);
                tp[1] = tt.x, tp[2] = tt.y;
                tp[3] = src[rowM0 - 1][colA + 3];
                bt[0] = src[rowM0 + 1][colA];
                const float2 bb = *reinterpret_cast<const float2 *>(&src[rowM0 + 1][colA + 1]);
                bt[1] = bb.x, bt[2] = bb.y;
                bt[3] = src[rowM0 + 1][colA + 3];
                // products shared between the lane's pixels (each equals the reference's product for that tap)
                const float m9b = m[1] * (9.0f / 16.0f), m9c = m[2] * (9.0f / 16.0f);
                const float m3a = m[0] * (3.0f / 16.0f), m3b = m[1] * (3.0f / 16.0f), m3c = m[2] * (3.0f / 16.0f), m3d = m[3] * (3.0f / 16.0f);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float * v = r ? bt : tp; // vertical neighbour row: above for even luma rows, below for odd
                    const float v3b = v[1] * (3.0f / 16.0f), v3c = v[2] * (3.0f / 16.0f);
                    const float v1a = v[0] * (1.0f / 16.0f), v1b = v[1] * (1.0f / 16.0f), v1c = v[2] * (1.0f / 16.0f), v1d = v[3] * (1.0f / 16.0f);
                    out[r][0] = ((m9b + m3a) + v3b) + v1a; // even pixel: horizontal neighbour on the left
                    out[r][1] = ((m9b + m3c) + v3b) + v1c; // odd pixel: on the right
                    out[r][2] = ((m9c + m3b) + v3c) + v1b;
                    out[r][3] = ((m9c + m3d) + v3c) + v1d;
                }
            } else { // 4:2:2: vertical neighbour is the sample itself (src/reformat.c:784-786)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float m[4];
                    m[0] = src[rowM0 + r][colA];
                    const float2 mm = *reinterpret_cast<const float2 *>(&src[rowM0 + r][colA + 1]);
                    m[1] = mm.x, m[2] = mm.y;
                    m[3] = src[rowM0 + r][colA + 3];
                    const float m9b = m[1] * (9.0f / 16.0f), m9c = m[2] * (9.0f / 16.0f);
                    const float m3a = m[0] * (3.0f / 16.0f), m3b = m[1] * (3.0f / 16.0f), m3c = m[2] * (3.0f / 16.0f), m3d = m[3] * (3.0f / 16.0f);
                    const float m1a = m[0] * (1.0f / 16.0f), m1b = m[1] * (1.0f / 16.0f), m1c = m[2] * (1.0f / 16.0f), m1d = m[3] * (1.0f / 16.0f);
                    out[r][0] = ((m9b + m3a) + m3b) + m1a;
                    out[r][1] = ((m9b + m3c) + m3b) + m1c;
                    out[r][2] = ((m9c + m3b) + m3c) + m1b;
                    out[r][3] = ((m9c + m3d) + m3c) + m1d;
                }
            }
        }
    }

    // ---- per-pixel arithmetic and stores ----
    const bool swapRB = (o.offB < o.offR);
    const bool alphaFirst = (NCH == 4) && (o.offA == 0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        PixelOut q[4];
        unsigned a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned unormY = kWide ? minU(yv[r][k], yuvMax) : yv[r][k];
            const unsigned unormA = needAlpha ? av[r][k] : 0u;
            a[k] = (unsigned)o.maxv;
            if (NCH == 4 && p.alphaSource == ALPHA_PLANE)
                a[k] = (s.depth == o.depth) ? unormA : rescaleAlpha(unormA, (float)s.maxv, o.maxf, o.maxv);
            if constexpr (FAST && sizeof(RT) == 1 && NCH == 4 && !HASMUL) {
                // 8-bit RGBA family: quantise, clamp and pack in one saturating byte conversion per channel
                float R, G, B;
                matrixRgb<true>(s, normYT<true>(unormY, s), cb[r][k], cr[r][k], R, G, B);
                unsigned word = a[k] << (8 * o.offA);
                word = packByte(R, o.maxf, (unsigned)o.offR, word);
                word = packByte(G, o.maxf, (unsigned)o.offG, word);
                word = packByte(B, o.maxf, (unsigned)o.offB, word);
                q[k].r = word;
            } else {
                q[k] = finishPixel<HASMUL, FAST>(p, normYT<FAST>(unormY, s), cb[r][k], cr[r][k], unormA, a[k]);
            }
        }
        uint8_t * dst = o.pixels + (size_t)(Y0 + r) * o.rowBytes + (size_t)X * (NCH * sizeof(RT));
        if constexpr (FAST && sizeof(RT) == 1 && NCH == 4 && !HASMUL) {
            uint4 w;
            w.x = q[0].r, w.y = q[1].r, w.z = q[2].r, w.w = q[3].r;
            *reinterpret_cast<uint4 *>(dst) = w;
        } else {
            store4<RT, NCH>(dst, q, a, swapRB, alphaFirst);
        }
    }
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL, bool FAST>
__global__ __launch_bounds__(256) void yuvToRgbTileKernel(YuvToRgbPlan p)
{
    __shared__ __attribute__((aligned(16))) float sU[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    __shared__ __attribute__((aligned(16))) float sV[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    tileBody<YT, SUB, BILINEAR, RT, NCH, HASMUL, FAST>(p, sU, sV);
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL, bool FAST>
__global__ __launch_bounds__(256) void yuvToRgbTileBatchKernel(const YuvToRgbPlan * __restrict__ table)
{
    __shared__ __attribute__((aligned(16))) float sU[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    __shared__ __attribute__((aligned(16))) float sV[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    __shared__ YuvToRgbPlan plan;
    {
        // one cooperative copy of the descriptor into LDS keeps it out of per-lane registers
        const uint32_t * src = reinterpret_cast<const uint32_t *>(&table[blockIdx.z]);
        uint32_t * dst = reinterpret_cast<uint32_t *>(&plan);
        const int t = threadIdx.y * kLanesX + threadIdx.x;
        for (int k = t; k < (int)(sizeof(YuvToRgbPlan) / 4); k += 256)
            dst[k] = src[k];
    }
    __syncthreads();
    tileBody<YT, SUB, BILINEAR, RT, NCH, HASMUL, FAST>(plan, sU, sV);
}

struct TileKey
{
    bool wideYuv;
    int sub;
    bool bilinear;
    bool wideRgb;
    int nch;
    bool hasMul;
    bool fast;
};

TileKey keyFor(const YuvToRgbPlan & p)
{
    TileKey k;
    k.wideYuv = p.yuv.chanBytes == 2;
    if (!p.yuv.hasColor)
        k.sub = SUB_400;
    else if (p.yuv.format == AVIF_PIXEL_FORMAT_YUV444)
        k.sub = SUB_444;
    else if (p.yuv.format == AVIF_PIXEL_FORMAT_YUV422)
        k.sub = SUB_422;
    else
        k.sub = SUB_420;
    k.bilinear = p.bilinear && (k.sub == SUB_420 || k.sub == SUB_422);
    k.wideRgb = p.rgb.chanBytes == 2;
    k.nch = p.rgb.hasAlpha ? 4 : 3;
    k.hasMul = (p.inLoopMul != MUL_NONE) || (p.postMul != MUL_NONE);
    k.fast = (p.tuning & TUNE_EXACT_RECIPROCAL) && (p.tuning & TUNE_SATURATING_PACK) && p.yuv.exactNorm && p.yuv.exactKg;
    return k;
}

bool aligned(const void * ptr, uint32_t rowBytes, uint32_t a)
{
    return ((uintptr_t)ptr % a) == 0 && (rowBytes % a) == 0;
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool MUL, bool FAST>
hipError_t launchOne(const YuvToRgbPlan * plan, const YuvToRgbPlan * table, uint32_t count, uint32_t w, uint32_t h, hipStream_t stream)
{
    const dim3 block(kLanesX, kLanesY);
    const dim3 grid((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH, count);
    if (table)
        hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, MUL, FAST>), grid, block, 0, stream, table);
    else
        hipLaunchKernelGGL((yuvToRgbTileKernel<YT, SUB, BIL, RT, NCH, MUL, FAST>), grid, block, 0, stream, *plan);
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL>
hipError_t launchRgbVariant(const TileKey & k, const YuvToRgbPlan * plan, const YuvToRgbPlan * table, uint32_t count, uint32_t w, uint32_t h, hipStream_t stream)
{
#define AVIFHIP_RGB_CASE(RT, NCH)                                                                      \
    return k.hasMul ? (k.fast ? launchOne<YT, SUB, BIL, RT, NCH, true, true>(plan, table, count, w, h, stream)    \
                              : launchOne<YT, SUB, BIL, RT, NCH, true, false>(plan, table, count, w, h, stream))  \
                    : (k.fast ? launchOne<YT, SUB, BIL, RT, NCH, false, true>(plan, table, count, w, h, stream)   \
                              : launchOne<YT, SUB, BIL, RT, NCH, false, false>(plan, table, count, w, h, stream))
    if (!k.wideRgb) {
        if (k.nch == 4) {
            AVIFHIP_RGB_CASE(uint8_t, 4);
        }
        AVIFHIP_RGB_CASE(uint8_t, 3);
    }
    if (k.nch == 4) {
        AVIFHIP_RGB_CASE(uint16_t, 4);
    }
    AVIFHIP_RGB_CASE(uint16_t, 3);
#undef AVIFHIP_RGB_CASE
}

template <typename YT>
hipError_t launchYuvVariant(const TileKey & k, const YuvToRgbPlan * plan, const YuvToRgbPlan * table, uint32_t count, uint32_t w, uint32_t h, hipStream_t stream)
{
    switch (k.sub) {
        case SUB_444: return launchRgbVariant<YT, SUB_444, false>(k, plan, table, count, w, h, stream);
        case SUB_400: return launchRgbVariant<YT, SUB_400, false>(k, plan, table, count, w, h, stream);
        case SUB_422:
            return k.bilinear ? launchRgbVariant<YT, SUB_422, true>(k, plan, table, count, w, h, stream)
                              : launchRgbVariant<YT, SUB_422, false>(k, plan, table, count, w, h, stream);
        default:
            return k.bilinear ? launchRgbVariant<YT, SUB_420, true>(k, plan, table, count, w, h, stream)
                              : launchRgbVariant<YT, SUB_420, false>(k, plan, table, count, w, h, stream);
    }
}

const char * kernelNameFor(const TileKey & k)
{
    static thread_local char name[96];
    static const char * subs[] = { "444", "422", "420", "400" };
    snprintf(name, sizeof(name), "yuv2rgb_tile<%s,%s,%s,%s%d%s%s>", k.wideYuv ? "u16" : "u8", subs[k.sub], k.bilinear ? "bilinear" : "nearest",
             k.nch == 4 ? "rgba" : "rgb", k.wideRgb ? 16 : 8, k.hasMul ? ",alphamul" : "", k.fast ? "" : ",ieee-div");
    return name;
}

} // namespace

bool tileYuvToRgbSupported(const YuvToRgbPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    if (p.arith != ARITH_FLOAT || p.identityCopy || s.mode != MODE_COEFF)
        return false;
    if (o.isGray || o.is565 || o.isFloat)
        return false;
    if (o.hasAlpha && p.alphaSource == ALPHA_KEEP)
        return false; // destination alpha bytes must stay untouched: per-channel stores only
    if (p.w < (uint32_t)kTileW || p.h < (uint32_t)kTileH)
        return false; // nothing but partial tiles
    // vector accesses: 4 samples per load, 4 pixels per store
    const uint32_t sampleVec = 4 * (uint32_t)s.chanBytes;
    if ((p.x0 % 8) != 0 || (p.y0 % 2) != 0)
        return false;
    if (!aligned(s.plane[0], s.rowBytes[0], sampleVec))
        return false;
    if (s.hasColor && (!aligned(s.plane[1], s.rowBytes[1], sampleVec) || !aligned(s.plane[2], s.rowBytes[2], sampleVec)))
        return false;
    if (((o.hasAlpha && p.alphaSource == ALPHA_PLANE) || p.inLoopMul != MUL_NONE) && !aligned(s.alpha, s.alphaRowBytes, sampleVec))
        return false;
    const int nch = o.hasAlpha ? 4 : 3;
    const uint32_t storeAlign = (nch == 4) ? 16u : (o.chanBytes == 1 ? 4u : 8u);
    if (!aligned(o.pixels, o.rowBytes, storeAlign))
        return false;
    return true;
}

int tileYuvToRgbVariant(const YuvToRgbPlan & plan)
{
    if (!tileYuvToRgbSupported(plan))
        return -1;
    const TileKey k = keyFor(plan);
    return (k.wideYuv ? 1 : 0) | (k.sub << 1) | ((k.bilinear ? 1 : 0) << 3) | ((k.wideRgb ? 1 : 0) << 4) | ((k.nch == 4 ? 1 : 0) << 5) |
           ((k.hasMul ? 1 : 0) << 6) | ((k.fast ? 1 : 0) << 7);
}

hipError_t launchYuvToRgbTile(const YuvToRgbPlan & plan, hipStream_t stream, const char ** kernelName)
{
    const TileKey k = keyFor(plan);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    return k.wideYuv ? launchYuvVariant<uint16_t>(k, &plan, nullptr, 1, plan.w, plan.h, stream)
                     : launchYuvVariant<uint8_t>(k, &plan, nullptr, 1, plan.w, plan.h, stream);
}

hipError_t launchYuvToRgbTileBatch(const YuvToRgbPlan * deviceTable, const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW,
                                   uint32_t maxH, hipStream_t stream, const char ** kernelName)
{
    const TileKey k = keyFor(representative);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    return k.wideYuv ? launchYuvVariant<uint16_t>(k, nullptr, deviceTable, count, maxW, maxH, stream)
                     : launchYuvVariant<uint8_t>(k, nullptr, deviceTable, count, maxW, maxH, stream);
}

bool tileRgbToYuvSupported(const RgbToYuvPlan &)
{
    return false;
}
hipError_t launchRgbToYuvTile(const RgbToYuvPlan &, hipStream_t, const char **)
{
    return hipErrorNotSupported;
}

} // namespace avifhip
