// tile_impl.h -- bandwidth-tuned YUV->RGB kernels for gfx950 (MI355X), instantiated per sample type by
// kernels_tile_u8.hip / kernels_tile_u16.hip.
//
// Scope: matrix-coefficient ("normal YUV") conversions into interleaved 3- or 4-channel RGB at 8-bit or 16-bit
// containers, from 8-bit or 16-bit-container 4:4:4 / 4:2:2 / 4:2:0 / 4:0:0 planes, nearest or bilinear chroma
// upsampling, alpha fill / copy / rescale and both flavours of alpha (un)premultiply -- every BASELINE
// configuration and what avifdec asks for.  Everything else (gray outputs, RGB565, half float, identity / YCgCo
// matrices, unaligned user buffers, divisors off the verified list) is served by kernels_generic.hip, whose
// per-pixel routine is also the fallback for pixel groups cut by the image border here.
//
// Structure (wave = 64 lanes, workgroup = 4 waves, one workgroup per 256x8-pixel tile):
//   * lane (tx,ty) owns 4 consecutive pixels of rows 2ty and 2ty+1 of the tile: one 4-sample vector load per plane
//     and row (row-coalesced), one 16-byte store per row for RGBA8 (1 KiB contiguous per wave instruction);
//   * bilinear chroma: the tile's chroma neighbourhood (4:2:0: 6 rows x 130 samples per plane) is normalised to
//     fp32 once per sample and staged in LDS as interleaved (Cb,Cr) pairs; each lane reads its 4x3 neighbourhood
//     with six 16-byte LDS loads and filters both planes at once with packed fp32 instructions, sharing the
//     9/16, 3/16, 1/16 products between its pixels (the reference re-reads and re-normalises up to four chroma
//     samples for every output pixel);
//   * workgroups of one XCD (blockIdx % 8) take a contiguous band of tiles, so the chroma halo rows shared by
//     vertically adjacent tiles are re-read from that XCD's L2 rather than from HBM;
//   * arithmetic: the reference's operations in the reference's order (pixel_math.h), issued two at a time as
//     v_pk_{add,mul,fma}_f32; divisions by plan constants use the exhaustively verified fma(x, hi, x*lo) form
//     (exactdiv.h); 8-bit outputs are quantised, clamped and packed by v_cvt_pk_u8_f32 executed in
//     round-toward-zero mode (it saturates to [0,255] and follows MODE.FP_ROUND: tests/tools/probe_cvt_mode.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <stdio.h>

#include "kernels.h"
#include "pixel_generic.h"
#include "pixel_math.h"
#include "tile_shared.h"

namespace avifhip {
namespace tile {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kTileW = 256;
constexpr int kTileH = 8;
constexpr int kLanesX = 64; // 4 pixels each
constexpr int kLanesY = 4;  // 2 rows each
// LDS chroma row: entry c+1 holds the (Cb,Cr) pair of chroma column cx0 + c, c in [-1, 128]; a lane's four
// columns 2tx-1 .. 2tx+2 are entries 2tx .. 2tx+3: two 16-byte aligned loads
constexpr int kChromaPitch = 132;
constexpr int kChromaRowsMax = 8;

__device__ __forceinline__ f2 splat(float v)
{
    return (f2) { v, v };
}
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}
// (cp - bias) / range for two samples, src/reformat.c:583,598 (verified reciprocal form)
__device__ __forceinline__ f2 norm2(f2 cp, float bias, RcpHL r)
{
    const f2 n = cp - splat(bias);
    return fma2(n, splat(r.hi), n * splat(r.lo));
}

template <typename T>
__device__ __forceinline__ void loadRaw4(const uint8_t * p, unsigned w[2])
{
    if constexpr (sizeof(T) == 1) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
        w[1] = 0;
    } else {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        w[0] = t.x;
        w[1] = t.y;
    }
}
template <typename T>
__device__ __forceinline__ void decode4(const unsigned w[2], unsigned v[4])
{
    if constexpr (sizeof(T) == 1) {
        v[0] = w[0] & 0xffu;
        v[1] = (w[0] >> 8) & 0xffu;
        v[2] = (w[0] >> 16) & 0xffu;
        v[3] = w[0] >> 24;
    } else {
        v[0] = w[0] & 0xffffu;
        v[1] = w[0] >> 16;
        v[2] = w[1] & 0xffffu;
        v[3] = w[1] >> 16;
    }
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
__device__ __forceinline__ int clampI(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}
// four samples of one plane as floats, clamped to the depth's maximum for 16-bit containers (src/reformat.c:712,821)
template <typename T>
__device__ __forceinline__ void samples4(const unsigned w[2], unsigned yuvMax, float f[4])
{
    unsigned v[4];
    decode4<T>(w, v);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        f[k] = (float)((sizeof(T) == 2) ? minU(v[k], yuvMax) : v[k]);
}

struct PixelOut
{
    unsigned r, g, b;
};

// alpha at the RGB depth from a plane sample: copy or depth rescale (src/alpha.c:84-103, verified reciprocal form)
__device__ __forceinline__ unsigned alphaFromPlane(const YuvToRgbPlan & p, unsigned sa)
{
    if (p.yuv.depth == p.rgb.depth)
        return sa;
    const float alphaF = divExact((float)sa, p.yuv.rcpMax);
    const int dstAlpha = (int)(0.5f + (alphaF * p.rgb.maxf));
    return (unsigned)clampInt(dstAlpha, 0, p.rgb.maxv);
}

// (un)premultiply on stored integers with the verified reciprocal for "/ maxF" (src/alpha.c:180-192, :367-381)
__device__ __forceinline__ unsigned alphaMulIntFast(const RgbSide & o, unsigned c, unsigned a, int mulMode)
{
    if (a >= (unsigned)o.maxv)
        return c;
    if (a == 0)
        return 0;
    if (mulMode == MUL_MULTIPLY)
        return (unsigned)roundHalfUp(divExact((float)c * (float)a, o.rcpMax));
    return unpremultiplyInt(c, a, o.maxf);
}

// General finish of one pixel from unclamped R,G,B: clamp, optional fp32 alpha multiply, quantise, optional integer
// alpha multiply (src/reformat.c:886-961, :1574-1585).
template <bool kHasMul>
__device__ __forceinline__ PixelOut finishPixel(const YuvToRgbPlan & p, float R, float G, float B, unsigned unormA, unsigned a)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    float Rc = clamp01(R), Gc = clamp01(G), Bc = clamp01(B);
    if (kHasMul && p.inLoopMul != MUL_NONE) {
        const float Ac = clamp01(divExact((float)minU(unormA, (unsigned)s.maxv), s.rcpMax));
        Rc = applyAlphaF(Rc, Ac, p.inLoopMul);
        Gc = applyAlphaF(Gc, Ac, p.inLoopMul);
        Bc = applyAlphaF(Bc, Ac, p.inLoopMul);
    }
    PixelOut q;
    q.r = quantize(Rc, o.maxf);
    q.g = quantize(Gc, o.maxf);
    q.b = quantize(Bc, o.maxf);
    if (kHasMul && p.postMul != MUL_NONE) {
        q.r = alphaMulIntFast(o, q.r, a, p.postMul);
        q.g = alphaMulIntFast(o, q.g, a, p.postMul);
        q.b = alphaMulIntFast(o, q.b, a, p.postMul);
    }
    return q;
}

template <typename V>
__device__ __forceinline__ void storeVec(V * dst, const V & v, bool nontemporal)
{
    if (nontemporal)
        __builtin_nontemporal_store(v, dst);
    else
        *dst = v;
}

// Store 4 consecutive pixels.  swapRB: B is the first colour channel; alphaFirst: A precedes colour.
template <typename RT, int NCH>
__device__ __forceinline__ void store4(uint8_t * dst, const PixelOut q[4], const unsigned a[4], bool swapRB, bool alphaFirst, bool nt)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    unsigned x[4], z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = swapRB ? q[k].b : q[k].r;
        z[k] = swapRB ? q[k].r : q[k].b;
    }
    if constexpr (sizeof(RT) == 1 && NCH == 4) {
        u4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = alphaFirst ? (a[k] | (x[k] << 8) | (q[k].g << 16) | (z[k] << 24)) : (x[k] | (q[k].g << 8) | (z[k] << 16) | (a[k] << 24));
        storeVec(reinterpret_cast<u4 *>(dst), w, nt);
    } else if constexpr (sizeof(RT) == 1 && NCH == 3) {
        // 12 bytes: x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3 (rows and pixel groups are 4-byte aligned)
        unsigned * d = reinterpret_cast<unsigned *>(dst);
        d[0] = x[0] | (q[0].g << 8) | (z[0] << 16) | (x[1] << 24);
        d[1] = q[1].g | (z[1] << 8) | (x[2] << 16) | (q[2].g << 24);
        d[2] = z[2] | (x[3] << 8) | (q[3].g << 16) | (z[3] << 24);
    } else if constexpr (sizeof(RT) == 2 && NCH == 4) {
        u4 w0, w1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = alphaFirst ? (a[k] | (x[k] << 16)) : (x[k] | (q[k].g << 16));
            const unsigned hi = alphaFirst ? (q[k].g | (z[k] << 16)) : (z[k] | (a[k] << 16));
            if (k < 2) {
                w0[(k & 1) * 2 + 0] = lo;
                w0[(k & 1) * 2 + 1] = hi;
            } else {
                w1[(k & 1) * 2 + 0] = lo;
                w1[(k & 1) * 2 + 1] = hi;
            }
        }
        storeVec(reinterpret_cast<u4 *>(dst), w0, nt);
        storeVec(reinterpret_cast<u4 *>(dst) + 1, w1, nt);
    } else { // 16-bit, 3 channels: 24 bytes = 3 x 8
        u2 w0, w1, w2;
        w0.x = x[0] | (q[0].g << 16);
        w0.y = z[0] | (x[1] << 16);
        w1.x = q[1].g | (z[1] << 16);
        w1.y = x[2] | (q[2].g << 16);
        w2.x = z[2] | (x[3] << 16);
        w2.y = q[3].g | (z[3] << 16);
        reinterpret_cast<u2 *>(dst)[0] = w0;
        reinterpret_cast<u2 *>(dst)[1] = w1;
        reinterpret_cast<u2 *>(dst)[2] = w2;
    }
    (void)sizeof(u3);
}

// 8-bit RGBA family: (uint8_t)(0.5f + clamp01(c) * 255) for the three colour channels of four pixels, inserted into
// words that already hold the alpha byte.  The inputs are t = 0.5f + c * 255 (unclamped c); v_cvt_pk_u8_f32 in
// round-toward-zero mode truncates like the C cast and saturates to [0, 255], and because t is monotonic in c the
// saturation selects the same byte as clamping c first.  The rounding mode is changed only inside this block.
__device__ __forceinline__ void packRgba8Row(unsigned w[4], const f2 br[4], const f2 g01, const f2 g23, unsigned slotR, unsigned slotG, unsigned slotB)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, %18, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %6, %18, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %8, %18, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %10, %18, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, %16, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, %16, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %9, %16, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %11, %16, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %12, %17, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %13, %17, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, %17, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %15, %17, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
                 : "v"(br[0].x), "v"(br[0].y), "v"(br[1].x), "v"(br[1].y), "v"(br[2].x), "v"(br[2].y), "v"(br[3].x), "v"(br[3].y), "v"(g01.x),
                   "v"(g01.y), "v"(g23.x), "v"(g23.y), "s"(slotR), "s"(slotG), "s"(slotB));
}

// 8-bit RGB / BGR: 12 bytes x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3 from t = 0.5f + c * 255 (see packRgba8Row)
__device__ __forceinline__ void packRgb8Row(unsigned w[3], const float x[4], const float g[4], const float z[4])
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %12, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %6, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %11, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %5, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %10, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(z[0]), "v"(z[1]), "v"(z[2]),
                   "v"(z[3]));
    // operand map: %3..%6 = x0..x3, %7..%10 = g0..g3, %11..%14 = z0..z3
    // w0 = x0 g0 z0 x1 ; w1 = g1 z1 x2 g2 ; w2 = z2 x3 g3 z3
}

// tile index of this workgroup: workgroups are dispatched round-robin over the 8 XCDs, so giving XCD x the x-th
// contiguous band of tiles keeps vertically adjacent tiles (which share chroma halo rows) on one L2
__device__ __forceinline__ uint32_t tileOfBlock(uint32_t b, uint32_t n, bool bands)
{
    if (!bands || n < 64)
        return b;
    const uint32_t per = n >> 3, rem = n & 7;
    const uint32_t xcd = b & 7, slot = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL>
__device__ __forceinline__ void runTile(const YuvToRgbPlan & p, uint32_t nBlocks, f2 (*sC)[kChromaPitch])
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    constexpr bool kWide = sizeof(YT) == 2;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const unsigned yuvMax = (unsigned)s.maxv;

    const uint32_t tilesX = (p.w + kTileW - 1) / kTileW;
    const uint32_t tilesY = (p.h + kTileH - 1) / kTileH;
    const uint32_t nTiles = tilesX * tilesY;
    const uint32_t tileIndex = tileOfBlock(blockIdx.x, nBlocks, (p.tuning & TUNE_XCD_BANDS) != 0 && (gridDim.z == 1 || (nBlocks & 7) == 0));
    if (tileIndex >= nTiles)
        return;
    const uint32_t trow = tileIndex / tilesX;
    const uint32_t tileX = (tileIndex - trow * tilesX) * kTileW;
    const uint32_t tileY = trow * kTileH;
    const bool edge = (tileX + kTileW > p.w) || (tileY + kTileH > p.h);
    const bool groupFull = tileX + 4 * tx + 3 < p.w;
    const uint32_t X = p.x0 + tileX + 4 * tx;
    const uint32_t Y0 = p.y0 + tileY + 2 * ty;
    const bool needAlpha = (NCH == 4 && p.alphaSource == ALPHA_PLANE) || (HASMUL && p.inLoopMul != MUL_NONE);
    const bool nt = (p.tuning & TUNE_NONTEMPORAL) != 0;

    // ---- bilinear: stage the tile's chroma neighbourhood in LDS (loads first, they are the longest chain) ----
    if constexpr (BILINEAR) {
        constexpr int kRows = (SUB == SUB_420) ? (kTileH / 2 + 2) : kTileH;
        const uint32_t cw = (p.canvasW + 1) >> 1;
        const uint32_t ch = (SUB == SUB_420) ? ((p.canvasH + 1) >> 1) : p.canvasH;
        const uint32_t cx0 = (p.x0 + tileX) >> 1;
        const uint32_t cy0 = (SUB == SUB_420) ? ((p.y0 + tileY) >> 1) : (p.y0 + tileY);
        const int t = ty * kLanesX + tx;
        // roles: lane t < 32*rows stages chroma group (row = t>>5, grp = t&31) of both planes; the LAST 2*rows lanes of
        // the workgroup stage one halo column each (row = h>>1, side = h&1)
        const int row = t >> 5, grp = t & 31;
        if (row < kRows) {
            // LDS row q holds canvas chroma row clamp(cy0 - 1 + q) for 4:2:0 or cy0 + q for 4:2:2; coordinates clamp
            // to the canvas: exactly the reference's border rule (src/reformat.c:768,784)
            const int cy = clampI((SUB == SUB_420) ? ((int)cy0 - 1 + row) : ((int)cy0 + row), 0, (int)ch - 1);
            const uint32_t cx = cx0 + 4 * grp;
            float fu[4], fv[4];
            if (cx + 3 < cw) {
                unsigned wu[2], wv[2];
                loadRaw4<YT>(s.plane[1] + (size_t)cy * s.rowBytes[1] + (size_t)cx * sizeof(YT), wu);
                loadRaw4<YT>(s.plane[2] + (size_t)cy * s.rowBytes[2] + (size_t)cx * sizeof(YT), wv);
                samples4<YT>(wu, yuvMax, fu);
                samples4<YT>(wv, yuvMax, fv);
            } else {
                // group cut by the right border of the canvas (edge tiles only)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cxk = clampI((int)cx + k, 0, (int)cw - 1);
                    const unsigned u = load1<YT>(s.plane[1], s.rowBytes[1], (uint32_t)cxk, (uint32_t)cy);
                    const unsigned v = load1<YT>(s.plane[2], s.rowBytes[2], (uint32_t)cxk, (uint32_t)cy);
                    fu[k] = (float)(kWide ? minU(u, yuvMax) : u);
                    fv[k] = (float)(kWide ? minU(v, yuvMax) : v);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                sC[row][1 + 4 * grp + k] = norm2((f2) { fu[k], fv[k] }, s.biasUV, s.rcpRangeUV);
        }
        const int h = kLanesX * kLanesY - 1 - t;
        if (h < 2 * kRows) {
            const int hrow = h >> 1, side = h & 1;
            const int cy = clampI((SUB == SUB_420) ? ((int)cy0 - 1 + hrow) : ((int)cy0 + hrow), 0, (int)ch - 1);
            const int cx = clampI(side ? (int)cx0 + 128 : (int)cx0 - 1, 0, (int)cw - 1);
            const unsigned u = load1<YT>(s.plane[1], s.rowBytes[1], (uint32_t)cx, (uint32_t)cy);
            const unsigned v = load1<YT>(s.plane[2], s.rowBytes[2], (uint32_t)cx, (uint32_t)cy);
            sC[hrow][side ? 129 : 0] = norm2((f2) { (float)(kWide ? minU(u, yuvMax) : u), (float)(kWide ? minU(v, yuvMax) : v) }, s.biasUV, s.rcpRangeUV);
        }
    }

    // ---- this lane's luma / alpha / co-sited chroma: issued before the barrier so they overlap the staging ----
    unsigned rawY[2][2], rawA[2][2], rawU[2][2], rawV[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool ok = groupFull && (tileY + 2 * ty + r < p.h);
        rawY[r][0] = rawY[r][1] = rawA[r][0] = rawA[r][1] = 0;
        rawU[r][0] = rawU[r][1] = rawV[r][0] = rawV[r][1] = 0;
        if (ok) {
            loadRaw4<YT>(s.plane[0] + (size_t)(Y0 + r) * s.rowBytes[0] + (size_t)X * sizeof(YT), rawY[r]);
            if (needAlpha)
                loadRaw4<YT>(s.alpha + (size_t)(Y0 + r) * s.alphaRowBytes + (size_t)X * sizeof(YT), rawA[r]);
            if constexpr (SUB == SUB_444) {
                loadRaw4<YT>(s.plane[1] + (size_t)(Y0 + r) * s.rowBytes[1] + (size_t)X * sizeof(YT), rawU[r]);
                loadRaw4<YT>(s.plane[2] + (size_t)(Y0 + r) * s.rowBytes[2] + (size_t)X * sizeof(YT), rawV[r]);
            } else if constexpr ((SUB == SUB_420 || SUB == SUB_422) && !BILINEAR) {
                // nearest: chroma samples (X>>1, X>>1 + 1) of chroma row (j >> shiftY); one aligned pair load per plane
                if (!(SUB == SUB_420 && r == 1)) {
                    const uint32_t cy = (SUB == SUB_420) ? (Y0 >> 1) : (Y0 + r);
                    const size_t off = (size_t)(X >> 1) * sizeof(YT);
                    if constexpr (sizeof(YT) == 1) {
                        rawU[r][0] = *reinterpret_cast<const uint16_t *>(s.plane[1] + (size_t)cy * s.rowBytes[1] + off);
                        rawV[r][0] = *reinterpret_cast<const uint16_t *>(s.plane[2] + (size_t)cy * s.rowBytes[2] + off);
                    } else {
                        rawU[r][0] = *reinterpret_cast<const uint32_t *>(s.plane[1] + (size_t)cy * s.rowBytes[1] + off);
                        rawV[r][0] = *reinterpret_cast<const uint32_t *>(s.plane[2] + (size_t)cy * s.rowBytes[2] + off);
                    }
                }
            }
        }
    }

    if constexpr (BILINEAR)
        __syncthreads();

    if (edge && !groupFull) {
        // pixel group cut by the right border: per-pixel routine for the pixels that exist
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            if (tileY + 2 * ty + r >= p.h)
                continue;
#pragma unroll 1
            for (int k = 0; k < 4; ++k)
                if (tileX + 4 * tx + k < p.w)
                    yuvToRgbPixel(p, X + k, Y0 + r);
        }
        return;
    }

    // ---- (Cb,Cr) for the lane's 2 x 4 pixels ----
    f2 uv[2][4];
    if constexpr (SUB == SUB_400) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                uv[r][k] = splat(0.5f);
    } else if constexpr (SUB == SUB_444) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float fu[4], fv[4];
            samples4<YT>(rawU[r], yuvMax, fu);
            samples4<YT>(rawV[r], yuvMax, fv);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                uv[r][k] = norm2((f2) { fu[k], fv[k] }, s.biasUV, s.rcpRangeUV);
        }
    } else if constexpr (!BILINEAR) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (SUB == SUB_420 && r == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    uv[1][k] = uv[0][k];
                break;
            }
            constexpr unsigned kMask = kWide ? 0xffffu : 0xffu;
            constexpr int kShift = kWide ? 16 : 8;
            unsigned u0 = rawU[r][0] & kMask, u1 = (rawU[r][0] >> kShift) & kMask;
            unsigned v0 = rawV[r][0] & kMask, v1 = (rawV[r][0] >> kShift) & kMask;
            if (kWide) {
                u0 = minU(u0, yuvMax), u1 = minU(u1, yuvMax), v0 = minU(v0, yuvMax), v1 = minU(v1, yuvMax);
            }
            const f2 c0 = norm2((f2) { (float)u0, (float)v0 }, s.biasUV, s.rcpRangeUV);
            const f2 c1 = norm2((f2) { (float)u1, (float)v1 }, s.biasUV, s.rcpRangeUV);
            uv[r][0] = uv[r][1] = c0;
            uv[r][2] = uv[r][3] = c1;
        }
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        // 4-tap filter on normalised samples, src/reformat.c:834-837: ((closest*9/16 + horizontal*3/16) + vertical*3/16)
        // + diagonal*1/16, evaluated for Cb and Cr at once; every product equals the reference's product for that tap
        const f2 k9 = splat(9.0f / 16.0f), k3 = splat(3.0f / 16.0f), k1 = splat(1.0f / 16.0f);
        auto loadRow = [&](int q, f2 m[4]) {
            const f4 lo = *reinterpret_cast<const f4 *>(&sC[q][2 * tx]);
            const f4 hi = *reinterpret_cast<const f4 *>(&sC[q][2 * tx + 2]);
            m[0] = lo.xy, m[1] = lo.zw, m[2] = hi.xy, m[3] = hi.zw;
        };
        if constexpr (SUB == SUB_420) {
            f2 m[4], v[2][4];
            loadRow(ty + 1, m);
            loadRow(ty, v[0]);     // even luma rows: vertical neighbour above
            loadRow(ty + 2, v[1]); // odd luma rows: below
            const f2 m9b = m[1] * k9, m9c = m[2] * k9;
            const f2 m3a = m[0] * k3, m3b = m[1] * k3, m3c = m[2] * k3, m3d = m[3] * k3;
            const f2 h0 = m9b + m3a, h1 = m9b + m3c, h2 = m9c + m3b, h3 = m9c + m3d;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f2 v3b = v[r][1] * k3, v3c = v[r][2] * k3;
                const f2 v1a = v[r][0] * k1, v1b = v[r][1] * k1, v1c = v[r][2] * k1, v1d = v[r][3] * k1;
                uv[r][0] = (h0 + v3b) + v1a; // even pixel: horizontal neighbour on the left
                uv[r][1] = (h1 + v3b) + v1c; // odd pixel: on the right
                uv[r][2] = (h2 + v3c) + v1b;
                uv[r][3] = (h3 + v3c) + v1d;
            }
        } else { // 4:2:2: the vertical neighbour is the sample itself (src/reformat.c:784-786)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f2 m[4];
                loadRow(2 * ty + r, m);
                const f2 m9b = m[1] * k9, m9c = m[2] * k9;
                const f2 m3a = m[0] * k3, m3b = m[1] * k3, m3c = m[2] * k3, m3d = m[3] * k3;
                const f2 m1a = m[0] * k1, m1b = m[1] * k1, m1c = m[2] * k1, m1d = m[3] * k1;
                uv[r][0] = ((m9b + m3a) + m3b) + m1a;
                uv[r][1] = ((m9b + m3c) + m3b) + m1c;
                uv[r][2] = ((m9c + m3b) + m3c) + m1b;
                uv[r][3] = ((m9c + m3d) + m3c) + m1d;
            }
        }
    }

    // ---- per-pixel arithmetic and stores ----
    const bool swapRB = (o.offB < o.offR);
    const bool alphaFirst = (NCH == 4) && (o.offA == 0);
    const f2 cBR = { s.twoOneMinusKb, s.twoOneMinusKr }; // (Cb,Cr) -> (B - Y, R - Y), src/reformat.c:874-875
    const f2 cUV = { s.kbOneMinusKb, s.krOneMinusKr };   // the two products of the green term, :876
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (edge && (tileY + 2 * ty + r >= p.h))
            continue;
        float fy[4];
        samples4<YT>(rawY[r], yuvMax, fy);
        const f2 y01 = norm2((f2) { fy[0], fy[1] }, s.biasY, s.rcpRangeY);
        const f2 y23 = norm2((f2) { fy[2], fy[3] }, s.biasY, s.rcpRangeY);
        const float yk[4] = { y01.x, y01.y, y23.x, y23.y };
        f2 br[4]; // (B, R) per pixel
        f2 g01, g23;
        if constexpr (SUB == SUB_400) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                br[k] = splat(yk[k]);
            g01 = y01, g23 = y23;
        } else {
            float sum[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                br[k] = splat(yk[k]) + cBR * uv[r][k];
                const f2 pr = cUV * uv[r][k];
                sum[k] = pr.y + pr.x; // (kr(1-kr)*Cr) + (kb(1-kb)*Cb)
            }
            // G = Y - (2*sum)/kg with 2/kg in verified reciprocal form
            const f2 s01 = { sum[0], sum[1] }, s23 = { sum[2], sum[3] };
            g01 = y01 - fma2(s01, splat(s.rcpKgTimes2.hi), s01 * splat(s.rcpKgTimes2.lo));
            g23 = y23 - fma2(s23, splat(s.rcpKgTimes2.hi), s23 * splat(s.rcpKgTimes2.lo));
        }

        unsigned av[4], a[4];
        decode4<YT>(rawA[r], av);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = (unsigned)o.maxv;
            if (NCH == 4 && p.alphaSource == ALPHA_PLANE)
                a[k] = alphaFromPlane(p, av[k]);
        }
        uint8_t * dst = o.pixels + (size_t)(Y0 + r) * o.rowBytes + (size_t)X * (NCH * sizeof(RT));

        if constexpr (sizeof(RT) == 1 && !HASMUL) {
            // 8-bit outputs: t = 0.5f + c * 255, then truncate + saturate + pack in one instruction per channel
            const f2 half = splat(0.5f), mx = splat(o.maxf);
            f2 tbr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tbr[k] = half + (br[k] * mx);
            const f2 tg01 = half + (g01 * mx), tg23 = half + (g23 * mx);
            if constexpr (NCH == 4) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w[k] = a[k] << (8 * o.offA);
                packRgba8Row(w, tbr, tg01, tg23, (unsigned)o.offR, (unsigned)o.offG, (unsigned)o.offB);
                storeVec(reinterpret_cast<u4 *>(dst), (u4) { w[0], w[1], w[2], w[3] }, nt);
            } else {
                float x[4], g[4] = { tg01.x, tg01.y, tg23.x, tg23.y }, z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    x[k] = swapRB ? tbr[k].x : tbr[k].y;
                    z[k] = swapRB ? tbr[k].y : tbr[k].x;
                }
                unsigned w[3];
                packRgb8Row(w, x, g, z);
                unsigned * d = reinterpret_cast<unsigned *>(dst);
                d[0] = w[0], d[1] = w[1], d[2] = w[2];
            }
        } else {
            const float gk[4] = { g01.x, g01.y, g23.x, g23.y };
            PixelOut q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = finishPixel<HASMUL>(p, br[k].y, gk[k], br[k].x, needAlpha ? av[k] : 0u, a[k]);
            store4<RT, NCH>(dst, q, a, swapRB, alphaFirst, nt);
        }
    }
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL>
__global__ __launch_bounds__(256) void yuvToRgbTileKernel(YuvToRgbPlan p)
{
    __shared__ __attribute__((aligned(16))) f2 sC[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    runTile<YT, SUB, BILINEAR, RT, NCH, HASMUL>(p, gridDim.x, sC);
}

template <typename YT, int SUB, bool BILINEAR, typename RT, int NCH, bool HASMUL>
__global__ __launch_bounds__(256) void yuvToRgbTileBatchKernel(const YuvToRgbPlan * __restrict__ table)
{
    __shared__ __attribute__((aligned(16))) f2 sC[BILINEAR ? kChromaRowsMax : 1][kChromaPitch];
    __shared__ YuvToRgbPlan plan;
    {
        // one cooperative copy of the job descriptor into LDS keeps it out of per-lane registers
        const uint32_t * src = reinterpret_cast<const uint32_t *>(&table[blockIdx.z]);
        uint32_t * dst = reinterpret_cast<uint32_t *>(&plan);
        const int t = threadIdx.y * kLanesX + threadIdx.x;
        for (int k = t; k < (int)(sizeof(YuvToRgbPlan) / 4); k += 256)
            dst[k] = src[k];
    }
    __syncthreads();
    runTile<YT, SUB, BILINEAR, RT, NCH, HASMUL>(plan, gridDim.x, sC);
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool MUL>
hipError_t launchOne(const TileLaunch & L)
{
    const dim3 block(kLanesX, kLanesY);
    const dim3 grid(L.blocksPerJob, 1, L.count);
    if (L.table)
        hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, MUL>), grid, block, 0, L.stream, L.table);
    else
        hipLaunchKernelGGL((yuvToRgbTileKernel<YT, SUB, BIL, RT, NCH, MUL>), grid, block, 0, L.stream, *L.plan);
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL>
hipError_t launchRgbVariant(const TileKey & k, const TileLaunch & L)
{
#define AVIFHIP_RGB_CASE(RT, NCH) return k.hasMul ? launchOne<YT, SUB, BIL, RT, NCH, true>(L) : launchOne<YT, SUB, BIL, RT, NCH, false>(L)
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
hipError_t launchYuvVariant(const TileKey & k, const TileLaunch & L)
{
    switch (k.sub) {
        case SUB_444: return launchRgbVariant<YT, SUB_444, false>(k, L);
        case SUB_400: return launchRgbVariant<YT, SUB_400, false>(k, L);
        case SUB_422: return k.bilinear ? launchRgbVariant<YT, SUB_422, true>(k, L) : launchRgbVariant<YT, SUB_422, false>(k, L);
        default: return k.bilinear ? launchRgbVariant<YT, SUB_420, true>(k, L) : launchRgbVariant<YT, SUB_420, false>(k, L);
    }
}

} // namespace tile
} // namespace avifhip
