// kernels_generic.hip -- the universal reformat kernels: every (format, depth, range, matrix,
// upsampling, alpha mode) combination libavif's reformat.c/alpha.c accept, one lane per output
// pixel (YUV->RGB, alpha passes) or per 2x2 block (RGB->YUV).  These are the correctness
// backbone; the tiled kernels in kernels_tile.hip take over the bandwidth-critical
// configurations and must agree with these bit for bit.
#include <hip/hip_runtime.h>

#include "exactdiv.h"
#include "kernels.h"
#include "pixel_fixed.h"
#include "pixel_generic.h"
#include "pixel_math.h"

namespace avifhip {

__global__ __launch_bounds__(256) void yuvToRgbGenericKernel(YuvToRgbPlan p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= p.w || j >= p.h)
        return;
    if (p.arith == ARITH_LIBYUV)
        yuvToRgbPixelFixed(p, p.x0 + i, p.y0 + j);
    else
        yuvToRgbPixel(p, p.x0 + i, p.y0 + j);
}

__global__ __launch_bounds__(256) void yuvToRgbGenericBatchKernel(const YuvToRgbPlan * __restrict__ table)
{
    const YuvToRgbPlan & p = table[blockIdx.z];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= p.w || j >= p.h)
        return;
    if (p.arith == ARITH_LIBYUV)
        yuvToRgbPixelFixed(p, p.x0 + i, p.y0 + j);
    else
        yuvToRgbPixel(p, p.x0 + i, p.y0 + j);
}

// --------------------------------------------------------------------------------------------
// Grid canvases: samples are fetched from the tile that holds them (src/read.c:1823-1877 would have copied them into one
// canvas first).  Used only for the few pixels next to tile seams.
struct GridReader
{
    const YuvSide & s;
    const GridGeometry & g;
    const GridTile * tiles; // the workgroup's copy in LDS
    // coordinate / tile size (canvas coordinates stay below 65536: exact)
    static __device__ __forceinline__ uint32_t divBy(uint32_t x, uint32_t d, uint32_t magic) { return d > 1 ? __umulhi(x, magic) : x; }
    __device__ __forceinline__ unsigned y(uint32_t x, uint32_t yy) const
    {
        const uint32_t tx = divBy(x, g.tileW, g.magicW), ty = divBy(yy, g.tileH, g.magicH);
        const GridTile & t = tiles[ty * g.columns + tx];
        return loadSample(t.plane[0], t.rowBytes[0], x - tx * g.tileW, yy - ty * g.tileH, s.chanBytes);
    }
    __device__ __forceinline__ unsigned c(int pl, uint32_t x, uint32_t yy) const
    {
        const uint32_t tx = divBy(x, g.tileCW, g.magicCW), ty = divBy(yy, g.tileCH, g.magicCH);
        const GridTile & t = tiles[ty * g.columns + tx];
        return loadSample(t.plane[pl], t.rowBytes[pl], x - tx * g.tileCW, yy - ty * g.tileCH, s.chanBytes);
    }
    __device__ __forceinline__ unsigned u(uint32_t x, uint32_t yy) const { return c(1, x, yy); }
    __device__ __forceinline__ unsigned v(uint32_t x, uint32_t yy) const { return c(2, x, yy); }
    __device__ __forceinline__ unsigned a(uint32_t x, uint32_t yy) const
    {
        const uint32_t tx = divBy(x, g.tileW, g.magicW), ty = divBy(yy, g.tileH, g.magicH);
        const GridTile & t = tiles[ty * g.columns + tx];
        const unsigned sa = loadSample(t.alpha, t.alphaRowBytes, x - tx * g.tileW, yy - ty * g.tileH, s.chanBytes);
        return s.alphaLimited ? limitedToFullAlpha(sa, (int)s.depth) : sa;
    }
};

// The two pixels either side of a seam look at the same four chroma positions (the co-sited sample of each and the other's: the neighbour
// every filter picks leans across the seam), so one lane converts the pair and fetches the quad once: a sample behind the tile table costs
// ~20 instructions and two dependent memory round trips, the selects that serve it from registers five.
template <class Base>
struct SeamPairReader
{
    const Base & base;
    uint32_t c0, c1, r0, r1; // chroma columns / rows of the quad
    unsigned qu[2][2], qv[2][2]; // [row][column]
    __device__ __forceinline__ unsigned y(uint32_t x, uint32_t yy) const { return base.y(x, yy); }
    __device__ __forceinline__ unsigned a(uint32_t x, uint32_t yy) const { return base.a(x, yy); }
    __device__ __forceinline__ unsigned pick(const unsigned (&q)[2][2], uint32_t x, uint32_t yy) const
    {
        const unsigned top = (x == c1) ? q[0][1] : q[0][0], bottom = (x == c1) ? q[1][1] : q[1][0];
        return (yy == r1) ? bottom : top;
    }
    __device__ __forceinline__ bool holds(uint32_t x, uint32_t yy) const { return (x == c0 || x == c1) && (yy == r0 || yy == r1); }
    __device__ __forceinline__ unsigned u(uint32_t x, uint32_t yy) const { return holds(x, yy) ? pick(qu, x, yy) : base.u(x, yy); }
    __device__ __forceinline__ unsigned v(uint32_t x, uint32_t yy) const { return holds(x, yy) ? pick(qv, x, yy) : base.v(x, yy); }
};

// one workgroup = 256 positions along one seam (vertical seams first); a lane = the two pixels either side of the seam at its position
template <bool LDS_TABLE>
__global__ __launch_bounds__(256) void yuvToRgbGridSeamKernel(YuvToRgbPlan p, GridGeometry g, const GridTile * __restrict__ tiles, uint32_t verticalSeams,
                                                              uint32_t blocksPerVertical, uint32_t blocksPerHorizontal)
{
    // the tile table (plane pointers and pitches of up to kSeamTiles tiles) is read ~10 times per pixel: once into LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char seamLds[];
    const GridTile * ldsTiles = tiles;
    if constexpr (LDS_TABLE) {
        ldsTiles = reinterpret_cast<const GridTile *>(seamLds);
        const uint32_t words = g.columns * g.rows * (uint32_t)(sizeof(GridTile) / 4);
        const uint32_t * src = reinterpret_cast<const uint32_t *>(tiles);
        uint32_t * dst = reinterpret_cast<uint32_t *>(seamLds);
        for (uint32_t k = threadIdx.x; k < words; k += blockDim.x)
            dst[k] = src[k];
        __syncthreads();
    }
    const uint32_t verticalBlocks = verticalSeams * blocksPerVertical;
    const bool vertical = blockIdx.x < verticalBlocks;
    uint32_t i0, j0, i1, j1;
    if (vertical) {
        const uint32_t seam = blockIdx.x / blocksPerVertical, t = (blockIdx.x - seam * blocksPerVertical) * blockDim.x + threadIdx.x;
        i0 = (seam + 1) * g.tileW - 1, i1 = i0 + 1, j0 = j1 = t;
    } else {
        const uint32_t b = blockIdx.x - verticalBlocks;
        const uint32_t seam = b / blocksPerHorizontal, t = (b - seam * blocksPerHorizontal) * blockDim.x + threadIdx.x;
        j0 = (seam + 1) * g.tileH - 1, j1 = j0 + 1, i0 = i1 = t;
    }
    if (i0 >= p.canvasW || j0 >= p.canvasH)
        return;
    const bool second = i1 < p.canvasW && j1 < p.canvasH; // (always, for seams inside the canvas)
    const GridReader rd { p.yuv, g, ldsTiles };
    // the quad: columns / rows of the first pixel's co-sited sample and of the one across the seam, held inside the canvas's planes
    const YuvSide & s = p.yuv;
    SeamPairReader<GridReader> pr { rd, 0, 0, 0, 0, {}, {} };
    const uint32_t ci = i0 >> s.shiftX, cj = j0 >> s.shiftY;
    auto lean = [](uint32_t full, uint32_t c, int lo, int hi) { return (uint32_t)clampInt((full & 1) ? (int)c + 1 : (int)c - 1, lo, hi); };
    pr.c0 = ci, pr.c1 = s.shiftX ? lean(i0, ci, p.cwinX0, p.cwinX1) : ci;
    pr.r0 = cj, pr.r1 = s.shiftY ? lean(j0, cj, p.cwinY0, p.cwinY1) : cj;
    pr.qu[0][0] = rd.u(pr.c0, pr.r0), pr.qu[0][1] = rd.u(pr.c1, pr.r0), pr.qu[1][0] = rd.u(pr.c0, pr.r1), pr.qu[1][1] = rd.u(pr.c1, pr.r1);
    pr.qv[0][0] = rd.v(pr.c0, pr.r0), pr.qv[0][1] = rd.v(pr.c1, pr.r0), pr.qv[1][0] = rd.v(pr.c0, pr.r1), pr.qv[1][1] = rd.v(pr.c1, pr.r1);
    if (p.arith == ARITH_LIBYUV) {
        yuvToRgbPixelFixedT(p, pr, i0, j0);
        if (second)
            yuvToRgbPixelFixedT(p, pr, i1, j1);
    } else {
        yuvToRgbPixelT(p, pr, i0, j0);
        if (second)
            yuvToRgbPixelT(p, pr, i1, j1);
    }
}

// ... and one lane per pixel (small grids: the kernel is a chain of memory round trips there, and a lane with one pixel has the shorter one --
// 7.0 us against 9.5 for the 82 thousand seam pixels of a 12-megapixel photograph; 17.2 against 14.3 for the 336 thousand of cfg5's canvas).
// blockIdx.y = seam line (2 per interior seam: vertical seams first), x = position along the line
template <bool LDS_TABLE>
__global__ __launch_bounds__(256) void yuvToRgbGridSeamPixelKernel(YuvToRgbPlan p, GridGeometry g, const GridTile * __restrict__ tiles, uint32_t verticalLines)
{
    // the tile table (plane pointers and pitches of up to kSeamTiles tiles) is read ~10 times per pixel: once into LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char seamLds[];
    const GridTile * ldsTiles = tiles;
    if constexpr (LDS_TABLE) {
        ldsTiles = reinterpret_cast<const GridTile *>(seamLds);
        const uint32_t words = g.columns * g.rows * (uint32_t)(sizeof(GridTile) / 4);
        const uint32_t * src = reinterpret_cast<const uint32_t *>(tiles);
        uint32_t * dst = reinterpret_cast<uint32_t *>(seamLds);
        for (uint32_t k = threadIdx.x; k < words; k += blockDim.x)
            dst[k] = src[k];
        __syncthreads();
    }
    const uint32_t line = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i, j;
    if (line < verticalLines) {
        i = (line / 2 + 1) * g.tileW - 1 + (line & 1);
        j = t;
    } else {
        const uint32_t l = line - verticalLines;
        j = (l / 2 + 1) * g.tileH - 1 + (l & 1);
        i = t;
    }
    if (i >= p.canvasW || j >= p.canvasH)
        return;
    const GridReader rd { p.yuv, g, ldsTiles };
    if (p.arith == ARITH_LIBYUV)
        yuvToRgbPixelFixedT(p, rd, i, j);
    else
        yuvToRgbPixelT(p, rd, i, j);
}

// --------------------------------------------------------------------------------------------
// RGB -> YUV, one lane per 2x2 block (edge blocks 1x2 / 2x1 / 1x1), src/reformat.c:295-470, with
// the alpha plane written in the same pass (src/reformat.c:545-569).
struct Yuvf
{
    float y, u, v;
};

__device__ __forceinline__ unsigned loadChannel(const uint8_t * px, int off, int chanBytes)
{
    return (chanBytes == 1) ? (unsigned)px[off] : (unsigned)*reinterpret_cast<const uint16_t *>(px + off);
}
__device__ __forceinline__ void storeSample(uint8_t * plane, uint32_t rowBytes, uint32_t x, uint32_t y, int chanBytes, int v)
{
    uint8_t * q = plane + (size_t)y * rowBytes + (size_t)x * chanBytes;
    if (chanBytes == 1)
        *q = (uint8_t)v;
    else
        *reinterpret_cast<uint16_t *>(q) = (uint16_t)v;
}

__device__ __forceinline__ float applyAlphaSrc(float c, float a, int mul) // src/reformat.c:333-357
{
    if (a == 0)
        return 0;
    if (a < 1.0f) {
        if (mul == MUL_MULTIPLY)
            return c * a;
        const float q = c / a;
        return (q < 1.0f) ? q : 1.0f;
    }
    return c;
}

__device__ __forceinline__ Yuvf rgbToYuvPixel(const RgbToYuvPlan & p, uint32_t i, uint32_t j)
{
    const RgbSide & o = p.rgb;
    const YuvSide & s = p.yuv;
    const uint8_t * px = o.pixels + (size_t)j * o.rowBytes + (size_t)i * o.pixBytes;
    float R = (float)loadChannel(px, o.offR, o.chanBytes) / o.maxf;
    float G = (float)loadChannel(px, o.offG, o.chanBytes) / o.maxf;
    float B = (float)loadChannel(px, o.offB, o.chanBytes) / o.maxf;
    if (p.mul != MUL_NONE) {
        const float a = (float)loadChannel(px, o.offA, o.chanBytes) / o.maxf;
        R = applyAlphaSrc(R, a, p.mul);
        G = applyAlphaSrc(G, a, p.mul);
        B = applyAlphaSrc(B, a, p.mul);
    }
    Yuvf out;
    if (s.mode == MODE_COEFF) { // :383-386
        const float Y = (s.kr * R) + (s.kg * G) + (s.kb * B);
        out.y = Y;
        out.u = (B - Y) / (2 * (1 - s.kb));
        out.v = (R - Y) / (2 * (1 - s.kr));
    } else if (s.mode == MODE_IDENTITY) { // :361-365
        out.y = G, out.u = B, out.v = R;
    } else if (s.mode == MODE_YCGCO) { // :366-370
        out.y = 0.5f * G + 0.25f * (R + B);
        out.u = 0.5f * G - 0.25f * (R + B);
        out.v = 0.5f * (R - B);
    } else { // YCgCo-Re / Ro, :371-381
        float t;
        t = R * o.maxf;
        const int Ri = (int)roundHalfUp((t < 0.0f) ? 0.0f : ((o.maxf < t) ? o.maxf : t));
        t = G * o.maxf;
        const int Gi = (int)roundHalfUp((t < 0.0f) ? 0.0f : ((o.maxf < t) ? o.maxf : t));
        t = B * o.maxf;
        const int Bi = (int)roundHalfUp((t < 0.0f) ? 0.0f : ((o.maxf < t) ? o.maxf : t));
        const int Co = Ri - Bi;
        const int tt = Bi + (Co >> 1);
        const int Cg = Gi - tt;
        out.y = (float)(tt + (Cg >> 1)) / s.rangeY;
        out.u = (float)Cg / s.rangeUV;
        out.v = (float)Co / s.rangeUV;
    }
    // luma (and 4:4:4 chroma) are stored right away, :389-406
    storeSample(s.plane[0], s.rowBytes[0], i, j, s.chanBytes, quantizeY(out.y, s));
    if (s.format == AVIF_PIXEL_FORMAT_YUV444) {
        storeSample(s.plane[1], s.rowBytes[1], i, j, s.chanBytes, quantizeUV(out.u, s));
        storeSample(s.plane[2], s.rowBytes[2], i, j, s.chanBytes, quantizeUV(out.v, s));
    }
    // alpha plane, src/alpha.c:9-149 via src/reformat.c:545-569
    if (p.alphaSource == ALPHA_FILL) {
        storeSample(s.alpha, s.alphaRowBytes, i, j, s.chanBytes, s.maxv);
    } else if (p.alphaSource == ALPHA_PLANE) {
        const unsigned sa = loadChannel(px, o.offA, o.chanBytes);
        const unsigned a = (o.depth == s.depth) ? sa : rescaleAlpha(sa, o.maxf, (float)s.maxv, s.maxv);
        storeSample(s.alpha, s.alphaRowBytes, i, j, s.chanBytes, (int)a);
    }
    return out;
}

__global__ __launch_bounds__(256) void rgbToYuvGenericKernel(RgbToYuvPlan p)
{
    const uint32_t bx = (p.rx0 >> 1) + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t by = (p.ry0 >> 1) + blockIdx.y * blockDim.y + threadIdx.y;
    const uint32_t oi = bx * 2, oj = by * 2;
    if (oi >= p.rx0 + p.rw || oj >= p.ry0 + p.rh)
        return;
    const YuvSide & s = p.yuv;
    const uint32_t bw = (oi + 1 >= p.width) ? 1 : 2;
    const uint32_t bh = (oj + 1 >= p.height) ? 1 : 2;

    // conversion order bJ-outer / bI-inner as in :306-307; blk[bI][bJ]
    Yuvf blk[2][2];
    for (uint32_t bJ = 0; bJ < bh; ++bJ)
        for (uint32_t bI = 0; bI < bw; ++bI)
            blk[bI][bJ] = rgbToYuvPixel(p, oi + bI, oj + bJ);

    if (s.format == AVIF_PIXEL_FORMAT_YUV420) { // :413-440
        float sumU = 0.0f, sumV = 0.0f;
        for (uint32_t bJ = 0; bJ < bh; ++bJ)
            for (uint32_t bI = 0; bI < bw; ++bI) {
                sumU += blk[bI][bJ].u;
                sumV += blk[bI][bJ].v;
            }
        const float total = (float)(bw * bh);
        storeSample(s.plane[1], s.rowBytes[1], bx, by, s.chanBytes, quantizeUV(sumU / total, s));
        storeSample(s.plane[2], s.rowBytes[2], bx, by, s.chanBytes, quantizeUV(sumV / total, s));
    } else if (s.format == AVIF_PIXEL_FORMAT_YUV422) { // :441-467
        for (uint32_t bJ = 0; bJ < bh; ++bJ) {
            float sumU = 0.0f, sumV = 0.0f;
            for (uint32_t bI = 0; bI < bw; ++bI) {
                sumU += blk[bI][bJ].u;
                sumV += blk[bI][bJ].v;
            }
            const float total = (float)bw;
            storeSample(s.plane[1], s.rowBytes[1], bx, oj + bJ, s.chanBytes, quantizeUV(sumU / total, s));
            storeSample(s.plane[2], s.rowBytes[2], bx, oj + bJ, s.chanBytes, quantizeUV(sumV / total, s));
        }
    }
}

// RGB -> YUV in libyuv's fixed point (8-bit BT.601, appendix D.5), one lane per 2x2 block: luma per pixel; the RGB of
// the block (4:2:0) or pair (4:2:2) is averaged per channel FIRST, with the last column / row replicated, and U, V come
// from the average.  The alpha plane is libavif's own pass (src/reformat.c:545-569), fused in.
__global__ __launch_bounds__(256) void rgbToYuvFixedKernel(RgbToYuvPlan p)
{
    const uint32_t bx = (p.rx0 >> 1) + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t by = (p.ry0 >> 1) + blockIdx.y * blockDim.y + threadIdx.y;
    const uint32_t oi = bx * 2, oj = by * 2;
    if (oi >= p.rx0 + p.rw || oj >= p.ry0 + p.rh)
        return;
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    const bool full = p.fxFullRange != 0;
    const uint32_t x1 = (oi + 1 < p.width) ? oi + 1 : oi, y1 = (oj + 1 < p.height) ? oj + 1 : oj;
    const Rgb8 c00 = fxLoadRgb(o, oi, oj), c10 = fxLoadRgb(o, x1, oj), c01 = fxLoadRgb(o, oi, y1), c11 = fxLoadRgb(o, x1, y1);
    const Rgb8 blk[2][2] = { { c00, c01 }, { c10, c11 } }; // [bI][bJ]
    const uint32_t bw = x1 - oi + 1, bh = y1 - oj + 1;
    for (uint32_t bJ = 0; bJ < bh; ++bJ) {
        for (uint32_t bI = 0; bI < bw; ++bI) {
            const uint32_t i = oi + bI, j = oj + bJ;
            s.plane[0][(size_t)j * s.rowBytes[0] + i] = (uint8_t)fxLuma(blk[bI][bJ], full);
            if (s.format == AVIF_PIXEL_FORMAT_YUV444) {
                s.plane[1][(size_t)j * s.rowBytes[1] + i] = (uint8_t)fxCb(blk[bI][bJ], full);
                s.plane[2][(size_t)j * s.rowBytes[2] + i] = (uint8_t)fxCr(blk[bI][bJ], full);
            }
            if (p.alphaSource == ALPHA_FILL) {
                s.alpha[(size_t)j * s.alphaRowBytes + i] = 255;
            } else if (p.alphaSource == ALPHA_PLANE) {
                s.alpha[(size_t)j * s.alphaRowBytes + i] = o.pixels[(size_t)j * o.rowBytes + (size_t)i * o.pixBytes + o.offA];
            }
        }
    }
    if (s.format == AVIF_PIXEL_FORMAT_YUV420) {
        Rgb8 m;
        m.r = (c00.r + c10.r + c01.r + c11.r + 2) >> 2;
        m.g = (c00.g + c10.g + c01.g + c11.g + 2) >> 2;
        m.b = (c00.b + c10.b + c01.b + c11.b + 2) >> 2;
        s.plane[1][(size_t)by * s.rowBytes[1] + bx] = (uint8_t)fxCb(m, full);
        s.plane[2][(size_t)by * s.rowBytes[2] + bx] = (uint8_t)fxCr(m, full);
    } else if (s.format == AVIF_PIXEL_FORMAT_YUV422) {
        for (uint32_t bJ = 0; bJ < bh; ++bJ) {
            const Rgb8 a = blk[0][bJ], b = blk[1][bJ];
            Rgb8 m;
            m.r = (a.r + b.r + 1) >> 1, m.g = (a.g + b.g + 1) >> 1, m.b = (a.b + b.b + 1) >> 1;
            s.plane[1][(size_t)(oj + bJ) * s.rowBytes[1] + bx] = (uint8_t)fxCb(m, full);
            s.plane[2][(size_t)(oj + bJ) * s.rowBytes[2] + bx] = (uint8_t)fxCr(m, full);
        }
    }
}

// gray source: Y from the gray channel, src/reformat.c:471-519 (chroma planes are filled separately)
__global__ __launch_bounds__(256) void grayToYuvGenericKernel(RgbToYuvPlan p)
{
    // (the region rx0 .. rx0 + rw of every row: the whole image, or the columns the tiled kernel leaves over)
    const uint32_t i = p.rx0 + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= p.rx0 + p.rw || i >= p.width || j >= p.height)
        return;
    const RgbSide & o = p.rgb;
    const YuvSide & s = p.yuv;
    const uint8_t * px = o.pixels + (size_t)j * o.rowBytes + (size_t)i * o.pixBytes;
    float g = (float)loadChannel(px, o.offGray, o.chanBytes) / o.maxf;
    if (p.mul != MUL_NONE) {
        const float a = (float)loadChannel(px, o.offA, o.chanBytes) / o.maxf;
        g = applyAlphaSrc(g, a, p.mul);
    }
    storeSample(s.plane[0], s.rowBytes[0], i, j, s.chanBytes, quantizeY(g, s));
    if (p.alphaSource == ALPHA_FILL) {
        storeSample(s.alpha, s.alphaRowBytes, i, j, s.chanBytes, s.maxv);
    } else if (p.alphaSource == ALPHA_PLANE) {
        const unsigned sa = loadChannel(px, o.offA, o.chanBytes);
        const unsigned a = (o.depth == s.depth) ? sa : rescaleAlpha(sa, o.maxf, (float)s.maxv, s.maxv);
        storeSample(s.alpha, s.alphaRowBytes, i, j, s.chanBytes, (int)a);
    }
}

// memset / avifMemset16 of the chroma planes to the half value, src/reformat.c:520-542
__global__ __launch_bounds__(256) void fillSamplesKernel(uint8_t * plane, size_t sampleCount, int chanBytes, unsigned value)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= sampleCount)
        return;
    if (chanBytes == 1)
        plane[k] = (uint8_t)value;
    else
        reinterpret_cast<uint16_t *>(plane)[k] = (uint16_t)value;
}

// --------------------------------------------------------------------------------------------
// in-place premultiply / unpremultiply, src/alpha.c:151-535
__global__ __launch_bounds__(256) void alphaMulGenericKernel(AlphaMulPlan p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= p.width || j >= p.height)
        return;
    const RgbSide & o = p.rgb;
    uint8_t * px = o.pixels + (size_t)j * o.rowBytes + (size_t)i * o.pixBytes;
    const unsigned a = loadChannel(px, o.offA, o.chanBytes);
    const int mode = p.unmultiply ? MUL_UNMULTIPLY : MUL_MULTIPLY;
    if (p.arith == ARITH_LIBYUV) { // ARGBAttenuate / ARGBUnattenuate: 8-bit RGBA / BGRA, every alpha value
        px[o.offR] = (uint8_t)fxAlphaMul(px[o.offR], a, mode);
        px[o.offG] = (uint8_t)fxAlphaMul(px[o.offG], a, mode);
        px[o.offB] = (uint8_t)fxAlphaMul(px[o.offB], a, mode);
        return;
    }
    if (a >= (unsigned)o.maxv)
        return; // opaque is a no-op
    const int nColour = o.isGray ? 1 : 3;
    const int offs[3] = { o.isGray ? o.offGray : o.offR, o.offG, o.offB };
    for (int k = 0; k < nColour; ++k) {
        const unsigned c = loadChannel(px, offs[k], o.chanBytes);
        const unsigned v = alphaMulInt(c, a, (unsigned)o.maxv, o.maxf, mode);
        if (o.chanBytes == 1)
            px[offs[k]] = (uint8_t)v;
        else
            *reinterpret_cast<uint16_t *>(px + offs[k]) = (uint16_t)v;
    }
}

// ... for 4-channel pixels in 16-byte aligned rows: lanes work on groups of 16 bytes (4 pixels of 8-bit channels, 2 of 16-bit ones), one
// load and one store per group instead of four byte loads and three byte stores per pixel.  Same results per channel; the alpha
// channel's bits pass through untouched.
// premultiply with "/ maxF" in the verified reciprocal form: floorf(c * a / maxF + 0.5f); the operand is never negative, so the truncating
// conversion is the floor; a == 0 needs no special case; a >= max leaves the channel untouched (src/alpha.c:180-192)
__device__ __forceinline__ unsigned premultiplyExact(unsigned c, unsigned a, unsigned maxv, RcpHL rcpMax)
{
    const unsigned m = (unsigned)(divExact((float)c * (float)a, rcpMax) + 0.5f);
    return (a >= maxv) ? c : m;
}

// which arithmetic an instantiation of the 16-byte kernel carries (one each: the five of them in one body made 12,000 instructions)
enum { AMUL_FX_MULTIPLY = 0, AMUL_FX_UNMULTIPLY, AMUL_EXACT_MULTIPLY, AMUL_IEEE_MULTIPLY, AMUL_IEEE_UNMULTIPLY, AMUL_RCP_UNMULTIPLY };

// what a pixel's alpha contributes to its three colour channels
struct AlphaOperand
{
    unsigned a;
    UnpremulRcp rcp; // AMUL_RCP_UNMULTIPLY: the pixel's exact reciprocal (pixel_math.h)
    unsigned ia;     // AMUL_FX_UNMULTIPLY: ARGBUnattenuate's 8.8 reciprocal of a (pixel_fixed.h: fxUnattenuateReciprocal)
};
template <int VARIANT>
__device__ __forceinline__ AlphaOperand alphaOperand(unsigned a)
{
    AlphaOperand A = { a, { 0.0f, 0.0f }, 0u };
    if constexpr (VARIANT == AMUL_RCP_UNMULTIPLY)
        A.rcp = unpremulRcp((float)(a ? a : 1u));
    if constexpr (VARIANT == AMUL_FX_UNMULTIPLY)
        A.ia = fxUnattenuateReciprocal(a);
    return A;
}

template <int VARIANT>
__device__ __forceinline__ unsigned alphaMulChannel(const RgbSide & o, unsigned c, const AlphaOperand & A, unsigned maxv, float maxf)
{
    if constexpr (VARIANT == AMUL_FX_MULTIPLY) {
        return fxAlphaMul(c, A.a, MUL_MULTIPLY);
    } else if constexpr (VARIANT == AMUL_FX_UNMULTIPLY) {
        return fxUnattenuateBy(c, A.ia); // the pixel's reciprocal formed once
    } else if constexpr (VARIANT == AMUL_EXACT_MULTIPLY) {
        return premultiplyExact(c, A.a, maxv, o.rcpMax);
    } else if constexpr (VARIANT == AMUL_RCP_UNMULTIPLY) {
        // (transparent and opaque pixels, which the reference answers with 0 / leaves alone, are settled per pixel by the caller: src/alpha.c:367-373)
        return min((unsigned)unpremulRcpArg((float)c, A.rcp, maxf), maxv);
    } else {
        return alphaMulInt(c, A.a, maxv, maxf, VARIANT == AMUL_IEEE_MULTIPLY ? MUL_MULTIPLY : MUL_UNMULTIPLY);
    }
}

// AMUL_RCP_UNMULTIPLY on the four 8-bit pixels of a group: bytes in and out without integer detours -- v_cvt_f32_ubyteN reads a channel
// straight from the pixel word, v_cvt_pk_u8_f32 under round-toward-zero (the floor of the non-negative argument, saturated at 255: the
// reference's min) writes it back in place; the alpha byte rides along in the word.
__device__ __forceinline__ void unpremultiplyGroup8(unsigned (&v)[4], bool alphaFirst)
{
    float t[4][3];
    unsigned keep[4], zero[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned w = v[q];
        const unsigned a = alphaFirst ? (w & 0xffu) : (w >> 24);
        const UnpremulRcp R = unpremulRcp(a ? (float)a : 1.0f);
        // the three colour bytes: 1 and 2 always, and the one at the other end from alpha
        t[q][0] = unpremulRcpArg((float)((w >> 8) & 0xffu), R, 255.0f);
        t[q][1] = unpremulRcpArg((float)((w >> 16) & 0xffu), R, 255.0f);
        t[q][2] = unpremulRcpArg(alphaFirst ? (float)(w >> 24) : (float)(w & 0xffu), R, 255.0f);
        keep[q] = (a == 255u) ? 1u : 0u; // opaque: left alone
        zero[q] = (a == 0u) ? 1u : 0u;   // transparent: colours 0
    }
    unsigned o[4] = { v[0], v[1], v[2], v[3] };
    const unsigned slotEnd = alphaFirst ? 3u : 0u;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %10, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %13, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %14, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, %16, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, %16, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, %16, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %15, %16, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])
                 : "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[2][0]), "v"(t[2][1]), "v"(t[2][2]),
                   "v"(t[3][0]), "v"(t[3][1]), "v"(t[3][2]), "s"(slotEnd));
    const unsigned alphaMask = alphaFirst ? 0x000000ffu : 0xff000000u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        v[q] = keep[q] ? v[q] : (zero[q] ? (v[q] & alphaMask) : o[q]);
}

// the N pixels of one 16-byte group; alpha is channel 0 (alphaFirst) or channel 3, channels 1 and 2 are colours either way
template <typename CT, int VARIANT>
__device__ __forceinline__ void alphaMulGroup(const AlphaMulPlan & p, unsigned (&v)[4], bool alphaFirst)
{
    const RgbSide & o = p.rgb;
    constexpr uint32_t N = (sizeof(CT) == 1) ? 4 : 2; // pixels per group
    if constexpr (sizeof(CT) == 1 && VARIANT == AMUL_RCP_UNMULTIPLY) {
        unpremultiplyGroup8(v, alphaFirst);
        return;
    }
#pragma unroll
    for (uint32_t q = 0; q < N; ++q) {
        if constexpr (sizeof(CT) == 1) {
            const unsigned w = v[q];
            const unsigned b0 = w & 0xffu, b3 = w >> 24;
            const AlphaOperand A = alphaOperand<VARIANT>(alphaFirst ? b0 : b3);
            const unsigned m1 = alphaMulChannel<VARIANT>(o, (w >> 8) & 0xffu, A, 255u, 255.0f);
            const unsigned m2 = alphaMulChannel<VARIANT>(o, (w >> 16) & 0xffu, A, 255u, 255.0f);
            const unsigned mx = alphaMulChannel<VARIANT>(o, alphaFirst ? b3 : b0, A, 255u, 255.0f); // the colour at the other end
            v[q] = (alphaFirst ? (b0 | (mx << 24)) : (mx | (b3 << 24))) | (m1 << 8) | (m2 << 16);
        } else {
            const unsigned lo = v[2 * q], hi = v[2 * q + 1];
            const unsigned c0 = lo & 0xffffu, c3 = hi >> 16;
            const unsigned a = alphaFirst ? c0 : c3;
            const AlphaOperand A = alphaOperand<VARIANT>(a);
            const unsigned m1 = alphaMulChannel<VARIANT>(o, lo >> 16, A, (unsigned)o.maxv, o.maxf);
            const unsigned m2 = alphaMulChannel<VARIANT>(o, hi & 0xffffu, A, (unsigned)o.maxv, o.maxf);
            const unsigned mx = alphaMulChannel<VARIANT>(o, alphaFirst ? c3 : c0, A, (unsigned)o.maxv, o.maxf);
            v[2 * q] = (alphaFirst ? c0 : mx) | (m1 << 16), v[2 * q + 1] = m2 | ((alphaFirst ? mx : c3) << 16);
            if constexpr (VARIANT == AMUL_RCP_UNMULTIPLY) {
                if (a >= (unsigned)o.maxv) // opaque: left alone, stray bits above the depth included
                    v[2 * q] = lo, v[2 * q + 1] = hi;
                else if (a == 0u) // transparent: colours 0
                    v[2 * q] = alphaFirst ? c0 : 0u, v[2 * q + 1] = alphaFirst ? 0u : (c3 << 16);
            }
        }
    }
}

// A lane owns kAlphaMulGroups groups of one row, 64 groups apart (every load and store instruction of the wave covers 1 KiB of the row),
// and requests all of them before the arithmetic starts: the divide sequences of the un-premultiply direction are long enough that
// one 16-byte request per lane left the memory system idle most of the time.
constexpr uint32_t kAlphaMulGroups = 4;

template <typename CT, int VARIANT>
__global__ __launch_bounds__(256) void alphaMulWideKernel(AlphaMulPlan p)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    constexpr uint32_t N = (sizeof(CT) == 1) ? 4 : 2; // pixels per group
    constexpr uint32_t G = kAlphaMulGroups;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= p.height)
        return;
    const RgbSide & o = p.rgb;
    uint8_t * row = o.pixels + (size_t)j * o.rowBytes;
    const uint32_t g0 = blockIdx.x * (64u * G) + threadIdx.x;
    const uint32_t slotA = (uint32_t)o.offA / sizeof(CT); // 0 or 3
    u4v in[G];
#pragma unroll
    for (uint32_t k = 0; k < G; ++k) {
        const uint32_t i = (g0 + 64u * k) * N;
        in[k] = u4v { 0u, 0u, 0u, 0u };
        if (i + N <= p.width)
            in[k] = *reinterpret_cast<const u4v *>(row + (size_t)(g0 + 64u * k) * 16u);
    }
#pragma unroll
    for (uint32_t k = 0; k < G; ++k) {
        const uint32_t i = (g0 + 64u * k) * N;
        uint8_t * px = row + (size_t)(g0 + 64u * k) * 16u;
        if (i + N <= p.width) {
            unsigned v[4] = { in[k].x, in[k].y, in[k].z, in[k].w };
            alphaMulGroup<CT, VARIANT>(p, v, slotA == 0);
            *reinterpret_cast<u4v *>(px) = u4v { v[0], v[1], v[2], v[3] };
        } else if (i < p.width) { // the row's last, partial group: the plain forms of the same results
            constexpr int kEdge = (VARIANT == AMUL_EXACT_MULTIPLY) ? (int)AMUL_IEEE_MULTIPLY : (VARIANT == AMUL_RCP_UNMULTIPLY) ? (int)AMUL_IEEE_UNMULTIPLY : VARIANT;
            for (uint32_t q = 0; i + q < p.width; ++q) {
                CT * c = reinterpret_cast<CT *>(px) + 4 * q;
                const unsigned a = c[slotA];
                for (uint32_t ch = 0; ch < 4; ++ch)
                    if (ch != slotA)
                        c[ch] = (VARIANT == AMUL_FX_UNMULTIPLY) ? (CT)fxAlphaMul(c[ch], a, MUL_UNMULTIPLY)
                                                                : (CT)alphaMulChannel<kEdge>(o, c[ch], alphaOperand<kEdge>(a), (unsigned)o.maxv, o.maxf);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// in-place integer -> half float pass, src/reformat.c:1419-1443
__global__ __launch_bounds__(256) void toF16GenericKernel(uint8_t * pixels, uint32_t rowBytes, uint32_t samplesPerRow, uint32_t rows, float multiplier)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= samplesPerRow || j >= rows)
        return;
    uint16_t * px = reinterpret_cast<uint16_t *>(pixels + (size_t)j * rowBytes) + i;
    *px = (uint16_t)toHalfBits(*px, multiplier);
}

// --------------------------------------------------------------------------------------------
// launchers
static inline dim3 gridFor(uint32_t w, uint32_t h, dim3 block, uint32_t z = 1)
{
    return dim3((w + block.x - 1) / block.x, (h + block.y - 1) / block.y, z);
}

hipError_t launchYuvToRgbGeneric(const YuvToRgbPlan & plan, hipStream_t stream)
{
    if (plan.w == 0 || plan.h == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    hipLaunchKernelGGL(yuvToRgbGenericKernel, gridFor(plan.w, plan.h, block), block, 0, stream, plan);
    return hipGetLastError();
}

hipError_t launchYuvToRgbGenericBatch(const YuvToRgbPlan * deviceTable, uint32_t count, uint32_t maxW, uint32_t maxH, hipStream_t stream)
{
    if (count == 0 || maxW == 0 || maxH == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    hipLaunchKernelGGL(yuvToRgbGenericBatchKernel, gridFor(maxW, maxH, block, count), block, 0, stream, deviceTable);
    return hipGetLastError();
}

hipError_t launchYuvToRgbGridSeams(const YuvToRgbPlan & canvasPlan, const GridGeometry & g, const GridTile * deviceTiles, bool vertical, bool horizontal,
                                   hipStream_t stream)
{
    const uint32_t vSeams = vertical ? g.columns - 1 : 0, hSeams = horizontal ? g.rows - 1 : 0;
    if (vSeams + hSeams == 0)
        return hipSuccess;
    const uint32_t bv = (canvasPlan.canvasH + 255) / 256, bh = (canvasPlan.canvasW + 255) / 256;
    const uint32_t blocks = vSeams * bv + hSeams * bh;
    GridGeometry gm = g;
    auto magic = [](uint32_t d) { return d > 1 ? (uint32_t)((((uint64_t)1 << 32) + d - 1) / d) : 0u; };
    gm.magicW = magic(g.tileW), gm.magicH = magic(g.tileH), gm.magicCW = magic(g.tileCW), gm.magicCH = magic(g.tileCH);
    const size_t ldsBytes = (size_t)g.columns * g.rows * sizeof(GridTile);
    const uint64_t seamPixels = 2ull * ((uint64_t)vSeams * canvasPlan.canvasH + (uint64_t)hSeams * canvasPlan.canvasW);
    bool pairs = seamPixels >= 200000;
    if (const char * e = getenv("AVIFHIP_SEAM_PAIRS")) // tests (small grids through the pair kernel) and A/B measurements
        pairs = atoi(e) != 0;
    if (!pairs && ldsBytes <= 48 * 1024) { // few seam pixels: one lane each (see yuvToRgbGridSeamPixelKernel)
        const uint32_t longest = canvasPlan.canvasW > canvasPlan.canvasH ? canvasPlan.canvasW : canvasPlan.canvasH;
        hipLaunchKernelGGL(yuvToRgbGridSeamPixelKernel<true>, dim3((longest + 255) / 256, 2 * (vSeams + hSeams)), dim3(256), ldsBytes, stream, canvasPlan, gm, deviceTiles, 2 * vSeams);
        return hipGetLastError();
    }
    if (ldsBytes <= 48 * 1024)
        hipLaunchKernelGGL(yuvToRgbGridSeamKernel<true>, dim3(blocks), dim3(256), ldsBytes, stream, canvasPlan, gm, deviceTiles, vSeams, bv, bh);
    else // more than a thousand tiles: the table stays in global memory
        hipLaunchKernelGGL(yuvToRgbGridSeamKernel<false>, dim3(blocks), dim3(256), 0, stream, canvasPlan, gm, deviceTiles, vSeams, bv, bh);
    return hipGetLastError();
}

// gray sources: the chroma planes, if any, are set to the half value over shiftedH * rowBytes bytes (padding included), src/reformat.c:520-542
hipError_t launchGrayChromaFill(const RgbToYuvPlan & plan, hipStream_t stream)
{
    const YuvSide & s = plan.yuv;
    const uint32_t shiftedH = (uint32_t)(((uint64_t)plan.height + s.shiftY) >> s.shiftY);
    const unsigned half = 1u << (s.depth - 1);
    for (int pl = 1; pl <= 2; ++pl) {
        if (!s.plane[pl])
            continue;
        const size_t samples = (size_t)shiftedH * s.rowBytes[pl] / (size_t)s.chanBytes;
        if (samples == 0)
            continue;
        const unsigned blocks = (unsigned)((samples + 255) / 256);
        hipLaunchKernelGGL(fillSamplesKernel, dim3(blocks), dim3(256), 0, stream, s.plane[pl], samples, s.chanBytes, half);
    }
    return hipGetLastError();
}

hipError_t launchRgbToYuvGeneric(const RgbToYuvPlan & plan, hipStream_t stream)
{
    if (plan.width == 0 || plan.height == 0 || plan.rw == 0 || plan.rh == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    if (plan.rgb.isGray) {
        hipLaunchKernelGGL(grayToYuvGenericKernel, gridFor(plan.rw, plan.height, block), block, 0, stream, plan);
        if (plan.rx0 != 0)
            return hipGetLastError(); // leftover columns of the tiled gray kernel: the chroma planes are the caller's
        return launchGrayChromaFill(plan, stream);
    } else {
        const uint32_t bw = (plan.rw + 1) / 2, bh = (plan.rh + 1) / 2;
        if (plan.arith == ARITH_LIBYUV)
            hipLaunchKernelGGL(rgbToYuvFixedKernel, gridFor(bw, bh, block), block, 0, stream, plan);
        else
            hipLaunchKernelGGL(rgbToYuvGenericKernel, gridFor(bw, bh, block), block, 0, stream, plan);
    }
    return hipGetLastError();
}

// ... 8 samples (16 bytes) per lane when the rows allow
__global__ __launch_bounds__(256) void toF16WideKernel(uint8_t * pixels, uint32_t rowBytes, uint32_t samplesPerRow, uint32_t rows, float multiplier)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 8, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= samplesPerRow || j >= rows)
        return;
    uint8_t * p = pixels + (size_t)j * rowBytes + (size_t)i * 2;
    if (i + 8 > samplesPerRow) {
        for (uint32_t k = 0; i + k < samplesPerRow; ++k)
            reinterpret_cast<uint16_t *>(p)[k] = (uint16_t)toHalfBits(reinterpret_cast<uint16_t *>(p)[k], multiplier);
        return;
    }
    u4v v = *reinterpret_cast<const u4v *>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        v[k] = toHalfBits(v[k] & 0xffffu, multiplier) | (toHalfBits(v[k] >> 16, multiplier) << 16);
    *reinterpret_cast<u4v *>(p) = v;
}

hipError_t launchToF16Generic(uint8_t * pixels, uint32_t rowBytes, uint32_t samplesPerRow, uint32_t rows, float multiplier, hipStream_t stream)
{
    if (samplesPerRow == 0 || rows == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    if (((uintptr_t)pixels % 16) == 0 && (rowBytes % 16) == 0) {
        hipLaunchKernelGGL(toF16WideKernel, gridFor((samplesPerRow + 7) / 8, rows, block), block, 0, stream, pixels, rowBytes, samplesPerRow, rows, multiplier);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(toF16GenericKernel, gridFor(samplesPerRow, rows, block), block, 0, stream, pixels, rowBytes, samplesPerRow, rows, multiplier);
    return hipGetLastError();
}

hipError_t launchAlphaMulGeneric(const AlphaMulPlan & plan, hipStream_t stream)
{
    if (plan.width == 0 || plan.height == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    const RgbSide & o = plan.rgb;
    const bool fourChannels = !o.isGray && !o.is565 && o.hasAlpha && o.pixBytes == 4 * o.chanBytes;
    if (fourChannels && ((uintptr_t)o.pixels % 16) == 0 && (o.rowBytes % 16) == 0) {
        const uint32_t groups = (o.chanBytes == 1) ? (plan.width + 3) / 4 : (plan.width + 1) / 2;
        const dim3 grid = gridFor((groups + kAlphaMulGroups - 1) / kAlphaMulGroups, plan.height, block);
        const int variant = (plan.arith == ARITH_LIBYUV) ? (plan.unmultiply ? AMUL_FX_UNMULTIPLY : AMUL_FX_MULTIPLY)
                            : plan.unmultiply            ? (plan.exactDiv ? AMUL_RCP_UNMULTIPLY : AMUL_IEEE_UNMULTIPLY)
                            : plan.exactDiv              ? AMUL_EXACT_MULTIPLY
                                                         : AMUL_IEEE_MULTIPLY;
        if (o.chanBytes == 1) {
            switch (variant) {
                case AMUL_FX_MULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_FX_MULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_FX_UNMULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_FX_UNMULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_EXACT_MULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_EXACT_MULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_IEEE_MULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_IEEE_MULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_RCP_UNMULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_RCP_UNMULTIPLY>), grid, block, 0, stream, plan); break;
                default: hipLaunchKernelGGL((alphaMulWideKernel<uint8_t, AMUL_IEEE_UNMULTIPLY>), grid, block, 0, stream, plan); break;
            }
        } else {
            switch (variant) { // (the fixed-point arithmetic is 8-bit only: plan.cpp)
                case AMUL_EXACT_MULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint16_t, AMUL_EXACT_MULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_IEEE_MULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint16_t, AMUL_IEEE_MULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_IEEE_UNMULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint16_t, AMUL_IEEE_UNMULTIPLY>), grid, block, 0, stream, plan); break;
                case AMUL_RCP_UNMULTIPLY: hipLaunchKernelGGL((alphaMulWideKernel<uint16_t, AMUL_RCP_UNMULTIPLY>), grid, block, 0, stream, plan); break;
                default: hipLaunchKernelGGL(alphaMulGenericKernel, gridFor(plan.width, plan.height, block), block, 0, stream, plan); break;
            }
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(alphaMulGenericKernel, gridFor(plan.width, plan.height, block), block, 0, stream, plan);
    return hipGetLastError();
}

} // namespace avifhip
