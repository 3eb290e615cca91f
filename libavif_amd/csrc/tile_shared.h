// tile_shared.h -- host-visible description of a tiled-kernel launch (shared by the dispatch TU kernels_tile.hip and the
// instantiation TUs).
#pragma once

#include <hip/hip_runtime.h>

#include <string.h>

#include "plan.h"

namespace avifhip {
namespace tile {

enum Subsampling : int { SUB_444 = 0, SUB_422 = 1, SUB_420 = 2, SUB_400 = 3 };

// Grid canvases in ONE launch (the seam-aware builds of the bilinear families: -DTILE_SEAMS, kernels_tile_inst.hip): where a job's chroma
// filter reaches over the edge of its chroma window, the sample comes from the neighbouring tile instead of being clamped -- what
// libavif gets by stitching the tiles into one canvas first (src/read.c:1823-1877) and filtering across the seams (src/reformat.c:766-816).
// Index 3 * v + h, v: 0 = the job's own rows, 1 = the tile above, 2 = the tile below; h: 0 = own columns, 1 = left, 2 = right.  Every
// pointer addresses CANVAS sample (0,0) of its tile's plane, virtually, like TileArgs::u / v (same pitch: the host checks), so one offset
// serves all nine.  Filled by linkTileBatchHalo (kernels_tile.hip); entries of neighbours that do not exist hold the job's own planes and
// are never read (TileArgs::haloSides stops the coordinates at the window there: the reference's border rule).  The kernels do not keep
// these in registers: every wave copies the table's nine entries into LDS when it starts (jobOf, tile_impl.h), and the staging lanes of a
// tile whose neighbourhood crosses a seam read their pair from there.
struct TileHalo
{
    struct Planes
    {
        const uint8_t *u, *v;
    } at[9];
};
enum HaloSides : uint32_t { HALO_ABOVE = 1, HALO_BELOW = 2, HALO_LEFT = 4, HALO_RIGHT = 8 };

// Everything a tiled kernel reads, distilled from a YuvToRgbPlan: small enough to live in scalar registers for the
// whole kernel (the full plan does not).  The kernel converts the w4 x h2 pixels at the rectangle origin, w4 a
// multiple of 4 and h2 a multiple of 2; the at most 3 columns / 1 row left over go to the universal kernel.
struct TileArgs
{
    const uint8_t * y; // luma / alpha planes and rgb: address of the rectangle's first sample
    const uint8_t * a;
    const uint8_t * u; // chroma planes: address of CANVAS sample (0,0) (border clamping needs canvas coordinates)
    const uint8_t * v;
    uint8_t * rgb;
    uint32_t yPitch, aPitch, uPitch, vPitch, rgbPitch;
    uint32_t w4, h2;
    int32_t cx0, cy0;  // chroma coordinates of the rectangle origin
    int32_t cxMin, cxMax, cyMin, cyMax; // chroma samples the job may read (inclusive): coordinates clamp into this window
    float biasY, biasUV;
    RcpHL rcpRangeY, rcpRangeUV, rcpKgTimes2, rcpYuvMax, rcpRgbMax;
    float cB, cR;      // 2(1-kb), 2(1-kr)
    float cU, cV;      // kb(1-kb), kr(1-kr)
    float rgbMaxF;
    uint32_t yuvMax, rgbMax;
    uint32_t slotR, slotG, slotB, slotA; // channel index inside a pixel
    // libyuv's "YVU trick" (src/reformat_libyuv.c:386-423), applied by both kernel families: `u` above addresses the plane
    // feeding the FIRST colour channel of a pixel (X: R for RGB orders, B for BGR orders), `v` the one feeding the third
    // (Z), and cB / cU (cR / cV) are the coefficients of that first (third) channel -- so no kernel selects channels per pixel
    uint32_t slotX, slotZ;
    uint32_t planesSwapped; // ... `u` addresses the image's V plane (host side: linkHalo hands the neighbours' planes over in the same order)
    int32_t alphaRescale;                // alpha plane depth differs from the rgb depth (src/alpha.c:84-103)
    // rgb->ignoreAlpha on a format with an alpha channel, fp32 arithmetic: the destination's alpha samples stay as they are (src/reformat.c:
    // 1449-1450 -- nothing writes them).  The kernels that take alpha from a plane serve it: a pixel's alpha is read from the destination
    // pixel itself and stored back with the colours (`a` then addresses the luma plane: loaded, not looked at)
    int32_t alphaKeep;
    float f16Mul;                        // half-float outputs (avifRGBImageToF16, src/reformat.c:1419-1443): the subnormal-trick multiplier, 0 = integer output
    int32_t inLoopMul, postMul;          // MulMode
    // which arithmetic the integer post-pass runs in (plan.h postMulFx): libyuv's ARGBAttenuate / ARGBUnattenuate, or the reference's own
    // fp32 form (src/alpha.c).  The two mix in a libyuv build: libyuv attenuates RGBA / BGRA only, whatever converted the pixels
    int32_t postMulFx;
    int32_t identityCopy;                // 8-bit full-range identity matrix: bytes are copied (src/reformat.c:1278-1309)
    int32_t identityMatrix;              // identity matrix otherwise: the planes are G, B, R on luma's scale (biasUV / rcpRangeUV hold luma's)
    // YCgCo (1: three adds on the normalised samples, src/reformat.c:853-858) and YCgCo-Re / -Ro (2: integer lifting from the luma code and
    // the chroma scaled back to codes, :859-871); `cgFirst`: the plane `u` above is Cg (B is the first colour channel), else `v` is
    int32_t ycgco, cgFirst;
    float yuvMaxF;
    uint32_t tuning;
    // fused crop / rotate / mirror (plan.h PixelMap): `rgb` is then the destination buffer's first pixel, and canvas pixel
    // (mapX0 + X, mapY0 + row) of the rectangle's pixel (X, row) goes where the map says.  Packed 16-bit kernels only.
    PixelMap map;
    int32_t mapX0, mapY0;
    // limited-range alpha planes (pre-1.0 files: avifImageLimitedToFullAlpha, src/read.c:6724-6764, per sample avifLimitedToFullY,
    // src/reformat.c:1778-1791): a' = clamp(((a - lo) * full + half) / range, 0, full), the division as multiply-high by
    // `magic` then >> `shift` (exact for every sample of the depth: distillArgs)
    struct AlphaLimited
    {
        int32_t on, lo, full, half;
        uint32_t magic, shift;
    } alphaLim;
    // ---- fixed-point (libyuv-arithmetic) kernels only: SURVEY.md appendix D.1-D.4 with the constants folded ----
    // libyuv's "YVU trick" (src/reformat_libyuv.c:386-423) is applied to the plane pointers: `u` above addresses the plane
    // feeding the FIRST colour byte of a pixel (X: R for RGB orders, B for BGR orders), `v` the one feeding the third (Z).
    // All terms are kept at 4x scale so that, after a clamp to [0, 65535], the result byte is byte 1 of the register
    // (clamp8(t >> 6) == clamp(4t, 0, 65535) >> 8) and v_perm_b32 packs it straight into place:
    //   y1 = (((y << yShl) | (y >> yShr)) * yMul) >> 16               (8-bit planes: (y * yMul8) >> 16, yMul8 = 0x0101 * yMul)
    //   X4 = (y1 << 2) + kX4 + cX4 * lo,  Z4 = (y1 << 2) + kZ4 + cZ4 * hi,  G4 = (y1 << 2) + kG4 - gLo4 * lo - gHi4 * hi
    //   with lo, hi = clamp8(upsampled chroma >> cShr) of the two planes, every plane sample first reduced by >> downshift
    struct Fx
    {
        uint32_t yShl, yShr, yMul, yMul8;
        int32_t kX4, kZ4, kG4;
        int32_t cX4, cZ4, gLo4, gHi4;
        uint32_t cShr, downshift;
        uint32_t selXG, selZA;   // v_perm_b32 selectors: byte 1 of (X4, G4) / byte 1 of Z4 and byte 0 of alpha into their slots
        uint32_t selXGm, selZAm; // the same for values already reduced to byte 0 (after attenuate / unattenuate)
        int32_t alphaMode;  // FxAlpha
        uint32_t alphaShift;
        // packed 16-bit kernels (tile_pk_impl.h): both halves of a word hold the same int16 --
        //   X = sat(pkCX * lo + y1), Z = sat(pkCZ * hi + y1), G = pkGHi * hi + (pkGLo * lo + y1), y1 = (y * yMul8 >> 16) + pkYb,
        // lo / hi = upsampled chroma - 128 of the planes `u` / `v`; then >> 6 and a clamp to a byte
        uint32_t pkYb, pkCX, pkCZ, pkGLo, pkGHi;
        // v_perm_b32 selectors placing (x0 g0 x1 g1) [second operand] and (z0 z1 a0 a1) [first operand] into pixel 0 / pixel 1 of a pair
        uint32_t pkSel0, pkSel1;
    } fx;
    // seam-aware builds (jobs that are tiles of one canvas): which sides have a neighbouring tile (HaloSides), and the neighbours' planes
    uint32_t haloSides;
    const TileHalo * haloRef; // where the TABLE holds this job's `halo`: set by the kernel in its private copy of the job (jobOf, tile_impl.h)
    TileHalo halo;
};


// Sequences (round 6): up to kSeqMaxFrames jobs that differ in their buffers only -- the frames of an image sequence, each large enough to
// fill the chip on its own -- converted by ONE launch of the single-image kernels: grid z = frame, the frame's five addresses taken from the
// kernel arguments (no device table, so no upload and no event between launches: avifhipImageYUVToRGBBatchAsync, api_batch.cpp).  A single
// image is a sequence of one.
constexpr uint32_t kSeqMaxFrames = 8;
struct SeqFrames
{
    struct Frame
    {
        const uint8_t *y, *a, *u, *v;
        uint8_t * rgb;
    } f[kSeqMaxFrames];
};
inline void seqSetFrame(SeqFrames & S, uint32_t k, const struct TileArgs & A);

// The tiled kernel's arguments for the w4 x h2 whole-group part of a plan's rectangle.
inline TileArgs distillArgs(const YuvToRgbPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    TileArgs A;
    memset(&A, 0, sizeof(A));
    A.y = s.plane[0] + (size_t)p.y0 * s.rowBytes[0] + (size_t)p.x0 * s.chanBytes;
    A.a = s.alpha ? s.alpha + (size_t)p.y0 * s.alphaRowBytes + (size_t)p.x0 * s.chanBytes : nullptr;
    const bool keepsAlpha = o.hasAlpha && !o.is565 && p.alphaSource == ALPHA_KEEP;
    A.u = s.plane[1];
    A.v = s.plane[2];
    A.rgb = o.map.on ? o.pixels : o.pixels + (size_t)p.y0 * o.rowBytes + (size_t)p.x0 * o.pixBytes;
    A.map = o.map;
    if (s.alphaLimited) {
        const int d = (int)s.depth;
        const uint32_t range = 219u << (d - 8);
        A.alphaLim.on = 1, A.alphaLim.lo = 16 << (d - 8), A.alphaLim.full = (1 << d) - 1, A.alphaLim.half = (int32_t)(range / 2);
        // n / range == mulhi(n, magic) >> shift for 0 <= n <= (full - lo) * full + half: magic = floor(2^(32 + shift) / range) + 1 exceeds
        // 2^(32 + shift) / range by less than 1, so the product's error stays below n / 2^32 ... n * range < 2^(32 + shift) suffices
        A.alphaLim.shift = (uint32_t)(d - 1); // 219 << (d - 8) < 2^d: magic < 2^32
        A.alphaLim.magic = (uint32_t)((((uint64_t)1 << (32 + A.alphaLim.shift)) / range) + 1);
    }
    A.mapX0 = (int32_t)p.x0, A.mapY0 = (int32_t)p.y0;
    A.yPitch = s.rowBytes[0], A.aPitch = s.alphaRowBytes, A.uPitch = s.rowBytes[1], A.vPitch = s.rowBytes[2], A.rgbPitch = o.rowBytes;
    A.w4 = p.w & ~3u;
    A.h2 = p.h & ~1u;
    const bool subX = s.hasColor && s.format != AVIF_PIXEL_FORMAT_YUV444;
    const bool subY = s.hasColor && s.format == AVIF_PIXEL_FORMAT_YUV420;
    A.cx0 = (int32_t)(subX ? p.x0 >> 1 : p.x0);
    A.cy0 = (int32_t)(subY ? p.y0 >> 1 : p.y0);
    A.cxMin = p.cwinX0, A.cxMax = p.cwinX1, A.cyMin = p.cwinY0, A.cyMax = p.cwinY1;
    A.biasY = s.biasY, A.biasUV = s.biasUV;
    A.rcpRangeY = s.rcpRangeY, A.rcpRangeUV = s.rcpRangeUV, A.rcpKgTimes2 = s.rcpKgTimes2, A.rcpYuvMax = s.rcpMax, A.rcpRgbMax = o.rcpMax;
    A.cB = s.twoOneMinusKb, A.cR = s.twoOneMinusKr, A.cU = s.kbOneMinusKb, A.cV = s.krOneMinusKr;
    A.rgbMaxF = o.maxf;
    A.yuvMax = (uint32_t)s.maxv, A.rgbMax = (uint32_t)o.maxv;
    A.slotR = (uint32_t)(o.offR / o.chanBytes), A.slotG = (uint32_t)(o.offG / o.chanBytes), A.slotB = (uint32_t)(o.offB / o.chanBytes);
    A.slotA = (uint32_t)(o.offA / o.chanBytes);
    const bool redFirstColour = A.slotR < A.slotB;
    A.slotX = redFirstColour ? A.slotR : A.slotB, A.slotZ = redFirstColour ? A.slotB : A.slotR;
    if (redFirstColour && p.arith != ARITH_LIBYUV) { // the fixed-point block below does its own swap
        const uint8_t * t = A.u;
        A.u = A.v, A.v = t;
        A.planesSwapped = 1;
        const uint32_t tp = A.uPitch;
        A.uPitch = A.vPitch, A.vPitch = tp;
        float tf = A.cB;
        A.cB = A.cR, A.cR = tf;
        tf = A.cU, A.cU = A.cV, A.cV = tf;
    }
    A.alphaRescale = (s.depth != o.depth) ? 1 : 0;
    if (keepsAlpha) {
        A.alphaKeep = 1, A.alphaRescale = 0, A.alphaLim.on = 0;
        A.a = A.y, A.aPitch = A.yPitch;
    }
    A.f16Mul = o.isFloat ? o.f16Multiplier : 0.0f;
    A.inLoopMul = p.inLoopMul, A.postMul = p.postMul, A.postMulFx = p.postMulFx;
    // (the 8-bit copy with an integer alpha (un)multiply behind it: the identity transform as arithmetic, which reproduces every code, then the
    //  post-pass of the kernels that carry alpha arithmetic -- the byte shuffle lives in the kernels without)
    const bool copyThenMul = p.identityCopy && p.postMul != MUL_NONE;
    A.identityCopy = (p.identityCopy && !copyThenMul) ? 1 : 0;
    A.identityMatrix = (p.arith != ARITH_LIBYUV && s.mode == MODE_IDENTITY && (!p.identityCopy || copyThenMul)) ? 1 : 0;
    if (A.identityMatrix)
        A.biasUV = s.biasY, A.rcpRangeUV = s.rcpRangeY; // src/reformat.c:587-589: identity reads chroma through luma's table
    A.ycgco = (p.arith == ARITH_LIBYUV) ? 0 : (s.mode == MODE_YCGCO ? 1 : ((s.mode == MODE_YCGCO_RE || s.mode == MODE_YCGCO_RO) ? 2 : 0));
    A.cgFirst = redFirstColour ? 0 : 1; // (the fp32 block above swapped the planes for red-first orders)
    A.yuvMaxF = (float)s.maxv;
    A.tuning = p.tuning;
    if (p.arith == ARITH_LIBYUV) {
        const FixedPointMatrix & m = p.fx;
        A.fx.yShl = (p.fxNative == 10) ? 6 : (p.fxNative == 12) ? 4 : 8;
        A.fx.yShr = (p.fxNative == 10) ? 4 : (p.fxNative == 12) ? 8 : 0;
        A.fx.yMul = (uint32_t)m.yg;
        A.fx.yMul8 = 0x0101u * (uint32_t)m.yg;
        const int kB = m.yb - 128 * m.ub, kR = m.yb - 128 * m.vr, kG = m.yb + 128 * (m.ug + m.vg);
        const bool redFirst = A.slotR < A.slotB;
        if (redFirst) { // X = R is fed by the V plane
            const uint8_t * t = A.u;
            A.u = A.v, A.v = t;
            A.planesSwapped = 1;
            const uint32_t tp = A.uPitch;
            A.uPitch = A.vPitch, A.vPitch = tp;
            A.fx.kX4 = 4 * kR, A.fx.cX4 = 4 * m.vr, A.fx.kZ4 = 4 * kB, A.fx.cZ4 = 4 * m.ub, A.fx.gLo4 = 4 * m.vg, A.fx.gHi4 = 4 * m.ug;
        } else {
            A.fx.kX4 = 4 * kB, A.fx.cX4 = 4 * m.ub, A.fx.kZ4 = 4 * kR, A.fx.cZ4 = 4 * m.vr, A.fx.gLo4 = 4 * m.ug, A.fx.gHi4 = 4 * m.vg;
        }
        A.fx.kG4 = 4 * kG;
        A.fx.cShr = (p.fxNative == 10) ? 2 : (p.fxNative == 12) ? 4 : 0;
        A.fx.downshift = (uint32_t)p.fxDownshift;
        // perm selectors: source bytes 4..7 = first operand, 0..3 = second operand, 12 = constant zero
        const uint32_t slotX = redFirst ? A.slotR : A.slotB, slotZ = redFirst ? A.slotB : A.slotR;
        auto place = [](uint32_t slotA_, uint32_t selA, uint32_t slotB_, uint32_t selB) {
            uint32_t sel = 0x0c0c0c0cu;
            sel = (sel & ~(0xffu << (8 * slotA_))) | (selA << (8 * slotA_));
            sel = (sel & ~(0xffu << (8 * slotB_))) | (selB << (8 * slotB_));
            return sel;
        };
        A.fx.selXG = place(slotX, 5, A.slotG, 1), A.fx.selXGm = place(slotX, 4, A.slotG, 0);
        if (o.hasAlpha) {
            A.fx.selZA = place(slotZ, 5, A.slotA, 0), A.fx.selZAm = place(slotZ, 4, A.slotA, 0);
        }
        A.fx.alphaMode = p.fxAlpha, A.fx.alphaShift = (uint32_t)p.fxAlphaShift;
        auto splat16 = [](int v) { return ((uint32_t)v & 0xffffu) * 0x00010001u; };
        A.fx.pkYb = splat16(m.yb);
        A.fx.pkCX = splat16(A.fx.cX4 / 4), A.fx.pkCZ = splat16(A.fx.cZ4 / 4);
        A.fx.pkGLo = splat16(-(A.fx.gLo4 / 4)), A.fx.pkGHi = splat16(-(A.fx.gHi4 / 4));
        const bool planeAlpha = o.hasAlpha && p.alphaSource == ALPHA_PLANE;
        // selector bytes: 0..3 = (x0 g0 x1 g1), 4..7 = (z0 z1 a0 a1), 12 = 0x00, 13 = 0xff
        uint32_t sel0 = 0x0c0c0c0cu, sel1 = 0x0c0c0c0cu;
        auto put = [](uint32_t sel, uint32_t slot, uint32_t v) { return (sel & ~(0xffu << (8 * slot))) | (v << (8 * slot)); };
        sel0 = put(sel0, slotX, 0), sel1 = put(sel1, slotX, 2);
        sel0 = put(sel0, A.slotG, 1), sel1 = put(sel1, A.slotG, 3);
        sel0 = put(sel0, slotZ, 4), sel1 = put(sel1, slotZ, 5);
        if (o.hasAlpha) {
            sel0 = put(sel0, A.slotA, planeAlpha ? 6u : 0x0du), sel1 = put(sel1, A.slotA, planeAlpha ? 7u : 0x0du);
        }
        A.fx.pkSel0 = sel0, A.fx.pkSel1 = sel1;
    }
    return A; // (no neighbours: haloSides = 0)
}

// Links a job of a batch to the eight tiles around its own (kernels.h TileNeighbours; index 3 * v + h as in TileHalo): `plane1` / `plane2` are
// the nine tiles' U / V planes, each addressing canvas sample (0,0) virtually.  distillArgs hands the planes to the kernels in the order of
// the pixel's colour channels (the "YVU trick"): the neighbours' follow the job's own.  Entries of absent neighbours hold the job's planes.
inline void linkHalo(TileArgs & T, const uint8_t * const plane1[9], const uint8_t * const plane2[9], bool above, bool below, bool left, bool right)
{
    const bool swapped = T.planesSwapped != 0; // (distillArgs' own decision: comparing pointers would misread a tile whose U and V planes are one buffer)
    const bool present[9] = { true, left, right, above, above && left, above && right, below, below && left, below && right };
    for (int d = 0; d < 9; ++d) {
        const uint8_t * p1 = present[d] ? plane1[d] : plane1[0];
        const uint8_t * p2 = present[d] ? plane2[d] : plane2[0];
        T.halo.at[d].u = swapped ? p2 : p1;
        T.halo.at[d].v = swapped ? p1 : p2;
    }
    T.haloSides = (above ? HALO_ABOVE : 0u) | (below ? HALO_BELOW : 0u) | (left ? HALO_LEFT : 0u) | (right ? HALO_RIGHT : 0u);
}

inline void seqSetFrame(SeqFrames & S, uint32_t k, const TileArgs & A)
{
    S.f[k].y = A.y, S.f[k].a = A.a, S.f[k].u = A.u, S.f[k].v = A.v, S.f[k].rgb = A.rgb;
}
inline SeqFrames seqOfOne(const TileArgs & A)
{
    SeqFrames S;
    for (uint32_t k = 0; k < kSeqMaxFrames; ++k)
        seqSetFrame(S, k, A);
    return S;
}
// two jobs that may share a sequence launch: everything but the five addresses agrees (the halo entries of an unlinked job are never read)
inline bool seqCompatible(const TileArgs & a, const TileArgs & b)
{
    TileArgs x = a, y = b;
    x.y = y.y = nullptr, x.a = y.a = nullptr, x.u = y.u = nullptr, x.v = y.v = nullptr, x.rgb = y.rgb = nullptr;
    x.haloRef = y.haloRef = nullptr;
    memset(&x.halo, 0, sizeof(x.halo)), memset(&y.halo, 0, sizeof(y.halo));
    return a.haloSides == 0 && b.haloSides == 0 && (a.a == nullptr) == (b.a == nullptr) && memcmp(&x, &y, sizeof(TileArgs)) == 0;
}

struct TileKey
{
    bool fixedPoint; // libyuv arithmetic (tile_fx_impl.h), 8-bit RGB outputs
    bool wideYuv;
    int sub;
    bool bilinear;
    bool wideRgb;
    int nch;
    bool alphaPlane; // alpha channel comes from the alpha plane (otherwise opaque / absent)
    bool hasMul;
    bool mapped;     // stores go through a PixelMap (fused crop / rotate / mirror)
    bool wideDownshift; // integer path on 16-bit containers: samples are reduced to 8 bits first (no high-bit-depth libyuv entry)
    int attenuate;      // integer path with libyuv's ARGBAttenuate (1) / ARGBUnattenuate (2) after the conversion: fused into the packed kernels
    bool gray;          // GRAY / GRAYA / AGRAY outputs: nch = 1 or 2, luma only
};

struct TileLaunch
{
    const TileArgs * args;  // single job (kernarg) ...
    const TileArgs * table; // ... or device table of `count` jobs
    // ... or a sequence: `args` with the addresses of `seqCount` frames (count stays 1: a frame's geometry is the single image's).  Families
    // without sequence kernels answer hipErrorNotSupported (kernels_tile.hip launchYuvToRgbTileSequence asks only those that have them)
    const SeqFrames * seq;
    uint32_t seqCount;
    uint32_t count;
    uint32_t blocksPerJob;  // workgroups covering the largest job: one per run of tiles (= bandsPerJob * runsPerJob)
    uint32_t bandsPerJob, runsPerJob;
    uint32_t canvasColumns; // batches: the jobs are the tiles of one canvas, row-major, this many per canvas row (tile_geom.h PkGeom::canvasColumns); 0 = no
    uint32_t stripsPerWave; // NS: 1 or 2 vertically consecutive 256x2 strips per wave (tile = 256 x 8*NS pixels)
    uint32_t tilesPerRun;   // vertically consecutive tiles one workgroup walks through (software-pipelined)
    // packed 16-bit kernels (tile_pk_impl.h): size of the largest job, and the tuning knobs (0 = automatic)
    uint32_t maxW4, maxH2;
    uint32_t pkStrips;      // strips (two luma rows) per wave: 2 or 4
    uint32_t wavesXLog2;    // waves of a workgroup side by side (1 << n), the rest stacked
    uint32_t chunkRows;     // tile rows per XCD chunk, 0 = plain raster order
    uint32_t shiftStrips;   // the tile grid starts this many strips ABOVE the rectangle (a multiple of the strips per wave; 0 but for quarter turns: launchSoloMapped)
    bool mapped;            // stores go through the jobs' PixelMap
    bool transposed;        // ... which turns rows into columns (quarter turns)
    int attenuate;          // TileKey::attenuate
    bool streamLoads;       // batches: the jobs' planes exceed what the Infinity Cache can hold -- luma / alpha rows as streaming loads
    bool solo;              // fp32 / 10-12-bit integer families: the wave-private kernels instead of the cooperative runs
    bool pkWide;            // 10-12-bit integer family without a post-pass: the packed 16-bit kernels (tile_pk_impl.h)
    bool wideDownshift;     // ... entered through the reduction to 8 bits (TileKey)
    bool seams;             // batches: jobs are tiles of one canvas with their neighbours linked (TileHalo) -- the seam-aware builds
    int alphaSel;           // fp32 kernels with pending alpha arithmetic: the one mode the job(s) ask for (computeTile MULSEL), 0 = all compiled in
    hipStream_t stream;
};


} // namespace tile
} // namespace avifhip
