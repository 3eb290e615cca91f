// tile_shared.h -- host-visible description of a tiled-kernel launch (shared by the dispatch TU kernels_tile.hip and the
// instantiation TUs).
#pragma once

#include <hip/hip_runtime.h>

#include <string.h>

#include "plan.h"

namespace avifhip {
namespace tile {

enum Subsampling : int { SUB_444 = 0, SUB_422 = 1, SUB_420 = 2, SUB_400 = 3 };

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
    int32_t cw, ch;    // chroma plane size of the canvas
    float biasY, biasUV;
    RcpHL rcpRangeY, rcpRangeUV, rcpKgTimes2, rcpYuvMax, rcpRgbMax;
    float cB, cR;      // 2(1-kb), 2(1-kr)
    float cU, cV;      // kb(1-kb), kr(1-kr)
    float rgbMaxF;
    uint32_t yuvMax, rgbMax;
    uint32_t slotR, slotG, slotB, slotA; // channel index inside a pixel
    int32_t alphaRescale;                // alpha plane depth differs from the rgb depth (src/alpha.c:84-103)
    int32_t inLoopMul, postMul;          // MulMode
    uint32_t tuning;
};


// The tiled kernel's arguments for the w4 x h2 whole-group part of a plan's rectangle.
inline TileArgs distillArgs(const YuvToRgbPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    TileArgs A;
    memset(&A, 0, sizeof(A));
    A.y = s.plane[0] + (size_t)p.y0 * s.rowBytes[0] + (size_t)p.x0 * s.chanBytes;
    A.a = s.alpha ? s.alpha + (size_t)p.y0 * s.alphaRowBytes + (size_t)p.x0 * s.chanBytes : nullptr;
    A.u = s.plane[1];
    A.v = s.plane[2];
    A.rgb = o.pixels + (size_t)p.y0 * o.rowBytes + (size_t)p.x0 * o.pixBytes;
    A.yPitch = s.rowBytes[0], A.aPitch = s.alphaRowBytes, A.uPitch = s.rowBytes[1], A.vPitch = s.rowBytes[2], A.rgbPitch = o.rowBytes;
    A.w4 = p.w & ~3u;
    A.h2 = p.h & ~1u;
    const bool subX = s.hasColor && s.format != AVIF_PIXEL_FORMAT_YUV444;
    const bool subY = s.hasColor && s.format == AVIF_PIXEL_FORMAT_YUV420;
    A.cx0 = (int32_t)(subX ? p.x0 >> 1 : p.x0);
    A.cy0 = (int32_t)(subY ? p.y0 >> 1 : p.y0);
    A.cw = (int32_t)(subX ? (p.canvasW + 1) >> 1 : p.canvasW);
    A.ch = (int32_t)(subY ? (p.canvasH + 1) >> 1 : p.canvasH);
    A.biasY = s.biasY, A.biasUV = s.biasUV;
    A.rcpRangeY = s.rcpRangeY, A.rcpRangeUV = s.rcpRangeUV, A.rcpKgTimes2 = s.rcpKgTimes2, A.rcpYuvMax = s.rcpMax, A.rcpRgbMax = o.rcpMax;
    A.cB = s.twoOneMinusKb, A.cR = s.twoOneMinusKr, A.cU = s.kbOneMinusKb, A.cV = s.krOneMinusKr;
    A.rgbMaxF = o.maxf;
    A.yuvMax = (uint32_t)s.maxv, A.rgbMax = (uint32_t)o.maxv;
    A.slotR = (uint32_t)(o.offR / o.chanBytes), A.slotG = (uint32_t)(o.offG / o.chanBytes), A.slotB = (uint32_t)(o.offB / o.chanBytes);
    A.slotA = (uint32_t)(o.offA / o.chanBytes);
    A.alphaRescale = (s.depth != o.depth) ? 1 : 0;
    A.inLoopMul = p.inLoopMul, A.postMul = p.postMul;
    A.tuning = p.tuning;
    return A;
}

struct TileKey
{
    bool wideYuv;
    int sub;
    bool bilinear;
    bool wideRgb;
    int nch;
    bool alphaPlane; // alpha channel comes from the alpha plane (otherwise opaque / absent)
    bool hasMul;
};

struct TileLaunch
{
    const TileArgs * args;  // single job (kernarg) ...
    const TileArgs * table; // ... or device table of `count` jobs
    uint32_t count;
    uint32_t blocksPerJob;  // workgroups covering the largest job: one per run of tiles
    uint32_t stripsPerWave; // NS: 1 or 2 vertically consecutive 256x2 strips per wave (tile = 256 x 8*NS pixels)
    uint32_t tilesPerRun;   // vertically consecutive tiles one workgroup walks through (software-pipelined)
    hipStream_t stream;
};

hipError_t launchTileU8(const TileKey & key, const TileLaunch & launch);
hipError_t launchTileU16(const TileKey & key, const TileLaunch & launch);

} // namespace tile
} // namespace avifhip
