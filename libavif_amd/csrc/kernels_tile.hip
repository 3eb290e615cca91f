// kernels_tile.hip -- host-side dispatch of the bandwidth-tuned tiled kernels (tile_impl.h): which plans they
// cover and which instantiation serves a plan (one workgroup per 256x8 tile).
#include <hip/hip_runtime.h>

#include <stdio.h>

#include "kernels.h"
#include "tile_shared.h"

namespace avifhip {

using namespace tile;

namespace {

constexpr uint32_t kTileW = 256, kTileH = 8;

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
    return k;
}

bool aligned(const void * ptr, uint32_t rowBytes, uint32_t a)
{
    return ((uintptr_t)ptr % a) == 0 && (rowBytes % a) == 0;
}

const char * kernelNameFor(const TileKey & k)
{
    static thread_local char name[112];
    static const char * subs[] = { "444", "422", "420", "400" };
    snprintf(name, sizeof(name), "yuv2rgb_tile<%s,%s,%s,%s%d%s>", k.wideYuv ? "u16" : "u8", subs[k.sub], k.bilinear ? "bilinear" : "nearest",
             k.nch == 4 ? "rgba" : "rgb", k.wideRgb ? 16 : 8, k.hasMul ? ",alphamul" : "");
    return name;
}

uint32_t tilesOf(uint32_t w, uint32_t h)
{
    return ((w + kTileW - 1) / kTileW) * ((h + kTileH - 1) / kTileH);
}

} // namespace

bool tileYuvToRgbSupported(const YuvToRgbPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    if (p.arith != ARITH_FLOAT || p.identityCopy || s.mode != MODE_COEFF)
        return false;
    if (!s.exactDiv)
        return false; // a divisor off the verified list (exactdiv.h): the universal kernel divides the IEEE way
    if (o.isGray || o.is565 || o.isFloat)
        return false;
    if (o.hasAlpha && p.alphaSource == ALPHA_KEEP)
        return false; // destination alpha bytes must stay untouched: per-channel stores only
    if (p.w < 64 || p.h < 2)
        return false; // too small to fill even one row of lanes: the per-pixel kernel is the better fit
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
           ((k.hasMul ? 1 : 0) << 6);
}

hipError_t launchYuvToRgbTile(const YuvToRgbPlan & plan, hipStream_t stream, const char ** kernelName)
{
    const TileKey k = keyFor(plan);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    TileLaunch L;
    L.plan = &plan;
    L.table = nullptr;
    L.count = 1;
    L.blocksPerJob = tilesOf(plan.w, plan.h);
    L.stream = stream;
    return k.wideYuv ? launchTileU16(k, L) : launchTileU8(k, L);
}

hipError_t launchYuvToRgbTileBatch(const YuvToRgbPlan * deviceTable, const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW,
                                   uint32_t maxH, hipStream_t stream, const char ** kernelName)
{
    const TileKey k = keyFor(representative);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    TileLaunch L;
    L.plan = nullptr;
    L.table = deviceTable;
    L.count = count;
    L.blocksPerJob = tilesOf(maxW, maxH);
    L.stream = stream;
    return k.wideYuv ? launchTileU16(k, L) : launchTileU8(k, L);
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
