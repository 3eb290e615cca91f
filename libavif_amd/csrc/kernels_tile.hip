// kernels_tile.hip -- host-side dispatch of the bandwidth-tuned tiled kernels (tile_impl.h): which plans they
// cover, which instantiation serves a plan, how the image is cut into tiles, and the hand-over of the at most
// 3 columns / 1 row that do not fill a 4x2 pixel group to the universal kernel.
#include <hip/hip_runtime.h>

#include <stdio.h>

#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "tile_shared.h"

namespace avifhip {

using namespace tile;

namespace tile {
#define AVIFHIP_DECLARE_TILE(name) hipError_t launchTile_##name(const TileKey &, const TileLaunch &);
AVIFHIP_DECLARE_TILE(u8_444n) AVIFHIP_DECLARE_TILE(u8_400n) AVIFHIP_DECLARE_TILE(u8_422n) AVIFHIP_DECLARE_TILE(u8_422b)
AVIFHIP_DECLARE_TILE(u8_420n) AVIFHIP_DECLARE_TILE(u8_420b) AVIFHIP_DECLARE_TILE(u16_444n) AVIFHIP_DECLARE_TILE(u16_400n)
AVIFHIP_DECLARE_TILE(u16_422n) AVIFHIP_DECLARE_TILE(u16_422b) AVIFHIP_DECLARE_TILE(u16_420n) AVIFHIP_DECLARE_TILE(u16_420b)
#undef AVIFHIP_DECLARE_TILE
#define AVIFHIP_DECLARE_TILE(name) hipError_t launchTileFx_##name(const TileKey &, const TileLaunch &);
AVIFHIP_DECLARE_TILE(u8_444n) AVIFHIP_DECLARE_TILE(u8_400n) AVIFHIP_DECLARE_TILE(u8_422n) AVIFHIP_DECLARE_TILE(u8_422b)
AVIFHIP_DECLARE_TILE(u8_420n) AVIFHIP_DECLARE_TILE(u8_420b) AVIFHIP_DECLARE_TILE(u16_444n) AVIFHIP_DECLARE_TILE(u16_400n)
AVIFHIP_DECLARE_TILE(u16_422n) AVIFHIP_DECLARE_TILE(u16_422b) AVIFHIP_DECLARE_TILE(u16_420n) AVIFHIP_DECLARE_TILE(u16_420b)
#undef AVIFHIP_DECLARE_TILE
// the seam-aware builds of the families that filter chroma (batch kernels only: tile_impl.h)
#define AVIFHIP_DECLARE_TILE(name) hipError_t launchTileSeams_##name(const TileKey &, const TileLaunch &); hipError_t launchTileFxSeams_##name(const TileKey &, const TileLaunch &);
AVIFHIP_DECLARE_TILE(u8_422b) AVIFHIP_DECLARE_TILE(u8_420b) AVIFHIP_DECLARE_TILE(u16_422b) AVIFHIP_DECLARE_TILE(u16_420b)
#undef AVIFHIP_DECLARE_TILE
} // namespace tile

namespace {

constexpr uint32_t kBandW = 256, kTargetBlocks = 2048;

TileKey keyFor(const YuvToRgbPlan & p)
{
    TileKey k;
    k.fixedPoint = p.arith == ARITH_LIBYUV;
    k.wideYuv = p.yuv.chanBytes == 2;
    if (!p.yuv.hasColor)
        k.sub = SUB_400;
    else if (p.yuv.format == AVIF_PIXEL_FORMAT_YUV444)
        k.sub = SUB_444;
    else if (p.yuv.format == AVIF_PIXEL_FORMAT_YUV422)
        k.sub = SUB_422;
    else
        k.sub = SUB_420;
    k.gray = p.rgb.isGray != 0;
    if (k.gray)
        k.sub = SUB_400; // gray = clamp01(Y): chroma is never read (src/reformat.c:886-892)
    k.bilinear = p.bilinear && (k.sub == SUB_420 || k.sub == SUB_422);
    k.wideRgb = p.rgb.chanBytes == 2;
    k.nch = k.gray ? (p.rgb.hasAlpha ? 2 : 1) : (p.rgb.is565 ? 2 : (p.rgb.hasAlpha ? 4 : 3));
    k.hasMul = (p.inLoopMul != MUL_NONE) || (p.postMul != MUL_NONE);
    // (ALPHA_KEEP -- rgb->ignoreAlpha -- runs the same kernels: the "plane" is the destination's own alpha channel, TileArgs::alphaKeep)
    k.alphaPlane = p.rgb.hasAlpha && !p.rgb.is565 && (p.alphaSource == ALPHA_PLANE || p.alphaSource == ALPHA_KEEP);
    k.mapped = p.rgb.map.on != 0;
    k.wideDownshift = k.fixedPoint && k.wideYuv && p.fxDownshift != 0;
    k.attenuate = (k.fixedPoint && p.postMul != MUL_NONE && p.postMulFx && k.nch == 4 && k.alphaPlane && (p.tuning & TUNE_COOPERATIVE) == 0) ? (p.postMul == MUL_MULTIPLY ? 1 : 2) : 0;
    return k;
}

bool aligned(const void * ptr, uint32_t rowBytes, uint32_t a)
{
    return ((uintptr_t)ptr % a) == 0 && (rowBytes % a) == 0;
}

const char * kernelNameFor(const TileKey & k, uint32_t tuning)
{
    static thread_local char name[112];
    static const char * subs[] = { "444", "422", "420", "400" };
    // ",pk16": the packed 16-bit kernels (tile_pk_impl.h) serve 8-bit planes of the integer path unless a post-pass follows
    const bool packed = k.fixedPoint && (!k.hasMul || (k.attenuate && !k.mapped)) && (!k.wideYuv || (tuning & TUNE_COOPERATIVE) == 0);
    snprintf(name, sizeof(name), "%s<%s,%s,%s,%s%d%s%s%s%s>", k.fixedPoint ? "yuv2rgb_fixed_tile" : "yuv2rgb_tile", k.wideYuv ? "u16" : "u8", subs[k.sub], k.bilinear ? "bilinear" : "nearest",
             k.gray ? (k.nch == 2 ? "graya" : "gray") : (k.nch == 4 ? "rgba" : (k.nch == 2 ? "rgb565_" : "rgb")), k.wideRgb ? 16 : 8, k.alphaPlane ? ",alpha" : "", k.hasMul ? ",alphamul" : "", packed ? ",pk16" : "", k.mapped ? ",mapped" : "");
    return name;
}

hipError_t launchFamily(const TileKey & k, const TileLaunch & L)
{
    if (L.seams && L.table && k.bilinear) { // tiles of one canvas, neighbours linked: chroma is filtered across the seams in this launch
        const bool s420 = k.sub == SUB_420;
        if (k.fixedPoint) {
            if (!k.wideYuv)
                return s420 ? launchTileFxSeams_u8_420b(k, L) : launchTileFxSeams_u8_422b(k, L);
            return s420 ? launchTileFxSeams_u16_420b(k, L) : launchTileFxSeams_u16_422b(k, L);
        }
        if (!k.wideYuv)
            return s420 ? launchTileSeams_u8_420b(k, L) : launchTileSeams_u8_422b(k, L);
        return s420 ? launchTileSeams_u16_420b(k, L) : launchTileSeams_u16_422b(k, L);
    }
    if (k.fixedPoint) {
        if (!k.wideYuv) {
            switch (k.sub) {
                case SUB_444: return launchTileFx_u8_444n(k, L);
                case SUB_400: return launchTileFx_u8_400n(k, L);
                case SUB_422: return k.bilinear ? launchTileFx_u8_422b(k, L) : launchTileFx_u8_422n(k, L);
                default: return k.bilinear ? launchTileFx_u8_420b(k, L) : launchTileFx_u8_420n(k, L);
            }
        }
        switch (k.sub) {
            case SUB_444: return launchTileFx_u16_444n(k, L);
            case SUB_400: return launchTileFx_u16_400n(k, L);
            case SUB_422: return k.bilinear ? launchTileFx_u16_422b(k, L) : launchTileFx_u16_422n(k, L);
            default: return k.bilinear ? launchTileFx_u16_420b(k, L) : launchTileFx_u16_420n(k, L);
        }
    }
    if (!k.wideYuv) {
        switch (k.sub) {
            case SUB_444: return launchTile_u8_444n(k, L);
            case SUB_400: return launchTile_u8_400n(k, L);
            case SUB_422: return k.bilinear ? launchTile_u8_422b(k, L) : launchTile_u8_422n(k, L);
            default: return k.bilinear ? launchTile_u8_420b(k, L) : launchTile_u8_420n(k, L);
        }
    }
    switch (k.sub) {
        case SUB_444: return launchTile_u16_444n(k, L);
        case SUB_400: return launchTile_u16_400n(k, L);
        case SUB_422: return k.bilinear ? launchTile_u16_422b(k, L) : launchTile_u16_422n(k, L);
        default: return k.bilinear ? launchTile_u16_420b(k, L) : launchTile_u16_420n(k, L);
    }
}

// Work decomposition: a workgroup walks down a run of `tilesPerRun` vertically consecutive tiles of 256 x (8*NS) pixels,
// NS = strips per wave in {1, 2}.  Longer runs hide the memory latency behind the previous tile's arithmetic; taller
// tiles amortise the chroma halo rows; short runs of small tiles keep small images spread over the whole chip.
void decompose(uint32_t tuning, uint32_t w4, uint32_t h2, uint32_t jobs, bool batch, TileLaunch * L)
{
    const uint32_t bands = (w4 + kBandW - 1) / kBandW;
    uint32_t ns = (tuning >> TUNE_STRIPS_SHIFT) & 0xfu;
    const uint64_t tiles8 = (uint64_t)bands * ((h2 + 7) / 8) * jobs;
    if (ns == 0)
        ns = 2 * tiles8 >= 3 * (uint64_t)kTargetBlocks ? 2 : 1; // from 4K frames up (tests/tools/geometry_sweep.py: 4K 11.1 -> 9.5 us)
    const bool forced = ((tuning >> TUNE_STRIPS_SHIFT) & 0xfu) != 0;
    ns = (ns >= 2 && (!batch || forced)) ? 2 : 1; // batches are made of small jobs: short tiles unless the sweep says otherwise
    const uint32_t tilesY = (h2 + 8 * ns - 1) / (8 * ns);
    uint32_t run = (tuning >> TUNE_RUN_SHIFT) & 0xfu;
    if (run == 0) {
        // about kTargetBlocks / 2 workgroups (4 per CU) in the launch, each walking 1..3 tiles: the sweep prefers fewer, longer-lived
        // workgroups (the next tile's loads fly during the current tile's arithmetic) over one tile per workgroup; 8K frames keep
        // their runs of 3
        const uint64_t tiles = (uint64_t)bands * tilesY * jobs;
        run = (uint32_t)((2 * tiles + kTargetBlocks / 2) / kTargetBlocks);
        run = run < 1 ? 1 : (run > 3 ? 3 : run);
    }
    L->stripsPerWave = ns;
    L->tilesPerRun = run;
    L->bandsPerJob = bands, L->runsPerJob = (tilesY + run - 1) / run;
    L->blocksPerJob = L->bandsPerJob * L->runsPerJob;
    L->canvasColumns = 0;
    // packed 16-bit integer kernels (tile_pk_impl.h) derive their own grid from these
    L->maxW4 = w4, L->maxH2 = h2;
    L->pkStrips = (tuning >> TUNE_STRIPS_SHIFT) & 0xfu;
    const uint32_t wx = (tuning >> TUNE_WAVESX_SHIFT) & 3u;
    // defaults from tests/tools/pk_sweep.py (profiles/r02_pk_sweep.txt): the four waves stacked (a 256 x 32 tile), one tile row per
    // XCD chunk -- 28.8 us per 8K frame with 4 frames cycled, 36.0 with 12; raster order: 34.0 / 37.0
    L->wavesXLog2 = wx ? wx - 1 : 0;
    const uint32_t chunk = (tuning >> TUNE_CHUNK_SHIFT) & 0xfu;
    L->chunkRows = (tuning & TUNE_XCD_BANDS) ? (chunk ? chunk : 1) : 0;
    L->solo = (tuning & TUNE_COOPERATIVE) == 0; // (launchYuvToRgbTile / ...Batch narrow it down per kernel family)
}

// Where the wave-private kernels of the fp32 / 10-12-bit integer families beat round 1's cooperative runs (profiles/r02_solo_ab.txt:
// cfg3 89.7 -> 85.3 us, fp32 cfg2 33.5 -> 32.0 us; but cfg5x64 284 -> 322 us, 4K fp32 cfg2 11.2 -> 11.8 us): whenever no chroma
// neighbourhood is staged, and for 8-bit planes on frames from ~6 megapixels up (round 2: 16).  Staged 16-bit planes keep the cooperative runs,
// whose software pipeline (next tile's loads in flight during a long fp32 compute phase) is worth more than the missing barrier.
bool soloPays(const TileKey & k, uint64_t pixels)
{
    if (!k.bilinear)
        return true;
    // (round 3, after the wave-private kernels' staging lost a round and their filter its re-reads: 4K fp32 cfg2 9.6 us cooperative, 8.4-9.0
    //  wave-private, profiles/r03_cfgs_bench.jsonl; a 1080p 10-bit tile still 6.0 against 6.4)
    return !k.wideYuv && pixels >= ((uint64_t)6 << 20);
}

// largest byte offset the kernel forms from a plane base must fit 32 bits
bool fits32(uint64_t rows, uint32_t pitch)
{
    return rows * (uint64_t)pitch < ((uint64_t)1 << 32);
}

} // namespace

bool tileYuvToRgbSupported(const YuvToRgbPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    if (p.arith == ARITH_LIBYUV) {
        // fixed-point kernels (tile_fx_impl.h): the post-pass is libyuv's own attenuate / unattenuate, or -- ARGB / ABGR, which libyuv does not
        // attenuate -- the reference's fp32 form over libyuv's bytes (4-channel 8-bit pixels, rows: the fp32 form lives in the 32-bit kernels)
        if (p.postMul != MUL_NONE && !p.postMulFx && (!o.hasAlpha || o.chanBytes != 1 || o.map.on || o.is565))
            return false;
        if ((s.hasColor != 0) != (s.format != AVIF_PIXEL_FORMAT_YUV400))
            return false;
        if (p.fxAlpha == FXA_FLOAT && s.depth != o.depth && !s.exactDiv)
            return false;
    } else {
        // libyuv's attenuate over the fp32 loops' bytes (8-bit RGBA / BGRA whose conversion libyuv has no entry for): the post-pass of the fp32
        // tiles in libyuv's arithmetic (rows, no pixel map)
        if (p.postMulFx && (o.chanBytes != 1 || !o.hasAlpha || o.map.on || o.is565 || p.postMul == MUL_NONE))
            return false;
        if (p.identityCopy) {
            // lossless RGB in 8-bit 4:4:4 planes: a byte shuffle inside the 4:4:4 kernel (no arithmetic, no divisors to verify) ...
            if (p.inLoopMul != MUL_NONE || s.chanBytes != 1 || o.chanBytes != 1 || s.format != AVIF_PIXEL_FORMAT_YUV444)
                return false;
            // ... unless an integer alpha (un)multiply follows the copy (src/reformat.c:1574-1585: premultiplied lossless images): then the
            // kernels with alpha arithmetic run the identity transform as arithmetic -- (uint8_t)(0.5f + (c / 255.0f) * 255.0f) is c for every
            // code, the error of the quotient is below 2e-5 -- and the post-pass on the codes (tile_shared.h: identityMatrix)
            if (p.postMul != MUL_NONE && !s.exactDiv)
                return false;
        } else {
            // matrix coefficients, the identity matrix at any depth / range (GBR planes: 4:4:4; without chroma every matrix is the same), or the
            // YCgCo family (three adds / integer lifting per pixel)
            const bool ycgco = s.mode == MODE_YCGCO || s.mode == MODE_YCGCO_RE || s.mode == MODE_YCGCO_RO;
            if (s.mode != MODE_COEFF && !ycgco && !(s.mode == MODE_IDENTITY && (s.format == AVIF_PIXEL_FORMAT_YUV444 || !s.hasColor)))
                return false;
            if (!s.exactDiv)
                return false; // a divisor off the verified list (exactdiv.h): the universal kernel divides the IEEE way
        }
    }
    if (o.isGray && p.arith == ARITH_LIBYUV)
        return false; // (libyuv has no gray entries: never the case)
    if (o.isFloat && (p.arith == ARITH_LIBYUV || o.chanBytes != 2))
        return false; // half-float outputs (Android's RGBA_F16 bitmaps): 16-bit containers, the fp32 kernels convert at the store
    if (o.is565) {
        // RGB565 (Android's bitmap format, android_jni/.../libavif_jni.cc:206-223): libyuv's I420ToRGB565Matrix / I422ToRGB565Matrix --
        // 8-bit 4:2:0 / 4:2:2 planes, nearest upsampling -- in the packed 16-bit kernels
        // (10- / 12-bit planes reach those two entries through Convert16To8Plane, src/reformat_libyuv.c:906-930: the packed kernels' front end
        //  for 16-bit containers, which unfiltered chroma always selects -- soloPays)
        const bool packed = p.arith == ARITH_LIBYUV && !p.bilinear && p.inLoopMul == MUL_NONE && p.postMul == MUL_NONE &&
                            (s.chanBytes == 1 || (p.fxDownshift != 0 && (p.tuning & TUNE_COOPERATIVE) == 0)) &&
                            (s.format == AVIF_PIXEL_FORMAT_YUV420 || s.format == AVIF_PIXEL_FORMAT_YUV422) && s.hasColor;
        // ... and from the fp32 arithmetic (10- / 12-bit sources, filtered chroma, avoidLibYUV) in the fp32 tiles, alpha arithmetic aside
        // (an alpha plane that the format drops is multiplied in inside the loop, src/reformat.c:1503-1511: the kernels with alpha arithmetic)
        const bool fp32 = p.arith != ARITH_LIBYUV && p.postMul == MUL_NONE;
        if (!(packed || fp32) || o.map.on)
            return false;
    }
    if (o.map.on) {
        // fused crop / rotate / mirror: the packed 16-bit kernels (3- and 4-byte pixels) and the wave-private fp32 kernels (4-channel pixels of
        // 4 or 8 bytes: tile_map_impl.h) store through the map; every other family leaves it to the universal kernel (the entry points
        // convert into scratch and run the transform pass instead: api_batch.cpp)
        const bool packed = p.arith == ARITH_LIBYUV && p.inLoopMul == MUL_NONE && p.postMul == MUL_NONE && (s.chanBytes == 1 || (p.tuning & TUNE_COOPERATIVE) == 0);
        const bool fp32Mapped = p.arith != ARITH_LIBYUV && !o.isGray && !o.is565 && o.hasAlpha && (o.pixBytes == 4 || o.pixBytes == 8);
        if (!(packed && (o.pixBytes == 4 || o.pixBytes == 3)) && !fp32Mapped)
            return false;
        if (((uintptr_t)o.pixels % 4) != 0 || (o.rowBytes % 4) != 0)
            return false;
    }
    // (limited-range alpha planes, pre-1.0 files, src/read.c:6818-6828: converted sample by sample inside the tiled kernels)
    if (o.hasAlpha && p.alphaSource == ALPHA_KEEP) {
        // destination alpha samples must stay as they are: the fp32 tiles read them back per pixel (TileArgs::alphaKeep); not behind a pixel
        // map, not with alpha arithmetic pending (the reference has none either: ignoreAlpha), not in the integer path (libyuv writes 255)
        if (p.arith == ARITH_LIBYUV || o.map.on || p.inLoopMul != MUL_NONE || p.postMul != MUL_NONE || o.is565)
            return false;
    }
    if (p.w < 64 || p.h < 2)
        return false; // too small to fill even one row of lanes: the per-pixel kernel is the better fit
    // vector accesses: 4 samples per load, 4 pixels per store
    const uint32_t sampleVec = 4 * (uint32_t)s.chanBytes;
    if ((p.x0 % 8) != 0 || (p.y0 % 2) != 0)
        return false;
    if (!aligned(s.plane[0], s.rowBytes[0], sampleVec))
        return false;
    if (s.hasColor && !o.isGray && (!aligned(s.plane[1], s.rowBytes[1], sampleVec) || !aligned(s.plane[2], s.rowBytes[2], sampleVec)))
        return false;
    const bool readsAlpha = (o.hasAlpha && p.alphaSource == ALPHA_PLANE) || p.inLoopMul != MUL_NONE || p.postMul != MUL_NONE;
    if (readsAlpha && (!s.alpha || !s.alphaRowBytes || !aligned(s.alpha, s.alphaRowBytes, sampleVec)))
        return false;
    if (p.postMul != MUL_NONE && p.alphaSource != ALPHA_PLANE)
        return false;
    const int nch = o.is565 ? 2 : (o.hasAlpha ? 4 : 3);
    const uint32_t storeAlign = o.map.on ? 1u : (o.isGray ? 4u * (uint32_t)o.pixBytes : ((nch == 4) ? 16u : (nch == 2 ? 2u : (o.chanBytes == 1 ? 4u : 8u))));
    if (!aligned(o.pixels, o.rowBytes, storeAlign))
        return false;
    // 32-bit lane offsets from the plane bases
    if (!fits32(p.canvasH, s.rowBytes[0]) || !fits32(o.map.on ? (o.map.transposed ? o.map.cw : o.map.ch) : p.canvasH, o.rowBytes) || (s.hasColor && (!fits32(p.canvasH, s.rowBytes[1]) || !fits32(p.canvasH, s.rowBytes[2]))) ||
        (readsAlpha && !fits32(p.canvasH, s.alphaRowBytes)))
        return false;
    return true;
}

int tileYuvToRgbVariant(const YuvToRgbPlan & plan)
{
    if (!tileYuvToRgbSupported(plan))
        return -1;
    const TileKey k = keyFor(plan);
    return (k.wideYuv ? 1 : 0) | (k.sub << 1) | ((k.bilinear ? 1 : 0) << 3) | ((k.wideRgb ? 1 : 0) << 4) | ((k.nch == 4 ? 1 : 0) << 5) | ((k.nch == 2 ? 1 : 0) << 10) |
           ((k.hasMul ? 1 : 0) << 6) | ((k.alphaPlane ? 1 : 0) << 7) | ((k.fixedPoint ? 1 : 0) << 8) | ((k.mapped ? 1 : 0) << 9) |
           ((k.wideDownshift ? 1 : 0) << 11) | ((k.gray ? 1 : 0) << 12) | ((k.attenuate & 3) << 13);
}

namespace {
// bytes of plane samples one pixel of a job reads
double planeBytesPerPixel(const YuvToRgbPlan & p, const TileKey & k)
{
    return (double)p.yuv.chanBytes * ((k.sub == SUB_444 ? 3.0 : k.sub == SUB_422 ? 2.0 : k.sub == SUB_420 ? 1.5 : 1.0) + ((k.alphaPlane || k.hasMul) ? 1.0 : 0.0));
}
// Planes beyond this many bytes per launch cannot be resident in the 256 MB Infinity Cache next to anything else: streaming loads
constexpr double kStreamPlaneBytes = 128.0 * 1048576.0;
} // namespace

// which single alpha mode a job with pending alpha arithmetic asks for (tile_impl.h computeTile MULSEL): 1 / 2 in-loop multiply / un-multiply,
// 3 / 4 integer post-multiply / un-multiply; 0: none of the four (or AVIFHIP_TUNING asks for the kernel with every mode compiled in)
static int alphaSelOf(const tile::TileArgs & a)
{
    if (a.tuning & TUNE_ALL_ALPHA_MODES)
        return 0;
    if (a.postMul == MUL_NONE)
        return a.inLoopMul == MUL_MULTIPLY ? 1 : (a.inLoopMul == MUL_UNMULTIPLY ? 2 : 0);
    if (a.inLoopMul == MUL_NONE)
        return a.postMul == MUL_MULTIPLY ? 3 : (a.postMul == MUL_UNMULTIPLY ? 4 : 0);
    return 0;
}

// the launch of one image's whole-group part (a sequence: of its first frame's, tile_shared.h SeqFrames)
static void singleLaunchOf(const YuvToRgbPlan & plan, const TileKey & k, const TileArgs & A, hipStream_t stream, TileLaunch * out)
{
    TileLaunch & L = *out;
    L.args = &A;
    L.alphaSel = alphaSelOf(A);
    L.table = nullptr;
    L.seq = nullptr, L.seqCount = 0;
    L.count = 1;
    L.seams = false;
    L.stream = stream;
    L.shiftStrips = 0;
    L.mapped = k.mapped, L.transposed = k.mapped && plan.rgb.map.transposed, L.streamLoads = false;
    // single images: plain loads (profiles/r02_stream_single_ab.txt: a frame converted again and again finds part of itself in the Infinity
    // Cache) unless TUNE_STREAM_LOADS asks the unfiltered 16-bit families for streaming ones (A/B measurements: tests/tools/stream_sweep.py)
    L.streamLoads = (plan.tuning & TUNE_STREAM_LOADS) != 0;
    decompose(plan.tuning, A.w4, A.h2, 1, false, &L);
    L.solo = L.solo && (soloPays(k, (uint64_t)A.w4 * A.h2) || (plan.tuning & TUNE_SOLO_ALWAYS));
    L.pkWide = (plan.tuning & TUNE_COOPERATIVE) == 0, L.wideDownshift = k.wideDownshift, L.attenuate = k.attenuate;
    // A frame whose planes and pixels together exceed the 256 MB Infinity Cache (8K 10-bit 4:4:4 + alpha -> RGBA16: 530 MB) streams from and to
    // HBM whatever the tile order, and its next frame finds nothing of it in the cache either.  The per-XCD bands of tall tiles that pay when
    // planes are cache-resident then only scatter the DRAM accesses: the unfiltered fp32 families (no chroma halo to share, so the tile shape is
    // free) take short wide tiles -- the four waves of a workgroup side by side, 1024 x 4 pixels -- in raster order, with streaming plane loads.
    // tests/tools/stream_sweep.py cfg3, two frames cycled, one box: 103.5 us -> 92.5 (0.64 -> 0.72; its byte-movement ceiling: 94.9 / 92 us
    // without / with streaming loads).  Explicit geometry bits in AVIFHIP_TUNING / avifhipSetTuning switch the rule off (A/B measurements).
    // 8-bit 4:2:0 planes filtered by the packed integer kernels (the headline's kernel), four waves stacked and one chroma neighbourhood per
    // workgroup: two strips per wave (256 x 16 tiles) at every size.  Four (256 x 32) were the rule from 2048 workgroups up until round 4; since
    // the workgroup shares its neighbourhood the taller tile buys nothing when the planes are cache-resident (8K, 4 frames cycled: 28.14 / 28.24
    // us, interleaved A/B) and costs 2.8 % when they stream (12 frames cycled: 34.98 -> 34.01 us; profiles/r05_stream_sweep_ab.jsonl)
    // (not behind a pixel map: quarter turns transpose 32-row tiles through LDS -- 16-row ones cost them 38 -> 63 us at 8K)
    if (k.fixedPoint && !k.wideYuv && k.bilinear && k.sub == SUB_420 && L.pkStrips == 0 && !k.mapped)
        L.pkStrips = 2;
    const bool geometryForced = (plan.tuning & (0xfu << TUNE_STRIPS_SHIFT | 3u << TUNE_WAVESX_SHIFT | 0xfu << TUNE_CHUNK_SHIFT | TUNE_STREAM_LOADS)) != 0 ||
                                (plan.tuning & TUNE_XCD_BANDS) == 0;
    if (!k.fixedPoint && !k.bilinear && L.solo && !k.mapped && !geometryForced &&
        (double)A.w4 * A.h2 * (planeBytesPerPixel(plan, k) + (double)plan.rgb.pixBytes) > 256.0 * 1048576.0) {
        L.chunkRows = 0, L.wavesXLog2 = 2, L.pkStrips = 2;
        L.streamLoads = true;
    }
}

// leftovers of one image: columns [w4, w) of every row, then row h2 of the columns before w4
static hipError_t launchLeftovers(const YuvToRgbPlan & plan, const TileArgs & A, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    if (A.w4 != plan.w) {
        YuvToRgbPlan rest = plan;
        rest.x0 = plan.x0 + A.w4, rest.w = plan.w - A.w4;
        e = launchYuvToRgbGeneric(rest, stream);
        if (e != hipSuccess)
            return e;
    }
    if (A.h2 != plan.h) {
        YuvToRgbPlan rest = plan;
        rest.y0 = plan.y0 + A.h2, rest.h = plan.h - A.h2, rest.w = A.w4;
        e = launchYuvToRgbGeneric(rest, stream);
    }
    return e;
}

// A sequence's frames must be large enough that the single image's launch geometry is the right one for each of them (smaller jobs are what
// the table batches are tuned for: grids of tiles)
static constexpr uint64_t kSequenceMinPixels = (uint64_t)2 << 20;

static_assert(kTileSequenceMax == kSeqMaxFrames, "kernels.h and tile_shared.h disagree on the frames of a sequence launch");
bool tileSequenceCompatible(const YuvToRgbPlan & first, const YuvToRgbPlan & other)
{
    if (tileYuvToRgbVariant(first) < 0 || tileYuvToRgbVariant(first) != tileYuvToRgbVariant(other) || first.rgb.map.on || other.rgb.map.on)
        return false;
    if (first.w != other.w || first.h != other.h || (uint64_t)first.w * first.h < kSequenceMinPixels)
        return false;
    return seqCompatible(distillArgs(first), distillArgs(other));
}

hipError_t launchYuvToRgbTileSequence(const YuvToRgbPlan * plans, uint32_t count, hipStream_t stream, const char ** kernelName)
{
    if (count == 0 || count > kSeqMaxFrames)
        return hipErrorInvalidValue;
    const YuvToRgbPlan & plan = plans[0];
    const TileKey k = keyFor(plan);
    const TileArgs A = distillArgs(plan);
    TileLaunch L;
    singleLaunchOf(plan, k, A, stream, &L);
    // the families with sequence kernels: the packed 16-bit integer kernels and the wave-private fp32 kernels, rows stored in place
    const bool packed = k.fixedPoint && !k.hasMul && (!k.wideYuv || L.pkWide);
    const bool soloFp32 = !k.fixedPoint && L.solo;
    if (k.mapped || !(packed || soloFp32))
        return hipErrorNotSupported;
    SeqFrames S = seqOfOne(A);
    for (uint32_t f = 1; f < count; ++f)
        seqSetFrame(S, f, distillArgs(plans[f]));
    L.seq = &S, L.seqCount = count;
    hipError_t e = launchFamily(k, L);
    if (e != hipSuccess)
        return e; // (hipErrorNotSupported: nothing was launched)
    if (kernelName)
        *kernelName = kernelNameFor(k, plan.tuning);
    for (uint32_t f = 0; f < count && e == hipSuccess; ++f)
        e = launchLeftovers(plans[f], A, stream);
    return e;
}

hipError_t launchYuvToRgbTile(const YuvToRgbPlan & plan, hipStream_t stream, const char ** kernelName)
{
    const TileKey k = keyFor(plan);
    if (kernelName)
        *kernelName = kernelNameFor(k, plan.tuning);
    const TileArgs A = distillArgs(plan);
    TileLaunch L;
    singleLaunchOf(plan, k, A, stream, &L);
    hipError_t e = launchFamily(k, L);
    if (e != hipSuccess)
        return e;
    return launchLeftovers(plan, A, stream);
}

size_t tileBatchTableBytes(uint32_t count)
{
    return (size_t)count * sizeof(TileArgs);
}

void fillTileBatchTable(const YuvToRgbPlan * plans, uint32_t count, void * hostTable)
{
    TileArgs * t = static_cast<TileArgs *>(hostTable);
    for (uint32_t k = 0; k < count; ++k)
        t[k] = distillArgs(plans[k]);
}

// Where one launch beats the tile batch plus the seam pass (A/B on one box, tests/tools/cfg_bench.py with AVIFHIP_GRID_SEAM_PASS = 1 / 0):
// the packed 16-bit kernels have registers to spare for the seam-aware loads (42 -> 48, same occupancy) and win everywhere -- a 12-megapixel
// photograph in 48 tiles 21.1 -> 12.9 us, cfg5's 64 tiles -> RGBA8 173.9 -> 165.9; the fp32 kernels pay for them with a step of occupancy
// (the cooperative 10-bit kernel: 76 -> 86 registers), so they win where the second launch is a large part of the call (the photograph:
// 21.4 -> 14.3 us) and lose where it is not (cfg5 -> RGBA(10): 263-269 -> 273 us; -> RGBA8 177.5 -> 183.9).
bool tileBatchLinksNeighbours(const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW, uint32_t maxH, int forced, uint32_t canvasColumns)
{
    const TileKey k = keyFor(representative);
    if (!k.bilinear)
        return false; // (nothing is filtered: seams need nothing)
    if (forced >= 0)
        return forced != 0;
    const bool packed = k.fixedPoint && (!k.hasMul || (k.attenuate && !k.mapped)) && (!k.wideYuv || (representative.tuning & TUNE_COOPERATIVE) == 0);
    if (packed || (uint64_t)maxW * maxH * count <= ((uint64_t)32 << 20))
        return true;
    // Round 5: large fp32 grids that walk along the canvas rows in the wave-private kernels (launchYuvToRgbTileBatch) read across the seams
    // themselves too -- in that order one launch beats the tile batch + seam pass (cfg5's canvas, interleaved on one box: 251.0 against 259.3
    // us), where the cooperative runs job by job had lost to it (283.2 against 279.6: round 4's rule, which TUNE_JOB_MAJOR / TUNE_COOPERATIVE keep)
    const double bytesPerPixel = (double)representative.yuv.chanBytes * (k.sub == SUB_444 ? 3.0 : k.sub == SUB_422 ? 2.0 : k.sub == SUB_420 ? 1.5 : 1.0) +
                                 (double)representative.rgb.pixBytes + (k.alphaPlane || k.hasMul ? (double)representative.yuv.chanBytes : 0.0);
    const bool streams = (double)maxW * maxH * count * bytesPerPixel > 192.0 * 1048576.0;
    return canvasColumns > 1 && streams && !k.fixedPoint && !k.mapped && (representative.tuning & (TUNE_COOPERATIVE | TUNE_JOB_MAJOR)) == 0;
}

void linkTileBatchHalo(void * hostTable, uint32_t job, const TileNeighbours & n)
{
    linkHalo(static_cast<TileArgs *>(hostTable)[job], n.plane1, n.plane2, n.above, n.below, n.left, n.right);
}

hipError_t launchYuvToRgbTileBatch(const void * deviceTileTable, const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW,
                                   uint32_t maxH, hipStream_t stream, const char ** kernelName, bool neighboursLinked, uint32_t canvasColumns)
{
    const TileKey k = keyFor(representative);
    if (kernelName)
        *kernelName = kernelNameFor(k, representative.tuning);
    TileLaunch L;
    L.args = nullptr;
    L.seq = nullptr, L.seqCount = 0;
    L.seams = neighboursLinked && k.bilinear;
    L.alphaSel = alphaSelOf(distillArgs(representative)); // (all jobs of a batch share their configuration)
    L.table = static_cast<const TileArgs *>(deviceTileTable);
    L.count = count;
    L.stream = stream;
    L.shiftStrips = 0;
    L.mapped = k.mapped, L.transposed = k.mapped && representative.rgb.map.transposed;
    // (upper bound from the largest job: batches are made of equally sized tiles)
    L.streamLoads = (double)maxW * maxH * count * planeBytesPerPixel(representative, k) > kStreamPlaneBytes;
    if (const char * e = getenv("AVIFHIP_STREAM_LOADS")) // diagnostics / A-B measurements only; exactly "1" or "0"
        if ((e[0] == '0' || e[0] == '1') && !e[1])
            L.streamLoads = e[0] == '1';
    decompose(representative.tuning, maxW & ~3u, maxH & ~1u, count, true, &L);
    L.solo = L.solo && (soloPays(k, (uint64_t)(maxW & ~3u) * (maxH & ~1u) * count) || (representative.tuning & TUNE_SOLO_ALWAYS));
    L.pkWide = (representative.tuning & TUNE_COOPERATIVE) == 0, L.wideDownshift = k.wideDownshift, L.attenuate = k.attenuate;
    const double bytesPerPixel = (double)representative.yuv.chanBytes * (k.sub == SUB_444 ? 3.0 : k.sub == SUB_422 ? 2.0 : k.sub == SUB_420 ? 1.5 : 1.0) +
                                 (double)representative.rgb.pixBytes + (k.alphaPlane || k.hasMul ? (double)representative.yuv.chanBytes : 0.0);
    const bool streams = (double)maxW * maxH * count * bytesPerPixel > 192.0 * 1048576.0; // beyond what the 256 MB Infinity Cache holds next to anything else
    const bool packed = k.fixedPoint && (!k.hasMul || (k.attenuate && !k.mapped)) && (!k.wideYuv || L.pkWide);
    // The tiles of one canvas (grids) that stream: workgroups along the rows of the CANVAS (tile_geom.h PkGeom::canvasColumns), so that a canvas
    // row's pixels leave in one sweep across the tiles that share it -- in the wave-private kernels, whose tiles need nothing from one another:
    // the packed ones with tall tiles (4 strips per wave), the fp32 ones in place of the cooperative runs.  Interleaved A/B on one box,
    // cfg5's 64 tiles of 1080p 10-bit into one canvas (profiles/r05_grid_ab.jsonl): -> RGBA8 168.6 us job by job (round 4) / 164.2 job by
    // job with 4 strips / 160.2 along the canvas with 4 strips / 184.1 along the canvas with 2; -> RGBA(10) 266.0 cooperative job by job
    // (round 4) / 281.7 cooperative along the canvas / 267.9 wave-private job by job / 256.4 wave-private along the canvas.  A grid that fits
    // the cache (the 12-megapixel photograph in 48 tiles) keeps the job-by-job order: 12.8 us against 14.1.  TUNE_JOB_MAJOR: round 4's order.
    L.canvasColumns = 0;
    if (canvasColumns > 1 && (streams || (representative.tuning & TUNE_CANVAS_ORDER)) && !k.mapped && (representative.tuning & TUNE_JOB_MAJOR) == 0) {
        if (packed) {
            L.canvasColumns = canvasColumns;
            if (L.pkStrips == 0)
                L.pkStrips = 4;
        } else if (!k.fixedPoint && (representative.tuning & TUNE_COOPERATIVE) == 0) {
            L.canvasColumns = canvasColumns;
            L.solo = true;
        }
    }
    // Linked grids in the packed kernels (tiles and seams in one launch): tall tiles at every size -- the photograph 12.8 -> 11.2 us
    if (L.seams && packed && L.pkStrips == 0)
        L.pkStrips = 4;
    // A batch whose bytes exceed the Infinity Cache (256 MB) streams from and to HBM whatever the tile order; the per-XCD chunks, which pay
    // when planes are cache-resident, then only scatter the DRAM accesses: plain raster order (tests/tools/pkbench_wide.hip, 64 tiles of
    // 1080p 10-bit -> RGBA8: 188 -> 175 us)
    if (streams && ((representative.tuning >> TUNE_CHUNK_SHIFT) & 0xfu) == 0) {
        if (k.fixedPoint && k.wideYuv && !k.hasMul && L.pkWide && L.pkStrips == 0 && L.chunkRows)
            L.pkStrips = 2; // 16-bit containers: twice the registers per strip -- shorter tiles in per-XCD rows (pkbench_wide: 174 -> 169 us)
        else
            L.chunkRows = 0;
    }
    return launchFamily(k, L);
}

} // namespace avifhip
