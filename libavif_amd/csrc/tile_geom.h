// tile_geom.h -- launch geometry shared by the tiled kernels whose waves work for themselves (one wave, one tile of
// 256 x 2*NSW pixels; tile_pk_impl.h and the wave-private kernels of tile_impl.h / tile_fx_impl.h): which tile a workgroup takes
// and how a job is cut into tiles.
#pragma once

#include <hip/hip_runtime.h>

#include "tile_shared.h"

// waves of a workgroup of the wave-private kernels (measurement builds may change it: tests/tools/seqbench.hip)
#ifndef AVIFHIP_PK_WAVES
#define AVIFHIP_PK_WAVES 4
#endif

namespace avifhip {
namespace tile {

// Geometry of a launch (kernel argument, lives in SGPRs)
struct PkGeom
{
    uint32_t wavesXLog2;  // waves of a workgroup side by side: 1 << wavesXLog2 in {1, 2, 4}
    uint32_t tilesX, nTiles;
    uint32_t magicTilesX; // ceil(2^32 / tilesX): tile / tilesX == mulhi(tile, magic) while tile * tilesX < 2^32; 0 when tilesX == 1
    uint32_t chunk;       // tiles per XCD chunk (a few tile rows), 0 = plain raster order
    uint32_t magicChunk;
    // quarter turns (launches that store through a transposing PixelMap): tiles are numbered DOWN the columns of the tile grid and a chunk is
    // one tile column, so the workgroups an XCD runs one after the other write neighbouring pieces of the same destination rows
    uint32_t columnMajor, tilesY, magicTilesY;
    // ... and their tile grid may start above the rectangle, by whole waves' worth of strips: the waves up there find no rows and only take
    // part in the transposition; what it buys is WHERE along a destination row a tile's 128-byte run starts (launchSoloMapped)
    uint32_t stripShift;
    // batches whose jobs are the tiles of ONE canvas, row-major, `canvasColumns` per canvas row (0: any other batch): workgroups are numbered
    // along the rows of the CANVAS -- grid x = (tile column of the canvas, tile of the job's tile row), y = tile row inside the job, z = tile row
    // of the canvas -- so that the pixels of a canvas row leave in one sweep across all the jobs that share it instead of job by job.  The
    // byte-movement ceiling of cfg5's canvas (64 tiles of 1080p into 15360 x 8640 RGBA16) runs 13 % faster in that order (236 against 273 us,
    // tests/tools/stream_sweep.py cfg5grid).  No XCD chunks in this order.
    uint32_t canvasColumns;
};

// which job of a batch (grid z, or the canvas order above) and which tile of that job's tile grid a workgroup takes
struct BatchWhere
{
    uint32_t job, tile;
};

// (The index arithmetic below is constexpr -- callable from host and device code alike -- so that tests/tools/geometry_check.cpp can walk
// every workgroup and wave of a launch on the CPU and count how often each strip of each band is visited: tests/test_host_plans.py.)
__attribute__((always_inline)) constexpr uint32_t mulHi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * b) >> 32);
}

__attribute__((always_inline)) constexpr uint32_t pkTileOf(uint32_t b, const PkGeom & g)
{
    if (g.chunk == 0)
        return b;
    // workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it): XCD x takes the x-th chunk of every
    // group of 8 chunks, so vertically adjacent tiles (which share chroma halo rows) mostly meet in one L2
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t sc = g.magicChunk ? mulHi32(slot, g.magicChunk) : slot, within = slot - sc * g.chunk;
    return (sc * 8u + xcd) * g.chunk + within;
}

__attribute__((always_inline)) constexpr BatchWhere pkBatchWhereOf(uint32_t bx, uint32_t by, uint32_t bz, const PkGeom & g)
{
    if (g.canvasColumns == 0)
        return BatchWhere { bz, pkTileOf(bx, g) };
    const uint32_t col = g.magicTilesX ? mulHi32(bx, g.magicTilesX) : bx; // bx / tilesX
    return BatchWhere { bz * g.canvasColumns + col, by * g.tilesX + (bx - col * g.tilesX) };
}
__device__ __forceinline__ BatchWhere pkBatchWhere(const PkGeom & g)
{
    return pkBatchWhereOf(blockIdx.x, blockIdx.y, blockIdx.z, g);
}
// ... and the grid that goes with it (host)
inline dim3 pkBatchGrid(const PkGeom & g, uint32_t blocks, uint32_t count)
{
    return g.canvasColumns ? dim3(g.tilesX * g.canvasColumns, g.tilesY, count / g.canvasColumns) : dim3(blocks, 1, count);
}

// The job of a workgroup of a sequence launch (tile_shared.h SeqFrames): the launch's arguments with the addresses of frame blockIdx.z.
// (Kernel arguments live in constant memory: the copy costs five scalar loads whose offset depends on the workgroup.)
__device__ __forceinline__ TileArgs seqJob(const TileArgs & A, const SeqFrames & S)
{
    TileArgs job = A;
    const SeqFrames::Frame f = S.f[blockIdx.z];
    job.y = f.y, job.a = f.a, job.u = f.u, job.v = f.v, job.rgb = f.rgb;
    return job;
}

// Where wave `wave` (0..3) of the workgroup that took `tile` works: its band of 256 pixels and its first strip of 2 rows (it owns
// `ns` consecutive strips)
struct PkPlace
{
    uint32_t band, strip0;
};
__attribute__((always_inline)) constexpr PkPlace pkPlaceOf(uint32_t tile, uint32_t wave, const PkGeom & g, uint32_t ns)
{
    const uint32_t wx = wave & ((1u << g.wavesXLog2) - 1u), wy = wave >> g.wavesXLog2;
    const uint32_t wavesY = (uint32_t)AVIFHIP_PK_WAVES >> g.wavesXLog2;
    uint32_t trow = 0, tcol = 0;
    if (g.columnMajor) {
        tcol = g.magicTilesY ? mulHi32(tile, g.magicTilesY) : tile, trow = tile - tcol * g.tilesY;
    } else {
        trow = g.magicTilesX ? mulHi32(tile, g.magicTilesX) : tile, tcol = tile - trow * g.tilesX;
    }
    return PkPlace { (tcol << g.wavesXLog2) + wx, (trow * wavesY + wy) * ns - g.stripShift }; // (wraps above the rectangle: 2 * strip0 >= h2 there too)
}

// The cooperative kernels' order (tile_impl.h runBlock, tile_fx_impl.h): workgroups are dispatched round-robin over the 8 XCDs;
// giving XCD x the x-th contiguous run of tiles keeps vertically adjacent tiles (which share chroma halo rows) on one L2.
__attribute__((always_inline)) constexpr uint32_t blockRemap(uint32_t b, uint32_t n, bool bands)
{
    if (!bands || n < 64)
        return b;
    const uint32_t per = n >> 3, rem = n & 7;
    const uint32_t xcd = b & 7, slot = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

// Quarter turns of a single image: how many strips above the rectangle the tile grid starts (a multiple of the strips per wave, at most
// three waves' worth).  A tile's rows leave as one 128-byte run per source column; where along the destination row the runs start is set
// by the row the tile grid starts on.  plan.h coverOfCrop picks the rectangle's first row with that in mind but cannot go above the canvas;
// the grid can start above the RECTANGLE, by whole waves, whose lanes then find no rows.  Same ranking: whole cache lines if any start gives
// them, else as far from an even split as possible.  Worth 12 % for 8-byte pixels and nothing measurable for 4-byte ones (the
// packed kernels, tried: 38.2 -> 38.4 us), so only the fp32 tiles use it (tile_impl.h launchSoloMapped).
inline uint32_t turnShiftStrips(const TileArgs & A, int pixelBytes, int stripsPerWave)
{
    if (pixelBytes != 4 && pixelBytes != 8)
        return 0;
    const PixelMap & m = A.map;
    const int waveRows = 2 * stripsPerWave, runPx = 4 * waveRows, runBytes = runPx * pixelBytes; // the four stacked waves' rows: 64 or 128 bytes
    int bestScore = -1;
    uint32_t best = 0;
    for (int w = 0; w < 4; ++w) {
        const int64_t d = (int64_t)A.mapY0 - w * waveRows - (int64_t)m.cy; // first row of the tile grid, in crop rows
        const int64_t startPx = m.sx > 0 ? (int64_t)m.kx + d : (int64_t)m.kx - d - (runPx - 1);
        const int off = (int)(((int64_t)(uintptr_t)A.rgb + startPx * pixelBytes) & (runBytes - 1));
        const int score = off == 0 ? 1000 : (off > runBytes / 2 ? off - runBytes / 2 : runBytes / 2 - off);
        if (score > bestScore)
            bestScore = score, best = (uint32_t)(w * stripsPerWave);
    }
    return best;
}

// Launch geometry: strips per wave, waves side by side, tile order (TuningBits; tests/tools/geometry_sweep.py)
inline void pkGeometry(const TileLaunch & L, uint32_t w4, uint32_t h2, uint32_t * nsw, PkGeom * g, uint32_t * blocks, bool oneStripKernels = false)
{
    const uint32_t bands = (w4 + 256u - 1) / 256u, strips = h2 / 2;
    uint32_t ns = L.pkStrips; // 0 = automatic
    if (ns != 2 && ns != 4 && !(ns == 1 && oneStripKernels))
        ns = ((uint64_t)bands * ((strips + 15) / 16) * L.count >= 2048) ? 4 : 2; // 4 strips per wave only if that still makes 2048 workgroups
                                                                                 // (8 per CU): a 4K frame runs 12 % faster with 2 (9.4 -> 8.2 us)
    uint32_t wxl = L.wavesXLog2 <= 2 ? L.wavesXLog2 : 2;
    while (wxl > 0 && (1u << wxl) > bands)
        --wxl;
    const uint32_t wavesX = 1u << wxl, wavesY = (uint32_t)AVIFHIP_PK_WAVES / wavesX;
    g->wavesXLog2 = wxl;
    g->tilesX = (bands + wavesX - 1) / wavesX;
    const uint32_t shift = L.shiftStrips - L.shiftStrips % ns; // whole waves
    g->stripShift = shift;
    const uint32_t tilesY = (strips + shift + ns * wavesY - 1) / (ns * wavesY);
    g->nTiles = g->tilesX * tilesY;
    auto magic = [](uint32_t d) { return d > 1 ? (uint32_t)((((uint64_t)1 << 32) + d - 1) / d) : 0u; };
    g->magicTilesX = magic(g->tilesX);
    g->columnMajor = L.transposed ? 1u : 0u;
    g->tilesY = tilesY, g->magicTilesY = magic(tilesY);
    // magic divisions are exact while tile * divisor < 2^32: tiles number far fewer than 2^16 (32768-pixel sides: 128 x 4096)
    g->chunk = L.transposed ? (L.chunkRows ? tilesY : 0u) : L.chunkRows * g->tilesX;
    g->canvasColumns = (L.table && L.canvasColumns > 1 && !L.mapped && L.count % L.canvasColumns == 0) ? L.canvasColumns : 0u;
    if (g->canvasColumns)
        g->chunk = 0;
    g->magicChunk = magic(g->chunk);
    *nsw = ns;
    // chunked order: padded to whole groups of 8 chunks (workgroups beyond the last tile leave at once)
    *blocks = g->chunk ? ((g->nTiles + 8 * g->chunk - 1) / (8 * g->chunk)) * 8 * g->chunk : g->nTiles;
}

} // namespace tile
} // namespace avifhip
