// tile_shared.h -- host-visible description of a tiled-kernel launch (shared by the dispatch TU and the two
// instantiation TUs).
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace avifhip {
namespace tile {

enum Subsampling : int { SUB_444 = 0, SUB_422 = 1, SUB_420 = 2, SUB_400 = 3 };

struct TileKey
{
    bool wideYuv;
    int sub;
    bool bilinear;
    bool wideRgb;
    int nch;
    bool hasMul;
};

struct TileLaunch
{
    const YuvToRgbPlan * plan;  // single job (kernarg) ...
    const YuvToRgbPlan * table; // ... or device table of `count` jobs
    uint32_t count;
    uint32_t blocksPerJob; // tiles of the largest job
    hipStream_t stream;
};

hipError_t launchTileU8(const TileKey & key, const TileLaunch & launch);
hipError_t launchTileU16(const TileKey & key, const TileLaunch & launch);

} // namespace tile
} // namespace avifhip
