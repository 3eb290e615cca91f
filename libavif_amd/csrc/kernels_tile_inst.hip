// kernels_tile_inst.hip -- one instantiation unit of the tiled YUV->RGB kernels (tile_impl.h).  The Makefile compiles
// this file once per (sample type, chroma layout, upsampling) with -DTILE_YT=... -DTILE_SUB=... -DTILE_BIL=... and
// -DTILE_FN=<entry point name>, so the twelve families build in parallel; the families that filter chroma a second time with -DTILE_SEAMS
// (batch kernels for grid canvases: tile_impl.h).
#include "tile_impl.h"

#if !defined(TILE_YT) || !defined(TILE_SUB) || !defined(TILE_BIL) || !defined(TILE_FN)
#error "compile with -DTILE_YT=<uint8_t|uint16_t> -DTILE_SUB=<SUB_4xx> -DTILE_BIL=<true|false> -DTILE_FN=<name>"
#endif

namespace avifhip {
namespace tile {
hipError_t TILE_FN(const TileKey & key, const TileLaunch & launch)
{
    return launchSubVariant<TILE_YT, TILE_SUB, TILE_BIL>(key, launch);
}
} // namespace tile
} // namespace avifhip
