// kernels_tile_fx_inst.hip -- one instantiation unit of the tiled fixed-point YUV->RGB kernels (tile_fx_impl.h).  The
// Makefile compiles this file once per (sample type, chroma layout, upsampling) with -DTILE_YT=... -DTILE_SUB=...
// -DTILE_BIL=... and -DTILE_FN=<entry point name>; the families that filter chroma a second time with -DTILE_SEAMS (tile_impl.h).
#include "tile_fx_impl.h"

#if !defined(TILE_YT) || !defined(TILE_SUB) || !defined(TILE_BIL) || !defined(TILE_FN)
#error "compile with -DTILE_YT=<uint8_t|uint16_t> -DTILE_SUB=<SUB_4xx> -DTILE_BIL=<true|false> -DTILE_FN=<name>"
#endif

namespace avifhip {
namespace tile {
hipError_t TILE_FN(const TileKey & key, const TileLaunch & launch)
{
    return launchFxVariant<TILE_YT, TILE_SUB, TILE_BIL>(key, launch);
}
} // namespace tile
} // namespace avifhip
