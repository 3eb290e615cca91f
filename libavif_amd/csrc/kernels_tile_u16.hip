// kernels_tile_u16.hip -- tiled YUV->RGB kernels for 10/12/16-bit planes in 16-bit containers (instantiations of tile_impl.h)
#include "tile_impl.h"

namespace avifhip {
namespace tile {
hipError_t launchTileU16(const TileKey & key, const TileLaunch & launch)
{
    return launchYuvVariant<uint16_t>(key, launch);
}
} // namespace tile
} // namespace avifhip
