// kernels_tile_u8.hip -- tiled YUV->RGB kernels for 8-bit planes (instantiations of tile_impl.h)
#include "tile_impl.h"

namespace avifhip {
namespace tile {
hipError_t launchTileU8(const TileKey & key, const TileLaunch & launch)
{
    return launchYuvVariant<uint8_t>(key, launch);
}
} // namespace tile
} // namespace avifhip
