// kernels_r2y_tile_inst.hip -- one instantiation unit of the tiled RGB -> YUV kernels (r2y_tile_impl.h); the Makefile
// compiles it once per RGB container type with -DR2Y_RT=<uint8_t|uint16_t> -DR2Y_FN=<entry point name>.
#include "r2y_tile_impl.h"

#if !defined(R2Y_RT) || !defined(R2Y_FN)
#error "compile with -DR2Y_RT=<uint8_t|uint16_t> -DR2Y_FN=<name>"
#endif

namespace avifhip {
namespace r2y {
hipError_t R2Y_FN(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & frames, uint32_t frameCount)
{
    return launchFamily<R2Y_RT>(key, args, blocks, stream, frames, frameCount);
}
#ifdef R2Y_GRAY_FN
hipError_t R2Y_GRAY_FN(int grayChannels, bool wideYuv, const GrayArgs & args, hipStream_t stream)
{
    return launchGray<R2Y_RT>(grayChannels, wideYuv, args, stream);
}
#endif
#ifdef R2Y_WITH_FX
hipError_t launchR2YTileFx(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & frames, uint32_t frameCount)
{
    return key.nch == 4 ? launchFxSub<4>(key.sub, args, blocks, stream, frames, frameCount) : launchFxSub<3>(key.sub, args, blocks, stream, frames, frameCount);
}
#endif
} // namespace r2y
} // namespace avifhip
