// r2y_tile_shared.h -- host-visible description of a tiled RGB -> YUV launch (shared by the dispatch code in
// kernels_r2y_tile.hip and the instantiation units).
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace avifhip {
namespace r2y {

enum Subsampling : int { SUB_444 = 0, SUB_422 = 1, SUB_420 = 2, SUB_400 = 3 };
enum AlphaMode : int {
    R2Y_ALPHA_NONE = 0,   // the image has no alpha plane
    R2Y_ALPHA_FILL = 1,   // opaque (source without alpha, or rgb->ignoreAlpha), avifFillAlpha src/alpha.c:9-35
    R2Y_ALPHA_COPY = 2,   // same depth: plain copy, src/alpha.c:44-79
    R2Y_ALPHA_RESCALE = 3 // depth rescale in fp32, src/alpha.c:84-103
};

// Everything the kernel reads, small enough to stay in scalar registers.  It converts the w4 x h2 pixels at the image
// origin (w4 a multiple of 4, h2 a multiple of 2); the leftover columns / row go to the universal kernel.
struct R2YArgs
{
    const uint8_t * rgb;
    uint8_t * y;
    uint8_t * u;
    uint8_t * v;
    uint8_t * a;
    uint32_t rgbPitch, yPitch, uPitch, vPitch, aPitch;
    uint32_t w4, h2;
    float kr, kg, kb;
    RcpHL rcpRgbMax, rcpCbDen, rcpCrDen;
    float rangeY, biasY, rangeUV, biasUV, yuvMaxF;
    uint32_t yuvMax;
    uint32_t slotR, slotB, slotA;
    int32_t alphaMode;
    uint32_t stripsPerWave;
};

struct R2YKey
{
    bool wideRgb, wideYuv;
    int nch, sub;
};

hipError_t launchR2YTileRgb8(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream);
hipError_t launchR2YTileRgb16(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream);

} // namespace r2y
} // namespace avifhip
