// kernels_bench.hip -- measurement support, not part of the conversion path: a zero-arithmetic kernel that moves exactly the
// bytes of an 8-bit 4:2:0 -> 4-byte-pixel conversion (every plane sample read once, every output byte written once, same
// lane-to-byte mapping and store policy as the tiled kernels).  bench.py reports its duration beside the conversion kernel's:
// the chip's own ceiling for this byte movement (HBM3E sustains less than its 8 TB/s pin rate, and less again for a
// write-heavy mix), against which the conversion kernel's roofline fraction can be read.
#include <hip/hip_runtime.h>

#include "api_internal.h"

namespace avifhip {
namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bandedTile(uint32_t b, uint32_t n)
{
    const uint32_t per = n >> 3, rem = n & 7, xcd = b & 7, slot = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

// Two access patterns, both one tile per workgroup, lane = 4 pixels of every row of its wave:
//   WAVES_X = 4, RPL = 2: four waves side by side (1024 x 2 pixels), raster order  -- best when nothing is cached (pattern_probe.hip)
//   WAVES_X = 1, RPL = 4: four waves stacked (256 x 16 pixels), one band of tiles per XCD -- best when the planes sit in the Infinity Cache
template <int WAVES_X, int RPL, bool BANDED>
__global__ __launch_bounds__(256) void streamCeilingKernel(const uint8_t * __restrict__ y, const uint8_t * __restrict__ u, const uint8_t * __restrict__ v,
                                                           uint8_t * __restrict__ rgba, uint32_t yPitch, uint32_t uPitch, uint32_t vPitch, uint32_t rgbPitch,
                                                           uint32_t w4, uint32_t h2, uint32_t tilesX)
{
    constexpr int WAVES_Y = 4 / WAVES_X;
    const uint32_t tile = BANDED ? bandedTile(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t trow = tile / tilesX, tcol = tile - trow * tilesX;
    const uint32_t wave = threadIdx.y, wx = wave % WAVES_X, wy = wave / WAVES_X;
    const uint32_t X = (tcol * WAVES_X + wx) * 256u + 4u * threadIdx.x;
    const uint32_t Y0 = (trow * WAVES_Y + wy) * RPL;
    if (X >= w4 || Y0 >= h2)
        return;
    unsigned wy_[RPL], cu[RPL], cv[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const uint32_t Y = Y0 + r < h2 ? Y0 + r : h2 - 1;
        wy_[r] = *reinterpret_cast<const unsigned *>(y + (size_t)Y * yPitch + X);
        cu[r] = cv[r] = 0;
        if (!(r & 1)) {
            cu[r] = *reinterpret_cast<const uint16_t *>(u + (size_t)(Y >> 1) * uPitch + (X >> 1));
            cv[r] = *reinterpret_cast<const uint16_t *>(v + (size_t)(Y >> 1) * vPitch + (X >> 1));
        }
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        if (Y0 + r >= h2)
            break;
        const unsigned c = cu[r & ~1] | (cv[r & ~1] << 16);
        u4 o;
        o.x = (wy_[r] & 0xff) | (c << 8);
        o.y = ((wy_[r] >> 8) & 0xff) | (c << 8);
        o.z = ((wy_[r] >> 16) & 0xff) | (c & 0xffffff00u);
        o.w = (wy_[r] >> 24) | (c & 0xffffff00u);
        __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(rgba + (size_t)(Y0 + r) * rgbPitch + (size_t)X * 4));
    }
}

bool launchCeiling(const avifImage * image, const avifRGBImage * rgb, int pattern, hipStream_t stream)
{
    if (!image || !rgb || image->depth != 8 || image->yuvFormat != AVIF_PIXEL_FORMAT_YUV420 || rgb->depth != 8 || rgbFormatChannelCount((int)rgb->format) != 4 ||
        !image->yuvPlanes[0] || !image->yuvPlanes[1] || !image->yuvPlanes[2] || !rgb->pixels)
        return false;
    const uint32_t w4 = image->width & ~3u, h2 = image->height & ~1u;
    if (!w4 || !h2 || (image->yuvRowBytes[0] & 3u) || (image->yuvRowBytes[1] & 1u) || (image->yuvRowBytes[2] & 1u) || (rgb->rowBytes & 15u) || ((uintptr_t)rgb->pixels & 15u))
        return false;
    if (pattern == 0) {
        const uint32_t tilesX = (w4 + 1023u) / 1024u;
        hipLaunchKernelGGL((streamCeilingKernel<4, 2, false>), dim3(tilesX * (h2 / 2)), dim3(64, 4), 0, stream, image->yuvPlanes[0], image->yuvPlanes[1],
                           image->yuvPlanes[2], rgb->pixels, image->yuvRowBytes[0], image->yuvRowBytes[1], image->yuvRowBytes[2], rgb->rowBytes, w4, h2, tilesX);
    } else {
        const uint32_t tilesX = (w4 + 255u) / 256u;
        hipLaunchKernelGGL((streamCeilingKernel<1, 4, true>), dim3(tilesX * ((h2 + 15) / 16)), dim3(64, 4), 0, stream, image->yuvPlanes[0], image->yuvPlanes[1],
                           image->yuvPlanes[2], rgb->pixels, image->yuvRowBytes[0], image->yuvRowBytes[1], image->yuvRowBytes[2], rgb->rowBytes, w4, h2, tilesX);
    }
    return hipGetLastError() == hipSuccess;
}

} // namespace
} // namespace avifhip

using namespace avifhip;
using namespace avifhip::api;

// Average milliseconds per launch of the byte-movement-only kernel, cycling over `count` device-resident frames like
// avifhipTimeYUVToRGBCycle (the RGB buffers are overwritten with meaningless bytes); the faster of the two access patterns
// above.  Negative when the frames are not 8-bit 4:2:0 planes with 4-byte pixels.
static double timeCeilingPattern(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, int pattern, hipStream_t stream)
{
    for (int k = 0; k < warmup; ++k)
        if (!launchCeiling(images[k % count], rgbs[k % count], pattern, stream))
            return -1.0;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)
        return -1.0;
    (void)hipEventRecord(t0, stream);
    for (int k = 0; k < iters; ++k)
        if (!launchCeiling(images[k % count], rgbs[k % count], pattern, stream))
            return -1.0;
    (void)hipEventRecord(t1, stream);
    float ms = -1.0f;
    if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
        ms = -1.0f;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return ms < 0 ? -1.0 : (double)ms / iters;
}

extern "C" double avifhipTimeStreamCeiling(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || count == 0 || !images || !rgbs || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    const double a = timeCeilingPattern(count, images, rgbs, warmup, iters, 0, stream);
    const double b = timeCeilingPattern(count, images, rgbs, warmup, iters, 1, stream);
    if (a < 0 || b < 0)
        return -1.0;
    return a < b ? a : b;
}
