// kernels_bench.hip -- measurement support, not part of the conversion path: a zero-arithmetic kernel that moves exactly the
// bytes of an 8-bit 4:2:0 -> 4-byte-pixel conversion (every plane sample read once, every output byte written once, same
// lane-to-byte mapping and store policy as the tiled kernels).  bench.py reports its duration beside the conversion kernel's:
// the chip's own ceiling for this byte movement (HBM3E sustains less than its 8 TB/s pin rate, and less again for a
// write-heavy mix), against which the conversion kernel's roofline fraction can be read.
#include <hip/hip_runtime.h>

#include "api_internal.h"

#include <vector>

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


// ---- the general mover (round 5): any plane layout of the tiled kernels, both directions, single jobs and batches --------------------------
// What a conversion of the job's shape has to move and nothing else: every sample of every plane the conversion reads (or writes) once, every
// pixel byte written (or read) once, lane = 4 pixels of every row of its wave, plane accesses of 4 samples per lane, pixel accesses of 16 bytes
// per lane at consecutive addresses (a wave instruction covers 1 KiB of a row), streaming stores -- the tiled kernels' own access shapes
// (tile_impl.h store4WideRgba, r2y_tile_impl.h loadStrip).  The bytes written are a mix of the bytes read, so that nothing can be elided.
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// four consecutive samples of a plane row: one 4-byte (8-bit samples) or 8-byte (16-bit containers) access, like tile_impl.h load4
template <int YB, bool NT = false>
__device__ __forceinline__ void moveLoad4(const uint8_t * p, unsigned (&w)[YB])
{
    if constexpr (YB == 1) {
        w[0] = NT ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p)) : *reinterpret_cast<const unsigned *>(p);
    } else {
        const u2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const u2 *>(p)) : *reinterpret_cast<const u2 *>(p);
        w[0] = t.x, w[1] = t.y;
    }
}
template <int YB>
__device__ __forceinline__ void moveStore4(uint8_t * p, const unsigned (&w)[YB], unsigned salt)
{
    if constexpr (YB == 1)
        __builtin_nontemporal_store(w[0] ^ salt, reinterpret_cast<unsigned *>(p));
    else
        __builtin_nontemporal_store((u2) { w[0] ^ salt, w[1] + salt }, reinterpret_cast<u2 *>(p));
}
struct MoveJob
{
    const uint8_t * plane[4]; // Y, U, V, A; nullptr = absent (A: not read by the conversion)
    uint8_t * pixels;
    uint32_t planePitch[4], pixelPitch;
    uint32_t w4, h2;
};

// `columns` != 0 (batches whose jobs are the tiles of ONE canvas, `columns` per canvas row): workgroups are numbered along the rows of the CANVAS
// -- grid x = (tile column of the grid, tile of the job's row), y = tile row inside the job, z = tile row of the grid -- so that the pixels of a
// canvas row leave in one sweep across all the jobs that share it, instead of job by job.
template <int YB, int PB, int WAVES_X, int RPW, bool BANDED, bool TO_RGB, bool NT = false>
__global__ __launch_bounds__(256) void streamMoveKernel(const MoveJob * __restrict__ jobs, uint32_t subX, uint32_t subY, uint32_t tilesX, uint32_t tilesPerJob, uint32_t columns)
{
    constexpr int WAVES_Y = 4 / WAVES_X;
    constexpr int VEC = PB / 4; // 16-byte pieces of a lane's 4 pixels
    uint32_t job = blockIdx.y, tile = BANDED ? bandedTile(blockIdx.x, gridDim.x) : blockIdx.x;
    if (columns) {
        const uint32_t c = blockIdx.x / tilesX;
        job = blockIdx.z * columns + c, tile = blockIdx.y * tilesX + (blockIdx.x - c * tilesX);
    }
    const MoveJob J = jobs[job];
    if (tile >= tilesPerJob)
        return;
    const uint32_t trow = tile / tilesX, tcol = tile - trow * tilesX;
    const uint32_t wave = threadIdx.y, wx = wave % WAVES_X, wy = wave / WAVES_X;
    const uint32_t band = tcol * WAVES_X + wx;
    const uint32_t X = band * 256u + 4u * threadIdx.x;
    const uint32_t Y0 = (trow * WAVES_Y + wy) * RPW;
    if (band * 256u >= J.w4 || Y0 >= J.h2)
        return;
    const bool laneValid = X < J.w4;
    const uint32_t Xc = laneValid ? X : 0u;
    const bool hasC = J.plane[1] != nullptr, hasA = J.plane[3] != nullptr;
    // lane l of pixel-store instruction h covers bytes [1024 * h + 16 * l, + 16) of the wave's row segment (4 * PB * 64 bytes)
    const uint32_t segBytes = (J.w4 - band * 256u < 256u ? J.w4 - band * 256u : 256u) * PB;
    if constexpr (TO_RGB) {
        unsigned ly[RPW][YB], lc[RPW][2][YB], la[RPW][YB];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const uint32_t Y = Y0 + r < J.h2 ? Y0 + r : J.h2 - 1;
            moveLoad4<YB, NT>(J.plane[0] + (size_t)Y * J.planePitch[0] + (size_t)Xc * YB, ly[r]);
#pragma unroll
            for (int k = 0; k < YB; ++k)
                la[r][k] = lc[r][0][k] = lc[r][1][k] = 0;
            if (hasA)
                moveLoad4<YB, NT>(J.plane[3] + (size_t)Y * J.planePitch[3] + (size_t)Xc * YB, la[r]);
            if (hasC && (!subY || !(r & 1))) { // (RPW is even and Y0 a multiple of it: r even <=> Y even)
                const uint32_t cy = Y >> subY;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint8_t * row = J.plane[1 + c] + (size_t)cy * J.planePitch[1 + c] + (size_t)(Xc >> subX) * YB;
                    if (subX) {
                        if constexpr (YB == 1)
                            lc[r][c][0] = *reinterpret_cast<const uint16_t *>(row);
                        else
                            lc[r][c][0] = *reinterpret_cast<const unsigned *>(row);
                    } else {
                        moveLoad4<YB, NT>(row, lc[r][c]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (Y0 + r >= J.h2)
                break;
            const int rc = subY ? (r & ~1) : r;
            unsigned mix = 0;
#pragma unroll
            for (int k = 0; k < YB; ++k)
                mix ^= ly[r][k] ^ la[r][k] ^ lc[rc][0][k] ^ (lc[rc][1][k] << 8);
            uint8_t * rowPx = J.pixels + (size_t)(Y0 + r) * J.pixelPitch + (size_t)band * 256u * PB;
#pragma unroll
            for (int h = 0; h < VEC; ++h) {
                const uint32_t byte = 1024u * h + 16u * threadIdx.x;
                if (byte < segBytes) {
                    const u4 o = { mix, mix + (unsigned)h, mix ^ 0x5a5a5a5au, mix + threadIdx.x };
                    __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(rowPx + byte));
                }
            }
        }
    } else {
        u4 px[RPW][VEC];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const uint32_t Y = Y0 + r < J.h2 ? Y0 + r : J.h2 - 1;
            const uint8_t * rowPx = J.pixels + (size_t)Y * J.pixelPitch + (size_t)band * 256u * PB;
#pragma unroll
            for (int h = 0; h < VEC; ++h) {
                const uint32_t byte = 1024u * h + 16u * threadIdx.x;
                const u4 * src = reinterpret_cast<const u4 *>(rowPx + (byte < segBytes ? byte : 0u));
                px[r][h] = NT ? __builtin_nontemporal_load(src) : *src;
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (Y0 + r >= J.h2 || !laneValid)
                break;
            unsigned mix[YB];
#pragma unroll
            for (int k = 0; k < YB; ++k) {
                mix[k] = 0;
#pragma unroll
                for (int h = 0; h < VEC; ++h)
                    mix[k] ^= px[r][h][k] ^ px[r][h][2 + k % 2];
            }
            const uint32_t Y = Y0 + r;
            moveStore4<YB>(const_cast<uint8_t *>(J.plane[0]) + (size_t)Y * J.planePitch[0] + (size_t)X * YB, mix, 0u);
            if (hasA)
                moveStore4<YB>(const_cast<uint8_t *>(J.plane[3]) + (size_t)Y * J.planePitch[3] + (size_t)X * YB, mix, 0xffu);
            if (hasC && (!subY || !(r & 1))) {
                const unsigned both = subY ? (mix[0] ^ px[r | 1][0][1]) : mix[0]; // 4:2:0: a chroma row is made of two pixel rows
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint8_t * row = const_cast<uint8_t *>(J.plane[1 + c]) + (size_t)(Y >> subY) * J.planePitch[1 + c] + (size_t)(X >> subX) * YB;
                    if (subX) {
                        if constexpr (YB == 1)
                            __builtin_nontemporal_store((uint16_t)(both >> (8 * c)), reinterpret_cast<uint16_t *>(row));
                        else
                            __builtin_nontemporal_store(both + (unsigned)c, reinterpret_cast<unsigned *>(row));
                    } else {
                        moveStore4<YB>(row, mix, 1u + (unsigned)c);
                    }
                }
            }
        }
    }
}

struct MoveShape
{
    int yb, pb;
    uint32_t subX, subY;
    uint32_t maxW4, maxH2;
    uint32_t columns; // jobs per canvas row when the jobs of a batch are the tiles of one canvas (row-major), else 0
};

// Fills `job` from an image / pixel pair; false when the pair is outside what the mover covers (the tiled kernels' own alignment rules).
// `alphaRead`: the conversion touches the alpha plane (TO_RGB: the plan reads it; encode: it writes one when the pixels carry alpha).
bool moveJobOf(const avifImage * image, const avifRGBImage * rgb, bool toRgb, MoveJob * job, MoveShape * shape)
{
    if (!image || !rgb || !image->yuvPlanes[0] || !rgb->pixels || rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return false;
    const int yb = image->depth > 8 ? 2 : 1, nch = rgbFormatChannelCount((int)rgb->format), pb = nch * (rgb->depth > 8 ? 2 : 1);
    if (pb != 4 && pb != 8)
        return false;
    const bool has444 = image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444, has400 = image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400;
    const uint32_t subX = (has444 || has400) ? 0u : 1u, subY = image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420 ? 1u : 0u;
    memset(job, 0, sizeof(*job));
    job->plane[0] = image->yuvPlanes[0], job->planePitch[0] = image->yuvRowBytes[0];
    if (!has400) {
        if (!image->yuvPlanes[1] || !image->yuvPlanes[2])
            return false;
        for (int c = 1; c <= 2; ++c)
            job->plane[c] = image->yuvPlanes[c], job->planePitch[c] = image->yuvRowBytes[c];
    }
    const bool pixelsCarryAlpha = nch == 4 && !rgb->ignoreAlpha;
    if (image->alphaPlane && image->alphaRowBytes && pixelsCarryAlpha)
        job->plane[3] = image->alphaPlane, job->planePitch[3] = image->alphaRowBytes;
    job->pixels = rgb->pixels, job->pixelPitch = rgb->rowBytes;
    job->w4 = image->width & ~3u, job->h2 = image->height & ~1u;
    if (!job->w4 || !job->h2 || ((uintptr_t)rgb->pixels & 15u) || (rgb->rowBytes & 15u))
        return false;
    for (int p = 0; p < 4; ++p) {
        if (!job->plane[p])
            continue;
        const uint32_t need = (p == 1 || p == 2) ? ((4u >> subX) * (uint32_t)yb) : 4u * (uint32_t)yb;
        if (((uintptr_t)job->plane[p] % need) || (job->planePitch[p] % need))
            return false;
    }
    if (shape->yb && (shape->yb != yb || shape->pb != pb || shape->subX != subX || shape->subY != subY))
        return false; // all jobs of a call share one shape
    shape->yb = yb, shape->pb = pb, shape->subX = subX, shape->subY = subY;
    shape->maxW4 = job->w4 > shape->maxW4 ? job->w4 : shape->maxW4, shape->maxH2 = job->h2 > shape->maxH2 ? job->h2 : shape->maxH2;
    return true;
}

// tile shapes / orders / load policies the mover knows: the ceiling of a shape is the fastest of them
constexpr int kMovePatterns = 13;
const char * const kMovePatternName[kMovePatterns] = { "1024x2 raster", "256x16 per-XCD bands", "1024x4 raster", "256x32 per-XCD bands", "512x4 raster",
                                                       "1024x2 raster, streaming loads", "256x16 per-XCD bands, streaming loads", "1024x4 raster, streaming loads",
                                                       "512x4 raster, streaming loads", "256x8 raster", "1024x4 along the canvas rows", "256x16 along the canvas rows",
                                                       "1024x4 along the canvas rows, streaming loads" };

template <int YB, int PB, bool TO_RGB>
bool launchMove(int pattern, const MoveJob * deviceJobs, uint32_t jobs, const MoveShape & s, hipStream_t stream)
{
    auto go = [&](auto kernel, uint32_t wavesX, uint32_t rpw, bool alongCanvas = false) {
        const uint32_t wavesY = 4u / wavesX;
        const uint32_t tilesX = (s.maxW4 + 256u * wavesX - 1) / (256u * wavesX), tilesY = (s.maxH2 + rpw * wavesY - 1) / (rpw * wavesY);
        if (alongCanvas) {
            if (!s.columns || jobs % s.columns)
                return false; // (separate buffers: there is no canvas)
            hipLaunchKernelGGL(kernel, dim3(tilesX * s.columns, tilesY, jobs / s.columns), dim3(64, 4), 0, stream, deviceJobs, s.subX, s.subY, tilesX, tilesX * tilesY, s.columns);
        } else {
            hipLaunchKernelGGL(kernel, dim3(tilesX * tilesY, jobs), dim3(64, 4), 0, stream, deviceJobs, s.subX, s.subY, tilesX, tilesX * tilesY, 0u);
        }
        return true;
    };
    switch (pattern) {
        case 0: return go(streamMoveKernel<YB, PB, 4, 2, false, TO_RGB>, 4, 2);
        case 1: return go(streamMoveKernel<YB, PB, 1, 4, true, TO_RGB>, 1, 4);
        case 2: return go(streamMoveKernel<YB, PB, 4, 4, false, TO_RGB>, 4, 4);
        case 3: return go(streamMoveKernel<YB, PB, 1, 8, true, TO_RGB>, 1, 8);
        case 4: return go(streamMoveKernel<YB, PB, 2, 2, false, TO_RGB>, 2, 2);
        case 5: return go(streamMoveKernel<YB, PB, 4, 2, false, TO_RGB, true>, 4, 2);
        case 6: return go(streamMoveKernel<YB, PB, 1, 4, true, TO_RGB, true>, 1, 4);
        case 7: return go(streamMoveKernel<YB, PB, 4, 4, false, TO_RGB, true>, 4, 4);
        case 8: return go(streamMoveKernel<YB, PB, 2, 2, false, TO_RGB, true>, 2, 2);
        case 9: return go(streamMoveKernel<YB, PB, 1, 2, false, TO_RGB>, 1, 2);
        case 10: return go(streamMoveKernel<YB, PB, 4, 4, false, TO_RGB>, 4, 4, true);
        case 11: return go(streamMoveKernel<YB, PB, 1, 4, false, TO_RGB>, 1, 4, true);
        default: return go(streamMoveKernel<YB, PB, 4, 4, false, TO_RGB, true>, 4, 4, true);
    }
}

// false: the pattern does not apply to this batch (nothing launched), or the launch failed (`failed` says which)
bool launchMoveShape(int pattern, bool toRgb, const MoveJob * deviceJobs, uint32_t jobs, const MoveShape & s, hipStream_t stream, bool * failed)
{
    bool launched;
    if (toRgb) {
        if (s.yb == 1)
            launched = s.pb == 4 ? launchMove<1, 4, true>(pattern, deviceJobs, jobs, s, stream) : launchMove<1, 8, true>(pattern, deviceJobs, jobs, s, stream);
        else
            launched = s.pb == 4 ? launchMove<2, 4, true>(pattern, deviceJobs, jobs, s, stream) : launchMove<2, 8, true>(pattern, deviceJobs, jobs, s, stream);
    } else {
        if (s.yb == 1)
            launched = s.pb == 4 ? launchMove<1, 4, false>(pattern, deviceJobs, jobs, s, stream) : launchMove<1, 8, false>(pattern, deviceJobs, jobs, s, stream);
        else
            launched = s.pb == 4 ? launchMove<2, 4, false>(pattern, deviceJobs, jobs, s, stream) : launchMove<2, 8, false>(pattern, deviceJobs, jobs, s, stream);
    }
    *failed = launched && hipGetLastError() != hipSuccess;
    return launched && !*failed;
}

} // namespace
} // namespace avifhip

using namespace avifhip;
using namespace avifhip::api;

// Average milliseconds per launch of the byte-movement-only kernel, cycling over `count` device-resident frames like
// avifhipTimeYUVToRGBCycle (the RGB buffers are overwritten with meaningless bytes); the faster of the two access patterns
// above.  Negative when the frames are not 8-bit 4:2:0 planes with 4-byte pixels.
// `warmup` untimed launches, then `iters` between two events on the launch stream: milliseconds per launch (< 0 on failure)
template <typename Launch>
static double timeCalls(void * hipStream, int warmup, int iters, Launch launch)
{
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (!launch(stream))
            return -1.0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (hipEventCreate(&t0) != hipSuccess)
        return -1.0;
    if (hipEventCreate(&t1) != hipSuccess) {
        (void)hipEventDestroy(t0);
        return -1.0;
    }
    bool ok = hipEventRecord(t0, stream) == hipSuccess;
    for (int k = 0; k < iters && ok; ++k)
        ok = launch(stream);
    float ms = -1.0f;
    ok = ok && hipEventRecord(t1, stream) == hipSuccess && hipEventSynchronize(t1) == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return ok ? (double)ms / iters : -1.0;
}

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

// the general mover over `groups` launches' worth of jobs, `perLaunch` jobs each (launch k moves group k % groups): the fastest of its patterns
static double timeMover(const std::vector<MoveJob> & jobs, uint32_t groups, uint32_t perLaunch, const MoveShape & shape, bool toRgb, int warmup, int iters, hipStream_t stream)
{
    MoveJob * deviceJobs = nullptr;
    if (hipMalloc(&deviceJobs, jobs.size() * sizeof(MoveJob)) != hipSuccess || hipMemcpy(deviceJobs, jobs.data(), jobs.size() * sizeof(MoveJob), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (deviceJobs)
            (void)hipFree(deviceJobs);
        return -1.0;
    }
    double best = -1.0;
    int bestPattern = -1;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool ok = hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
    for (int pattern = 0; ok && pattern < kMovePatterns; ++pattern) {
        bool failed = false;
        if (!launchMoveShape(pattern, toRgb, deviceJobs, perLaunch, shape, stream, &failed)) {
            ok = !failed;
            continue; // (the pattern does not apply to this batch)
        }
        for (int k = 0; ok && k < warmup; ++k)
            ok = launchMoveShape(pattern, toRgb, deviceJobs + (size_t)(k % groups) * perLaunch, perLaunch, shape, stream, &failed);
        ok = ok && hipEventRecord(t0, stream) == hipSuccess;
        for (int k = 0; ok && k < iters; ++k)
            ok = launchMoveShape(pattern, toRgb, deviceJobs + (size_t)(k % groups) * perLaunch, perLaunch, shape, stream, &failed);
        float ms = -1.0f;
        ok = ok && hipEventRecord(t1, stream) == hipSuccess && hipEventSynchronize(t1) == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess;
        if (ok && (best < 0 || ms / iters < best))
            best = (double)ms / iters, bestPattern = pattern;
        if (ok && getenv("AVIFHIP_CEILING_TRACE")) // every pattern's time, not just the fastest one's (which access shape costs what)
            fprintf(stderr, "avifhip ceiling: %-48s %8.3f us\n", kMovePatternName[pattern], 1e3 * ms / iters);
    }
    if (t0)
        (void)hipEventDestroy(t0);
    if (t1)
        (void)hipEventDestroy(t1);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(deviceJobs);
    if (!ok) {
        (void)hipGetLastError();
        return -1.0;
    }
    static thread_local char name[96];
    snprintf(name, sizeof(name), "stream_ceiling<%s,%s,%s>", toRgb ? "planes->pixels" : "pixels->planes", perLaunch > 1 ? "batch" : "single", kMovePatternName[bestPattern]);
    tls.lastKernel = name;
    return best;
}

static double timeCeilingGeneral(uint32_t count, const avifImage * const * images, const avifRGBImage * const * rgbs, bool toRgb, bool batch, int warmup, int iters, void * hipStream,
                                 uint32_t perLaunch = 0)
{
    if (perLaunch && (perLaunch > count || count % perLaunch))
        return -1.0;
    if (iters <= 0 || count == 0 || !images || !rgbs || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    std::vector<MoveJob> jobs(count);
    MoveShape shape;
    memset(&shape, 0, sizeof(shape));
    for (uint32_t k = 0; k < count; ++k) {
        if (!moveJobOf(images[k], rgbs[k], toRgb, &jobs[k], &shape)) {
            setError("avifhipTimeStreamCeiling*: job %u is outside the mover's shapes (4- or 8-byte pixels, planes and pixels aligned like the tiled kernels ask, one shape per call)", k);
            return -1.0;
        }
        if (toRgb && jobs[k].plane[3]) {
            // the conversion reads the alpha plane only when the plan says so (alpha into the pixels, or pending alpha arithmetic)
            YuvToRgbPlan plan;
            if (makeYuvToRgbPlan(images[k], rgbs[k], nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan) != AVIF_RESULT_OK)
                return -1.0;
            if (!(plan.alphaSource == ALPHA_PLANE || plan.inLoopMul != MUL_NONE || plan.postMul != MUL_NONE))
                jobs[k].plane[3] = nullptr;
        }
    }
    if (batch && count > 1) {
        // the tiles of one canvas, row-major: the leading jobs whose pixels start inside the first job's first pixel row share a canvas row
        uint32_t columns = 1;
        while (columns < count && jobs[columns].pixelPitch == jobs[0].pixelPitch && jobs[columns].pixels > jobs[0].pixels && jobs[columns].pixels < jobs[0].pixels + jobs[0].pixelPitch)
            ++columns;
        shape.columns = (columns > 1 && count % columns == 0) ? columns : 0;
    }
    if (perLaunch) // a sequence walked perLaunch frames at a time (avifhipTimeYUVToRGBBatchCycle)
        return timeMover(jobs, count / perLaunch, perLaunch, shape, toRgb, warmup, iters, pickStream(hipStream));
    return batch ? timeMover(jobs, 1, count, shape, toRgb, warmup, iters, pickStream(hipStream)) : timeMover(jobs, count, 1, shape, toRgb, warmup, iters, pickStream(hipStream));
}

extern "C" double avifhipTimeStreamCeiling(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || count == 0 || !images || !rgbs || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    // 8-bit 4:2:0 planes into 4-byte pixels without an alpha plane: the headline's shape keeps rounds 2-4's kernel (comparable figures)
    bool headlineShape = true;
    for (uint32_t k = 0; k < count && headlineShape; ++k)
        headlineShape = images[k] && rgbs[k] && images[k]->depth == 8 && images[k]->yuvFormat == AVIF_PIXEL_FORMAT_YUV420 && rgbs[k]->depth == 8 &&
                        rgbFormatChannelCount((int)rgbs[k]->format) == 4 && !(images[k]->alphaPlane && !rgbs[k]->ignoreAlpha);
    if (!headlineShape)
        return timeCeilingGeneral(count, images, rgbs, true, false, warmup, iters, hipStream);
    const double a = timeCeilingPattern(count, images, rgbs, warmup, iters, 0, stream);
    const double b = timeCeilingPattern(count, images, rgbs, warmup, iters, 1, stream);
    if (a < 0 || b < 0)
        return -1.0;
    tls.lastKernel = a < b ? "stream_ceiling<planes->pixels,single,1024x2 raster>" : "stream_ceiling<planes->pixels,single,256x16 per-XCD bands>";
    return a < b ? a : b;
}

// ---- plane scaling: every sample of the source planes read once, every sample of the destination planes written once ----
namespace {
struct PlanePair
{
    const uint8_t * src;
    uint8_t * dst;
    uint32_t srcPitch, dstPitch, srcWidthBytes, dstWidthBytes, srcRows, dstRows;
};
struct ScaleMove
{
    PlanePair plane[4];
    uint32_t planes;
};
// one workgroup = 256 lanes x 16 bytes of ROWS consecutive rows of one plane, source rows first and destination rows after them (grid y = plane)
template <int ROWS>
__global__ __launch_bounds__(256) void scaleMoveKernel(ScaleMove M, uint32_t chunksPerRow)
{
    const PlanePair P = M.plane[blockIdx.y];
    const uint32_t group = blockIdx.x / chunksPerRow, chunk = blockIdx.x - group * chunksPerRow;
    const uint32_t byte = (chunk * 256u + threadIdx.x) * 16u;
    u4 v[ROWS];
    unsigned seen = 0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const uint32_t row = group * ROWS + k;
        v[k] = (u4) { byte, row, chunk, 0x5a5a5a5au };
        if (row < P.srcRows && byte + 16u <= P.srcWidthBytes)
            v[k] = *reinterpret_cast<const u4 *>(P.src + (size_t)row * P.srcPitch + byte);
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const uint32_t row = group * ROWS + k;
        if (row < P.srcRows) {
            seen ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        } else if (row - P.srcRows < P.dstRows && byte + 16u <= P.dstWidthBytes) {
            __builtin_nontemporal_store(v[k], reinterpret_cast<u4 *>(P.dst + (size_t)(row - P.srcRows) * P.dstPitch + byte));
        }
    }
    if (seen == 0x9e3779b9u && byte + 16u <= P.dstWidthBytes) // (never, in effect: the loads must not be optimised away)
        __builtin_nontemporal_store(v[0], reinterpret_cast<u4 *>(P.dst + byte));
}
} // namespace

extern "C" double avifhipTimeStreamCeilingScale(const avifImage * src, avifImage * dst, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || !src || !dst || src->depth != dst->depth || src->yuvFormat != dst->yuvFormat || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    ScaleMove M;
    memset(&M, 0, sizeof(M));
    const PlaneGeometry gs = planeGeometry(src), gd = planeGeometry(dst);
    uint32_t maxRows = 0, maxWidth = 0;
    for (int p = 0; p < 4; ++p) {
        const uint8_t * sp = (p < 3) ? src->yuvPlanes[p] : src->alphaPlane;
        uint8_t * dp = (p < 3) ? dst->yuvPlanes[p] : dst->alphaPlane;
        const uint32_t sPitch = (p < 3) ? src->yuvRowBytes[p] : src->alphaRowBytes, dPitch = (p < 3) ? dst->yuvRowBytes[p] : dst->alphaRowBytes;
        if (!sp || !dp || ((p == 1 || p == 2) && src->yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
            continue;
        if ((((uintptr_t)sp | (uintptr_t)dp | sPitch | dPitch) & 15u) != 0) {
            setError("avifhipTimeStreamCeilingScale: planes and pitches must be multiples of 16 bytes");
            return -1.0;
        }
        PlanePair & P = M.plane[M.planes++];
        P.src = sp, P.dst = dp, P.srcPitch = sPitch, P.dstPitch = dPitch;
        P.srcWidthBytes = gs.widthBytes[p] & ~15u, P.dstWidthBytes = gd.widthBytes[p] & ~15u, P.srcRows = gs.rows[p], P.dstRows = gd.rows[p];
        maxRows = P.srcRows + P.dstRows > maxRows ? P.srcRows + P.dstRows : maxRows;
        maxWidth = P.srcWidthBytes > maxWidth ? P.srcWidthBytes : maxWidth;
        maxWidth = P.dstWidthBytes > maxWidth ? P.dstWidthBytes : maxWidth;
    }
    if (!M.planes || !maxWidth)
        return -1.0;
    const uint32_t chunksPerRow = (maxWidth + 4095u) / 4096u;
    double ms = -1.0;
    for (int rowsPerGroup : { 1, 4, 8 }) { // the fastest of three depths of requests in flight per lane
        const double t = timeCalls(hipStream, warmup, iters, [&](hipStream_t s) {
            const dim3 grid(chunksPerRow * ((maxRows + rowsPerGroup - 1) / rowsPerGroup), M.planes);
            if (rowsPerGroup == 1)
                hipLaunchKernelGGL(scaleMoveKernel<1>, grid, dim3(256), 0, s, M, chunksPerRow);
            else if (rowsPerGroup == 4)
                hipLaunchKernelGGL(scaleMoveKernel<4>, grid, dim3(256), 0, s, M, chunksPerRow);
            else
                hipLaunchKernelGGL(scaleMoveKernel<8>, grid, dim3(256), 0, s, M, chunksPerRow);
            return hipGetLastError() == hipSuccess;
        });
        if (t < 0)
            return -1.0;
        ms = (ms < 0 || t < ms) ? t : ms;
    }
    if (ms >= 0)
        tls.lastKernel = "stream_ceiling<scale: source planes read, destination planes written>";
    return ms;
}

extern "C" double avifhipTimeStreamCeilingRGBToYUV(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    return timeCeilingGeneral(count, images, rgbs, false, false, warmup, iters, hipStream);
}

extern "C" double avifhipTimeStreamCeilingBatch(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    return timeCeilingGeneral(count, images, rgbs, true, true, warmup, iters, hipStream);
}

extern "C" double avifhipTimeStreamCeilingBatchCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                     int iters, void * hipStream)
{
    return timeCeilingGeneral(count, images, rgbs, true, false, warmup, iters, hipStream, perLaunch ? perLaunch : 1);
}

extern "C" double avifhipTimeStreamCeilingRGBToYUVBatchCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                             int iters, void * hipStream)
{
    return timeCeilingGeneral(count, images, rgbs, false, false, warmup, iters, hipStream, perLaunch ? perLaunch : 1);
}
