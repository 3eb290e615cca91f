// r2y_tile_shared.h -- host-visible description of a tiled RGB -> YUV launch (shared by the dispatch code in
// kernels_r2y_tile.hip and the instantiation units).
#pragma once

#include <hip/hip_runtime.h>

#include <string.h>

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
    uint32_t xcdBands; // workgroups of one XCD (blockIdx % 8) take a contiguous run of tiles (r2yBlockOf)
    uint32_t identity; // identity matrix: the planes are G, B, R, each quantised on luma's scale (rangeUV / biasUV hold luma's)
    // the other matrices without coefficients (src/reformat.c:368-381): MODE_YCGCO (three adds on the normalised channels) and
    // MODE_YCGCO_RE / _RO (integer lifting on the channel codes, then "/ range" in the verified reciprocal form); MODE_COEFF otherwise
    int32_t matrixMode;
    RcpHL rcpRangeY, rcpRangeUV;
    float rgbMaxF;
    // pending alpha (un)multiply, applied to the normalised channels before the matrix (src/reformat.c:325-358); 4-channel sources only
    int32_t mulMode;
    // fixed-point kernels (libyuv's 8-bit BT.601 arithmetic), coefficients per MEMORY-order colour
    // channel (c0 = first colour byte, c1 = G, c2 = third colour byte), so that no channel swap is needed:
    //   Y = (y0*c0 + y1*c1 + y2*c2 + yBias) >> 8,  U = (u0*m0 + u1*m1 + u2*m2 + 0x8000) >> 8,  V likewise,
    //   m = the per-channel average of the covered pixels, (a+b+c+d+2) >> 2 or (a+b+1) >> 1
    struct Fx
    {
        int32_t y0, y1, y2, yBias;
        int32_t u0, u1, u2;
        int32_t v0, v1, v2;
    } fx;
};

// Sequences (round 6, like tile_shared.h SeqFrames in the other direction): up to kR2YSeqMaxFrames frames that differ in their buffers only,
// encoded by ONE launch of the single-image kernels -- grid z = frame, the frame's five addresses taken from the kernel arguments.
// A single image is a sequence of one.
constexpr uint32_t kR2YSeqMaxFrames = 8;
struct R2YSeqFrames
{
    struct Frame
    {
        const uint8_t * rgb;
        uint8_t *y, *u, *v, *a;
    } f[kR2YSeqMaxFrames];
};
inline void r2ySeqSetFrame(R2YSeqFrames & S, uint32_t k, const R2YArgs & A)
{
    S.f[k].rgb = A.rgb, S.f[k].y = A.y, S.f[k].u = A.u, S.f[k].v = A.v, S.f[k].a = A.a;
}
inline R2YSeqFrames r2ySeqOfOne(const R2YArgs & A)
{
    R2YSeqFrames S;
    for (uint32_t k = 0; k < kR2YSeqMaxFrames; ++k)
        r2ySeqSetFrame(S, k, A);
    return S;
}
// everything but the five addresses agrees (both filled by the same code from a zeroed struct: padding included)
inline bool r2ySeqCompatible(const R2YArgs & a, const R2YArgs & b)
{
    R2YArgs x = a, y = b;
    x.rgb = y.rgb = nullptr, x.y = y.y = nullptr, x.u = y.u = nullptr, x.v = y.v = nullptr, x.a = y.a = nullptr;
    return (a.u == nullptr) == (b.u == nullptr) && (a.v == nullptr) == (b.v == nullptr) && (a.a == nullptr) == (b.a == nullptr) && memcmp(&x, &y, sizeof(R2YArgs)) == 0;
}

struct R2YKey
{
    bool fixedPoint; // libyuv arithmetic: 8-bit RGB -> 8-bit planes
    bool wideRgb, wideYuv;
    int nch, sub;
    bool hasMul; // pending alpha (un)multiply (kernel name only: a wave-uniform branch inside the kernels)
};

// gray sources (r2y_tile_impl.h grayToYuvTileKernel)
struct GrayArgs
{
    const uint8_t * gray; // source pixels
    uint8_t * y;
    uint8_t * a;
    uint32_t grayPitch, yPitch, aPitch;
    uint32_t wP, height; // columns converted here (a multiple of the pixels per lane) and rows
    RcpHL rcpRgbMax;
    float rangeY, biasY, yuvMaxF;
    uint32_t yuvMax;
    uint32_t alphaFirst; // AGRAY
    int32_t alphaMode;   // R2Y_ALPHA_*
    int32_t mulMode;
};
hipError_t launchR2YGray8(int grayChannels, bool wideYuv, const GrayArgs & args, hipStream_t stream);
hipError_t launchR2YGray16(int grayChannels, bool wideYuv, const GrayArgs & args, hipStream_t stream);

// (`frames`: the launch's frames -- r2ySeqOfOne(args) for a single image -- and how many of them: grid z)
hipError_t launchR2YTileRgb8(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & frames, uint32_t frameCount);
hipError_t launchR2YTileRgb16(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & frames, uint32_t frameCount);
hipError_t launchR2YTileFx(const R2YKey & key, const R2YArgs & args, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & frames, uint32_t frameCount);

} // namespace r2y
} // namespace avifhip
