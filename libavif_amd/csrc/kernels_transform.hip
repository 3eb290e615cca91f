// kernels_transform.hip -- clean-aperture crop, rotation and mirroring of an interleaved pixel buffer as one permutation
// pass (apps/shared/avifutil.c:667-825: avifRGBImageSetViewRect + avifRGBImageRotate + avifRGBImageMirror, which the
// reference runs as a view, a full copy and an in-place swap pass).  Pure byte movement: 2 x pixelBytes per pixel of HBM
// traffic.  One workgroup moves a 32 x 32 tile of destination pixels; quarter-turn rotations go through an LDS tile so
// that both the loads (along source rows) and the stores (along destination rows) are coalesced.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace avifhip {

namespace {

constexpr int naturalAlign(int px)
{
    return (px % 8 == 0) ? 8 : (px % 4 == 0) ? 4 : (px % 2 == 0) ? 2 : 1;
}
// one pixel, moved as a unit; ALIGNED = the buffers allow the pixel's natural alignment (else byte-granular moves)
template <int PX, bool ALIGNED>
struct alignas(ALIGNED ? naturalAlign(PX) : 1) Pixel
{
    uint8_t b[PX];
};

// destination pixel (x, y) -> pixel (i, j) of the cropped source
__device__ __forceinline__ void sourceOf(const TransformArgs & A, uint32_t x, uint32_t y, uint32_t * i, uint32_t * j)
{
    // undo the mirror (applied last, :819-825), then the rotation (:711-740)
    if (A.mirror == 1)
        x = A.dw - 1 - x;
    else if (A.mirror == 0)
        y = A.dh - 1 - y;
    switch (A.angle) {
        case 1: *i = A.cw - 1 - y, *j = x; break;             // source (i, j) went to (j, cw - 1 - i)
        case 2: *i = A.cw - 1 - x, *j = A.ch - 1 - y; break;  // ... to (cw - 1 - i, ch - 1 - j)
        case 3: *i = y, *j = A.ch - 1 - x; break;             // ... to (ch - 1 - j, i)
        default: *i = x, *j = y; break;
    }
}

template <int PX, bool ALIGNED>
__global__ __launch_bounds__(256) void transformRowsKernel(TransformArgs A)
{
    // angle 0 / 2: destination rows are source rows (possibly reversed): one lane per pixel is already coalesced
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.dw || y >= A.dh)
        return;
    uint32_t i, j;
    sourceOf(A, x, y, &i, &j);
    *reinterpret_cast<Pixel<PX, ALIGNED> *>(A.dst + (size_t)y * A.dstPitch + (size_t)x * PX) =
        *reinterpret_cast<const Pixel<PX, ALIGNED> *>(A.src + (size_t)j * A.srcPitch + (size_t)i * PX);
}

// angle 0 / 2 with 4-byte-aligned 4- or 8-byte pixels: 16 bytes per lane and access.  A destination group of N = 16 / PX
// consecutive pixels is a run of N consecutive source pixels, read forwards or (mirrored / half turn) backwards.
template <int PX>
__global__ __launch_bounds__(256) void transformRowsWideKernel(TransformArgs A)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    constexpr uint32_t N = 16 / PX;
    const uint32_t x = (blockIdx.x * 64 + threadIdx.x) * N, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.dw || y >= A.dh)
        return;
    uint8_t * dst = A.dst + (size_t)y * A.dstPitch + (size_t)x * PX;
    if (x + N > A.dw) { // the row's last, partial group: pixel by pixel
        for (uint32_t k = 0; x + k < A.dw; ++k) {
            uint32_t i, j;
            sourceOf(A, x + k, y, &i, &j);
            *reinterpret_cast<Pixel<PX, true> *>(dst + k * PX) = *reinterpret_cast<const Pixel<PX, true> *>(A.src + (size_t)j * A.srcPitch + (size_t)i * PX);
        }
        return;
    }
    uint32_t i0, j0, i1, j1;
    sourceOf(A, x, y, &i0, &j0);
    sourceOf(A, x + N - 1, y, &i1, &j1); // same source row; i1 = i0 + N - 1 or i0 - (N - 1)
    const bool reversed = i1 < i0;
    const u4v v = *reinterpret_cast<const u4v *>(A.src + (size_t)j0 * A.srcPitch + (size_t)(reversed ? i1 : i0) * PX);
    u4v o = v;
    if (reversed) {
        if constexpr (PX == 4)
            o = (u4v) { v.w, v.z, v.y, v.x };
        else
            o = (u4v) { v.z, v.w, v.x, v.y };
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u4v *>(dst));
}

template <int PX, bool ALIGNED>
__global__ __launch_bounds__(256) void transformTransposeKernel(TransformArgs A)
{
    __shared__ Pixel<PX, ALIGNED> tile[32][33];
    const uint32_t X0 = blockIdx.x * 32, Y0 = blockIdx.y * 32;
    const uint32_t X1 = min(X0 + 32, A.dw) - 1, Y1 = min(Y0 + 32, A.dh) - 1;
    // the destination tile is the image of a 32 x 32 source tile: its origin is the smaller corner
    uint32_t ia, ja, ib, jb;
    sourceOf(A, X0, Y0, &ia, &ja);
    sourceOf(A, X1, Y1, &ib, &jb);
    const uint32_t i0 = min(ia, ib), j0 = min(ja, jb);
    const uint32_t tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + tx, j = j0 + ty + 8 * k;
        if (i < A.cw && j < A.ch)
            tile[ty + 8 * k][tx] = *reinterpret_cast<const Pixel<PX, ALIGNED> *>(A.src + (size_t)j * A.srcPitch + (size_t)i * PX);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = X0 + tx, y = Y0 + ty + 8 * k;
        if (x < A.dw && y < A.dh) {
            uint32_t i, j;
            sourceOf(A, x, y, &i, &j);
            *reinterpret_cast<Pixel<PX, ALIGNED> *>(A.dst + (size_t)y * A.dstPitch + (size_t)x * PX) = tile[j - j0][i - i0];
        }
    }
}

// Quarter turns of 4- or 8-byte pixels with 16 bytes per lane on BOTH sides (the 32 x 32 kernel above moves one pixel per lane and access:
// 128-byte row pieces, a vector-memory instruction per 256 bytes).  A workgroup takes the source tile of 64 columns x 16 N rows
// (N = 16 / PX pixels per 16 bytes) behind a destination tile of 16 N columns x 64 rows: 1024 16-byte loads along source rows into an LDS
// tile whose 16-byte chunks are XOR-swizzled by the row group, then every lane gathers the N pixels of one destination chunk column-wise
// (two lanes per bank) and stores them: a wave instruction writes 4 destination rows of 256 contiguous bytes.
template <int PX>
__global__ __launch_bounds__(256) void transformTransposeWideKernel(TransformArgs A)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr uint32_t N = 16 / PX, W = PX / 4;          // pixels per chunk, dwords per pixel
    constexpr uint32_t TW = 16 * N, TH = 64;             // destination tile
    constexpr uint32_t SR = TW, CH = 64 / N;             // source tile: SR rows of CH chunks (64 pixels)
    __shared__ __attribute__((aligned(16))) u4v tile[SR * CH]; // 16 KiB
    const uint32_t X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    const uint32_t X1 = min(X0 + TW, A.dw) - 1, Y1 = min(Y0 + TH, A.dh) - 1;
    uint32_t ia, ja, ib, jb;
    sourceOf(A, X0, Y0, &ia, &ja);
    sourceOf(A, X1, Y1, &ib, &jb);
    const uint32_t i0 = min(ia, ib), j0 = min(ja, jb);
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t idx = t + 256 * k, jj = idx / CH, c = idx % CH;
        const uint32_t i = i0 + N * c, j = j0 + jj;
        u4v v = { 0, 0, 0, 0 };
        if (j < A.ch && i < A.cw) {
            const uint8_t * p = A.src + (size_t)j * A.srcPitch + (size_t)i * PX;
            if (i + N <= A.cw) {
                v = *reinterpret_cast<const u4u *>(p);
            } else { // the row's last, partial chunk
                for (uint32_t q = 0; i + q < A.cw; ++q)
                    for (uint32_t w = 0; w < W; ++w)
                        v[q * W + w] = reinterpret_cast<const uint32_t *>(p)[q * W + w];
            }
        }
        tile[jj * CH + (c ^ ((jj / N) & (CH - 1)))] = v;
    }
    __syncthreads();
    const uint32_t * words = reinterpret_cast<const uint32_t *>(tile);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t idx = t + 256 * k, yy = idx / 16, gx = idx % 16;
        const uint32_t x = X0 + N * gx, y = Y0 + yy;
        if (x >= A.dw || y >= A.dh)
            continue;
        u4v o = { 0, 0, 0, 0 };
#pragma unroll
        for (uint32_t q = 0; q < N; ++q) {
            uint32_t i, j;
            sourceOf(A, min(x + q, A.dw - 1), y, &i, &j);
            const uint32_t jj = j - j0, ii = i - i0;
            const uint32_t chunk = jj * CH + ((ii / N) ^ ((jj / N) & (CH - 1)));
#pragma unroll
            for (uint32_t w = 0; w < W; ++w)
                o[q * W + w] = words[chunk * 4 + (ii % N) * W + w];
        }
        uint8_t * d = A.dst + (size_t)y * A.dstPitch + (size_t)x * PX;
        if (x + N <= A.dw) {
            __builtin_nontemporal_store(o, reinterpret_cast<u4v *>(d));
        } else {
            for (uint32_t q = 0; x + q < A.dw; ++q)
                for (uint32_t w = 0; w < W; ++w)
                    reinterpret_cast<uint32_t *>(d)[q * W + w] = o[q * W + w];
        }
    }
}

template <int PX, bool ALIGNED>
hipError_t launchFor(const TransformArgs & A, hipStream_t stream)
{
    if (A.angle == 1 || A.angle == 3) {
        if constexpr ((PX == 4 || PX == 8) && ALIGNED) {
            if (((uintptr_t)A.dst % 16) == 0 && (A.dstPitch % 16) == 0 && ((uintptr_t)A.src % 4) == 0 && (A.srcPitch % 4) == 0) {
                constexpr uint32_t TW = 16 * (16 / PX);
                hipLaunchKernelGGL((transformTransposeWideKernel<PX>), dim3((A.dw + TW - 1) / TW, (A.dh + 63) / 64), dim3(256), 0, stream, A);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((transformTransposeKernel<PX, ALIGNED>), dim3((A.dw + 31) / 32, (A.dh + 31) / 32), dim3(32, 8), 0, stream, A);
    } else if constexpr ((PX == 4 || PX == 8) && ALIGNED) {
        // 16-byte stores need a 16-byte aligned destination; loads only the pixels' own 4-byte alignment
        if (((uintptr_t)A.dst % 16) == 0 && (A.dstPitch % 16) == 0 && ((uintptr_t)A.src % 4) == 0 && (A.srcPitch % 4) == 0) {
            constexpr uint32_t N = 16 / PX;
            hipLaunchKernelGGL((transformRowsWideKernel<PX>), dim3(((A.dw + N - 1) / N + 63) / 64, (A.dh + 3) / 4), dim3(64, 4), 0, stream, A);
        } else {
            hipLaunchKernelGGL((transformRowsKernel<PX, ALIGNED>), dim3((A.dw + 63) / 64, (A.dh + 3) / 4), dim3(64, 4), 0, stream, A);
        }
    } else {
        hipLaunchKernelGGL((transformRowsKernel<PX, ALIGNED>), dim3((A.dw + 63) / 64, (A.dh + 3) / 4), dim3(64, 4), 0, stream, A);
    }
    return hipGetLastError();
}

template <int PX>
hipError_t launchPx(const TransformArgs & A, hipStream_t stream)
{
    constexpr uintptr_t a = (uintptr_t)naturalAlign(PX);
    const bool aligned = ((uintptr_t)A.src % a) == 0 && ((uintptr_t)A.dst % a) == 0 && (A.srcPitch % a) == 0 && (A.dstPitch % a) == 0;
    return aligned ? launchFor<PX, true>(A, stream) : launchFor<PX, false>(A, stream);
}

} // namespace

hipError_t launchRgbTransform(const TransformArgs & A, uint32_t pixelBytes, hipStream_t stream)
{
    if (A.dw == 0 || A.dh == 0)
        return hipSuccess;
    switch (pixelBytes) { // gray 8/16, RGB565, RGB 8/16, RGBA 8/16
        case 1: return launchPx<1>(A, stream);
        case 2: return launchPx<2>(A, stream);
        case 3: return launchPx<3>(A, stream);
        case 4: return launchPx<4>(A, stream);
        case 6: return launchPx<6>(A, stream);
        case 8: return launchPx<8>(A, stream);
        default: return hipErrorInvalidValue;
    }
}

} // namespace avifhip
