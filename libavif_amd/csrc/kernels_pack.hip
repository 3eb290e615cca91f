// kernels_pack.hip -- row packing for the file writers next to the reformat path (second half):
//   * Y4M frame payload (apps/shared/y4m.c:603-618): the planes Y, U, V (, A) written row by row without their pitch padding,
//     16-bit samples little-endian as stored;
//   * PNG row data (apps/shared/avifpng.c:865-880): the pixel rows without padding, 16-bit samples byte-swapped to big-endian
//     (what libpng's png_set_swap does to every row on the CPU before deflating it).
// One pass over device-resident data: destination = a linear byte stream, lane = 16 consecutive destination bytes.  Rows whose
// width, pitch and base are multiples of 16 bytes move as 16-byte loads and stores; anything else goes dword by dword with the
// row breaks resolved per dword.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace avifhip {
namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned swap16(unsigned v)
{
    return __builtin_amdgcn_perm(0u, v, 0x02030001u); // bytes 1 0 3 2
}

template <bool SWAP>
__global__ __launch_bounds__(256) void packRowsWideKernel(PackArgs A)
{
    // every row is a whole number of 16-byte groups: group g of row r
    const uint32_t groupsPerRow = A.widthBytes >> 4;
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= (uint64_t)groupsPerRow * A.rows)
        return;
    const uint32_t r = (uint32_t)(g / groupsPerRow), c = (uint32_t)(g - (uint64_t)r * groupsPerRow);
    u4 v = *reinterpret_cast<const u4 *>(A.src + (size_t)r * A.srcPitch + ((size_t)c << 4));
    if (SWAP)
        v = (u4) { swap16(v.x), swap16(v.y), swap16(v.z), swap16(v.w) };
    __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(A.dst + (size_t)r * A.dstPitch + ((size_t)c << 4)));
}

template <bool SWAP>
__global__ __launch_bounds__(256) void packRowsKernel(PackArgs A)
{
    // destination byte stream of rows * widthBytes bytes at dst (dstPitch == widthBytes), one dword per lane
    const uint64_t total = (uint64_t)A.widthBytes * A.rows;
    const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i >= total)
        return;
    unsigned v = 0;
    uint32_t r = (uint32_t)(i / A.widthBytes), c = (uint32_t)(i - (uint64_t)r * A.widthBytes);
    const uint32_t n = (total - i < 4) ? (uint32_t)(total - i) : 4u;
    for (uint32_t k = 0; k < n; ++k) {
        v |= (unsigned)A.src[(size_t)r * A.srcPitch + c] << (8 * k);
        if (++c == A.widthBytes)
            c = 0, ++r;
    }
    if (SWAP)
        v = swap16(v); // rows hold whole 16-bit samples and start on even stream offsets: a dword never splits a sample
    uint8_t * d = A.dst + i;
    if (n == 4 && (((uintptr_t)d) & 3u) == 0) {
        *reinterpret_cast<unsigned *>(d) = v;
    } else {
        for (uint32_t k = 0; k < n; ++k)
            d[k] = (uint8_t)(v >> (8 * k));
    }
}

// The other direction (api.cpp uploadRows): rows that arrived from the host as ONE block -- source pitch, width and base of any alignment --
// into rows at a 4-byte aligned destination pitch.  Lane = one destination dword, its source bytes read one by one.
__global__ __launch_bounds__(256) void unpackRowsKernel(PackArgs A)
{
    const uint32_t dwordsPerRow = (A.widthBytes + 3u) >> 2;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)dwordsPerRow * A.rows)
        return;
    const uint32_t r = (uint32_t)(i / dwordsPerRow), c = (uint32_t)(i - (uint64_t)r * dwordsPerRow) * 4u;
    const uint8_t * s = A.src + (size_t)r * A.srcPitch + c;
    uint8_t * d = A.dst + (size_t)r * A.dstPitch + c;
    const uint32_t n = A.widthBytes - c < 4u ? A.widthBytes - c : 4u;
    if (n == 4u) {
        *reinterpret_cast<unsigned *>(d) = (unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16) | ((unsigned)s[3] << 24);
    } else {
        for (uint32_t k = 0; k < n; ++k)
            d[k] = s[k];
    }
}

} // namespace

hipError_t launchUnpackRows(const PackArgs & A, hipStream_t stream)
{
    if (!A.rows || !A.widthBytes)
        return hipSuccess;
    if ((A.dstPitch & 3u) || ((uintptr_t)A.dst & 3u))
        return hipErrorInvalidValue;
    const uint64_t dwords = (uint64_t)((A.widthBytes + 3u) >> 2) * A.rows;
    hipLaunchKernelGGL(unpackRowsKernel, dim3((unsigned)((dwords + 255) / 256)), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launchPackRows(const PackArgs & A, hipStream_t stream)
{
    if (!A.rows || !A.widthBytes)
        return hipSuccess;
    const bool wide = (A.widthBytes & 15u) == 0 && (A.srcPitch & 15u) == 0 && (A.dstPitch & 15u) == 0 && (((uintptr_t)A.src | (uintptr_t)A.dst) & 15u) == 0;
    if (wide) {
        const uint64_t groups = (uint64_t)(A.widthBytes >> 4) * A.rows;
        const dim3 grid((unsigned)((groups + 255) / 256));
        if (A.swap16)
            hipLaunchKernelGGL(packRowsWideKernel<true>, grid, dim3(256), 0, stream, A);
        else
            hipLaunchKernelGGL(packRowsWideKernel<false>, grid, dim3(256), 0, stream, A);
    } else {
        const uint64_t dwords = ((uint64_t)A.widthBytes * A.rows + 3) / 4;
        const dim3 grid((unsigned)((dwords + 255) / 256));
        if (A.swap16)
            hipLaunchKernelGGL(packRowsKernel<true>, grid, dim3(256), 0, stream, A);
        else
            hipLaunchKernelGGL(packRowsKernel<false>, grid, dim3(256), 0, stream, A);
    }
    return hipGetLastError();
}

} // namespace avifhip
