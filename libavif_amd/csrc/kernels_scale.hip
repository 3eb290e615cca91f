// kernels_scale.hip -- plane scaling (avifImageScale, reference src/scale.c:23-201): one lane per destination sample,
// evaluated from the host-derived schedule (scale_plan.h).  Sample arithmetic of the vendored libyuv scaler: 7-bit column
// blend for 8-bit samples (scale_common.c:192-195), 16-bit column blend for 16-bit samples (:254-258), 8-bit row fractions
// (InterpolateRow_C, row_common.c:44-104), box sums with the reference's accumulator widths and fixed-point reciprocal
// (scale.c:29-150), 9:3:3:1 for the 2x upsamplers (scale_common.c:29-112).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace avifhip {

namespace {

template <bool WIDE>
__device__ __forceinline__ int sampleOf(const uint8_t * plane, uint32_t rowBytes, int x, int y)
{
    const uint8_t * p = plane + (size_t)y * rowBytes;
    return WIDE ? (int)reinterpret_cast<const uint16_t *>(p)[x] : (int)p[x];
}
template <bool WIDE>
__device__ __forceinline__ int blendColumns(int a, int b, int f)
{
    if (!WIDE)
        return (int)(uint8_t)(a + ((((f >> 9) * (b - a)) + 0x40) >> 7));
    return (int)(uint16_t)(a + (int)((((int64_t)f * ((int64_t)b - a)) + 0x8000) >> 16));
}
__device__ __forceinline__ int blendRows(int a, int b, int yf)
{
    if (yf == 0)
        return a;
    if (yf == 128)
        return (a + b + 1) >> 1;
    return (a * (256 - yf) + b * yf + 128) >> 8;
}

// one destination sample
template <bool WIDE>
__device__ __forceinline__ int scaledSample(const ScaleArgs & A, int ca, int cb, int ra, int rb, int rf)
{
    switch (A.mode) {
        case SCALE_POINT_MODE:
            return sampleOf<WIDE>(A.src, A.srcPitch, ca, ra);
        case SCALE_DOWN_MODE: {
            const int c1 = min(ca + 1, A.srcW - 1);
            const int v0 = blendRows(sampleOf<WIDE>(A.src, A.srcPitch, ca, ra), sampleOf<WIDE>(A.src, A.srcPitch, ca, rb), rf);
            const int v1 = blendRows(sampleOf<WIDE>(A.src, A.srcPitch, c1, ra), sampleOf<WIDE>(A.src, A.srcPitch, c1, rb), rf);
            return blendColumns<WIDE>(v0, v1, cb);
        }
        case SCALE_UP_MODE: {
            const int c1 = min(ca + 1, A.srcW - 1);
            const int h0 = blendColumns<WIDE>(sampleOf<WIDE>(A.src, A.srcPitch, ca, ra), sampleOf<WIDE>(A.src, A.srcPitch, c1, ra), cb);
            const int h1 = blendColumns<WIDE>(sampleOf<WIDE>(A.src, A.srcPitch, ca, rb), sampleOf<WIDE>(A.src, A.srcPitch, c1, rb), cb);
            return blendRows(h0, h1, rf);
        }
        case SCALE_BOX_MODE: {
            // rows are accumulated per column in uint16_t (8-bit samples: wraps like ScaleAddRow_C's row buffer) or uint32_t
            uint32_t sum = 0;
            for (int c = 0; c < cb; ++c) {
                uint32_t colSum = 0;
                for (int r = 0; r < rb; ++r)
                    colSum += (uint32_t)sampleOf<WIDE>(A.src, A.srcPitch, ca + c, ra + r);
                sum += WIDE ? colSum : (colSum & 0xffffu);
            }
            const uint32_t scale = (uint32_t)(65536 / (max(cb, 1) * rb));
            return WIDE ? (int)(uint16_t)((sum * scale) >> 16) : (int)(uint8_t)((sum * scale) >> 16);
        }
        default: { // 2x upsamplers
            const int nn = sampleOf<WIDE>(A.src, A.srcPitch, ca, ra), nf = sampleOf<WIDE>(A.src, A.srcPitch, cb, ra);
            const int fn = sampleOf<WIDE>(A.src, A.srcPitch, ca, rb), ff = sampleOf<WIDE>(A.src, A.srcPitch, cb, rb);
            return (9 * nn + 3 * nf + 3 * fn + ff + 8) >> 4;
        }
    }
}

// One lane per destination sample.  (Four samples per lane with one 4-byte store were measured 1.5-1.8x SLOWER: the source
// gathers of neighbouring lanes stop sharing cache lines; the kernel is bound by its gathers, not by its byte stores.)
template <bool WIDE>
__global__ __launch_bounds__(256) void scalePlaneKernel(ScaleArgs A)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= A.dstW || j >= A.dstH)
        return;
    const int out = scaledSample<WIDE>(A, A.colA[i], A.colB[i], A.rowA[j], A.rowB[j], A.rowF[j]);
    uint8_t * d = A.dst + (size_t)j * A.dstPitch;
    if (WIDE)
        reinterpret_cast<uint16_t *>(d)[i] = (uint16_t)out;
    else
        d[i] = (uint8_t)out;
}

} // namespace

hipError_t launchScalePlane(const ScaleArgs & A, bool wide, hipStream_t stream)
{
    if (A.dstW <= 0 || A.dstH <= 0)
        return hipSuccess;
    const dim3 grid((A.dstW + 63) / 64, (A.dstH + 3) / 4), block(64, 4);
    if (wide)
        hipLaunchKernelGGL(scalePlaneKernel<true>, grid, block, 0, stream, A);
    else
        hipLaunchKernelGGL(scalePlaneKernel<false>, grid, block, 0, stream, A);
    return hipGetLastError();
}

} // namespace avifhip
