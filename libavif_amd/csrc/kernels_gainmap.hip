// kernels_gainmap.hip -- gain-map application (avifRGBImageApplyGainMap, reference src/gainmap.c:73-315), one lane per
// pixel.  Every libm call of the reference is a table here (gainmap_plan.h explains why that is exact): the base samples'
// linear light and the gain-map samples' gains are looked up, the tone-mapping arithmetic in between is the reference's
// fp32 / fp64 multiply-adds in its order (contraction off), and the output transfer function + quantisation is a binary
// search among the function's fp32 steps (in LDS up to 12-bit outputs).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace avifhip {

namespace {

constexpr float kF16Multiplier = 1.9259299444e-34f; // src/reformat.c:1411
constexpr uint32_t kStepsInLds = 8192; // both pieces of a 12-bit output

__device__ __forceinline__ float f16ToFloat(uint32_t code) // avifF16ToFloat, src/reformat.c:1849-1854
{
    return __uint_as_float(code << 13) / kF16Multiplier;
}
__device__ __forceinline__ uint32_t floatToF16(float v) // avifFloatToF16, :1842-1847
{
    return (__float_as_uint(v * kF16Multiplier) >> 13) & 0xffffu;
}

// the three colour sample codes and alpha of pixel p (avifGetRGBAPixel, :1856-1897); alpha as the float the reference carries
__device__ __forceinline__ void readPixel(const uint8_t * p, const GainMapPixelLayout & L, uint32_t code[3], float & alpha)
{
    if (L.channelBytes > 1) {
        code[0] = *reinterpret_cast<const uint16_t *>(p + L.offR), code[1] = *reinterpret_cast<const uint16_t *>(p + L.offG);
        code[2] = *reinterpret_cast<const uint16_t *>(p + L.offB);
        const uint32_t a = L.hasAlpha ? *reinterpret_cast<const uint16_t *>(p + L.offA) : ((1u << L.depth) - 1);
        alpha = L.isFloat ? (L.hasAlpha ? f16ToFloat(a) : 1.0f) : (float)a / L.maxF;
    } else if (L.is565) {
        const uint32_t v = *reinterpret_cast<const uint16_t *>(p);
        const uint32_t r5 = (v >> 11) & 0x1f, g6 = (v >> 5) & 0x3f, b5 = v & 0x1f;
        code[0] = ((r5 << 3) | (r5 >> 2)) & 0xff, code[1] = ((g6 << 2) | (g6 >> 4)) & 0xff, code[2] = ((b5 << 3) | (b5 >> 2)) & 0xff;
        alpha = 1.0f;
    } else {
        code[0] = p[L.offR], code[1] = p[L.offG], code[2] = p[L.offB];
        alpha = L.hasAlpha ? (float)p[L.offA] / L.maxF : 1.0f;
    }
}

// (T)(0.5f + v * max) of avifSetRGBAPixel, :1920-1937; an out-of-range product converts like the reference's compiled code
// does on x86-64 (through int32, then truncated to the container)
__device__ __forceinline__ uint32_t quantise(float v, const GainMapPixelLayout & L)
{
    if (L.isFloat)
        return floatToF16(v);
    const uint32_t mask = (L.channelBytes > 1) ? 0xffffu : 0xffu;
    return (uint32_t)(int32_t)(0.5f + (v * L.maxF)) & mask;
}

__device__ __forceinline__ void writePixel(uint8_t * p, const GainMapPixelLayout & L, const uint32_t code[3], uint32_t alphaCode)
{
    if (L.channelBytes > 1) {
        *reinterpret_cast<uint16_t *>(p + L.offR) = (uint16_t)code[0], *reinterpret_cast<uint16_t *>(p + L.offG) = (uint16_t)code[1];
        *reinterpret_cast<uint16_t *>(p + L.offB) = (uint16_t)code[2];
        if (L.hasAlpha)
            *reinterpret_cast<uint16_t *>(p + L.offA) = (uint16_t)alphaCode;
    } else if (L.is565) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)((code[2] >> 3) | ((code[1] >> 2) << 5) | ((code[0] >> 3) << 11)); // :619-633
    } else {
        p[L.offR] = (uint8_t)code[0], p[L.offG] = (uint8_t)code[1], p[L.offB] = (uint8_t)code[2];
        if (L.hasAlpha)
            p[L.offA] = (uint8_t)alphaCode;
    }
}

// avifLinearRGBConvertColorSpace, src/colrconvert.c:186-195: fp64 products and sums in the reference's order, rounded to fp32
__device__ __forceinline__ void convertPrimaries(float v[3], const double M[9])
{
    const double x = v[0], y = v[1], z = v[2];
    const double r0 = M[0] * x + M[1] * y + M[2] * z, r1 = M[3] * x + M[4] * y + M[5] * z, r2 = M[6] * x + M[7] * y + M[8] * z;
    v[0] = (float)r0, v[1] = (float)r1, v[2] = (float)r2;
}

// the output code of linear value x: the largest k with steps[k] <= x in the piece (x < 0, x >= 0) x belongs to
__device__ __forceinline__ uint32_t codeOf(float x, const float * steps, uint32_t maxCode, uint32_t nanCode)
{
    if (x != x)
        return nanCode;
    if (!(x < 0.0f))
        steps += maxCode + 1;
    uint32_t lo = 0, hi = maxCode;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (steps[mid] <= x)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void gainMapApplyKernel(GainMapArgs A)
{
    __shared__ float ldsSteps[kStepsInLds];
    const bool stepsInLds = A.convert && (2 * (A.maxCode + 1) <= kStepsInLds);
    if (stepsInLds) {
        for (uint32_t k = threadIdx.y * 64 + threadIdx.x; k < 2 * (A.maxCode + 1); k += 256)
            ldsSteps[k] = A.steps[k];
        __syncthreads();
    }
    const float * steps = stepsInLds ? ldsSteps : A.steps;

    const uint32_t i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    const bool inside = i < A.width && j < A.height;
    float pixelMax = 0.0f, toneMax = 0.0f;
    bool sawNan = false;
    if (inside) {
        uint32_t code[3];
        float alpha;
        readPixel(A.base + (size_t)j * A.basePitch + (size_t)i * A.baseL.pixelBytes, A.baseL, code, alpha);
        uint32_t outCode[3];
        if (!A.convert) { // :155-166 without a change of transfer function or primaries
#pragma unroll
            for (int c = 0; c < 3; ++c)
                outCode[c] = quantise(A.baseL.isFloat ? f16ToFloat(code[c]) : (float)code[c] / A.baseL.maxF, A.outL);
        } else {
            float v[3] = { A.baseLut[code[0]], A.baseLut[code[1]], A.baseLut[code[2]] };
            if (A.inConv)
                convertPrimaries(v, A.inM);
            if (A.gain) { // :236-270
                const uint8_t * g = A.gain + (size_t)j * A.gainPitch + (size_t)i * 4 * ((A.gainDepth > 8) ? 2 : 1);
                const uint32_t n = 1u << A.gainDepth;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t gc = (A.gainDepth > 8) ? reinterpret_cast<const uint16_t *>(g)[c] : g[c];
                    const float tone = (v[c] + A.baseOffset[c]) * A.gainLut[c * n + min(gc, n - 1)] - A.altOffset[c];
                    if (tone > toneMax)
                        toneMax = tone;
                    if (tone > pixelMax)
                        pixelMax = tone;
                    v[c] = tone;
                }
                if (A.outConv)
                    convertPrimaries(v, A.outM);
                sawNan = (v[0] != v[0]) || (v[1] != v[1]) || (v[2] != v[2]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                outCode[c] = codeOf(v[c], steps, A.maxCode, A.nanCode);
        }
        writePixel(A.out + (size_t)j * A.outPitch + (size_t)i * A.outL.pixelBytes, A.outL, outCode, A.outL.hasAlpha ? quantise(alpha, A.outL) : 0);
    }
    if (A.gain) { // wave-level reduction, then one atomic per wave and statistic
        double sum = (double)pixelMax;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            toneMax = fmaxf(toneMax, __shfl_xor(toneMax, m));
            sum += __shfl_xor(sum, m);
        }
        const unsigned long long nanLanes = __ballot(sawNan);
        if (threadIdx.x == 0) {
            atomicMax(&A.stats->maxBits, __float_as_uint(toneMax));
            atomicAdd(&A.stats->sum, sum);
            if (nanLanes)
                atomicOr(&A.stats->nan, 1);
        }
    }
}

} // namespace

hipError_t launchGainMapApply(const GainMapArgs & A, hipStream_t stream)
{
    if (!A.width || !A.height)
        return hipSuccess;
    const dim3 grid((A.width + 63) / 64, (A.height + 3) / 4), block(64, 4);
    hipLaunchKernelGGL(gainMapApplyKernel, grid, block, 0, stream, A);
    return hipGetLastError();
}

} // namespace avifhip
