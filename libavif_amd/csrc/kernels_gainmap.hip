// kernels_gainmap.hip -- gain-map application (avifRGBImageApplyGainMap, reference src/gainmap.c:73-315), one lane per
// pixel.  Every libm call of the reference is a table here (gainmap_plan.h explains why that is exact): the base samples'
// linear light and the gain-map samples' gains are looked up, the tone-mapping arithmetic in between is the reference's
// fp32 / fp64 multiply-adds in its order (contraction off), and the output transfer function + quantisation is a binary
// search among the function's fp32 steps (in LDS up to 12-bit outputs).
#include <hip/hip_runtime.h>

#include <atomic>

#include "gainmap_steps.h"
#include "kernels.h"
#include "pixel_math.h"

#include <type_traits>

namespace avifhip {

namespace {

constexpr float kF16Multiplier = 1.9259299444e-34f; // src/reformat.c:1411

__device__ __forceinline__ float f16ToFloat(uint32_t code) // avifF16ToFloat, src/reformat.c:1849-1854
{
    return __uint_as_float(code << 13) / kF16Multiplier;
}
__device__ __forceinline__ uint32_t floatToF16(float v) // avifFloatToF16, :1842-1847
{
    return (__float_as_uint(v * kF16Multiplier) >> 13) & 0xffffu;
}

// the three colour sample codes and alpha of pixel p (avifGetRGBAPixel, :1856-1897); alpha as the float the reference carries.
// `vector`: 4-channel pixels at naturally aligned addresses are moved with one 4- or 8-byte access (the per-channel
// accesses of the general path cost one vector-memory instruction each)
__device__ __forceinline__ void readPixel(const uint8_t * p, const GainMapPixelLayout & L, bool vector, uint32_t code[3], float & alpha)
{
    if (vector && L.pixelBytes == 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
        code[0] = (w >> (8 * L.offR)) & 0xff, code[1] = (w >> (8 * L.offG)) & 0xff, code[2] = (w >> (8 * L.offB)) & 0xff;
        alpha = (float)((w >> (8 * L.offA)) & 0xff) / L.maxF;
    } else if (vector && L.pixelBytes == 8) {
        const uint2 w = *reinterpret_cast<const uint2 *>(p);
        auto pick = [&](uint32_t off) -> uint32_t { return (((off & 4) ? w.y : w.x) >> (8 * (off & 3))) & 0xffff; };
        code[0] = pick(L.offR), code[1] = pick(L.offG), code[2] = pick(L.offB);
        const uint32_t a = pick(L.offA);
        alpha = L.isFloat ? f16ToFloat(a) : (float)a / L.maxF;
    } else if (L.channelBytes > 1) {
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

__device__ __forceinline__ void writePixel(uint8_t * p, const GainMapPixelLayout & L, bool vector, const uint32_t code[3], uint32_t alphaCode)
{
    if (vector && L.pixelBytes == 4) {
        *reinterpret_cast<uint32_t *>(p) = ((code[0] & 0xff) << (8 * L.offR)) | ((code[1] & 0xff) << (8 * L.offG)) | ((code[2] & 0xff) << (8 * L.offB)) |
                                           ((alphaCode & 0xff) << (8 * L.offA));
    } else if (vector && L.pixelBytes == 8) {
        uint2 w = { 0, 0 };
        auto place = [&](uint32_t off, uint32_t v) {
            const uint32_t s = (v & 0xffff) << (8 * (off & 3));
            if (off & 4)
                w.y |= s;
            else
                w.x |= s;
        };
        place(L.offR, code[0]), place(L.offG, code[1]), place(L.offB, code[2]), place(L.offA, alphaCode);
        *reinterpret_cast<uint2 *>(p) = w;
    } else if (L.channelBytes > 1) {
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

// The output codes of three linear values: for each the largest k with steps[k] <= x in the piece (x < 0, x >= 0) x belongs
// to (each piece has `entries` steps, steps[0] = -inf).  The guide brackets the answer for x >= 0 (64 buckets per octave of
// x: one or two codes wide for the log-like curves, a dozen at the top of a linear one), a bisection inside the bracket
// finishes; the three channels' chains of table reads advance together.
struct StepSearch
{
    const float * steps;
    const uint16_t * guide;
    uint32_t entries, maxCode, nanCode, firstBits, shift, buckets;
};
__device__ __forceinline__ void codesOf(const float x[3], const StepSearch & S, uint32_t code[3])
{
    uint32_t lo[3], hi[3], base[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t bits = __float_as_uint(x[c]);
        if (x[c] < 0.0f || x[c] != x[c]) { // the x < 0 piece (BT.1361, IEC 61966-2-4 only reach above code 0 there): whole range
            base[c] = 0, lo[c] = 0, hi[c] = S.maxCode;
        } else {
            base[c] = S.entries;
            const uint32_t b = (bits - S.firstBits) >> S.shift;
            if (bits < S.firstBits)
                lo[c] = 0, hi[c] = S.guide[0];
            else if (b >= S.buckets)
                lo[c] = S.guide[S.buckets], hi[c] = S.maxCode;
            else
                lo[c] = S.guide[b], hi[c] = S.guide[b + 1];
        }
    }
    while ((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2])) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t mid = (lo[c] + hi[c] + 1) >> 1; // lo == hi: mid == lo, steps[lo] <= x holds, nothing changes
            const bool up = S.steps[base[c] + mid] <= x[c];
            lo[c] = (lo[c] < hi[c] && up) ? mid : lo[c];
            hi[c] = (lo[c] < hi[c] && !up) ? mid - 1 : hi[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        code[c] = (x[c] != x[c]) ? S.nanCode : lo[c];
}

// Statistics of a launch (rgbMaxLinear, rgbSumLinear, "a NaN appeared": src/gainmap.c:257-287) from every lane's own: wave reduction
// -> workgroup reduction through LDS -> one partial per workgroup, stored straight into pinned host memory; the caller adds them up in
// index order once the stream has drained.  (One atomic per wave on a single address cost 3 ms on a 4K image; a ticket counter --
// one atomic per workgroup -- so that the last workgroup could do the sum, 78 us of a 110 us kernel: atomics on one address are
// serialised across the eight L2s.)  WAVES: waves per workgroup (64 x WAVES threads).
template <int WAVES>
__device__ __forceinline__ void finishStatistics(const GainMapArgs & A, float toneMax, double sum, unsigned long long nanLanes, uint8_t * scratch)
{
    // `scratch`: 16 * WAVES bytes of LDS, 8-byte aligned, that no lane of the workgroup still reads as something else
    double * const waveSums = reinterpret_cast<double *>(scratch);
    float * const waveMaxima = reinterpret_cast<float *>(waveSums + WAVES);
    uint32_t * const waveNan = reinterpret_cast<uint32_t *>(waveMaxima + WAVES);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        toneMax = fmaxf(toneMax, __shfl_xor(toneMax, m));
        sum += __shfl_xor(sum, m);
    }
    if (threadIdx.x == 0)
        waveMaxima[threadIdx.y] = toneMax, waveSums[threadIdx.y] = sum, waveNan[threadIdx.y] = nanLanes ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        GainMapPartial mine = { waveSums[0], waveMaxima[0], waveNan[0] };
#pragma unroll
        for (int w = 1; w < WAVES; ++w)
            mine.sum += waveSums[w], mine.max = fmaxf(mine.max, waveMaxima[w]), mine.nan |= waveNan[w];
        A.partials[blockIdx.x] = mine;
    }
}

// The one-read locator of the output code (gainmap_plan.h: GainMapSteps::locator), for tables in LDS at byte address L.base + ...
struct Locator
{
    int32_t first, last;  // the bit patterns the first bucket starts and the last bucket ends at (in every lane: v_med3_i32 takes one scalar)
    uint32_t shift, base; // bucket width; LDS address of the table minus 4 x (first >> shift), modulo 2^32
};
__device__ __forceinline__ uint32_t locate(float x, const Locator & L)
{
    // as signed integers the bit patterns of negative values sort below those of every x >= 0, NaNs of either sign beyond the ends
    uint32_t b;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(b) : "v"(__float_as_uint(x)), "v"(L.first), "v"(L.last));
    // `first` is a multiple of the bucket width: the offset inside the bucket is that of the pattern itself
    const uint32_t e = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)(((b >> L.shift) << 2) + L.base);
    // the code in the low 12 bits, the entry's threshold field still above them: whoever packs the code takes its low byte (8-bit
    // outputs) or its low 16 bits (deeper ones: the host keeps the bucket width at 2^16 or less there, so bits 12-15 are clear)
    return e + (((b << (32 - L.shift)) > e) ? 1u : 0u);
}

// Persistent workgroups walking tiles of 64 x 4 pixels with a grid stride.  LDS_TABLES: the three tables (steps, base lookup,
// gain lookup) are copied to LDS once per workgroup and addressed as LDS -- a pointer that may be either LDS or global memory
// compiles to flat loads, which the searches cannot afford; the host picks this variant when everything fits (api_gainmap.cpp).
// TABLES: 0 -- tables in global memory; 1 -- steps, guide, base and gain tables in LDS; 2 -- base and gain tables and the LOCATOR in LDS
// (integer outputs up to 12 bits whose curve has one: no search, as in the fast kernel below, which serves the 4-channel layouts)
template <int TABLES>
__global__ __launch_bounds__(256) void gainMapApplyKernel(GainMapArgs A, uint32_t tilesX, uint32_t tiles)
{
    constexpr bool LDS_TABLES = TABLES == 1;
    extern __shared__ float ldsTables[];
    const float * steps = A.steps;
    const float * baseLut = A.baseLut;
    const float * gainLut = A.gainLut;
    const uint16_t * guide = A.guide;
    Locator L = { 0, 0, A.locShift, 0 };
    if constexpr (TABLES == 2) {
        const uint32_t t = threadIdx.y * 64 + threadIdx.x;
        for (uint32_t k = t; k < A.ldsBaseLut; k += 256)
            ldsTables[k] = A.baseLut[k];
        for (uint32_t k = t; k < A.ldsGainLut; k += 256)
            ldsTables[A.ldsBaseLut + k] = A.gainLut[k];
        for (uint32_t k = t; k < A.locBuckets; k += 256)
            reinterpret_cast<uint32_t *>(ldsTables)[A.ldsBaseLut + A.ldsGainLut + k] = A.locator[k];
        __syncthreads();
        baseLut = ldsTables, gainLut = ldsTables + A.ldsBaseLut;
        // (LDS addresses: the dynamic block follows the kernel's static LDS)
        L.base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)(ldsTables + A.ldsBaseLut + A.ldsGainLut) - 4 * (A.locFirstBits >> A.locShift);
        asm volatile("v_mov_b32 %0, %1" : "=v"(L.first) : "s"(A.locFirstBits));
        asm volatile("v_mov_b32 %0, %1" : "=v"(L.last) : "s"(A.locFirstBits + ((A.locBuckets << A.locShift) - 1)));
    }
    if constexpr (LDS_TABLES) {
        const uint32_t t = threadIdx.y * 64 + threadIdx.x;
        for (uint32_t k = t; k < A.ldsSteps; k += 256)
            ldsTables[k] = A.steps[k];
        for (uint32_t k = t; k < A.ldsBaseLut; k += 256)
            ldsTables[A.ldsSteps + k] = A.baseLut[k];
        for (uint32_t k = t; k < A.ldsGainLut; k += 256)
            ldsTables[A.ldsSteps + A.ldsBaseLut + k] = A.gainLut[k];
        uint16_t * ldsGuide = reinterpret_cast<uint16_t *>(ldsTables + A.ldsSteps + A.ldsBaseLut + A.ldsGainLut);
        for (uint32_t k = t; k <= A.guideBuckets; k += 256)
            ldsGuide[k] = A.guide[k];
        __syncthreads();
        steps = ldsTables, baseLut = ldsTables + A.ldsSteps, gainLut = ldsTables + A.ldsSteps + A.ldsBaseLut;
        guide = ldsGuide;
    }
    const StepSearch search = { steps, guide, A.stepEntries, A.maxCode, A.nanCode, A.guideFirstBits, A.guideShift, A.guideBuckets };
    const bool baseVector = A.baseL.hasAlpha && (((uintptr_t)A.base | A.basePitch) & (A.baseL.pixelBytes - 1)) == 0;
    const bool outVector = A.outL.hasAlpha && (((uintptr_t)A.out | A.outPitch) & (A.outL.pixelBytes - 1)) == 0;
    const uint32_t gainPixelBytes = 4 * ((A.gainDepth > 8) ? 2 : 1);
    const bool gainVector = A.gain && (((uintptr_t)A.gain | A.gainPitch) & (gainPixelBytes - 1)) == 0;

    float toneMax = 0.0f; // statistics of this lane over all its pixels
    double sum = 0.0;
    bool sawNan = false;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t i = (tile % tilesX) * 64 + threadIdx.x, j = (tile / tilesX) * 4 + threadIdx.y;
        if (i >= A.width || j >= A.height)
            continue;
        uint32_t code[3];
        float alpha;
        readPixel(A.base + (size_t)j * A.basePitch + (size_t)i * A.baseL.pixelBytes, A.baseL, baseVector, code, alpha);
        uint32_t outCode[3];
        if (!A.convert) { // :155-166 without a change of transfer function or primaries
#pragma unroll
            for (int c = 0; c < 3; ++c)
                outCode[c] = quantise(A.baseL.isFloat ? f16ToFloat(code[c]) : (float)code[c] / A.baseL.maxF, A.outL);
        } else {
            float v[3] = { baseLut[code[0]], baseLut[code[1]], baseLut[code[2]] };
            if (A.inConv)
                convertPrimaries(v, A.inM);
            if (A.gain) { // :236-270
                const uint8_t * g = A.gain + (size_t)j * A.gainPitch + (size_t)i * gainPixelBytes;
                const uint32_t n = 1u << A.gainDepth;
                uint32_t gcode[3];
                if (gainVector && A.gainDepth <= 8) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(g);
                    gcode[0] = w & 0xff, gcode[1] = (w >> 8) & 0xff, gcode[2] = (w >> 16) & 0xff;
                } else if (gainVector) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(g);
                    gcode[0] = w.x & 0xffff, gcode[1] = w.x >> 16, gcode[2] = w.y & 0xffff;
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        gcode[c] = (A.gainDepth > 8) ? reinterpret_cast<const uint16_t *>(g)[c] : g[c];
                }
                float pixelMax = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float tone = (v[c] + A.baseOffset[c]) * gainLut[c * n + min(gcode[c], n - 1)] - A.altOffset[c];
                    if (tone > toneMax)
                        toneMax = tone;
                    if (tone > pixelMax)
                        pixelMax = tone;
                    v[c] = tone;
                }
                sum += (double)pixelMax;
                if (A.pixelMax)
                    A.pixelMax[(size_t)j * A.width + i] = pixelMax;
                if (A.outConv)
                    convertPrimaries(v, A.outM);
                sawNan = sawNan || (v[0] != v[0]) || (v[1] != v[1]) || (v[2] != v[2]);
            }
            if constexpr (TABLES == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) // (the low 12 bits: the entry's threshold field rides above them)
                    outCode[c] = (v[c] != v[c]) ? A.nanCode : (locate(v[c], L) & 0xfffu);
            } else {
                codesOf(v, search, outCode);
            }
        }
        writePixel(A.out + (size_t)j * A.outPitch + (size_t)i * A.outL.pixelBytes, A.outL, outVector, outCode, A.outL.hasAlpha ? quantise(alpha, A.outL) : 0);
    }
    if (A.gain) {
        __shared__ __attribute__((aligned(8))) uint8_t statistics[16 * 4];
        finishStatistics<4>(A, toneMax, sum, __ballot(sawNan), statistics);
    }
}

// ---- the fast kernel ---------------------------------------------------------------------------------------------------
// 4-channel integer pixels on both sides (4 or 8 bytes each), a gain map, all tables in LDS at fixed places (host: api_gainmap.cpp
// decides).  Four neighbouring pixels per lane: 16- / 32-byte accesses.  Per pixel: one byte permutation brings the base pixel into R, G, B, A
// order whatever its layout (v_perm_b32 with a wave-uniform selector), three reads of the base table, three of the gain table, ONE read
// of the locator per channel for the output code (gainmap_plan.h: GainMapSteps::locator -- no search), one of the alpha table, one
// or two permutations into the output layout; in between the reference's fp32 / fp64 arithmetic in its order.  The kernel does not look
// for NaNs (AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE, src/gainmap.c:277-281): the host sends a call here only when the tables and coefficients
// bound every intermediate value below FLT_MAX -- no infinity, hence no NaN; the general kernel serves the rest.
// Sample codes above the image's depth (garbage in a 16-bit container) read whatever lies at that place of the LDS.

// LDS layout, in bytes.  With 4-byte pixels (8-bit samples) every table base is a compile-time constant of the instantiation (it rides in
// the offset field of the ds_read); 16-bit base samples get room for 12 bits; gain-map tables of more than 8 bits follow the locator, at
// a place and with a stride the launch knows (one more instruction per read).
template <int BASE_BYTES, int GAIN_BYTES>
struct FastLds
{
    static constexpr uint32_t kBaseEntries = (BASE_BYTES == 4) ? 256 : 4096;
    // GAIN_BYTES < 4: the gain map arrives as 8-bit planes and the kernel converts them itself -- 0 in the reference's fp32 arithmetic
    // (normalised luma and chroma of every sample code behind the gain tables), 1 in libyuv's fixed point (the same layout, those two unused)
    static constexpr bool kPlanes = GAIN_BYTES < 4;
    static constexpr uint32_t kBaseLut = 0, kAlphaLut = kBaseLut + 4 * kBaseEntries, kGainLut = kAlphaLut + 2 * kBaseEntries,
                              kNormY = kGainLut + 3 * 4 * 256, kNormUV = kNormY + 4 * 256,
                              kLocator = kGainLut + ((GAIN_BYTES == 4) ? 3 * 4 * 256 : kPlanes ? 5 * 4 * 256 : 0);
};

// kFastPixels neighbouring pixels of BYTES each, moved 16 bytes at a time (4-byte alignment is all the wide accesses need).  Pixels are read
// and written once: streaming accesses keep them from displacing one another in the L2.
constexpr int kFastPixels = 4; // per lane
#ifndef AVIFHIP_GAINMAP_ROWS
#define AVIFHIP_GAINMAP_ROWS 8
#endif
constexpr int kFastRows = AVIFHIP_GAINMAP_ROWS; // = waves per workgroup: 256 x 8 pixels per workgroup and step
#ifndef AVIFHIP_GAINMAP_DEPTH
#define AVIFHIP_GAINMAP_DEPTH 2 // tiles in flight per wave, the one being worked on included (3, round 6: 89 registers instead of 80, no faster -- 24.2 / 24.3 us on one box)
#endif
#ifndef AVIFHIP_GAINMAP_NT_LOADS
#define AVIFHIP_GAINMAP_NT_LOADS 0
#endif
#ifndef AVIFHIP_GAINMAP_NT_STORES
#define AVIFHIP_GAINMAP_NT_STORES 1
#endif
typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef Quad QuadA4 __attribute__((aligned(4)));
template <int BYTES>
struct PixelRun
{
    uint32_t w[kFastPixels * BYTES / 4];
    __device__ __forceinline__ void load(const uint8_t * p)
    {
#pragma unroll
        for (int k = 0; k < kFastPixels * BYTES / 16; ++k) {
            const QuadA4 * q = reinterpret_cast<const QuadA4 *>(p) + k;
            const Quad v = AVIFHIP_GAINMAP_NT_LOADS ? __builtin_nontemporal_load(q) : *q;
            w[4 * k] = v.x, w[4 * k + 1] = v.y, w[4 * k + 2] = v.z, w[4 * k + 3] = v.w;
        }
    }
    __device__ __forceinline__ void store(uint8_t * p) const
    {
#pragma unroll
        for (int k = 0; k < kFastPixels * BYTES / 16; ++k) {
            QuadA4 * q = reinterpret_cast<QuadA4 *>(p) + k;
            const Quad v = { w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
            if (AVIFHIP_GAINMAP_NT_STORES)
                __builtin_nontemporal_store(v, q);
            else
                *q = v;
        }
    }
};

// four neighbouring samples of each plane of an 8-bit gain map (the kernel converts them itself)
struct PlaneRun
{
    uint32_t y, u, v;
    // (a run that ends a row whose width is no multiple of four starts at any byte)
    static __device__ __forceinline__ uint32_t four(const uint8_t * p)
    {
        if (__builtin_expect(((uintptr_t)p & 3) == 0, 1))
            return *reinterpret_cast<const uint32_t *>(p);
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
};

// (code & 0xff) << 2 in one instruction: the compiler finds the sub-dword operand for bytes 1-3 but not for byte 0
__device__ __forceinline__ uint32_t byte0Times4(uint32_t w, uint32_t two)
{
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    return r;
}
__device__ __forceinline__ uint32_t word0Times4(uint32_t w, uint32_t two)
{
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(two), "v"(w));
    return r;
}

// CONV: bit 0 -- the base image's primaries differ from the gain-map math's (inM), bit 1 -- the output's do (outM)
template <int BASE_BYTES, int OUT_BYTES, int GAIN_BYTES, int CONV>
__global__ __launch_bounds__(64 * kFastRows) void gainMapApplyFastKernel(GainMapArgs A, uint32_t tilesX, uint32_t tiles, uint32_t stepX, uint32_t stepY)
{
    using Lds = FastLds<BASE_BYTES, GAIN_BYTES>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t gainR = (GAIN_BYTES != 8) ? Lds::kGainLut : Lds::kLocator + 16 * ((A.locBuckets + 3) / 4);
    const uint32_t gainG = gainR + ((GAIN_BYTES != 8) ? 1024 : (4u << A.gainDepth)), gainB = gainG + ((GAIN_BYTES != 8) ? 1024 : (4u << A.gainDepth));
    Locator L = { 0, 0, A.locShift, Lds::kLocator - 4 * (A.locFirstBits >> A.locShift) };
    // (the two bounds in vector registers for the whole kernel: left to itself the compiler copies them over from scalar ones at every use)
    asm volatile("v_mov_b32 %0, %1" : "=v"(L.first) : "s"(A.locFirstBits));
    asm volatile("v_mov_b32 %0, %1" : "=v"(L.last) : "s"(A.locFirstBits + ((A.locBuckets << A.locShift) - 1)));
    const uint32_t selBaseX = A.selBase[0], selBaseY = A.selBase[1], selOutX = A.selOut[0], selOutY = A.selOut[1];
    const float bo0 = A.baseOffset[0], bo1 = A.baseOffset[1], bo2 = A.baseOffset[2];
    const float ao0 = A.altOffset[0], ao1 = A.altOffset[1], ao2 = A.altOffset[2];
    uint32_t two; // a shift count that has to sit in a register (SDWA takes no inline constants)
    asm volatile("v_mov_b32 %0, 2" : "=v"(two));
    // the dynamic LDS starts at address 0 (the kernel declares no other): tables are read at integer addresses, table base in the
    // instruction's offset field
    typedef __attribute__((address_space(3))) const float * LdsFloat;
    typedef __attribute__((address_space(3))) const uint16_t * LdsU16;
    auto lutF = [&](uint32_t byteOffset) -> float { return *(LdsFloat)(uintptr_t)byteOffset; };

    float toneMax = 0.0f;
    double sum = 0.0;
    // Workgroups walk tiles of 256 x 8 pixels (four neighbouring pixels per lane, a row per wave) with a grid stride; a lane whose run would
    // cross the end of the row moves it left to end there: the pixels it shares with its neighbour are computed (and stored, same bytes)
    // twice but counted once.  The next tile's pixels are requested before the current ones are worked on (one tile's loads per wave do not
    // cover the memory latency), into the other of two register sets: the loop body is written out for both, so nothing is copied.
    struct Tile
    {
        PixelRun<BASE_BYTES> bp;
        std::conditional_t<Lds::kPlanes, PlaneRun, PixelRun<Lds::kPlanes ? 4 : GAIN_BYTES>> gp;
        uint32_t i, i0, j;
        bool live;
    };
    uint32_t tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    auto request = [&](Tile & T, uint32_t tile) {
        T.live = false;
        if (tile >= tiles)
            return;
        T.i = tx * (64 * kFastPixels) + threadIdx.x * kFastPixels, T.j = ty * kFastRows + threadIdx.y;
        tx += stepX, ty += stepY;
        if (tx >= tilesX)
            tx -= tilesX, ++ty;
        T.live = T.i < A.width && T.j < A.height;
        if (T.live) {
            T.i0 = min(T.i, A.width - kFastPixels);
            T.bp.load(A.base + (size_t)T.j * A.basePitch + (size_t)T.i0 * BASE_BYTES);
            if constexpr (!Lds::kPlanes) {
                T.gp.load(A.gain + (size_t)T.j * A.gainPitch + (size_t)T.i0 * GAIN_BYTES);
            } else {
                T.gp.y = PlaneRun::four(A.gain + (size_t)T.j * A.gainPitch + T.i0);
                if (A.gainU) {
                    const size_t at = (size_t)T.j * A.gainPitchUV + T.i0;
                    T.gp.u = PlaneRun::four(A.gainU + at), T.gp.v = PlaneRun::four(A.gainV + at);
                }
            }
        }
    };
    // the gain map's own conversion at one pixel (planes): the R, G, B codes avifImageYUVToRGB gives the samples, side by side in one word
    // like a pixel of the converted copy (pixel_generic.h / pixel_fixed.h: the universal kernels' per-pixel routines, restated on tables)
    const GainMapPlaneConversion & K = A.gainConv;
    auto convertGainFixed = [&](uint32_t y, uint32_t u, uint32_t v) -> uint32_t {
        uint32_t r, g, b;
        if (K.identityCopy) { // src/reformat.c:1278-1309
            g = y, b = u, r = v;
        } else {
            // (every factor fits 24 bits -- 16-bit y * 0x0101, coefficients below 2^15, chroma within +-128 --: the full-rate 24-bit multiplies
            //  give the products exactly; left as 32-bit ones they are quarter-rate v_mul_lo_u32 / v_mad_u64_u32)
            const int y1 = (int)(__umul24(y * 0x0101u, (uint32_t)K.fx.yg) >> 16) + K.fx.yb;
            if (!K.hasColor) { // I400ToARGBMatrix: chroma 128
                r = g = b = (uint32_t)clampInt(y1 >> 6, 0, 255);
            } else {
                const int ub = (int)u - 128, vb = (int)v - 128;
                b = (uint32_t)clampInt((y1 + __mul24(K.fx.ub, ub)) >> 6, 0, 255);
                g = (uint32_t)clampInt((y1 - (__mul24(K.fx.ug, ub) + __mul24(K.fx.vg, vb))) >> 6, 0, 255);
                r = (uint32_t)clampInt((y1 + __mul24(K.fx.vr, vb)) >> 6, 0, 255);
            }
        }
        return r | (g << 8) | (b << 16);
    };
    // fp32: the operands of the quantiser's truncation, 0.5f + (c * 255.0f) of the UNCLAMPED channels (src/reformat.c:952-961 runs on channels
    // clamped to [0, 1]; packGainCodes explains why the clamp can wait)
    auto convertGainFloat = [&](uint32_t y, uint32_t u, uint32_t v, float t[3]) {
        const float Y = lutF(Lds::kNormY + (y << 2));
        float R = Y, G = Y, B = Y;
        if (K.hasColor) {
            const float Cb = lutF(Lds::kNormUV + (u << 2)), Cr = lutF(Lds::kNormUV + (v << 2));
            if (K.mode == MODE_COEFF) { // src/reformat.c:874-884
                R = Y + K.twoOneMinusKr * Cr;
                B = Y + K.twoOneMinusKb * Cb;
                const float sum = (K.krOneMinusKr * Cr) + (K.kbOneMinusKb * Cb);
                G = Y - __builtin_fmaf(sum, K.rcpKgTimes2.hi, sum * K.rcpKgTimes2.lo); // (2 * sum) / kg, exactdiv.h
            } else if (K.mode == MODE_IDENTITY) {
                G = Y, B = Cb, R = Cr;
            } else { // MODE_YCGCO, :851-857
                const float h = Y - Cb;
                G = Y + Cb, B = h - Cr, R = h + Cr;
            }
        }
        t[0] = 0.5f + (R * 255.0f), t[1] = 0.5f + (G * 255.0f), t[2] = 0.5f + (B * 255.0f);
    };
    // PLAIN: R, G, B, A order on both sides (the usual case): the three byte permutations per pixel fall away
    auto work = [&](const Tile & T, auto plain) {
        constexpr bool PLAIN = decltype(plain)::value;
        if (!T.live)
            return;
        PixelRun<OUT_BYTES> op;
        float pixelMax[kFastPixels];
        uint32_t gainWord[kFastPixels]; // (planes) the four pixels of the gain map, converted before anything else of the tile is touched
        if constexpr (GAIN_BYTES == 1) {
#pragma unroll
            for (int p = 0; p < kFastPixels; ++p)
                gainWord[p] = convertGainFixed((T.gp.y >> (8 * p)) & 0xff, (T.gp.u >> (8 * p)) & 0xff, (T.gp.v >> (8 * p)) & 0xff);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (GAIN_BYTES == 0) {
            float t[kFastPixels][3];
#pragma unroll
            for (int p = 0; p < kFastPixels; ++p)
                convertGainFloat((T.gp.y >> (8 * p)) & 0xff, (T.gp.u >> (8 * p)) & 0xff, (T.gp.v >> (8 * p)) & 0xff, t[p]);
            // (uint8_t)(0.5f + AVIF_CLAMP(c, 0, 1) * 255.0f) of twelve channels.  v_cvt_pk_u8_f32 converts with the current rounding mode and
            // saturates to [0, 255]; under round-toward-zero that is the truncation of every operand in [0, 256), 0 for the negative ones and
            // 255 from 256 up.  The operand of an unclamped channel c is 0.5f + c * 255.0f: the same number where 0 <= c <= 1; below 0.5
            // (code 0, or negative: saturated to 0) where c < 0, as for the clamped 0; at least 255.5 (code 255, or saturated) where c > 1, as
            // for the clamped 1.  The rounding mode is changed only inside the block.
            static_assert(kFastPixels == 4, "the block below converts four pixels");
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                         "v_cvt_pk_u8_f32 %0, %4, 0, 0\n\t"
                         "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                         "v_cvt_pk_u8_f32 %2, %10, 0, 0\n\t"
                         "v_cvt_pk_u8_f32 %3, %13, 0, 0\n\t"
                         "v_cvt_pk_u8_f32 %0, %5, 1, %0\n\t"
                         "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                         "v_cvt_pk_u8_f32 %2, %11, 1, %2\n\t"
                         "v_cvt_pk_u8_f32 %3, %14, 1, %3\n\t"
                         "v_cvt_pk_u8_f32 %0, %6, 2, %0\n\t"
                         "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                         "v_cvt_pk_u8_f32 %2, %12, 2, %2\n\t"
                         "v_cvt_pk_u8_f32 %3, %15, 2, %3\n\t"
                         "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                         : "=&v"(gainWord[0]), "=&v"(gainWord[1]), "=&v"(gainWord[2]), "=&v"(gainWord[3])
                         : "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[2][0]), "v"(t[2][1]), "v"(t[2][2]),
                           "v"(t[3][0]), "v"(t[3][1]), "v"(t[3][2]));
        }
#pragma unroll
        for (int p = 0; p < kFastPixels; ++p) {
            // byte offsets into the tables: sample code x entry size
            uint32_t r4, g4, b4, a2, gr4, gg4, gb4;
            if constexpr (BASE_BYTES == 4) {
                const uint32_t w = PLAIN ? T.bp.w[p] : __builtin_amdgcn_perm(T.bp.w[p], T.bp.w[p], selBaseX); // R, G, B, A from byte 0 up
                r4 = byte0Times4(w, two), g4 = ((w >> 8) & 0xff) << 2, b4 = ((w >> 16) & 0xff) << 2, a2 = (w >> 24) << 1;
            } else {
                const uint32_t x = PLAIN ? T.bp.w[2 * p] : __builtin_amdgcn_perm(T.bp.w[2 * p + 1], T.bp.w[2 * p], selBaseX);
                const uint32_t y = PLAIN ? T.bp.w[2 * p + 1] : __builtin_amdgcn_perm(T.bp.w[2 * p + 1], T.bp.w[2 * p], selBaseY);
                r4 = word0Times4(x, two), g4 = (x >> 16) << 2, b4 = word0Times4(y, two), a2 = (y >> 16) << 1;
            }
            if constexpr (GAIN_BYTES != 8) { // the gain map is RGBA8 (avifRGBImageSetDefaults), or its planes converted above
                uint32_t w;
                if constexpr (Lds::kPlanes)
                    w = gainWord[p];
                else
                    w = T.gp.w[p];
                gr4 = byte0Times4(w, two), gg4 = ((w >> 8) & 0xff) << 2, gb4 = ((w >> 16) & 0xff) << 2;
            } else {
                const uint32_t x = T.gp.w[2 * p], y = T.gp.w[2 * p + 1];
                gr4 = word0Times4(x, two), gg4 = (x >> 16) << 2, gb4 = word0Times4(y, two);
            }
            float v[3] = { lutF(Lds::kBaseLut + r4), lutF(Lds::kBaseLut + g4), lutF(Lds::kBaseLut + b4) };
            uint32_t alphaCode = *(LdsU16)(uintptr_t)(Lds::kAlphaLut + a2);
#ifdef AVIFHIP_GAINMAP_PROBE // tests/tools/gmbench.hip: what the kernel costs without one of its parts (results are wrong then)
            if (A.fast & 16)
                v[0] = (float)r4, v[1] = (float)g4, v[2] = (float)b4, alphaCode = a2;
#endif
            if constexpr (CONV & 1)
                convertPrimaries(v, A.inM);
            // :236-270
#ifdef AVIFHIP_GAINMAP_PROBE
            if (A.fast & 16) {
                v[0] = (v[0] + bo0) * (float)gr4 - ao0, v[1] = (v[1] + bo1) * (float)gg4 - ao1, v[2] = (v[2] + bo2) * (float)gb4 - ao2;
            } else
#endif
            {
                v[0] = (v[0] + bo0) * lutF(gainR + gr4) - ao0;
                v[1] = (v[1] + bo1) * lutF(gainG + gg4) - ao1;
                v[2] = (v[2] + bo2) * lutF(gainB + gb4) - ao2;
            }
            pixelMax[p] = fmaxf(fmaxf(fmaxf(0.0f, v[0]), v[1]), v[2]); // a NaN leaves the maximum as it is, like the reference's comparisons
            if constexpr (CONV & 2) {
#ifdef AVIFHIP_GAINMAP_PROBE
                if (!(A.fast & 8))
#endif
                    convertPrimaries(v, A.outM);
            }
            uint32_t c0, c1, c2;
#ifdef AVIFHIP_GAINMAP_PROBE
            if (A.fast & 4)
                c0 = __float_as_uint(v[0]) >> 20, c1 = __float_as_uint(v[1]) >> 20, c2 = __float_as_uint(v[2]) >> 20;
            else
#endif
                c0 = locate(v[0], L), c1 = locate(v[1], L), c2 = locate(v[2], L);
            if constexpr (OUT_BYTES == 4) { // the codes' low bytes side by side, then into the output's order
                const uint32_t rg = __builtin_amdgcn_perm(c1, c0, 0x0c0c0400u), ba = __builtin_amdgcn_perm(alphaCode, c2, 0x0c0c0400u);
                op.w[p] = PLAIN ? (rg | (ba << 16)) : __builtin_amdgcn_perm(ba, rg, selOutX);
            } else { // their low halves
                const uint32_t rg = __builtin_amdgcn_perm(c1, c0, 0x05040100u), ba = __builtin_amdgcn_perm(alphaCode, c2, 0x05040100u);
                op.w[2 * p] = PLAIN ? rg : __builtin_amdgcn_perm(ba, rg, selOutX), op.w[2 * p + 1] = PLAIN ? ba : __builtin_amdgcn_perm(ba, rg, selOutY);
            }
        }
#ifdef AVIFHIP_GAINMAP_PROBE
        if (A.fast & 128) { // 64 scalar instructions more per tile: do they take issue slots from the vector ALU?
#pragma unroll
            for (int k = 0; k < 64; ++k)
                asm volatile("s_mov_b32 s90, 0" ::: "s90");
        }
        if (A.fast & 256) { // 64 vector instructions more per tile, for comparison
#pragma unroll
            for (int k = 0; k < 64; ++k)
                asm volatile("v_mov_b32 %0, %0" : "+v"(op.w[0]));
        }
        if (!(A.fast & 32) || op.w[0] == 0xdeadbeefu)
#endif
            op.store(A.out + (size_t)T.j * A.outPitch + (size_t)T.i0 * OUT_BYTES);
        if (A.pixelMax) { // (wave-uniform; a run moved left at the end of a row stores the pixels it shares with its neighbour again: same values)
            float * pm = A.pixelMax + (size_t)T.j * A.width + T.i0;
#pragma unroll
            for (int p = 0; p < kFastPixels; ++p)
                pm[p] = pixelMax[p];
        }
        toneMax = fmaxf(toneMax, fmaxf(fmaxf(pixelMax[0], pixelMax[1]), fmaxf(pixelMax[2], pixelMax[3])));
        if (__builtin_expect(T.i0 == T.i, 1)) {
            sum += ((double)pixelMax[0] + (double)pixelMax[1]) + ((double)pixelMax[2] + (double)pixelMax[3]);
        } else { // the end of a row whose width is no multiple of four: the neighbouring lane counts the first pixels of this run
#pragma unroll
            for (int p = 0; p < kFastPixels; ++p)
                if (T.i0 + p >= T.i)
                    sum += (double)pixelMax[p];
        }
    };
    Tile T0, T1;
    request(T0, blockIdx.x); // the first pixels are on their way while the tables move into the LDS
#if AVIFHIP_GAINMAP_DEPTH == 3
    Tile T2;
    request(T1, blockIdx.x + gridDim.x);
#endif
#ifdef AVIFHIP_GAINMAP_PROBE
    if (!(A.fast & 64))
#endif
    {
        // all four tables in one sweep of 16-byte moves, every lane's loads in flight before its first LDS write (a loop of load - write
        // round trips took 4 of the kernel's 37 us on a 4K image); the host keeps the tables 16-byte aligned and padded
        const uint32_t nBase = 1u << A.baseL.depth, nGain = 1u << A.gainDepth;
        const uint32_t gainLds = (GAIN_BYTES != 8) ? Lds::kGainLut : Lds::kLocator + 16 * ((A.locBuckets + 3) / 4);
        const uint32_t q0 = nBase / 4, q1 = q0 + nBase / 8, q2 = q1 + (3 * nGain) / 4, q3 = q2 + (A.locBuckets + 3) / 4; // in 16-byte units
        const uint32_t t = threadIdx.y * 64 + threadIdx.x;
        constexpr int kMoves = (int)(kGainMapFastLdsBytes / 16 / (64 * kFastRows));
        Quad held[kMoves];
#pragma unroll
        for (int m = 0; m < kMoves; ++m) {
            const uint32_t q = t + m * 64 * kFastRows;
            if (q < q3) {
                const Quad * src = (q < q0) ? reinterpret_cast<const Quad *>(A.baseLut) + q
                                 : (q < q1) ? reinterpret_cast<const Quad *>(A.alphaLut) + (q - q0)
                                 : (q < q2) ? reinterpret_cast<const Quad *>(A.gainLut) + (q - q1)
                                            : reinterpret_cast<const Quad *>(A.locator) + (q - q2);
                held[m] = *src;
            }
        }
#pragma unroll
        for (int m = 0; m < kMoves; ++m) {
            const uint32_t q = t + m * 64 * kFastRows;
            if (q < q3) {
                const uint32_t at = (q < q0) ? Lds::kBaseLut + 16 * q
                                  : (q < q1) ? Lds::kAlphaLut + 16 * (q - q0)
                                  : (q < q2) ? gainLds + 16 * (q - q1)
                                             : Lds::kLocator + 16 * (q - q2);
                *reinterpret_cast<Quad *>(lds + at) = held[m];
            }
        }
        if constexpr (GAIN_BYTES == 0) {
            // unormFloatTableY / unormFloatTableUV of the gain map's conversion (src/reformat.c:575-603), with the reference's own division;
            // identity mode reuses the luma table for chroma (:587-589)
            {
                for (uint32_t k = t; k < 512; k += 64 * kFastRows) {
                    const bool chroma = k >= 256 && K.mode != MODE_IDENTITY;
                    const float code = (float)(k & 255);
                    *reinterpret_cast<float *>(lds + Lds::kNormY + 4 * k) = chroma ? (code - K.biasUV) / K.rangeUV : (code - K.biasY) / K.rangeY;
                }
            }
        }
        __syncthreads();
    }
    const bool plain = (BASE_BYTES == 4 ? selBaseX == 0x03020100u : (selBaseX == 0x03020100u && selBaseY == 0x07060504u)) &&
                       (OUT_BYTES == 4 ? selOutX == 0x05040100u : (selOutX == 0x03020100u && selOutY == 0x07060504u));
#if AVIFHIP_GAINMAP_DEPTH == 3
    // (round 6, tried) two tiles requested ahead of the one being worked on
    if (plain) {
        for (uint32_t tile = blockIdx.x; tile < tiles; tile += 3 * gridDim.x) {
            request(T2, tile + 2 * gridDim.x);
            work(T0, std::true_type{});
            request(T0, tile + 3 * gridDim.x);
            work(T1, std::true_type{});
            request(T1, tile + 4 * gridDim.x);
            work(T2, std::true_type{});
        }
    } else {
        for (uint32_t tile = blockIdx.x; tile < tiles; tile += 3 * gridDim.x) {
            request(T2, tile + 2 * gridDim.x);
            work(T0, std::false_type{});
            request(T0, tile + 3 * gridDim.x);
            work(T1, std::false_type{});
            request(T1, tile + 4 * gridDim.x);
            work(T2, std::false_type{});
        }
    }
#else
    if (plain) {
        for (uint32_t tile = blockIdx.x; tile < tiles; tile += 2 * gridDim.x) {
            request(T1, tile + gridDim.x);
            work(T0, std::true_type{});
            request(T0, tile + 2 * gridDim.x);
            work(T1, std::true_type{});
        }
    } else {
        for (uint32_t tile = blockIdx.x; tile < tiles; tile += 2 * gridDim.x) {
            request(T1, tile + gridDim.x);
            work(T0, std::false_type{});
            request(T0, tile + 2 * gridDim.x);
            work(T1, std::false_type{});
        }
    }
#endif
    __syncthreads(); // the tables are done with: their place serves the reduction
    finishStatistics<kFastRows>(A, toneMax, sum, 0, lds); // no NaN: the host sends a call here only when it can prove that (api_gainmap.cpp)
}

// ---- gain-map computation -----------------------------------------------------------------------------------------
// Round 6: a lane owns FOUR consecutive pixels of a row (one 16-byte load per image for 4-byte pixels, 16-byte stores of the ratio planes),
// a workgroup a run of 1024 pixels of a row per step; the linear-light tables sit in LDS up to 12-bit samples; the bucket of a ratio is
// found from a guess (the buckets are equally wide in log2: a line through two of the table's steps, v_log_f32) corrected against the
// exact steps; histogram counts are aggregated inside the wave before they meet the LDS atomics.  The arithmetic per pixel is what it was
// (the reference's, in its order): 762 us -> see profiles/r06_gainmap_compute.txt for the 4K pair of profiles/r05_gainmap_kernel_stats.txt.

constexpr uint32_t kComputeRun = 1024;        // pixels of a row a workgroup takes per step (256 lanes x 4)
constexpr uint32_t kComputeLdsLutMax = 4096;  // entries per linear-light table that go to LDS (12-bit samples)

// which run of which row workgroup-step `unit` is
struct ComputeRuns
{
    uint32_t runsX, units;
};
__host__ __device__ __forceinline__ ComputeRuns computeRuns(uint32_t width, uint32_t height)
{
    const uint32_t runsX = (width + kComputeRun - 1) / kComputeRun;
    return { runsX, runsX * height };
}

// the sample codes of up to four consecutive pixels starting at pixel i of the row at `row`
__device__ __forceinline__ void readCodes4(const uint8_t * row, uint32_t i, uint32_t n, const GainMapPixelLayout & L, bool vector, bool vector16, uint32_t code[4][3])
{
    if (vector16 && n == 4) { // 4-byte pixels, rows aligned to 16 bytes, i a multiple of 4
        const uint4 w = *reinterpret_cast<const uint4 *>(row + (size_t)i * 4);
        const uint32_t word[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int k = 0; k < 4; ++k)
            code[k][0] = (word[k] >> (8 * L.offR)) & 0xff, code[k][1] = (word[k] >> (8 * L.offG)) & 0xff, code[k][2] = (word[k] >> (8 * L.offB)) & 0xff;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        code[k][0] = code[k][1] = code[k][2] = 0;
        if ((uint32_t)k < n) {
            float alpha;
            readPixel(row + (size_t)(i + k) * L.pixelBytes, L, vector, code[k], alpha);
        }
    }
}

struct ComputeLayout
{
    bool baseVector, altVector, baseVector16, altVector16;
};
__device__ __forceinline__ ComputeLayout computeLayout(const GainMapComputeArgs & A)
{
    ComputeLayout c;
    c.baseVector = A.baseL.hasAlpha && (((uintptr_t)A.base | A.basePitch) & (A.baseL.pixelBytes - 1)) == 0;
    c.altVector = A.altL.hasAlpha && (((uintptr_t)A.alt | A.altPitch) & (A.altL.pixelBytes - 1)) == 0;
    c.baseVector16 = c.baseVector && A.baseL.pixelBytes == 4 && (((uintptr_t)A.base | A.basePitch) & 15) == 0;
    c.altVector16 = c.altVector && A.altL.pixelBytes == 4 && (((uintptr_t)A.alt | A.altPitch) & 15) == 0;
    return c;
}

// the workgroup's copies of the two linear-light tables (LDSLUT: both have at most kComputeLdsLutMax entries)
template <bool LDSLUT>
struct ComputeLuts
{
    const float * base;
    const float * alt;
};
template <bool LDSLUT>
__device__ __forceinline__ ComputeLuts<LDSLUT> stageLuts(const GainMapComputeArgs & A, float * lds)
{
    ComputeLuts<LDSLUT> T;
    if constexpr (LDSLUT) {
        const uint32_t nb = A.baseLutEntries, na = A.altLutEntries;
        for (uint32_t k = threadIdx.x; k < nb; k += 256)
            lds[k] = A.baseLut[k];
        for (uint32_t k = threadIdx.x; k < na; k += 256)
            lds[nb + k] = A.altLut[k];
        __syncthreads();
        T.base = lds, T.alt = lds + nb;
    } else {
        T.base = A.baseLut, T.alt = A.altLut;
    }
    return T;
}

template <int N>
__device__ __forceinline__ void blockReduceStore(float v[N], const bool isMax[N], float * out)
{
    __shared__ float scratch[4][N];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float o = __shfl_xor(v[k], m);
            v[k] = isMax[k] ? fmaxf(v[k], o) : fminf(v[k], o);
        }
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; ++k)
            scratch[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float r = scratch[0][k];
            for (int w = 1; w < 4; ++w)
                r = isMax[k] ? fmaxf(r, scratch[w][k]) : fminf(r, scratch[w][k]);
            out[k] = r;
        }
    }
}

// pass 0: minima of the converted side's channels (the other image is not read)
template <bool LDSLUT>
__global__ __launch_bounds__(256) void gainMapChannelMinKernel(GainMapComputeArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float ldsLut[];
    const ComputeLuts<LDSLUT> T = stageLuts<LDSLUT>(A, ldsLut);
    const ComputeLayout lay = computeLayout(A);
    const ComputeRuns R = computeRuns(A.width, A.height);
    const bool alt = A.convertAlt != 0;
    const uint8_t * image = alt ? A.alt : A.base;
    const uint32_t pitch = alt ? A.altPitch : A.basePitch;
    const GainMapPixelLayout & L = alt ? A.altL : A.baseL;
    const float * lut = alt ? T.alt : T.base;
    float mn[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t unit = blockIdx.x; unit < R.units; unit += gridDim.x) {
        const uint32_t j = unit / R.runsX, i = (unit - j * R.runsX) * kComputeRun + 4 * threadIdx.x;
        if (i >= A.width)
            continue;
        const uint32_t n = A.width - i < 4 ? A.width - i : 4;
        uint32_t code[4][3];
        readCodes4(image + (size_t)j * pitch, i, n, L, alt ? lay.altVector : lay.baseVector, alt ? lay.altVector16 : lay.baseVector16, code);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k >= n)
                break;
            float v[3] = { lut[code[k][0]], lut[code[k][1]], lut[code[k][2]] };
            convertPrimaries(v, A.M);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                mn[c] = (mn[c] < v[c]) ? mn[c] : v[c]; // AVIF_MIN: a NaN candidate replaces the minimum, like the reference's macro
        }
    }
    const bool isMax[3] = { false, false, false };
    blockReduceStore<3>(mn, isMax, A.partials + (size_t)blockIdx.x * 8);
}

// between passes 0 and 1: the offsets that keep the converted side's channels positive (src/gainmap.c:646-660) from pass 0's minima, by one
// workgroup (a few thousand floats), so that pass 1 can follow pass 0 without the host in between
struct GainMapOffsetsArgs
{
    const float * minima;
    uint32_t groups;
    int32_t useBaseColorSpace;
    float base[3], alt[3];
    float * out;
};
__global__ __launch_bounds__(256) void gainMapOffsetsKernel(GainMapOffsetsArgs A)
{
    float mn[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t g = threadIdx.x; g < A.groups; g += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            mn[c] = fminf(mn[c], A.minima[(size_t)g * 8 + c]);
    const bool isMax[3] = { false, false, false };
    __shared__ float folded[3];
    blockReduceStore<3>(mn, isMax, folded);
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = (int)threadIdx.x;
        const float channelMin = folded[c], maxOffset = 0.1f;
        float base = A.base[c], alt = A.alt[c];
        if (channelMin < -1e-10f) {
            if (A.useBaseColorSpace) {
                const float o = alt - channelMin;
                alt = (o < maxOffset) ? o : maxOffset;
            } else {
                const float o = base - channelMin;
                base = (o < maxOffset) ? o : maxOffset;
            }
        }
        A.out[c] = base, A.out[3 + c] = alt;
    }
}

// pass 1: ratios + per-workgroup [baseMax, altMax, minRatio x 3, maxRatio x 3]
template <bool LDSLUT, int CHANNELS, int FAST> // FAST: 0 -- any layout; 4 / 8 -- whole 16-byte runs, the alternate image's pixels of 4 / 8 bytes
__global__ __launch_bounds__(256) void gainMapRatioKernel(GainMapComputeArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float ldsLut[];
    ComputeLuts<LDSLUT> tables = { nullptr, nullptr }; // (staged below, behind the first step's loads where they can be issued early)
    const ComputeLayout lay = computeLayout(A);
    const ComputeRuns R = computeRuns(A.width, A.height);
    const size_t numPixels = (size_t)A.width * A.height;
    constexpr int channels = CHANNELS; // 1: A.singleChannel (a compile-time count keeps the per-channel arrays in registers)
    const bool vectorStores = (A.width & 3u) == 0; // (every plane of the ratio buffer then starts on 16 bytes, and so does every lane's run)
    float acc[8] = { 1.0f, 1.0f, __builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f, 0.0f, 0.0f };
    // (the offsets may come from device memory: gainMapOffsetsKernel folded pass 0's minima while the host did not wait)
    float baseOffset[3] = { A.baseOffset[0], A.baseOffset[1], A.baseOffset[2] }, altOffset[3] = { A.altOffset[0], A.altOffset[1], A.altOffset[2] };
    if (A.offsets) { // (a kernel argument: uniform over the launch; six scalar loads)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            baseOffset[c] = A.offsets[c], altOffset[c] = A.offsets[3 + c];
    }
    // what a lane does with the codes of its (up to) four pixels at (i .. i + n - 1, j)
    auto process = [&](uint32_t j, uint32_t i, uint32_t n, const uint32_t (&bc)[4][3], const uint32_t (&ac)[4][3]) {
        float ratio[CHANNELS][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float b[3] = { tables.base[bc[k][0]], tables.base[bc[k][1]], tables.base[bc[k][2]] };
            float a[3] = { tables.alt[ac[k][0]], tables.alt[ac[k][1]], tables.alt[ac[k][2]] };
            if (A.convertAlt)
                convertPrimaries(a, A.M);
            if (A.convertBase)
                convertPrimaries(b, A.M);
            const bool live = (uint32_t)k < n;
#pragma unroll
            for (int c = 0; c < channels; ++c) {
                float base = b[c], alt = a[c];
                if constexpr (CHANNELS == 1) { // :699-703, products and sums in the reference's order
                    base = A.yCoeffs[0] * b[0] + A.yCoeffs[1] * b[1] + A.yCoeffs[2] * b[2];
                    alt = A.yCoeffs[0] * a[0] + A.yCoeffs[1] * a[1] + A.yCoeffs[2] * a[2];
                }
                const float q = (alt + altOffset[c]) / (base + baseOffset[c]);
                const float r = (q > 1e-10f) ? q : 1e-10f; // AVIF_MAX(ratio, kEpsilon): NaN -> epsilon
                ratio[c][k] = r;
                if (live) {
                    if (base > acc[0])
                        acc[0] = base;
                    if (alt > acc[1])
                        acc[1] = alt;
                    acc[2 + c] = fminf(acc[2 + c], r), acc[5 + c] = fmaxf(acc[5 + c], r);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < channels; ++c) {
            float * dst = A.ratios + (size_t)c * numPixels + (size_t)j * A.width + i;
            if (vectorStores) {
                *reinterpret_cast<float4 *>(dst) = make_float4(ratio[c][0], ratio[c][1], ratio[c][2], ratio[c][3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((uint32_t)k < n)
                        dst[k] = ratio[c][k];
            }
        }
    };
    // Whole 16-byte runs on both sides (4- or 8-byte pixels, rows on 16 bytes, a width of whole quads -- what every RGBA image of libavif's
    // own allocation is): the next step's pixels are requested before this step's are worked on, the first step's before the tables are staged
    // (round 6: a workgroup's two steps had been load - wait - compute - store, twice, behind the staging of its tables).  Wave-uniform.
    // FAST: a kernel of its own (launchGainMapRatios decides: gainMapRatioFast), without the general path's five pixel layouts
    if constexpr (FAST != 0) {
        constexpr bool alt8 = FAST == 8;
        struct Raw
        {
            uint4 b, a0, a1;
            uint32_t j, i;
            bool live;
        };
        auto issue = [&](Raw & Q, uint32_t unit) {
            Q.live = unit < R.units;
            if (!Q.live)
                return;
            Q.j = unit / R.runsX, Q.i = (unit - Q.j * R.runsX) * kComputeRun + 4 * threadIdx.x;
            Q.live = Q.i < A.width;
            if (!Q.live)
                return;
            Q.b = *reinterpret_cast<const uint4 *>(A.base + (size_t)Q.j * A.basePitch + (size_t)Q.i * 4);
            if constexpr (alt8) {
                const uint4 * q = reinterpret_cast<const uint4 *>(A.alt + (size_t)Q.j * A.altPitch + (size_t)Q.i * 8);
                Q.a0 = q[0], Q.a1 = q[1];
            } else {
                Q.a0 = *reinterpret_cast<const uint4 *>(A.alt + (size_t)Q.j * A.altPitch + (size_t)Q.i * 4);
            }
        };
        auto work = [&](const Raw & Q) {
            if (!Q.live)
                return;
            uint32_t bc[4][3], ac[4][3];
            const uint32_t bw[4] = { Q.b.x, Q.b.y, Q.b.z, Q.b.w };
#pragma unroll
            for (int k = 0; k < 4; ++k)
                bc[k][0] = (bw[k] >> (8 * A.baseL.offR)) & 0xff, bc[k][1] = (bw[k] >> (8 * A.baseL.offG)) & 0xff, bc[k][2] = (bw[k] >> (8 * A.baseL.offB)) & 0xff;
            if constexpr (alt8) { // (readPixel's 8-byte form)
                const uint32_t lo[4] = { Q.a0.x, Q.a0.z, Q.a1.x, Q.a1.z }, hi[4] = { Q.a0.y, Q.a0.w, Q.a1.y, Q.a1.w };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    auto pick = [&](uint32_t off) -> uint32_t { return (((off & 4) ? hi[k] : lo[k]) >> (8 * (off & 3))) & 0xffff; };
                    ac[k][0] = pick(A.altL.offR), ac[k][1] = pick(A.altL.offG), ac[k][2] = pick(A.altL.offB);
                }
            } else {
                const uint32_t aw[4] = { Q.a0.x, Q.a0.y, Q.a0.z, Q.a0.w };
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    ac[k][0] = (aw[k] >> (8 * A.altL.offR)) & 0xff, ac[k][1] = (aw[k] >> (8 * A.altL.offG)) & 0xff, ac[k][2] = (aw[k] >> (8 * A.altL.offB)) & 0xff;
            }
            process(Q.j, Q.i, 4, bc, ac);
        };
        Raw Q0, Q1;
        issue(Q0, blockIdx.x);
        const ComputeLuts<LDSLUT> T = stageLuts<LDSLUT>(A, ldsLut);
        tables = T;
        for (uint32_t unit = blockIdx.x; unit < R.units; unit += 2 * gridDim.x) {
            issue(Q1, unit + gridDim.x);
            work(Q0);
            issue(Q0, unit + 2 * gridDim.x);
            work(Q1);
        }
    } else {
        const ComputeLuts<LDSLUT> T = stageLuts<LDSLUT>(A, ldsLut);
        tables = T;
        for (uint32_t unit = blockIdx.x; unit < R.units; unit += gridDim.x) {
            const uint32_t j = unit / R.runsX, i = (unit - j * R.runsX) * kComputeRun + 4 * threadIdx.x;
            if (i >= A.width)
                continue;
            const uint32_t n = A.width - i < 4 ? A.width - i : 4;
            uint32_t bc[4][3], ac[4][3];
            readCodes4(A.base + (size_t)j * A.basePitch, i, n, A.baseL, lay.baseVector, lay.baseVector16, bc);
            readCodes4(A.alt + (size_t)j * A.altPitch, i, n, A.altL, lay.altVector, lay.altVector16, ac);
            process(j, i, n, bc, ac);
        }
    }
    const bool isMax[8] = { true, true, false, false, false, true, true, true };
    blockReduceStore<8>(acc, isMax, A.partials + (size_t)blockIdx.x * 8);
}

// largest k with steps[k] <= x; `entries` a power of two, steps[0] = -inf, NaN padding
__device__ __forceinline__ uint32_t stepIndex(const float * steps, uint32_t entries, float x)
{
    uint32_t pos = 0;
    for (uint32_t s = entries >> 1; s; s >>= 1)
        pos += (steps[pos + s] <= x) ? s : 0;
    return pos;
}
// ... from a guess: the steps of a bucket table are equally spaced in log2(x), so a line through two of them (GainMapStepTable::guessA / B,
// gainmap_plan.cpp) lands within a bucket of the answer; the exact steps then decide.  Correct for any guess (a bad one only walks longer);
// `last`: the last finite step's index.
__device__ __forceinline__ uint32_t stepIndexGuessed(const float * steps, uint32_t last, float a, float b, float x)
{
    const float g = a * __log2f(x) + b;
    const uint32_t m = (g >= 0.0f) ? (g < (float)last ? (uint32_t)g : last) : 0u; // (NaN: 0)
    return stepIndexFromGuess(steps, last, m, x); // gainmap_steps.h: four steps around the guess decide, the walks serve the rest
}

struct GainMapStepTables
{
    GainMapStepTable t[3];
};

// Histograms are privatised: each (persistent) workgroup counts one channel at a time in LDS (<= 10000 buckets = 40 KB) and adds
// its non-zero counts to the global histogram at the end -- one global atomic per sample took 5.9 ms on a 4K image (the
// distribution is peaked: most samples fall into a few buckets).  For the same reason the lanes of a wave first agree on what they hold:
// up to four rounds of "the first lane's bucket, counted over the wave, added once" before the rest go to the LDS atomics one by one.
constexpr uint32_t kHistogramLdsBuckets = 10240;
// Few, large workgroups: what a workgroup adds to the global histogram at the end is one atomic per non-zero counter -- 2048 workgroups of 256
// lanes spent more time flushing a spread-out histogram (4 300 buckets x 3 channels x 2048) than counting (round 6: 140 us of 4K's 8.3 MP)
constexpr uint32_t kHistogramThreads = 1024, kHistogramGroups = 512;

__device__ __forceinline__ void histogramAdd(uint32_t * local, uint32_t bucket, bool valid)
{
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
#pragma unroll 1
    for (int round = 0; round < 4 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1; // (wave-uniform: from the ballot)
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bucket, leader);
        const unsigned long long same = __ballot(valid && bucket == b) & todo;
        if ((int)lane == leader)
            atomicAdd(&local[b], (uint32_t)__popcll(same));
        todo &= ~same;
        if (__popcll(same) < 12)
            break; // a spread-out wave: the rounds cost more than the conflicts of the plain atomics they save
    }
    if ((todo >> lane) & 1ull)
        atomicAdd(&local[bucket], 1u);
}

__global__ __launch_bounds__(1024) void gainMapHistogramKernel(const float * ratios, size_t numPixels, int channels, GainMapStepTables T, uint32_t * h0,
                                                              uint32_t * h1, uint32_t * h2, uint32_t capacity)
{
    // `capacity` counters, then as many steps (the channel's: the corrections of a guess read two or three of them per sample).  Sized at launch
    // by the widest channel -- a few hundred buckets for a real HDR / SDR pair, 10 000 at most -- so that several workgroups share a CU
    extern __shared__ __attribute__((aligned(16))) uint32_t histogramLds[];
    uint32_t * local = histogramLds;
    float * ldsSteps = reinterpret_cast<float *>(histogramLds + capacity);
    uint32_t * const hist[3] = { h0, h1, h2 };
    const size_t quads = numPixels >> 2; // (ratio planes start on 16 bytes when numPixels is a multiple of 4; the tail below otherwise)
    for (int c = 0; c < channels; ++c) {
        if (!T.t[c].entries)
            continue;
        const uint32_t buckets = T.t[c].flip + 1;
        const uint32_t last = T.t[c].flip, flip = T.t[c].flip;
        const bool flipped = T.t[c].flipped != 0;
        const float ga = T.t[c].guessA, gb = T.t[c].guessB;
        const bool guessed = ga != 0.0f; // (workgroup-uniform)
        for (uint32_t k = threadIdx.x; k < buckets; k += kHistogramThreads) {
            local[k] = 0;
            if (guessed)
                ldsSteps[k] = T.t[c].steps[k];
        }
        __syncthreads();
        const float * plane = ratios + (size_t)c * numPixels;
        const bool aligned = (((uintptr_t)plane) & 15) == 0;
        const float * steps = T.t[c].steps;
        auto bucketOf = [&](float x) -> uint32_t {
            const uint32_t m = guessed ? stepIndexGuessed(ldsSteps, last, ga, gb, x) : stepIndex(steps, T.t[c].entries, x);
            return flipped ? flip - m : m;
        };
        // (uniform trip count over the workgroup's waves: the ballots of histogramAdd need every lane of a wave in the loop)
        const size_t stride = (size_t)gridDim.x * kHistogramThreads;
        if (aligned) {
            // two 16-byte loads in flight per lane; eight chains of table reads side by side
            for (size_t k0 = (size_t)blockIdx.x * kHistogramThreads; k0 < quads; k0 += 2 * stride) {
                const size_t ka = k0 + threadIdx.x, kb = ka + stride;
                const bool va = ka < quads, vb = kb < quads;
                float4 p = make_float4(1.0f, 1.0f, 1.0f, 1.0f), q = p;
                if (va)
                    p = reinterpret_cast<const float4 *>(plane)[ka];
                if (vb)
                    q = reinterpret_cast<const float4 *>(plane)[kb];
                const uint32_t b0 = bucketOf(p.x), b1 = bucketOf(p.y), b2 = bucketOf(p.z), b3 = bucketOf(p.w);
                const uint32_t b4 = bucketOf(q.x), b5 = bucketOf(q.y), b6 = bucketOf(q.z), b7 = bucketOf(q.w);
                histogramAdd(local, b0, va);
                histogramAdd(local, b1, va);
                histogramAdd(local, b2, va);
                histogramAdd(local, b3, va);
                histogramAdd(local, b4, vb);
                histogramAdd(local, b5, vb);
                histogramAdd(local, b6, vb);
                histogramAdd(local, b7, vb);
            }
            for (size_t k = quads * 4 + (size_t)blockIdx.x * kHistogramThreads + threadIdx.x; k < numPixels; k += stride)
                atomicAdd(&local[bucketOf(plane[k])], 1u);
        } else {
            for (size_t k = (size_t)blockIdx.x * kHistogramThreads + threadIdx.x; k < numPixels; k += stride)
                atomicAdd(&local[bucketOf(plane[k])], 1u);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < buckets; k += kHistogramThreads)
            if (local[k])
                atomicAdd(&hist[c][k], local[k]);
        __syncthreads();
    }
}

// pass 3: codes.  The code steps of the three channels sit in LDS up to 12-bit maps (LDSSTEPS); a lane owns four consecutive pixels of a row.
template <bool LDSSTEPS, int CHANNELS>
__global__ __launch_bounds__(256) void gainMapQuantiseKernel(const float * ratios, uint32_t width, uint32_t height, GainMapStepTables T,
                                                             uint8_t * rgba, uint32_t rgbaPitch, uint32_t depth)
{
    constexpr int channels = CHANNELS;
    extern __shared__ __attribute__((aligned(16))) float ldsSteps[];
    const float * steps[3] = { T.t[0].steps, T.t[1].steps, T.t[2].steps };
    if constexpr (LDSSTEPS) {
        uint32_t at = 0;
        for (int c = 0; c < channels; ++c) {
            for (uint32_t k = threadIdx.x; k < T.t[c].entries; k += 256)
                ldsSteps[at + k] = T.t[c].steps[k];
            steps[c] = ldsSteps + at;
            at += T.t[c].entries;
        }
        __syncthreads();
    }
    const size_t numPixels = (size_t)width * height;
    const uint32_t maxCode = (1u << depth) - 1;
    const ComputeRuns R = computeRuns(width, height);
    const bool vectorLoads = (width & 3u) == 0 && (((uintptr_t)ratios) & 15) == 0;
    const bool vectorStores = (((uintptr_t)rgba | rgbaPitch) & 15) == 0;
    for (uint32_t unit = blockIdx.x; unit < R.units; unit += gridDim.x) {
        const uint32_t j = unit / R.runsX, i = (unit - j * R.runsX) * kComputeRun + 4 * threadIdx.x;
        if (i >= width)
            continue;
        const uint32_t n = width - i < 4 ? width - i : 4;
        uint32_t code[3][4];
#pragma unroll
        for (int c = 0; c < channels; ++c) {
            const float * src = ratios + (size_t)c * numPixels + (size_t)j * width + i;
            float v[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
            if (vectorLoads) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((uint32_t)k < n)
                        v[k] = src[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!T.t[c].entries) {
                    code[c][k] = 0;
                } else {
                    // (gamma 1: the codes are equally spaced in log2 of the ratio between the range's ends -- a guess, corrected against the steps)
                    const uint32_t m = (T.t[c].guessA != 0.0f) ? stepIndexGuessed(steps[c], T.t[c].flip, T.t[c].guessA, T.t[c].guessB, v[k])
                                                               : stepIndex(steps[c], T.t[c].entries, v[k]);
                    code[c][k] = T.t[c].flipped ? T.t[c].flip - m : m;
                }
            }
        }
        if constexpr (CHANNELS == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                code[1][k] = code[2][k] = code[0][k];
        }
        uint8_t * p = rgba + (size_t)j * rgbaPitch;
        if (depth > 8) {
            uint2 px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                px[k] = { code[0][k] | (code[1][k] << 16), code[2][k] | (maxCode << 16) };
            if (vectorStores && n == 4) {
                reinterpret_cast<uint4 *>(p)[i / 2] = make_uint4(px[0].x, px[0].y, px[1].x, px[1].y);
                reinterpret_cast<uint4 *>(p)[i / 2 + 1] = make_uint4(px[2].x, px[2].y, px[3].x, px[3].y);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((uint32_t)k < n)
                        reinterpret_cast<uint2 *>(p)[i + k] = px[k];
            }
        } else {
            uint32_t px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                px[k] = code[0][k] | (code[1][k] << 8) | (code[2][k] << 16) | (maxCode << 24);
            if (vectorStores && n == 4) {
                reinterpret_cast<uint4 *>(p)[i / 4] = make_uint4(px[0], px[1], px[2], px[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((uint32_t)k < n)
                        reinterpret_cast<uint32_t *>(p)[i + k] = px[k];
            }
        }
    }
}

} // namespace

size_t gainMapFastLdsBytes(uint32_t basePixelBytes, uint32_t gainDepth, uint32_t locBuckets, bool planes)
{
    const size_t fixed = (basePixelBytes == 4) ? FastLds<4, 8>::kLocator : FastLds<8, 8>::kLocator; // base and alpha tables
    return fixed + ((size_t)12 << gainDepth) + (planes ? 2 * 4 * 256 : 0) + (size_t)((locBuckets + 3) / 4) * 16;
}

hipError_t launchGainMapApply(const GainMapArgs & A, hipStream_t stream, uint32_t * partials)
{
    *partials = 0;
    if (!A.width || !A.height)
        return hipSuccess;
    if (A.fast) {
        // 256 x 8 pixels per workgroup and step.  As many workgroups as the CUs hold at once -- by the tables' LDS and by the registers of
        // the instantiation (the variants with two primaries conversions need more than the 64 that leave room for eight waves per SIMD) --
        // or fewer, so that every workgroup makes the same number of steps when the tiles divide that way.
        const uint32_t tilesX = (A.width + 64 * kFastPixels - 1) / (64 * kFastPixels), tiles = tilesX * ((A.height + kFastRows - 1) / kFastRows);
        const size_t lds = gainMapFastLdsBytes(A.baseL.pixelBytes, A.gainDepth, A.locBuckets, A.gainPlanes != 0);
        const uint32_t byLds = (uint32_t)((160 * 1024) / (lds + 512));
        const int conv = (A.inConv ? 1 : 0) | (A.outConv ? 2 : 0);
        // (gain map: RGBA8, RGBA16, or its own 8-bit planes)
        const bool planesFixed = A.gainConv.fixedPoint || A.gainConv.identityCopy;
        const int key = 4 * ((A.baseL.pixelBytes == 8 ? 2 : 0) | (A.outL.pixelBytes == 8 ? 1 : 0)) + (A.gainPlanes ? (planesFixed ? 3 : 2) : (A.gainDepth > 8 ? 1 : 0));
        static std::atomic<int> wavesPerSimd[16][4]; // of each instantiation, from its register count (0: not asked yet; any thread may ask: same answer)
        uint32_t groups = 0;
        auto launch = [&](auto kernel) {
            int waves = wavesPerSimd[key][conv].load(std::memory_order_relaxed);
            if (!waves) {
                hipFuncAttributes attr;
                waves = 4;
                if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)) == hipSuccess && attr.numRegs > 0)
                    waves = 512 / ((attr.numRegs + 7) & ~7);
                else
                    (void)hipGetLastError();
#ifdef AVIFHIP_GAINMAP_PROBE
                printf("numRegs %d sharedSizeBytes %zu maxThreadsPerBlock %d\n", attr.numRegs, attr.sharedSizeBytes, attr.maxThreadsPerBlock);
#endif
                waves = (waves > 8) ? 8 : ((waves < 2) ? 2 : waves);
                wavesPerSimd[key][conv].store(waves, std::memory_order_relaxed);
            }
            const uint32_t byRegisters = (uint32_t)(4 * waves / kFastRows);
            const uint32_t perCu = (byLds < byRegisters) ? byLds : byRegisters;
            const uint32_t resident = 256 * (perCu < 1 ? 1 : perCu);
            const uint32_t steps = (tiles + resident - 1) / resident;
            groups = (tiles + steps - 1) / steps;
#ifdef AVIFHIP_GAINMAP_PROBE
            if (getenv("GM_GROUPS"))
                groups = (uint32_t)atoi(getenv("GM_GROUPS"));
#endif
            hipLaunchKernelGGL(kernel, dim3(groups), dim3(64, kFastRows), lds, stream, A, tilesX, tiles, groups % tilesX, groups / tilesX);
        };
        auto byConv = [&](auto b, auto o, auto g) {
            constexpr int B = decltype(b)::value, O = decltype(o)::value, G = decltype(g)::value;
            switch (conv) {
                case 0: launch(gainMapApplyFastKernel<B, O, G, 0>); break;
                case 1: launch(gainMapApplyFastKernel<B, O, G, 1>); break;
                case 2: launch(gainMapApplyFastKernel<B, O, G, 2>); break;
                default: launch(gainMapApplyFastKernel<B, O, G, 3>); break;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        auto byGain = [&](auto b, auto o) {
            switch (key % 4) {
                case 0: byConv(b, o, I4{}); break;
                case 1: byConv(b, o, I8{}); break;
                case 2: byConv(b, o, I0{}); break;
                default: byConv(b, o, I1{}); break;
            }
        };
        switch (key / 4) {
            case 0: byGain(I4{}, I4{}); break;
            case 1: byGain(I4{}, I8{}); break;
            case 2: byGain(I8{}, I4{}); break;
            default: byGain(I8{}, I8{}); break;
        }
        *partials = groups;
        return hipGetLastError();
    }
    const uint32_t tilesX = (A.width + 63) / 64, tiles = tilesX * ((A.height + 3) / 4);
    const uint32_t groups = tiles < kGainMapMaxGroups ? tiles : kGainMapMaxGroups;
    *partials = A.gain ? groups : 0;
    const size_t lds = A.ldsSteps ? (size_t)(A.ldsSteps + A.ldsBaseLut + A.ldsGainLut) * sizeof(float) + ((size_t)A.guideBuckets + 2) * sizeof(uint16_t) : 0;
    if (A.ldsLocator)
        hipLaunchKernelGGL(gainMapApplyKernel<2>, dim3(groups), dim3(64, 4), (size_t)(A.ldsBaseLut + A.ldsGainLut + A.locBuckets) * sizeof(float), stream, A, tilesX, tiles);
    else if (lds)
        hipLaunchKernelGGL(gainMapApplyKernel<1>, dim3(groups), dim3(64, 4), lds, stream, A, tilesX, tiles);
    else
        hipLaunchKernelGGL(gainMapApplyKernel<0>, dim3(groups), dim3(64, 4), 0, stream, A, tilesX, tiles);
    return hipGetLastError();
}

uint32_t gainMapComputeGroups(uint32_t width, uint32_t height)
{
    // (one or two runs per workgroup: 2048 workgroups with four runs each measured 5 % slower -- the passes wait on memory, not on the table staging)
    const uint32_t units = computeRuns(width, height).units, most = kGainMapMaxGroups;
    return units < most ? (units ? units : 1u) : most;
}

static bool computeLutsFitLds(const GainMapComputeArgs & A)
{
    return A.baseLutEntries && A.altLutEntries && A.baseLutEntries <= kComputeLdsLutMax && A.altLutEntries <= kComputeLdsLutMax;
}

hipError_t launchGainMapChannelMin(const GainMapComputeArgs & A, hipStream_t stream)
{
    const uint32_t groups = gainMapComputeGroups(A.width, A.height);
    if (computeLutsFitLds(A))
        hipLaunchKernelGGL(gainMapChannelMinKernel<true>, dim3(groups), dim3(256), (A.baseLutEntries + A.altLutEntries) * sizeof(float), stream, A);
    else
        hipLaunchKernelGGL(gainMapChannelMinKernel<false>, dim3(groups), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launchGainMapOffsets(const float * minima, uint32_t groups, bool useBaseColorSpace, const float base[3], const float alt[3], float * out, hipStream_t stream)
{
    GainMapOffsetsArgs A = { minima, groups, useBaseColorSpace ? 1 : 0, { base[0], base[1], base[2] }, { alt[0], alt[1], alt[2] }, out };
    hipLaunchKernelGGL(gainMapOffsetsKernel, dim3(1), dim3(256), 0, stream, A);
    return hipGetLastError();
}

// whole 16-byte runs on both sides (gainMapRatioKernel's FAST form): 4-channel pixels of 4 or 8 bytes, bases and pitches on 16 bytes, whole quads
static bool gainMapRatioFast(const GainMapComputeArgs & A)
{
    const bool base16 = A.baseL.hasAlpha && A.baseL.pixelBytes == 4 && (((uintptr_t)A.base | A.basePitch) & 15) == 0;
    const bool alt16 = A.altL.hasAlpha && (A.altL.pixelBytes == 4 || A.altL.pixelBytes == 8) && (((uintptr_t)A.alt | A.altPitch) & 15) == 0;
    return (A.width & 3u) == 0 && base16 && alt16;
}

hipError_t launchGainMapRatios(const GainMapComputeArgs & A, hipStream_t stream)
{
    const uint32_t groups = gainMapComputeGroups(A.width, A.height);
    const uint32_t lds = (A.baseLutEntries + A.altLutEntries) * (uint32_t)sizeof(float);
    const bool fast = gainMapRatioFast(A);
    auto launch = [&](auto ldsLut, auto channels) {
        constexpr bool L = decltype(ldsLut)::value;
        constexpr int C = decltype(channels)::value;
        if (fast && A.altL.pixelBytes == 8)
            hipLaunchKernelGGL((gainMapRatioKernel<L, C, 8>), dim3(groups), dim3(256), L ? lds : 0, stream, A);
        else if (fast)
            hipLaunchKernelGGL((gainMapRatioKernel<L, C, 4>), dim3(groups), dim3(256), L ? lds : 0, stream, A);
        else
            hipLaunchKernelGGL((gainMapRatioKernel<L, C, 0>), dim3(groups), dim3(256), L ? lds : 0, stream, A);
    };
    using One = std::integral_constant<int, 1>;
    using Three = std::integral_constant<int, 3>;
    if (computeLutsFitLds(A)) {
        if (A.singleChannel)
            launch(std::true_type{}, One{});
        else
            launch(std::true_type{}, Three{});
    } else {
        if (A.singleChannel)
            launch(std::false_type{}, One{});
        else
            launch(std::false_type{}, Three{});
    }
    return hipGetLastError();
}

hipError_t launchGainMapHistogram(const float * ratios, size_t numPixels, int channels, const GainMapStepTable tables[3], uint32_t * const histograms[3],
                                  hipStream_t stream)
{
    GainMapStepTables T;
    for (int c = 0; c < 3; ++c)
        T.t[c] = tables[c];
    const size_t want = (numPixels + 4 * kHistogramThreads - 1) / (4 * kHistogramThreads);
    const uint32_t groups = (uint32_t)(want < kHistogramGroups ? (want ? want : 1) : kHistogramGroups);
    uint32_t capacity = 64;
    for (int c = 0; c < channels; ++c)
        if (T.t[c].entries && T.t[c].flip + 1 > capacity)
            capacity = T.t[c].flip + 1;
    capacity = (capacity + 63u) & ~63u;
    if (capacity > kHistogramLdsBuckets)
        return hipErrorInvalidValue; // (the reference caps its histogram at 10 000 buckets, src/gainmap.c:393)
    hipLaunchKernelGGL(gainMapHistogramKernel, dim3(groups), dim3(kHistogramThreads), 2 * capacity * sizeof(uint32_t), stream, ratios, numPixels, channels, T, histograms[0],
                       histograms[1], histograms[2], capacity);
    return hipGetLastError();
}

hipError_t launchGainMapQuantise(const float * ratios, uint32_t width, uint32_t height, int channels, const GainMapStepTable tables[3], uint8_t * rgba,
                                 uint32_t rgbaPitch, uint32_t depth, hipStream_t stream)
{
    GainMapStepTables T;
    for (int c = 0; c < 3; ++c)
        T.t[c] = tables[c];
    const uint32_t units = computeRuns(width, height).units;
    const uint32_t groups = units < 8192 ? (units ? units : 1u) : 8192u; // (2048: 45 -> 52 us for a 4K map)
    size_t entries = 0;
    for (int c = 0; c < channels; ++c)
        entries += T.t[c].entries;
    const size_t lds = (entries ? entries : 1) * sizeof(float);
    if (entries <= 3 * 4096) {
        if (channels == 1)
            hipLaunchKernelGGL((gainMapQuantiseKernel<true, 1>), dim3(groups), dim3(256), lds, stream, ratios, width, height, T, rgba, rgbaPitch, depth);
        else
            hipLaunchKernelGGL((gainMapQuantiseKernel<true, 3>), dim3(groups), dim3(256), lds, stream, ratios, width, height, T, rgba, rgbaPitch, depth);
    } else {
        if (channels == 1)
            hipLaunchKernelGGL((gainMapQuantiseKernel<false, 1>), dim3(groups), dim3(256), 0, stream, ratios, width, height, T, rgba, rgbaPitch, depth);
        else
            hipLaunchKernelGGL((gainMapQuantiseKernel<false, 3>), dim3(groups), dim3(256), 0, stream, ratios, width, height, T, rgba, rgbaPitch, depth);
    }
    return hipGetLastError();
}

} // namespace avifhip
