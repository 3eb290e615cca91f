// r2y_tile_impl.h -- bandwidth-tuned RGB -> YUV kernels for gfx950 (the encode direction, avifImageRGBToYUV,
// src/reformat.c:275-470 + the alpha plane pass :545-569), instantiated by kernels_r2y_tile_*.hip.
//
// Scope: interleaved 3- or 4-channel RGB at 8-bit or 16-bit containers into 8-bit or 16-bit-container 4:4:4 / 4:2:2 /
// 4:2:0 / 4:0:0 planes with the matrix-coefficient ("normal YUV") transform, no alpha (un)multiply, with the alpha
// plane (copy / depth rescale / opaque fill) written in the same pass.  Gray sources, identity / YCgCo matrices, pending
// alpha multiplies, unaligned buffers, divisors off the verified list and the <= 3 columns / <= 1 row that do not fill
// a 4x2 pixel group go to kernels_generic.hip.
//
// Structure: wave = 64 lanes; a lane owns 4 consecutive pixels of 2 rows (two 2x2 chroma blocks): one 16-byte load per
// row for RGBA8 (1 KiB contiguous per wave instruction), one 4-byte luma (and alpha) store per row, one 2-byte store per
// chroma plane.  A wave walks down `stripsPerWave` strips (256 x 2 pixels) with the next strip's loads in flight while
// the current one is computed.  No LDS: every input byte is used by exactly one lane.
//
// Arithmetic: the reference's fp32 operations in the reference's order, no contraction; the three divisions by plan
// constants (channel maximum, 2(1-kb), 2(1-kr)) use the exhaustively verified reciprocal form (exactdiv.h).
#pragma once

#include <hip/hip_runtime.h>

#include "pixel_math.h"
#include "r2y_tile_shared.h"

namespace avifhip {
namespace r2y {

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kLanes = 64;
constexpr int kWaves = 4;

// the lane's 4 pixels of one row, undecoded
template <typename RT, int NCH>
struct RawRow
{
    static constexpr int kWords = 4 * NCH * (int)sizeof(RT) / 4; // 3, 4, 6 or 8 dwords
    unsigned w[kWords];
};

template <typename RT, int NCH>
__device__ __forceinline__ RawRow<RT, NCH> loadRow(const uint8_t * base, uint32_t off)
{
    RawRow<RT, NCH> r;
    constexpr int kWords = RawRow<RT, NCH>::kWords;
    if constexpr (kWords == 4) {
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(base + off));
        r.w[0] = t.x, r.w[1] = t.y, r.w[2] = t.z, r.w[3] = t.w;
    } else if constexpr (kWords == 8) {
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(base + off));
        const u4 s = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(base + off + 16));
        r.w[0] = t.x, r.w[1] = t.y, r.w[2] = t.z, r.w[3] = t.w, r.w[4] = s.x, r.w[5] = s.y, r.w[6] = s.z, r.w[7] = s.w;
    } else if constexpr (kWords == 3) { // 12 bytes, 4-byte aligned
        const unsigned * p = reinterpret_cast<const unsigned *>(base + off);
        r.w[0] = __builtin_nontemporal_load(p), r.w[1] = __builtin_nontemporal_load(p + 1), r.w[2] = __builtin_nontemporal_load(p + 2);
    } else { // 24 bytes, 8-byte aligned
        const u2 * p = reinterpret_cast<const u2 *>(base + off);
        const u2 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2);
        r.w[0] = a.x, r.w[1] = a.y, r.w[2] = b.x, r.w[3] = b.y, r.w[4] = c.x, r.w[5] = c.y;
    }
    return r;
}

// channel `ch` (0 .. NCH-1, in memory order) of pixel `px` (0 .. 3) of a raw row
template <typename RT, int NCH>
__device__ __forceinline__ unsigned channelOf(const RawRow<RT, NCH> & r, int px, int ch)
{
    const int idx = px * NCH + ch; // compile-time after unrolling
    if constexpr (sizeof(RT) == 1)
        return (r.w[idx >> 2] >> (8 * (idx & 3))) & 0xffu;
    else
        return (r.w[idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
}

struct Yuvf
{
    float y, u, v;
};

// AVIF_CLAMP((int)floorf(v * range + bias + 0.5f), 0, max), src/reformat.c:197-219.  v_cvt_u32_f32 truncates toward zero
// and returns 0 for every negative operand: for t >= 0 truncation is the floor, for t < 0 the floor is negative and the
// reference's clamp returns 0 as well; the min restores the upper clamp.
__device__ __forceinline__ int toUNorm(float v, float range, float bias, int maxv)
{
    const float t = ((v * range) + bias) + 0.5f;
    unsigned q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(t));
    return (int)min(q, (unsigned)maxv);
}

template <typename YT>
__device__ __forceinline__ void store4Samples(uint8_t * base, uint32_t off, const int q[4])
{
    if constexpr (sizeof(YT) == 1) {
        __builtin_nontemporal_store((unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24),
                                    reinterpret_cast<unsigned *>(base + off));
    } else {
        const u2 w = { (unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16) };
        __builtin_nontemporal_store(w, reinterpret_cast<u2 *>(base + off));
    }
}
template <typename YT>
__device__ __forceinline__ void store2Samples(uint8_t * base, uint32_t off, int q0, int q1)
{
    if constexpr (sizeof(YT) == 1)
        __builtin_nontemporal_store((uint16_t)((unsigned)q0 | ((unsigned)q1 << 8)), reinterpret_cast<uint16_t *>(base + off));
    else
        __builtin_nontemporal_store((unsigned)q0 | ((unsigned)q1 << 16), reinterpret_cast<unsigned *>(base + off));
}

template <typename RT, int NCH>
struct StripRaw
{
    RawRow<RT, NCH> row[2];
};

template <typename RT, int NCH>
__device__ __forceinline__ void loadStrip(const R2YArgs & A, uint32_t sy, uint32_t Xc, StripRaw<RT, NCH> & S)
{
    constexpr uint32_t kPix = NCH * sizeof(RT);
    const uint32_t syc = sy < A.h2 ? sy : 0; // absent strips load (and discard) the first one
    S.row[0] = loadRow<RT, NCH>(A.rgb, syc * A.rgbPitch + Xc * kPix);
    S.row[1] = loadRow<RT, NCH>(A.rgb, (syc + 1) * A.rgbPitch + Xc * kPix);
}

template <typename RT, int NCH, typename YT, int SUB>
__device__ __forceinline__ void computeStrip(const R2YArgs & A, uint32_t sy, uint32_t X, bool laneValid, const StripRaw<RT, NCH> & S)
{
    constexpr uint32_t BPS = sizeof(YT);
    const int yuvMax = (int)A.yuvMax;
    const bool alphaFirst = (NCH == 4) && (A.slotA == 0);
    const bool swapRB = A.slotB < A.slotR;
    const unsigned colourShift = alphaFirst ? 8u : 0u, alphaShift = alphaFirst ? 0u : 24u;
    (void)colourShift, (void)alphaShift;
    Yuvf c[2][4];
    int yq[2][4], aq[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // memory-order channels -> (first colour, G, last colour, alpha)
            unsigned c0, c1, c2, ca = 0;
            if constexpr (NCH == 4 && sizeof(RT) == 1) {
                // one dword per pixel: alpha-first layouts shift the colour bytes down instead of selecting per channel
                const unsigned w = S.row[r].w[i];
                const unsigned cw = w >> colourShift;
                c0 = cw & 0xffu, c1 = (cw >> 8) & 0xffu, c2 = (cw >> 16) & 0xffu;
                ca = (w >> alphaShift) & 0xffu;
            } else {
                c0 = channelOf<RT, NCH>(S.row[r], i, 0), c1 = channelOf<RT, NCH>(S.row[r], i, 1), c2 = channelOf<RT, NCH>(S.row[r], i, 2);
                if constexpr (NCH == 4) {
                    const unsigned c3 = channelOf<RT, NCH>(S.row[r], i, 3);
                    ca = alphaFirst ? c0 : c3;
                    c0 = alphaFirst ? c1 : c0, c1 = alphaFirst ? c2 : c1, c2 = alphaFirst ? c3 : c2;
                }
            }
            // "Unpack RGB into normalized float", src/reformat.c:312-323: channel / maxChannelF
            const float x = divExact((float)c0, A.rcpRgbMax), G = divExact((float)c1, A.rcpRgbMax), z = divExact((float)c2, A.rcpRgbMax);
            const float R = swapRB ? z : x, B = swapRB ? x : z;
            const float Y = ((A.kr * R) + (A.kg * G)) + (A.kb * B); // :383
            c[r][i].y = Y;
            c[r][i].u = divExact(B - Y, A.rcpCbDen); // (B - Y) / (2 * (1 - kb)), :384
            c[r][i].v = divExact(R - Y, A.rcpCrDen); // (R - Y) / (2 * (1 - kr)), :385
            yq[r][i] = toUNorm(Y, A.rangeY, A.biasY, yuvMax);
            if (A.alphaMode == R2Y_ALPHA_COPY) {
                aq[r][i] = (int)ca; // plain strided copy, src/alpha.c:44-79
            } else if (A.alphaMode == R2Y_ALPHA_RESCALE) {
                const float alphaF = divExact((float)ca, A.rcpRgbMax); // src/alpha.c:93-96
                aq[r][i] = clampInt((int)(0.5f + (alphaF * A.yuvMaxF)), 0, yuvMax);
            } else {
                aq[r][i] = yuvMax; // avifFillAlpha
            }
        }
    }
    if (!laneValid)
        return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        store4Samples<YT>(A.y, (sy + r) * A.yPitch + X * BPS, yq[r]);
        if (A.alphaMode != R2Y_ALPHA_NONE)
            store4Samples<YT>(A.a, (sy + r) * A.aPitch + X * BPS, aq[r]);
    }
    if constexpr (SUB == SUB_444) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int uq[4], vq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uq[i] = toUNorm(c[r][i].u, A.rangeUV, A.biasUV, yuvMax);
                vq[i] = toUNorm(c[r][i].v, A.rangeUV, A.biasUV, yuvMax);
            }
            store4Samples<YT>(A.u, (sy + r) * A.uPitch + X * BPS, uq);
            store4Samples<YT>(A.v, (sy + r) * A.vPitch + X * BPS, vq);
        }
    } else if constexpr (SUB == SUB_420) {
        // sum in the reference's order (bJ outer, bI inner, from 0.0f), then / 4 (exact), src/reformat.c:416-426
        int uq[2], vq[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float su = ((c[0][2 * b].u + c[0][2 * b + 1].u) + c[1][2 * b].u) + c[1][2 * b + 1].u;
            const float sv = ((c[0][2 * b].v + c[0][2 * b + 1].v) + c[1][2 * b].v) + c[1][2 * b + 1].v;
            uq[b] = toUNorm(su * 0.25f, A.rangeUV, A.biasUV, yuvMax);
            vq[b] = toUNorm(sv * 0.25f, A.rangeUV, A.biasUV, yuvMax);
        }
        store2Samples<YT>(A.u, (sy >> 1) * A.uPitch + (X >> 1) * BPS, uq[0], uq[1]);
        store2Samples<YT>(A.v, (sy >> 1) * A.vPitch + (X >> 1) * BPS, vq[0], vq[1]);
    } else if constexpr (SUB == SUB_422) {
#pragma unroll
        for (int r = 0; r < 2; ++r) { // :444-453
            int uq[2], vq[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                uq[b] = toUNorm((c[r][2 * b].u + c[r][2 * b + 1].u) * 0.5f, A.rangeUV, A.biasUV, yuvMax);
                vq[b] = toUNorm((c[r][2 * b].v + c[r][2 * b + 1].v) * 0.5f, A.rangeUV, A.biasUV, yuvMax);
            }
            store2Samples<YT>(A.u, (sy + r) * A.uPitch + (X >> 1) * BPS, uq[0], uq[1]);
            store2Samples<YT>(A.v, (sy + r) * A.vPitch + (X >> 1) * BPS, vq[0], vq[1]);
        }
    }
}

// ---- libyuv's fixed point (8-bit RGB -> 8-bit planes, BT.601, appendix D.5): same loads, stores and strip walk ----
struct Rgb3
{
    int c0, c1, c2;
};
__device__ __forceinline__ int fxDot(const Rgb3 & p, int k0, int k1, int k2, int bias)
{
    // operands fit 24 bits: full-rate multiplies
    return (__mul24(k0, p.c0) + __mul24(k1, p.c1) + __mul24(k2, p.c2) + bias) >> 8;
}

template <int NCH, int SUB>
__device__ __forceinline__ void computeStripFx(const R2YArgs & A, uint32_t sy, uint32_t X, bool laneValid, const StripRaw<uint8_t, NCH> & S)
{
    const bool alphaFirst = (NCH == 4) && (A.slotA == 0);
    const unsigned colourShift = alphaFirst ? 8u : 0u, alphaShift = alphaFirst ? 0u : 24u;
    (void)colourShift, (void)alphaShift;
    const R2YArgs::Fx & F = A.fx;
    Rgb3 px[2][4];
    int yq[2][4], aq[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned ca = 255;
            if constexpr (NCH == 4) {
                const unsigned w = S.row[r].w[i];
                const unsigned cw = w >> colourShift;
                px[r][i].c0 = (int)(cw & 0xffu), px[r][i].c1 = (int)((cw >> 8) & 0xffu), px[r][i].c2 = (int)((cw >> 16) & 0xffu);
                ca = (w >> alphaShift) & 0xffu;
            } else {
                px[r][i].c0 = (int)channelOf<uint8_t, 3>(S.row[r], i, 0), px[r][i].c1 = (int)channelOf<uint8_t, 3>(S.row[r], i, 1);
                px[r][i].c2 = (int)channelOf<uint8_t, 3>(S.row[r], i, 2);
            }
            yq[r][i] = fxDot(px[r][i], F.y0, F.y1, F.y2, F.yBias);
            aq[r][i] = (A.alphaMode == R2Y_ALPHA_COPY) ? (int)ca : 255; // libavif's own alpha pass, src/reformat.c:545-569
        }
    }
    if (!laneValid)
        return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        store4Samples<uint8_t>(A.y, (sy + r) * A.yPitch + X, yq[r]);
        if (A.alphaMode != R2Y_ALPHA_NONE)
            store4Samples<uint8_t>(A.a, (sy + r) * A.aPitch + X, aq[r]);
    }
    if constexpr (SUB == SUB_444) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int uq[4], vq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uq[i] = fxDot(px[r][i], F.u0, F.u1, F.u2, 0x8000);
                vq[i] = fxDot(px[r][i], F.v0, F.v1, F.v2, 0x8000);
            }
            store4Samples<uint8_t>(A.u, (sy + r) * A.uPitch + X, uq);
            store4Samples<uint8_t>(A.v, (sy + r) * A.vPitch + X, vq);
        }
    } else if constexpr (SUB == SUB_420) {
        int uq[2], vq[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) { // the block's RGB is averaged first, per channel, then U and V come from the average
            const Rgb3 &p00 = px[0][2 * b], &p10 = px[0][2 * b + 1], &p01 = px[1][2 * b], &p11 = px[1][2 * b + 1];
            Rgb3 m;
            m.c0 = (p00.c0 + p10.c0 + p01.c0 + p11.c0 + 2) >> 2;
            m.c1 = (p00.c1 + p10.c1 + p01.c1 + p11.c1 + 2) >> 2;
            m.c2 = (p00.c2 + p10.c2 + p01.c2 + p11.c2 + 2) >> 2;
            uq[b] = fxDot(m, F.u0, F.u1, F.u2, 0x8000);
            vq[b] = fxDot(m, F.v0, F.v1, F.v2, 0x8000);
        }
        store2Samples<uint8_t>(A.u, (sy >> 1) * A.uPitch + (X >> 1), uq[0], uq[1]);
        store2Samples<uint8_t>(A.v, (sy >> 1) * A.vPitch + (X >> 1), vq[0], vq[1]);
    } else if constexpr (SUB == SUB_422) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int uq[2], vq[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const Rgb3 &p0 = px[r][2 * b], &p1 = px[r][2 * b + 1];
                Rgb3 m;
                m.c0 = (p0.c0 + p1.c0 + 1) >> 1, m.c1 = (p0.c1 + p1.c1 + 1) >> 1, m.c2 = (p0.c2 + p1.c2 + 1) >> 1;
                uq[b] = fxDot(m, F.u0, F.u1, F.u2, 0x8000);
                vq[b] = fxDot(m, F.v0, F.v1, F.v2, 0x8000);
            }
            store2Samples<uint8_t>(A.u, (sy + r) * A.uPitch + (X >> 1), uq[0], uq[1]);
            store2Samples<uint8_t>(A.v, (sy + r) * A.vPitch + (X >> 1), vq[0], vq[1]);
        }
    }
}

template <int NCH, int SUB>
__global__ __launch_bounds__(256) void rgbToYuvTileFxKernel(R2YArgs A)
{
    const uint32_t bands = (A.w4 + 255) / 256;
    const uint32_t band = blockIdx.x % bands, chunk = blockIdx.x / bands;
    const uint32_t X = band * 256 + 4 * threadIdx.x;
    const bool laneValid = X < A.w4;
    const uint32_t Xc = laneValid ? X : 0;
    StripRaw<uint8_t, NCH> cur;
    uint32_t sy = (chunk * A.stripsPerWave * kWaves + threadIdx.y) * 2;
    loadStrip<uint8_t, NCH>(A, sy, Xc, cur);
    for (uint32_t s = 0; s < A.stripsPerWave; ++s) {
        if (sy >= A.h2)
            break;
        StripRaw<uint8_t, NCH> nxt;
        const bool more = s + 1 < A.stripsPerWave;
        if (more)
            loadStrip<uint8_t, NCH>(A, sy + 2 * kWaves, Xc, nxt);
        computeStripFx<NCH, SUB>(A, sy, X, laneValid, cur);
        if (!more)
            break;
        cur = nxt;
        sy += 2 * kWaves;
    }
}

template <int NCH>
hipError_t launchFxSub(int sub, const R2YArgs & A, uint32_t blocks, hipStream_t stream)
{
    const dim3 block(kLanes, kWaves);
    switch (sub) {
        case SUB_444: hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB_444>), dim3(blocks), block, 0, stream, A); break;
        case SUB_422: hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB_422>), dim3(blocks), block, 0, stream, A); break;
        case SUB_420: hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB_420>), dim3(blocks), block, 0, stream, A); break;
        default: hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB_400>), dim3(blocks), block, 0, stream, A); break;
    }
    return hipGetLastError();
}

template <typename RT, int NCH, typename YT, int SUB>
__global__ __launch_bounds__(256) void rgbToYuvTileKernel(R2YArgs A)
{
    const uint32_t bands = (A.w4 + 255) / 256;
    const uint32_t band = blockIdx.x % bands, chunk = blockIdx.x / bands;
    const uint32_t X = band * 256 + 4 * threadIdx.x;
    const bool laneValid = X < A.w4;
    const uint32_t Xc = laneValid ? X : 0;
    // the 4 waves of a workgroup take adjacent strips; a wave's next strip is 4 strips further down
    const uint32_t first = (chunk * A.stripsPerWave * kWaves + threadIdx.y) * 2;
    StripRaw<RT, NCH> cur;
    uint32_t sy = first;
    loadStrip<RT, NCH>(A, sy, Xc, cur);
    for (uint32_t s = 0; s < A.stripsPerWave; ++s) {
        if (sy >= A.h2)
            break;
        StripRaw<RT, NCH> nxt;
        const bool more = s + 1 < A.stripsPerWave;
        if (more)
            loadStrip<RT, NCH>(A, sy + 2 * kWaves, Xc, nxt);
        computeStrip<RT, NCH, YT, SUB>(A, sy, X, laneValid, cur);
        if (!more)
            break;
        cur = nxt;
        sy += 2 * kWaves;
    }
}

template <typename RT, int NCH, typename YT, int SUB>
hipError_t launchOne(const R2YArgs & A, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB>), dim3(blocks), dim3(kLanes, kWaves), 0, stream, A);
    return hipGetLastError();
}

template <typename RT, int NCH, typename YT>
hipError_t launchSub(int sub, const R2YArgs & A, uint32_t blocks, hipStream_t stream)
{
    switch (sub) {
        case SUB_444: return launchOne<RT, NCH, YT, SUB_444>(A, blocks, stream);
        case SUB_422: return launchOne<RT, NCH, YT, SUB_422>(A, blocks, stream);
        case SUB_420: return launchOne<RT, NCH, YT, SUB_420>(A, blocks, stream);
        default: return launchOne<RT, NCH, YT, SUB_400>(A, blocks, stream);
    }
}

template <typename RT>
hipError_t launchFamily(const R2YKey & k, const R2YArgs & A, uint32_t blocks, hipStream_t stream)
{
    if (k.nch == 4)
        return k.wideYuv ? launchSub<RT, 4, uint16_t>(k.sub, A, blocks, stream) : launchSub<RT, 4, uint8_t>(k.sub, A, blocks, stream);
    return k.wideYuv ? launchSub<RT, 3, uint16_t>(k.sub, A, blocks, stream) : launchSub<RT, 3, uint8_t>(k.sub, A, blocks, stream);
}

} // namespace r2y
} // namespace avifhip
