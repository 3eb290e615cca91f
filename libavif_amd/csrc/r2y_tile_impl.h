// r2y_tile_impl.h -- bandwidth-tuned RGB -> YUV kernels for gfx950 (the encode direction, avifImageRGBToYUV,
// src/reformat.c:275-470 + the alpha plane pass :545-569), instantiated by kernels_r2y_tile_*.hip.
//
// Scope: interleaved 3- or 4-channel RGB at 8-bit or 16-bit containers into 8-bit or 16-bit-container 4:4:4 / 4:2:2 /
// 4:2:0 / 4:0:0 planes with the matrix-coefficient ("normal YUV") transform, the identity matrix, YCgCo and YCgCo-Re / -Ro, a pending alpha
// multiply / un-multiply on the normalised channels (src/reformat.c:325-358), with the alpha plane (copy / depth rescale / opaque fill)
// written in the same pass; gray sources in their own kernel at the end of this file.  Unaligned buffers, divisors off the verified
// list and the <= 3 columns / <= 1 row that do not fill a 4x2 pixel group go to kernels_generic.hip.
//
// Structure: wave = 64 lanes; a lane owns 4 consecutive pixels of 2 rows (two 2x2 chroma blocks): one 16-byte load per
// row for RGBA8 (1 KiB contiguous per wave instruction), one 4-byte luma (and alpha) store per row, one 2-byte store per
// chroma plane.  A wave takes `stripsPerWave` (1, 2 or 4) vertically consecutive strips (256 x 2 pixels), all loads issued up
// front.  No LDS: every input byte is used by exactly one lane.
//
// Arithmetic: the reference's fp32 operations in the reference's order, no contraction; the three divisions by plan
// constants (channel maximum, 2(1-kb), 2(1-kr)) use the exhaustively verified reciprocal form (exactdiv.h).
#pragma once

#include <hip/hip_runtime.h>

#include "pixel_math.h"
#include "r2y_tile_shared.h"

namespace avifhip {
namespace r2y {

// plane stores: streaming (non-temporal) unless -DAVIFHIP_R2Y_PLAIN_STORES (A/B measurements)
template <typename V>
__device__ __forceinline__ void storeOut(V v, V * dst)
{
#ifdef AVIFHIP_R2Y_PLAIN_STORES
    *dst = v;
#else
    __builtin_nontemporal_store(v, dst);
#endif
}

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

// plain loads: streaming (non-temporal) loads, round 1's choice, cost 12 % here (4K RGBA8 -> 4:2:0 10.3 -> 9.0 us, 8K 37.3 -> 33.3 us)
template <typename RT, int NCH>
__device__ __forceinline__ RawRow<RT, NCH> loadRow(const uint8_t * base, uint32_t off)
{
    RawRow<RT, NCH> r;
    constexpr int kWords = RawRow<RT, NCH>::kWords;
    const unsigned * p = reinterpret_cast<const unsigned *>(base + off);
    if constexpr (kWords == 4 || kWords == 8) {
#pragma unroll
        for (int h = 0; h < kWords / 4; ++h) {
            const u4 t = reinterpret_cast<const u4 *>(p)[h];
            r.w[4 * h] = t.x, r.w[4 * h + 1] = t.y, r.w[4 * h + 2] = t.z, r.w[4 * h + 3] = t.w;
        }
    } else if constexpr (kWords == 3) {
        r.w[0] = p[0], r.w[1] = p[1], r.w[2] = p[2];
    } else {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const u2 t = reinterpret_cast<const u2 *>(p)[h];
            r.w[2 * h] = t.x, r.w[2 * h + 1] = t.y;
        }
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
__device__ __forceinline__ int truncClamp(float t, int maxv)
{
    unsigned q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(t));
    return (int)min(q, (unsigned)maxv);
}
__device__ __forceinline__ int toUNorm(float v, float range, float bias, int maxv)
{
    return truncClamp(((v * range) + bias) + 0.5f, maxv);
}

template <typename YT>
__device__ __forceinline__ void store4Samples(uint8_t * base, uint32_t off, const int q[4])
{
    if constexpr (sizeof(YT) == 1) {
        storeOut((unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24),
                                    reinterpret_cast<unsigned *>(base + off));
    } else {
        const u2 w = { (unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16) };
        storeOut(w, reinterpret_cast<u2 *>(base + off));
    }
}
template <typename YT>
__device__ __forceinline__ void store2Samples(uint8_t * base, uint32_t off, int q0, int q1)
{
    if constexpr (sizeof(YT) == 1)
        storeOut((uint16_t)((unsigned)q0 | ((unsigned)q1 << 8)), reinterpret_cast<uint16_t *>(base + off));
    else
        storeOut((unsigned)q0 | ((unsigned)q1 << 16), reinterpret_cast<unsigned *>(base + off));
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

// (3-channel pixels are read with three dword loads per lane and row.  The decode direction's cure for 3-byte STORES -- 16-byte accesses
//  at consecutive addresses, redistributed through a wave-private LDS buffer -- was measured for these LOADS and is slower: 8K RGB8 ->
//  4:2:0 32.5 us against 29.7 us, 4K 9.5 against 8.8: partial-line reads are absorbed by the caches, partial-line writes are not.)

// ---- fp32 arithmetic on pixel PAIRS: gfx950 multiplies and adds two fp32 lanes per instruction (v_pk_mul_f32 / v_pk_add_f32 /
//      v_pk_fma_f32, each component rounded exactly like the scalar instruction), so horizontally adjacent pixels share every
//      normalisation, matrix and quantisation instruction ----
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float v)
{
    return (f2) { v, v };
}
// x / d for a plan constant d on the verified list (exactdiv.h), both components: bit-identical to the IEEE quotients
__device__ __forceinline__ f2 div2(f2 x, RcpHL r)
{
    return __builtin_elementwise_fma(x, splat2(r.hi), x * splat2(r.lo));
}
// t = v * range + bias + 0.5f, the operand of the reference's floorf (src/reformat.c:197-219), in the reference's order
__device__ __forceinline__ f2 unormOperand(f2 v, float range, float bias)
{
    return ((v * splat2(range)) + splat2(bias)) + splat2(0.5f);
}
// AVIF_CLAMP((int)floorf(t), 0, 255) of four operands, packed little-endian.  v_cvt_pk_u8_f32 converts with the current rounding
// mode and saturates to [0, 255]: under round-toward-zero that is the floor for t >= 0 and 0 for every negative t -- the same
// byte as the reference's floor-then-clamp.  The rounding mode is changed only inside the block.
__device__ __forceinline__ unsigned packU8x4(float a, float b, float c, float d)
{
    unsigned w;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 3, %0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w)
                 : "v"(a), "v"(b), "v"(c), "v"(d));
    return w;
}

// four samples of a row from two operand pairs
template <typename YT>
__device__ __forceinline__ void storeRow4(uint8_t * base, uint32_t off, f2 t01, f2 t23, int maxv)
{
    if constexpr (sizeof(YT) == 1) {
        storeOut(packU8x4(t01.x, t01.y, t23.x, t23.y), reinterpret_cast<unsigned *>(base + off));
    } else {
        const int q[4] = { truncClamp(t01.x, maxv), truncClamp(t01.y, maxv), truncClamp(t23.x, maxv), truncClamp(t23.y, maxv) };
        store4Samples<YT>(base, off, q);
    }
}

// Pending alpha (un)multiply on the normalised channels of a pixel pair (src/reformat.c:325-358); a = alpha / max.
//   multiply:   a == 0 -> 0, a < 1 -> c * a, else c: the product with min(a, 1) is all three cases.
//   unmultiply: a == 0 -> 0, a < 1 -> min(c / a, 1), else c.  The three divisions share their divisor: r = RN(1 / a) from v_rcp_f32 and one
//               Newton step, then q = fma(fma(-q0, a, c), r, q0), q0 = c * r, is the correctly rounded quotient -- enumerated for every pair
//               of channel codes (c, a) of every depth (tests/tools/verify_fp32_shortcuts.cpp).
// (per pixel: the divisor d -- 1 where the reference leaves the channel alone or answers 0 --, its exact reciprocal r -- 0 where the answer
//  is 0, so that the quotient comes out as 0 without a select per channel -- and the upper clamp: 1 below full alpha, none otherwise)
struct UnmulOperand
{
    float d, r, lim;
};
__device__ __forceinline__ UnmulOperand unmulOperand(float a)
{
    const bool zero = a == 0.0f, below1 = a < 1.0f;
    UnmulOperand U;
    U.d = (zero || !below1) ? 1.0f : a;
    const float r0 = __builtin_amdgcn_rcpf(U.d);
    const float r = __builtin_fmaf(__builtin_fmaf(-U.d, r0, 1.0f), r0, r0);
    U.r = zero ? 0.0f : r;
    U.lim = below1 ? 1.0f : __builtin_inff();
    return U;
}
__device__ __forceinline__ float unmulChannel(float c, const UnmulOperand & U)
{
    const float q0 = c * U.r;
    return fminf(__builtin_fmaf(__builtin_fmaf(-q0, U.d, c), U.r, q0), U.lim);
}
// ... with the divisor and its reciprocal of every 8-bit alpha code in LDS (unmulTable: built by the workgroup when the kernel starts, by
// the instructions above, so the entries ARE unmulOperand's): the pixel reads its pair instead of forming it -- two compares, three selects,
// v_rcp_f32 (a quarter-rate instruction) and its Newton step less per pixel -- and the three channels of a pixel pair go through the packed
// multiply / fused multiply-add.  The upper clamp is min(q, 1) everywhere: where unmulOperand lifts it (a == 1) the quotient is c <= 1.
struct UnmulEntry
{
    float d, r;
};
__device__ __forceinline__ void unmulPairFromTable(const UnmulEntry * table, unsigned a0, unsigned a1, f2 & x, f2 & g, f2 & z)
{
    const UnmulEntry e0 = table[a0], e1 = table[a1];
    const f2 d = { e0.d, e1.d }, r = { e0.r, e1.r }, one = splat2(1.0f);
    const f2 qx = x * r, qg = g * r, qz = z * r;
    x = __builtin_elementwise_min(__builtin_elementwise_fma(__builtin_elementwise_fma(-qx, d, x), r, qx), one);
    g = __builtin_elementwise_min(__builtin_elementwise_fma(__builtin_elementwise_fma(-qg, d, g), r, qg), one);
    z = __builtin_elementwise_min(__builtin_elementwise_fma(__builtin_elementwise_fma(-qz, d, z), r, qz), one);
}
__device__ __forceinline__ void alphaOnPair(bool multiply, f2 a, f2 & x, f2 & g, f2 & z)
{
    if (multiply) {
        const f2 am = { fminf(a.x, 1.0f), fminf(a.y, 1.0f) };
        x = x * am, g = g * am, z = z * am;
        return;
    }
    const UnmulOperand U0 = unmulOperand(a.x), U1 = unmulOperand(a.y);
    x = (f2) { unmulChannel(x.x, U0), unmulChannel(x.y, U1) };
    g = (f2) { unmulChannel(g.x, U0), unmulChannel(g.y, U1) };
    z = (f2) { unmulChannel(z.x, U0), unmulChannel(z.y, U1) };
}
// (int)avifRoundf(AVIF_CLAMP(c * maxF, 0, maxF)): a normalised channel back to its code for the YCgCo-R lifting, src/reformat.c:375-377
__device__ __forceinline__ int liftCode(float c, float maxf)
{
    const float v = c * maxf;
    const float cl = (v < 0.0f) ? 0.0f : ((maxf < v) ? maxf : v);
    return (int)floorf(cl + 0.5f);
}

// PLAIN: matrix coefficients or the identity matrix and no pending alpha arithmetic -- what nearly every encode is.  It is a kernel of its
// own (launchOne picks): with the YCgCo family and the alpha (un)multiply compiled into the same loop as wave-uniform branches, the
// one-strip kernel that serves 4K frames took 120 vector registers (half the occupancy) and cfg4 went from 9.0 to 14.9 us.
// IDENT (PLAIN kernels only): the identity matrix, known when the code is generated -- left as a run-time test the compiler computes BOTH
// transforms and picks per value (three v_cndmask_b32 per pixel of the 25 instructions the matrix-coefficient path took).
template <typename RT, int NCH, typename YT, int SUB, bool SWAP, bool PLAIN, bool IDENT>
__device__ __forceinline__ void computeStripT(const R2YArgs & A, uint32_t sy, uint32_t X, bool laneValid, const StripRaw<RT, NCH> & S, const UnmulEntry * unmulTable)
{
    constexpr uint32_t BPS = sizeof(YT);
    const int yuvMax = (int)A.yuvMax;
    const bool alphaFirst = (NCH == 4) && (A.slotA == 0);
    const unsigned colourShift = alphaFirst ? 8u : 0u;
    (void)colourShift;
    f2 tY[2][2], U[2][2], V[2][2]; // [row][pixel pair]: luma as the quantiser's operand, chroma normalised
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // memory-order colour channels (first colour, G, last colour) of the pair's two pixels
            unsigned c0[2], c1[2], c2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * p + h;
                if constexpr (NCH == 4 && sizeof(RT) == 1) {
                    // one dword per pixel: alpha-first layouts shift the colour bytes down instead of selecting per channel
                    const unsigned cw = S.row[r].w[i] >> colourShift;
                    c0[h] = cw & 0xffu, c1[h] = (cw >> 8) & 0xffu, c2[h] = (cw >> 16) & 0xffu;
                } else {
                    c0[h] = channelOf<RT, NCH>(S.row[r], i, 0), c1[h] = channelOf<RT, NCH>(S.row[r], i, 1), c2[h] = channelOf<RT, NCH>(S.row[r], i, 2);
                    if constexpr (NCH == 4) {
                        const unsigned c3 = channelOf<RT, NCH>(S.row[r], i, 3);
                        c0[h] = alphaFirst ? c1[h] : c0[h], c1[h] = alphaFirst ? c2[h] : c1[h], c2[h] = alphaFirst ? c3 : c2[h];
                    }
                }
            }
            // "Unpack RGB into normalized float", src/reformat.c:312-323: channel / maxChannelF
            const f2 x = div2((f2) { (float)c0[0], (float)c0[1] }, A.rcpRgbMax);
            const f2 G = div2((f2) { (float)c1[0], (float)c1[1] }, A.rcpRgbMax);
            const f2 z = div2((f2) { (float)c2[0], (float)c2[1] }, A.rcpRgbMax);
            f2 xs = x, Gs = G, zs = z;
            if constexpr (NCH == 4 && !PLAIN) {
                if (A.mulMode != MUL_NONE) { // wave-uniform: pending alpha (un)multiply on the normalised channels, src/reformat.c:325-358
                    unsigned ca[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        ca[h] = channelOf<RT, NCH>(S.row[r], 2 * p + h, alphaFirst ? 0 : 3);
                    if (sizeof(RT) == 1 && A.mulMode == MUL_UNMULTIPLY) { // (8-bit channels: the alpha code addresses the workgroup's table)
                        unmulPairFromTable(unmulTable, ca[0], ca[1], xs, Gs, zs);
                    } else {
                        const f2 a = div2((f2) { (float)ca[0], (float)ca[1] }, A.rcpRgbMax);
                        alphaOnPair(A.mulMode == MUL_MULTIPLY, a, xs, Gs, zs);
                    }
                }
            }
            const f2 R = SWAP ? zs : xs, B = SWAP ? xs : zs;
            const f2 G2 = Gs;
            if (!PLAIN && A.matrixMode == MODE_YCGCO) { // wave-uniform, src/reformat.c:368-372: 0.5 G +- 0.25 (R + B), 0.5 (R - B)
                const f2 hg = splat2(0.5f) * G2, q = splat2(0.25f) * (R + B);
                U[r][p] = hg - q, V[r][p] = splat2(0.5f) * (R - B);
                tY[r][p] = unormOperand(hg + q, A.rangeY, A.biasY);
            } else if (!PLAIN && (A.matrixMode == MODE_YCGCO_RE || A.matrixMode == MODE_YCGCO_RO)) { // integer lifting on the channel codes, :373-383
                float yv[2], uv2[2], vv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int Ri = liftCode(h ? R.y : R.x, A.rgbMaxF), Gi = liftCode(h ? G2.y : G2.x, A.rgbMaxF), Bi = liftCode(h ? B.y : B.x, A.rgbMaxF);
                    const int Co = Ri - Bi, t = Bi + (Co >> 1), Cg = Gi - t;
                    yv[h] = divExact((float)(t + (Cg >> 1)), A.rcpRangeY);
                    uv2[h] = divExact((float)Cg, A.rcpRangeUV);
                    vv[h] = divExact((float)Co, A.rcpRangeUV);
                }
                U[r][p] = (f2) { uv2[0], uv2[1] }, V[r][p] = (f2) { vv[0], vv[1] };
                tY[r][p] = unormOperand((f2) { yv[0], yv[1] }, A.rangeY, A.biasY);
            } else if (PLAIN ? IDENT : (A.identity != 0)) { // wave-uniform: GBR planes (lossless RGB), src/reformat.c:362-366 -- Y = G, U = B, V = R, all three on luma's scale
                U[r][p] = B, V[r][p] = R;
                tY[r][p] = unormOperand(G2, A.rangeY, A.biasY);
            } else {
                const f2 Y = ((splat2(A.kr) * R) + (splat2(A.kg) * G2)) + (splat2(A.kb) * B); // :383
                U[r][p] = div2(B - Y, A.rcpCbDen); // (B - Y) / (2 * (1 - kb)), :384
                V[r][p] = div2(R - Y, A.rcpCrDen); // (R - Y) / (2 * (1 - kr)), :385
                tY[r][p] = unormOperand(Y, A.rangeY, A.biasY);
            }
        }
    }
    // alpha plane: four samples per row
    unsigned aw[2] = { 0, 0 };
    int aq[2][4];
    const bool alphaBytes = NCH == 4 && sizeof(RT) == 1 && sizeof(YT) == 1 && A.alphaMode == R2Y_ALPHA_COPY; // wave-uniform
    if (A.alphaMode != R2Y_ALPHA_NONE) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (NCH == 4 && sizeof(RT) == 1 && sizeof(YT) == 1) {
                if (alphaBytes) { // plain strided copy, src/alpha.c:44-79: the four alpha bytes gathered by three byte permutes
                    const unsigned sel = alphaFirst ? 0x0c0c0400u : 0x0c0c0703u;
                    const unsigned a01 = __builtin_amdgcn_perm(S.row[r].w[1], S.row[r].w[0], sel), a23 = __builtin_amdgcn_perm(S.row[r].w[3], S.row[r].w[2], sel);
                    aw[r] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
                    continue;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned ca = 0;
                if constexpr (NCH == 4)
                    ca = channelOf<RT, NCH>(S.row[r], i, alphaFirst ? 0 : 3);
                if (A.alphaMode == R2Y_ALPHA_COPY) {
                    aq[r][i] = (int)ca;
                } else if (A.alphaMode == R2Y_ALPHA_RESCALE) {
                    const float alphaF = divExact((float)ca, A.rcpRgbMax); // src/alpha.c:93-96
                    aq[r][i] = clampInt((int)(0.5f + (alphaF * A.yuvMaxF)), 0, yuvMax);
                } else {
                    aq[r][i] = yuvMax; // avifFillAlpha
                }
            }
        }
    }
    if (!laneValid)
        return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        storeRow4<YT>(A.y, (sy + r) * A.yPitch + X * BPS, tY[r][0], tY[r][1], yuvMax);
        if (A.alphaMode != R2Y_ALPHA_NONE) {
            if (alphaBytes)
                storeOut(aw[r], reinterpret_cast<unsigned *>(A.a + ((sy + r) * A.aPitch + X)));
            else
                store4Samples<YT>(A.a, (sy + r) * A.aPitch + X * BPS, aq[r]);
        }
    }
    if constexpr (SUB == SUB_444) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            storeRow4<YT>(A.u, (sy + r) * A.uPitch + X * BPS, unormOperand(U[r][0], A.rangeUV, A.biasUV), unormOperand(U[r][1], A.rangeUV, A.biasUV), yuvMax);
            storeRow4<YT>(A.v, (sy + r) * A.vPitch + X * BPS, unormOperand(V[r][0], A.rangeUV, A.biasUV), unormOperand(V[r][1], A.rangeUV, A.biasUV), yuvMax);
        }
    } else if constexpr (SUB == SUB_420) {
        // sum in the reference's order (bJ outer, bI inner, from 0.0f), then / 4 (exact), src/reformat.c:416-426; block b is pixel pair b
        // of both rows.  (u of block 0, u of block 1) and (v, v) then share the quantiser's instructions.
        const f2 su = { ((U[0][0].x + U[0][0].y) + U[1][0].x) + U[1][0].y, ((U[0][1].x + U[0][1].y) + U[1][1].x) + U[1][1].y };
        const f2 sv = { ((V[0][0].x + V[0][0].y) + V[1][0].x) + V[1][0].y, ((V[0][1].x + V[0][1].y) + V[1][1].x) + V[1][1].y };
        const f2 tu = unormOperand(su * splat2(0.25f), A.rangeUV, A.biasUV), tv = unormOperand(sv * splat2(0.25f), A.rangeUV, A.biasUV);
        if constexpr (sizeof(YT) == 1) {
            const unsigned uv = packU8x4(tu.x, tu.y, tv.x, tv.y);
            storeOut((uint16_t)uv, reinterpret_cast<uint16_t *>(A.u + ((sy >> 1) * A.uPitch + (X >> 1))));
            storeOut((uint16_t)(uv >> 16), reinterpret_cast<uint16_t *>(A.v + ((sy >> 1) * A.vPitch + (X >> 1))));
        } else {
            store2Samples<YT>(A.u, (sy >> 1) * A.uPitch + (X >> 1) * BPS, truncClamp(tu.x, yuvMax), truncClamp(tu.y, yuvMax));
            store2Samples<YT>(A.v, (sy >> 1) * A.vPitch + (X >> 1) * BPS, truncClamp(tv.x, yuvMax), truncClamp(tv.y, yuvMax));
        }
    } else if constexpr (SUB == SUB_422) {
#pragma unroll
        for (int r = 0; r < 2; ++r) { // :444-453
            const f2 su = { U[r][0].x + U[r][0].y, U[r][1].x + U[r][1].y }, sv = { V[r][0].x + V[r][0].y, V[r][1].x + V[r][1].y };
            const f2 tu = unormOperand(su * splat2(0.5f), A.rangeUV, A.biasUV), tv = unormOperand(sv * splat2(0.5f), A.rangeUV, A.biasUV);
            store2Samples<YT>(A.u, (sy + r) * A.uPitch + (X >> 1) * BPS, truncClamp(tu.x, yuvMax), truncClamp(tu.y, yuvMax));
            store2Samples<YT>(A.v, (sy + r) * A.vPitch + (X >> 1) * BPS, truncClamp(tv.x, yuvMax), truncClamp(tv.y, yuvMax));
        }
    }
}

template <typename RT, int NCH, typename YT, int SUB, bool PLAIN, bool IDENT>
__device__ __forceinline__ void computeStrip(const R2YArgs & A, uint32_t sy, uint32_t X, bool laneValid, const StripRaw<RT, NCH> & S, const UnmulEntry * unmulTable)
{
    // which memory-order colour channel is red decides the operand ORDER of the luma sum (fp32 addition is not associative):
    // one wave-uniform branch instead of selects per pixel.
    // (IDENT -- the identity matrix in the PLAIN kernels, 4:4:4 or 4:0:0 planes only, src/reformat.c:141-144 -- is a kernel of its own since
    //  round 6: as a branch inside the PLAIN kernels (round 5) it cost them their registers -- 72 against 60, 110 with one strip per wave --
    //  and the lossless encode ran at 44-53 us per 8K frame where the all-modes kernel took 38.4)
    static_assert(!IDENT || (PLAIN && (SUB == SUB_444 || SUB == SUB_400)), "the identity matrix takes 4:4:4 or 4:0:0 planes");
    if (A.slotB < A.slotR)
        computeStripT<RT, NCH, YT, SUB, true, PLAIN, IDENT>(A, sy, X, laneValid, S, unmulTable);
    else
        computeStripT<RT, NCH, YT, SUB, false, PLAIN, IDENT>(A, sy, X, laneValid, S, unmulTable);
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

// Tile order.  Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8, observed; only speed depends on it).  In plain raster order
// the 8 XCDs interleave their accesses tile by tile over the whole frame; giving XCD x the x-th contiguous run of tiles lets every XCD (its own
// L2, its own share of the memory channels' queues) walk one band of the image.  The byte-movement ceiling of the encode direction's shape runs
// 15 % faster that way on 4K frames that stream (tests/tools/stream_sweep.py cfg4: 9.05 us in per-XCD bands, 10.4 in raster order).
__device__ __forceinline__ uint32_t r2yBlockOf(uint32_t b, uint32_t n, bool xcdBands)
{
    if (!xcdBands || n < 64)
        return b;
    const uint32_t per = n >> 3, rem = n & 7, xcd = b & 7, slot = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

// the job of a workgroup of a sequence launch (r2y_tile_shared.h R2YSeqFrames): the launch's arguments with the addresses of frame blockIdx.z
__device__ __forceinline__ R2YArgs r2ySeqJob(const R2YArgs & A, const R2YSeqFrames & S)
{
    R2YArgs job = A;
    const R2YSeqFrames::Frame f = S.f[blockIdx.z];
    job.rgb = f.rgb, job.y = f.y, job.u = f.u, job.v = f.v, job.a = f.a;
    return job;
}

template <int NCH, int SUB, int NS>
__global__ __launch_bounds__(256) void rgbToYuvTileFxKernel(R2YArgs A0, R2YSeqFrames S)
{
    const R2YArgs A = r2ySeqJob(A0, S);
    const uint32_t bands = (A.w4 + 255) / 256;
    const uint32_t block = r2yBlockOf(blockIdx.x, gridDim.x, A.xcdBands != 0);
    const uint32_t band = block % bands, chunk = block / bands;
    const uint32_t X = band * 256 + 4 * threadIdx.x;
    const bool laneValid = X < A.w4;
    const uint32_t Xc = laneValid ? X : 0;
    const uint32_t first = (chunk * kWaves + threadIdx.y) * (2 * NS);
    if (first >= A.h2)
        return;
    StripRaw<uint8_t, NCH> raw[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        loadStrip<uint8_t, NCH>(A, first + 2 * s, Xc, raw[s]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (first + 2 * s >= A.h2)
            break;
        computeStripFx<NCH, SUB>(A, first + 2 * s, X, laneValid, raw[s]);
    }
}

template <int NCH, int SUB>
void launchFxOne(const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    const dim3 block(kLanes, kWaves), grid(blocks, 1, frames);
    if (A.stripsPerWave >= 4)
        hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB, 4>), grid, block, 0, stream, A, S);
    else if (A.stripsPerWave >= 2)
        hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB, 2>), grid, block, 0, stream, A, S);
    else
        hipLaunchKernelGGL((rgbToYuvTileFxKernel<NCH, SUB, 1>), grid, block, 0, stream, A, S);
}

template <int NCH>
hipError_t launchFxSub(int sub, const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    switch (sub) {
        case SUB_444: launchFxOne<NCH, SUB_444>(A, blocks, stream, S, frames); break;
        case SUB_422: launchFxOne<NCH, SUB_422>(A, blocks, stream, S, frames); break;
        case SUB_420: launchFxOne<NCH, SUB_420>(A, blocks, stream, S, frames); break;
        default: launchFxOne<NCH, SUB_400>(A, blocks, stream, S, frames); break;
    }
    return hipGetLastError();
}

// One wave, one tile of 256 x 2*NS pixels (NS vertically consecutive strips), every load issued before the first result is needed; the
// four waves of a workgroup are stacked and independent.  (Round 1 walked a wave down its strips with the next strip's loads in
// flight; like in the decode direction, many short-lived waves keep the memory pipes fuller: 4K RGBA8 -> 4:2:0 10.9 -> 9.9 us.)
template <typename RT, int NCH, typename YT, int SUB, int NS, bool PLAIN, bool IDENT = false>
__global__ __launch_bounds__(256) void rgbToYuvTileKernel(R2YArgs A0, R2YSeqFrames S)
{
    const R2YArgs A = r2ySeqJob(A0, S);
    const uint32_t bands = (A.w4 + 255) / 256;
    const uint32_t block = r2yBlockOf(blockIdx.x, gridDim.x, A.xcdBands != 0);
    const uint32_t band = block % bands, chunk = block / bands;
    const uint32_t X = band * 256 + 4 * threadIdx.x;
    const bool laneValid = X < A.w4;
    const uint32_t Xc = laneValid ? X : 0;
    const uint32_t first = (chunk * kWaves + threadIdx.y) * (2 * NS);
    // pending un-multiply of 8-bit pixels: (divisor, reciprocal) of every alpha code, one entry per thread of the workgroup (before any
    // wave leaves: the barrier is the workgroup's)
    constexpr bool kTable = !PLAIN && NCH == 4 && sizeof(RT) == 1;
    __shared__ UnmulEntry unmulTable[kTable ? 256 : 1];
    const bool tabled = kTable && A.mulMode == MUL_UNMULTIPLY; // wave-uniform
    if (first >= A.h2 && !tabled)
        return;
    StripRaw<RT, NCH> raw[NS];
    if (first < A.h2) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            loadStrip<RT, NCH>(A, first + 2 * s, Xc, raw[s]);
    }
    if constexpr (kTable) {
        if (tabled) {
            const unsigned code = threadIdx.y * kLanes + threadIdx.x; // 0..255 (workgroups are kLanes x kWaves = 64 x 4)
            const float a = divExact((float)code, A.rcpRgbMax);
            const UnmulOperand U = unmulOperand(a);
            unmulTable[code] = { U.d, U.r };
            __syncthreads();
            if (first >= A.h2)
                return;
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (first + 2 * s >= A.h2) // wave-uniform
            break;
        computeStrip<RT, NCH, YT, SUB, PLAIN, IDENT>(A, first + 2 * s, X, laneValid, raw[s], unmulTable);
    }
}

template <typename RT, int NCH, typename YT, int SUB, bool PLAIN>
hipError_t launchOnePlainOrNot(const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    const dim3 grid(blocks, 1, frames), block(kLanes, kWaves);
    if constexpr (PLAIN && (SUB == SUB_444 || SUB == SUB_400)) {
        if (A.identity) { // lossless RGB in GBR planes (avifenc -l): its own kernels
            if (A.stripsPerWave >= 4)
                hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 4, true, true>), grid, block, 0, stream, A, S);
            else if (A.stripsPerWave >= 2)
                hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 2, true, true>), grid, block, 0, stream, A, S);
            else
                hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 1, true, true>), grid, block, 0, stream, A, S);
            return hipGetLastError();
        }
    }
    if (A.stripsPerWave >= 4)
        hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 4, PLAIN>), grid, block, 0, stream, A, S);
    else if (A.stripsPerWave >= 2)
        hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 2, PLAIN>), grid, block, 0, stream, A, S);
    else if constexpr (PLAIN)
        hipLaunchKernelGGL((rgbToYuvTileKernel<RT, NCH, YT, SUB, 1, PLAIN>), grid, block, 0, stream, A, S);
    else
        return hipErrorInvalidValue; // (the rare modes run two strips per wave or more: kernels_r2y_tile.hip)
    return hipGetLastError();
}

template <typename RT, int NCH, typename YT, int SUB>
hipError_t launchOne(const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    const bool plain = A.mulMode == MUL_NONE && A.matrixMode != MODE_YCGCO && A.matrixMode != MODE_YCGCO_RE && A.matrixMode != MODE_YCGCO_RO;
    return plain ? launchOnePlainOrNot<RT, NCH, YT, SUB, true>(A, blocks, stream, S, frames) : launchOnePlainOrNot<RT, NCH, YT, SUB, false>(A, blocks, stream, S, frames);
}

template <typename RT, int NCH, typename YT>
hipError_t launchSub(int sub, const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    switch (sub) {
        case SUB_444: return launchOne<RT, NCH, YT, SUB_444>(A, blocks, stream, S, frames);
        case SUB_422: return launchOne<RT, NCH, YT, SUB_422>(A, blocks, stream, S, frames);
        case SUB_420: return launchOne<RT, NCH, YT, SUB_420>(A, blocks, stream, S, frames);
        default: return launchOne<RT, NCH, YT, SUB_400>(A, blocks, stream, S, frames);
    }
}

template <typename RT>
hipError_t launchFamily(const R2YKey & k, const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    if (k.nch == 4)
        return k.wideYuv ? launchSub<RT, 4, uint16_t>(k.sub, A, blocks, stream, S, frames) : launchSub<RT, 4, uint8_t>(k.sub, A, blocks, stream, S, frames);
    return k.wideYuv ? launchSub<RT, 3, uint16_t>(k.sub, A, blocks, stream, S, frames) : launchSub<RT, 3, uint8_t>(k.sub, A, blocks, stream, S, frames);
}

// ---- gray sources (GRAY / GRAYA / AGRAY -> the luma plane, src/reformat.c:471-519; the chroma planes, if any, are set to the half value by
//      the caller and the alpha plane is written here: copy / depth rescale / opaque fill).  A lane owns the PPL pixels of one 16-byte load,
//      a wave 64 of them of ROWS consecutive rows (all loads issued first); luma and alpha leave as 16-byte stores where PPL samples fill
//      them.  Arithmetic: g = code / max in the verified reciprocal form, the pending alpha (un)multiply of alphaOnPair, the quantiser
//      of the colour kernels ----
template <typename RT, int GCH, typename YT, bool AFIRST>
__global__ __launch_bounds__(256) void grayToYuvTileKernel(GrayArgs A)
{
    constexpr int PPL = 16 / (GCH * (int)sizeof(RT)); // pixels per lane: 16 (GRAY8), 8 (GRAYA8, GRAY16), 4 (GRAYA16)
    constexpr int ROWS = 2;
    const uint32_t X = (blockIdx.x * kLanes + threadIdx.x) * PPL;
    const uint32_t row0 = (blockIdx.y * kWaves + threadIdx.y) * ROWS;
    if (X >= A.wP || row0 >= A.height)
        return;
    u4 raw[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t j = (row0 + r < A.height) ? row0 + r : row0;
        raw[r] = *reinterpret_cast<const u4 *>(A.gray + (size_t)j * A.grayPitch + (size_t)X * (GCH * sizeof(RT)));
    }
    // (every branch below is wave-uniform and spans all of a row's pixels: a first version that tested the alpha mode per pixel spent its
    //  time in 800 scalar branches per wave -- 37 us for an 8K GRAY8 frame)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (row0 + r >= A.height)
            break;
        const unsigned w[4] = { raw[r].x, raw[r].y, raw[r].z, raw[r].w };
        // channel k (0 .. GCH-1, memory order) of pixel i
        auto chan = [&](int i, int k) -> unsigned {
            const int idx = i * GCH + k;
            if constexpr (sizeof(RT) == 1)
                return (w[idx >> 2] >> (8 * (idx & 3))) & 0xffu;
            else
                return (w[idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
        };
        float g[PPL];
        unsigned ca[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            g[i] = divExact((float)chan(i, (GCH == 2 && AFIRST) ? 1 : 0), A.rcpRgbMax);
            ca[i] = (GCH == 2) ? chan(i, AFIRST ? 0 : 1) : 0u;
        }
        if constexpr (GCH == 2) {
            if (A.mulMode == MUL_MULTIPLY) {
#pragma unroll
                for (int i = 0; i < PPL; ++i)
                    g[i] = g[i] * fminf(divExact((float)ca[i], A.rcpRgbMax), 1.0f);
            } else if (A.mulMode == MUL_UNMULTIPLY) {
#pragma unroll
                for (int i = 0; i < PPL; ++i)
                    g[i] = unmulChannel(g[i], unmulOperand(divExact((float)ca[i], A.rcpRgbMax)));
            }
        }
        // the lane's PPL samples of a plane row as dwords, stored 16 bytes at a time where that many exist (4-byte stores at a lane stride of
        // 16 made every store instruction touch a quarter of each cache line: 31 us for an 8K GRAY8 frame)
        constexpr int kWords = PPL * (int)sizeof(YT) / 4; // 4 (GRAY8 -> 8-bit), 8 (GRAY8 -> 16-bit), 2, 4, 1 or 2
        auto storePlaneRow = [&](uint8_t * dst, const unsigned (&wd)[kWords]) {
            if constexpr (kWords >= 4) {
#pragma unroll
                for (int h = 0; h < kWords / 4; ++h)
                    storeOut((u4) { wd[4 * h], wd[4 * h + 1], wd[4 * h + 2], wd[4 * h + 3] }, reinterpret_cast<u4 *>(dst + 16 * h));
            } else if constexpr (kWords == 2) {
                storeOut((u2) { wd[0], wd[1] }, reinterpret_cast<u2 *>(dst));
            } else {
                storeOut(wd[0], reinterpret_cast<unsigned *>(dst));
            }
        };
        unsigned yw[kWords];
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                t[i] = ((g[4 * q + i] * A.rangeY) + A.biasY) + 0.5f;
            if constexpr (sizeof(YT) == 1) {
                yw[q] = packU8x4(t[0], t[1], t[2], t[3]);
            } else {
                yw[2 * q] = (unsigned)truncClamp(t[0], (int)A.yuvMax) | ((unsigned)truncClamp(t[1], (int)A.yuvMax) << 16);
                yw[2 * q + 1] = (unsigned)truncClamp(t[2], (int)A.yuvMax) | ((unsigned)truncClamp(t[3], (int)A.yuvMax) << 16);
            }
        }
        storePlaneRow(A.y + (size_t)(row0 + r) * A.yPitch + (size_t)X * sizeof(YT), yw);
        if (A.alphaMode != R2Y_ALPHA_NONE) {
            unsigned aq[PPL];
            if (A.alphaMode == R2Y_ALPHA_COPY) {
#pragma unroll
                for (int i = 0; i < PPL; ++i)
                    aq[i] = ca[i];
            } else if (A.alphaMode == R2Y_ALPHA_RESCALE) {
#pragma unroll
                for (int i = 0; i < PPL; ++i)
                    aq[i] = (unsigned)clampInt((int)(0.5f + (divExact((float)ca[i], A.rcpRgbMax) * A.yuvMaxF)), 0, (int)A.yuvMax);
            } else {
#pragma unroll
                for (int i = 0; i < PPL; ++i)
                    aq[i] = A.yuvMax;
            }
            unsigned aw[kWords];
#pragma unroll
            for (int q = 0; q < PPL / 4; ++q) {
                if constexpr (sizeof(YT) == 1) {
                    aw[q] = aq[4 * q] | (aq[4 * q + 1] << 8) | (aq[4 * q + 2] << 16) | (aq[4 * q + 3] << 24);
                } else {
                    aw[2 * q] = aq[4 * q] | (aq[4 * q + 1] << 16);
                    aw[2 * q + 1] = aq[4 * q + 2] | (aq[4 * q + 3] << 16);
                }
            }
            storePlaneRow(A.a + (size_t)(row0 + r) * A.aPitch + (size_t)X * sizeof(YT), aw);
        }
    }
}

template <typename RT>
hipError_t launchGray(int gch, bool wideYuv, const GrayArgs & A, hipStream_t stream)
{
    const uint32_t ppl = 16u / ((uint32_t)gch * (uint32_t)sizeof(RT));
    const dim3 block(kLanes, kWaves), grid((A.wP / ppl + kLanes - 1) / kLanes, (A.height + 2 * kWaves - 1) / (2 * kWaves));
    if (gch == 1) {
        if (wideYuv)
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 1, uint16_t, false>), grid, block, 0, stream, A);
        else
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 1, uint8_t, false>), grid, block, 0, stream, A);
    } else if (A.alphaFirst) {
        if (wideYuv)
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 2, uint16_t, true>), grid, block, 0, stream, A);
        else
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 2, uint8_t, true>), grid, block, 0, stream, A);
    } else {
        if (wideYuv)
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 2, uint16_t, false>), grid, block, 0, stream, A);
        else
            hipLaunchKernelGGL((grayToYuvTileKernel<RT, 2, uint8_t, false>), grid, block, 0, stream, A);
    }
    return hipGetLastError();
}

} // namespace r2y
} // namespace avifhip
