// tile_pk_impl.h -- the integer path's tiled YUV->RGB kernels for 8-bit (and, behind a front end, 10/12-bit) planes and 8-bit RGB outputs, in wavefront-wide PACKED
// 16-bit arithmetic (v_pk_mad_i16 / v_pk_ashrrev_i16 / v_sat_pk_u8_i16 / v_perm_b32), instantiated by kernels_tile_fx_inst.hip.
// What it computes is libyuv's fixed point exactly (I420ToARGBMatrixFilter and its relatives as
// src/reformat_libyuv.c:544-1108 dispatches them):
//     y1 = ((y * 0x0101 * yg) >> 16) + yb;  b = clamp8((y1 + ub * (u - 128)) >> 6);  g = clamp8((y1 - (ug * (u - 128) + vg * (v - 128))) >> 6); ...
// How it is arranged for gfx950:
//   * work unit = one wave = 256 x (2 * NSW) pixels (NSW strips of two luma rows); a lane owns 4 consecutive pixels of every row:
//     one dword load per plane and row, one 16-byte non-temporal store per row (1 KiB contiguous per wave instruction).  The four
//     waves of a workgroup sit side by side (a 1024-pixel-wide tile) or stacked; they never wait for each other, except at ONE
//     workgroup barrier where they share something: the chroma halo rows of stacked 4:2:0 bilinear tiles (pkRunBlock), the LDS
//     transposition of fused quarter turns (pkTransposeStore).  Every load of the wave's tile is issued before the first result
//     is needed; occupancy (8 waves per SIMD) hides the rest;
//   * workgroups take tiles in per-XCD chunks of one tile row (planes resident in the Infinity Cache), or in raster order (batches
//     that stream from HBM: tests/tools/pattern_probe.hip, pkbench_wide.hip);
//   * bilinear chroma: the chroma neighbourhood (NSW + 2 rows x 136 columns per wave for 4:2:0) is staged in LDS
//     as one word per column holding both planes, (u | v << 16) * 16 + 0x08080808.  The filter 9:3:3:1 runs on both planes
//     at once with 32-bit shift-adds.  The constant carries libyuv's rounding (+8 per tap sum of 16) AND flips the top bit of
//     the result byte: after the sum, byte 1 / byte 3 of a word hold (u' - 128) / (v' - 128) as signed bytes -- what the matrix
//     wants.  Fields are allowed to wrap: the 32-bit sums are exact modulo 2^32, a carry out of the low field reaches only
//     the fraction bits of the high field (its low byte is a multiple of 16, +1 never reaches byte 3);
//   * matrix: pixels are processed as PAIRS, one 16-bit lane each.  v_perm_b32 with its sign-replicating selectors builds
//     (u0 - 128 | u1 - 128) and (v0 - 128 | v1 - 128) from two filtered words; luma is one 24-bit multiply per pixel whose upper
//     halves a v_perm_b32 pairs up; then X = mad(U, cX, Y) with saturation (y1 + 128 * 127 exceeds int16 for limited-range
//     blue), Z = mad(V, cZ, Y), G = mad(V, -gHi, mad(U, -gLo, Y)); >> 6 per pair; v_sat_pk_u8_i16 clamps a pair to bytes;
//     three v_perm_b32 interleave two pixels with their alpha.  9.5 VALU instructions per pixel for what took 18 as 32-bit
//     scalar code (tile_fx_impl.h), all full rate;
//   * libyuv's "YVU trick" (src/reformat_libyuv.c:386-423) is applied to the plane pointers (tile_shared.h): `u` feeds the
//     first colour byte X.
// Scope: 8-bit planes 4:4:4 / 4:2:2 / 4:2:0 / 4:0:0, nearest or bilinear, RGB / BGR / RGBA / BGRA / ARGB / ABGR, alpha opaque
// or copied from the plane.  10- and 12-bit planes (template parameter WIDE) enter the same arithmetic through one of libavif's two
// routes (src/reformat_libyuv.c:714-772, :906-930):
//   * WIDE_NATIVE: libyuv's I010 / I210 / I410 / I012 entries -- y32 = (y << 6) | (y >> 4), chroma filtered at the planes' depth and only
//     then cut to a byte (clamp255(c >> 2)): staged words keep 16-bit fields (u | v << 16) + rounding, the filtered fields are paired up
//     per plane and reduced with three packed instructions (shift, min, subtract 128);
//   * WIDE_DOWNSHIFT: no high-bit-depth entry exists, libavif reduces every sample to 8 bits first (clamp255(s >> (depth - 8))) and calls
//     the 8-bit entry: samples are narrowed right after the load, everything downstream is the 8-bit kernel.
// The attenuate / unattenuate post-pass stays with tile_fx_impl.h.
#pragma once

#include <type_traits>

#include "pixel_fixed.h"
#include "tile_impl.h"

namespace avifhip {
namespace tile {
inline namespace AVIFHIP_TILE_BUILD { // (tile_impl.h: the plain and the seam-aware build of a family)

constexpr int kPkPitch = 140;   // words per staged chroma row: entry c + 5 holds chroma column cxb + c, c in [-4, 131]
constexpr int kPkGroups = 17;   // 8-column groups per staged row
constexpr int kPkRowsPerRound = 3; // 3 x 17 = 51 of the wave's 64 lanes load 8 columns of both planes per round

enum PkWide { WIDE_NONE = 0, WIDE_NATIVE = 1, WIDE_DOWNSHIFT = 2 }; // 8-bit planes | 16-bit containers through libyuv's high-bit-depth entries | ... reduced to 8 bits first

// what a lane holds of a row: 4 samples (a dword of bytes, or two dwords of 16-bit samples) / of a staged chroma row: 8 columns
template <int WIDE>
struct PkTypes
{
    typedef typename std::conditional<WIDE != WIDE_NONE, u2, unsigned>::type Row4;
    typedef typename std::conditional<WIDE != WIDE_NONE, u4, u2>::type Col8;
    static constexpr uint32_t kBytes = (WIDE != WIDE_NONE) ? 2 : 1;
};

template <int SUB, int NSW>
struct PkStage
{
    static constexpr int kRows = (SUB == SUB_420) ? NSW + 2 : 2 * NSW; // 4:2:2: one chroma row per luma row
    static constexpr int kRounds = (kRows + kPkRowsPerRound - 1) / kPkRowsPerRound;
    static constexpr unsigned kScale = (SUB == SUB_420) ? 16u : 64u;    // tap sums of 16 (9+3+3+1) / 4 (3+1), to 256
    // per staged sample: rounding (weight sum / 2, scaled) + the top-bit flip (0x8000), divided by the weight sum
    static constexpr unsigned kBias = (SUB == SUB_420) ? 0x08080808u : 0x20202020u;
};

// ---- packed 16-bit instructions (two pixels per instruction).  Operands named `k` are wave-uniform (kernel arguments or
//      constants) and are read straight from scalar registers: one SGPR per VALU instruction is free ----
__device__ __forceinline__ unsigned pkMad(unsigned a, unsigned k, unsigned c)
{
    unsigned d;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
// saturating: the exact a * k + c held to [-32768, 32767] per half (tests/tools/probe_pk.hip)
__device__ __forceinline__ unsigned pkMadSat(unsigned a, unsigned k, unsigned c)
{
    unsigned d;
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pkAddK(unsigned a, unsigned k)
{
    unsigned d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
__device__ __forceinline__ unsigned pkSubK(unsigned a, unsigned k)
{
    unsigned d;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
// logical shifts of both halves by a wave-uniform amount (both halves of `k` hold it), unsigned minimum with a wave-uniform pair
__device__ __forceinline__ unsigned pkLshrK(unsigned a, unsigned k)
{
    unsigned d;
    asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(d) : "s"(k), "v"(a));
    return d;
}
__device__ __forceinline__ unsigned pkLshlK(unsigned a, unsigned k)
{
    unsigned d;
    asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "s"(k), "v"(a));
    return d;
}
__device__ __forceinline__ unsigned pkMinK(unsigned a, unsigned k)
{
    unsigned d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
// clamp255(s >> shift) - 128 on a pair of 16-bit samples: what libyuv's 16-bit readers do to chroma before the matrix
__device__ __forceinline__ unsigned pkChromaFromWide(unsigned pair, unsigned shiftSplat)
{
    return pkSubK(pkMinK(pkLshrK(pair, shiftSplat), 0x00ff00ffu), 0x00800080u);
}
// four 16-bit samples (two dwords) to four bytes clamp255(s >> shift)
__device__ __forceinline__ unsigned pkNarrow(u2 r, unsigned shiftSplat)
{
    const unsigned lo = pkMinK(pkLshrK(r.x, shiftSplat), 0x00ff00ffu), hi = pkMinK(pkLshrK(r.y, shiftSplat), 0x00ff00ffu);
    return __builtin_amdgcn_perm(hi, lo, 0x06040200u);
}
// arithmetic shift right of both halves by 6 (the inline constant's low half serves both)
__device__ __forceinline__ unsigned pkAshr6(unsigned a)
{
    unsigned d;
    asm("v_pk_ashrrev_i16 %0, 6, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a));
    return d;
}
// bits 0..15 = { clamp(lo, 0, 255), clamp(hi, 0, 255) }; bits 16..31 are not written (never read by the callers)
__device__ __forceinline__ unsigned satPkU8(unsigned a)
{
    unsigned d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
// 3 * a on a packed word, as the shift-and-add instruction (from (a << 1) + a the compiler builds a quarter-rate 32-bit multiply)
__device__ __forceinline__ unsigned pkTimes3(unsigned a)
{
    unsigned d;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ unsigned add3(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- chroma neighbourhood: loads of one staging round (8 columns of both planes per lane) ----
template <int SUB, int NSW, int WIDE, bool HALO>
__device__ __forceinline__ void pkStageLoad(const TileArgs & A, int cxb, int rowBase, int round, int slotMin, int slotMax, typename PkTypes<WIDE>::Col8 & uD,
                                            typename PkTypes<WIDE>::Col8 & vD)
{
    typedef PkStage<SUB, NSW> ST;
    typedef typename PkTypes<WIDE>::Col8 Col8;
    constexpr uint32_t B = PkTypes<WIDE>::kBytes;
    const int lane = threadIdx.x;
    const int rr = (lane * 241) >> 12; // lane / 17
    const int j = lane - rr * kPkGroups;
    const int i = round * kPkRowsPerRound + rr;
    uD = Col8 {};
    vD = Col8 {};
    if (rr < kPkRowsPerRound && i < ST::kRows && i >= slotMin && i <= slotMax) {
        // coordinates clamp to the job's chroma window (haloRow, tile_impl.h: the whole plane unless the canvas is a grid of separately
        // stored tiles; the seam-aware builds read one sample beyond it from the neighbouring tile)
        const HaloRow hr = haloRow<HALO>(A, rowBase + i);
        const int cxa = cxb - 4 + 8 * j;
        const uint32_t uRow = (uint32_t)hr.cy * A.uPitch, vRow = (uint32_t)hr.cy * A.vPitch;
        if (cxa >= A.cxMin && cxa + 7 <= A.cxMax) {
            uD = *reinterpret_cast<const Col8 *>(hr.u + (uRow + (uint32_t)cxa * B));
            vD = *reinterpret_cast<const Col8 *>(hr.v + (vRow + (uint32_t)cxa * B));
        } else {
            // group cut by the left or right border of the window
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint8_t *pu, *pv;
                const uint32_t cx = haloColumn<HALO>(A, hr, cxa + k, pu, pv);
                if constexpr (WIDE != WIDE_NONE) {
                    const unsigned u = *reinterpret_cast<const uint16_t *>(pu + (uRow + cx * 2u)), v = *reinterpret_cast<const uint16_t *>(pv + (vRow + cx * 2u));
                    uD[k >> 1] |= u << (16 * (k & 1));
                    vD[k >> 1] |= v << (16 * (k & 1));
                } else {
                    const unsigned u = pu[uRow + cx], v = pv[vRow + cx];
                    uD[k >> 2] |= u << (8 * (k & 3));
                    vD[k >> 2] |= v << (8 * (k & 3));
                }
            }
        }
    }
}

// ... and their conversion into staged words
// WIDE_NATIVE: fields keep the planes' depth (held to 12 bits so that no tap sum leaves its field: 16 * 4095 + 8 < 65536 -- libyuv's own
// 16-bit row functions are defined on that domain, ScaleRowUp2_Bilinear_12); WIDE_DOWNSHIFT: samples are cut to
// bytes (`shiftSplat` = depth - 8 in both halves) and staged like 8-bit ones.
template <int SUB, int NSW, int WIDE>
__device__ __forceinline__ void pkStageStore(int round, int slotMin, int slotMax, const typename PkTypes<WIDE>::Col8 & uD, const typename PkTypes<WIDE>::Col8 & vD,
                                             unsigned shiftSplat, unsigned * ring)
{
    typedef PkStage<SUB, NSW> ST;
    const int lane = threadIdx.x;
    const int rr = (lane * 241) >> 12;
    const int j = lane - rr * kPkGroups;
    const int i = round * kPkRowsPerRound + rr;
    if (rr < kPkRowsPerRound && i < ST::kRows && i >= slotMin && i <= slotMax) {
        unsigned w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // (u_k | v_k << 16): bytes 0..3 of the selector address the second operand, 4..7 the first, 12 is zero
            if constexpr (WIDE == WIDE_NONE) {
                const unsigned p = __builtin_amdgcn_perm(vD[k >> 2], uD[k >> 2], 0x0c040c00u + (unsigned)(k & 3) * 0x00010001u);
                w[k] = __umul24(p, ST::kScale) + ST::kBias;
            } else {
                const unsigned p = __builtin_amdgcn_perm(vD[k >> 1], uD[k >> 1], (k & 1) ? 0x07060302u : 0x05040100u); // (u_k | v_k << 16), 16-bit fields
                if constexpr (WIDE == WIDE_DOWNSHIFT)
                    w[k] = __umul24(pkMinK(pkLshrK(p, shiftSplat), 0x00ff00ffu), ST::kScale) + ST::kBias;
                else
                    w[k] = pkMinK(p, 0x0fff0fffu); // (the rounding joins in the filter: a per-sample bias would be multiplied by the weight sum)
            }
        }
        unsigned * row = ring + i * kPkPitch + 8 * j; // entries 8j + 1 .. 8j + 8
        row[1] = w[0];
        *reinterpret_cast<u2 *>(row + 2) = (u2) { w[1], w[2] };
        *reinterpret_cast<u2 *>(row + 4) = (u2) { w[3], w[4] };
        *reinterpret_cast<u2 *>(row + 6) = (u2) { w[5], w[6] };
        row[8] = w[7];
    }
}

__device__ __forceinline__ void pkReadRow(const unsigned * ring, int q, unsigned m[4])
{
    const u2 * src = reinterpret_cast<const u2 *>(ring + q * kPkPitch + 2 * (int)threadIdx.x + 4); // columns 2tx-1 .. 2tx+2
    const u2 lo = src[0], hi = src[1];
    m[0] = lo.x, m[1] = lo.y, m[2] = hi.x, m[3] = hi.y;
}

// ---- one luma row of a lane: 4 pixels from their luma dword, (U, V) pairs and alpha dword ----
// Up[p] / Vp[p]: (c0 - 128 | c1 - 128) as two int16 for pixel pair p.
// MAPPED: the four pixel words are handed back in `out` instead of being stored (the caller stores them through the PixelMap).
// luma of a lane's 4 pixels as two pairs (y1 of pixel 0 | y1 of pixel 1), y1 = ((y32 * yg) >> 16) + yb
template <int WIDE>
__device__ __forceinline__ void pkLuma(const TileArgs & A, typename PkTypes<WIDE>::Row4 raw, unsigned Y[2])
{
    const TileArgs::Fx & F = A.fx;
    if constexpr (WIDE == WIDE_NATIVE) {
        // y32 = (y << 6) | (y >> 4) for 10-bit planes ((y << 4) | (y >> 8) for 12): the sample's bits replicated down to fill 16
        const unsigned shl = F.yShl * 0x00010001u, shr = F.yShr * 0x00010001u;
        const unsigned s0 = pkLshlK(raw.x, shl) | pkLshrK(raw.x, shr), s1 = pkLshlK(raw.y, shl) | pkLshrK(raw.y, shr);
        const unsigned k0 = __umul24(s0 & 0xffffu, F.yMul), k1 = __umul24(s0 >> 16, F.yMul);
        const unsigned k2 = __umul24(s1 & 0xffffu, F.yMul), k3 = __umul24(s1 >> 16, F.yMul);
        Y[0] = pkAddK(__builtin_amdgcn_perm(k1, k0, 0x07060302u), F.pkYb);
        Y[1] = pkAddK(__builtin_amdgcn_perm(k3, k2, 0x07060302u), F.pkYb);
    } else {
        unsigned yraw;
        if constexpr (WIDE == WIDE_DOWNSHIFT)
            yraw = pkNarrow(raw, F.downshift * 0x00010001u);
        else
            yraw = raw;
        const unsigned k0 = __umul24(yraw & 0xffu, F.yMul8), k1 = __umul24((yraw >> 8) & 0xffu, F.yMul8);
        const unsigned k2 = __umul24((yraw >> 16) & 0xffu, F.yMul8), k3 = __umul24(yraw >> 24, F.yMul8);
        Y[0] = pkAddK(__builtin_amdgcn_perm(k1, k0, 0x07060302u), F.pkYb);
        Y[1] = pkAddK(__builtin_amdgcn_perm(k3, k2, 0x07060302u), F.pkYb);
    }
}

// alpha of a lane's 4 pixels as four bytes
template <int WIDE>
__device__ __forceinline__ unsigned pkAlpha(const TileArgs & A, typename PkTypes<WIDE>::Row4 raw)
{
    if constexpr (WIDE == WIDE_NONE) {
        if (A.alphaLim.on) { // limited-range alpha plane (wave-uniform): the four bytes to full range
            const unsigned a0 = alphaToFullRange(A, raw & 0xffu), a1 = alphaToFullRange(A, (raw >> 8) & 0xffu);
            const unsigned a2 = alphaToFullRange(A, (raw >> 16) & 0xffu), a3 = alphaToFullRange(A, raw >> 24);
            return a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        }
        return raw;
    } else {
        const TileArgs::Fx & F = A.fx;
        if (!A.alphaLim.on && F.alphaMode == FXA_SHIFT) // libyuv's alpha twins: clamp255(a >> (depth - 8)), both wave-uniform
            return pkNarrow(raw, F.alphaShift * 0x00010001u);
        unsigned av[4] = { raw.x & 0xffffu, raw.x >> 16, raw.y & 0xffffu, raw.y >> 16 };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (A.alphaLim.on)
                av[i] = alphaToFullRange(A, av[i]);
            // libyuv's 255, then avifReformatAlpha over it (src/reformat.c:1464-1486): a shift or the fp32 rescale
            av[i] = (F.alphaMode == FXA_SHIFT) ? minU(av[i] >> F.alphaShift, 255u) : alphaFromPlane(A, av[i]);
        }
        return av[0] | (av[1] << 8) | (av[2] << 16) | (av[3] << 24);
    }
}

// `xchg` / `segBytes`: 3-byte pixels only -- the wave's exchange buffer and the bytes of its row segment that exist (storeRowContiguous)
// ATT: libyuv's ARGBAttenuate on the three colour bytes, (c * a + 255) >> 8 (appendix D.4; what libavif runs after the conversion when the
// pixels are to be premultiplied, src/reformat.c:1574-1585 -> src/alpha.c:163) -- on pixel pairs, before the bytes are packed.
// ATT == 2: libyuv's ARGBUnattenuate (appendix D.4; src/reformat_libyuv.c:1138-1161 -- what libavif runs after the conversion when a premultiplied
// image is wanted as straight-alpha pixels): t = ((c * 0x101) * ia) >> 16 with ia the fixed-point reciprocal of the pixel's alpha from the
// workgroup's LDS table `recip` (pkBuildReciprocals: 0, 0xffff, 0x10000 / a, 0x100 for a = 0, 1, 2..254, 255), then libyuv's signed saturating
// pack: t >= 0x8000 becomes 0 (the a == 1, c >= 128 artefact), everything else min(t, 255).  Two 16 x 16-bit products per channel and pixel pair
// (v_mul_u32_u24: both factors are below 2^16), their upper halves gathered into one register and clamped by v_sat_pk_u8_i16, whose signed
// reading of the halves IS the artefact.
template <int SUB, int NCH, bool APLANE, bool MAPPED, int ATT = 0>
__device__ __forceinline__ void pkRow(const TileArgs & A, const unsigned Y[2], unsigned araw, const unsigned Up[2], const unsigned Vp[2], uint32_t off, bool laneValid,
                                      unsigned out[4], WideRowExchange * xchg, uint32_t segBytes, const unsigned * recip = nullptr)
{
    const TileArgs::Fx & F = A.fx;
    unsigned px[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned X, G, Z;
#ifdef AVIFHIP_ABLATE_MATRIX // measurement only (tests/tools/pkbench.hip): no matrix, no clamps
        if constexpr (true) {
            X = Up[p], G = Vp[p], Z = Y[p];
        } else
#endif
        if constexpr (SUB == SUB_400) {
            X = G = Z = satPkU8(pkAshr6(Y[p]));
        } else {
            X = pkMadSat(Up[p], F.pkCX, Y[p]);
            Z = pkMadSat(Vp[p], F.pkCZ, Y[p]);
            G = pkMad(Vp[p], F.pkGHi, pkMad(Up[p], F.pkGLo, Y[p]));
            X = satPkU8(pkAshr6(X));
            G = satPkU8(pkAshr6(G));
            Z = satPkU8(pkAshr6(Z));
        }
        if constexpr (ATT == 2) {
            const unsigned i0 = recip[(araw >> (16 * p)) & 0xffu], i1 = recip[(araw >> (16 * p + 8)) & 0xffu];
            auto unatt = [&](unsigned c) { // c = (c0 c1 . .)
                const unsigned p0 = __umul24(__builtin_amdgcn_perm(0u, c, 0x0c0c0000u), i0); // c0 * 0x101 = (c0 c0 . .)
                const unsigned p1 = __umul24(__builtin_amdgcn_perm(0u, c, 0x0c0c0101u), i1);
                return satPkU8(__builtin_amdgcn_perm(p1, p0, 0x07060302u)); // (t0 | t1 << 16) -> (clamp(t0) clamp(t1) . .)
            };
            X = unatt(X), G = unatt(G), Z = unatt(Z);
        } else if constexpr (ATT == 1) {
            // colour bytes (c0 c1 . .) and alpha bytes spread to 16-bit pairs, (c * a + 255) >> 8 per half (65280 at most), bytes gathered again
            const unsigned Ap = __builtin_amdgcn_perm(0u, araw, p ? 0x0c030c02u : 0x0c010c00u);
            auto att = [&](unsigned c) {
                unsigned d;
                const unsigned c2 = __builtin_amdgcn_perm(0u, c, 0x0c010c00u);
                asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(c2), "v"(Ap), "s"(0x00ff00ffu));
                return __builtin_amdgcn_perm(0u, d, 0x0c0c0301u); // byte 1 of each half
            };
            X = att(X), G = att(G), Z = att(Z);
        }
        if constexpr (NCH == 2) {
            // RGB565 (I420ToRGB565Matrix / I422ToRGB565Matrix: b >> 3 | (g >> 2) << 5 | (r >> 3) << 11, src/reformat.c:619): both pixels of
            // the pair at once, one per 16-bit half (masks first, so plain 32-bit shifts cannot leak across the halves)
            const unsigned xh = __builtin_amdgcn_perm(0u, X, 0x0c010c00u) & 0x00f800f8u; // x0 . x1 .
            const unsigned gh = __builtin_amdgcn_perm(0u, G, 0x0c010c00u) & 0x00fc00fcu;
            const unsigned zh = __builtin_amdgcn_perm(0u, Z, 0x0c010c00u) & 0x00f800f8u;
            px[p] = (zh << 8) | ((gh << 3) | (xh >> 3));
            continue;
        }
        const unsigned XG = __builtin_amdgcn_perm(G, X, 0x05010400u); // x0 g0 x1 g1
        unsigned ZA = Z;                                              // z0 z1 . .
        if constexpr (APLANE)
            ZA = __builtin_amdgcn_perm(araw, Z, p ? 0x07060100u : 0x05040100u); // z0 z1 a0 a1
        px[2 * p] = __builtin_amdgcn_perm(ZA, XG, F.pkSel0);
        px[2 * p + 1] = __builtin_amdgcn_perm(ZA, XG, F.pkSel1);
    }
    if constexpr (MAPPED) {
        out[0] = px[0], out[1] = px[1], out[2] = px[2], out[3] = px[3];
        return;
    }
    if constexpr (NCH == 3) {
        // pixels are (x g z .): 12 bytes x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3, handed to 16-byte stores through the wave's LDS buffer
        const unsigned w[3] = { __builtin_amdgcn_perm(px[1], px[0], 0x04020100u), __builtin_amdgcn_perm(px[2], px[1], 0x05040201u),
                                __builtin_amdgcn_perm(px[3], px[2], 0x06050402u) };
        storeRowContiguous<3>(A.rgb, off - 12u * (uint32_t)threadIdx.x, w, segBytes, *xchg);
        return;
    }
    if (!laneValid)
        return;
    if constexpr (NCH == 2) {
        storeVec(A.rgb, off, (u2a2) { px[0], px[1] }, true); // four 16-bit pixels (2-byte aligned rows)
    } else if constexpr (NCH == 4) {
        storeVec(A.rgb, off, (u4) { px[0], px[1], px[2], px[3] }, true);
    }
}

// (c0 - 128 | c1 - 128) from two filtered words (byte 1 = u - 128, byte 3 = v - 128 as signed bytes)
__device__ __forceinline__ void pkPairsFromWords(const unsigned w[4], unsigned Up[2], unsigned Vp[2])
{
    // selectors 8..11 replicate the sign of byte 1 / 3 of the second / first operand
    Up[0] = __builtin_amdgcn_perm(w[1], w[0], 0x0a050801u);
    Vp[0] = __builtin_amdgcn_perm(w[1], w[0], 0x0b070903u);
    Up[1] = __builtin_amdgcn_perm(w[3], w[2], 0x0a050801u);
    Vp[1] = __builtin_amdgcn_perm(w[3], w[2], 0x0b070903u);
}

// ... from two filtered words with 16-bit fields (u | v << 16) at the planes' depth times the filter's weight sum
__device__ __forceinline__ void pkPairsFromWideWords(const unsigned w[4], unsigned shiftSplat, unsigned Up[2], unsigned Vp[2])
{
    Up[0] = pkChromaFromWide(__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), shiftSplat);
    Vp[0] = pkChromaFromWide(__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), shiftSplat);
    Up[1] = pkChromaFromWide(__builtin_amdgcn_perm(w[3], w[2], 0x05040100u), shiftSplat);
    Vp[1] = pkChromaFromWide(__builtin_amdgcn_perm(w[3], w[2], 0x07060302u), shiftSplat);
}
template <int WIDE>
__device__ __forceinline__ void pkPairs(const unsigned w[4], unsigned shiftSplat, unsigned Up[2], unsigned Vp[2])
{
    if constexpr (WIDE == WIDE_NATIVE)
        pkPairsFromWideWords(w, shiftSplat, Up, Vp);
    else
        pkPairsFromWords(w, Up, Vp);
}

// Raw (undecoded) data of one wave tile (256 x 2*NSW pixels) as loaded by one lane; lives in registers while the previous tile
// is computed.
template <int SUB, bool BIL, bool APLANE, int NSW, int WIDE>
struct PkRaw
{
    static constexpr bool kStaged = BIL && (SUB == SUB_420 || SUB == SUB_422);
    static constexpr bool kOwnChroma = !kStaged && SUB != SUB_400;
    typedef typename PkTypes<WIDE>::Row4 Row4;
    typedef typename PkTypes<WIDE>::Col8 Col8;
    typedef typename std::conditional<SUB == SUB_444, Row4, unsigned>::type Own; // 4:4:4: four samples per row | nearest: the lane's two samples
    Col8 uD[kStaged ? PkStage<SUB, NSW>::kRounds : 1], vD[kStaged ? PkStage<SUB, NSW>::kRounds : 1]; // this lane's share of the neighbourhood
    Row4 y[2 * NSW], a[APLANE ? 2 * NSW : 1];
    Own u[kOwnChroma ? 2 * NSW : 1], v[kOwnChroma ? 2 * NSW : 1];
};

// where a wave works: band (256-pixel column) and first strip (pair of luma rows) of its tile
struct PkSpot
{
    uint32_t band, strip0;
    // staged chroma rows (slots 0 .. kRows-1 of the wave's neighbourhood) this wave loads and stages itself; the others are staged by
    // the waves above and below it in the workgroup (pkRunBlock: shared halo rows)
    int slotMin, slotMax;
};

// ---- every load of a wave tile ----
template <int SUB, bool BIL, bool APLANE, int NSW, int WIDE, bool STREAM>
__device__ __forceinline__ void pkLoad(const TileArgs & A, const PkSpot & w, PkRaw<SUB, BIL, APLANE, NSW, WIDE> & R)
{
    typedef PkRaw<SUB, BIL, APLANE, NSW, WIDE> RawT;
    typedef PkStage<SUB, NSW> ST;
    typedef typename RawT::Row4 Row4;
    constexpr uint32_t B = PkTypes<WIDE>::kBytes;
    const uint32_t bandX = w.band * (uint32_t)kBandW;
    const uint32_t X = bandX + 4u * (uint32_t)threadIdx.x;
    const uint32_t Xc = X < A.w4 ? X : 0u; // absent lanes load (and discard) the row's first group
    const uint32_t strips = A.h2 >> 1;
    auto stagedLoads = [&]() {
#ifdef AVIFHIP_ABLATE_STAGE
        if constexpr (false) {
#else
        if constexpr (RawT::kStaged) {
#endif
            const int cxb = A.cx0 + (int)(bandX >> 1);
            const int rowBase = (SUB == SUB_420) ? A.cy0 + (int)w.strip0 - 1 : A.cy0 + 2 * (int)w.strip0;
            if (haloNeeded(A, rowBase, rowBase + ST::kRows - 1, cxb - 1, cxb + 128)) { // (wave-uniform; never in the plain builds)
#pragma unroll
                for (int t = 0; t < ST::kRounds; ++t)
                    pkStageLoad<SUB, NSW, WIDE, true>(A, cxb, rowBase, t, w.slotMin, w.slotMax, R.uD[t], R.vD[t]);
            } else {
#pragma unroll
                for (int t = 0; t < ST::kRounds; ++t)
                    pkStageLoad<SUB, NSW, WIDE, false>(A, cxb, rowBase, t, w.slotMin, w.slotMax, R.uD[t], R.vD[t]);
            }
        }
    };
    // Which loads go first (tests/tools/pkbench.hip / pkbench_wide.hip with -DAVIFHIP_LUMA_FIRST / -DAVIFHIP_CHROMA_FIRST): 8-bit planes run
    // 2 % faster with the luma (and alpha) rows ahead of the neighbourhood (8K: 28.8 -> 28.1 us; 12 frames cycled 36.2 -> 35.6), 16-bit
    // containers 2 % faster the other way round (64 tiles of 1080p: 170.7 vs 174.1 us)
#if defined(AVIFHIP_LUMA_FIRST)
    constexpr bool kLumaFirst = true;
#elif defined(AVIFHIP_CHROMA_FIRST)
    constexpr bool kLumaFirst = false;
#else
    constexpr bool kLumaFirst = WIDE == WIDE_NONE;
#endif
    if constexpr (!kLumaFirst)
        stagedLoads();
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const uint32_t st = w.strip0 + (uint32_t)s;
        const uint32_t sy = 2u * (st < strips ? st : strips - 1u); // absent strips load (and discard) the last one
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // Planes too large to stay in the Infinity Cache (batches of tiles: TileLaunch::streamLoads) are read with streaming loads --
            // 64 tiles of 1080p 10-bit: 173.8 -> 160.7 us; planes that ARE cache-resident must not be (an 8K frame cycled with three
            // others: 28.3 -> 35.8 us): tests/tools/pkbench_wide.hip with -DPKB_STREAM=true.  A template parameter, not a
            // run-time branch: the compiler merges the two branches' loads and the merged load loses its non-temporal mark.
            if constexpr (STREAM) {
                R.y[2 * s + r] = __builtin_nontemporal_load(reinterpret_cast<const Row4 *>(A.y + ((sy + r) * A.yPitch + Xc * B)));
                if constexpr (APLANE)
                    R.a[2 * s + r] = __builtin_nontemporal_load(reinterpret_cast<const Row4 *>(A.a + ((sy + r) * A.aPitch + Xc * B)));
            } else {
                R.y[2 * s + r] = *reinterpret_cast<const Row4 *>(A.y + ((sy + r) * A.yPitch + Xc * B));
                if constexpr (APLANE)
                    R.a[2 * s + r] = *reinterpret_cast<const Row4 *>(A.a + ((sy + r) * A.aPitch + Xc * B));
            }
            if constexpr (SUB == SUB_444) {
                const Row4 * up = reinterpret_cast<const Row4 *>(A.u + (((uint32_t)A.cy0 + sy + r) * A.uPitch + ((uint32_t)A.cx0 + Xc) * B));
                const Row4 * vp = reinterpret_cast<const Row4 *>(A.v + (((uint32_t)A.cy0 + sy + r) * A.vPitch + ((uint32_t)A.cx0 + Xc) * B));
                if constexpr (STREAM) {
                    R.u[2 * s + r] = __builtin_nontemporal_load(up), R.v[2 * s + r] = __builtin_nontemporal_load(vp);
                } else {
                    R.u[2 * s + r] = *up, R.v[2 * s + r] = *vp;
                }
            } else if constexpr (RawT::kOwnChroma) {
                // nearest: chroma samples (X >> 1, X >> 1 + 1) of the row's chroma row, one aligned pair per plane
                if (!(SUB == SUB_420 && r == 1)) {
                    const uint32_t cy = (uint32_t)A.cy0 + ((SUB == SUB_420) ? (sy >> 1) : (sy + r));
                    const uint32_t cx = (uint32_t)A.cx0 + (Xc >> 1);
                    if constexpr (WIDE != WIDE_NONE) {
                        R.u[2 * s + r] = *reinterpret_cast<const uint32_t *>(A.u + (cy * A.uPitch + cx * 2u));
                        R.v[2 * s + r] = *reinterpret_cast<const uint32_t *>(A.v + (cy * A.vPitch + cx * 2u));
                    } else {
                        R.u[2 * s + r] = *reinterpret_cast<const uint16_t *>(A.u + (cy * A.uPitch + cx));
                        R.v[2 * s + r] = *reinterpret_cast<const uint16_t *>(A.v + (cy * A.vPitch + cx));
                    }
                } else {
                    R.u[2 * s + r] = R.v[2 * s + r] = 0;
                }
            }
        }
    }
    if constexpr (kLumaFirst)
        stagedLoads();
}

// ---- the chroma neighbourhood into the wave's LDS block.  Wave-private: the LDS instructions of one wave execute in order;
//      the fences keep the compiler from moving the reads above the writes ----
template <int SUB, bool BIL, bool APLANE, int NSW, int WIDE>
__device__ __forceinline__ void pkStage(const TileArgs & A, const PkSpot & w, bool shared, const PkRaw<SUB, BIL, APLANE, NSW, WIDE> & R, unsigned * ring)
{
#ifdef AVIFHIP_ABLATE_STAGE // measurement only (tests/tools/pkbench_wide.hip): nothing is staged, the filter reads whatever the LDS holds
    if constexpr (false) {
#else
    if constexpr (PkRaw<SUB, BIL, APLANE, NSW, WIDE>::kStaged) {
#endif
        const unsigned shiftSplat = A.fx.downshift * 0x00010001u;
#pragma unroll
        for (int t = 0; t < PkStage<SUB, NSW>::kRounds; ++t)
            pkStageStore<SUB, NSW, WIDE>(t, w.slotMin, w.slotMax, R.uD[t], R.vD[t], shiftSplat, ring);
        if (shared) { // workgroup-uniform: the stacked waves read each other's boundary rows -- the kernel's one workgroup barrier
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ---- stores through a PixelMap (fused crop / rotate / mirror).  A pixel word holds (x g z a) or (x g z .) ----
// (u4a4, tile_impl.h: 16-byte accesses at dword alignment -- crops start anywhere)

template <int NCH>
__device__ __forceinline__ void pkStorePixel(uint8_t * dst, unsigned px)
{
    if constexpr (NCH == 4) {
        *reinterpret_cast<unsigned *>(dst) = px;
    } else {
        dst[0] = (uint8_t)px, dst[1] = (uint8_t)(px >> 8), dst[2] = (uint8_t)(px >> 16);
    }
}

// rows stay rows: the lane's four pixels of canvas row j, first at canvas column i
template <int NCH>
__device__ __forceinline__ void pkStoreMappedRow(const TileArgs & A, const unsigned px[4], uint32_t i, uint32_t j)
{
    const PixelMap & m = A.map;
    const uint32_t ii = i - m.cx, jj = j - m.cy;
    if (jj >= m.ch)
        return;
    uint8_t * row = A.rgb + (size_t)(uint32_t)(m.sy * (int32_t)jj + m.ky) * A.rgbPitch;
    if (NCH == 4 && ii < m.cw && m.cw - ii >= 4u) { // all four inside the crop: one 16-byte store, forwards or mirrored
        const bool fwd = m.sx > 0;
        const uint32_t x = (uint32_t)(fwd ? (int32_t)ii + m.kx : m.kx - (int32_t)(ii + 3u));
        const u4 v = fwd ? (u4) { px[0], px[1], px[2], px[3] } : (u4) { px[3], px[2], px[1], px[0] };
        if (A.tuning & TUNE_NONTEMPORAL)
            __builtin_nontemporal_store(v, reinterpret_cast<u4a4 *>(row + (size_t)x * 4u));
        else
            *reinterpret_cast<u4a4 *>(row + (size_t)x * 4u) = v;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t iik = ii + (uint32_t)k;
        if (iik < m.cw)
            pkStorePixel<NCH>(row + (size_t)(uint32_t)(m.sx * (int32_t)iik + m.kx) * NCH, px[k]);
    }
}

// rows become columns (quarter turns): the lane's 4 columns x ROWS rows, first at canvas (i, j); `rowsValid` rows exist.  Each
// source column is a run of ROWS consecutive destination pixels; plain stores: the pieces of one destination line come from
// different waves and meet in L2 (tests/tools/transpose_probe.hip: 64 us for an 8K frame against 383 us non-temporal)
template <int NCH, int ROWS>
__device__ __forceinline__ void pkStoreMappedColumns(const TileArgs & A, const unsigned px[ROWS][4], uint32_t i, uint32_t j, uint32_t rowsValid)
{
    const PixelMap & m = A.map;
    const uint32_t ii = i - m.cx, jj = j - m.cy;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t iic = ii + (uint32_t)c;
        if (iic >= m.cw)
            continue;
        uint8_t * row = A.rgb + (size_t)(uint32_t)(m.sy * (int32_t)iic + m.ky) * A.rgbPitch;
#pragma unroll
        for (int q = 0; q < ROWS / 4; ++q) {
            const uint32_t jq = jj + 4u * (uint32_t)q;
            if (NCH == 4 && 4u * (uint32_t)q + 4u <= rowsValid && jq < m.ch && m.ch - jq >= 4u) {
                const bool fwd = m.sx > 0;
                const uint32_t x = (uint32_t)(fwd ? (int32_t)jq + m.kx : m.kx - (int32_t)(jq + 3u));
                const u4 v = fwd ? (u4) { px[4 * q][c], px[4 * q + 1][c], px[4 * q + 2][c], px[4 * q + 3][c] }
                                 : (u4) { px[4 * q + 3][c], px[4 * q + 2][c], px[4 * q + 1][c], px[4 * q][c] };
                *reinterpret_cast<u4a4 *>(row + (size_t)x * 4u) = v;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t jr = jq + (uint32_t)r;
                if (4u * (uint32_t)q + (uint32_t)r < rowsValid && jr < m.ch)
                    pkStorePixel<NCH>(row + (size_t)(uint32_t)(m.sx * (int32_t)jr + m.kx) * NCH, px[4 * q + r][c]);
            }
        }
    }
}

// Quarter turns, the four waves stacked: the workgroup's tile (256 columns x ROWS = 8 * NSW rows) is transposed through LDS, so that a
// source COLUMN leaves as one run of ROWS consecutive destination pixels (128 bytes for 32 rows) and a store instruction writes 8 such
// runs, 16 bytes per lane -- whole cache lines instead of the 32-byte pieces of pkStoreMappedColumns.  Layout: word (y, x) of the tile
// at y * 256 + (x ^ 4 * ((y >> 2) & 7)): rows are written with aligned 16-byte LDS stores (the swizzle moves whole 4-word groups) and
// read column-wise with two lanes per bank.  `tile` is shared by the workgroup; the caller's barrier separates it from the chroma
// neighbourhoods it overlays.
template <int NCH, int NSW>
__device__ __forceinline__ void pkTransposeWrite(unsigned * tile, uint32_t wy, const unsigned (*held)[4])
{
    const uint32_t l = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 2 * NSW; ++r) {
        const uint32_t y = wy * (uint32_t)(2 * NSW) + (uint32_t)r;
        *reinterpret_cast<u4 *>(tile + y * 256u + 4u * (l ^ ((y >> 2) & 7u))) = (u4) { held[r][0], held[r][1], held[r][2], held[r][3] };
    }
}

// (the way out is tile_map_impl.h's mapTransposeStore: the same layout, everything invariant hoisted out of its loop)
template <int NCH, int NSW>
__device__ __forceinline__ void pkTransposeStore(const TileArgs & A, const unsigned * tile, uint32_t wv, uint32_t bandX0, uint32_t row0)
{
    mapTransposeStore<1, NCH, 8u * (uint32_t)NSW>(A, tile, wv, bandX0, row0); // (32 rows, or 16 for small jobs)
}

// ---- filter, matrix, stores of a wave tile ----
template <int SUB, bool BIL, int NCH, bool APLANE, int NSW, bool MAPPED, int WIDE, int ATT = 0>
__device__ __forceinline__ void pkCompute(const TileArgs & A, const PkSpot & w, const PkRaw<SUB, BIL, APLANE, NSW, WIDE> & R, const unsigned * ring, WideRowExchange * xchg,
                                          unsigned * xposeTile, uint32_t wy, const unsigned * recip = nullptr)
{
    constexpr bool kStaged = PkRaw<SUB, BIL, APLANE, NSW, WIDE>::kStaged;
    // 16-bit containers, wave-uniform shift pairs: filtered fields (weight sum 16 or 4) / plain samples down to a byte
    const unsigned shFiltered = (((SUB == SUB_420) ? 4u : 2u) + A.fx.cShr) * 0x00010001u;
    const unsigned shSample = (A.fx.downshift + A.fx.cShr) * 0x00010001u;
    unsigned held[MAPPED ? 2 * NSW : 1][4]; // quarter turns: every row of the tile, stored column-wise at the end
    const uint32_t X = w.band * (uint32_t)kBandW + 4u * (uint32_t)threadIdx.x;
    const bool laneValid = X < A.w4;
    const uint32_t strips = A.h2 >> 1;
    const uint32_t bandX0 = w.band * (uint32_t)kBandW;
    const uint32_t segBytes = ((A.w4 - bandX0 < (uint32_t)kBandW) ? A.w4 - bandX0 : (uint32_t)kBandW) * (uint32_t)NCH;
    unsigned mA[4] = { 0, 0, 0, 0 }, mB[4] = { 0, 0, 0, 0 }, tA1 = 0, tA2 = 0, tB1 = 0, tB2 = 0;
    if constexpr (kStaged && SUB == SUB_420) {
        pkReadRow(ring, 0, mA);
        pkReadRow(ring, 1, mB);
        tA1 = pkTimes3(mA[1]), tA2 = pkTimes3(mA[2]);
        tB1 = pkTimes3(mB[1]), tB2 = pkTimes3(mB[2]);
    }
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const uint32_t st = w.strip0 + (uint32_t)s;
        const bool stripValid = st < strips; // wave-uniform
        const uint32_t sy = 2u * st;
        unsigned Up[2][2], Vp[2][2]; // [luma row of the strip][pixel pair]
        if constexpr (SUB == SUB_400) {
            Up[0][0] = Up[0][1] = Up[1][0] = Up[1][1] = 0, Vp[0][0] = Vp[0][1] = Vp[1][0] = Vp[1][1] = 0;
        } else if constexpr (kStaged && SUB == SUB_420) {
            // Scale2RowUp_Bilinear: (9 near + 3 horizontal + 3 vertical + 1 diagonal + 8) >> 4; rows: A above, B co-sited, C below
            unsigned mC[4];
            pkReadRow(ring, s + 2, mC);
            const unsigned tC1 = pkTimes3(mC[1]), tC2 = pkTimes3(mC[2]);
#ifdef AVIFHIP_ABLATE_FILTER // measurement only: no filter arithmetic
            pkPairsFromWords(mB, Up[0], Vp[0]);
            pkPairsFromWords(mC, Up[1], Vp[1]);
#else
            unsigned p1 = pkTimes3(tB1), p2 = pkTimes3(tB2); // 9 x
            if constexpr (WIDE == WIDE_NATIVE)
                p1 += 0x00080008u, p2 += 0x00080008u; // the filter's "+ 8" (8-bit fields carry it in their staging bias)
            const unsigned a0 = pkTimes3(mB[0]) + p1, a1 = p1 + tB2, a2 = p2 + tB1, a3 = pkTimes3(mB[3]) + p2;
            const unsigned we[4] = { add3(a0, tA1, mA[0]), add3(a1, tA1, mA[2]), add3(a2, tA2, mA[1]), add3(a3, tA2, mA[3]) }; // even row leans up
            const unsigned wo[4] = { add3(a0, tC1, mC[0]), add3(a1, tC1, mC[2]), add3(a2, tC2, mC[1]), add3(a3, tC2, mC[3]) }; // odd row down
            pkPairs<WIDE>(we, shFiltered, Up[0], Vp[0]);
            pkPairs<WIDE>(wo, shFiltered, Up[1], Vp[1]);
#endif
#pragma unroll
            for (int k = 0; k < 4; ++k)
                mA[k] = mB[k], mB[k] = mC[k];
            tA1 = tB1, tA2 = tB2, tB1 = tC1, tB2 = tC2;
        } else if constexpr (kStaged) { // 4:2:2, ScaleRowUp2_Linear: (3 near + far + 2) >> 2
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned m[4];
                pkReadRow(ring, 2 * s + r, m);
                unsigned t1 = pkTimes3(m[1]), t2 = pkTimes3(m[2]);
                if constexpr (WIDE == WIDE_NATIVE)
                    t1 += 0x00020002u, t2 += 0x00020002u;
                const unsigned wd[4] = { t1 + m[0], t1 + m[2], t2 + m[1], t2 + m[3] };
                pkPairs<WIDE>(wd, shFiltered, Up[r], Vp[r]);
            }
        } else if constexpr (SUB == SUB_444) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if constexpr (WIDE != WIDE_NONE) { // a dword is a pair already
                    Up[r][0] = pkChromaFromWide(R.u[2 * s + r].x, shSample), Up[r][1] = pkChromaFromWide(R.u[2 * s + r].y, shSample);
                    Vp[r][0] = pkChromaFromWide(R.v[2 * s + r].x, shSample), Vp[r][1] = pkChromaFromWide(R.v[2 * s + r].y, shSample);
                } else {
                    const unsigned u = R.u[2 * s + r], v = R.v[2 * s + r];
                    Up[r][0] = pkSubK(__builtin_amdgcn_perm(0u, u, 0x0c010c00u), 0x00800080u);
                    Up[r][1] = pkSubK(__builtin_amdgcn_perm(0u, u, 0x0c030c02u), 0x00800080u);
                    Vp[r][0] = pkSubK(__builtin_amdgcn_perm(0u, v, 0x0c010c00u), 0x00800080u);
                    Vp[r][1] = pkSubK(__builtin_amdgcn_perm(0u, v, 0x0c030c02u), 0x00800080u);
                }
            }
        } else { // nearest 4:2:2 / 4:2:0: both pixels of a pair share their chroma sample
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (SUB == SUB_420 && r == 1) {
                    Up[1][0] = Up[0][0], Up[1][1] = Up[0][1], Vp[1][0] = Vp[0][0], Vp[1][1] = Vp[0][1];
                    break;
                }
                if constexpr (WIDE != WIDE_NONE) { // (c0 | c1) reduced once, then each half doubled
                    const unsigned u = pkChromaFromWide(R.u[2 * s + r], shSample), v = pkChromaFromWide(R.v[2 * s + r], shSample);
                    Up[r][0] = __builtin_amdgcn_perm(0u, u, 0x01000100u), Up[r][1] = __builtin_amdgcn_perm(0u, u, 0x03020302u);
                    Vp[r][0] = __builtin_amdgcn_perm(0u, v, 0x01000100u), Vp[r][1] = __builtin_amdgcn_perm(0u, v, 0x03020302u);
                } else {
                    const unsigned u = R.u[2 * s + r], v = R.v[2 * s + r];
                    Up[r][0] = pkSubK(__builtin_amdgcn_perm(0u, u, 0x0c000c00u), 0x00800080u);
                    Up[r][1] = pkSubK(__builtin_amdgcn_perm(0u, u, 0x0c010c01u), 0x00800080u);
                    Vp[r][0] = pkSubK(__builtin_amdgcn_perm(0u, v, 0x0c000c00u), 0x00800080u);
                    Vp[r][1] = pkSubK(__builtin_amdgcn_perm(0u, v, 0x0c010c01u), 0x00800080u);
                }
            }
        }
        if (stripValid) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned px[4], Y[2], araw = 0;
                pkLuma<WIDE>(A, R.y[2 * s + r], Y);
                if constexpr (APLANE)
                    araw = pkAlpha<WIDE>(A, R.a[2 * s + r]);
                pkRow<SUB, NCH, APLANE, MAPPED, ATT>(A, Y, araw, Up[r], Vp[r], (sy + r) * A.rgbPitch + X * (uint32_t)NCH, laneValid, px, xchg, segBytes, recip);
                if constexpr (MAPPED) {
                    if (A.map.transposed) { // wave-uniform
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            held[2 * s + r][k] = px[k];
                    } else if (laneValid) {
                        pkStoreMappedRow<NCH>(A, px, (uint32_t)A.mapX0 + X, (uint32_t)A.mapY0 + sy + (uint32_t)r);
                    }
                }
            }
        } else if constexpr (MAPPED) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    held[2 * s + r][k] = 0;
        }
    }
    if constexpr (MAPPED) {
        if (A.map.transposed && xposeTile) { // workgroup-uniform: every wave of the workgroup is here, with or without rows of its own
            __syncthreads(); // the tile overlays the chroma neighbourhoods: every wave is done reading them
            pkTransposeWrite<NCH, NSW>(xposeTile, wy, held);
            __syncthreads();
            pkTransposeStore<NCH, NSW>(A, xposeTile, wy, bandX0, 2u * (w.strip0 - wy * (uint32_t)NSW));
        } else if (A.map.transposed && laneValid && w.strip0 < strips) {
            const uint32_t left = strips - w.strip0; // strips of the tile that exist (at least one)
            pkStoreMappedColumns<NCH, 2 * NSW>(A, held, (uint32_t)A.mapX0 + X, (uint32_t)A.mapY0 + 2u * w.strip0, 2u * (left < (uint32_t)NSW ? left : (uint32_t)NSW));
        }
    }
}

// The waves of a workgroup work for themselves (one barrier at most, see below): one wave, one tile of 256 x 2*NSW pixels, every load issued up
// front.  One tile per wave and many short-lived workgroups is deliberate: a persistent variant (k x 256 workgroups walking over
// their tiles with the next tile's loads in flight) measured 10-20% SLOWER on 8K frames, with frames streaming from HBM as well
// as from the Infinity Cache (tests/tools/pk_sweep.py, profiles/r02_pk_sweep_persistent.txt) -- the dispatcher refilling 32 waves per CU in
// tile order keeps the memory pipes fuller than a software pipeline one tile deep does.
// LDS of a workgroup: the waves' chroma blocks (16-byte aligned in total), then their exchange buffers where rows of 3-byte pixels are stored
template <int SUB, bool BIL, int NCH, int NSW, bool MAPPED>
struct PkLds
{
    static constexpr int kRingWordsRaw = (BIL && (SUB == SUB_420 || SUB == SUB_422)) ? PkStage<SUB, NSW>::kRows * kPkPitch : 1;
    static constexpr int kRingWords = (NCH == 3 && !MAPPED) ? ((kRingWordsRaw + 3) & ~3) : kRingWordsRaw;
    static constexpr int kXposeWords = MAPPED ? 8 * NSW * 256 : 0; // quarter turns: the workgroup's tile, overlaying the chroma blocks
    static constexpr int kPlain = AVIFHIP_PK_WAVES * kRingWords + ((NCH == 3 && !MAPPED) ? AVIFHIP_PK_WAVES * (int)(sizeof(WideRowExchange) / 4) : 0);
    static constexpr int kWords = kPlain > kXposeWords ? kPlain : kXposeWords;
};

template <int SUB, bool BIL, int NCH, bool APLANE, int NSW, bool MAPPED, int WIDE, bool STREAM, int ATT = 0>
__device__ __forceinline__ void pkRunBlock(const TileArgs & A, const PkGeom & g, unsigned * lds, uint32_t tile)
{
    typedef PkRaw<SUB, BIL, APLANE, NSW, WIDE> RawT;
    constexpr int kRingWords = PkLds<SUB, BIL, NCH, NSW, MAPPED>::kRingWords;
    if (tile >= g.nTiles)
        return;
    // un-attenuate: the reciprocal of every alpha code, one entry per thread of the workgroup (64 x 4), behind the chroma blocks; built before any
    // wave leaves -- the barrier is the workgroup's
    unsigned * recip = nullptr;
    if constexpr (ATT == 2) {
        recip = lds + PkLds<SUB, BIL, NCH, NSW, MAPPED>::kPlain;
        const unsigned code = threadIdx.y * (unsigned)kLanesX + threadIdx.x;
        recip[code] = fxUnattenuateReciprocal(code);
        __syncthreads();
    }
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const PkPlace place = pkPlaceOf(tile, wave, g, (uint32_t)NSW);
    const uint32_t wy = wave >> g.wavesXLog2, wavesY = (uint32_t)AVIFHIP_PK_WAVES >> g.wavesXLog2; // the wave's row among the workgroup's stacked waves
    PkSpot w;
    w.band = place.band;
    w.strip0 = place.strip0;
    // 4:2:0 with the four waves stacked: consecutive waves' chroma neighbourhoods overlap in two rows (the halo above and below each
    // wave's NSW rows).  When frames stream from HBM those re-reads are not served by a cache any more (tests/tools/pkbench_wide.hip:
    // the staged kernel costs exactly the halo's bytes more than nearest upsampling), so the workgroup stages ONE neighbourhood of
    // 4 * NSW + 2 rows -- every wave its own NSW rows, the first and the last wave one halo row each -- and meets at one barrier.
    const bool shared = RawT::kStaged && SUB == SUB_420 && g.wavesXLog2 == 0 && (A.tuning & TUNE_PRIVATE_HALO) == 0; // workgroup-uniform
    const bool bandValid = w.band * (uint32_t)kBandW < A.w4, rowsValid = 2u * w.strip0 < A.h2;
    // quarter turns through LDS (pkTransposeStore) need the four waves stacked, all of them to the end
    const bool xpose = MAPPED && A.map.transposed && g.wavesXLog2 == 0 && (A.tuning & TUNE_PRIVATE_HALO) == 0; // workgroup-uniform
    if (!bandValid || (!rowsValid && !shared && !xpose))
        return; // tiles at the right / bottom edge: a wave without work simply leaves (sharing: its rows are still its neighbour's halo)
    w.slotMin = (shared && wy > 0) ? 1 : 0;
    // (every staged row of the wave's own neighbourhood -- NSW + 2 for 4:2:0, 2 * NSW for 4:2:2 -- unless the wave below stages the last one.
    //  Until round 5 this read NSW + 1 for both layouts: right for 4:2:0, two rows short for 4:2:2 with four strips per wave -- the geometry
    //  that images of ~8 megapixels and more select -- whose rows 6 and 7 of every wave were filtered from unstaged LDS.  No test converted a
    //  4:2:2 image that large; tests/test_gpu_parity_libyuv.py::test_packed_kernels_at_every_tile_height does now.)
    w.slotMax = (shared && wy + 1 < wavesY) ? NSW : PkStage<SUB, NSW>::kRows - 1;
    unsigned * ring = shared ? lds + wy * (uint32_t)(NSW * kPkPitch) : lds + wave * (uint32_t)kRingWords;
    // 3-byte pixels, stored as rows: one exchange buffer per wave behind the chroma blocks
    WideRowExchange * xchg = (NCH == 3 && !MAPPED) ? reinterpret_cast<WideRowExchange *>(lds + AVIFHIP_PK_WAVES * PkLds<SUB, BIL, NCH, NSW, MAPPED>::kRingWords) + wave : nullptr;
    RawT raw;
    pkLoad<SUB, BIL, APLANE, NSW, WIDE, STREAM>(A, w, raw);
    pkStage<SUB, BIL, APLANE, NSW, WIDE>(A, w, shared, raw, ring);
    if (!rowsValid && !xpose)
        return;
    pkCompute<SUB, BIL, NCH, APLANE, NSW, MAPPED, WIDE, ATT>(A, w, raw, ring, xchg, xpose ? lds : nullptr, wy, recip);
}

// (grid z = frame of a sequence, tile_shared.h SeqFrames: a single image is a sequence of one)
// (STREAM = false for single images: their planes are assumed cache-resident -- just decoded / uploaded / produced; sequence launches whose
//  bytes exceed the Infinity Cache take streaming luma / alpha loads, launchPkMapped)
template <int SUB, bool BIL, int NCH, bool APLANE, int NSW, bool MAPPED, int WIDE, bool STREAM = false>
__global__ __launch_bounds__(64 * AVIFHIP_PK_WAVES) void yuvToRgbPkKernel(TileArgs A, PkGeom g, SeqFrames S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[]; // PkLds<...>::kPlain words, or kWords for quarter turns (launchPkMapped)
    pkRunBlock<SUB, BIL, NCH, APLANE, NSW, MAPPED, WIDE, STREAM>(seqJob(A, S), g, lds, pkTileOf(blockIdx.x, g));
}

template <int SUB, bool BIL, int NCH, bool APLANE, int NSW, bool MAPPED, int WIDE, bool STREAM = false>
__global__ __launch_bounds__(256) void yuvToRgbPkBatchKernel(const TileArgs * __restrict__ table, PkGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[]; // PkLds<...>::kPlain words, or kWords for quarter turns (launchPkMapped)
    const BatchWhere where = pkBatchWhere(g);
    const TileArgs job = jobOf(table, where.job); // a private copy: see yuvToRgbTileBatchKernel (tile_impl.h)
    pkRunBlock<SUB, BIL, NCH, APLANE, NSW, MAPPED, WIDE, STREAM>(job, g, lds, where.tile);
}

// ... with libyuv's attenuate (ATT = 1: premultiplied outputs) or un-attenuate (ATT = 2: premultiplied images into straight-alpha pixels) pass
// fused in (alpha from the plane, 4-byte pixels, rows)
template <int SUB, bool BIL, int NSW, int WIDE, int ATT>
__global__ __launch_bounds__(256) void yuvToRgbPkAttenuateKernel(TileArgs A, PkGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    pkRunBlock<SUB, BIL, 4, true, NSW, false, WIDE, false, ATT>(A, g, lds, pkTileOf(blockIdx.x, g));
}
template <int SUB, bool BIL, int NSW, int WIDE, int ATT>
__global__ __launch_bounds__(256) void yuvToRgbPkAttenuateBatchKernel(const TileArgs * __restrict__ table, PkGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const BatchWhere where = pkBatchWhere(g);
    const TileArgs job = jobOf(table, where.job);
    pkRunBlock<SUB, BIL, 4, true, NSW, false, WIDE, false, ATT>(job, g, lds, where.tile);
}

template <int SUB, bool BIL, int WIDE, int ATT>
hipError_t launchPkAttenuate(const TileLaunch & L)
{
    uint32_t nsw, blocks;
    PkGeom g;
    pkGeometry(L, L.maxW4, L.maxH2, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid = pkBatchGrid(g, blocks, L.count);
    constexpr uint32_t kTable = (ATT == 2) ? 256u : 0u; // words: the reciprocal of every alpha code (pkRunBlock)
    const uint32_t lds4 = 4u * ((uint32_t)PkLds<SUB, BIL, 4, 4, false>::kPlain + kTable), lds2 = 4u * ((uint32_t)PkLds<SUB, BIL, 4, 2, false>::kPlain + kTable);
    if (L.seq)
        return hipErrorNotSupported;
    if (L.table) {
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbPkAttenuateBatchKernel<SUB, BIL, 4, WIDE, ATT>), grid, block, lds4, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbPkAttenuateBatchKernel<SUB, BIL, 2, WIDE, ATT>), grid, block, lds2, L.stream, L.table, g);
    } else {
        if (nsw == 4)
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbPkAttenuateKernel<SUB, BIL, 4, WIDE, ATT>), grid, block, lds4, L.stream, *L.args, g);
        else
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbPkAttenuateKernel<SUB, BIL, 2, WIDE, ATT>), grid, block, lds2, L.stream, *L.args, g);
    }
    return hipGetLastError();
}

template <int SUB, bool BIL, int NCH, bool APLANE, bool MAPPED, int WIDE = WIDE_NONE>
hipError_t launchPkMapped(const TileLaunch & L)
{
    uint32_t nsw, blocks;
    PkGeom g;
    pkGeometry(L, L.maxW4, L.maxH2, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid = pkBatchGrid(g, blocks, L.count);
    // LDS: the waves' chroma blocks (+ exchange buffers); quarter turns overlay them with the workgroup's transposition tile
    const bool xpose = MAPPED && L.transposed;
    const uint32_t lds4 = 4u * (uint32_t)(xpose ? PkLds<SUB, BIL, NCH, 4, MAPPED>::kWords : PkLds<SUB, BIL, NCH, 4, MAPPED>::kPlain);
    const uint32_t lds2 = 4u * (uint32_t)(xpose ? PkLds<SUB, BIL, NCH, 2, MAPPED>::kWords : PkLds<SUB, BIL, NCH, 2, MAPPED>::kPlain);
    if (L.table && L.streamLoads && !MAPPED) { // batches whose planes exceed the Infinity Cache
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbPkBatchKernel<SUB, BIL, NCH, APLANE, 4, false, WIDE, true>), grid, block, lds4, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbPkBatchKernel<SUB, BIL, NCH, APLANE, 2, false, WIDE, true>), grid, block, lds2, L.stream, L.table, g);
    } else if (L.table) {
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbPkBatchKernel<SUB, BIL, NCH, APLANE, 4, MAPPED, WIDE>), grid, block, lds4, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbPkBatchKernel<SUB, BIL, NCH, APLANE, 2, MAPPED, WIDE>), grid, block, lds2, L.stream, L.table, g);
    } else {
        if (L.seq && MAPPED)
            return hipErrorNotSupported; // (a pixel map is per job)
        const SeqFrames S = L.seq ? *L.seq : seqOfOne(*L.args);
        const dim3 frames(grid.x, 1, L.seq ? L.seqCount : 1u);
        (void)S, (void)frames; // (seam-aware builds compile no single launches)
        if (nsw == 4)
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbPkKernel<SUB, BIL, NCH, APLANE, 4, MAPPED, WIDE>), frames, block, lds4, L.stream, *L.args, g, S);
        else
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbPkKernel<SUB, BIL, NCH, APLANE, 2, MAPPED, WIDE>), frames, block, lds2, L.stream, *L.args, g, S);
    }
    return hipGetLastError();
}

template <int SUB, bool BIL, int NCH, bool APLANE>
hipError_t launchPk(const TileLaunch & L)
{
    return L.mapped ? launchPkMapped<SUB, BIL, NCH, APLANE, true>(L) : launchPkMapped<SUB, BIL, NCH, APLANE, false>(L);
}

// 16-bit containers: libyuv's high-bit-depth entries, or libavif's reduction to 8 bits followed by the 8-bit entry (plan.cpp: fxDownshift)
template <int SUB, bool BIL, int NCH, bool APLANE>
hipError_t launchPkWide(const TileLaunch & L)
{
    if (L.mapped)
        return L.wideDownshift ? launchPkMapped<SUB, BIL, NCH, APLANE, true, WIDE_DOWNSHIFT>(L) : launchPkMapped<SUB, BIL, NCH, APLANE, true, WIDE_NATIVE>(L);
    return L.wideDownshift ? launchPkMapped<SUB, BIL, NCH, APLANE, false, WIDE_DOWNSHIFT>(L) : launchPkMapped<SUB, BIL, NCH, APLANE, false, WIDE_NATIVE>(L);
}

} // namespace AVIFHIP_TILE_BUILD
} // namespace tile
} // namespace avifhip
