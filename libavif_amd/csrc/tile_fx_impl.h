// tile_fx_impl.h -- bandwidth-tuned YUV->RGB kernels for the reference's INTEGER path (libyuv's fixed-point arithmetic,
// SURVEY.md appendix D.1-D.4, as dispatched by src/reformat_libyuv.c), instantiated by kernels_tile_fx_inst.hip.
//
// Same work decomposition, loads, software pipeline and stores as the fp32 tiled kernels (tile_impl.h); what differs is
// the arithmetic, which is integer-only:
//   * bilinear chroma: the tile's chroma neighbourhood is staged in LDS as RAW samples, one 32-bit word per chroma
//     column holding (u | v << 16); libyuv's 9:3:3:1 filter with its +8 >> 4 rounding is evaluated on both planes at
//     once with plain 32-bit shifts and adds (no field exceeds 16 * 1023 + 8, so no carry crosses the halves), sharing
//     the 9x / 3x terms between a lane's 8 pixels.  Coordinates clamp to the canvas exactly like the fp32 staging:
//     with a duplicated neighbour the 2-D formula (12a + 4b + 8) >> 4 IS libyuv's edge formula (3a + b + 2) >> 2, and
//     (3a + a + 2) >> 2 == a, so the first column / row and the last column of an even width / last row of an even height
//     need no special case (the last column of an ODD width never reaches these kernels: leftover columns go to the
//     universal kernel);
//   * matrix: one 24-bit multiply for luma, four multiply-adds for the chroma terms with the biases folded in, shifts,
//     integer clamps, byte packing.
// Scope: 8-bit RGB / BGR / RGBA / BGRA / ARGB / ABGR outputs from 8-bit planes, 10-bit planes through the I010 family or
// any depth through the downshift route; alpha opaque / from the plane (shift or fp32 rescale); ARGBAttenuate /
// ARGBUnattenuate post-pass.
// Since round 2 these kernels are the fall-back of the packed 16-bit ones (tile_pk_impl.h, dispatched from launchOneFx below):
// what they still serve by default is ARGBUnattenuate, and the attenuate pass behind a pixel map; with TUNE_COOPERATIVE they take
// the 10/12-bit planes back for A/B runs.
#pragma once

#include "pixel_fixed.h"
#include "tile_impl.h"
#include "tile_pk_impl.h"

namespace avifhip {
namespace tile {
inline namespace AVIFHIP_TILE_BUILD { // (tile_impl.h: the plain and the seam-aware build of a family)

// LDS row of staged chroma: entry c+5 holds (u | v << 16) of chroma column cxb + c, c in [-4, 131]
constexpr int kFxRowPitch = 140;

__device__ __forceinline__ unsigned fxReduce(unsigned v, unsigned downshift)
{
    return downshift ? minU(v >> downshift, 255u) : v;
}
// ... for the packed filter: samples of 16-bit containers are additionally held to 12 bits so that no field of the packed
// sums can carry into its neighbour (fxChromaBilinear of the universal kernel applies the same bound; samples inside
// their nominal depth never reach it)
template <typename YT>
__device__ __forceinline__ unsigned fxReduceForFilter(unsigned v, unsigned downshift)
{
    if constexpr (sizeof(YT) == 2)
        return downshift ? minU(v >> downshift, 255u) : minU(v, 4095u);
    else
        return v;
}

// 8-bit planes through a filter: the staged words are pre-scaled by the filter's divisor-to-256 factor (4:2:0: sums of weight
// 16, x16; 4:2:2: weight 4, x64), so that the filtered, rounded and shifted sample lands in byte 1 of its 16-bit field --
// where the matrix multiplies read it with a byte select instead of a bit-field extract per plane and pixel.  No field
// overflows: 255 * 256 + 128 < 65536.
template <typename YT, int SUB, bool BIL>
struct FxPrescale
{
    static constexpr bool kOn = BIL && sizeof(YT) == 1 && (SUB == SUB_420 || SUB == SUB_422);
    static constexpr int kShift = !kOn ? 0 : (SUB == SUB_420 ? 4 : 6);
};

template <typename YT, int SUB, bool NEEDA, int NS, int WAVES = 4>
__device__ __forceinline__ void stageTileFx(const TileArgs & A, const TileRaw<YT, SUB, true, NEEDA, NS, WAVES> & T, unsigned (*rows)[kFxRowPitch])
{
    typedef StageRows<SUB, NS, WAVES> SR;
    const int t = ((WAVES == 1) ? 0 : (int)threadIdx.y * kLanesX) + (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < SR::kRounds; ++j) {
        int row, grp;
        if (SR::placeOf(t, j, row, grp)) {
            unsigned u[4], v[4];
            decode4<YT>(T.su[j], u);
            decode4<YT>(T.sv[j], v);
            unsigned * dst = &rows[row][4 * grp + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dst[k] = (fxReduceForFilter<YT>(u[k], A.fx.downshift) | (fxReduceForFilter<YT>(v[k], A.fx.downshift) << 16)) << FxPrescale<YT, SUB, true>::kShift;
        }
    }
    if constexpr (SR::kSplitHalo) { // (tile_impl.h StageRows: the side samples, one lane each)
        if (t < 2 * SR::kRows)
            rows[t >> 1][(t & 1) ? 4 * 33 + 1 : 4] = (fxReduceForFilter<YT>(T.hu, A.fx.downshift) | (fxReduceForFilter<YT>(T.hv, A.fx.downshift) << 16)) << FxPrescale<YT, SUB, true>::kShift;
    }
}

// 3x and 9x of a packed (u | v << 16) word.  Spelled as the shift-and-add instruction: from (x << 3) + x the compiler builds
// x * 9 and, where an addend follows, v_mad_u64_u32 -- a quarter-rate 32 x 32 multiplier for what one full-rate
// v_lshl_add_u32 does (six of them per 8 pixels of the 4:2:0 filter).
__device__ __forceinline__ unsigned fx3(unsigned x)
{
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned fx9(unsigned x)
{
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 3, %1" : "=v"(r) : "v"(x));
    return r;
}


template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS, int WAVES = 4>
__device__ __forceinline__ void computeTileFx(const TileArgs & A, const BandCtx & c, uint32_t tileY, const TileRaw<YT, SUB, BIL, APLANE || HASMUL, NS, WAVES> & T,
                                              unsigned (*rows)[kFxRowPitch])
{
    constexpr bool kWide = sizeof(YT) == 2;
    constexpr bool kNeedA = APLANE || HASMUL;
    const int tx = threadIdx.x, wv = (WAVES == 1) ? 0 : (int)threadIdx.y;
    const bool nt = (A.tuning & TUNE_NONTEMPORAL) != 0;
    const uint32_t X = c.X;
    const bool laneValid = c.laneValid;
    const StripRaw<YT, SUB, BIL, kNeedA> * raw = T.raw;
    const TileArgs::Fx & F = A.fx;

#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const uint32_t sy = tileY + 2 * (wv * NS + k);
        if (sy >= A.h2)
            break;

        // ---- upsampled (u, v) of the lane's 2 x 4 pixels, at the planes' (reduced) depth, still packed u | v << 16 ----
        unsigned uv[2][4];
        if constexpr (SUB == SUB_400) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    uv[r][i] = (128u << F.cShr) | ((128u << F.cShr) << 16);
        } else if constexpr (SUB == SUB_444) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned u[4], v[4];
                decode4<YT>(raw[k].u[r], u);
                decode4<YT>(raw[k].v[r], v);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    uv[r][i] = fxReduce(u[i], F.downshift) | (fxReduce(v[i], F.downshift) << 16);
            }
        } else if constexpr (!BIL) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (SUB == SUB_420 && r == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        uv[1][i] = uv[0][i];
                    break;
                }
                constexpr unsigned kMask = kWide ? 0xffffu : 0xffu;
                constexpr int kShift = kWide ? 16 : 8;
                const unsigned u0 = raw[k].u[r].w[0] & kMask, u1 = (raw[k].u[r].w[0] >> kShift) & kMask;
                const unsigned v0 = raw[k].v[r].w[0] & kMask, v1 = (raw[k].v[r].w[0] >> kShift) & kMask;
                const unsigned c0 = fxReduce(u0, F.downshift) | (fxReduce(v0, F.downshift) << 16);
                const unsigned c1 = fxReduce(u1, F.downshift) | (fxReduce(v1, F.downshift) << 16);
                uv[r][0] = uv[r][1] = c0;
                uv[r][2] = uv[r][3] = c1;
            }
        } else {
            auto loadRow = [&](int q, unsigned m[4]) {
                const u2 * src = reinterpret_cast<const u2 *>(&rows[q][2 * tx + 4]); // columns 2tx-1 .. 2tx+2, 8-byte aligned
                const u2 lo = src[0], hi = src[1];
                m[0] = lo.x, m[1] = lo.y, m[2] = hi.x, m[3] = hi.y;
            };
            if constexpr (SUB == SUB_420) {
                // Scale2RowUp_Bilinear: (9 near + 3 horizontal + 3 vertical + 1 diagonal + 8) >> 4 per plane
                const int qm = 1 + wv * NS + k;
                unsigned m[4], f[2][4];
                loadRow(qm, m);
                loadRow(qm - 1, f[0]); // even luma rows lean to the chroma row above
                loadRow(qm + 1, f[1]); // odd luma rows to the one below
                constexpr unsigned kRound = 0x00080008u << FxPrescale<YT, SUB, BIL>::kShift;
                const unsigned n9b = fx9(m[1]), n9c = fx9(m[2]);
                const unsigned h0 = n9b + fx3(m[0]) + kRound, h1 = n9b + fx3(m[2]) + kRound;
                const unsigned h2 = n9c + fx3(m[1]) + kRound, h3 = n9c + fx3(m[3]) + kRound;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned g3b = fx3(f[r][1]), g3c = fx3(f[r][2]);
                    uv[r][0] = h0 + g3b + f[r][0]; // even pixel: horizontal neighbour on the left
                    uv[r][1] = h1 + g3b + f[r][2]; // odd pixel: on the right
                    uv[r][2] = h2 + g3c + f[r][1];
                    uv[r][3] = h3 + g3c + f[r][3];
                }
            } else { // 4:2:2, ScaleRowUp2_Linear: (3 near + far + 2) >> 2
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    unsigned m[4];
                    loadRow(2 * (wv * NS + k) + r, m);
                    constexpr unsigned kRound = 0x00020002u << FxPrescale<YT, SUB, BIL>::kShift;
                    const unsigned n3b = fx3(m[1]) + kRound, n3c = fx3(m[2]) + kRound;
                    uv[r][0] = n3b + m[0];
                    uv[r][1] = n3b + m[2];
                    uv[r][2] = n3c + m[1];
                    uv[r][3] = n3c + m[3];
                }
            }
        }

        // ---- matrix, alpha, stores ----
        // the filter's final ">> SH" is folded into the field extraction: uv holds the un-shifted sums
        constexpr int SH = (BIL ? (SUB == SUB_420 ? 4 : 2) : 0) + FxPrescale<YT, SUB, BIL>::kShift; // 8 with the pre-scale: byte 1
        constexpr unsigned kField = BIL ? 0xfffu : 0xffffu; // filtered fields are 12 bits wide after the shift, plain samples 16
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            unsigned yv[4], av[4] = { 0, 0, 0, 0 };
            decode4<YT>(raw[k].y[r], yv);
            if constexpr (kNeedA) {
                decode4<YT>(raw[k].a[r], av);
                if (A.alphaLim.on) { // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[i] = alphaToFullRange(A, av[i]);
                }
            }
            unsigned X4[4], G4[4], Z4[4], a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // operands fit 24 bits (y32 < 2^16, yMul < 2^15, yMul8 < 2^23, chroma < 2^8, coefficients < 2^10): full-rate multiplies
                unsigned y1;
                int lo, hi;
                if constexpr (!kWide) {
                    y1 = __umul24(yv[i], F.yMul8) >> 16;
                    lo = (int)((uv[r][i] >> SH) & 0xffu), hi = (int)((uv[r][i] >> (16 + SH)) & 0xffu);
                } else {
                    const unsigned y = fxReduce(yv[i], F.downshift);
                    y1 = __umul24((y << F.yShl) | (y >> F.yShr), F.yMul) >> 16;
                    lo = (int)minU(((uv[r][i] >> SH) & kField) >> F.cShr, 255u), hi = (int)minU(((uv[r][i] >> (16 + SH)) & kField) >> F.cShr, 255u);
                }
                const int yx = (int)(y1 << 2) + F.kX4, yz = (int)(y1 << 2) + F.kZ4, yg = (int)(y1 << 2) + F.kG4;
                const int x = yx + __mul24(F.cX4, lo);
                const int z = yz + __mul24(F.cZ4, hi);
                const int g = (yg - __mul24(F.gLo4, lo)) - __mul24(F.gHi4, hi);
                X4[i] = (unsigned)min(max(x, 0), 65535);
                Z4[i] = (unsigned)min(max(z, 0), 65535);
                G4[i] = (unsigned)min(max(g, 0), 65535);
                if constexpr (APLANE) {
                    if (F.alphaMode == FXA_SHIFT)
                        a[i] = minU(av[i] >> F.alphaShift, 255u);
                    else
                        a[i] = alphaFromPlane(A, av[i]); // avifReformatAlpha over libyuv's 255: copy or fp32 rescale
                } else {
                    a[i] = 255u;
                }
            }
            if constexpr (NCH == 4) {
                u4 w;
                if constexpr (HASMUL) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        unsigned x = X4[i] >> 8, g = G4[i] >> 8, z = Z4[i] >> 8;
                        if (!A.postMulFx) {
                            // (wave-uniform) ARGB / ABGR: libyuv attenuates RGBA / BGRA only, so the reference runs its own fp32 post-pass over
                            // libyuv's bytes (src/alpha.c:171-246, 358-430)
                            x = alphaMulInt(x & 0xffu, a[i], 255u, 255.0f, A.postMul), g = alphaMulInt(g & 0xffu, a[i], 255u, 255.0f, A.postMul);
                            z = alphaMulInt(z & 0xffu, a[i], 255u, 255.0f, A.postMul);
                        } else if (A.postMul == MUL_MULTIPLY) { // ARGBAttenuate / ARGBUnattenuate
                            x = fxAttenuate(x, a[i]), g = fxAttenuate(g, a[i]), z = fxAttenuate(z, a[i]);
                        } else if (A.postMul == MUL_UNMULTIPLY) {
                            const unsigned ia = fxUnattenuateReciprocal(a[i]); // once per pixel, no integer division
                            x = fxUnattenuateBy(x, ia), g = fxUnattenuateBy(g, ia), z = fxUnattenuateBy(z, ia);
                        }
                        w[i] = __builtin_amdgcn_perm(x, g, F.selXGm) | __builtin_amdgcn_perm(z, a[i], F.selZAm);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        w[i] = __builtin_amdgcn_perm(X4[i], G4[i], F.selXG) | __builtin_amdgcn_perm(Z4[i], a[i], F.selZA);
                }
                if (laneValid)
                    storeVec(A.rgb, (sy + r) * A.rgbPitch + X * 4, w, nt);
            } else {
                // 12 bytes: x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3, each byte = byte 1 of its 4x-scale value
                constexpr unsigned kLoPair = 0x0c0c0105u, kHiPair = 0x01050c0cu;
                const unsigned w0 = __builtin_amdgcn_perm(X4[0], G4[0], kLoPair) | __builtin_amdgcn_perm(Z4[0], X4[1], kHiPair);
                const unsigned w1 = __builtin_amdgcn_perm(G4[1], Z4[1], kLoPair) | __builtin_amdgcn_perm(X4[2], G4[2], kHiPair);
                const unsigned w2 = __builtin_amdgcn_perm(Z4[2], X4[3], kLoPair) | __builtin_amdgcn_perm(G4[3], Z4[3], kHiPair);
                if (laneValid) {
                    const uint32_t off = (sy + r) * A.rgbPitch + X * 3;
                    storeVec(A.rgb, off, w0, nt);
                    storeVec(A.rgb, off + 4, w1, nt);
                    storeVec(A.rgb, off + 8, w2, nt);
                }
            }
        }
    }
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__device__ __forceinline__ void runBlockFx(const TileArgs & A, uint32_t tilesPerRun, unsigned (*rows)[BIL ? StageRows<SUB, NS>::kRows : 1][kFxRowPitch])
{
    constexpr int kTileH = 8 * NS;
    constexpr bool kNeedA = APLANE || HASMUL;
    const uint32_t bands = (A.w4 + kBandW - 1) / kBandW;
    const uint32_t tilesY = (A.h2 + kTileH - 1) / kTileH;
    const uint32_t runsY = (tilesY + tilesPerRun - 1) / tilesPerRun;
    const uint32_t nRuns = bands * runsY;
    const uint32_t run = blockRemap(blockIdx.x, gridDim.x, (A.tuning & TUNE_XCD_BANDS) != 0 && (gridDim.z == 1 || (gridDim.x & 7) == 0));
    if (run >= nRuns)
        return;
    const uint32_t rrow = run / bands;
    const uint32_t bandX = (run - rrow * bands) * kBandW;
    const uint32_t firstTile = rrow * tilesPerRun;
    const uint32_t nTiles = (tilesY - firstTile < tilesPerRun) ? (tilesY - firstTile) : tilesPerRun;

    BandCtx c;
    c.bandX = bandX;
    c.X = bandX + 4 * threadIdx.x;
    c.laneValid = c.X < A.w4;
    c.Xc = c.laneValid ? c.X : 0;
    c.cxb = A.cx0 + (int)(bandX >> 1);

    // double-buffered LDS, one barrier per tile (see runBlock in tile_impl.h)
    TileRaw<YT, SUB, BIL, kNeedA, NS> cur;
    uint32_t tileY = firstTile * kTileH;
    loadTile<YT, SUB, BIL, kNeedA, NS>(A, c, tileY, cur);
    if constexpr (BIL) {
        stageTileFx<YT, SUB, kNeedA, NS>(A, cur, rows[0]);
        __syncthreads();
    }
    for (uint32_t i = 0; i < nTiles; ++i) {
        const bool more = i + 1 < nTiles;
        TileRaw<YT, SUB, BIL, kNeedA, NS> nxt;
        if (more)
            loadTile<YT, SUB, BIL, kNeedA, NS>(A, c, tileY + kTileH, nxt);
        computeTileFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS>(A, c, tileY, cur, rows[i & 1]);
        if (!more)
            break;
        if constexpr (BIL) {
            stageTileFx<YT, SUB, kNeedA, NS>(A, nxt, rows[(i + 1) & 1]);
            __syncthreads();
        }
        cur = nxt;
        tileY += kTileH;
    }
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__global__ __launch_bounds__(256) void yuvToRgbTileFxKernel(TileArgs A, uint32_t tilesPerRun)
{
    __shared__ __attribute__((aligned(16))) unsigned rows[BIL ? 2 : 1][BIL ? StageRows<SUB, NS>::kRows : 1][kFxRowPitch];
    runBlockFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS>(A, tilesPerRun, rows);
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__global__ __launch_bounds__(256) void yuvToRgbTileFxBatchKernel(const TileArgs * __restrict__ table, uint32_t tilesPerRun)
{
    __shared__ __attribute__((aligned(16))) unsigned rows[BIL ? 2 : 1][BIL ? StageRows<SUB, NS>::kRows : 1][kFxRowPitch];
    const TileArgs job = jobOf<true>(table); // a private copy: see yuvToRgbTileBatchKernel (tile_impl.h)
    runBlockFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS>(job, tilesPerRun, rows);
}

// ---- every wave for itself (see runSolo, tile_impl.h) ----
template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__device__ __forceinline__ void runSoloFx(const TileArgs & A, const PkGeom & g, unsigned * lds)
{
    constexpr bool kNeedA = APLANE || HASMUL;
    typedef StageRows<SUB, NS, 1> SR;
    const uint32_t tile = pkTileOf(blockIdx.x, g);
    if (tile >= g.nTiles)
        return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const PkPlace place = pkPlaceOf(tile, wave, g, (uint32_t)NS);
    const uint32_t bandX = place.band * (uint32_t)kBandW;
    const uint32_t tileY = place.strip0 * 2u;
    if (bandX >= A.w4 || tileY >= A.h2)
        return;
    BandCtx c;
    c.bandX = bandX;
    c.X = bandX + 4 * threadIdx.x;
    c.laneValid = c.X < A.w4;
    c.Xc = c.laneValid ? c.X : 0;
    c.cxb = A.cx0 + (int)(bandX >> 1);
    TileRaw<YT, SUB, BIL, kNeedA, NS, 1> raw;
    loadTile<YT, SUB, BIL, kNeedA, NS, 1>(A, c, tileY, raw);
    unsigned(*rows)[kFxRowPitch] = reinterpret_cast<unsigned(*)[kFxRowPitch]>(lds + (size_t)wave * (BIL ? SR::kRows : 1) * kFxRowPitch);
    if constexpr (BIL) {
        stageTileFx<YT, SUB, kNeedA, NS, 1>(A, raw, rows);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    computeTileFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS, 1>(A, c, tileY, raw, rows);
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__global__ __launch_bounds__(256) void yuvToRgbTileFxSoloKernel(TileArgs A, PkGeom g)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[kWavesPerBlock * (BIL ? StageRows<SUB, NS, 1>::kRows : 1) * kFxRowPitch];
    runSoloFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS>(A, g, lds);
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool HASMUL, int NS>
__global__ __launch_bounds__(256) void yuvToRgbTileFxSoloBatchKernel(const TileArgs * __restrict__ table, PkGeom g)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[kWavesPerBlock * (BIL ? StageRows<SUB, NS, 1>::kRows : 1) * kFxRowPitch];
    const TileArgs job = jobOf(table);
    runSoloFx<YT, SUB, BIL, NCH, APLANE, HASMUL, NS>(job, g, lds);
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool MUL>
hipError_t launchSoloFx(const TileLaunch & L)
{
    uint32_t nsw, blocks;
    PkGeom g;
    pkGeometry(L, L.maxW4, L.maxH2, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid(blocks, 1, L.count);
    if (L.table) {
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbTileFxSoloBatchKernel<YT, SUB, BIL, NCH, APLANE, MUL, 4>), grid, block, 0, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbTileFxSoloBatchKernel<YT, SUB, BIL, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, L.table, g);
    } else {
        if (nsw == 4)
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileFxSoloKernel<YT, SUB, BIL, NCH, APLANE, MUL, 4>), grid, block, 0, L.stream, *L.args, g);
        else
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileFxSoloKernel<YT, SUB, BIL, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, *L.args, g);
    }
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL, int NCH, bool APLANE, bool MUL>
hipError_t launchOneFx(const TileLaunch & L)
{
    // 8-bit planes without a post-pass: the packed 16-bit kernels (tile_pk_impl.h)
    if constexpr (sizeof(YT) == 1 && !MUL)
        return launchPk<SUB, BIL, NCH, APLANE>(L);
    // libyuv's ARGBAttenuate / ARGBUnattenuate after the conversion (premultiplied outputs / premultiplied images into straight pixels): the packed
    // kernels with the pass fused in
    if constexpr (MUL) {
        if (L.attenuate == 1 && !L.mapped) {
            if constexpr (sizeof(YT) == 1)
                return launchPkAttenuate<SUB, BIL, WIDE_NONE, 1>(L);
            else if (L.pkWide)
                return L.wideDownshift ? launchPkAttenuate<SUB, BIL, WIDE_DOWNSHIFT, 1>(L) : launchPkAttenuate<SUB, BIL, WIDE_NATIVE, 1>(L);
        }
        if (L.attenuate == 2 && !L.mapped) { // (round 5: the un-attenuate pass too)
            if constexpr (sizeof(YT) == 1)
                return launchPkAttenuate<SUB, BIL, WIDE_NONE, 2>(L);
            else if (L.pkWide)
                return L.wideDownshift ? launchPkAttenuate<SUB, BIL, WIDE_DOWNSHIFT, 2>(L) : launchPkAttenuate<SUB, BIL, WIDE_NATIVE, 2>(L);
        }
    }
    // 10- / 12-bit planes without a post-pass: the same kernels behind a front end for 16-bit containers
    if constexpr (sizeof(YT) == 2 && !MUL) {
        if (L.pkWide)
            return launchPkWide<SUB, BIL, NCH, APLANE>(L);
    }
    if (L.seq)
        return hipErrorNotSupported; // (sequences: the packed kernels above)
    if (L.solo)
        return launchSoloFx<YT, SUB, BIL, NCH, APLANE, MUL>(L);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid(L.blocksPerJob, 1, L.count);
    if (L.table && L.stripsPerWave >= 2)
        hipLaunchKernelGGL((yuvToRgbTileFxBatchKernel<YT, SUB, BIL, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, L.table, L.tilesPerRun);
    else if (L.table)
        hipLaunchKernelGGL((yuvToRgbTileFxBatchKernel<YT, SUB, BIL, NCH, APLANE, MUL, 1>), grid, block, 0, L.stream, L.table, L.tilesPerRun);
    else if (L.stripsPerWave >= 2)
        AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileFxKernel<YT, SUB, BIL, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, *L.args, L.tilesPerRun);
    else
        AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileFxKernel<YT, SUB, BIL, NCH, APLANE, MUL, 1>), grid, block, 0, L.stream, *L.args, L.tilesPerRun);
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL>
hipError_t launchFxVariant(const TileKey & k, const TileLaunch & L)
{
    if constexpr (sizeof(YT) == 1 && !BIL && (SUB == SUB_420 || SUB == SUB_422)) {
        if (k.nch == 2) // RGB565: the packed 16-bit kernels only (tileYuvToRgbSupported admits nothing else)
            return launchPk<SUB, BIL, 2, false>(L);
    }
    if constexpr (sizeof(YT) == 2 && !BIL && (SUB == SUB_420 || SUB == SUB_422)) {
        if (k.nch == 2) // ... from 10- / 12-bit planes: Convert16To8Plane in the kernel's front end, then the same arithmetic
            return L.wideDownshift ? launchPkMapped<SUB, BIL, 2, false, false, WIDE_DOWNSHIFT>(L) : hipErrorInvalidValue;
    }
    if (k.nch == 3)
        return launchOneFx<YT, SUB, BIL, 3, false, false>(L);
    if (k.hasMul)
        return launchOneFx<YT, SUB, BIL, 4, true, true>(L);
    return k.alphaPlane ? launchOneFx<YT, SUB, BIL, 4, true, false>(L) : launchOneFx<YT, SUB, BIL, 4, false, false>(L);
}

} // namespace AVIFHIP_TILE_BUILD
} // namespace tile
} // namespace avifhip
