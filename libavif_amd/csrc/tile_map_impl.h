// tile_map_impl.h -- stores of the fp32 tiles through a PixelMap (plan.h): the application's clean-aperture crop, rotation and mirror
// (avifApplyTransforms, apps/shared/avifutil.c:787-825) fused into the conversion's stores, for 4-channel pixels of 4 bytes (RGBA8) and
// 8 bytes (RGBA in 16-bit containers: what avifImageYUVToRGB hands out by default for 10- and 12-bit images, src/avif.c:704).  The packed
// integer kernels have their own 4-byte twins (tile_pk_impl.h: pkStoreMappedRow, pkTransposeWrite / pkTransposeStore); the layouts here
// follow them.
//
// A pixel is PW = 1 or 2 dwords.  Rows stay rows for no rotation / half turns: a lane's 4 pixels leave as 16-byte stores, reversed when
// mirrored, at whatever dword alignment the crop leaves; 8-byte pixels first go through the wave's exchange buffer so that every store
// instruction still covers one contiguous KiB (tile_impl.h store4WideRgba has the measurements).  Quarter turns: the workgroup's tile of
// 256 columns x ROWS rows is transposed through LDS, so that a source COLUMN leaves as one run of ROWS consecutive destination pixels
// (ROWS * 4 * PW = 128 bytes) and a store instruction writes 8 such runs, 16 bytes per lane.
#pragma once

#include <hip/hip_runtime.h>

#include "tile_shared.h"

namespace avifhip {
namespace tile {

typedef unsigned mu4 __attribute__((ext_vector_type(4)));
typedef unsigned mu4a4 __attribute__((ext_vector_type(4), aligned(4))); // crops start anywhere: 16-byte accesses at dword alignment
typedef unsigned mu2 __attribute__((ext_vector_type(2)));
typedef unsigned mu2a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int PW>
__device__ __forceinline__ void mapStorePixel(uint8_t * dst, const unsigned (&px)[PW])
{
    if constexpr (PW == 1)
        *reinterpret_cast<unsigned *>(dst) = px[0];
    else
        *reinterpret_cast<mu2a4 *>(dst) = (mu2) { px[0], px[1] };
}

// destination address of cropped-canvas coordinates (ii, jj) for rows-stay-rows maps
__device__ __forceinline__ uint8_t * mapRowBase(const TileArgs & A, uint32_t jj)
{
    return A.rgb + (size_t)(uint32_t)(A.map.sy * (int32_t)jj + A.map.ky) * A.rgbPitch;
}

// rows stay rows: the lane's four pixels of canvas row j, first at canvas column i (4-byte pixels: one 16-byte store per lane)
template <int PW>
__device__ __forceinline__ void mapStoreRow(const TileArgs & A, const unsigned (&px)[4][PW], uint32_t i, uint32_t j, bool nt)
{
    const PixelMap & m = A.map;
    const uint32_t ii = i - m.cx, jj = j - m.cy;
    if (jj >= m.ch)
        return;
    uint8_t * row = mapRowBase(A, jj);
    constexpr uint32_t PB = 4u * PW;
    const bool fwd = m.sx > 0;
    if (ii < m.cw && m.cw - ii >= 4u) { // all four inside the crop
        const uint32_t x = (uint32_t)(fwd ? (int32_t)ii + m.kx : m.kx - (int32_t)(ii + 3u));
        uint8_t * dst = row + (size_t)x * PB;
        if constexpr (PW == 1) {
            const mu4 v = fwd ? (mu4) { px[0][0], px[1][0], px[2][0], px[3][0] } : (mu4) { px[3][0], px[2][0], px[1][0], px[0][0] };
            if (nt)
                __builtin_nontemporal_store(v, reinterpret_cast<mu4a4 *>(dst));
            else
                *reinterpret_cast<mu4a4 *>(dst) = v;
        } else {
            const mu4 lo = fwd ? (mu4) { px[0][0], px[0][1], px[1][0], px[1][1] } : (mu4) { px[3][0], px[3][1], px[2][0], px[2][1] };
            const mu4 hi = fwd ? (mu4) { px[2][0], px[2][1], px[3][0], px[3][1] } : (mu4) { px[1][0], px[1][1], px[0][0], px[0][1] };
            if (nt) {
                __builtin_nontemporal_store(lo, reinterpret_cast<mu4a4 *>(dst));
                __builtin_nontemporal_store(hi, reinterpret_cast<mu4a4 *>(dst + 16));
            } else {
                *reinterpret_cast<mu4a4 *>(dst) = lo;
                *reinterpret_cast<mu4a4 *>(dst + 16) = hi;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t iik = ii + (uint32_t)k;
        if (iik < m.cw)
            mapStorePixel<PW>(row + (size_t)(uint32_t)(m.sx * (int32_t)iik + m.kx) * PB, px[k]);
    }
}

// ... 8-byte pixels through the wave's exchange buffer (2 KiB, private to the wave): written in lane order (32 bytes per lane), read back so
// that store instruction h covers the segment's pixels 128 h .. 128 h + 127 with lane l holding pixels 128 h + 2 l and + 1 -- one
// contiguous KiB per instruction in the destination too, forwards or mirrored.  Must be called by every lane of the wave.
// `bandI`: canvas column of the wave's first pixel; `w4`: columns of the rectangle beyond which lanes hold nothing.
__device__ __forceinline__ void mapStoreRowWide(const TileArgs & A, const unsigned (&px)[4][2], uint32_t bandI, uint32_t bandX, uint32_t j, mu4 * xchg)
{
    const PixelMap & m = A.map;
    const uint32_t l = threadIdx.x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    xchg[2 * l] = (mu4) { px[0][0], px[0][1], px[1][0], px[1][1] };
    xchg[2 * l + 1] = (mu4) { px[2][0], px[2][1], px[3][0], px[3][1] };
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const mu4 s[2] = { xchg[l], xchg[64 + l] };
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t jj = j - m.cy;
    if (jj >= m.ch)
        return;
    uint8_t * row = mapRowBase(A, jj);
    const bool fwd = m.sx > 0;
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t p = 128u * h + 2u * l; // the pair's first pixel within the wave's segment
        if (bandX + p >= A.w4)
            continue; // (w4 is a multiple of 4: pairs are whole or absent)
        const uint32_t ii = bandI + p - m.cx;
        const bool in0 = ii < m.cw, in1 = ii + 1u < m.cw; // (ii wraps to a huge value left of the crop)
        if (in0 && in1) {
            const uint32_t x = (uint32_t)(fwd ? (int32_t)ii + m.kx : m.kx - (int32_t)(ii + 1u));
            const mu4 v = fwd ? s[h] : (mu4) { s[h].z, s[h].w, s[h].x, s[h].y };
            __builtin_nontemporal_store(v, reinterpret_cast<mu4a4 *>(row + (size_t)x * 8u));
        } else {
            if (in0)
                *reinterpret_cast<mu2a4 *>(row + (size_t)(uint32_t)(m.sx * (int32_t)ii + m.kx) * 8u) = (mu2) { s[h].x, s[h].y };
            if (in1)
                *reinterpret_cast<mu2a4 *>(row + (size_t)(uint32_t)(m.sx * (int32_t)(ii + 1u) + m.kx) * 8u) = (mu2) { s[h].z, s[h].w };
        }
    }
}

// ---- quarter turns: the workgroup's tile (256 columns x ROWS rows, the four waves stacked, RW = ROWS / 4 rows each) through LDS ----
// Layout, in pixels: (y, x) at y * 256 + (x ^ swizzle(y)); the swizzle moves whole 16-byte groups, so rows go in with aligned 16-byte LDS
// stores, and spreads the rows of one column over the banks, so columns come out two lanes per bank.
template <int PW>
__device__ __forceinline__ uint32_t mapSwizzle(uint32_t y)
{
    if constexpr (PW == 1)
        return 4u * ((y >> 2) & 7u); // 4-pixel groups, rows 4 apart share a bank group (tile_pk_impl.h pkTransposeWrite)
    else
        return 2u * (y & 15u);       // 2-pixel groups: 16 rows x 2 banks cover the 64 banks once
}
template <int PW>
struct MapTile
{
    static constexpr uint32_t kRows = (PW == 1) ? 32u : 16u;     // rows of the workgroup's tile: 128-byte runs either way
    static constexpr uint32_t kWords = kRows * 256u * (uint32_t)PW; // 32 KiB
};

// one row of the lane's 4 pixels into the tile at tile row y
template <int PW>
__device__ __forceinline__ void mapTransposeWrite(unsigned * tile, uint32_t y, const unsigned (&px)[4][PW])
{
    const uint32_t x0 = 4u * threadIdx.x;
    if constexpr (PW == 1) {
        *reinterpret_cast<mu4 *>(tile + y * 256u + (x0 ^ mapSwizzle<1>(y))) = (mu4) { px[0][0], px[1][0], px[2][0], px[3][0] };
    } else {
        const uint32_t s = mapSwizzle<2>(y);
        *reinterpret_cast<mu4 *>(tile + 2u * (y * 256u + (x0 ^ s))) = (mu4) { px[0][0], px[0][1], px[1][0], px[1][1] };
        *reinterpret_cast<mu4 *>(tile + 2u * (y * 256u + ((x0 + 2u) ^ s))) = (mu4) { px[2][0], px[2][1], px[3][0], px[3][1] };
    }
}

// the workgroup's tile out, column by column: wave `wv` (0..3) takes 64 of the 256 columns.  `bandX0`: the tile's first column within the
// rectangle, `row0`: its first row within the rectangle; rows / columns beyond the rectangle (w4 x h2) hold nothing.
// PB: bytes per pixel -- 4 or 8 (one 16-byte store per lane and run piece), or 3 (the packed kernels' RGB8: pixels arrive as words, leave as
// three bytes each).  Everything that does not depend on the iteration is formed once per lane: which tile rows the lane gathers, where
// they land along the destination row, whether they exist; an iteration only advances the source column -- the first version recomputed
// all of it per iteration and spent more instructions on storing a quarter turn than on converting the pixels (61 against 36 per pixel for
// 10-bit -> RGBA10, profiles/r03_tail_pmc.txt).
template <int PW, int PB = 4 * PW, uint32_t ROWS = MapTile<PW>::kRows>
__device__ __forceinline__ void mapTransposeStore(const TileArgs & A, const unsigned * tile, uint32_t wv, uint32_t bandX0, uint32_t row0)
{
    constexpr uint32_t PPL = 4u / (uint32_t)PW; // pixels per lane and store: 16 bytes
    constexpr uint32_t RL = ROWS / PPL, RUNS = 64u / RL;                   // lanes per run, runs per store instruction: 8 and 8
    const PixelMap & m = A.map;
    const uint32_t l = threadIdx.x, t = l % RL;
    const bool fwd = m.sx > 0;
    // ---- per lane, once: its PPL tile rows in ascending destination order (backwards when the turn reverses them) ----
    uint32_t rowWord[PPL], swz[PPL], xDst[PPL];
    bool ok[PPL], all = true;
#pragma unroll
    for (uint32_t k = 0; k < PPL; ++k) {
        const uint32_t p = PPL * t + k, yr = fwd ? p : ROWS - 1u - p;
        const uint32_t jj = (uint32_t)A.mapY0 + row0 + yr - m.cy;
        rowWord[k] = yr * 256u, swz[k] = mapSwizzle<PW>(yr);
        xDst[k] = (uint32_t)(m.sx * (int32_t)jj + m.kx);
        ok[k] = row0 + yr < A.h2 && jj < m.ch;
        all = all && ok[k];
    }
    // ---- the lane's first source column; an iteration moves RUNS columns on: one destination row pitch times +-RUNS further ----
    uint32_t xs = wv * 64u + l / RL;
    const uint32_t iiBase = (uint32_t)A.mapX0 + bandX0 - m.cx; // (wraps when the tile starts left of the crop: ii is compared unsigned)
    // (signed: a lane whose first column lies left of the crop starts at a negative row and steps into the image)
    uint8_t * dstRow = A.rgb + (ptrdiff_t)(m.sy * (int32_t)(iiBase + xs) + m.ky) * (ptrdiff_t)A.rgbPitch;
    const ptrdiff_t rowStep = (ptrdiff_t)m.sy * (ptrdiff_t)RUNS * (ptrdiff_t)A.rgbPitch;
#pragma unroll
    for (uint32_t q = 0; q < 64u / RUNS; ++q, xs += RUNS, dstRow += rowStep) {
        if (bandX0 + xs >= A.w4 || iiBase + xs >= m.cw)
            continue;
        unsigned px[PPL][PW];
#pragma unroll
        for (uint32_t k = 0; k < PPL; ++k) {
            const uint32_t e = rowWord[k] + (xs ^ swz[k]);
            if constexpr (PW == 1) {
                px[k][0] = tile[e];
            } else {
                const mu2 v = *reinterpret_cast<const mu2 *>(tile + 2u * e);
                px[k][0] = v.x, px[k][1] = v.y;
            }
        }
        if (PB != 3 && all) {
            mu4 v;
            if constexpr (PW == 1)
                v = (mu4) { px[0][0], px[1][0], px[2][0], px[3][0] };
            else
                v = (mu4) { px[0][0], px[0][1], px[1][0], px[1][1] };
            *reinterpret_cast<mu4a4 *>(dstRow + (size_t)xDst[0] * PB) = v;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < PPL; ++k) {
                if (!ok[k])
                    continue;
                uint8_t * dst = dstRow + (size_t)xDst[k] * PB;
                if constexpr (PB == 3)
                    dst[0] = (uint8_t)px[k][0], dst[1] = (uint8_t)(px[k][0] >> 8), dst[2] = (uint8_t)(px[k][0] >> 16);
                else
                    mapStorePixel<PW>(dst, px[k]);
            }
        }
    }
}

} // namespace tile
} // namespace avifhip
