// tile_impl.h -- bandwidth-tuned YUV->RGB kernels for gfx950 (MI355X), instantiated by the kernels_tile_*.hip TUs.
//
// Scope: matrix-coefficient ("normal YUV") conversions into interleaved 3- or 4-channel RGB at 8-bit or 16-bit
// containers, from 8-bit or 16-bit-container 4:4:4 / 4:2:2 / 4:2:0 / 4:0:0 planes, nearest or bilinear chroma
// upsampling, alpha fill / copy / rescale and both flavours of alpha (un)premultiply -- every BASELINE
// configuration and what avifdec asks for -- plus half-float outputs, gray layouts (1 or 2 channels from the 4:0:0
// instantiations) and the identity matrix (round 2).  Everything else (the YCgCo family, destinations whose alpha
// bytes must stay untouched, unaligned user buffers, divisors off the verified list, the <= 3 columns and <= 1 row
// that do not fill a 4x2 pixel group) is served by kernels_generic.hip; RGB565 by the packed kernels (tile_pk_impl.h).
//
// Two structures share the arithmetic (computeTile).  Cooperative runs, round 1 (wave = 64 lanes, workgroup = 4 waves, one
// workgroup per tile of 256 x (8*NS) pixels, described below), kept for staged 16-bit planes and small frames; wave-private tiles
// (runSolo, the packed kernels' geometry: tile_geom.h) wherever an A/B run favoured them (kernels_tile.hip soloPays):
//   * wave w owns the NS vertically consecutive strips (256 x 2 pixels) w*NS .. w*NS+NS-1; lane tx owns 4 consecutive
//     pixels of both rows of each strip: one 4-sample vector load per plane and row (row-coalesced), one 16-byte
//     store per row for RGBA8 (1 KiB contiguous per wave instruction);
//   * every load of the tile is issued before the first result is needed; stores of earlier strips drain while later
//     strips are computed;
//   * bilinear chroma: the tile's chroma neighbourhood (4:2:0: 4*NS+2 rows x 136 samples per plane) is normalised to
//     fp32 once per sample and staged in LDS as interleaved (Cb,Cr) pairs; each lane reads its 4x3 neighbourhood
//     with six 16-byte LDS loads and filters both planes at once with packed fp32 instructions, sharing the
//     9/16, 3/16, 1/16 products between its pixels (the reference re-reads and re-normalises up to four chroma
//     samples for every output pixel);
//   * workgroups of one XCD (blockIdx % 8) take a contiguous run of tiles, so the chroma halo rows shared by
//     vertically adjacent tiles are re-read from that XCD's L2 rather than from HBM;
//   * arithmetic: the reference's operations in the reference's order, no contraction; divisions by plan constants
//     use the exhaustively verified fma(x, hi, x*lo) form (exactdiv.h); 8-bit outputs are quantised, clamped and
//     packed by v_cvt_pk_u8_f32 executed in round-toward-zero mode (it saturates to [0,255] and follows
//     MODE.FP_ROUND: tests/tools/probe_cvt_mode.hip);
//   * the kernel arguments are a distilled TileArgs (tile_shared.h) that stays in scalar registers; addresses are
//     uniform base + 32-bit lane offset.
#pragma once

#include <hip/hip_runtime.h>

#include "exactdiv.h"
#include "pixel_fixed.h"
#include "pixel_math.h"
#include "tile_geom.h"
#include "tile_map_impl.h"
#include "tile_shared.h"

// The seam-aware build of a family (-DTILE_SEAMS: kernels_tile_inst.hip, kernels_tile_fx_inst.hip) compiles these headers a second time, with
// the staging loads below reaching into the neighbouring tiles of a grid canvas (TileHalo).  Its kernels are batch kernels only and live in a
// namespace of their own, so that both builds link into one library.
#ifdef TILE_SEAMS
#define AVIFHIP_TILE_BUILD seams
#define AVIFHIP_SINGLE_LAUNCH(...) ((void)0) // (single images have no neighbours: not compiled)
#else
#define AVIFHIP_TILE_BUILD plain
#define AVIFHIP_SINGLE_LAUNCH(...) hipLaunchKernelGGL(__VA_ARGS__)
#endif

namespace avifhip {
namespace tile {
inline namespace AVIFHIP_TILE_BUILD {

#ifdef TILE_SEAMS
constexpr bool kSeams = true;
#else
constexpr bool kSeams = false;
#endif

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u2a2 __attribute__((ext_vector_type(2), aligned(2))); // four RGB565 pixels: rows of 16-bit pixels are 2-byte aligned, no more
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kBandW = 256; // pixels per strip row: 64 lanes x 4 pixels
constexpr int kLanesX = 64;
constexpr int kWavesPerBlock = 4;
// LDS rows of normalised chroma: entry c+5 holds the (Cb,Cr) pair of chroma column cxb + c, c in [-4, 131]
// (34 aligned 4-sample groups per row); a lane's four columns 2tx-1 .. 2tx+2 are entries 2tx+4 .. 2tx+7: two
// 16-byte aligned loads.
constexpr int kRowPitch = 140;
constexpr int kStageGroups = 34;
// chroma rows staged for a tile of WAVES * NS strips (2 * WAVES * NS luma rows) by WAVES waves: the four waves of a workgroup
// together (WAVES = 4, one barrier per tile), or every wave for itself (WAVES = 1: wave-private LDS, no barrier)
// A task = four consecutive samples of one row (a 4-byte / 8-byte load per plane).  The row needs 1 + 128 + 1 samples: WAVES = 4 takes them as
// 34 aligned groups (the band's 32 and one either side, of which one sample is used); a single wave would spend a whole extra round of its
// 64 lanes on those 2 x kRows side groups (6 rows x 34 = 204 tasks: a fourth round for 12 of them), so WAVES = 1 stages the band's 32 groups
// in rounds (kSplitHalo: 6 x 32 = 192 tasks, three rounds exactly) and the two side SAMPLES of every row by one lane each, once.
template <int SUB, int NS, int WAVES = 4>
struct StageRows
{
    static constexpr int kRows = (SUB == SUB_420) ? (WAVES * NS + 2) : (2 * WAVES * NS);
    static constexpr bool kSplitHalo = WAVES == 1;
    static constexpr int kGroups = kSplitHalo ? 32 : kStageGroups; // groups per row that go through the rounds
    static constexpr int kFirstGroup = kSplitHalo ? 1 : 0;         // ... the first of them, counted from the group left of the band
    static constexpr int kTasks = kRows * kGroups;
    static constexpr int kThreads = 64 * WAVES;
    static constexpr int kRounds = (kTasks + kThreads - 1) / kThreads;
    // Seam-aware build, the four waves together: the 64 tasks of a wave and round lie in at most three consecutive rows, and which tile a
    // ROW lies in -- the job's own, the one above or the one below -- is wave-uniform: loadNeighbourhood issues a wave's loads row by row,
    // each with a scalar base
    static constexpr bool kUniformRows = kSeams && WAVES == 4;
    static constexpr int kRowsPerWaveRound = (64 + kGroups - 2) / kGroups + 1;
    static_assert(!kSplitHalo || 2 * kRows <= 64, "one lane per side sample");
    // row and group (0 = the group left of the band) thread `t` of the workgroup stages in round `j`; false: none
    static __device__ __forceinline__ bool placeOf(int t, int j, int & row, int & grp)
    {
        const int task = t + kThreads * j;
        row = task / kGroups;
        grp = task - row * kGroups + kFirstGroup;
        return task < kTasks;
    }
};

__device__ __forceinline__ f2 splat(float v)
{
    return (f2) { v, v };
}
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}
// (cp - bias) / range for two samples, src/reformat.c:583,598 (verified reciprocal form)
__device__ __forceinline__ f2 norm2(f2 cp, float bias, RcpHL r)
{
    const f2 n = cp - splat(bias);
    return fma2(n, splat(r.hi), n * splat(r.lo));
}
__device__ __forceinline__ unsigned minU(unsigned a, unsigned b)
{
    return a < b ? a : b;
}
__device__ __forceinline__ int clampI(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

// Four consecutive samples as one vector load from a uniform base plus a 32-bit lane offset.
template <typename T>
struct Raw4
{
    unsigned w[sizeof(T) == 1 ? 1 : 2];
};
template <typename T>
__device__ __forceinline__ Raw4<T> load4(const uint8_t * base, uint32_t off)
{
    Raw4<T> r;
    if constexpr (sizeof(T) == 1) {
        r.w[0] = *reinterpret_cast<const uint32_t *>(base + off);
    } else {
        const u2 t = *reinterpret_cast<const u2 *>(base + off);
        r.w[0] = t.x;
        r.w[1] = t.y;
    }
    return r;
}
// ... as a streaming (non-temporal) load: planes that cannot stay in the Infinity Cache (batches: TileLaunch::streamLoads; tile_pk_impl.h
// pkLoad has the measurements and the reason for a template parameter)
template <typename T, bool STREAM>
__device__ __forceinline__ Raw4<T> load4s(const uint8_t * base, uint32_t off)
{
    if constexpr (!STREAM) {
        return load4<T>(base, off);
    } else {
        Raw4<T> r;
        if constexpr (sizeof(T) == 1) {
            r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + off));
        } else {
            const u2 t = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(base + off));
            r.w[0] = t.x;
            r.w[1] = t.y;
        }
        return r;
    }
}
template <typename T>
__device__ __forceinline__ unsigned load1(const uint8_t * base, uint32_t off)
{
    return (unsigned)*reinterpret_cast<const T *>(base + off);
}
template <typename T>
__device__ __forceinline__ void decode4(const Raw4<T> & r, unsigned v[4])
{
    if constexpr (sizeof(T) == 1) {
        v[0] = r.w[0] & 0xffu;
        v[1] = (r.w[0] >> 8) & 0xffu;
        v[2] = (r.w[0] >> 16) & 0xffu;
        v[3] = r.w[0] >> 24;
    } else {
        v[0] = r.w[0] & 0xffffu;
        v[1] = r.w[0] >> 16;
        v[2] = r.w[1] & 0xffffu;
        v[3] = r.w[1] >> 16;
    }
}
// ... as floats, clamped to the depth's maximum for 16-bit containers (src/reformat.c:712,821)
template <typename T>
__device__ __forceinline__ void samples4(const Raw4<T> & r, unsigned yuvMax, float f[4])
{
    unsigned v[4];
    decode4<T>(r, v);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        f[k] = (float)((sizeof(T) == 2) ? minU(v[k], yuvMax) : v[k]);
}

struct PixelOut
{
    unsigned r, g, b;
};

// a limited-range alpha sample as full range (TileArgs::AlphaLimited)
__device__ __forceinline__ unsigned alphaToFullRange(const TileArgs & A, unsigned v)
{
    const int n = ((int)v - A.alphaLim.lo) * A.alphaLim.full + A.alphaLim.half;
    const unsigned q = __umulhi((unsigned)max(n, 0), A.alphaLim.magic) >> A.alphaLim.shift; // a negative n truncates to <= 0 and clamps to 0
    return minU(q, (unsigned)A.alphaLim.full);
}

// alpha at the RGB depth from a plane sample: copy or depth rescale (src/alpha.c:84-103, verified reciprocal form)
__device__ __forceinline__ unsigned alphaRescaled(const TileArgs & A, unsigned sa)
{
    const float alphaF = divExact((float)sa, A.rcpYuvMax);
    const int dstAlpha = (int)(0.5f + (alphaF * A.rgbMaxF));
    return (unsigned)clampInt(dstAlpha, 0, (int)A.rgbMax);
}
__device__ __forceinline__ unsigned alphaFromPlane(const TileArgs & A, unsigned sa)
{
    return A.alphaRescale ? alphaRescaled(A, sa) : sa;
}

// ---- quantisation and alpha (un)premultiply: the reference's values by cheaper instruction sequences, each one enumerated
//      (tests/tools/verify_fp32_shortcuts.cpp, run by tests/test_exact_division.py) ----

// (T)(0.5f + clamp01(c) * max) without clamping first, the multiply and the add in ONE instruction.  Two facts:
//   * fmaf(c, max, 0.5f) truncates to the same integer as 0.5f + (c * max) for EVERY binary32 c and max in {255, 1023, 4095, 65535}
//     (the sum's ulp is never finer than the product's where an integer boundary could be crossed; enumerated over all 2^32 operands);
//   * t is monotonic in c, v_cvt_u32_f32 truncates toward zero like the C cast and returns 0 for every negative operand, and the min
//     restores the upper clamp (c >= 1 gives t >= max + 0.5f): identical to clamp-then-quantise for every finite c.
__device__ __forceinline__ float quantizeArg(float c, float maxf)
{
    return __builtin_fmaf(c, maxf, 0.5f);
}
__device__ __forceinline__ unsigned truncU32(float t)
{
    unsigned q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(t));
    return q;
}
__device__ __forceinline__ unsigned quantizeSat(float c, float maxf, unsigned maxv)
{
    return minU(truncU32(quantizeArg(c, maxf)), maxv);
}

// AVIF_CLAMP(x, 0.0f, 1.0f) (include/avif/internal.h:18) as the VOP3 clamp modifier of the instruction that produces x: the clamp
// costs nothing.  (A -0.0f may come out as +0.0f; every consumer below maps both to the same integer.)
__device__ __forceinline__ float sat01(float x)
{
    float r;
    asm("v_max_f32_e64 %0, %1, %1 clamp" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float addSat01(float a, float b)
{
    float r;
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float subSat01(float a, float b)
{
    float r;
    asm("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmaSat01(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// In-loop alpha of the reference's slow path (src/reformat.c:894-947) for the three clamped colours of one pixel and the pixel's
// normalised alpha Ac in [0, 1]; returns the quantiser's arguments t = c' * max + 0.5f (one rounding, see quantizeArg).
//   multiply:   Ac == 0 -> 0, Ac < 1 -> c * Ac, else c.  c * 0 is 0 and c * 1 is c: the product alone is all three cases.
//   unmultiply: Ac == 0 -> 0, Ac < 1 -> min(c / Ac, 1), else c.  The three IEEE divisions share their divisor, so its reciprocal is formed
//               once: v_rcp_f32 (1 ulp) and one Newton step give the correctly rounded r = RN(1 / Ac) for every Ac = a / max the alpha plane
//               can produce, and then q = fma(fma(-q0, Ac, c), r, q0), q0 = c * r, IS the correctly rounded quotient (Markstein's
//               correction) -- enumerated for every alpha code of the 8-, 10- and 12-bit planes against every c in {0} U [2^-40, 1]
//               (smaller positive c cannot arise: c is a sum of terms that are multiples of 2^-34), estimates off by up to 2 ulp.
//               The final fma carries the clamp (min(q, 1)); c / 1 comes out as c.  16-bit alpha planes keep the IEEE division.
struct InLoopAlpha
{
    float Ac, r;
    bool zero;
};
template <bool UNMUL, bool WIDE>
__device__ __forceinline__ InLoopAlpha inLoopAlpha(const TileArgs & A, unsigned unormA)
{
    InLoopAlpha L;
    const float af = (float)(WIDE ? minU(unormA, A.yuvMax) : unormA); // (8-bit samples cannot exceed the maximum)
    L.Ac = fmaSat01(af, A.rcpYuvMax.hi, af * A.rcpYuvMax.lo); // clamp01(unormA / yuvMax), src/reformat.c:896
    L.r = 0.0f, L.zero = false;
    if constexpr (UNMUL) {
        L.zero = L.Ac == 0.0f;
        const float d = L.zero ? 1.0f : L.Ac;
        const float r0 = __builtin_amdgcn_rcpf(d);
        L.r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
        L.Ac = d;
    }
    return L;
}
// 8-bit alpha planes: (divisor, reciprocal) of every alpha code in LDS, built by the workgroup when the kernel starts with inLoopAlpha's own
// instructions (buildUnmulTable) -- a pixel reads its pair instead of forming it (a conversion, the normalisation, a compare, a select,
// v_rcp_f32 -- a quarter-rate instruction -- and its Newton step less per pixel).  The reciprocal of alpha 0 is stored as 0: the quotient
// then comes out as 0 (q0 = c * 0, fma(c, 0, 0)) without the select per channel.
struct UnmulEntry
{
    float d, r;
};
__device__ __forceinline__ UnmulEntry * unmulTable()
{
    __shared__ UnmulEntry table[256];
    return table;
}
// (every thread of the workgroup, before any wave leaves the kernel)
__device__ __forceinline__ void buildUnmulTable(const TileArgs & A)
{
    const unsigned code = threadIdx.y * blockDim.x + threadIdx.x; // workgroups are 64 x 4
    const InLoopAlpha L = inLoopAlpha<true, false>(A, code);
    unmulTable()[code & 255u] = { L.Ac, L.zero ? 0.0f : L.r };
    __syncthreads();
}
// quantiser arguments of a pixel's (first, third) colours and of green, un-multiplied by the alpha code `a`
__device__ __forceinline__ void unmulFromTable(const TileArgs & A, unsigned a, float X, float G, float Z, f2 & tbr, float & tg)
{
    const UnmulEntry e = unmulTable()[a];
    const f2 c = { X, Z }, d = splat(e.d), r = splat(e.r);
    const f2 q0 = c * r;
    const f2 er = __builtin_elementwise_fma(-q0, d, c);
    const f2 q = { fmaSat01(er.x, e.r, q0.x), fmaSat01(er.y, e.r, q0.y) };
    tbr = __builtin_elementwise_fma(q, splat(A.rgbMaxF), splat(0.5f));
    const float g0 = G * e.r;
    tg = quantizeArg(fmaSat01(__builtin_fmaf(-g0, e.d, G), e.r, g0), A.rgbMaxF);
}
// first thing in a kernel whose pixels may be un-multiplied from an 8-bit alpha plane (MULSEL: computeTile's; 0 = the job says)
template <typename YT, bool HASMUL, int MULSEL>
__device__ __forceinline__ void prepareAlphaTables(const TileArgs & A)
{
    if constexpr (HASMUL && sizeof(YT) == 1 && (MULSEL == 0 || MULSEL == 2)) {
        if (MULSEL == 2 || A.inLoopMul == MUL_UNMULTIPLY) // wave-uniform (workgroup-uniform: one job per workgroup)
            buildUnmulTable(A);
    }
}
template <bool UNMUL>
__device__ __forceinline__ float inLoopChannel(const TileArgs & A, float c, const InLoopAlpha & L)
{
    if constexpr (!UNMUL) {
        return quantizeArg(c * L.Ac, A.rgbMaxF);
    } else {
        const float q0 = c * L.r;
        const float q = fmaSat01(__builtin_fmaf(-q0, L.Ac, c), L.r, q0);
        return quantizeArg(L.zero ? 0.0f : q, A.rgbMaxF);
    }
}
// ... with the IEEE division (alpha planes deeper than 12 bits)
__device__ __forceinline__ float inLoopChannelIeee(const TileArgs & A, float c, float Ac)
{
    return quantizeArg(applyAlphaF(c, Ac, MUL_UNMULTIPLY), A.rgbMaxF);
}

// The integer post-pass of the reference's fast paths (src/alpha.c:180-192, :367-381) on one pixel's quantised colours.
//   multiply:   floorf(c * a / maxF + 0.5f) with "/ maxF" in the verified reciprocal form; a == 0 needs no special case (0 / maxF + 0.5f
//               truncates to 0); a >= max leaves the pixel untouched (:180-183).
//   unmultiply: min(floorf(c * maxF / a + 0.5f), maxF) with the pixel's exact reciprocal formed once (pixel_math.h unpremulRcp: every
//               depth); a == 0 answers 0, a >= max leaves the channel alone (:367-373).
struct PostAlpha
{
    unsigned a;
    float af;
    UnpremulRcp rcp;
};
template <bool UNMUL>
__device__ __forceinline__ PostAlpha postAlpha(const TileArgs & A, unsigned a)
{
    PostAlpha P;
    P.a = a, P.af = (float)a, P.rcp = { 0.0f, 0.0f };
    if constexpr (UNMUL)
        P.rcp = unpremulRcp(a ? P.af : 1.0f);
    return P;
}
template <bool UNMUL>
__device__ __forceinline__ unsigned postChannel(const TileArgs & A, unsigned c, const PostAlpha & P)
{
    if constexpr (!UNMUL) {
        const unsigned m = truncU32(divExact((float)c * P.af, A.rcpRgbMax) + 0.5f);
        return (P.a >= A.rgbMax) ? c : m;
    } else {
        const unsigned q = minU(truncU32(unpremulRcpArg((float)c, P.rcp, A.rgbMaxF)), A.rgbMax);
        return (P.a >= A.rgbMax) ? c : (P.a == 0u ? 0u : q);
    }
}

// RGB output is written once and never read back by the kernel: non-temporal stores keep 130+ MB of it from
// displacing the input planes in L2 / Infinity Cache (tests/tools/membw2.hip: 33 -> 26.5 us for cfg2's byte
// movement when frames stream from HBM).
template <typename V>
__device__ __forceinline__ void storeVec(uint8_t * base, uint32_t off, const V & v, bool)
{
    V * dst = reinterpret_cast<V *>(base + off);
#ifdef AVIFHIP_PLAIN_STORES
    *dst = v;
#else
    __builtin_nontemporal_store(v, dst);
#endif
}

// Store 4 consecutive pixels.  swapRB: B is the first colour channel; alphaFirst: A precedes colour.
template <typename RT, int NCH>
__device__ __forceinline__ void store4(uint8_t * base, uint32_t off, const PixelOut q[4], const unsigned a[4], bool swapRB, bool alphaFirst, bool nt)
{
    unsigned x[4], z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = swapRB ? q[k].b : q[k].r;
        z[k] = swapRB ? q[k].r : q[k].b;
    }
    if constexpr (sizeof(RT) == 1 && NCH == 4) {
        u4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = alphaFirst ? (a[k] | (x[k] << 8) | (q[k].g << 16) | (z[k] << 24)) : (x[k] | (q[k].g << 8) | (z[k] << 16) | (a[k] << 24));
        storeVec(base, off, w, nt);
    } else if constexpr (sizeof(RT) == 1 && NCH == 3) {
        // 12 bytes: x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3 (rows and pixel groups are 4-byte aligned)
        storeVec(base, off, x[0] | (q[0].g << 8) | (z[0] << 16) | (x[1] << 24), nt);
        storeVec(base, off + 4, q[1].g | (z[1] << 8) | (x[2] << 16) | (q[2].g << 24), nt);
        storeVec(base, off + 8, z[2] | (x[3] << 8) | (q[3].g << 16) | (z[3] << 24), nt);
    } else if constexpr (sizeof(RT) == 2 && NCH == 4) {
        // never reached: 16-bit RGBA rows go through store4WideRgba (contiguous store instructions)
        static_assert(sizeof(RT) != 2 || NCH != 4, "use store4WideRgba");
    } else { // 16-bit, 3 channels: 24 bytes = 3 x 8
        storeVec(base, off, (u2) { x[0] | (q[0].g << 16), z[0] | (x[1] << 16) }, nt);
        storeVec(base, off + 8, (u2) { q[1].g | (z[1] << 16), x[2] | (q[2].g << 16) }, nt);
        storeVec(base, off + 16, (u2) { z[2] | (x[3] << 16), q[3].g | (z[3] << 16) }, nt);
    }
}

// Gray layouts (GRAY, GRAYA, AGRAY): 4, 8 or 16 bytes per lane, one store.  The value is the pixel's G channel: with no chroma every
// colour channel equals clamp01(Y) and goes through the same alpha multiply and quantiser as the reference's `grayc`
// (src/reformat.c:886-892, :894-947, :952-961).
template <typename RT, int NCH>
__device__ __forceinline__ void storeGray4(uint8_t * base, uint32_t off, const PixelOut q[4], const unsigned a[4], bool alphaFirst, bool nt)
{
    if constexpr (sizeof(RT) == 1 && NCH == 1) {
        storeVec(base, off, q[0].g | (q[1].g << 8) | (q[2].g << 16) | (q[3].g << 24), nt);
    } else if constexpr (sizeof(RT) == 1) {
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p[k] = alphaFirst ? (a[k] | (q[k].g << 8)) : (q[k].g | (a[k] << 8));
        storeVec(base, off, (u2) { p[0] | (p[1] << 16), p[2] | (p[3] << 16) }, nt);
    } else if constexpr (NCH == 1) {
        storeVec(base, off, (u2) { q[0].g | (q[1].g << 16), q[2].g | (q[3].g << 16) }, nt);
    } else {
        u4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = alphaFirst ? (a[k] | (q[k].g << 16)) : (q[k].g | (a[k] << 16));
        storeVec(base, off, w, nt);
    }
}

// 16-bit RGBA: a lane's 4 pixels are 32 bytes.  Two 16-byte stores at (32*lane, 32*lane + 16) make every store
// instruction touch only half of each cache line (tests/tools/membw3.hip: cfg3's bytes take 111 us that way and 74 us
// with contiguous instructions), so the wave first re-distributes its 2 KiB row segment through a wave-private LDS
// buffer: written in lane order (32 bytes per lane), read back so that store instruction h covers bytes
// [1024*h, 1024*h + 1024) with lane l writing the 16 bytes at 16*l.  Must be called by every lane of the wave.
// q[k].r / q[k].b hold the first / third colour channel; rowOff addresses the band's first pixel.
struct WideRowExchange
{
    u4 w[128]; // 2 KiB: one row segment of one wave
};

__device__ __forceinline__ void store4WideRgba(uint8_t * base, uint32_t rowOff, const PixelOut q[4], const unsigned a[4], bool alphaFirst, uint32_t bandX,
                                               uint32_t w4, WideRowExchange & xchg)
{
    u4 w0, w1;
    if (alphaFirst) { // uniform: A X G Z
        w0 = (u4) { a[0] | (q[0].r << 16), q[0].g | (q[0].b << 16), a[1] | (q[1].r << 16), q[1].g | (q[1].b << 16) };
        w1 = (u4) { a[2] | (q[2].r << 16), q[2].g | (q[2].b << 16), a[3] | (q[3].r << 16), q[3].g | (q[3].b << 16) };
    } else { // X G Z A
        w0 = (u4) { q[0].r | (q[0].g << 16), q[0].b | (a[0] << 16), q[1].r | (q[1].g << 16), q[1].b | (a[1] << 16) };
        w1 = (u4) { q[2].r | (q[2].g << 16), q[2].b | (a[2] << 16), q[3].r | (q[3].g << 16), q[3].b | (a[3] << 16) };
    }
    const int l = threadIdx.x;
    // the buffer is private to this wave, whose LDS accesses execute in program order; the fences only keep the compiler
    // from moving the reads above the writes (other lanes' writes) or the next row's writes above these reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    xchg.w[2 * l] = w0;
    xchg.w[2 * l + 1] = w1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u4 s0 = xchg.w[l], s1 = xchg.w[64 + l];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // slot l of instruction h holds the pixels of lane 32*h + (l >> 1)
    if (bandX + 4u * (uint32_t)(l >> 1) < w4)
        __builtin_nontemporal_store(s0, reinterpret_cast<u4 *>(base + rowOff + 16u * (uint32_t)l));
    if (bandX + 4u * (uint32_t)(32 + (l >> 1)) < w4)
        __builtin_nontemporal_store(s1, reinterpret_cast<u4 *>(base + rowOff + 1024u + 16u * (uint32_t)l));
}

// 3-channel pixels: a lane's 4 pixels are NW = 3 (8-bit) or 6 (16-bit) dwords, and dword stores at a stride of 12 / 24 bytes make each of
// the NW store instructions touch a third of every cache line of the row segment (8K identity copies: 49 us that way, against 34.5 us
// for the 4-byte pixels' single 16-byte store per lane).  Same cure as for 16-bit RGBA: the wave re-distributes its segment (768 or 1536
// bytes) through its private LDS buffer so that a store instruction writes 16 bytes per lane at consecutive addresses.  Must be called by
// every lane of the wave; `validBytes` (a multiple of 4) is how much of the segment exists (bands cut by the right edge).
typedef unsigned u4a4 __attribute__((ext_vector_type(4), aligned(4))); // rows of 3-channel pixels are only dword-aligned

template <int NW>
__device__ __forceinline__ void storeRowContiguous(uint8_t * base, uint32_t rowOff, const unsigned (&w)[NW], uint32_t validBytes, WideRowExchange & xchg)
{
    unsigned * words = reinterpret_cast<unsigned *>(xchg.w);
    const uint32_t l = threadIdx.x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int k = 0; k < NW; ++k)
        words[NW * l + k] = w[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr uint32_t kChunks = 16 * NW; // 16-byte pieces of the segment
    constexpr int kInstr = (kChunks + 63) / 64;
    u4 s[kInstr];
#pragma unroll
    for (int h = 0; h < kInstr; ++h)
        s[h] = xchg.w[(64u * h + l) < kChunks ? 64u * h + l : 0u];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < kInstr; ++h) {
        const uint32_t chunk = 64u * h + l, byte = 16u * chunk;
        if (chunk >= kChunks || byte >= validBytes)
            continue;
        uint8_t * dst = base + rowOff + byte;
        if (byte + 16u <= validBytes) {
            __builtin_nontemporal_store(s[h], reinterpret_cast<u4a4 *>(dst));
        } else { // the segment ends inside this piece
#pragma unroll
            for (uint32_t d = 0; d < 3; ++d)
                if (byte + 4u * d < validBytes)
                    __builtin_nontemporal_store(s[h][d], reinterpret_cast<unsigned *>(dst + 4u * d));
        }
    }
}

// ... of four finished pixels (q[k].r / q[k].b: first / third colour channel)
template <typename RT>
__device__ __forceinline__ void store4Rgb3(uint8_t * base, uint32_t rowOff, const PixelOut q[4], uint32_t validBytes, WideRowExchange & xchg)
{
    if constexpr (sizeof(RT) == 1) {
        const unsigned w[3] = { q[0].r | (q[0].g << 8) | (q[0].b << 16) | (q[1].r << 24), q[1].g | (q[1].b << 8) | (q[2].r << 16) | (q[2].g << 24),
                                q[2].b | (q[3].r << 8) | (q[3].g << 16) | (q[3].b << 24) };
        storeRowContiguous<3>(base, rowOff, w, validBytes, xchg);
    } else {
        const unsigned w[6] = { q[0].r | (q[0].g << 16), q[0].b | (q[1].r << 16), q[1].g | (q[1].b << 16),
                                q[2].r | (q[2].g << 16), q[2].b | (q[3].r << 16), q[3].g | (q[3].b << 16) };
        storeRowContiguous<6>(base, rowOff, w, validBytes, xchg);
    }
}

// 8-bit RGBA family: (uint8_t)(0.5f + clamp01(c) * 255) for the three colour channels of four pixels, inserted into
// copies of words that hold the alpha byte.  The inputs are t = 0.5f + c * 255 (unclamped c); v_cvt_pk_u8_f32 in
// round-toward-zero mode truncates like the C cast and saturates to [0, 255], and because t is monotonic in c the
// saturation selects the same byte as clamping c first.  The rounding mode is changed only inside this block.
__device__ __forceinline__ void packRgba8Row(unsigned w[4], const unsigned aw[4], const f2 br[4], const float g[4], unsigned slotR, unsigned slotG, unsigned slotB)
{
    // operands: %4/%5 = (B,R) of pixel 0 ... %10/%11 of pixel 3; %12..%15 = G; %16..%19 = alpha words; %20,%21,%22 = slots R,G,B
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, %22, %16\n\t"
                 "v_cvt_pk_u8_f32 %1, %6, %22, %17\n\t"
                 "v_cvt_pk_u8_f32 %2, %8, %22, %18\n\t"
                 "v_cvt_pk_u8_f32 %3, %10, %22, %19\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, %20, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, %20, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %9, %20, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %11, %20, %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %12, %21, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %13, %21, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, %21, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %15, %21, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(br[0].x), "v"(br[0].y), "v"(br[1].x), "v"(br[1].y), "v"(br[2].x), "v"(br[2].y), "v"(br[3].x), "v"(br[3].y), "v"(g[0]),
                   "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(aw[0]), "v"(aw[1]), "v"(aw[2]), "v"(aw[3]), "s"(slotR), "s"(slotG), "s"(slotB));
}

// 8-bit RGB / BGR: 12 bytes x0 g0 z0 x1 | g1 z1 x2 g2 | z2 x3 g3 z3 from t = 0.5f + c * 255 (see packRgba8Row)
__device__ __forceinline__ void packRgb8Row(unsigned w[3], const float x[4], const float g[4], const float z[4])
{
    // operand map: %3..%6 = x0..x3, %7..%10 = g0..g3, %11..%14 = z0..z3
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %12, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %6, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %11, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %5, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %10, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(z[0]), "v"(z[1]), "v"(z[2]),
                   "v"(z[3]));
}

// a + b of two halves of one register pair, as ONE scalar v_add_f32: written in C the two sums of neighbouring pixels are paired into a
// v_pk_add_f32 behind three v_mov_b32 that bring the halves side by side (12 cycles per pixel pair against 5, profiles/r03_valu_rate.txt)
__device__ __forceinline__ float addScalar(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// YCgCo family, one pixel: (first colour, green, third colour) unclamped from normalised luma, the two normalised chroma samples in plane
// order (`u` plane, `v` plane of TileArgs) and the luma code (src/reformat.c:853-871).
__device__ __forceinline__ void ycgcoPixel(const TileArgs & A, float Y, f2 uvp, unsigned unormY, float & X, float & G, float & Z)
{
    const float cg = A.cgFirst ? uvp.x : uvp.y, co = A.cgFirst ? uvp.y : uvp.x;
    float R, B;
    if (A.ycgco == 1) {
        const float t = Y - cg;
        G = Y + cg, B = t - co, R = t + co;
    } else { // YCgCo-Re / -Ro: lifting on integers, then "/ maxF" in the verified reciprocal form
        const int Cg = (int)floorf((cg * A.yuvMaxF) + 0.5f), Co = (int)floorf((co * A.yuvMaxF) + 0.5f);
        const int t = (int)unormY - (Cg >> 1);
        const int gi = clampI(t + Cg, 0, (int)A.rgbMax), bi = clampI(t - (Co >> 1), 0, (int)A.rgbMax), ri = clampI(bi + Co, 0, (int)A.rgbMax);
        G = divExact((float)gi, A.rcpRgbMax), B = divExact((float)bi, A.rcpRgbMax), R = divExact((float)ri, A.rcpRgbMax);
    }
    X = A.cgFirst ? B : R, Z = A.cgFirst ? R : B; // cgFirst <=> blue is the first colour channel
}

// ---- where a staging lane finds a chroma sample.  HALO = false: coordinates clamp into the job's chroma window (the whole plane of the
//      canvas unless the canvas is a grid of separately stored tiles) -- exactly the reference's border rule (src/reformat.c:768,784): the
//      neighbour of an edge sample is the sample itself, which is also libyuv's ((3a + a + 2) >> 2 == a).  HALO = true (seam-aware builds,
//      tiles whose neighbourhood crosses a seam): one sample beyond the window on every side with a neighbouring tile, read from that
//      tile's plane (TileHalo) ----
// the neighbours' planes of the wave's job, in LDS (seam-aware builds: jobOf puts them there)
__device__ __forceinline__ TileHalo::Planes * haloOfWave()
{
    __shared__ __attribute__((aligned(16))) TileHalo::Planes held[4][9]; // (workgroups are 64 x 4: one copy per wave, no workgroup barrier needed)
    return held[threadIdx.y];
}

// the private copy of a batch kernel's job (read through the table pointer, every field would be re-loaded after each store: the compiler
// cannot rule out that the RGB stores alias the table -- ~20 scalar loads per tile inside the pipelined loop)
template <bool COOPERATIVE = false>
__device__ __forceinline__ TileArgs jobOf(const TileArgs * __restrict__ table, uint32_t jobIndex = blockIdx.z)
{
    // (the copy first: behind the fences below the table's fields would no longer qualify for scalar loads, and the job would live in
    //  vector registers -- 140 instead of 76 in the cooperative 10-bit kernel)
    TileArgs job = table[jobIndex];
    if constexpr (kSeams && COOPERATIVE) {
        job.haloRef = &table[jobIndex].halo; // the four waves together stage whole rows: scalar loads from the table where a row needs them
    } else if constexpr (kSeams) {
        TileHalo::Planes * mine = haloOfWave();
        if (threadIdx.x < 9)
            mine[threadIdx.x] = table[jobIndex].halo.at[threadIdx.x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return job;
}

// Wave-uniform: does the chroma neighbourhood of a tile -- rows cyFirst..cyLast, columns cxFirst..cxLast -- cross a seam of the canvas?
// (Tiles away from the seams, most of a large canvas, take the plain loads: scalar base, 32-bit lane offset.)
__device__ __forceinline__ bool haloNeeded(const TileArgs & A, int cyFirst, int cyLast, int cxFirst, int cxLast)
{
    if constexpr (!kSeams)
        return false;
    const uint32_t s = A.haloSides;
    return ((s & HALO_ABOVE) && cyFirst < A.cyMin) || ((s & HALO_BELOW) && cyLast > A.cyMax) || ((s & HALO_LEFT) && cxFirst < A.cxMin) ||
           ((s & HALO_RIGHT) && cxLast > A.cxMax);
}

struct HaloRow
{
    int cy;               // canvas chroma row to read
    int index;            // 0: in the job's own tile, 3: in the tile above, 6: below (TileHalo)
    const uint8_t *u, *v; // the planes holding the row's samples inside the window's columns
};
template <bool HALO>
__device__ __forceinline__ HaloRow haloRow(const TileArgs & A, int cyRaw)
{
    HaloRow r;
    if constexpr (!HALO) {
        r.cy = clampI(cyRaw, A.cyMin, A.cyMax);
        r.index = 0;
        r.u = A.u, r.v = A.v;
    } else {
        r.cy = clampI(cyRaw, A.cyMin - ((A.haloSides & HALO_ABOVE) ? 1 : 0), A.cyMax + ((A.haloSides & HALO_BELOW) ? 1 : 0));
        r.index = r.cy < A.cyMin ? 3 : (r.cy > A.cyMax ? 6 : 0);
        const TileHalo::Planes p = haloOfWave()[r.index];
        r.u = p.u, r.v = p.v;
    }
    return r;
}
// column `cxRaw` of that row: the column to read and the planes holding it
template <bool HALO>
__device__ __forceinline__ uint32_t haloColumn(const TileArgs & A, const HaloRow & r, int cxRaw, const uint8_t *& pu, const uint8_t *& pv)
{
    if constexpr (!HALO) {
        pu = A.u, pv = A.v;
        return (uint32_t)clampI(cxRaw, A.cxMin, A.cxMax);
    } else {
        const int cx = clampI(cxRaw, A.cxMin - ((A.haloSides & HALO_LEFT) ? 1 : 0), A.cxMax + ((A.haloSides & HALO_RIGHT) ? 1 : 0));
        const TileHalo::Planes p = haloOfWave()[r.index + (cx < A.cxMin ? 1 : (cx > A.cxMax ? 2 : 0))];
        pu = p.u, pv = p.v;
        return (uint32_t)cx;
    }
}

// Raw (undecoded) data of one strip (256 pixels x 2 rows) as loaded by one lane.
template <typename YT, int SUB, bool BIL, bool NEEDA>
struct StripRaw
{
    static constexpr bool kOwnChroma = SUB == SUB_444 || ((SUB == SUB_420 || SUB == SUB_422) && !BIL);
    Raw4<YT> y[2];
    Raw4<YT> a[NEEDA ? 2 : 1];
    // 4:4:4: one vector per row | nearest 4:2:x: w[0] = the lane's two chroma samples of the row
    Raw4<YT> u[kOwnChroma ? 2 : 1];
    Raw4<YT> v[kOwnChroma ? 2 : 1];
};

// Raw (undecoded) data of one tile as loaded by one lane: its share of the chroma neighbourhood to stage, and the
// luma / alpha / co-sited chroma of its own strips.  Lives in registers while the previous tile is computed.
template <typename YT, int SUB, bool BIL, bool NEEDA, int NS, int WAVES = 4>
struct TileRaw
{
    Raw4<YT> su[BIL ? StageRows<SUB, NS, WAVES>::kRounds : 1], sv[BIL ? StageRows<SUB, NS, WAVES>::kRounds : 1];
    unsigned hu, hv; // StageRows::kSplitHalo: the side sample this lane brings (lane 2 * row + side, side 0 = left of the band)
    StripRaw<YT, SUB, BIL, NEEDA> raw[NS];
};

// Per-lane constants of the band a workgroup walks down.
struct BandCtx
{
    uint32_t bandX; // first pixel column of the band (wave row segment), relative to the rectangle
    uint32_t X;     // first pixel column of this lane, relative to the rectangle
    uint32_t Xc;    // ... clamped into the rectangle for loads
    bool laneValid; // the lane's 4-pixel group exists (w4 is a multiple of 4: groups are whole or absent)
    int cxb;        // canvas chroma column of the band's first sample
};

// this lane's share of a tile's chroma neighbourhood; LDS row 0 holds canvas chroma row `rowBase`
template <typename YT, int SUB, bool NEEDA, int NS, int WAVES, bool HALO>
__device__ __forceinline__ void loadNeighbourhood(const TileArgs & A, const BandCtx & c, int rowBase, TileRaw<YT, SUB, true, NEEDA, NS, WAVES> & T)
{
    constexpr bool kWide = sizeof(YT) == 2;
    constexpr uint32_t BPS = sizeof(YT);
    typedef StageRows<SUB, NS, WAVES> SR;
    const int tx = threadIdx.x, wv = (WAVES == 1) ? 0 : (int)threadIdx.y;
    if constexpr (SR::kUniformRows) {
        // Seam-aware build, the four waves together.  Which tile a ROW lies in is wave-uniform, so the row's planes come from the table by
        // scalar loads (constant address space) and its loads keep the plain build's form, scalar base + 32-bit lane offset -- per-lane
        // pointers cost this kernel ten vector registers and a step of occupancy (profiles/README.md).  The rows of the job's own tile go in
        // one pass, the row just above or below its window -- at most one each, and only in the tiles along the seam -- in a pass of its own
        // with the neighbour's planes.  Groups cut by the window's left or right border (the tiles along the job's sides only) fetch their
        // two pairs of planes from the table per lane.
        typedef const __attribute__((address_space(4))) TileHalo * HaloTable;
        typedef const __attribute__((address_space(1))) uint8_t * GlobalPlane;
        typedef const __attribute__((address_space(1))) TileHalo * HaloTableG;
        const HaloTable HT = (HaloTable)A.haloRef;
        const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
        const int above = (A.haloSides & HALO_ABOVE) ? 1 : 0, below = (A.haloSides & HALO_BELOW) ? 1 : 0;
        const int left = (A.haloSides & HALO_LEFT) ? 1 : 0, right = (A.haloSides & HALO_RIGHT) ? 1 : 0;
        const int t = w * kLanesX + tx;
#pragma unroll
        for (int j = 0; j < SR::kRounds; ++j) {
            int myRow, grp;
            const bool mine = SR::placeOf(t, j, myRow, grp);
            const int firstRow = (w * kLanesX + SR::kThreads * j) / SR::kGroups; // wave-uniform
            const int cxa = c.cxb - 4 + 4 * grp;
            const bool whole = cxa >= A.cxMin && cxa + 3 <= A.cxMax;
            T.su[j].w[0] = T.sv[j].w[0] = 0;
            if constexpr (kWide)
                T.su[j].w[1] = T.sv[j].w[1] = 0;
            // first the rows of the job's own tile -- every row of most tiles: ONE load per plane, exactly the plain build's (a first version
            // went row by row for every tile: three masked loads per plane and wave, cfg5's canvas 8 % behind the plain kernel) ...
            const int cyRaw = rowBase + myRow;
            const bool foreign = (above && cyRaw < A.cyMin) || (below && cyRaw > A.cyMax);
            if (mine && whole && !foreign) {
                const int cy = clampI(cyRaw, A.cyMin, A.cyMax);
                T.su[j] = load4<YT>(A.u, (uint32_t)cy * A.uPitch + (uint32_t)cxa * BPS);
                T.sv[j] = load4<YT>(A.v, (uint32_t)cy * A.vPitch + (uint32_t)cxa * BPS);
            }
            // ... then the row just above / below the window, where this wave stages it (wave-uniform: the rows of a wave and round are
            // lastRow - firstRow + 1 <= kRowsPerWaveRound consecutive ones)
            const int lastRow = (w * kLanesX + SR::kThreads * j + kLanesX - 1) / SR::kGroups;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int cy = side ? A.cyMax + 1 : A.cyMin - 1; // the neighbour's row the filter reaches (rows beyond it belong to absent strips)
                const int row = cy - rowBase;
                if (!(side ? below : above) || row < firstRow || row > lastRow || row >= SR::kRows)
                    continue;
                GlobalPlane ru = (GlobalPlane)HT->at[side ? 6 : 3].u, rv = (GlobalPlane)HT->at[side ? 6 : 3].v;
                // (the row's planes pinned in scalar registers, and a fence for the compiler: left alone it merges the passes into ONE load per
                //  lane through a selected 64-bit pointer -- flat loads and 300 selects, the very registers this form is here to save)
                asm volatile("" : "+s"(ru), "+s"(rv) : : "memory");
                if (mine && whole && myRow == row) {
                    T.su[j] = load4<YT>((const uint8_t *)ru, (uint32_t)cy * A.uPitch + (uint32_t)cxa * BPS);
                    T.sv[j] = load4<YT>((const uint8_t *)rv, (uint32_t)cy * A.vPitch + (uint32_t)cxa * BPS);
                }
            }
            if (mine && !whole) {
                // group cut by the left or right border of the window: samples inside it from the row's tile, samples beyond it from that
                // tile's left / right neighbour (or clamped, where the canvas ends)
                const int cy = clampI(rowBase + myRow, A.cyMin - above, A.cyMax + below);
                const int vi = cy < A.cyMin ? 3 : (cy > A.cyMax ? 6 : 0);
                const HaloTableG G = (HaloTableG)A.haloRef;
                const bool leftCut = cxa < A.cxMin;
                const int si = vi + (leftCut ? left : 2 * right); // (no neighbour on that side: the row's own tile, coordinates clamped)
                const uint8_t *rowU = G->at[vi].u, *rowV = G->at[vi].v, *sideU = G->at[si].u, *sideV = G->at[si].v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cx = clampI(cxa + k, A.cxMin - left, A.cxMax + right);
                    const bool beyond = cx < A.cxMin || cx > A.cxMax;
                    const GlobalPlane pu = (GlobalPlane)(beyond ? sideU : rowU), pv = (GlobalPlane)(beyond ? sideV : rowV);
                    const unsigned u = load1<YT>((const uint8_t *)pu, (uint32_t)cy * A.uPitch + (uint32_t)cx * BPS);
                    const unsigned v = load1<YT>((const uint8_t *)pv, (uint32_t)cy * A.vPitch + (uint32_t)cx * BPS);
                    if constexpr (!kWide) {
                        T.su[j].w[0] |= u << (8 * k);
                        T.sv[j].w[0] |= v << (8 * k);
                    } else {
                        T.su[j].w[k >> 1] |= u << (16 * (k & 1));
                        T.sv[j].w[k >> 1] |= v << (16 * (k & 1));
                    }
                }
            }
        }
    } else {
        const int t = wv * kLanesX + tx;
#pragma unroll
        for (int j = 0; j < SR::kRounds; ++j) {
            int row, grp;
            if (SR::placeOf(t, j, row, grp)) {
                // coordinates clamp to the job's chroma window (the whole plane of the canvas unless the canvas is a grid of
                // separately stored tiles): exactly the reference's border rule (src/reformat.c:768,784) -- the neighbour
                // of an edge sample is the sample itself
                const HaloRow hr = haloRow<HALO>(A, rowBase + row);
                const int cy = hr.cy;
                const int cxa = c.cxb - 4 + 4 * grp;
                if (cxa >= A.cxMin && cxa + 3 <= A.cxMax) {
                    T.su[j] = load4<YT>(hr.u, (uint32_t)cy * A.uPitch + (uint32_t)cxa * BPS);
                    T.sv[j] = load4<YT>(hr.v, (uint32_t)cy * A.vPitch + (uint32_t)cxa * BPS);
                } else {
                    // group cut by the left or right border of the window
                    T.su[j].w[0] = T.sv[j].w[0] = 0;
                    if constexpr (kWide)
                        T.su[j].w[1] = T.sv[j].w[1] = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint8_t *pu, *pv;
                        const uint32_t cx = haloColumn<HALO>(A, hr, cxa + k, pu, pv);
                        const unsigned u = load1<YT>(pu, (uint32_t)cy * A.uPitch + cx * BPS);
                        const unsigned v = load1<YT>(pv, (uint32_t)cy * A.vPitch + cx * BPS);
                        if constexpr (!kWide) {
                            T.su[j].w[0] |= u << (8 * k);
                            T.sv[j].w[0] |= v << (8 * k);
                        } else {
                            T.su[j].w[k >> 1] |= u << (16 * (k & 1));
                            T.sv[j].w[k >> 1] |= v << (16 * (k & 1));
                        }
                    }
                }
            }
        }
    }
    if constexpr (SR::kSplitHalo) {
        T.hu = T.hv = 0;
        if (tx < 2 * SR::kRows) {
            const HaloRow hr = haloRow<HALO>(A, rowBase + (tx >> 1));
            const uint8_t *pu, *pv;
            const uint32_t cx = haloColumn<HALO>(A, hr, (tx & 1) ? c.cxb + 128 : c.cxb - 1, pu, pv);
            T.hu = load1<YT>(pu, (uint32_t)hr.cy * A.uPitch + cx * BPS);
            T.hv = load1<YT>(pv, (uint32_t)hr.cy * A.vPitch + cx * BPS);
        }
    }
}

// issue every load of the tile whose first luma row (relative to the rectangle) is tileY
template <typename YT, int SUB, bool BIL, bool NEEDA, int NS, int WAVES = 4, bool STREAM = false>
__device__ __forceinline__ void loadTile(const TileArgs & A, const BandCtx & c, uint32_t tileY, TileRaw<YT, SUB, BIL, NEEDA, NS, WAVES> & T)
{
    constexpr bool kWide = sizeof(YT) == 2;
    constexpr uint32_t BPS = sizeof(YT);
    const int wv = (WAVES == 1) ? 0 : (int)threadIdx.y;
    // ---- bilinear: this lane's share of the tile's chroma neighbourhood (first: it heads the longest chain) ----
    if constexpr (BIL) {
        // canvas chroma row held by LDS row 0
        const int rowBase = (SUB == SUB_420) ? A.cy0 + (int)(tileY >> 1) - 1 : A.cy0 + (int)tileY;
        if constexpr (StageRows<SUB, NS, WAVES>::kUniformRows) // (seam-aware build, the four waves together: every tile the same way)
            loadNeighbourhood<YT, SUB, NEEDA, NS, WAVES, true>(A, c, rowBase, T);
        else if (haloNeeded(A, rowBase, rowBase + StageRows<SUB, NS, WAVES>::kRows - 1, c.cxb - 1, c.cxb + 128)) // (wave-uniform; never in the plain builds)
            loadNeighbourhood<YT, SUB, NEEDA, NS, WAVES, true>(A, c, rowBase, T);
        else
            loadNeighbourhood<YT, SUB, NEEDA, NS, WAVES, false>(A, c, rowBase, T);
    }
    // ---- this wave's luma / alpha / co-sited chroma for all of its strips ----
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const uint32_t sy = tileY + 2 * (wv * NS + k);
        const uint32_t syc = sy < A.h2 ? sy : 0; // absent strips load (and discard) the first one
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            T.raw[k].y[r] = load4s<YT, STREAM>(A.y, (syc + r) * A.yPitch + c.Xc * BPS);
            if constexpr (NEEDA)
                T.raw[k].a[r] = load4s<YT, STREAM>(A.a, (syc + r) * A.aPitch + c.Xc * BPS);
            if constexpr (SUB == SUB_444) {
                T.raw[k].u[r] = load4s<YT, STREAM>(A.u, ((uint32_t)A.cy0 + syc + r) * A.uPitch + ((uint32_t)A.cx0 + c.Xc) * BPS);
                T.raw[k].v[r] = load4s<YT, STREAM>(A.v, ((uint32_t)A.cy0 + syc + r) * A.vPitch + ((uint32_t)A.cx0 + c.Xc) * BPS);
            } else if constexpr ((SUB == SUB_420 || SUB == SUB_422) && !BIL) {
                // nearest: chroma samples (X>>1, X>>1 + 1) of chroma row (j >> shiftY); one aligned pair load per plane
                if (!(SUB == SUB_420 && r == 1)) {
                    const uint32_t cy = (uint32_t)A.cy0 + ((SUB == SUB_420) ? (syc >> 1) : (syc + r));
                    const uint32_t cx = (uint32_t)A.cx0 + (c.Xc >> 1);
                    if constexpr (!kWide) {
                        T.raw[k].u[r].w[0] = *reinterpret_cast<const uint16_t *>(A.u + (cy * A.uPitch + cx));
                        T.raw[k].v[r].w[0] = *reinterpret_cast<const uint16_t *>(A.v + (cy * A.vPitch + cx * 1));
                    } else {
                        T.raw[k].u[r].w[0] = *reinterpret_cast<const uint32_t *>(A.u + (cy * A.uPitch + cx * 2));
                        T.raw[k].v[r].w[0] = *reinterpret_cast<const uint32_t *>(A.v + (cy * A.vPitch + cx * 2));
                    }
                }
            }
        }
    }
}

// bilinear: normalise this lane's share of the chroma neighbourhood and put it into LDS
template <typename YT, int SUB, bool NEEDA, int NS, int WAVES = 4>
__device__ __forceinline__ void stageTile(const TileArgs & A, const TileRaw<YT, SUB, true, NEEDA, NS, WAVES> & T, f2 (*rows)[kRowPitch])
{
    typedef StageRows<SUB, NS, WAVES> SR;
    const int t = ((WAVES == 1) ? 0 : (int)threadIdx.y * kLanesX) + (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < SR::kRounds; ++j) {
        int row, grp;
        if (SR::placeOf(t, j, row, grp)) {
            float fu[4], fv[4];
            samples4<YT>(T.su[j], A.yuvMax, fu);
            samples4<YT>(T.sv[j], A.yuvMax, fv);
            f2 * dst = &rows[row][4 * grp + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dst[k] = norm2((f2) { fu[k], fv[k] }, A.biasUV, A.rcpRangeUV);
        }
    }
    if constexpr (SR::kSplitHalo) { // the sample left of the band (column 4 of the row: the last of group 0) and right of it (133: the first of group 33)
        if (t < 2 * SR::kRows) {
            const unsigned u = (sizeof(YT) == 2) ? minU(T.hu, A.yuvMax) : T.hu, v = (sizeof(YT) == 2) ? minU(T.hv, A.yuvMax) : T.hv;
            rows[t >> 1][(t & 1) ? 4 * 33 + 1 : 4] = norm2((f2) { (float)u, (float)v }, A.biasUV, A.rcpRangeUV);
        }
    }
}

// MAP (4-channel pixels, wave-private tiles only): stores go through the job's PixelMap (tile_map_impl.h) -- MAP_ROWS: rows stay rows, stored
// straight away; MAP_TURNED: quarter turns, into `held` (2 * NS rows of the lane's four pixels, 4 * PW words each), which the caller
// transposes through LDS.  Separate instantiations: the rows of a whole tile held in registers cost the row-wise kernels their occupancy.
enum MapMode : int { MAP_NONE = 0, MAP_ROWS = 1, MAP_TURNED = 2 };
// MULSEL (kernels with pending alpha arithmetic): 0 = the mode is read from the job (a wave-uniform branch around every row: all of them
// compiled in; only AVIFHIP_TUNING's TUNE_ALL_ALPHA_MODES asks for it now), 1 / 2 = in-loop multiply / un-multiply, 3 / 4 = integer post-
// multiply / un-multiply compiled in alone (selOfAlphaModes) -- 1 and 3 are what premultiplied destinations (Android's bitmaps, cfg3) ask for.  With every mode in one kernel cfg3's took 76 KB of code and 116 registers (four waves per SIMD); its own: 28 KB, 94
// registers.  The big kernel's speed depends on where the compiler happens to lay its blocks (84 us in one build, 100-102 in two others
// that differed only in code it never executes: the eight unrolled row bodies a wave walks through are spread over more code than the
// instruction cache holds); the small one ran at 78.8-82.6 us in every build measured.
template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, int WAVES = 4, int MAP = MAP_NONE, int MULSEL = 0>
__device__ __forceinline__ void computeTile(const TileArgs & A, const BandCtx & c, uint32_t tileY,
                                            const TileRaw<YT, SUB, BIL, APLANE || HASMUL, NS, WAVES> & T, f2 (*rows)[kRowPitch], WideRowExchange * xchg,
                                            unsigned * held = nullptr)
{
    constexpr bool MAPPED = MAP != MAP_NONE;
    static_assert(!MAPPED || (NCH == 4 && WAVES == 1), "mapped stores: 4-channel pixels, wave-private tiles");
    constexpr int PW = (sizeof(RT) == 1) ? 1 : 2; // dwords per 4-channel pixel
    constexpr bool kWide = sizeof(YT) == 2;
    constexpr bool kNeedA = APLANE || HASMUL;
    constexpr bool IS565 = NCH == 5; // RGB565: one 16-bit word per pixel (template code 5; 8-bit RT)
    // (HASMUL: an image with an alpha plane converted to RGB565 is premultiplied in the loop -- src/reformat.c:1503-1511 -- there is no post-pass)
    static_assert(!IS565 || (sizeof(RT) == 1 && !APLANE && !MAPPED), "RGB565: 8-bit channels, no alpha channel");
    constexpr uint32_t kPixBytes = IS565 ? 2u : NCH * sizeof(RT);
    const int tx = threadIdx.x, wv = (WAVES == 1) ? 0 : (int)threadIdx.y; // WAVES == 1: `xchg` is this wave's own exchange buffer
    const bool nt = (A.tuning & TUNE_NONTEMPORAL) != 0;
    const unsigned yuvMax = A.yuvMax;
    const uint32_t X = c.X;
    const bool laneValid = c.laneValid;
    const StripRaw<YT, SUB, BIL, kNeedA> * raw = T.raw;
    const bool alphaFirst = (NCH == 4 || NCH == 2) && (A.slotA == 0);
    // (first colour X, third colour Z) of a pixel from the two chroma planes in TileArgs order (tile_shared.h): for BGR
    // orders (Cb,Cr) -> (B - Y, R - Y), src/reformat.c:874-875; for RGB orders the planes and coefficients arrive swapped
    const f2 cBR = { A.cB, A.cR };
    const f2 cUV = { A.cU, A.cV }; // the two products of the green term, :876 (their sum is commutative)
    const unsigned opaqueWord = A.rgbMax << (8 * A.slotA); // 8-bit RGBA: the alpha byte in place
#ifdef AVIFHIP_PROBE_MULMODE // instruction-count probes only (tests/tools/isa_count.py): one alpha mode compiled in
    const int inLoopMode = (AVIFHIP_PROBE_MULMODE) <= 2 ? (AVIFHIP_PROBE_MULMODE) : MUL_NONE, postMode = (AVIFHIP_PROBE_MULMODE) > 2 ? (AVIFHIP_PROBE_MULMODE) - 2 : MUL_NONE;
#else
    const int inLoopMode = MULSEL == 0 ? A.inLoopMul : (MULSEL == 1 ? (int)MUL_MULTIPLY : (MULSEL == 2 ? (int)MUL_UNMULTIPLY : (int)MUL_NONE));
    const int postMode = MULSEL == 0 ? A.postMul : (MULSEL == 3 ? (int)MUL_MULTIPLY : (MULSEL == 4 ? (int)MUL_UNMULTIPLY : (int)MUL_NONE));
#endif
    // 3-channel pixels: bytes of the band's row segment that exist (storeRowContiguous)
    const uint32_t segBytes = ((A.w4 - c.bandX < (uint32_t)kBandW) ? A.w4 - c.bandX : (uint32_t)kBandW) * kPixBytes;

    struct // 4:2:0 bilinear: what the strip's co-sited and lower chroma rows hand to the next strip (see there)
    {
        f2 m3b, m3c, m1[4], b[4], b3b, b3c, b1[4];
    } carry;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const uint32_t sy = tileY + 2 * (wv * NS + k);
        if (sy >= A.h2)
            break;

        // ---- (Cb,Cr) for the lane's 2 x 4 pixels ----
        f2 uv[2][4];
        if constexpr (SUB == SUB_400) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    uv[r][i] = splat(0.5f);
        } else if constexpr (SUB == SUB_444) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float fu[4], fv[4];
                samples4<YT>(raw[k].u[r], yuvMax, fu);
                samples4<YT>(raw[k].v[r], yuvMax, fv);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    uv[r][i] = norm2((f2) { fu[i], fv[i] }, A.biasUV, A.rcpRangeUV);
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
                unsigned u0 = raw[k].u[r].w[0] & kMask, u1 = (raw[k].u[r].w[0] >> kShift) & kMask;
                unsigned v0 = raw[k].v[r].w[0] & kMask, v1 = (raw[k].v[r].w[0] >> kShift) & kMask;
                if (kWide) {
                    u0 = minU(u0, yuvMax), u1 = minU(u1, yuvMax), v0 = minU(v0, yuvMax), v1 = minU(v1, yuvMax);
                }
                const f2 c0 = norm2((f2) { (float)u0, (float)v0 }, A.biasUV, A.rcpRangeUV);
                const f2 c1 = norm2((f2) { (float)u1, (float)v1 }, A.biasUV, A.rcpRangeUV);
                uv[r][0] = uv[r][1] = c0;
                uv[r][2] = uv[r][3] = c1;
            }
        } else {
            // 4-tap filter on normalised samples, src/reformat.c:834-837: ((closest*9/16 + horizontal*3/16) + vertical*3/16)
            // + diagonal*1/16, evaluated for Cb and Cr at once; every product equals the reference's product for that tap.
            const f2 k9 = splat(9.0f / 16.0f), k3 = splat(3.0f / 16.0f), k1 = splat(1.0f / 16.0f);
            auto loadRow = [&](int q, f2 m[4]) {
                const f2 * src = &rows[q][2 * tx + 4];
                const f4 lo = *reinterpret_cast<const f4 *>(src);
                const f4 hi = *reinterpret_cast<const f4 *>(src + 2);
                m[0] = lo.xy, m[1] = lo.zw, m[2] = hi.xy, m[3] = hi.zw;
            };
            if constexpr (SUB == SUB_420) {
                // A chroma row serves three strips in turn: as the row below (its products by 3/16 of the two middle columns and by 1/16 of all
                // four), as the co-sited row (adds 9/16 of the middle columns, 3/16 of the outer ones) and as the row above (nothing new).  The
                // loop over the wave's strips is unrolled, so a row's samples and products stay in registers from one strip to the next: one
                // row read from LDS and ten products per strip instead of three rows and eighteen.  Same products, same sums as before.
                const int qm = 1 + wv * NS + k; // LDS row of the strip's co-sited chroma row
                f2 m[4], m3b, m3c, m1[4], a3b, a3c, a1[4];
                if (k == 0) {
                    f2 a[4];
                    loadRow(qm, m);
                    loadRow(qm - 1, a); // even luma rows: vertical neighbour above
                    m3b = m[1] * k3, m3c = m[2] * k3;
                    a3b = a[1] * k3, a3c = a[2] * k3;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        a1[i] = a[i] * k1; // (m1: the next strip's row above -- filled below)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        m1[i] = m[i] * k1;
                } else {
                    a3b = carry.m3b, a3c = carry.m3c;
                    m3b = carry.b3b, m3c = carry.b3c;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        a1[i] = carry.m1[i], m[i] = carry.b[i], m1[i] = carry.b1[i];
                }
                f2 b[4], b1[4];
                loadRow(qm + 1, b); // odd luma rows: below
                const f2 b3b = b[1] * k3, b3c = b[2] * k3;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    b1[i] = b[i] * k1;
                const f2 m9b = m[1] * k9, m9c = m[2] * k9;
                const f2 m3a = m[0] * k3, m3d = m[3] * k3;
                const f2 h0 = m9b + m3a, h1 = m9b + m3c, h2 = m9c + m3b, h3 = m9c + m3d;
                uv[0][0] = (h0 + a3b) + a1[0]; // even pixel: horizontal neighbour on the left
                uv[0][1] = (h1 + a3b) + a1[2]; // odd pixel: on the right
                uv[0][2] = (h2 + a3c) + a1[1];
                uv[0][3] = (h3 + a3c) + a1[3];
                uv[1][0] = (h0 + b3b) + b1[0];
                uv[1][1] = (h1 + b3b) + b1[2];
                uv[1][2] = (h2 + b3c) + b1[1];
                uv[1][3] = (h3 + b3c) + b1[3];
                carry.m3b = m3b, carry.m3c = m3c, carry.b3b = b3b, carry.b3c = b3c;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    carry.m1[i] = m1[i], carry.b[i] = b[i], carry.b1[i] = b1[i];
            } else { // 4:2:2: the vertical neighbour is the sample itself (src/reformat.c:784-786)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    f2 m[4];
                    loadRow(2 * (wv * NS + k) + r, m);
                    const f2 m9b = m[1] * k9, m9c = m[2] * k9;
                    const f2 m3a = m[0] * k3, m3b = m[1] * k3, m3c = m[2] * k3, m3d = m[3] * k3;
                    const f2 m1a = m[0] * k1, m1b = m[1] * k1, m1c = m[2] * k1, m1d = m[3] * k1;
                    uv[r][0] = ((m9b + m3a) + m3b) + m1a;
                    uv[r][1] = ((m9b + m3c) + m3b) + m1c;
                    uv[r][2] = ((m9c + m3b) + m3c) + m1b;
                    uv[r][3] = ((m9c + m3d) + m3c) + m1d;
                }
            }
        }

        // ---- per-pixel arithmetic and stores ----
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float fy[4];
            samples4<YT>(raw[k].y[r], yuvMax, fy);
            const f2 y01 = norm2((f2) { fy[0], fy[1] }, A.biasY, A.rcpRangeY);
            const f2 y23 = norm2((f2) { fy[2], fy[3] }, A.biasY, A.rcpRangeY);
            const float yk[4] = { y01.x, y01.y, y23.x, y23.y };

            unsigned av[4] = { 0, 0, 0, 0 }, a[4];
            if constexpr (kNeedA) {
                decode4<YT>(raw[k].a[r], av);
                if (A.alphaLim.on) { // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[i] = alphaToFullRange(A, av[i]);
                }
            }
            if (APLANE && A.alphaRescale) { // wave-uniform, and a branch around all four pixels (per pixel the compiler computes the rescale and selects)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = alphaRescaled(A, av[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = APLANE ? av[i] : A.rgbMax;
            }
            const uint32_t off = (sy + r) * A.rgbPitch + X * kPixBytes;
            const uint32_t bandOff = (sy + r) * A.rgbPitch + c.bandX * kPixBytes;
            if constexpr (APLANE && !HASMUL && !MAPPED) {
                if (A.alphaKeep) { // wave-uniform: rgb->ignoreAlpha -- the pixel keeps the alpha sample it has (read here, stored back with the colours)
                    // the lane's four pixels as whole words (4 x kPixBytes = 8, 16 or 32 bytes), the alpha sample of each picked out
                    constexpr int kWords = (int)kPixBytes; // 32-bit words of four pixels
                    unsigned old[kWords];
#pragma unroll
                    for (int wd = 0; wd < kWords; ++wd)
                        old[wd] = 0;
                    if (laneValid) {
                        if constexpr (kWords == 2) {
                            const u2 t = *reinterpret_cast<const u2 *>(A.rgb + off);
                            old[0] = t.x, old[1] = t.y;
                        } else {
#pragma unroll
                            for (int q4 = 0; q4 < kWords / 4; ++q4) {
                                const u4 t = *reinterpret_cast<const u4 *>(A.rgb + (off + 16u * (uint32_t)q4));
                                old[4 * q4] = t.x, old[4 * q4 + 1] = t.y, old[4 * q4 + 2] = t.z, old[4 * q4 + 3] = t.w;
                            }
                        }
                    }
                    const uint32_t slotBits = A.slotA * 8u * (uint32_t)sizeof(RT); // bit position of the alpha sample inside its pixel
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // pixel i occupies kPixBytes * 8 bits from bit i * kPixBytes * 8 of `old`
                        if constexpr (kPixBytes == 8) { // two words per pixel: the sample's word chosen by a wave-uniform select
                            const unsigned wsel = (slotBits >= 32u) ? old[2 * i + 1] : old[2 * i];
                            a[i] = (wsel >> (slotBits & 31u)) & 0xffffu;
                        } else if constexpr (kPixBytes == 4) {
                            a[i] = (old[i] >> slotBits) & (sizeof(RT) == 1 ? 0xffu : 0xffffu);
                        } else { // 2-byte pixels (GRAYA8 / AGRAY8): two per word
                            a[i] = (old[i >> 1] >> (16u * (uint32_t)(i & 1) + slotBits)) & 0xffu;
                        }
                    }
                }
            }

            // finished pixel words through the job's PixelMap: canvas pixel (mapX0 + X .. + 3, mapY0 + row)
            auto emitMapped = [&](const unsigned (&px)[4][PW]) {
                if constexpr (MAP == MAP_TURNED) { // quarter turns leave through the caller's LDS transposition
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int wd = 0; wd < PW; ++wd)
                            held[((2 * k + r) * 4 + i) * PW + wd] = px[i][wd];
                } else if constexpr (MAP == MAP_ROWS) {
                    if constexpr (PW == 1) {
                        if (laneValid)
                            mapStoreRow<1>(A, px, (uint32_t)A.mapX0 + X, (uint32_t)A.mapY0 + sy + (uint32_t)r, nt);
                    } else {
                        mapStoreRowWide(A, px, (uint32_t)A.mapX0 + c.bandX, c.bandX, (uint32_t)A.mapY0 + sy + (uint32_t)r, reinterpret_cast<mu4 *>(xchg[wv].w));
                    }
                }
            };
            // RGB565 (Android's bitmap format; here from the fp32 arithmetic: 10- / 12-bit sources, filtered chroma, avoidLibYUV): the format has no
            // channel offsets, so the 'first' colour is blue and the 'third' red (tile_shared.h distillArgs); four pixels = 8 bytes per lane, at
            // the 2-byte alignment a row of 16-bit pixels has (odd widths, padded pitches: the hardware splits what straddles).
            // b >> 3 | (g >> 2) << 5 | (r >> 3) << 11 of the 8-bit channels, src/reformat.c:619-626
            auto emit565 = [&](const unsigned (&b8)[4], const unsigned (&g8)[4], const unsigned (&r8)[4]) {
                unsigned h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    h[i] = pack565(r8[i], g8[i], b8[i]);
                if (laneValid)
                    storeVec(A.rgb, off, (u2a2) { h[0] | (h[1] << 16), h[2] | (h[3] << 16) }, nt);
            };
            // finished integers (q[i].r / q[i].b: first / third colour channel) to their pixels
            auto emitQ = [&](PixelOut (&q)[4]) {
                if constexpr (sizeof(RT) == 2) {
                    if (A.f16Mul != 0.0f) { // wave-uniform: avifRGBImageToF16 (src/reformat.c:1419-1443) on every channel, alpha included
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            q[i].r = toHalfBits(q[i].r, A.f16Mul), q[i].g = toHalfBits(q[i].g, A.f16Mul), q[i].b = toHalfBits(q[i].b, A.f16Mul);
                            a[i] = toHalfBits(a[i], A.f16Mul);
                        }
                    }
                }
                if constexpr (MAPPED) {
                    unsigned px[4][PW];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (PW == 1) {
                            px[i][0] = alphaFirst ? (a[i] | (q[i].r << 8) | (q[i].g << 16) | (q[i].b << 24)) : (q[i].r | (q[i].g << 8) | (q[i].b << 16) | (a[i] << 24));
                        } else {
                            px[i][0] = alphaFirst ? (a[i] | (q[i].r << 16)) : (q[i].r | (q[i].g << 16));
                            px[i][1] = alphaFirst ? (q[i].g | (q[i].b << 16)) : (q[i].b | (a[i] << 16));
                        }
                    }
                    emitMapped(px);
                } else if constexpr (sizeof(RT) == 2 && NCH == 4) {
                    store4WideRgba(A.rgb, bandOff, q, a, alphaFirst, c.bandX, A.w4, xchg[wv]);
                } else if constexpr (NCH == 3) {
                    store4Rgb3<RT>(A.rgb, bandOff, q, segBytes, xchg[wv]);
                } else if constexpr (NCH <= 2) {
                    if (laneValid)
                        storeGray4<RT, NCH>(A.rgb, off, q, a, alphaFirst, nt);
                } else if constexpr (IS565) {
                    const unsigned b8[4] = { q[0].r, q[1].r, q[2].r, q[3].r }, g8[4] = { q[0].g, q[1].g, q[2].g, q[3].g }, r8[4] = { q[0].b, q[1].b, q[2].b, q[3].b };
                    emit565(b8, g8, r8);
                } else {
                    if (laneValid)
                        store4<RT, NCH>(A.rgb, off, q, a, false, alphaFirst, nt);
                }
            };
            // the quantiser's arguments t = c * max + 0.5f (quantizeArg) of the row's (first, third) colours and of green to their pixels:
            // 8-bit colour outputs truncate, saturate and pack in one instruction per channel (packRgba8Row), the others through v_cvt_u32_f32
            auto emitT = [&](const f2 (&tbr)[4], const float (&tg)[4]) {
                if constexpr (sizeof(RT) == 1 && NCH >= 3) {
                    if constexpr (NCH == 4) {
                        unsigned w[4], aw[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            aw[i] = APLANE ? (a[i] << (8 * A.slotA)) : opaqueWord;
                        packRgba8Row(w, aw, tbr, tg, A.slotZ, A.slotG, A.slotX);
                        if constexpr (MAPPED) {
                            const unsigned px[4][PW] = { { w[0] }, { w[1] }, { w[2] }, { w[3] } };
                            emitMapped(px);
                        } else if (laneValid) {
                            storeVec(A.rgb, off, (u4) { w[0], w[1], w[2], w[3] }, nt);
                        }
                    } else if constexpr (IS565) {
                        unsigned b8[4], g8[4], r8[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            b8[i] = minU(truncU32(tbr[i].x), 255u), g8[i] = minU(truncU32(tg[i]), 255u), r8[i] = minU(truncU32(tbr[i].y), 255u);
                        emit565(b8, g8, r8);
                    } else {
                        float x[4], z[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            x[i] = tbr[i].x;
                            z[i] = tbr[i].y;
                        }
                        unsigned w[3];
                        packRgb8Row(w, x, tg, z);
                        storeRowContiguous<3>(A.rgb, bandOff, w, segBytes, xchg[wv]);
                    }
                } else {
                    PixelOut q[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        q[i].r = minU(truncU32(tbr[i].x), A.rgbMax);
                        q[i].g = minU(truncU32(tg[i]), A.rgbMax);
                        q[i].b = minU(truncU32(tbr[i].y), A.rgbMax);
                    }
                    emitQ(q);
                }
            };

            if constexpr (SUB == SUB_444 && sizeof(YT) == 1 && sizeof(RT) == 1 && !HASMUL) {
                // identity matrix, 8 bits in and out, full range (lossless RGB in 4:4:4 planes): G = Y, B = Cb, R = Cr, a byte shuffle
                // (avifImageIdentity8ToRGB8ColorFullRange, src/reformat.c:1278-1309); `u` feeds the first colour channel (tile_shared.h)
                if (A.identityCopy) { // wave-uniform
                    unsigned yb[4], ub[4], vb[4];
                    decode4<YT>(raw[k].y[r], yb);
                    decode4<YT>(raw[k].u[r], ub);
                    decode4<YT>(raw[k].v[r], vb);
                    PixelOut q[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        q[i].r = ub[i], q[i].g = yb[i], q[i].b = vb[i];
                    if constexpr (NCH == 3) {
                        store4Rgb3<RT>(A.rgb, bandOff, q, segBytes, xchg[wv]);
                    } else if constexpr (IS565) {
                        emit565(ub, yb, vb);
                    } else if constexpr (MAPPED) {
                        unsigned px[4][PW];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            px[i][0] = alphaFirst ? (a[i] | (q[i].r << 8) | (q[i].g << 16) | (q[i].b << 24)) : (q[i].r | (q[i].g << 8) | (q[i].b << 16) | (a[i] << 24));
                        emitMapped(px);
                    } else {
                        if (laneValid)
                            store4<RT, NCH>(A.rgb, off, q, a, false, alphaFirst, nt);
                    }
                    continue;
                }
            }

            if constexpr (HASMUL) {
                if (inLoopMode != MUL_NONE) { // wave-uniform: the slow path's alpha arithmetic in fp32 before quantisation (src/reformat.c:894-947)
                    // clamped colours: the matrix in scalar form so that each channel's last instruction carries the clamp
                    float X[4], G[4], Z[4];
                    if constexpr (SUB == SUB_400) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            X[i] = G[i] = Z[i] = sat01(yk[i]);
                    } else if (SUB == SUB_444 && A.identityMatrix) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            X[i] = sat01(uv[r][i].x), G[i] = sat01(yk[i]), Z[i] = sat01(uv[r][i].y);
                    } else if (A.ycgco) {
                        unsigned yc[4];
                        decode4<YT>(raw[k].y[r], yc);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x, g, z;
                            ycgcoPixel(A, yk[i], uv[r][i], kWide ? minU(yc[i], yuvMax) : yc[i], x, g, z);
                            X[i] = sat01(x), G[i] = sat01(g), Z[i] = sat01(z);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            X[i] = addSat01(yk[i], A.cB * uv[r][i].x);
                            Z[i] = addSat01(yk[i], A.cR * uv[r][i].y);
                            const float sum = (A.cV * uv[r][i].y) + (A.cU * uv[r][i].x);
                            G[i] = subSat01(yk[i], __builtin_fmaf(sum, A.rcpKgTimes2.hi, sum * A.rcpKgTimes2.lo));
                        }
                    }
                    f2 tbr[4];
                    float tg[4];
                    if (inLoopMode == MUL_MULTIPLY) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const InLoopAlpha L = inLoopAlpha<false, kWide>(A, av[i]);
                            tbr[i] = (f2) { inLoopChannel<false>(A, X[i], L), inLoopChannel<false>(A, Z[i], L) };
                            tg[i] = inLoopChannel<false>(A, G[i], L);
                        }
                    } else if constexpr (!kWide) { // 8-bit alpha planes: the workgroup's table (built by runBlock / runSolo / runSoloMapped)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            unmulFromTable(A, av[i], X[i], G[i], Z[i], tbr[i], tg[i]);
                    } else if (A.yuvMax <= 4095u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const InLoopAlpha L = inLoopAlpha<true, kWide>(A, av[i]);
                            tbr[i] = (f2) { inLoopChannel<true>(A, X[i], L), inLoopChannel<true>(A, Z[i], L) };
                            tg[i] = inLoopChannel<true>(A, G[i], L);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float Ac = inLoopAlpha<false, kWide>(A, av[i]).Ac;
                            tbr[i] = (f2) { inLoopChannelIeee(A, X[i], Ac), inLoopChannelIeee(A, Z[i], Ac) };
                            tg[i] = inLoopChannelIeee(A, G[i], Ac);
                        }
                    }
                    emitT(tbr, tg);
                    continue;
                }
            }

            f2 br[4]; // unclamped (first, third) colour per pixel
            float g[4];
            if constexpr (SUB == SUB_400) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    br[i] = splat(yk[i]);
                    g[i] = yk[i];
                }
            } else if (SUB == SUB_444 && A.identityMatrix) { // wave-uniform: GBR planes, src/reformat.c:855-858 -- G = Y, B = Cb, R = Cr (normalised on luma's scale)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    br[i] = uv[r][i];
                    g[i] = yk[i];
                }
            } else if (A.ycgco) { // wave-uniform: YCgCo / YCgCo-Re / -Ro
                unsigned yc[4];
                decode4<YT>(raw[k].y[r], yc);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x, z;
                    ycgcoPixel(A, yk[i], uv[r][i], kWide ? minU(yc[i], yuvMax) : yc[i], x, g[i], z);
                    br[i] = (f2) { x, z };
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    br[i] = splat(yk[i]) + cBR * uv[r][i];
                    const f2 pr = cUV * uv[r][i];
                    // (kr(1-kr)*Cr) + (kb(1-kb)*Cb).  (The kernels with pending alpha arithmetic and unfiltered chroma keep the C form: they are bound
                    // by memory, and cfg3's lost 19 % to the very same instructions laid out differently around this block -- 76 KB of code.)
                    const float sum = (BIL || !HASMUL) ? addScalar(pr.y, pr.x) : pr.y + pr.x;
                    // G = Y - (2*sum)/kg with 2/kg in verified reciprocal form
                    g[i] = yk[i] - __builtin_fmaf(sum, A.rcpKgTimes2.hi, sum * A.rcpKgTimes2.lo);
                }
            }

            if constexpr (HASMUL) {
                // fast paths: quantise, then the integer post-pass (src/reformat.c:1574-1585)
                PixelOut q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[i].r = quantizeSat(br[i].x, A.rgbMaxF, A.rgbMax);
                    q[i].g = quantizeSat(g[i], A.rgbMaxF, A.rgbMax);
                    q[i].b = quantizeSat(br[i].y, A.rgbMaxF, A.rgbMax);
                }
                if (sizeof(RT) == 1 && A.postMulFx) {
                    // (wave-uniform) a libyuv build runs libyuv's ARGBAttenuate / ARGBUnattenuate over 8-bit RGBA / BGRA pixels whoever converted them
                    // (src/alpha.c:163,350): here over the fp32 loops' bytes -- sources libyuv has no entry for
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        q[i].r = fxAlphaMul(q[i].r, a[i], postMode), q[i].g = fxAlphaMul(q[i].g, a[i], postMode), q[i].b = fxAlphaMul(q[i].b, a[i], postMode);
                } else if (postMode == MUL_MULTIPLY) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const PostAlpha P = postAlpha<false>(A, a[i]);
                        q[i].r = postChannel<false>(A, q[i].r, P), q[i].g = postChannel<false>(A, q[i].g, P), q[i].b = postChannel<false>(A, q[i].b, P);
                    }
                } else if (postMode == MUL_UNMULTIPLY) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const PostAlpha P = postAlpha<true>(A, a[i]);
                        q[i].r = postChannel<true>(A, q[i].r, P), q[i].g = postChannel<true>(A, q[i].g, P), q[i].b = postChannel<true>(A, q[i].b, P);
                    }
                }
                emitQ(q);
            } else {
                const f2 half = splat(0.5f), mx = splat(A.rgbMaxF);
                f2 tbr[4];
                float tg[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    tbr[i] = fma2(br[i], mx, half);
                    tg[i] = quantizeArg(g[i], A.rgbMaxF);
                }
                emitT(tbr, tg);
            }
        }
    }
}


// One workgroup walks down `tilesPerRun` vertically consecutive tiles (256 x 8*NS pixels each) of one band.  The loads
// of tile i+1 are in flight while tile i is computed and stored, so a wave waits for memory once per run, not once
// per tile; vertically consecutive tiles also re-read their shared chroma halo rows from the nearest cache.
template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, bool STREAM = false>
__device__ __forceinline__ void runBlock(const TileArgs & A, uint32_t tilesPerRun, f2 (*rows)[BIL ? StageRows<SUB, NS>::kRows : 1][kRowPitch],
                                         WideRowExchange * xchg, uint32_t canvasColumns = 0)
{
    constexpr int kTileH = 8 * NS;
    constexpr bool kNeedA = APLANE || HASMUL;
    prepareAlphaTables<YT, HASMUL, 0>(A);
    const uint32_t bands = (A.w4 + kBandW - 1) / kBandW;
    const uint32_t tilesY = (A.h2 + kTileH - 1) / kTileH;
    const uint32_t runsY = (tilesY + tilesPerRun - 1) / tilesPerRun;
    const uint32_t nRuns = bands * runsY;
    uint32_t rrow, band;
    if (canvasColumns) { // tiles of one canvas, workgroups along the canvas rows (tile_geom.h PkGeom::canvasColumns): x = (canvas column, band), y = run row
        const uint32_t bandsPerJob = gridDim.x / canvasColumns;
        band = blockIdx.x % bandsPerJob, rrow = blockIdx.y;
        if (band >= bands || rrow >= runsY)
            return;
    } else {
        const uint32_t run = blockRemap(blockIdx.x, gridDim.x, (A.tuning & TUNE_XCD_BANDS) != 0 && (gridDim.z == 1 || (gridDim.x & 7) == 0));
        if (run >= nRuns)
            return;
        rrow = run / bands, band = run - rrow * bands;
    }
    const uint32_t bandX = band * kBandW;
    const uint32_t firstTile = rrow * tilesPerRun;
    const uint32_t nTiles = (tilesY - firstTile < tilesPerRun) ? (tilesY - firstTile) : tilesPerRun;

    BandCtx c;
    c.bandX = bandX;
    c.X = bandX + 4 * threadIdx.x;
    c.laneValid = c.X < A.w4;
    c.Xc = c.laneValid ? c.X : 0; // absent lanes load (and discard) the row's first group
    c.cxb = A.cx0 + (int)(bandX >> 1);

    // Two LDS buffers: while tile i is computed from one, the neighbourhood of tile i+1 (whose loads were issued before
    // the computation started) is staged into the other -- one barrier per tile, and no wave waits for the slowest one
    // between "done reading" and "may overwrite".
    TileRaw<YT, SUB, BIL, kNeedA, NS> cur;
    uint32_t tileY = firstTile * kTileH;
    loadTile<YT, SUB, BIL, kNeedA, NS, 4, STREAM>(A, c, tileY, cur);
    if constexpr (BIL) {
        stageTile<YT, SUB, kNeedA, NS>(A, cur, rows[0]);
        __syncthreads();
    }
    for (uint32_t i = 0; i < nTiles; ++i) {
        const bool more = i + 1 < nTiles;
        TileRaw<YT, SUB, BIL, kNeedA, NS> nxt;
        if (more)
            loadTile<YT, SUB, BIL, kNeedA, NS, 4, STREAM>(A, c, tileY + kTileH, nxt);
        computeTile<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS>(A, c, tileY, cur, rows[i & 1], xchg);
        if (!more)
            break;
        if constexpr (BIL) {
            stageTile<YT, SUB, kNeedA, NS>(A, nxt, rows[(i + 1) & 1]);
            __syncthreads();
        }
        cur = nxt;
        tileY += kTileH;
    }
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS>
__global__ __launch_bounds__(256) void yuvToRgbTileKernel(TileArgs A, uint32_t tilesPerRun)
{
    __shared__ __attribute__((aligned(16))) f2 rows[BIL ? 2 : 1][BIL ? StageRows<SUB, NS>::kRows : 1][kRowPitch];
    if constexpr ((sizeof(RT) == 2 && NCH == 4) || NCH == 3) {
        __shared__ WideRowExchange xchg[kWavesPerBlock]; // one per wave, 16-bit RGBA only
        runBlock<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS>(A, tilesPerRun, rows, xchg);
    } else {
        runBlock<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS>(A, tilesPerRun, rows, nullptr);
    }
}

// one launch for a table of jobs (grid z = job); the descriptor is read with scalar loads
template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, bool STREAM = false>
__global__ __launch_bounds__(256) void yuvToRgbTileBatchKernel(const TileArgs * __restrict__ table, uint32_t tilesPerRun, uint32_t canvasColumns)
{
    __shared__ __attribute__((aligned(16))) f2 rows[BIL ? 2 : 1][BIL ? StageRows<SUB, NS>::kRows : 1][kRowPitch];
    // a private copy of the job: read through the table pointer, every field would be re-loaded after each store (the compiler
    // cannot rule out that the RGB stores alias the table) -- ~20 scalar loads per tile inside the pipelined loop
    const uint32_t jobIndex = canvasColumns ? blockIdx.z * canvasColumns + blockIdx.x / (gridDim.x / canvasColumns) : blockIdx.z;
    const TileArgs job = jobOf<true>(table, jobIndex);
    if constexpr ((sizeof(RT) == 2 && NCH == 4) || NCH == 3) {
        __shared__ WideRowExchange xchg[kWavesPerBlock];
        runBlock<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM>(job, tilesPerRun, rows, xchg, canvasColumns);
    } else {
        runBlock<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM>(job, tilesPerRun, rows, nullptr, canvasColumns);
    }
}

// ---- every wave for itself (the structure of tile_pk_impl.h): one wave = one tile of 256 x 2*NS pixels, every load issued up
//      front, the chroma neighbourhood in a wave-private LDS block, NO workgroup barrier; tiles in per-XCD chunks (tile_geom.h).
//      Replaces the cooperative runs above wherever it measured faster (launchOne) ----
template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, bool STREAM = false, int MULSEL = 0>
__device__ __forceinline__ void runSolo(const TileArgs & A, const PkGeom & g, f2 * lds, WideRowExchange * xchg, uint32_t tile)
{
    constexpr bool kNeedA = APLANE || HASMUL;
    typedef StageRows<SUB, NS, 1> SR;
    prepareAlphaTables<YT, HASMUL, MULSEL>(A);
    if (tile >= g.nTiles)
        return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const PkPlace place = pkPlaceOf(tile, wave, g, (uint32_t)NS);
    const uint32_t bandX = place.band * (uint32_t)kBandW;
    const uint32_t tileY = place.strip0 * 2u;
    if (bandX >= A.w4 || tileY >= A.h2)
        return; // no barrier from here on: a wave without work simply leaves
    BandCtx c;
    c.bandX = bandX;
    c.X = bandX + 4 * threadIdx.x;
    c.laneValid = c.X < A.w4;
    c.Xc = c.laneValid ? c.X : 0;
    c.cxb = A.cx0 + (int)(bandX >> 1);
    TileRaw<YT, SUB, BIL, kNeedA, NS, 1> raw;
    loadTile<YT, SUB, BIL, kNeedA, NS, 1, STREAM>(A, c, tileY, raw);
    f2(*rows)[kRowPitch] = reinterpret_cast<f2(*)[kRowPitch]>(lds + (size_t)wave * (BIL ? SR::kRows : 1) * kRowPitch);
    if constexpr (BIL) {
        stageTile<YT, SUB, kNeedA, NS, 1>(A, raw, rows);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    computeTile<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, 1, MAP_NONE, MULSEL>(A, c, tileY, raw, rows, xchg ? xchg + wave : nullptr);
}

// (STREAM: single images of 16-bit planes too large for the Infinity Cache -- 8K 4:4:4 + alpha is 265 MB of planes -- may take streaming
//  loads: kernels_tile.hip launchYuvToRgbTile)
template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, int MULSEL = 0, bool STREAM = false>
__global__ __launch_bounds__(256) void yuvToRgbTileSoloKernel(TileArgs A, PkGeom g, SeqFrames S)
{
    __shared__ __attribute__((aligned(16))) f2 lds[kWavesPerBlock * (BIL ? StageRows<SUB, NS, 1>::kRows : 1) * kRowPitch];
    const TileArgs job = seqJob(A, S); // (grid z = frame of a sequence, tile_shared.h SeqFrames: a single image is a sequence of one)
    if constexpr ((sizeof(RT) == 2 && NCH == 4) || NCH == 3) {
        __shared__ WideRowExchange xchg[kWavesPerBlock];
        runSolo<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM, MULSEL>(job, g, lds, xchg, pkTileOf(blockIdx.x, g));
    } else {
        runSolo<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM, MULSEL>(job, g, lds, nullptr, pkTileOf(blockIdx.x, g));
    }
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool HASMUL, int NS, bool STREAM = false, int MULSEL = 0>
__global__ __launch_bounds__(256) void yuvToRgbTileSoloBatchKernel(const TileArgs * __restrict__ table, PkGeom g)
{
    __shared__ __attribute__((aligned(16))) f2 lds[kWavesPerBlock * (BIL ? StageRows<SUB, NS, 1>::kRows : 1) * kRowPitch];
    const BatchWhere where = pkBatchWhere(g);
    const TileArgs job = jobOf(table, where.job); // a private copy: see yuvToRgbTileBatchKernel
    if constexpr ((sizeof(RT) == 2 && NCH == 4) || NCH == 3) {
        __shared__ WideRowExchange xchg[kWavesPerBlock];
        runSolo<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM, MULSEL>(job, g, lds, xchg, where.tile);
    } else {
        runSolo<YT, SUB, BIL, RT, NCH, APLANE, HASMUL, NS, STREAM, MULSEL>(job, g, lds, nullptr, where.tile);
    }
}

// ---- ... storing through the job's PixelMap (fused crop / rotate / mirror; 4-channel pixels).  LDS is sized at launch (SoloMapLds): the
//      waves' chroma blocks, then their exchange buffers (8-byte pixels); for quarter turns the workgroup's transposition tile (32 KiB)
//      overlays both behind a barrier.  Quarter turns need the four waves stacked and ALL of them to the end, with or without rows of
//      their own: every wave stores a quarter of the tile's columns ----
template <typename YT, int SUB, bool BIL, typename RT, int NS>
struct SoloMapLds
{
    static constexpr uint32_t kStageBytes = (uint32_t)(kWavesPerBlock * (BIL ? StageRows<SUB, NS, 1>::kRows : 1) * kRowPitch * sizeof(f2));
    static constexpr uint32_t kXchgBytes = (sizeof(RT) == 2) ? (uint32_t)(kWavesPerBlock * sizeof(WideRowExchange)) : 0u;
    static constexpr uint32_t kPlain = kStageBytes + kXchgBytes;
    static constexpr uint32_t kTileBytes = 4u * MapTile<(sizeof(RT) == 1) ? 1 : 2>::kWords;
    static constexpr bool kTurns = 8 * NS == (int)MapTile<(sizeof(RT) == 1) ? 1 : 2>::kRows; // the tile's rows = the four waves' rows
    static constexpr uint32_t kTurned = kPlain > kTileBytes ? kPlain : kTileBytes;
};

template <typename YT, int SUB, bool BIL, typename RT, bool APLANE, bool HASMUL, int NS, bool TURNED>
__device__ __forceinline__ void runSoloMapped(const TileArgs & A, const PkGeom & g, uint8_t * ldsBytes)
{
    constexpr bool kNeedA = APLANE || HASMUL;
    constexpr int PW = (sizeof(RT) == 1) ? 1 : 2;
    typedef StageRows<SUB, NS, 1> SR;
    typedef SoloMapLds<YT, SUB, BIL, RT, NS> LDS;
    static_assert(!TURNED || LDS::kTurns, "quarter turns: the workgroup's rows must be the transposition tile's");
    prepareAlphaTables<YT, HASMUL, 0>(A);
    const uint32_t tile = pkTileOf(blockIdx.x, g);
    if (tile >= g.nTiles)
        return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const PkPlace place = pkPlaceOf(tile, wave, g, (uint32_t)NS);
    const uint32_t bandX = place.band * (uint32_t)kBandW;
    const uint32_t tileY = place.strip0 * 2u;
    if (bandX >= A.w4)
        return; // (stacked waves share their band: the whole workgroup leaves)
    const bool rowsValid = tileY < A.h2;
    if (!rowsValid && !TURNED)
        return;
    BandCtx c;
    c.bandX = bandX;
    c.X = bandX + 4 * threadIdx.x;
    c.laneValid = c.X < A.w4;
    c.Xc = c.laneValid ? c.X : 0;
    c.cxb = A.cx0 + (int)(bandX >> 1);
    unsigned held[TURNED ? 2 * NS * 4 * PW : 1];
    if constexpr (TURNED) {
#pragma unroll
        for (int i = 0; i < 2 * NS * 4 * PW; ++i)
            held[i] = 0;
    }
    if (rowsValid) {
        TileRaw<YT, SUB, BIL, kNeedA, NS, 1> raw;
        loadTile<YT, SUB, BIL, kNeedA, NS, 1, false>(A, c, tileY, raw);
        f2(*rows)[kRowPitch] = reinterpret_cast<f2(*)[kRowPitch]>(reinterpret_cast<f2 *>(ldsBytes) + (size_t)wave * (BIL ? SR::kRows : 1) * kRowPitch);
        if constexpr (BIL) {
            stageTile<YT, SUB, kNeedA, NS, 1>(A, raw, rows);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        WideRowExchange * xchg = (sizeof(RT) == 2) ? reinterpret_cast<WideRowExchange *>(ldsBytes + LDS::kStageBytes) + wave : nullptr;
        computeTile<YT, SUB, BIL, RT, 4, APLANE, HASMUL, NS, 1, TURNED ? MAP_TURNED : MAP_ROWS>(A, c, tileY, raw, rows, xchg, held);
    }
    if constexpr (TURNED) {
        unsigned * tileWords = reinterpret_cast<unsigned *>(ldsBytes);
        __syncthreads(); // the tile overlays the chroma blocks and exchange buffers: every wave is done with them
#pragma unroll
        for (int rr = 0; rr < 2 * NS; ++rr) {
            unsigned px[4][PW];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int wd = 0; wd < PW; ++wd)
                    px[i][wd] = held[(rr * 4 + i) * PW + wd];
            mapTransposeWrite<PW>(tileWords, wave * (uint32_t)(2 * NS) + (uint32_t)rr, px);
        }
        __syncthreads();
        mapTransposeStore<PW>(A, tileWords, wave, bandX, tileY - wave * (uint32_t)(2 * NS));
    }
}

template <typename YT, int SUB, bool BIL, typename RT, bool APLANE, bool HASMUL, int NS, bool TURNED>
__global__ __launch_bounds__(256) void yuvToRgbTileSoloMappedKernel(TileArgs A, PkGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsMapped[]; // SoloMapLds<...>::kPlain bytes, kTurned for quarter turns
    runSoloMapped<YT, SUB, BIL, RT, APLANE, HASMUL, NS, TURNED>(A, g, ldsMapped);
}

template <typename YT, int SUB, bool BIL, typename RT, bool APLANE, bool HASMUL, int NS, bool TURNED>
__global__ __launch_bounds__(256) void yuvToRgbTileSoloMappedBatchKernel(const TileArgs * __restrict__ table, PkGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsMapped[];
    const TileArgs job = jobOf(table); // a private copy: see yuvToRgbTileBatchKernel
    runSoloMapped<YT, SUB, BIL, RT, APLANE, HASMUL, NS, TURNED>(job, g, ldsMapped);
}

template <typename YT, int SUB, bool BIL, typename RT, bool APLANE, bool MUL>
hipError_t launchSoloMapped(const TileLaunch & L)
{
    // quarter turns: the workgroup's rows must be the transposition tile's (tile_map_impl.h MapTile: 32 rows of 4-byte pixels, 16 of
    // 8-byte pixels), the four waves stacked, tiles numbered down the columns (tile_geom.h); rows: the automatic choice
    constexpr int kTurnNS = (sizeof(RT) == 1) ? 4 : 2;
    if (L.seq)
        return hipErrorNotSupported; // (a pixel map is per job)
    TileLaunch M = L;
    if (L.transposed) {
        M.pkStrips = kTurnNS, M.wavesXLog2 = 0;
        if (L.args)
            M.shiftStrips = turnShiftStrips(*L.args, 4 * (int)sizeof(RT), kTurnNS);
    }
    uint32_t nsw, blocks;
    PkGeom g;
    pkGeometry(M, M.maxW4, M.maxH2, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid(blocks, 1, L.count);
    if (L.transposed) {
        const uint32_t lds = SoloMapLds<YT, SUB, BIL, RT, kTurnNS>::kTurned;
        if (L.table)
            hipLaunchKernelGGL((yuvToRgbTileSoloMappedBatchKernel<YT, SUB, BIL, RT, APLANE, MUL, kTurnNS, true>), grid, block, lds, L.stream, L.table, g);
        else
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloMappedKernel<YT, SUB, BIL, RT, APLANE, MUL, kTurnNS, true>), grid, block, lds, L.stream, *L.args, g);
        return hipGetLastError();
    }
    const uint32_t lds4 = SoloMapLds<YT, SUB, BIL, RT, 4>::kPlain, lds2 = SoloMapLds<YT, SUB, BIL, RT, 2>::kPlain;
    if (L.table) {
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbTileSoloMappedBatchKernel<YT, SUB, BIL, RT, APLANE, MUL, 4, false>), grid, block, lds4, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbTileSoloMappedBatchKernel<YT, SUB, BIL, RT, APLANE, MUL, 2, false>), grid, block, lds2, L.stream, L.table, g);
    } else {
        if (nsw == 4)
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloMappedKernel<YT, SUB, BIL, RT, APLANE, MUL, 4, false>), grid, block, lds4, L.stream, *L.args, g);
        else
            AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloMappedKernel<YT, SUB, BIL, RT, APLANE, MUL, 2, false>), grid, block, lds2, L.stream, *L.args, g);
    }
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool MUL, int MULSEL>
hipError_t launchSoloSel(const TileLaunch & L, uint32_t nsw, const PkGeom & g, dim3 grid, dim3 block)
{
    if (L.table && L.streamLoads && sizeof(YT) == 2) { // batches of 16-bit planes beyond the Infinity Cache: streaming loads
        if constexpr (sizeof(YT) == 2) {
            if (nsw == 4)
                hipLaunchKernelGGL((yuvToRgbTileSoloBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 4, true, MULSEL>), grid, block, 0, L.stream, L.table, g);
            else
                hipLaunchKernelGGL((yuvToRgbTileSoloBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2, true, MULSEL>), grid, block, 0, L.stream, L.table, g);
        }
    } else if (L.table) {
        if (nsw == 4)
            hipLaunchKernelGGL((yuvToRgbTileSoloBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 4, false, MULSEL>), grid, block, 0, L.stream, L.table, g);
        else
            hipLaunchKernelGGL((yuvToRgbTileSoloBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2, false, MULSEL>), grid, block, 0, L.stream, L.table, g);
    } else {
        const SeqFrames S = L.seq ? *L.seq : seqOfOne(*L.args);
        const dim3 frames(grid.x, 1, L.seq ? L.seqCount : 1u);
        (void)S, (void)frames; // (seam-aware builds compile no single launches)
        if (L.streamLoads && sizeof(YT) == 2 && !BIL) {
            if constexpr (sizeof(YT) == 2 && !BIL) {
                if (nsw == 4)
                    AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 4, MULSEL, true>), frames, block, 0, L.stream, *L.args, g, S);
                else
                    AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2, MULSEL, true>), frames, block, 0, L.stream, *L.args, g, S);
            }
        } else {
            if (nsw == 4)
                AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 4, MULSEL>), frames, block, 0, L.stream, *L.args, g, S);
            else
                AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileSoloKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2, MULSEL>), frames, block, 0, L.stream, *L.args, g, S);
        }
    }
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool MUL>
hipError_t launchSolo(const TileLaunch & L)
{
    uint32_t nsw, blocks;
    PkGeom g;
    pkGeometry(L, L.maxW4, L.maxH2, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock);
    const dim3 grid = pkBatchGrid(g, blocks, L.count);
    if constexpr (MUL) {
        // jobs with pending alpha arithmetic: the kernel with that one mode compiled in (computeTile MULSEL; kernels_tile.hip alphaSelOf)
        switch (L.alphaSel) {
            case 1: return launchSoloSel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 1>(L, nsw, g, grid, block);
            case 2: return launchSoloSel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2>(L, nsw, g, grid, block);
            case 3: return launchSoloSel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 3>(L, nsw, g, grid, block);
            case 4: return launchSoloSel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 4>(L, nsw, g, grid, block);
            default: break;
        }
    }
    return launchSoloSel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 0>(L, nsw, g, grid, block);
}

template <typename YT, int SUB, bool BIL, typename RT, int NCH, bool APLANE, bool MUL>
hipError_t launchOne(const TileLaunch & L)
{
    if (L.solo)
        return launchSolo<YT, SUB, BIL, RT, NCH, APLANE, MUL>(L);
    if (L.seq)
        return hipErrorNotSupported; // (the cooperative runs know single jobs and tables)
    const dim3 block(kLanesX, kWavesPerBlock);
    // (tiles of one canvas: workgroups along the canvas rows -- tile_geom.h PkGeom::canvasColumns)
    const uint32_t columns = (L.table && L.canvasColumns > 1 && !L.mapped && L.count % L.canvasColumns == 0) ? L.canvasColumns : 0u;
    const dim3 grid = columns ? dim3(L.bandsPerJob * columns, L.runsPerJob, L.count / columns) : dim3(L.blocksPerJob, 1, L.count);
    if (L.table && L.streamLoads && sizeof(YT) == 2) {
        if constexpr (sizeof(YT) == 2) {
            if (L.stripsPerWave >= 2)
                hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2, true>), grid, block, 0, L.stream, L.table, L.tilesPerRun, columns);
            else
                hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 1, true>), grid, block, 0, L.stream, L.table, L.tilesPerRun, columns);
        }
    } else if (L.table && L.stripsPerWave >= 2)
        hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, L.table, L.tilesPerRun, columns);
    else if (L.table)
        hipLaunchKernelGGL((yuvToRgbTileBatchKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 1>), grid, block, 0, L.stream, L.table, L.tilesPerRun, columns);
    else if (L.stripsPerWave >= 2)
        AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 2>), grid, block, 0, L.stream, *L.args, L.tilesPerRun);
    else
        AVIFHIP_SINGLE_LAUNCH((yuvToRgbTileKernel<YT, SUB, BIL, RT, NCH, APLANE, MUL, 1>), grid, block, 0, L.stream, *L.args, L.tilesPerRun);
    return hipGetLastError();
}

template <typename YT, int SUB, bool BIL, typename RT>
hipError_t launchAlphaVariant(const TileKey & k, const TileLaunch & L)
{
    if (!k.gray && k.nch == 2) { // RGB565 (tileYuvToRgbSupported: 8-bit channels, no map; alpha arithmetic in the loop only: wave-private kernels)
        if constexpr (sizeof(RT) == 1)
            return k.hasMul ? launchSolo<YT, SUB, BIL, RT, 5, false, true>(L) : launchOne<YT, SUB, BIL, RT, 5, false, false>(L);
        else
            return hipErrorInvalidValue;
    }
    if constexpr (SUB == SUB_400) { // gray layouts read luma (and alpha) only, whatever the image's chroma layout: wave-private kernels
        if (k.nch == 1)
            return k.hasMul ? launchSolo<YT, SUB, BIL, RT, 1, false, true>(L) : launchSolo<YT, SUB, BIL, RT, 1, false, false>(L);
        if (k.nch == 2) {
            if (k.hasMul)
                return launchSolo<YT, SUB, BIL, RT, 2, true, true>(L);
            return k.alphaPlane ? launchSolo<YT, SUB, BIL, RT, 2, true, false>(L) : launchSolo<YT, SUB, BIL, RT, 2, false, false>(L);
        }
    }
    if (k.nch == 3)
        return k.hasMul ? launchOne<YT, SUB, BIL, RT, 3, false, true>(L) : launchOne<YT, SUB, BIL, RT, 3, false, false>(L);
    if (L.mapped) { // fused crop / rotate / mirror (4-channel pixels only: tileYuvToRgbSupported)
        if (k.hasMul)
            return launchSoloMapped<YT, SUB, BIL, RT, true, true>(L);
        return k.alphaPlane ? launchSoloMapped<YT, SUB, BIL, RT, true, false>(L) : launchSoloMapped<YT, SUB, BIL, RT, false, false>(L);
    }
    if (k.hasMul)
        return launchOne<YT, SUB, BIL, RT, 4, true, true>(L);
    return k.alphaPlane ? launchOne<YT, SUB, BIL, RT, 4, true, false>(L) : launchOne<YT, SUB, BIL, RT, 4, false, false>(L);
}

template <typename YT, int SUB, bool BIL>
hipError_t launchSubVariant(const TileKey & k, const TileLaunch & L)
{
    return k.wideRgb ? launchAlphaVariant<YT, SUB, BIL, uint16_t>(k, L) : launchAlphaVariant<YT, SUB, BIL, uint8_t>(k, L);
}

} // namespace AVIFHIP_TILE_BUILD
} // namespace tile
} // namespace avifhip
