// kernels_scale.hip -- plane scaling (avifImageScale, reference src/scale.c:23-201): one lane per destination sample,
// evaluated from the host-derived schedule (scale_plan.h).  Sample arithmetic of the vendored libyuv scaler: 7-bit column
// blend for 8-bit samples (scale_common.c:192-195), 16-bit column blend for 16-bit samples (:254-258), 8-bit row fractions
// (InterpolateRow_C, row_common.c:44-104), box sums with the reference's accumulator widths and fixed-point reciprocal
// (scale.c:29-150), 9:3:3:1 for the 2x upsamplers (scale_common.c:29-112).
#include <hip/hip_runtime.h>

#include "kernels.h"

// destination samples are written once and not read again by the kernel: streaming stores (A/B: -DAVIFHIP_SCALE_PLAIN_STORES)
#ifdef AVIFHIP_SCALE_PLAIN_STORES
#define SCALE_STORE(ptr, value) (*(ptr) = (value))
#else
#define SCALE_STORE(ptr, value) __builtin_nontemporal_store((value), (ptr))
#endif

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

// ---- row-staged kernel ---------------------------------------------------------------------------------------------
// The gather kernel above issues one byte-sized memory instruction per source sample and per destination sample, and a
// wave's memory instruction costs the same address-path cycles whether it moves 64 bytes or 1 KiB: it runs at 6-11 % of
// the HBM roofline.  Here a wave owns 256 consecutive destination columns (4 per lane) of `rowsPerWave` destination rows.
// It first copies every source-row segment those samples need into a wave-private LDS block with 16-byte loads (all
// issued before the first is waited for), then gathers from LDS (whose address path is not the bottleneck) and stores 4
// samples per lane at once.  No workgroup barrier: a wave's LDS accesses execute in program order; the fences only
// restrain the compiler.  The host (scaleStagedPlan, api_scale.cpp) guarantees the block fits.  The chunks are 16-byte ALIGNED
// pieces of the address space, so the first / last chunk of a segment may begin before / end after the bytes asked for
// (even before the first or after the last byte of the plane): every chunk contains at least one byte of the segment, an
// aligned 16-byte chunk never straddles a page, hence the load touches no page the plane does not own; the extra bytes
// are never used.  All planes of an image are scaled by ONE launch (blockIdx.z): the launch-to-first-store latency of a
// kernel of this shape is as long as its streaming time, three of them back to back tripled it.
constexpr int kStagedCols = 256;

typedef uint32_t u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int waveMin(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int waveMax(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        v = max(v, __shfl_xor(v, m));
    return v;
}

template <bool WIDE>
__device__ __forceinline__ int ldsSample(const uint8_t * stage, uint32_t off)
{
    return WIDE ? (int)*reinterpret_cast<const uint16_t *>(stage + off) : (int)stage[off];
}

template <bool WIDE, int MODE>
__device__ __forceinline__ void scalePlaneStaged(const ScaleArgs & A, int rowsPerWave, int rowsCap, uint32_t segPitch, int boxWidth, u4v * stageAll)
{
    constexpr uint32_t kBps = WIDE ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = blockIdx.x * kStagedCols + lane * 4;
    const int jBase = (blockIdx.y * 4 + wv) * rowsPerWave;
    if (jBase >= A.dstH || (int)(blockIdx.x * kStagedCols) >= A.dstW) // the grid covers the largest plane of the launch
        return;
    const int jEnd = min(jBase + rowsPerWave, A.dstH);
    uint8_t * stage = reinterpret_cast<uint8_t *>(stageAll) + (uint32_t)wv * (uint32_t)rowsCap * segPitch;

    // the lane's four destination columns (clamped copies beyond the last column keep the loads in range)
    int ca[4], cb[4], c1[4];
    int lo = 0x7fffffff, hi = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = min(i0 + s, A.dstW - 1);
        ca[s] = A.colA[i], cb[s] = A.colB[i];
        c1[s] = (MODE == SCALE_UP2_MODE) ? cb[s] : (MODE == SCALE_BOX_MODE) ? ca[s] + cb[s] - 1 : (MODE == SCALE_POINT_MODE) ? ca[s] : min(ca[s] + 1, A.srcW - 1);
        lo = min(lo, min(ca[s], c1[s])), hi = max(hi, max(ca[s], c1[s]));
    }
    lo = __builtin_amdgcn_readfirstlane(waveMin(lo)), hi = __builtin_amdgcn_readfirstlane(waveMax(hi));
    const uint32_t loByte = (uint32_t)lo * kBps, segBytes = (uint32_t)(hi + 1 - lo) * kBps;

    // the row schedule of the wave's destination rows: lane l holds row jBase + l (rowsPerWave <= 64), so that the row loop
    // below broadcasts from registers instead of waiting for a dependent global load per row
    const int jMine = min(jBase + lane, A.dstH - 1);
    const int raMine = A.rowA[jMine], rbMine = A.rowB[jMine], rfMine = A.rowF[jMine];
    const int lastMine = (MODE == SCALE_BOX_MODE) ? raMine + rbMine - 1 : (MODE == SCALE_POINT_MODE) ? raMine : rbMine;
    const bool rowMine = jBase + lane < jEnd;
    const int rLo = __builtin_amdgcn_readfirstlane(waveMin(rowMine ? min(raMine, lastMine) : 0x7fffffff));
    const int rHi = __builtin_amdgcn_readfirstlane(waveMax(rowMine ? max(raMine, lastMine) : 0));
    const int nRows = rHi - rLo + 1;

    // ---- stage: rows rLo..rHi, bytes [loByte, loByte + segBytes), as aligned 16-byte chunks of absolute addresses ----
    const uint32_t chunksMax = (15u + segBytes + 15u) >> 4;
    for (uint32_t k0 = 0; k0 < chunksMax; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        for (int r0 = 0; r0 < nRows; r0 += 4) {
            u4v v[4];
            bool on[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uintptr_t first = (uintptr_t)(A.src + (size_t)(rLo + r0 + q) * A.srcPitch + loByte);
                const uint32_t shift = (uint32_t)(first & 15u);
                on[q] = (r0 + q < nRows) && k < ((shift + segBytes + 15u) >> 4);
                if (on[q])
                    v[q] = *reinterpret_cast<const u4v *>((first & ~(uintptr_t)15) + 16u * k);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (on[q])
                    *reinterpret_cast<u4v *>(stage + (uint32_t)(r0 + q) * segPitch + 16u * k) = v[q];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // per-lane byte offsets of the two source columns of each sample inside a staged row; the uniform part of an address
    // (row, alignment shift of that row) is added per row
    uint32_t oA[4], oB[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        oA[s] = (uint32_t)(ca[s] - lo) * kBps, oB[s] = (uint32_t)(c1[s] - lo) * kBps;
    const uint32_t srcLo = (uint32_t)(uintptr_t)A.src + loByte; // low address bits decide the 16-byte phase
    auto rowBase = [&](int r) -> uint32_t { return (uint32_t)(r - rLo) * segPitch + ((srcLo + (uint32_t)r * A.srcPitch) & 15u); };

    for (int j = jBase; j < jEnd; ++j) {
        const int ra = __builtin_amdgcn_readlane(raMine, j - jBase), rb = __builtin_amdgcn_readlane(rbMine, j - jBase);
        const int rf = __builtin_amdgcn_readlane(rfMine, j - jBase);
        int out[4];
        if constexpr (MODE == SCALE_BOX_MODE) {
            uint32_t sum[4] = { 0, 0, 0, 0 };
            // 8-bit boxes of one width 4 / 8 whose first byte is dword-aligned in the staged rows (uniform: segment start and
            // row pitch multiples of 4): the lane's 4 boxes are `boxWidth` consecutive dwords of each row, summed a dword at a
            // time by v_sad_u8 against zero -- 1-2 instructions per box and row instead of a read and an add per sample
            const bool dwords = !WIDE && boxWidth && (((uint32_t)(uintptr_t)A.src | A.srcPitch | loByte) & 3u) == 0;
            if (dwords) {
                for (int r = ra; r < ra + rb; ++r) {
                    const uint32_t * row = reinterpret_cast<const uint32_t *>(stage + rowBase(r) + oA[0]);
                    if (boxWidth == 4) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            sum[s] = __builtin_amdgcn_sad_u8(row[s], 0u, sum[s]);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            sum[s] = __builtin_amdgcn_sad_u8(row[2 * s + 1], 0u, __builtin_amdgcn_sad_u8(row[2 * s], 0u, sum[s]));
                    }
                }
            } else {
                for (int r = ra; r < ra + rb; ++r) {
                    const uint8_t * row = stage + rowBase(r);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        for (uint32_t o = oA[s]; o <= oB[s]; o += kBps)
                            sum[s] += (uint32_t)ldsSample<WIDE>(row, o);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t scale = (uint32_t)(65536 / (cb[s] * rb));
                out[s] = WIDE ? (int)(uint16_t)((sum[s] * scale) >> 16) : (int)(uint8_t)((sum[s] * scale) >> 16);
            }
        } else {
            const uint8_t * rowA = stage + rowBase(ra);
            const uint8_t * rowB = stage + rowBase(MODE == SCALE_POINT_MODE ? ra : rb);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int a0 = ldsSample<WIDE>(rowA, oA[s]);
                if constexpr (MODE == SCALE_POINT_MODE) {
                    out[s] = a0;
                } else {
                    const int a1 = ldsSample<WIDE>(rowA, oB[s]);
                    const int b0 = ldsSample<WIDE>(rowB, oA[s]), b1 = ldsSample<WIDE>(rowB, oB[s]);
                    if constexpr (MODE == SCALE_DOWN_MODE)
                        out[s] = blendColumns<WIDE>(blendRows(a0, b0, rf), blendRows(a1, b1, rf), cb[s]);
                    else if constexpr (MODE == SCALE_UP_MODE)
                        out[s] = blendRows(blendColumns<WIDE>(a0, a1, cb[s]), blendColumns<WIDE>(b0, b1, cb[s]), rf);
                    else
                        out[s] = (9 * a0 + 3 * a1 + 3 * b0 + b1 + 8) >> 4;
                }
            }
        }

        uint8_t * d = A.dst + (size_t)j * A.dstPitch + (size_t)i0 * kBps;
        const bool aligned = (((uintptr_t)(A.dst + (size_t)j * A.dstPitch)) & (WIDE ? 7u : 3u)) == 0; // uniform
        if (i0 + 3 < A.dstW && aligned) {
            if (WIDE) {
                uint2 w = { (uint32_t)out[0] | ((uint32_t)out[1] << 16), (uint32_t)out[2] | ((uint32_t)out[3] << 16) };
                {
                    typedef unsigned u2v __attribute__((ext_vector_type(2)));
                    SCALE_STORE(reinterpret_cast<u2v *>(d), ((u2v) { w.x, w.y }));
                }
            } else {
                SCALE_STORE(reinterpret_cast<uint32_t *>(d), (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24));
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (i0 + s < A.dstW) {
                    if (WIDE)
                        reinterpret_cast<uint16_t *>(d)[s] = (uint16_t)out[s];
                    else
                        d[s] = (uint8_t)out[s];
                }
            }
        }
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void scalePlanesStagedKernel(ScaleStagedLaunch L)
{
    extern __shared__ u4v stageAll[]; // 4 waves x rowsCap staged rows x segPitch bytes
    const int p = blockIdx.z;
    const ScaleArgs & A = L.plane[p];
    const int rpw = L.staging[p].rowsPerWave, cap = L.staging[p].rowsCap, bw = L.staging[p].boxWidth;
    const uint32_t pitch = L.staging[p].segPitch;
    switch (A.mode) { // uniform
        case SCALE_POINT_MODE: scalePlaneStaged<WIDE, SCALE_POINT_MODE>(A, rpw, cap, pitch, 0, stageAll); break;
        case SCALE_DOWN_MODE: scalePlaneStaged<WIDE, SCALE_DOWN_MODE>(A, rpw, cap, pitch, 0, stageAll); break;
        case SCALE_UP_MODE: scalePlaneStaged<WIDE, SCALE_UP_MODE>(A, rpw, cap, pitch, 0, stageAll); break;
        case SCALE_BOX_MODE: scalePlaneStaged<WIDE, SCALE_BOX_MODE>(A, rpw, cap, pitch, bw, stageAll); break;
        default: scalePlaneStaged<WIDE, SCALE_UP2_MODE>(A, rpw, cap, pitch, 0, stageAll); break;
    }
}

// ---- window kernel (8-bit samples, point / bilinear / 2x modes) ---------------------------------------------------
// When the source columns of a lane's four destination samples span at most 8 bytes (every upscale, downscales up to
// 1.75x; host-checked), the lane reads them as ONE 8-byte window per source row straight from global memory
// (neighbouring lanes' windows overlap or abut, so a wave's load covers one contiguous piece of the row), v_perm_b32
// with per-lane selectors computed once spreads them into packed 16-bit pairs, and the reference's integer arithmetic
// runs two samples per instruction (v_pk_*_u16 / _i16: every intermediate of the 8-bit formulas fits 16 bits).  No LDS,
// no staging prologue, ~6 VALU instructions per destination sample instead of ~25.
// What bounds it (profiles/r01_scale_pmc.txt: TA_BUSY 70-80 % of the kernel): the CU's vector-memory path handles a
// wave's load or store at 4 lanes per clock WHATEVER the lanes' access size, so the 4-byte-per-lane stores and the
// 8+4-byte window loads run that path at a quarter of what 16-byte accesses reach.  (16 destination samples per lane
// with 16-byte stores needs windows of up to 32 bytes, which v_perm_b32 cannot address; unaligned 8-byte LDS windows
// over a staged row were measured 2.4x SLOWER than this kernel.)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

// The 8-byte window at byte address p (any alignment) from DWORD-ALIGNED loads + v_alignbyte (misaligned vector loads
// work on gfx950 but are slower).  Reads the aligned dwords d0, d1 that contain the window's first bytes and, only when
// the window is not itself aligned, the dword after them (otherwise d1 again), so that every dword touched contains at
// least one byte of the window: no page is touched that the row does not own.
__device__ __forceinline__ u2v loadWindow(const uint8_t * p)
{
    const uintptr_t a = (uintptr_t)p;
    const uint32_t sh = (uint32_t)a & 3u;
    const uint32_t * q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[sh ? 2 : 1];
    return (u2v) { __builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh) };
}

__device__ __forceinline__ us2 asPair(uint32_t v)
{
    return __builtin_bit_cast(us2, v);
}
__device__ __forceinline__ us2 splatPair(int v)
{
    return (us2) { (unsigned short)v, (unsigned short)v };
}
// blendRows on two samples at once; yf is uniform
__device__ __forceinline__ us2 blendRowsPair(us2 a, us2 b, int yf)
{
    if (yf == 0)
        return a;
    if (yf == 128)
        return (a + b + splatPair(1)) >> splatPair(1);
    return (a * splatPair(256 - yf) + b * splatPair(yf) + splatPair(128)) >> splatPair(8);
}
// blendColumns (8-bit samples) on two samples at once: a + (((f >> 9) * (b - a)) + 0x40) >> 7, which lies between a and b
__device__ __forceinline__ us2 blendColumnsPair(us2 a, us2 b, ss2 f7)
{
    const ss2 d = __builtin_bit_cast(ss2, b) - __builtin_bit_cast(ss2, a);
    const ss2 m = (f7 * d + (ss2) { 0x40, 0x40 }) >> (ss2) { 7, 7 };
    return __builtin_bit_cast(us2, __builtin_bit_cast(ss2, a) + m);
}

template <int MODE>
__device__ __forceinline__ void scalePlaneWindow(const ScaleArgs & A, int rowsPerWave)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = blockIdx.x * kStagedCols + lane * 4;
    const int jBase = (blockIdx.y * 4 + wv) * rowsPerWave;
    if (jBase >= A.dstH || (int)(blockIdx.x * kStagedCols) >= A.dstW)
        return;
    const int jEnd = min(jBase + rowsPerWave, A.dstH);

    // the lane's four destination columns.  The column tables are padded with copies of their last entry up to the next multiple of 16 + 16
    // (api_scale.cpp) -- NOT up to the end of the block's 256 columns: lanes past the plane's last column take the entries of the last group of
    // four that has one.  (Until round 6 they read colA[i0 ..] unclamped: beyond the padding for every plane narrower than its block, i.e. into
    // the tables behind or past the upload into whatever the scratch allocation held before.  The entries decide the lane's load address: zeros
    // -- fresh device memory -- are harmless, pixels of a buffer another context has just freed are a "Memory access fault by GPU" a gigabyte
    // below the plane.  That was the open fault of round 5: gain maps scaled after the device farm had recycled its buffers.)
    const int iTable = min(i0, (A.dstW - 1) & ~3);
    const int4 ca4 = *reinterpret_cast<const int4 *>(A.colA + iTable), cb4 = *reinterpret_cast<const int4 *>(A.colB + iTable);
    const int ca[4] = { ca4.x, ca4.y, ca4.z, ca4.w }, cb[4] = { cb4.x, cb4.y, cb4.z, cb4.w };
    int c1[4];
    int cmin = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        c1[s] = (MODE == SCALE_UP2_MODE) ? cb[s] : (MODE == SCALE_POINT_MODE) ? ca[s] : min(ca[s] + 1, A.srcW - 1);
        cmin = min(cmin, min(ca[s], c1[s]));
    }
    const int wstart = min(cmin, A.srcW - 8); // the window never leaves the row (srcW >= 8, host-checked)
    // selectors: byte k of the window = selector value k; 0x0c = constant zero
    auto pairSel = [&](int x, int y) -> uint32_t { return (uint32_t)(x - wstart) | 0x0c00u | ((uint32_t)(y - wstart) << 16) | 0x0c000000u; };
    const uint32_t selNearLo = pairSel(ca[0], ca[1]), selNearHi = pairSel(ca[2], ca[3]);
    const uint32_t selFarLo = pairSel(c1[0], c1[1]), selFarHi = pairSel(c1[2], c1[3]);
    const ss2 f7Lo = { (short)(cb[0] >> 9), (short)(cb[1] >> 9) }, f7Hi = { (short)(cb[2] >> 9), (short)(cb[3] >> 9) };

    // lane l holds the row schedule of destination row jBase + l
    const int jMine = min(jBase + lane, A.dstH - 1);
    const int raMine = A.rowA[jMine], rbMine = A.rowB[jMine], rfMine = A.rowF[jMine];

    // four destination rows per step: their window loads are issued before the first is used
    constexpr int kStep = 4;
    for (int j0 = jBase; j0 < jEnd; j0 += kStep) {
        u2v wa[kStep], wb[kStep];
        int rf[kStep];
#pragma unroll
        for (int q = 0; q < kStep; ++q) {
            const int jj = min(j0 + q, jEnd - 1) - jBase; // rows past the end repeat the last one (loads only)
            const int ra = __builtin_amdgcn_readlane(raMine, jj), rb = __builtin_amdgcn_readlane(rbMine, jj);
            rf[q] = __builtin_amdgcn_readlane(rfMine, jj);
            wa[q] = loadWindow(A.src + (size_t)ra * A.srcPitch + wstart);
            if constexpr (MODE != SCALE_POINT_MODE)
                wb[q] = loadWindow(A.src + (size_t)rb * A.srcPitch + wstart);
        }
#pragma unroll
        for (int q = 0; q < kStep; ++q) {
            const int j = j0 + q;
            if (j >= jEnd)
                break;
            us2 outLo, outHi;
            if constexpr (MODE == SCALE_POINT_MODE) {
                outLo = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selNearLo)), outHi = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selNearHi));
            } else {
                const us2 anLo = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selNearLo)), anHi = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selNearHi));
                const us2 afLo = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selFarLo)), afHi = asPair(__builtin_amdgcn_perm(wa[q].y, wa[q].x, selFarHi));
                const us2 bnLo = asPair(__builtin_amdgcn_perm(wb[q].y, wb[q].x, selNearLo)), bnHi = asPair(__builtin_amdgcn_perm(wb[q].y, wb[q].x, selNearHi));
                const us2 bfLo = asPair(__builtin_amdgcn_perm(wb[q].y, wb[q].x, selFarLo)), bfHi = asPair(__builtin_amdgcn_perm(wb[q].y, wb[q].x, selFarHi));
                if constexpr (MODE == SCALE_UP2_MODE) {
                    outLo = (anLo * splatPair(9) + (afLo + bnLo) * splatPair(3) + bfLo + splatPair(8)) >> splatPair(4);
                    outHi = (anHi * splatPair(9) + (afHi + bnHi) * splatPair(3) + bfHi + splatPair(8)) >> splatPair(4);
                } else if constexpr (MODE == SCALE_DOWN_MODE) {
                    outLo = blendColumnsPair(blendRowsPair(anLo, bnLo, rf[q]), blendRowsPair(afLo, bfLo, rf[q]), f7Lo);
                    outHi = blendColumnsPair(blendRowsPair(anHi, bnHi, rf[q]), blendRowsPair(afHi, bfHi, rf[q]), f7Hi);
                } else {
                    outLo = blendRowsPair(blendColumnsPair(anLo, afLo, f7Lo), blendColumnsPair(bnLo, bfLo, f7Lo), rf[q]);
                    outHi = blendRowsPair(blendColumnsPair(anHi, afHi, f7Hi), blendColumnsPair(bnHi, bfHi, f7Hi), rf[q]);
                }
            }
            const uint32_t lo = __builtin_bit_cast(uint32_t, outLo), hi = __builtin_bit_cast(uint32_t, outHi);
            uint8_t * d = A.dst + (size_t)j * A.dstPitch + (size_t)i0;
            const bool aligned = (((uintptr_t)(A.dst + (size_t)j * A.dstPitch)) & 3u) == 0; // uniform
            if (i0 + 3 < A.dstW && aligned) {
                SCALE_STORE(reinterpret_cast<uint32_t *>(d), __builtin_amdgcn_perm(hi, lo, 0x06040200u));
            } else {
                const uint32_t out[4] = { lo & 0xffu, (lo >> 16) & 0xffu, hi & 0xffu, (hi >> 16) & 0xffu };
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (i0 + s < A.dstW)
                        d[s] = (uint8_t)out[s];
            }
        }
    }
}

// ---- 2x on both axes (ScalePlaneUp2_Bilinear, scale.c:500-528; ScaleRowUp2_Bilinear_C: (9 near + 3 + 3 + 1 diagonal + 8) >> 4 with the
//      neighbours of edge samples duplicated; the last destination column takes far = near, scale_plan.cpp upsample2Axis) ----
// The window kernel spends a vector-memory instruction per 4 destination samples; here a lane owns 8 source columns = 16 destination
// columns: one 8-byte load (+ two halo bytes) per SOURCE row, one 16-byte streaming store per DESTINATION row.  The filter is separable in
// exact integers: H = 3 near + far + 2 per source row (<= 1022: 16-bit pairs, v_pk_mad_u16), out = (3 H_near + H_far) >> 4 -- the +2s sum up
// to the +8.  A wave slides over kDoubleRows source rows with H of three rows in registers.
#ifndef AVIFHIP_DOUBLE_ROWS
#define AVIFHIP_DOUBLE_ROWS 4
#endif
constexpr int kDoubleRows = AVIFHIP_DOUBLE_ROWS, kDoubleCols = 512;

__device__ __forceinline__ unsigned pkMad3(unsigned a, unsigned c) // a * 3 + c on both 16-bit halves
{
    unsigned d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(0x00030003u), "v"(c));
    return d;
}

struct DoubledRow
{
    unsigned h[8]; // (H of destination column 2i | H of column 2i + 1) for the lane's source columns i = 0 .. 7
};

__device__ __forceinline__ DoubledRow doubleRowHorizontally(const ScaleArgs & A, int row, int c0, bool fullLane)
{
    const uint8_t * src = A.src + (size_t)row * A.srcPitch;
    unsigned w0, w1;
    if (fullLane) {
        const uint2 t = *reinterpret_cast<const uint2 *>(src + c0);
        w0 = t.x, w1 = t.y;
    } else { // the lane holding the row's last columns: missing ones repeat the last sample
        w0 = w1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned v = src[min(c0 + i, A.srcW - 1)];
            if (i < 4)
                w0 |= v << (8 * i);
            else
                w1 |= v << (8 * (i - 4));
        }
    }
    const unsigned sl = src[max(c0 - 1, 0)], sr = src[min(c0 + 8, A.srcW - 1)];
    DoubledRow R;
    // v_perm_b32: selector bytes 0..3 address the second operand, 4..7 the first, 12 is zero
    const unsigned f0 = __builtin_amdgcn_perm(w0, sl, 0x0c050c00u);  // (s[-1] | s[1] << 16)
    const unsigned f7 = __builtin_amdgcn_perm(sr, w1, 0x0c040c02u);  // (s[6] | s[8] << 16)
    unsigned f[8] = { f0, 0, 0, 0, 0, 0, 0, f7 };
#pragma unroll
    for (int i = 1; i < 7; ++i)
        f[i] = __builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (unsigned)(i - 1) | ((unsigned)(i + 1) << 16));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned s2 = __builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (unsigned)i | ((unsigned)i << 16)); // (s[i] | s[i] << 16)
        R.h[i] = pkMad3(s2, f[i]) + 0x00020002u;
    }
    // the last destination column of an odd width is an even one whose far neighbour is itself (upsample2Axis: lastIsEdge)
    if (A.dstW & 1) {
        const int iLast = A.srcW - 1 - c0;
        if (iLast >= 0 && iLast < 8) {
            const unsigned v = src[A.srcW - 1];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == iLast)
                    R.h[i] = (R.h[i] & 0xffff0000u) | (4u * v + 2u);
        }
    }
    return R;
}

__global__ __launch_bounds__(256) void scalePlanesDoublingKernel(ScaleStagedLaunch L)
{
    const ScaleArgs & A = L.plane[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (int)blockIdx.x * kDoubleCols + 8 * lane;
    const int j0 = ((int)blockIdx.y * 4 + wave) * kDoubleRows;
    if (c0 >= A.srcW || j0 >= A.srcH)
        return;
    const bool fullLane = c0 + 8 <= A.srcW;
    const int d0 = 2 * c0;
    const bool fullStore = d0 + 16 <= A.dstW;
    DoubledRow prev = doubleRowHorizontally(A, max(j0 - 1, 0), c0, fullLane);
    DoubledRow cur = doubleRowHorizontally(A, j0, c0, fullLane);
#pragma unroll
    for (int r = 0; r < kDoubleRows; ++r) {
        const int j = j0 + r;
        if (j >= A.srcH) // wave-uniform
            break;
        const DoubledRow next = doubleRowHorizontally(A, min(j + 1, A.srcH - 1), c0, fullLane);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int dj = 2 * j + half;
            if (dj >= A.dstH)
                break;
            const DoubledRow & far = half ? next : prev;
            unsigned o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, pkMad3(cur.h[i], far.h[i])) >> splatPair(4));
            typedef unsigned u4s __attribute__((ext_vector_type(4)));
            const u4s px = { __builtin_amdgcn_perm(o[1], o[0], 0x06040200u), __builtin_amdgcn_perm(o[3], o[2], 0x06040200u),
                             __builtin_amdgcn_perm(o[5], o[4], 0x06040200u), __builtin_amdgcn_perm(o[7], o[6], 0x06040200u) };
            uint8_t * d = A.dst + (size_t)dj * A.dstPitch + (size_t)d0;
            if (fullStore) {
                __builtin_nontemporal_store(px, reinterpret_cast<u4s *>(d));
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (d0 + k < A.dstW)
                        d[k] = (uint8_t)(px[k >> 2] >> (8 * (k & 3)));
            }
        }
        prev = cur, cur = next;
    }
}

// ---- exact N x N boxes, N in {4, 8} (thumbnails at exactly 1/4 or 1/8: ScalePlaneBox with every box on the N-grid; ScaleAddRow sums rows
//      in 16 bits -- no wrap below 258 rows -- and ScaleAddCols multiplies by 65536 / N^2, an exact shift) ----
// A lane owns 4 destination samples: N / 4 16-byte loads per source row (1 or 2 KiB contiguous per wave instruction, streaming), N rows,
// one v_sad_u8 per dword, one dword store.  No LDS, no tables.  The row-staged kernel copies the same bytes through LDS first.
template <int N>
__device__ __forceinline__ void scalePlaneExactBox(const ScaleArgs & A)
{
    typedef unsigned u4s __attribute__((ext_vector_type(4)));
    constexpr int kLoads = N / 4;     // 16-byte loads per source row
    constexpr int kRows = 8 / N;      // destination rows per wave: 8 source rows (32 / 64 registers of pixels) in flight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (int)blockIdx.x * 256 + 4 * lane;
    const int j0 = ((int)blockIdx.y * 4 + wave) * kRows;
    if (i0 >= A.dstW || j0 >= A.dstH)
        return;
    const bool whole = i0 + 4 <= A.dstW;
    u4s raw[kRows][N][kLoads];
    if (whole) {
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const int j = min(j0 + q, A.dstH - 1); // rows past the end repeat the last one (loads only)
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int h = 0; h < kLoads; ++h)
                    raw[q][r][h] = *reinterpret_cast<const u4s *>(A.src + (size_t)(j * N + r) * A.srcPitch + (size_t)i0 * N + 16 * h);
        }
    }
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int j = j0 + q;
        if (j >= A.dstH) // wave-uniform
            break;
        uint32_t sum[4] = { 0, 0, 0, 0 };
        if (whole) {
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (N == 4) {
                        sum[s] = __builtin_amdgcn_sad_u8(raw[q][r][0][s], 0u, sum[s]);
                    } else {
                        const u4s & t = raw[q][r][s >> 1];
                        sum[s] = __builtin_amdgcn_sad_u8(t[2 * (s & 1) + 1], 0u, __builtin_amdgcn_sad_u8(t[2 * (s & 1)], 0u, sum[s]));
                    }
                }
            }
            constexpr int kShift = (N == 4) ? 4 : 6; // (sum * (65536 / N^2)) >> 16
            *reinterpret_cast<uint32_t *>(A.dst + (size_t)j * A.dstPitch + (size_t)i0) =
                (sum[0] >> kShift) | ((sum[1] >> kShift) << 8) | ((sum[2] >> kShift) << 16) | ((sum[3] >> kShift) << 24);
        } else { // the row's last, partial group of destination samples
            for (int s = 0; i0 + s < A.dstW; ++s) {
                uint32_t t = 0;
                for (int r = 0; r < N; ++r)
                    for (int c = 0; c < N; ++c)
                        t += A.src[(size_t)(j * N + r) * A.srcPitch + (size_t)(i0 + s) * N + c];
                A.dst[(size_t)j * A.dstPitch + (size_t)(i0 + s)] = (uint8_t)((t * (65536u / (N * N))) >> 16);
            }
        }
    }
}

__global__ __launch_bounds__(256) void scalePlanesExactBoxKernel(ScaleStagedLaunch L)
{
    const ScaleArgs & A = L.plane[blockIdx.z];
    if (L.staging[blockIdx.z].boxWidth == 4) // uniform
        scalePlaneExactBox<4>(A);
    else
        scalePlaneExactBox<8>(A);
}

__global__ __launch_bounds__(256) void scalePlanesWindowKernel(ScaleStagedLaunch L)
{
    const int p = blockIdx.z;
    const ScaleArgs & A = L.plane[p];
    const int rpw = L.staging[p].rowsPerWave;
    switch (A.mode) { // uniform
        case SCALE_POINT_MODE: scalePlaneWindow<SCALE_POINT_MODE>(A, rpw); break;
        case SCALE_DOWN_MODE: scalePlaneWindow<SCALE_DOWN_MODE>(A, rpw); break;
        case SCALE_UP_MODE: scalePlaneWindow<SCALE_UP_MODE>(A, rpw); break;
        default: scalePlaneWindow<SCALE_UP2_MODE>(A, rpw); break;
    }
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

bool scaleExactBoxCovers(const ScaleArgs & A)
{
    return A.mode == SCALE_BOX_MODE && (((uintptr_t)A.src | A.srcPitch) & 15u) == 0 && (((uintptr_t)A.dst | A.dstPitch) & 3u) == 0;
}

hipError_t launchScalePlanesExactBox(const ScaleStagedLaunch & L, hipStream_t stream)
{
    if (L.count <= 0)
        return hipSuccess;
    unsigned gx = 1, gy = 1;
    for (int p = 0; p < L.count; ++p) {
        const ScaleArgs & A = L.plane[p];
        const int rowsPerGroup = 4 * (8 / L.staging[p].boxWidth);
        const unsigned bx = (unsigned)(A.dstW + 255) / 256, by = (unsigned)(A.dstH + rowsPerGroup - 1) / rowsPerGroup;
        gx = bx > gx ? bx : gx, gy = by > gy ? by : gy;
    }
    hipLaunchKernelGGL(scalePlanesExactBoxKernel, dim3(gx, gy, (unsigned)L.count), dim3(256), 0, stream, L);
    return hipGetLastError();
}

bool scaleDoublingCovers(const ScaleArgs & A)
{
    return A.mode == SCALE_UP2_MODE && A.srcW >= 8 && (((uintptr_t)A.src | A.srcPitch) & 3u) == 0 && (((uintptr_t)A.dst | A.dstPitch) & 15u) == 0;
}

hipError_t launchScalePlanesDoubling(const ScaleStagedLaunch & L, hipStream_t stream)
{
    if (L.count <= 0)
        return hipSuccess;
    unsigned gx = 1, gy = 1;
    for (int p = 0; p < L.count; ++p) {
        const ScaleArgs & A = L.plane[p];
        const unsigned bx = (unsigned)(A.srcW + kDoubleCols - 1) / kDoubleCols, by = (unsigned)(A.srcH + 4 * kDoubleRows - 1) / (4 * kDoubleRows);
        gx = bx > gx ? bx : gx, gy = by > gy ? by : gy;
    }
    hipLaunchKernelGGL(scalePlanesDoublingKernel, dim3(gx, gy, (unsigned)L.count), dim3(256), 0, stream, L);
    return hipGetLastError();
}

hipError_t launchScalePlanesStaged(const ScaleStagedLaunch & L, bool wide, bool window, hipStream_t stream)
{
    if (L.count <= 0)
        return hipSuccess;
    unsigned gx = 1, gy = 1;
    size_t lds = 0;
    for (int p = 0; p < L.count; ++p) {
        const ScaleArgs & A = L.plane[p];
        const ScaleStaging & st = L.staging[p];
        const unsigned segs = (unsigned)(A.dstW + kStagedCols - 1) / kStagedCols, groups = (unsigned)(A.dstH + 4 * st.rowsPerWave - 1) / (4 * st.rowsPerWave);
        gx = segs > gx ? segs : gx, gy = groups > gy ? groups : gy;
        const size_t need = window ? 0 : (size_t)4 * st.rowsCap * st.segPitch;
        lds = need > lds ? need : lds;
    }
    const dim3 grid(gx, gy, (unsigned)L.count), block(256);
    if (window)
        hipLaunchKernelGGL(scalePlanesWindowKernel, grid, block, 0, stream, L);
    else if (wide)
        hipLaunchKernelGGL(scalePlanesStagedKernel<true>, grid, block, lds, stream, L);
    else
        hipLaunchKernelGGL(scalePlanesStagedKernel<false>, grid, block, lds, stream, L);
    return hipGetLastError();
}

} // namespace avifhip
