// kernels_r2y_tile.hip -- host-side dispatch of the bandwidth-tuned RGB -> YUV kernels (r2y_tile_impl.h): which plans
// they cover, how the image is cut into strips, and the hand-over of the at most 3 columns / 1 row that do not fill a
// 4x2 pixel group to the universal kernel.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "r2y_tile_shared.h"

namespace avifhip {

using namespace r2y;

namespace {

bool alignedTo(const void * ptr, uint32_t rowBytes, uint32_t a)
{
    return ((uintptr_t)ptr % a) == 0 && (rowBytes % a) == 0;
}
bool fits32(uint64_t rows, uint32_t pitch)
{
    return rows * (uint64_t)pitch < ((uint64_t)1 << 32);
}

R2YKey keyFor(const RgbToYuvPlan & p)
{
    R2YKey k;
    k.fixedPoint = p.arith == ARITH_LIBYUV;
    k.wideRgb = p.rgb.chanBytes == 2;
    k.wideYuv = p.yuv.chanBytes == 2;
    k.nch = p.rgb.hasAlpha ? 4 : 3;
    k.hasMul = p.mul != MUL_NONE;
    switch (p.yuv.format) {
        case AVIF_PIXEL_FORMAT_YUV444: k.sub = SUB_444; break;
        case AVIF_PIXEL_FORMAT_YUV422: k.sub = SUB_422; break;
        case AVIF_PIXEL_FORMAT_YUV420: k.sub = SUB_420; break;
        default: k.sub = SUB_400; break;
    }
    return k;
}

const char * kernelNameFor(const R2YKey & k)
{
    static thread_local char name[96];
    static const char * subs[] = { "444", "422", "420", "400" };
    snprintf(name, sizeof(name), "%s<%s%d,%s,%s%s>", k.fixedPoint ? "rgb2yuv_fixed_tile" : "rgb2yuv_tile", k.nch == 4 ? "rgba" : "rgb", k.wideRgb ? 16 : 8,
             k.wideYuv ? "u16" : "u8", subs[k.sub], k.hasMul ? ",alphamul" : "");
    return name;
}

} // namespace

bool tileRgbToYuvSupported(const RgbToYuvPlan & p)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    if (o.is565)
        return false;
    if (o.isGray) {
        // gray sources (GRAY / GRAYA / AGRAY -> luma): 16-byte loads, 4-sample stores; the reference's own loop in either arithmetic
        // (libyuv is never asked for gray sources, src/reformat.c:255)
        if (!s.exactDiv || p.rx0 != 0 || p.ry0 != 0 || p.rw != p.width || p.rh != p.height || p.width < 64)
            return false;
        if (p.mul != MUL_NONE && !o.hasAlpha)
            return false;
        const uint32_t ybps = (uint32_t)s.chanBytes, ppl = 16u / ((o.hasAlpha ? 2u : 1u) * (uint32_t)o.chanBytes);
        const uint32_t planeVec = ppl * ybps < 16u ? ppl * ybps : 16u; // a lane's samples of a plane row leave as one store (16 bytes at most)
        if (!alignedTo(o.pixels, o.rowBytes, 16) || !alignedTo(s.plane[0], s.rowBytes[0], planeVec))
            return false;
        if (p.alphaSource != ALPHA_KEEP && (!s.alpha || !alignedTo(s.alpha, s.alphaRowBytes, planeVec)))
            return false;
        return true;
    }
    if (p.mul != MUL_NONE && (p.arith != ARITH_FLOAT || !o.hasAlpha))
        return false; // pending alpha (un)multiply: 4-channel sources of the fp32 arithmetic (libyuv is never asked: src/reformat.c:255)
    if (p.arith == ARITH_FLOAT) {
        if (s.mode == MODE_IDENTITY) {
            // lossless RGB as GBR planes (avifenc -l): no matrix, one division (channel / maximum); 4:4:4 or monochrome only
            if (!s.exactDiv || (s.format != AVIF_PIXEL_FORMAT_YUV444 && s.format != AVIF_PIXEL_FORMAT_YUV400))
                return false;
        } else if (s.mode == MODE_YCGCO || s.mode == MODE_YCGCO_RE || s.mode == MODE_YCGCO_RO) {
            // three adds on the normalised channels / integer lifting on the codes and "/ range": the channel maximum and the ranges must be
            // on the verified list
            if (!s.exactDiv)
                return false;
        } else {
            if (s.mode != MODE_COEFF)
                return false;
            if (!s.exactDivEncode)
                return false; // a divisor off the verified lists (exactdiv.h): the universal kernel divides the IEEE way
        }
    } else if (o.chanBytes != 1 || s.chanBytes != 1) {
        return false; // libyuv converts 8-bit to 8-bit only (src/reformat_libyuv.c:277)
    }
    if (p.rx0 != 0 || p.ry0 != 0 || p.rw != p.width || p.rh != p.height)
        return false;
    if (p.width < 64 || p.height < 2)
        return false;
    const uint32_t bps = (uint32_t)s.chanBytes;
    const int nch = o.hasAlpha ? 4 : 3;
    const uint32_t loadAlign = (nch == 4) ? 16u : (o.chanBytes == 1 ? 4u : 8u);
    if (!alignedTo(o.pixels, o.rowBytes, loadAlign))
        return false;
    if (!alignedTo(s.plane[0], s.rowBytes[0], 4 * bps))
        return false;
    if (s.format != AVIF_PIXEL_FORMAT_YUV400) {
        const uint32_t chromaAlign = (s.format == AVIF_PIXEL_FORMAT_YUV444) ? 4 * bps : 2 * bps;
        if (!s.plane[1] || !s.plane[2] || !alignedTo(s.plane[1], s.rowBytes[1], chromaAlign) || !alignedTo(s.plane[2], s.rowBytes[2], chromaAlign))
            return false;
    }
    if (p.alphaSource != ALPHA_KEEP && (!s.alpha || !alignedTo(s.alpha, s.alphaRowBytes, 4 * bps)))
        return false;
    if (!fits32(p.height, o.rowBytes) || !fits32(p.height, s.rowBytes[0]) || !fits32(p.height, s.alphaRowBytes) || !fits32(p.height, s.rowBytes[1]) ||
        !fits32(p.height, s.rowBytes[2]))
        return false;
    return true;
}

namespace {
hipError_t launchGrayToYuvTile(const RgbToYuvPlan & p, hipStream_t stream, const char ** kernelName)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    const int gch = o.hasAlpha ? 2 : 1;
    if (kernelName) {
        static thread_local char name[64];
        snprintf(name, sizeof(name), "gray2yuv_tile<%s%d,%s%s>", gch == 2 ? "graya" : "gray", o.chanBytes == 2 ? 16 : 8, s.chanBytes == 2 ? "u16" : "u8", p.mul != MUL_NONE ? ",alphamul" : "");
        *kernelName = name;
    }
    GrayArgs A;
    memset(&A, 0, sizeof(A));
    const uint32_t ppl = 16u / ((uint32_t)gch * (uint32_t)o.chanBytes);
    A.gray = o.pixels, A.y = s.plane[0], A.a = (p.alphaSource != ALPHA_KEEP) ? s.alpha : nullptr;
    A.grayPitch = o.rowBytes, A.yPitch = s.rowBytes[0], A.aPitch = s.alphaRowBytes;
    A.wP = p.width - p.width % ppl, A.height = p.height;
    A.rcpRgbMax = o.rcpMax;
    A.rangeY = s.rangeY, A.biasY = s.biasY, A.yuvMax = (uint32_t)s.maxv, A.yuvMaxF = (float)s.maxv;
    A.alphaFirst = (gch == 2 && o.offA == 0) ? 1u : 0u;
    A.alphaMode = R2Y_ALPHA_NONE;
    if (p.alphaSource == ALPHA_FILL)
        A.alphaMode = R2Y_ALPHA_FILL;
    else if (p.alphaSource == ALPHA_PLANE)
        A.alphaMode = (o.depth == s.depth) ? R2Y_ALPHA_COPY : R2Y_ALPHA_RESCALE;
    A.mulMode = p.mul;
    hipError_t e = (o.chanBytes == 2) ? launchR2YGray16(gch, s.chanBytes == 2, A, stream) : launchR2YGray8(gch, s.chanBytes == 2, A, stream);
    if (e != hipSuccess)
        return e;
    if (A.wP != p.width) { // the columns that do not fill a lane's 16 bytes
        RgbToYuvPlan rest = p;
        rest.rx0 = A.wP, rest.rw = p.width - A.wP;
        e = launchRgbToYuvGeneric(rest, stream);
        if (e != hipSuccess)
            return e;
    }
    return launchGrayChromaFill(p, stream);
}
} // namespace

namespace {
// the tiled kernel's arguments for the whole-group part of a colour plan, and the workgroups of its launch
uint32_t colourArgsOf(const RgbToYuvPlan & p, const R2YKey & k, uint32_t tuning, R2YArgs * out)
{
    const YuvSide & s = p.yuv;
    const RgbSide & o = p.rgb;
    R2YArgs & A = *out;
    memset(&A, 0, sizeof(A));
    A.rgb = o.pixels;
    A.y = s.plane[0], A.u = s.plane[1], A.v = s.plane[2], A.a = s.alpha;
    A.rgbPitch = o.rowBytes, A.yPitch = s.rowBytes[0], A.uPitch = s.rowBytes[1], A.vPitch = s.rowBytes[2], A.aPitch = s.alphaRowBytes;
    A.w4 = p.width & ~3u, A.h2 = p.height & ~1u;
    A.kr = s.kr, A.kg = s.kg, A.kb = s.kb;
    A.rcpRgbMax = o.rcpMax, A.rcpCbDen = s.rcpCbDen, A.rcpCrDen = s.rcpCrDen;
    A.rangeY = s.rangeY, A.biasY = s.biasY, A.rangeUV = s.rangeUV, A.biasUV = s.biasUV;
    A.identity = (p.arith == ARITH_FLOAT && s.mode == MODE_IDENTITY) ? 1 : 0;
    A.matrixMode = (p.arith == ARITH_FLOAT) ? s.mode : MODE_COEFF;
    A.rcpRangeY = s.rcpRangeY, A.rcpRangeUV = s.rcpRangeUV, A.rgbMaxF = o.maxf;
    A.mulMode = p.mul;
    if (A.identity)
        A.rangeUV = s.rangeY, A.biasUV = s.biasY; // src/reformat.c:205-211: identity quantises chroma like luma
    A.yuvMax = (uint32_t)s.maxv, A.yuvMaxF = (float)s.maxv;
    A.slotR = (uint32_t)(o.offR / o.chanBytes), A.slotB = (uint32_t)(o.offB / o.chanBytes), A.slotA = (uint32_t)(o.offA / o.chanBytes);
    A.alphaMode = R2Y_ALPHA_NONE;
    if (p.alphaSource == ALPHA_FILL)
        A.alphaMode = R2Y_ALPHA_FILL;
    else if (p.alphaSource == ALPHA_PLANE)
        A.alphaMode = (o.depth == s.depth) ? R2Y_ALPHA_COPY : R2Y_ALPHA_RESCALE;

    // decomposition: a wave takes 1, 2 or 4 vertically consecutive strips (tests/tools/cfg_bench.py with AVIFHIP_R2Y_SPW)
    const uint32_t bands = (A.w4 + 255) / 256;
    const uint32_t strips = A.h2 / 2;
    const uint64_t waveStrips = (uint64_t)bands * strips;
    // Two strips from the size on at which one strip per wave no longer fits the chip at once (8 192 waves: 1 024 SIMDs x 8): a 4K frame's
    // 16 200 strips are 8 100 waves, all resident from the first cycle to the last instead of 2.26 rounds of short ones -- 8.67 -> 7.70 us
    // for a frame the caches hold, no difference (10.63 / 10.65 us) for frames that stream; 1080p frames want one strip (4.34 / 5.00 us),
    // four strips lose everywhere (interleaved A/B, tests/tools/spw_ab.py)
    uint32_t spw = waveStrips > 8192 ? 2 : 1;
    // (the identity matrix -- lossless GBR planes -- has next to no arithmetic between a wave's loads and its four planes' stores: two strips
    //  per wave ran an 8K RGBA8 frame in 44-53 us, box to box, one strip in 37.5 = 0.885; round 6, AVIFHIP_R2Y_SPW sweep)
    if (p.arith == ARITH_FLOAT && s.mode == MODE_IDENTITY && p.mul == MUL_NONE)
        spw = 1;
    if (const char * e = getenv("AVIFHIP_R2Y_SPW")) // diagnostics / A-B measurements only
        spw = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : spw;
    spw = spw >= 4 ? 4 : (spw >= 2 ? 2 : 1);
    const bool plain = A.mulMode == MUL_NONE && A.matrixMode != MODE_YCGCO && A.matrixMode != MODE_YCGCO_RE && A.matrixMode != MODE_YCGCO_RO;
    if (!plain && !k.fixedPoint && spw < 2)
        spw = 2; // the kernels of the rare modes exist with two strips per wave or more (r2y_tile_impl.h launchOnePlainOrNot)
    A.stripsPerWave = spw;
    A.xcdBands = (tuning & TUNE_R2Y_RASTER) ? 0u : 1u;
    const uint32_t chunks = (strips + 4 * spw - 1) / (4 * spw);
    if (k.fixedPoint) { // appendix D.5, coefficients in memory order of the colour channels
        const bool full = p.fxFullRange != 0;
        const int yr = full ? 77 : 66, yg = full ? 150 : 129, yb = full ? 29 : 25;
        const int ur = full ? -43 : -38, ug = full ? -85 : -74, ub = full ? 128 : 112;
        const int vr = full ? 128 : 112, vg = full ? -107 : -94, vb = full ? -21 : -18;
        const bool redFirst = A.slotR < A.slotB;
        A.fx.y0 = redFirst ? yr : yb, A.fx.y1 = yg, A.fx.y2 = redFirst ? yb : yr, A.fx.yBias = full ? 128 : 0x1080;
        A.fx.u0 = redFirst ? ur : ub, A.fx.u1 = ug, A.fx.u2 = redFirst ? ub : ur;
        A.fx.v0 = redFirst ? vr : vb, A.fx.v1 = vg, A.fx.v2 = redFirst ? vb : vr;
    }
    return bands * chunks;
}

// leftovers: columns [w4, width) of every row, then row h2 of the columns before w4 (even origins: whole 2x2 blocks)
hipError_t colourLeftovers(const RgbToYuvPlan & p, const R2YArgs & A, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    if (A.w4 != p.width) {
        RgbToYuvPlan rest = p;
        rest.rx0 = A.w4, rest.rw = p.width - A.w4;
        e = launchRgbToYuvGeneric(rest, stream);
        if (e != hipSuccess)
            return e;
    }
    if (A.h2 != p.height) {
        RgbToYuvPlan rest = p;
        rest.ry0 = A.h2, rest.rh = p.height - A.h2, rest.rw = A.w4;
        e = launchRgbToYuvGeneric(rest, stream);
    }
    return e;
}

hipError_t launchColour(const R2YKey & k, const R2YArgs & A, uint32_t blocks, hipStream_t stream, const R2YSeqFrames & S, uint32_t frames)
{
    return k.fixedPoint ? launchR2YTileFx(k, A, blocks, stream, S, frames)
                        : (k.wideRgb ? launchR2YTileRgb16(k, A, blocks, stream, S, frames) : launchR2YTileRgb8(k, A, blocks, stream, S, frames));
}
} // namespace

hipError_t launchRgbToYuvTile(const RgbToYuvPlan & p, hipStream_t stream, const char ** kernelName, uint32_t tuning)
{
    if (p.rgb.isGray)
        return launchGrayToYuvTile(p, stream, kernelName);
    const R2YKey k = keyFor(p);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    R2YArgs A;
    const uint32_t blocks = colourArgsOf(p, k, tuning, &A);
    const hipError_t e = launchColour(k, A, blocks, stream, r2ySeqOfOne(A), 1);
    if (e != hipSuccess)
        return e;
    return colourLeftovers(p, A, stream);
}

// Sequences: frames that differ in their buffers only, each large enough for the single image's launch geometry to be the right one
static_assert(kRgbToYuvSequenceMax == kR2YSeqMaxFrames, "kernels.h and r2y_tile_shared.h disagree on the frames of a sequence launch");
bool tileRgbToYuvSequenceCompatible(const RgbToYuvPlan & first, const RgbToYuvPlan & other, uint32_t tuning)
{
    if (first.rgb.isGray || other.rgb.isGray || !tileRgbToYuvSupported(first) || !tileRgbToYuvSupported(other))
        return false;
    if (first.width != other.width || first.height != other.height || (uint64_t)first.width * first.height < ((uint64_t)2 << 20))
        return false;
    const R2YKey ka = keyFor(first), kb = keyFor(other);
    if (ka.fixedPoint != kb.fixedPoint || ka.wideRgb != kb.wideRgb || ka.wideYuv != kb.wideYuv || ka.nch != kb.nch || ka.sub != kb.sub || ka.hasMul != kb.hasMul)
        return false;
    R2YArgs a, b;
    return colourArgsOf(first, ka, tuning, &a) == colourArgsOf(other, kb, tuning, &b) && r2ySeqCompatible(a, b);
}

hipError_t launchRgbToYuvTileSequence(const RgbToYuvPlan * plans, uint32_t count, hipStream_t stream, const char ** kernelName, uint32_t tuning)
{
    if (count == 0 || count > kR2YSeqMaxFrames)
        return hipErrorInvalidValue;
    const R2YKey k = keyFor(plans[0]);
    if (kernelName)
        *kernelName = kernelNameFor(k);
    R2YArgs A;
    const uint32_t blocks = colourArgsOf(plans[0], k, tuning, &A);
    R2YSeqFrames S = r2ySeqOfOne(A);
    for (uint32_t f = 1; f < count; ++f) {
        R2YArgs B;
        (void)colourArgsOf(plans[f], k, tuning, &B);
        r2ySeqSetFrame(S, f, B);
    }
    hipError_t e = launchColour(k, A, blocks, stream, S, count);
    for (uint32_t f = 0; f < count && e == hipSuccess; ++f)
        e = colourLeftovers(plans[f], A, stream);
    return e;
}

} // namespace avifhip
