// plan.h -- host-derived description of one reformat job, consumed by the HIP kernels.
// POD only: it travels to the device by value (kernarg) or in a descriptor table (batch launches).
#pragma once

#include <stdint.h>

#include "avifhip.h"

namespace avifhip {

enum ReformatMode : int { MODE_COEFF = 0, MODE_IDENTITY = 1, MODE_YCGCO = 2, MODE_YCGCO_RE = 3, MODE_YCGCO_RO = 4 };
enum MulMode : int { MUL_NONE = 0, MUL_MULTIPLY = 1, MUL_UNMULTIPLY = 2 };
enum AlphaSource : int {
    ALPHA_KEEP = 0,  // destination alpha bytes (if any) are not written
    ALPHA_FILL = 1,  // opaque                      (avifFillAlpha, src/alpha.c:9)
    ALPHA_PLANE = 2  // from the alpha plane, copy or depth rescale (avifReformatAlpha, src/alpha.c:37)
};
enum Arith : int { ARITH_FLOAT = 0, ARITH_LIBYUV = 1 };
// How the A channel of a fixed-point (libyuv-arithmetic) conversion is produced, src/reformat_libyuv.c:956-959,982-1100
enum FxAlpha : int {
    FXA_OPAQUE = 0, // libyuv's own 255 (also when rgb->ignoreAlpha: libyuv always writes the channel)
    FXA_SHIFT = 1,  // an *Alpha* entry: plane sample >> fxAlphaShift, saturated to 255
    FXA_FLOAT = 2   // libyuv wrote 255, then avifReformatAlpha (src/alpha.c:37-149) overwrote it: copy or fp32 rescale
};

// 1/d of a plan constant d, split into RN(1/d) and the rounded remainder.  For the divisors on the verified list
// (exactdiv.h)  fma(x, hi, x * lo)  equals the correctly rounded IEEE-754 binary32 quotient x / d bit for bit.
struct RcpHL
{
    float hi, lo;
};

// Where a converted canvas pixel goes when the decode-side transforms are fused into the conversion (SURVEY.md 8f rank 1:
// avifApplyTransforms, apps/shared/avifutil.c:787-825 = clean-aperture crop as a view, avifRGBImageRotate :687-741,
// avifRGBImageMirror :745-785, in that order).  Canvas pixel (i, j) with ii = i - cx < cw, jj = j - cy < ch lands at
//     (x, y) = (sx * ii + kx, sy * jj + ky)            rows stay rows      (no rotation / half turn)
//     (x, y) = (sx * jj + kx, sy * ii + ky)            rows become columns (quarter turns)
// of the destination buffer (`pixels`, `rowBytes` of the RgbSide); pixels outside the crop are dropped.
struct PixelMap
{
    int32_t on;
    int32_t transposed;
    int32_t sx, sy, kx, ky;
    uint32_t cx, cy, cw, ch;
};
inline PixelMap makePixelMap(uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, int angle, int mirror /* -1 none, 0 top<->bottom, 1 left<->right */)
{
    PixelMap m;
    m.on = 1, m.cx = cx, m.cy = cy, m.cw = cw, m.ch = ch;
    m.transposed = angle & 1;
    switch (angle & 3) { // anti-clockwise quarter turns: source (ii, jj) -> (jj, cw-1-ii) -> (cw-1-ii, ch-1-jj) -> (ch-1-jj, ii)
        case 1: m.sx = 1, m.kx = 0, m.sy = -1, m.ky = (int32_t)cw - 1; break;
        case 2: m.sx = -1, m.kx = (int32_t)cw - 1, m.sy = -1, m.ky = (int32_t)ch - 1; break;
        case 3: m.sx = -1, m.kx = (int32_t)ch - 1, m.sy = 1, m.ky = 0; break;
        default: m.sx = 1, m.kx = 0, m.sy = 1, m.ky = 0; break;
    }
    const int32_t dw = (int32_t)((angle & 1) ? ch : cw), dh = (int32_t)((angle & 1) ? cw : ch);
    if (mirror == 1)
        m.sx = -m.sx, m.kx = dw - 1 - m.kx;
    else if (mirror == 0)
        m.sy = -m.sy, m.ky = dh - 1 - m.ky;
    return m;
}

// The crop, grown to the left / upwards to an origin the tile kernels take (x a multiple of 8, y even: kernels_tile.hip
// tileYuvToRgbSupported): only what the crop keeps is converted; the few extra columns / rows are dropped by the map.
// A quarter turn stores 128-byte runs along destination rows, one per tile row band (tile_map_impl.h mapTransposeStore), and where those
// start within a 128-byte line is set by the first row converted.  Measured at 8K -> RGBA16 (cfg_bench tail90_rgba10, one box): runs that
// are whole lines 377 k megapixels/s, split 32 + 96 bytes 341 k, 48 + 80 339 k, 64 + 64 306 k.  So the rectangle starts up to one run's
// pixels above the crop, on the row that makes the runs whole lines, or failing that keeps them furthest from an even split.
inline avifCropRect coverOfCrop(const avifCropRect & r, const PixelMap & map, uintptr_t pixelsAddress, uint32_t pixelBytes)
{
    const uint32_t x0 = r.x & ~7u;
    uint32_t y0 = r.y & ~1u;
    if (map.transposed && (pixelBytes == 4 || pixelBytes == 8)) {
        const int64_t runPx = 128 / pixelBytes;
        int bestScore = -1;
        uint32_t bestY = y0;
        for (uint32_t y = y0;; y -= 2) {
            const int64_t d = (int64_t)y - (int64_t)r.y; // first converted row, in crop rows (<= 0)
            const int64_t startPx = map.sx > 0 ? (int64_t)map.kx + d : (int64_t)map.kx - d - (runPx - 1);
            const uint32_t off = (uint32_t)(((int64_t)pixelsAddress + startPx * (int64_t)pixelBytes) & 127);
            const int score = off == 0 ? 1000 : ((int)off > 64 ? (int)off - 64 : 64 - (int)off);
            if (score > bestScore)
                bestScore = score, bestY = y;
            if (off == 0 || y < 2 || (int64_t)(y0 - y) + 2 >= runPx)
                break;
        }
        y0 = bestY;
    }
    return avifCropRect { x0, y0, r.x + r.width - x0, r.y + r.height - y0 };
}

// Interleaved-pixel side (avifRGBColorSpaceInfo, include/avif/internal.h:297-309)
struct RgbSide
{
    uint8_t * pixels;
    uint32_t rowBytes;
    uint32_t depth;
    int32_t format; // avifRGBFormat
    int32_t chanBytes, pixBytes;
    int32_t offR, offG, offB, offA, offGray;
    int32_t hasAlpha, isGray, is565, isFloat;
    int32_t maxv;
    float maxf;
    float f16Multiplier; // src/reformat.c:1411,1429-1430
    RcpHL rcpMax;        // 1 / maxf (premultiply, src/alpha.c:189)
    PixelMap map;        // fused crop / rotate / mirror (off unless the caller asks: avifhip*TransformedAsync)
};

// Planar side (avifYUVColorSpaceInfo, include/avif/internal.h:314-331)
struct YuvSide
{
    uint8_t * plane[3];
    uint8_t * alpha;
    uint32_t rowBytes[3];
    uint32_t alphaRowBytes;
    uint32_t depth;
    int32_t format; // avifPixelFormat
    int32_t chanBytes;
    int32_t shiftX, shiftY;
    int32_t hasColor; // chroma planes present and format != 400
    int32_t alphaLimited; // the alpha plane holds limited-range samples: avifLimitedToFullY first (src/read.c:6724-6764)
    int32_t limited;
    int32_t maxv;
    int32_t mode; // ReformatMode
    float kr, kg, kb;
    float biasY, biasUV, rangeY, rangeUV;
    // products the reference forms from kr/kb before touching pixel data (src/reformat.c:874-876)
    float twoOneMinusKr; // 2*(1-kr)
    float twoOneMinusKb; // 2*(1-kb)
    float krOneMinusKr;  // kr*(1-kr)
    float kbOneMinusKb;  // kb*(1-kb)
    // Division by plan constants without the IEEE divide sequence (RcpHL above).  exactDiv is set only when every
    // divisor of the plan is on the verified list (exactdiv.h); the tiled kernels require it, the universal kernels
    // always use '/'.
    RcpHL rcpRangeY, rcpRangeUV;
    RcpHL rcpKgTimes2; // 2 / kg: (2 * x) / kg == x * (2 / kg) up to the same single rounding (power-of-two scaling)
    RcpHL rcpMax;      // 1 / maxv (alpha normalisation and depth rescale, src/reformat.c:897, src/alpha.c:93)
    int32_t exactDiv;
    // encode direction (src/reformat.c:384-385): 1 / (2*(1-kb)), 1 / (2*(1-kr)); exactDivEncode when both denominators
    // and the RGB channel maximum are on the verified lists
    RcpHL rcpCbDen, rcpCrDen;
    int32_t exactDivEncode;
};

// libyuv YuvConstants as black-box verified in SURVEY.md Appendix D.1
struct FixedPointMatrix
{
    int32_t yg, yb, ub, ug, vg, vr;
};

struct YuvToRgbPlan
{
    YuvSide yuv;
    RgbSide rgb;
    uint32_t canvasW, canvasH; // edge rules are evaluated against these (src/reformat.c:768,784)
    uint32_t x0, y0, w, h;     // rectangle converted by this job
    // Chroma samples this job may read, in chroma-plane coordinates (inclusive): neighbours selected by the filter rules are
    // clamped into the window.  The whole plane by default (a no-op: the rules never leave the plane); a tile's own samples
    // when the "canvas" is a grid of separately stored tiles (avifhipGridYUVToRGBAsync), whose seams a second pass redoes.
    int32_t cwinX0, cwinX1, cwinY0, cwinY1;
    int32_t bilinear;          // 4-tap chroma filter requested (420/422 only)
    int32_t alphaSource;       // AlphaSource
    int32_t inLoopMul;         // MulMode applied in fp32 before quantisation (slow path, :894-947)
    int32_t postMul;           // MulMode applied on the quantised integers (fast path + :1574-1585)
    int32_t identityCopy;      // src/reformat.c:1278-1309
    int32_t arith;             // Arith: which arithmetic converts the colour channels
    // ---- arith == ARITH_LIBYUV ----
    FixedPointMatrix fx;
    int32_t fxNative;          // 8, 10 (I010/I210/I410: y<<6|y>>4, chroma>>2 after upsampling) or 12 (I012: y<<4|y>>8, chroma>>4)
    int32_t fxDownshift;       // Convert16To8Plane before an 8-bit entry: every sample >> fxDownshift, saturated to 255
    int32_t fxMono;            // I400ToARGBMatrix: chroma = 128
    int32_t fxAlpha;           // FxAlpha
    int32_t fxAlphaShift;      // FXA_SHIFT: 0 (8-bit), 2 (native 10-bit) or the downshift
    // ---- both arithmetics ----
    int32_t postMulFx;         // the integer post-pass is libyuv's ARGBAttenuate / ARGBUnattenuate (appendix D.4)
    int32_t mulOfTheCall;      // the MulMode avifImageYUVToRGB derives for these arguments (src/reformat.c:1662-1677), whatever this job applies of it
    uint32_t tuning;           // TuningBits: performance knobs that never change results
};

enum TuningBits : uint32_t {
    TUNE_XCD_BANDS = 1u << 0,     // tiles of one XCD form a contiguous band of the image (chroma halo rows hit its L2)
    TUNE_NONTEMPORAL = 1u << 1,   // streaming (nt) stores for the RGB output
    TUNE_SOLO_ALWAYS = 1u << 3,   // ... the wave-private kernels for every family (A/B measurements)
    TUNE_COOPERATIVE = 1u << 2,   // fp32 / 10-12-bit integer families: round 1's cooperative runs (four waves stage together, one barrier
                                  // per tile) instead of the wave-private kernels (A/B measurements)
    TUNE_ALL_ALPHA_MODES = 1u << 5, // fp32 tiles with a pending alpha multiply: the kernel that carries every alpha mode, not the one compiled for the job's (A/B measurements)
    TUNE_STREAM_LOADS = 1u << 6,  // single images of 16-bit planes, unfiltered chroma: streaming (non-temporal) plane loads (A/B measurements)
    TUNE_R2Y_RASTER = 1u << 7,    // RGB -> YUV tiles in plain raster order instead of per-XCD bands (A/B measurements)
    TUNE_CANVAS_ORDER = 1u << 25, // grids: workgroups along the rows of the canvas whatever the canvas's size (tests: small grids through the order large ones take)
    TUNE_JOB_MAJOR = 1u << 24,    // grids: workgroups job by job (rounds 1-4) instead of along the rows of the canvas (A/B measurements)
    TUNE_PRIVATE_HALO = 1u << 4,  // packed kernels, 4:2:0 bilinear: every wave stages its own chroma halo rows (no workgroup barrier; A/B measurements)
    TUNE_DEFAULT = TUNE_XCD_BANDS,
    TUNE_STRIPS_SHIFT = 8,        // bits 8..11: forced strips per wave (tile height / 8) of the tiled kernels, 0 = automatic
    TUNE_RUN_SHIFT = 12,          // bits 12..15: forced tiles per workgroup run, 0 = automatic
    // packed 16-bit integer kernels (tile_pk_impl.h): strips per wave from TUNE_STRIPS_SHIFT (2 or 4), and
    TUNE_WAVESX_SHIFT = 16,       // bits 16..17: 1 + log2(waves of a workgroup side by side), 0 = automatic
    TUNE_CHUNK_SHIFT = 20         // bits 20..23: tile rows per XCD chunk when TUNE_XCD_BANDS is set (0 = automatic), raster order otherwise
};

struct RgbToYuvPlan
{
    YuvSide yuv; // destination planes
    RgbSide rgb; // source pixels (const in practice)
    uint32_t width, height;
    uint32_t rx0, ry0, rw, rh; // region converted by this job (even origin; edge blocks are decided against width/height)
    int32_t mul;         // MulMode applied in fp32 (src/reformat.c:325-358)
    int32_t alphaSource; // ALPHA_KEEP (no alpha plane) / ALPHA_FILL / ALPHA_PLANE(= from rgb alpha channel)
    int32_t arith;       // ARITH_LIBYUV: appendix D.5 (8-bit BT.601 only)
    int32_t fxFullRange;
};

struct AlphaMulPlan
{
    RgbSide rgb;
    uint32_t width, height;
    int32_t unmultiply;
    int32_t arith; // ARITH_LIBYUV: ARGBAttenuate / ARGBUnattenuate (8-bit RGBA / BGRA)
    int32_t exactDiv; // the channel maximum is on the verified list (exactdiv.h): "/ maxF" may use the reciprocal form
};

// arithMode (avifhipArithmetic): which libavif BUILD the result must equal.
//   AUTO   a libavif built with libyuv (the default build): libyuv's fixed-point arithmetic wherever that build hands
//          the work to libyuv -- honouring rgb->avoidLibYUV exactly like src/reformat.c:1453 and :264 do, and NOT
//          honouring it for (un)premultiply exactly like src/alpha.c:163,350 -- the fp32 path everywhere else;
//   FLOAT  a libavif built without libyuv: fp32 everywhere;
//   LIBYUV as AUTO but rgb->avoidLibYUV is ignored.
// Host-side derivation (plan.cpp). Return an avifResult; AVIF_RESULT_OK means the plan is valid.
// colorOnly: the job libavif hands to its accelerated-backend hook (avifImageYUVToRGBLibYUV, include/avif/internal.h:
// 349-363): colour conversion without any alpha (un)multiply or half-float pass (the caller runs those afterwards),
// alpha channel written only when reformatAlpha.
avifResult makeYuvToRgbPlan(const avifImage * image, const avifRGBImage * rgb, const avifCropRect * rect, int arithMode, uint32_t tuning, YuvToRgbPlan * out,
                            bool colorOnly = false, bool reformatAlpha = false);
// The plan of (image, rgb, rect) from the plan `proto` already made for (protoImage, protoRgb) with the same arithmetic and tuning, when the
// two jobs differ in nothing a plan is derived from except where their buffers are and which rectangle they convert -- the tiles of a grid,
// the frames of a sequence.  Returns false when they do differ (the caller then makes the plan from scratch); a 48-tile photograph costs
// one derivation (colour coefficients, reciprocals, libyuv's dispatch) instead of 48.
bool rebindYuvToRgbPlan(const YuvToRgbPlan & proto, const avifImage * protoImage, const avifRGBImage * protoRgb, const avifImage * image, const avifRGBImage * rgb,
                        const avifCropRect * rect, YuvToRgbPlan * out, avifResult * result);
avifResult makeRgbToYuvPlan(const avifImage * image, const avifRGBImage * rgb, int arithMode, RgbToYuvPlan * out);
avifResult makeAlphaMulPlan(const avifRGBImage * rgb, bool unmultiply, int arithMode, AlphaMulPlan * out);

// kr/kg/kb for an image, reference src/colr.c:156-189 (avifCalcYUVCoefficients)
void calcYuvCoefficients(const avifImage * image, float * kr, float * kg, float * kb);
bool rgbFormatHasAlpha(int format);
bool rgbFormatIsGray(int format);
int rgbFormatChannelCount(int format);

} // namespace avifhip
