// kernels.h -- launch entry points of the HIP kernels (kernels_generic.hip, kernels_tile.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace avifhip {

// universal one-lane-per-pixel kernels (every combination the reference accepts)
hipError_t launchYuvToRgbGeneric(const YuvToRgbPlan & plan, hipStream_t stream);
hipError_t launchYuvToRgbGenericBatch(const YuvToRgbPlan * deviceTable, uint32_t count, uint32_t maxW, uint32_t maxH, hipStream_t stream);
hipError_t launchRgbToYuvGeneric(const RgbToYuvPlan & plan, hipStream_t stream);

// A canvas stored as a grid of separate tile images (avifhipGridYUVToRGBAsync): where canvas sample (x, y) of each plane lives
struct GridTile
{
    const uint8_t * plane[3];
    const uint8_t * alpha;
    uint32_t rowBytes[3];
    uint32_t alphaRowBytes;
};
struct GridGeometry
{
    uint32_t columns, rows;
    uint32_t tileW, tileH;   // luma / alpha samples per tile
    uint32_t tileCW, tileCH; // chroma samples per tile
    // ceil(2^32 / d) of the four sizes (filled by launchYuvToRgbGridSeams): x / d == mulhi(x, magic) for every canvas coordinate (x * d < 2^32)
    uint32_t magicW, magicH, magicCW, magicCH;
};
// Re-converts the pixels next to interior tile seams (the only ones whose chroma filter reaches into a neighbouring tile):
// luma columns k*tileW-1, k*tileW when `vertical`, luma rows k*tileH-1, k*tileH when `horizontal`.  `canvasPlan` describes the
// whole canvas (its plane pointers are not used); `deviceTiles` has columns*rows entries.
hipError_t launchYuvToRgbGridSeams(const YuvToRgbPlan & canvasPlan, const GridGeometry & geometry, const GridTile * deviceTiles, bool vertical,
                                   bool horizontal, hipStream_t stream);
hipError_t launchAlphaMulGeneric(const AlphaMulPlan & plan, hipStream_t stream);
// in-place uint16 -> IEEE half over rows of `samplesPerRow` samples, src/reformat.c:1419-1443
hipError_t launchToF16Generic(uint8_t * pixels, uint32_t rowBytes, uint32_t samplesPerRow, uint32_t rows, float multiplier, hipStream_t stream);

// bandwidth-tuned tiled kernels; return false from the *Supported predicates when a plan is not covered
bool tileYuvToRgbSupported(const YuvToRgbPlan & plan);
// id of the tiled-kernel instantiation serving `plan`, or -1 when only the generic kernel can
int tileYuvToRgbVariant(const YuvToRgbPlan & plan);
hipError_t launchYuvToRgbTile(const YuvToRgbPlan & plan, hipStream_t stream, const char ** kernelName);
// Sequences (tile_shared.h SeqFrames): `count` <= 8 jobs that differ in their buffers only, each a frame large enough for the single image's
// launch geometry, in ONE launch of the single-image kernels with the frames' addresses in the kernel arguments -- no device table, nothing
// uploaded, no event between launches.  tileSequenceCompatible: `other` may share a launch with `first`; the launcher answers
// hipErrorNotSupported (and launches nothing) for kernel families without sequence kernels.
constexpr uint32_t kTileSequenceMax = 8;
bool tileSequenceCompatible(const YuvToRgbPlan & first, const YuvToRgbPlan & other);
hipError_t launchYuvToRgbTileSequence(const YuvToRgbPlan * plans, uint32_t count, hipStream_t stream, const char ** kernelName);
// batch launches read a device table of distilled descriptors: tileBatchTableBytes(count) bytes, written on the host
// by fillTileBatchTable.  They convert the whole-group part (w & ~3, h & ~1) of every job; the caller hands the
// leftover columns/rows to launchYuvToRgbGenericBatch.
size_t tileBatchTableBytes(uint32_t count);
void fillTileBatchTable(const YuvToRgbPlan * plans, uint32_t count, void * hostTable);
hipError_t launchYuvToRgbTileBatch(const void * deviceTileTable, const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW,
                                   uint32_t maxH, hipStream_t stream, const char ** kernelName, bool neighboursLinked = false, uint32_t canvasColumns = 0);
// Jobs that are the tiles of ONE canvas (grid images): a job whose neighbours are linked filters chroma across the seams inside the tiled
// kernels (the seam-aware builds, tile_impl.h) -- no second pass.  Index 3 * v + h of the plane arrays: v 0 = the job's own tile, 1 = the
// tile above, 2 = below; h 0 = own, 1 = left, 2 = right; every pointer addresses canvas sample (0,0) of the tile's U / V plane, virtually
// (entries of absent neighbours are ignored).  All tiles must share the job's chroma pitches.
struct TileNeighbours
{
    const uint8_t * plane1[9];
    const uint8_t * plane2[9];
    bool above, below, left, right;
};
// whether a batch of linkable jobs is to be linked: its kernel family filters chroma at all, and one launch measured faster than the tile
// batch followed by the seam pass for this family and size (`forced`: -1 = by that measurement, 0 = never, 1 = whenever chroma is filtered)
bool tileBatchLinksNeighbours(const YuvToRgbPlan & representative, uint32_t count, uint32_t maxW, uint32_t maxH, int forced, uint32_t canvasColumns = 0);
// after fillTileBatchTable; launchYuvToRgbTileBatch is then told `neighboursLinked`
void linkTileBatchHalo(void * hostTable, uint32_t job, const TileNeighbours & neighbours);
hipError_t launchGrayChromaFill(const RgbToYuvPlan & plan, hipStream_t stream);
bool tileRgbToYuvSupported(const RgbToYuvPlan & plan);
hipError_t launchRgbToYuvTile(const RgbToYuvPlan & plan, hipStream_t stream, const char ** kernelName, uint32_t tuning = TUNE_DEFAULT);
// Sequences of the encode direction (r2y_tile_shared.h R2YSeqFrames), like launchYuvToRgbTileSequence: `count` <= 8 colour frames of 2
// megapixels or more that differ in their buffers only, ONE launch of the single-image kernel
constexpr uint32_t kRgbToYuvSequenceMax = 8;
bool tileRgbToYuvSequenceCompatible(const RgbToYuvPlan & first, const RgbToYuvPlan & other, uint32_t tuning = TUNE_DEFAULT);
hipError_t launchRgbToYuvTileSequence(const RgbToYuvPlan * plans, uint32_t count, hipStream_t stream, const char ** kernelName, uint32_t tuning = TUNE_DEFAULT);

// crop + rotate + mirror of an interleaved pixel buffer in one pass (kernels_transform.hip)
struct TransformArgs
{
    const uint8_t * src; // first pixel of the cropped source rectangle
    uint8_t * dst;
    uint32_t srcPitch, dstPitch;
    uint32_t cw, ch;  // cropped source size
    uint32_t dw, dh;  // destination size
    int32_t angle;    // 0..3, multiples of 90 degrees anti-clockwise
    int32_t mirror;   // -1 none, 0 about the horizontal axis (top <-> bottom), 1 about the vertical axis (left <-> right)
};
hipError_t launchRgbTransform(const TransformArgs & args, uint32_t pixelBytes, hipStream_t stream);

// row packing for the Y4M / PNG writers (kernels_pack.hip): rows of `widthBytes` bytes from a pitched source into a destination
// whose pitch is widthBytes (a byte stream) or, for the 16-byte fast path, any multiple of 16
struct PackArgs
{
    const uint8_t * src;
    uint8_t * dst;
    uint32_t srcPitch, dstPitch;
    uint32_t widthBytes, rows;
    int32_t swap16; // swap the bytes of every 16-bit sample (little-endian -> big-endian)
};
hipError_t launchPackRows(const PackArgs & args, hipStream_t stream);
// rows at any source pitch / alignment -> rows at a 4-byte aligned destination pitch (dstPitch, dst: multiples of 4)
hipError_t launchUnpackRows(const PackArgs & args, hipStream_t stream);

// plane scaling (kernels_scale.hip): schedule tables live in device memory, modes as in scale_plan.h
enum { SCALE_POINT_MODE = 0, SCALE_DOWN_MODE = 1, SCALE_UP_MODE = 2, SCALE_BOX_MODE = 3, SCALE_UP2_MODE = 4 };
struct ScaleArgs
{
    const uint8_t * src;
    uint8_t * dst;
    uint32_t srcPitch, dstPitch;
    int32_t srcW, srcH, dstW, dstH;
    int32_t mode;
    const int32_t * colA; // dstW entries each
    const int32_t * colB;
    const int32_t * rowA; // dstH entries each
    const int32_t * rowB;
    const int32_t * rowF;
};
// Parameters of the row-staged kernel (wide loads of source-row segments into LDS, 4 destination samples per lane), chosen by
// the host so that the block a wave stages -- the segments of every source row its `rowsPerWave` destination rows of 256
// columns read -- fits kScaleStageBytes.  rowsPerWave == 0 (or a null pointer): the one-lane-per-sample gather kernel.
constexpr int kScaleStageBytes = 16384;
struct ScaleStaging
{
    int rowsPerWave = 0;
    int rowsCap = 0;       // source rows a wave stages at most
    uint32_t segPitch = 0; // bytes between staged rows (a multiple of 16, segment + alignment slack)
    int boxWidth = 0;      // 8-bit boxes of one width w in {4, 8} with colA[i] = colA[0] + i * w: the kernel may sum dwords (v_sad_u8)
};
hipError_t launchScalePlane(const ScaleArgs & args, bool wide, hipStream_t stream); // gather kernel, one plane
struct ScaleStagedLaunch
{
    ScaleArgs plane[4];
    ScaleStaging staging[4];
    int count;
};
// every plane in one launch.  `window`: the LDS-free window kernel (8-bit samples, point / bilinear / 2x modes, source width >= 8,
// the source columns of every aligned group of 4 destination columns span <= 8 samples; only rowsPerWave of the staging is used;
// column tables padded with copies of their last entry to a multiple of 4)
hipError_t launchScalePlanesStaged(const ScaleStagedLaunch & launch, bool wide, bool window, hipStream_t stream);
// 8-bit planes doubled on both axes (ScalePlaneUp2_Bilinear): needs no schedule tables; source rows dword-aligned, destination rows 16-byte aligned
// 8-bit planes reduced by exact N x N boxes, N in {4, 8} (staging[].boxWidth = N): source rows 16-byte aligned, destination rows dword-aligned
bool scaleExactBoxCovers(const ScaleArgs & args);
hipError_t launchScalePlanesExactBox(const ScaleStagedLaunch & launch, hipStream_t stream);
bool scaleDoublingCovers(const ScaleArgs & args);
hipError_t launchScalePlanesDoubling(const ScaleStagedLaunch & launch, hipStream_t stream);

// Sample Transform expression evaluation (kernels_sato.hip), one lane per sample of one plane
constexpr int kSatoMaxTokens = 64, kSatoMaxInputs = 32;
struct SatoArgs
{
    uint8_t * dst;
    uint32_t dstPitch;
    int32_t dstWide;
    int32_t width, height;
    int32_t maxValue;
    int32_t numTokens;
    struct Token
    {
        int32_t type;  // avifSampleTransformTokenType
        int32_t value; // constant, or 0-based input index
    } tokens[kSatoMaxTokens];
};
struct SatoInputs // device table: the plane of every input image item
{
    const uint8_t * plane[kSatoMaxInputs];
    uint32_t pitch[kSatoMaxInputs];
    int32_t wide[kSatoMaxInputs];
};
hipError_t launchSato(const SatoArgs & args, const SatoInputs * deviceInputs, hipStream_t stream);

// Gain-map application (kernels_gainmap.hip; avifRGBImageApplyGainMap, reference src/gainmap.c:73-315): one lane per pixel.
// All transcendentals arrive as host-built tables (gainmap_plan.h).
struct GainMapPixelLayout // avifRGBColorSpaceInfo, src/reformat.c:32-117
{
    uint32_t channelBytes, pixelBytes, offR, offG, offB, offA;
    int32_t hasAlpha, is565, isFloat;
    uint32_t depth;
    float maxF;
};
struct GainMapStats
{
    uint32_t maxBits; // bits of max(0, every tone-mapped linear value): rgbMaxLinear, src/gainmap.c:257-259
    int32_t nan;      // a tone-mapped value was NaN, :277-281
    double sum;       // sum over pixels of max(0, r, g, b): rgbSumLinear, :287
};
struct GainMapPartial // one workgroup's share of the statistics
{
    double sum;
    float max;
    uint32_t nan;
};
// The gain map's own avifImageYUVToRGB (src/gainmap.c:185-212: into avifRGBImageSetDefaults' RGBA of the map's depth) for one pixel of an 8-bit
// 4:4:4 / 4:0:0 map, as the fast apply kernel computes it itself instead of reading a converted copy: what of a YuvToRgbPlan that takes.
struct GainMapPlaneConversion
{
    int32_t fixedPoint;   // libyuv's arithmetic (pixel_fixed.h: fxMatrix) / the reference's fp32 (pixel_math.h: yuvToRgbCore)
    int32_t hasColor;     // 4:4:4 with chroma planes (else: monochrome)
    int32_t identityCopy; // src/reformat.c:1278-1309: G, B, R are the samples
    int32_t mode;         // fp32: MODE_COEFF / MODE_IDENTITY / MODE_YCGCO
    FixedPointMatrix fx;
    float biasY, rangeY, biasUV, rangeUV;
    float twoOneMinusKr, twoOneMinusKb, krOneMinusKr, kbOneMinusKb;
    RcpHL rcpKgTimes2;    // (the plan's divisors are on the verified list: YuvSide::exactDiv)
};

struct GainMapArgs
{
    const uint8_t * base;
    uint8_t * out;
    const uint8_t * gain; // the gain map as RGBA of gainDepth bits (avifRGBImageSetDefaults layout), or null: weight 0
    uint32_t basePitch, outPitch, gainPitch, gainDepth;
    // gainPlanes: `gain` is the luma plane of an 8-bit gain map instead (gainPitch its pitch), gainU / gainV its 4:4:4 chroma planes (null:
    // 4:0:0) -- the fast kernel converts every pixel itself (no conversion launch, no 4-byte-per-pixel intermediate)
    int32_t gainPlanes;
    uint32_t gainPitchUV;
    const uint8_t * gainU;
    const uint8_t * gainV;
    GainMapPlaneConversion gainConv;
    GainMapPixelLayout baseL, outL;
    uint32_t width, height;
    const float * baseLut;  // linear light of every base sample code
    const float * gainLut;  // 3 x (1 << gainDepth): exp2f(log2 gain * weight) per channel and gain-map sample code
    const float * steps;    // output steps of the output transfer function: 2 pieces (x < 0, x >= 0) x stepEntries
    const uint16_t * guide; // GainMapSteps::guide (gainmap_plan.h): brackets of the search over the x >= 0 piece
    uint32_t guideFirstBits, guideShift, guideBuckets;
    uint32_t maxCode, nanCode, stepEntries; // stepEntries: entries per piece of `steps`, a power of two
    uint32_t ldsSteps, ldsBaseLut, ldsGainLut; // entries of the tables when the kernel is to keep ALL of them (and the guide) in LDS, else all 0
    uint32_t ldsLocator;    // the general kernel keeps the base and gain tables and the locator (below) in LDS instead, and searches nothing
    // the fast kernel (4-channel integer pixels on both sides, a gain map, tables that fit the LDS): the output code through
    // GainMapSteps::locator with one table read, alpha through a table of output alpha codes per base alpha code
    const uint32_t * locator;
    const uint16_t * alphaLut; // 1 << baseL.depth entries
    uint32_t locFirstBits, locShift, locBuckets;
    uint32_t selBase[2], selOut[2]; // v_perm_b32 selectors: base pixel -> R, G, B, A order; codes in that order -> output pixel
    int32_t fast;
    int32_t convert;        // linearise, (convert primaries, apply the gain,) re-encode; 0: requantise the samples as they are
    int32_t inConv, outConv;
    double inM[9], outM[9]; // avifLinearRGBConvertColorSpace coefficients, row-major
    float baseOffset[3], altOffset[3];
    // statistics: one partial per workgroup, in pinned host memory (kGainMapMaxGroups entries); the caller adds them up in index order
    GainMapPartial * partials;
    // exact light levels (opt-in, api_gainmap.cpp): every pixel's max(0, r, g, b) of the tone-mapped linear values, `width` floats per row --
    // the host adds them up in the reference's order (src/gainmap.c:293: one fp32 accumulator, raster order); null otherwise
    float * pixelMax;
};
constexpr uint32_t kGainMapMaxGroups = 4096; // persistent workgroups of the apply kernel
// LDS the fast kernel may fill with tables (two workgroups per CU keep eight waves resident)
constexpr size_t kGainMapFastLdsBytes = 64 * 1024;
// what the fast kernel needs for 4- / 8-byte base pixels, a gain map of that depth and a locator of that many buckets
// (planes: the kernel converts the gain map's planes itself -- two more tables of 256 floats)
size_t gainMapFastLdsBytes(uint32_t basePixelBytes, uint32_t gainDepth, uint32_t locBuckets, bool planes = false);
// *partials: how many entries of args.partials the launch fills (0: no statistics)
hipError_t launchGainMapApply(const GainMapArgs & args, hipStream_t stream, uint32_t * partials);

// Gain-map computation (avifRGBImageComputeGainMap, reference src/gainmap.c:535-843): three passes over the pixels.
struct GainMapComputeArgs
{
    const uint8_t * base;
    const uint8_t * alt;
    uint32_t basePitch, altPitch;
    GainMapPixelLayout baseL, altL;
    uint32_t width, height;
    const float * baseLut; // linear light per sample code (gainmap_plan.h)
    const float * altLut;
    uint32_t baseLutEntries, altLutEntries; // tables of at most 4096 entries each are staged in LDS by the kernels (0: unknown, read from memory)
    int32_t convertAlt, convertBase; // at most one: which side goes through M into the other's primaries (:676-684)
    double M[9];
    int32_t singleChannel;
    float yCoeffs[3];
    float baseOffset[3], altOffset[3];
    float * ratios;   // channels x width*height: max((alt + offset) / (base + offset), 1e-10), :711-712
    float * partials; // kGainMapMaxGroups x 8 floats, see the kernels
    // pass 1 can take its offsets from device memory -- no host round trip between passes 0 and 1 (round 6): `offsets` = the six floats
    // launchGainMapOffsets left (base x 3, alternate x 3); nullptr: baseOffset / altOffset above are final
    const float * offsets;
};
// pass 0 (only when the primaries differ): per-workgroup minima of the converted side's channels, min(0, .), :624-645.
// partials[g * 8 + c], c < 3
hipError_t launchGainMapChannelMin(const GainMapComputeArgs & args, hipStream_t stream);
// between the two: pass 0's partials (`groups` x 8 floats) folded into the offsets that keep the converted side's channels positive (:646-660), by
// one workgroup: out[0..2] = base offsets, out[3..5] = alternate offsets, starting from `base` / `alt`.  The fold drops NaN partials (fminf) where
// the reference's AVIF_MIN lets them through: the host repeats it in the reference's order and runs pass 1 again if the two disagree.
hipError_t launchGainMapOffsets(const float * minima, uint32_t groups, bool useBaseColorSpace, const float base[3], const float alt[3], float * out, hipStream_t stream);
// pass 1: ratios + per-workgroup [baseMax, altMax (both >= 1, :663-664), minRatio[3], maxRatio[3]]
hipError_t launchGainMapRatios(const GainMapComputeArgs & args, hipStream_t stream);
struct GainMapStepTable
{
    const float * steps; // monotone steps over the ratio (gainmap_plan.h), `entries` (a power of two) floats
    uint32_t entries;
    uint32_t flip;       // index = flip - m when `flipped` (negative sign), else m
    int32_t flipped;
    // bucket tables (pass 2): index ~ guessA * log2(x) + guessB, corrected against the steps by the kernel (any values are safe; 0 / 0: bisection)
    float guessA, guessB;
};
// workgroups of passes 0 and 1 for an image: the host reads that many partials
uint32_t gainMapComputeGroups(uint32_t width, uint32_t height);
// pass 2: histogram[c][bucket] += 1 for every sample; channels with tables[c].entries == 0 are skipped
hipError_t launchGainMapHistogram(const float * ratios, size_t numPixels, int channels, const GainMapStepTable tables[3], uint32_t * const histograms[3],
                                  hipStream_t stream);
// pass 3: the gain map as RGBA of `depth` bits (avifRGBImageSetDefaults layout), alpha opaque; tables[c].entries == 0: code 0 (:766-773)
hipError_t launchGainMapQuantise(const float * ratios, uint32_t width, uint32_t height, int channels, const GainMapStepTable tables[3], uint8_t * rgba,
                                 uint32_t rgbaPitch, uint32_t depth, hipStream_t stream);

} // namespace avifhip
