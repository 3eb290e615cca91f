/*
 * avif_abi.h -- layout-compatible mirror of the libavif boundary types used by
 * the pixel-reformat path (avifImage / avifRGBImage and their enums).
 *
 * This header is NOT a copy of libavif's avif.h: it declares only the fields
 * the reformat path reads or writes, under the names a libavif consumer uses,
 * so that the same C source compiles against either header.  When libavif's
 * own <avif/avif.h> has already been included (AVIF_AVIF_H defined) this file
 * declares nothing and libavif's definitions are used as they are.
 *
 * Layout facts mirrored (libavif v1.2.0 .. v1.4.2, LP64):
 *   avifImage      sizeof 224: width@0 height@4 depth@8 yuvFormat@12 yuvRange@16
 *                  yuvChromaSamplePosition@20 yuvPlanes@24 yuvRowBytes@48
 *                  imageOwnsYUVPlanes@60 alphaPlane@64 alphaRowBytes@72
 *                  imageOwnsAlphaPlane@76 alphaPremultiplied@80 (icc@88)
 *                  colorPrimaries@104 transferCharacteristics@106
 *                  matrixCoefficients@108            (reference include/avif/avif.h:777-851)
 *   avifRGBImage   sizeof 64: width@0 height@4 depth@8 format@12 chromaUpsampling@16
 *                  chromaDownsampling@20 avoidLibYUV@24 ignoreAlpha@28
 *                  alphaPremultiplied@32 isFloat@36 maxThreads@40 pixels@48
 *                  rowBytes@56                        (reference include/avif/avif.h:996-1018)
 * The offsets are enforced below with static assertions and were probed
 * against the reference header with offsetof() ("Boundary").
 */
#ifndef AVIFHIP_AVIF_ABI_H
#define AVIFHIP_AVIF_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifndef AVIF_AVIF_H /* libavif's own header not in use: provide the mirror */
#define AVIFHIP_ABI_MIRROR 1

#ifdef __cplusplus
extern "C" {
#endif

typedef int avifBool; /* avif.h:87 */
#define AVIF_TRUE 1
#define AVIF_FALSE 0

#define AVIF_PLANE_COUNT_YUV 3 /* avif.h:112 */

enum /* avifChannelIndex, avif.h:137-146 */
{
    AVIF_CHAN_Y = 0,
    AVIF_CHAN_U = 1,
    AVIF_CHAN_V = 2,
    AVIF_CHAN_A = 3
};

enum /* avifPlanesFlag, avif.h:127-133 */
{
    AVIF_PLANES_YUV = 1,
    AVIF_PLANES_A = 2,
    AVIF_PLANES_ALL = 0xff
};
typedef uint32_t avifPlanesFlags;

/* Result codes: only the values the reformat path can return are named (avif.h:163-205). */
typedef enum avifResult
{
    AVIF_RESULT_OK = 0,
    AVIF_RESULT_UNKNOWN_ERROR = 1,
    AVIF_RESULT_REFORMAT_FAILED = 5,
    AVIF_RESULT_BMFF_PARSE_FAILED = 9,
    AVIF_RESULT_INVALID_IMAGE_GRID = 18,
    AVIF_RESULT_INVALID_ARGUMENT = 24,
    AVIF_RESULT_NOT_IMPLEMENTED = 25,
    AVIF_RESULT_OUT_OF_MEMORY = 26,
    AVIF_RESULT_INTERNAL_ERROR = 29,
    AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE = 32
} avifResult;

typedef enum avifPixelFormat /* avif.h:280-289 */
{
    AVIF_PIXEL_FORMAT_NONE = 0,
    AVIF_PIXEL_FORMAT_YUV444 = 1,
    AVIF_PIXEL_FORMAT_YUV422 = 2,
    AVIF_PIXEL_FORMAT_YUV420 = 3,
    AVIF_PIXEL_FORMAT_YUV400 = 4,
    AVIF_PIXEL_FORMAT_COUNT = 5
} avifPixelFormat;

typedef enum avifChromaSamplePosition /* avif.h:310-316 */
{
    AVIF_CHROMA_SAMPLE_POSITION_UNKNOWN = 0,
    AVIF_CHROMA_SAMPLE_POSITION_VERTICAL = 1,
    AVIF_CHROMA_SAMPLE_POSITION_COLOCATED = 2
} avifChromaSamplePosition;

typedef enum avifRange /* avif.h:322-328 */
{
    AVIF_RANGE_LIMITED = 0,
    AVIF_RANGE_FULL = 1
} avifRange;

/* CICP code points are 16-bit integers in libavif (avif.h:357,387,414). */
typedef uint16_t avifColorPrimaries;
typedef uint16_t avifTransferCharacteristics;
typedef uint16_t avifMatrixCoefficients;

enum /* avif.h:394-413 */
{
    AVIF_MATRIX_COEFFICIENTS_IDENTITY = 0,
    AVIF_MATRIX_COEFFICIENTS_BT709 = 1,
    AVIF_MATRIX_COEFFICIENTS_UNSPECIFIED = 2,
    AVIF_MATRIX_COEFFICIENTS_FCC = 4,
    AVIF_MATRIX_COEFFICIENTS_BT470BG = 5,
    AVIF_MATRIX_COEFFICIENTS_BT601 = 6,
    AVIF_MATRIX_COEFFICIENTS_SMPTE240 = 7,
    AVIF_MATRIX_COEFFICIENTS_YCGCO = 8,
    AVIF_MATRIX_COEFFICIENTS_BT2020_NCL = 9,
    AVIF_MATRIX_COEFFICIENTS_BT2020_CL = 10,
    AVIF_MATRIX_COEFFICIENTS_SMPTE2085 = 11,
    AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL = 12,
    AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_CL = 13,
    AVIF_MATRIX_COEFFICIENTS_ICTCP = 14,
    AVIF_MATRIX_COEFFICIENTS_YCGCO_RE = 16,
    AVIF_MATRIX_COEFFICIENTS_YCGCO_RO = 17,
    AVIF_MATRIX_COEFFICIENTS_LAST = 18
};

typedef struct avifCropRect /* avif.h:547-553 */
{
    uint32_t x, y, width, height;
} avifCropRect;

/*
 * avifImage: the leading, version-stable part of libavif's struct, followed by
 * an opaque tail so that sizeof matches (224).  Never rely on the tail; images
 * that must be handed to libavif proper have to come from avifImageCreate().
 */
typedef struct avifImage
{
    uint32_t width;
    uint32_t height;
    uint32_t depth; /* 8, 10, 12 (16 accepted by the state check); >8 => uint16_t samples */

    avifPixelFormat yuvFormat;
    avifRange yuvRange;
    avifChromaSamplePosition yuvChromaSamplePosition;
    uint8_t * yuvPlanes[AVIF_PLANE_COUNT_YUV];
    uint32_t yuvRowBytes[AVIF_PLANE_COUNT_YUV];
    avifBool imageOwnsYUVPlanes;

    uint8_t * alphaPlane;
    uint32_t alphaRowBytes;
    avifBool imageOwnsAlphaPlane;
    avifBool alphaPremultiplied;

    uint64_t avifhipOpaqueIcc_[2]; /* avifRWData icc @88 (pointer + size) */

    avifColorPrimaries colorPrimaries;
    avifTransferCharacteristics transferCharacteristics;
    avifMatrixCoefficients matrixCoefficients;

    uint8_t avifhipOpaqueTail_[224 - 110]; /* clli, transforms, exif, xmp, properties, gainMap */
} avifImage;

typedef enum avifRGBFormat /* avif.h:948-971 */
{
    AVIF_RGB_FORMAT_RGB = 0,
    AVIF_RGB_FORMAT_RGBA = 1,
    AVIF_RGB_FORMAT_ARGB = 2,
    AVIF_RGB_FORMAT_BGR = 3,
    AVIF_RGB_FORMAT_BGRA = 4,
    AVIF_RGB_FORMAT_ABGR = 5,
    AVIF_RGB_FORMAT_RGB_565 = 6,
    AVIF_RGB_FORMAT_GRAY = 7,
    AVIF_RGB_FORMAT_GRAYA = 8,
    AVIF_RGB_FORMAT_AGRAY = 9,
    AVIF_RGB_FORMAT_COUNT = 10
} avifRGBFormat;

typedef enum avifChromaUpsampling /* avif.h:975-983 */
{
    AVIF_CHROMA_UPSAMPLING_AUTOMATIC = 0,
    AVIF_CHROMA_UPSAMPLING_FASTEST = 1,
    AVIF_CHROMA_UPSAMPLING_BEST_QUALITY = 2,
    AVIF_CHROMA_UPSAMPLING_NEAREST = 3,
    AVIF_CHROMA_UPSAMPLING_BILINEAR = 4
} avifChromaUpsampling;

typedef enum avifChromaDownsampling /* avif.h:985-992 */
{
    AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC = 0,
    AVIF_CHROMA_DOWNSAMPLING_FASTEST = 1,
    AVIF_CHROMA_DOWNSAMPLING_BEST_QUALITY = 2,
    AVIF_CHROMA_DOWNSAMPLING_AVERAGE = 3,
    AVIF_CHROMA_DOWNSAMPLING_SHARP_YUV = 4
} avifChromaDownsampling;

typedef struct avifRGBImage /* avif.h:996-1018 */
{
    uint32_t width;
    uint32_t height;
    uint32_t depth; /* 8, 10, 12, 16; >8 => uint16_t channels */
    avifRGBFormat format;
    avifChromaUpsampling chromaUpsampling;
    avifChromaDownsampling chromaDownsampling;
    avifBool avoidLibYUV;
    avifBool ignoreAlpha;
    avifBool alphaPremultiplied;
    avifBool isFloat;
    int maxThreads;

    uint8_t * pixels;
    uint32_t rowBytes;
} avifRGBImage;

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* !AVIF_AVIF_H */

/* Layout contract, checked whichever header supplied the types. */
#if defined(__cplusplus)
#define AVIFHIP_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define AVIFHIP_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
AVIFHIP_STATIC_ASSERT(sizeof(avifRGBImage) == 64, "avifRGBImage layout");
AVIFHIP_STATIC_ASSERT(offsetof(avifRGBImage, pixels) == 48, "avifRGBImage.pixels");
AVIFHIP_STATIC_ASSERT(offsetof(avifRGBImage, rowBytes) == 56, "avifRGBImage.rowBytes");
AVIFHIP_STATIC_ASSERT(sizeof(avifImage) == 224, "avifImage layout");
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, yuvPlanes) == 24, "avifImage.yuvPlanes");
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, yuvRowBytes) == 48, "avifImage.yuvRowBytes");
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, alphaPlane) == 64, "avifImage.alphaPlane");
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, alphaPremultiplied) == 80, "avifImage.alphaPremultiplied");
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, matrixCoefficients) == 108, "avifImage.matrixCoefficients");
#ifdef AVIFHIP_ABI_MIRROR /* (members of the mirror only) */
AVIFHIP_STATIC_ASSERT(offsetof(avifImage, avifhipOpaqueIcc_) == 88 && offsetof(avifImage, avifhipOpaqueTail_) == 110, "avifImage.icc / .clli");
#endif


/* ---- gain maps (avifRGBImageApplyGainMap, reference src/gainmap.c): boundary types, avif.h:236-250, :419-453, :582-610, :630-711 ---- */
#ifdef AVIFHIP_ABI_MIRROR
#ifdef __cplusplus
extern "C" {
#endif
enum /* avifColorPrimaries values, avif.h:336-355 */
{
    AVIF_COLOR_PRIMARIES_UNKNOWN = 0,
    AVIF_COLOR_PRIMARIES_BT709 = 1,
    AVIF_COLOR_PRIMARIES_UNSPECIFIED = 2,
    AVIF_COLOR_PRIMARIES_BT470M = 4,
    AVIF_COLOR_PRIMARIES_BT470BG = 5,
    AVIF_COLOR_PRIMARIES_BT601 = 6,
    AVIF_COLOR_PRIMARIES_SMPTE240 = 7,
    AVIF_COLOR_PRIMARIES_GENERIC_FILM = 8,
    AVIF_COLOR_PRIMARIES_BT2020 = 9,
    AVIF_COLOR_PRIMARIES_XYZ = 10,
    AVIF_COLOR_PRIMARIES_SMPTE431 = 11,
    AVIF_COLOR_PRIMARIES_SMPTE432 = 12,
    AVIF_COLOR_PRIMARIES_EBU3213 = 22
};
enum /* avifTransferCharacteristics values, avif.h:364-385 */
{
    AVIF_TRANSFER_CHARACTERISTICS_UNKNOWN = 0,
    AVIF_TRANSFER_CHARACTERISTICS_BT709 = 1,
    AVIF_TRANSFER_CHARACTERISTICS_UNSPECIFIED = 2,
    AVIF_TRANSFER_CHARACTERISTICS_BT470M = 4,
    AVIF_TRANSFER_CHARACTERISTICS_BT470BG = 5,
    AVIF_TRANSFER_CHARACTERISTICS_BT601 = 6,
    AVIF_TRANSFER_CHARACTERISTICS_SMPTE240 = 7,
    AVIF_TRANSFER_CHARACTERISTICS_LINEAR = 8,
    AVIF_TRANSFER_CHARACTERISTICS_LOG100 = 9,
    AVIF_TRANSFER_CHARACTERISTICS_LOG100_SQRT10 = 10,
    AVIF_TRANSFER_CHARACTERISTICS_IEC61966 = 11,
    AVIF_TRANSFER_CHARACTERISTICS_BT1361 = 12,
    AVIF_TRANSFER_CHARACTERISTICS_SRGB = 13,
    AVIF_TRANSFER_CHARACTERISTICS_BT2020_10BIT = 14,
    AVIF_TRANSFER_CHARACTERISTICS_BT2020_12BIT = 15,
    AVIF_TRANSFER_CHARACTERISTICS_PQ = 16,
    AVIF_TRANSFER_CHARACTERISTICS_SMPTE2084 = 16,
    AVIF_TRANSFER_CHARACTERISTICS_SMPTE428 = 17,
    AVIF_TRANSFER_CHARACTERISTICS_HLG = 18
};
typedef struct avifRWData
{
    uint8_t * data;
    size_t size;
} avifRWData;
#define AVIF_DIAGNOSTICS_ERROR_BUFFER_SIZE 256
typedef struct avifDiagnostics
{
    char error[AVIF_DIAGNOSTICS_ERROR_BUFFER_SIZE];
} avifDiagnostics;
typedef struct avifSignedFraction
{
    int32_t n;
    uint32_t d;
} avifSignedFraction;
typedef struct avifUnsignedFraction
{
    uint32_t n;
    uint32_t d;
} avifUnsignedFraction;
typedef struct avifContentLightLevelInformationBox
{
    uint16_t maxCLL;
    uint16_t maxPALL;
} avifContentLightLevelInformationBox;
typedef struct avifGainMap
{
    struct avifImage * image; /* the gain map pixels (YUV planes; CICP fields ignored) */
    avifSignedFraction gainMapMin[3];
    avifSignedFraction gainMapMax[3];
    avifUnsignedFraction gainMapGamma[3];
    avifSignedFraction baseOffset[3];
    avifSignedFraction alternateOffset[3];
    avifUnsignedFraction baseHdrHeadroom;
    avifUnsignedFraction alternateHdrHeadroom;
    avifBool useBaseColorSpace;
    avifRWData altICC;
    avifColorPrimaries altColorPrimaries;
    avifTransferCharacteristics altTransferCharacteristics;
    avifMatrixCoefficients altMatrixCoefficients;
    avifRange altYUVRange;
    uint32_t altDepth;
    uint32_t altPlaneCount;
    avifContentLightLevelInformationBox altCLLI;
} avifGainMap;
#ifdef __cplusplus
}
#endif
AVIFHIP_STATIC_ASSERT(sizeof(avifGainMap) == 192, "avifGainMap layout");
AVIFHIP_STATIC_ASSERT(offsetof(avifGainMap, baseHdrHeadroom) == 128, "avifGainMap.baseHdrHeadroom");
AVIFHIP_STATIC_ASSERT(offsetof(avifGainMap, useBaseColorSpace) == 144, "avifGainMap.useBaseColorSpace");
AVIFHIP_STATIC_ASSERT(offsetof(avifGainMap, altColorPrimaries) == 168, "avifGainMap.altColorPrimaries");
#endif /* AVIFHIP_ABI_MIRROR */

/* Sample Transform tokens ('sato' derived image items): libavif declares these in its INTERNAL header
 * (include/avif/internal.h:179-228); mirrored here, layout-identical (sizeof(avifSampleTransformToken) == 12), unless that
 * header is in use. */
#ifndef AVIF_INTERNAL_H
#ifdef __cplusplus
extern "C" {
#endif
typedef enum avifSampleTransformBitDepth
{
    AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_8 = 0,
    AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_16 = 1,
    AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_32 = 2,
    AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_64 = 3
} avifSampleTransformBitDepth;
typedef enum avifSampleTransformTokenType
{
    AVIF_SAMPLE_TRANSFORM_CONSTANT = 0,
    AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX = 1,
    AVIF_SAMPLE_TRANSFORM_FIRST_UNARY_OPERATOR = 64,
    AVIF_SAMPLE_TRANSFORM_NEGATION = 64,
    AVIF_SAMPLE_TRANSFORM_ABSOLUTE = 65,
    AVIF_SAMPLE_TRANSFORM_NOT = 66,
    AVIF_SAMPLE_TRANSFORM_BSR = 67,
    AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR = 128,
    AVIF_SAMPLE_TRANSFORM_SUM = 128,
    AVIF_SAMPLE_TRANSFORM_DIFFERENCE = 129,
    AVIF_SAMPLE_TRANSFORM_PRODUCT = 130,
    AVIF_SAMPLE_TRANSFORM_QUOTIENT = 131,
    AVIF_SAMPLE_TRANSFORM_AND = 132,
    AVIF_SAMPLE_TRANSFORM_OR = 133,
    AVIF_SAMPLE_TRANSFORM_XOR = 134,
    AVIF_SAMPLE_TRANSFORM_POW = 135,
    AVIF_SAMPLE_TRANSFORM_MIN = 136,
    AVIF_SAMPLE_TRANSFORM_MAX = 137,
    AVIF_SAMPLE_TRANSFORM_RESERVED = 138
} avifSampleTransformTokenType;
typedef struct avifSampleTransformToken
{
    avifSampleTransformTokenType type;
    int32_t constant;            /* AVIF_SAMPLE_TRANSFORM_CONSTANT */
    uint8_t inputImageItemIndex; /* AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX, 1-based */
} avifSampleTransformToken;
#ifdef __cplusplus
}
#endif
#endif /* AVIF_INTERNAL_H */

#endif /* AVIFHIP_AVIF_ABI_H */
