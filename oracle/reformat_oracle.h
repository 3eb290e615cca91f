/*
 * reformat_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of libavif's pixel-reformat path, used only as the checker
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
 * shipped library (libavif_amd/csrc) never includes, links or calls this.
 *
 * Two arithmetic families are restated:
 *   oracle*            the reference's built-in floating-point path
 *                      (/root/reference/src/reformat.c, src/alpha.c), pinned
 *                      against the reference compiled from source
 *                      (oracle/_ref/libavif_ref.so, see oracle/Makefile);
 *   oracleLibyuv*      the fixed-point path libavif dispatches to in libyuv
 *                      (third-party, pinned 5d03bf9 / LIBYUV_VERSION 1949, source
 *                      absent from /root/reference; restated from SURVEY.md
 *                      Appendix D and pinned against Pillow's bundled
 *                      libavif 1.4.1 + libyuv 1922 binary).
 */
#ifndef AVIFHIP_REFORMAT_ORACLE_H
#define AVIFHIP_REFORMAT_ORACLE_H

#include "avifhip/avif_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/reformat.c:1649 (avifImageYUVToRGB), avoidLibYUV semantics (float path) */
avifResult oracleImageYUVToRGB(const avifImage * image, avifRGBImage * rgb);
/* reference src/reformat.c:221 (avifImageRGBToYUV); planes that are NULL are malloc'ed like src/avif.c:431 */
avifResult oracleImageRGBToYUV(avifImage * image, const avifRGBImage * rgb);
/* reference src/alpha.c:151 / :338 */
avifResult oracleRGBImagePremultiplyAlpha(avifRGBImage * rgb);
avifResult oracleRGBImageUnpremultiplyAlpha(avifRGBImage * rgb);
/* reference src/reformat.c:1778-1840 */
int oracleLimitedToFullY(uint32_t depth, int v);
int oracleLimitedToFullUV(uint32_t depth, int v);
int oracleFullToLimitedY(uint32_t depth, int v);
int oracleFullToLimitedUV(uint32_t depth, int v);

/*
 * Same conversion as oracleImageYUVToRGB but only for the sub-rectangle `rect`
 * of a stitched canvas, with the chroma-edge rules of src/reformat.c:768,784
 * evaluated against the CANVAS: the result equals the
 * corresponding rectangle of a whole-canvas conversion.  rgb describes the
 * whole RGB canvas.  rect->x / rect->y must be even for subsampled formats.
 */
avifResult oracleImageYUVToRGBRect(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * rect);

/*
 * Decode-side tail of a grid image, the way the reference runs it: every tile copied into the
 * canvas with the last column / row cropped (avifDecoderDataCopyTileToImage, src/read.c:1823-1877 = avifImageSetViewRect
 * + avifImageCopySamples), limited-range alpha tiles converted to full range sample by sample first
 * (avifImageLimitedToFullAlpha, src/read.c:6724-6764), then avifImageYUVToRGB on the canvas.  colorTiles / alphaTiles:
 * rows*columns images, row-major (alphaTiles may be NULL); metadata is taken from colorTiles[0].  libyuvBuild selects the
 * integer-path restatement (a libavif built with libyuv) for the conversion.
 */
typedef struct oracleGrid
{
    uint32_t rows, columns;
    uint32_t outputWidth, outputHeight;
} oracleGrid;
avifResult oracleGridYUVToRGB(const oracleGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                              avifBool alphaIsLimitedRange, avifRGBImage * rgb, int libyuvBuild);

/*
 * The pixel transforms libavif's tools apply to the converted RGB image, apps/shared/avifutil.c:787-825
 * (avifApplyTransforms): clean-aperture crop (:667-682), rotation by angle * 90 degrees anti-clockwise (:687-743), mirror
 * about the horizontal (axis 0) / vertical (axis 1) axis (:745-785), in that order.  Writes the transformed image into
 * dst (caller-allocated, transformed size).  crop may be NULL; rotate / mirror false = box absent.
 */
avifResult oracleRGBImageTransform(avifRGBImage * dst, const avifRGBImage * src, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                   avifBool mirror, uint8_t axis);

/*
 * avifImageApplyOperations (include/avif/internal.h:247-254, src/sampletransform.c:284-421): the postfix expression `tokens`
 * evaluated per sample of the selected planes in saturating 32-bit arithmetic, clamped to the destination depth.
 */
avifResult oracleImageApplyOperations(avifImage * dstImage, avifSampleTransformBitDepth bitDepth, uint32_t numTokens,
                                      const avifSampleTransformToken * tokens, uint8_t numInputImageItems, const avifImage * const * inputImageItems,
                                      avifPlanesFlags planes);

/*
 * avifImageScale (src/scale.c:23-201; vendored libyuv scaler under kFilterBox, third_party/libyuv/source/scale*.c): every
 * plane of `image` is replaced by its scaled version (scale_oracle.c).  The diagnostics argument of the reference is dropped.
 */
avifResult oracleImageScale(avifImage * image, uint32_t dstWidth, uint32_t dstHeight);

/*
 * The reference's INTEGER path: what a libavif built with libyuv computes (libyuv_oracle.c).
 * oracleLibyuv<Entry> == that build's avif<Entry>, end to end: libyuv's fixed-point arithmetic wherever libavif
 * dispatches to libyuv (src/reformat_libyuv.c; honours rgb->avoidLibYUV like src/reformat.c:1453 and :264), the fp32
 * path for everything else.  Note that (un)premultiply always tries libyuv first (src/alpha.c:163,350).
 */
avifResult oracleLibyuvImageYUVToRGB(const avifImage * image, avifRGBImage * rgb);
avifResult oracleLibyuvImageRGBToYUV(avifImage * image, const avifRGBImage * rgb);
avifResult oracleLibyuvRGBImagePremultiplyAlpha(avifRGBImage * rgb);
avifResult oracleLibyuvRGBImageUnpremultiplyAlpha(avifRGBImage * rgb);
/* The four backend hooks themselves (include/avif/internal.h:346-378): AVIF_RESULT_NOT_IMPLEMENTED for every
 * combination libavif would not hand to libyuv. */
avifResult oracleLibyuvHookYUVToRGB(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, avifBool * alphaReformatted);
avifResult oracleLibyuvHookRGBToYUV(avifImage * image, const avifRGBImage * rgb);
avifResult oracleLibyuvHookPremultiplyAlpha(avifRGBImage * rgb);
avifResult oracleLibyuvHookUnpremultiplyAlpha(avifRGBImage * rgb);

/* ---- gain-map application (gainmap_oracle.c): avifRGBImageApplyGainMap, reference src/gainmap.c:73-315 ---- */
avifResult oracleRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                      avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                      avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                      avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, int libyuvBuild);
avifResult oracleRGBImageComputeGainMap(const avifRGBImage * baseRgb, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC,
                                        const avifRGBImage * altRgb, avifColorPrimaries altPrimaries, avifTransferCharacteristics altTC,
                                        avifGainMap * gainMap, int libyuvBuild);
avifResult oracleFindMinMaxWithoutOutliers(const float * gainMapF, size_t numPixels, float * rangeMin, float * rangeMax);
int oracleDoubleToSignedFraction(double v, avifSignedFraction * fraction);
int oracleDoubleToUnsignedFraction(double v, avifUnsignedFraction * fraction);
avifResult oracleGainMapValidateMetadata(const avifGainMap * gainMap);
float oracleGainMapWeight(float hdrHeadroom, const avifGainMap * gainMap);
float oracleTransferFunction(int transferCharacteristics, int direction, float v);
int oracleColorPrimariesComputeRGBToRGBMatrix(int srcPrimaries, int dstPrimaries, double coeffs[3][3]);

#ifdef __cplusplus
}
#endif

#endif
