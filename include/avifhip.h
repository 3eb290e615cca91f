/*
 * avifhip.h -- C ABI of libavifhip.so: libavif's pixel-reformat path
 * (avifImageYUVToRGB / avifImageRGBToYUV / alpha premultiply) executed by
 * hand-written HIP kernels on AMD MI355X (gfx950).
 *
 * The boundary is libavif's own: the same avifImage / avifRGBImage structs, the
 * same avifResult codes, the same argument meaning and error behaviour as the
 * reference functions each entry point replaces (cited per function,
 * file:line relative to the libavif source tree).  Plain C pointers and
 * sizes only.  Buffers reachable from the structs may live in host memory
 * (they are staged through HBM) or already in device memory (they are used in
 * place); the library classifies each pointer itself.
 *
 * Arithmetic contract (see DESIGN.md "Parity"): results are byte-identical to one of the two libavif builds,
 * selected by avifhipSetArithmetic() / the AVIFHIP_ARITHMETIC environment variable:
 *   AUTO (default)  a libavif built WITH libyuv (the stock build): libyuv's fixed-point arithmetic for every
 *                   combination libavif dispatches to libyuv (src/reformat_libyuv.c) -- honouring rgb->avoidLibYUV
 *                   the way src/reformat.c:1453 and :264 do, and asking libyuv first for 8-bit RGBA/BGRA
 *                   (un)premultiply whatever avoidLibYUV says, the way src/alpha.c:163,350 do -- and libavif's
 *                   built-in fp32 arithmetic (src/reformat.c, src/alpha.c) for everything else;
 *   FLOAT           a libavif built WITHOUT libyuv: the built-in fp32 arithmetic everywhere;
 *   LIBYUV          as AUTO with rgb->avoidLibYUV ignored.
 */
#ifndef AVIFHIP_H
#define AVIFHIP_H

#include "avifhip/avif_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define AVIFHIP_API __attribute__((visibility("default")))

/* ---- the four public conversion entry points (libavif seam A) ------------------------------- */

/* Replaces avifImageYUVToRGB, include/avif/avif.h:1032, src/reformat.c:1649-1748.
 * One fused kernel does colour conversion, chroma upsampling, alpha copy/fill/rescale,
 * (un)premultiply and the half-float pass that the reference runs as up to four passes. */
AVIFHIP_API avifResult avifhipImageYUVToRGB(const avifImage * image, avifRGBImage * rgb);

/* Replaces avifImageRGBToYUV, include/avif/avif.h:1031, src/reformat.c:221-571.
 * Planes that are NULL are allocated with malloc exactly like avifImageAllocatePlanes
 * (src/avif.c:431-490) and marked image-owned. */
AVIFHIP_API avifResult avifhipImageRGBToYUV(avifImage * image, const avifRGBImage * rgb);

/* Replace avifRGBImagePremultiplyAlpha / avifRGBImageUnpremultiplyAlpha,
 * include/avif/avif.h:1037-1038, src/alpha.c:151-336 / :338-535 (in place). */
AVIFHIP_API avifResult avifhipRGBImagePremultiplyAlpha(avifRGBImage * rgb);
AVIFHIP_API avifResult avifhipRGBImageUnpremultiplyAlpha(avifRGBImage * rgb);

/* ---- libavif's accelerated-backend hooks (seam B, include/avif/internal.h:349-386) ---------------- */

/* The job libavif's avifImageYUVToRGBImpl hands to avifImageYUVToRGBLibYUV (src/reformat.c:1453-1461,
 * src/reformat_libyuv.c:932-1108): colour conversion WITHOUT alpha (un)multiply and without the half-float pass (the
 * caller runs those afterwards through the hooks below); the alpha channel is written (from the alpha plane, or
 * opaque) only when reformatAlpha is set. */
AVIFHIP_API avifResult avifhipImageYUVToRGBColorOnly(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha);
/* The same job with libavif's NEXT steps folded in: after AVIF_RESULT_OK from its colour hook, avifImageYUVToRGBImpl calls
 * avifRGBImagePremultiplyAlpha / avifRGBImageUnpremultiplyAlpha on the same pixels when an alpha (un)multiply is pending and
 * avifRGBImageToF16 when rgb->isFloat (src/reformat.c:1574-1590) -- each another hook call that would move a host-resident image across
 * the bus both ways.  When those steps can be computed in the same pass (the integer post-pass after the conversion: what a
 * libyuv-backed libavif computes too), the pixels handed back are FINAL and *folded says which follow-up calls the caller must answer
 * with AVIF_RESULT_OK without touching the pixels (AVIFHIP_FOLDED_*); otherwise *folded is 0 and the result is
 * avifhipImageYUVToRGBColorOnly's.  integration/reformat_libyuv_hip.c keeps the one-shot bookkeeping. */
#define AVIFHIP_FOLDED_PREMULTIPLY 1u
#define AVIFHIP_FOLDED_UNPREMULTIPLY 2u
#define AVIFHIP_FOLDED_TO_F16 4u
AVIFHIP_API avifResult avifhipImageYUVToRGBHook(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, uint32_t * folded);
/* Replaces avifRGBImageToF16 / avifRGBImageToF16LibYUV (src/reformat.c:1419-1443): in-place uint16 -> IEEE half. */
AVIFHIP_API avifResult avifhipRGBImageToF16(avifRGBImage * rgb);

/* Rectangles of a HOST-resident canvas, converted into the same rectangles of a host-resident RGB canvas: what one rank of a
 * tile farm does with its share of an AVIF grid (tiles stitched by src/read.c:1823-1877, then converted), or one band worker with
 * its band.  Each rectangle equals the same rectangle of a whole-canvas avifImageYUVToRGB byte for byte (chroma edge rules are
 * evaluated against the canvas, seams see their neighbours).  Only the plane samples the rectangles need cross the host link
 * (their own plus the one-sample chroma halo of the bilinear filter), only the rectangles come back; uploads, kernels and
 * downloads of successive rectangles overlap.  Rectangle origins must lie on the chroma grid (like avifImageSetViewRect,
 * src/avif.c:335-337); pixels outside the rectangles are left untouched.  Synchronous. */
AVIFHIP_API avifResult avifhipImageYUVToRGBRects(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * rects, uint32_t count);
/* The bytes that call moves over the host link for these rectangles (needs no device): *bytesUp, *bytesDown. */
AVIFHIP_API avifResult avifhipPlanRectTransfers(const avifImage * canvas, const avifRGBImage * rgbCanvas, const avifCropRect * rects, uint32_t count, uint64_t * bytesUp,
                                                uint64_t * bytesDown);
/* ... and what the calling thread's last host-resident conversion (avifhipImageYUVToRGBRects, avifhipImageYUVToRGB, avifhipImageRGBToYUV, the
 * in-place alpha / half-float passes) actually moved over the host link(s). */
AVIFHIP_API void avifhipLastTransferBytes(uint64_t * bytesUp, uint64_t * bytesDown);

/* ---- device-resident / asynchronous variants ------------------------------------------------- */

/* Same conversions with every buffer already in device memory, enqueued on `hipStream`
 * (a hipStream_t passed as void*; NULL = the calling thread's library stream) WITHOUT
 * synchronising.  No pointer classification, no staging, no allocation.
 * avifhipImageRGBToYUVAsync requires all destination planes to be present. */
AVIFHIP_API avifResult avifhipImageYUVToRGBAsync(const avifImage * image, avifRGBImage * rgb, void * hipStream);
AVIFHIP_API avifResult avifhipImageRGBToYUVAsync(avifImage * image, const avifRGBImage * rgb, void * hipStream);
AVIFHIP_API avifResult avifhipRGBImagePremultiplyAlphaAsync(avifRGBImage * rgb, void * hipStream);
AVIFHIP_API avifResult avifhipRGBImageUnpremultiplyAlphaAsync(avifRGBImage * rgb, void * hipStream);

/* `count` device-resident RGB -> YUV conversions enqueued on `hipStream` (the frames of an image sequence on their way into an encoder;
 * src/reformat.c:161-519 per frame, same rules and results as `count` calls of avifhipImageRGBToYUVAsync, destination planes allocated
 * by the caller).  Frames of 2 megapixels or more that differ in their buffers only share launches of the single-image kernel, up to 8
 * frames per launch, so that a launch's ramp and tail are paid once per 8 frames; any other mix is converted frame by frame. */
AVIFHIP_API avifResult avifhipImageRGBToYUVBatchAsync(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, void * hipStream);

/* Grid tiles / sequence frames.  Converts the sub-rectangle `rect` of a stitched
 * canvas; chroma edge rules (src/reformat.c:768,784) are evaluated against the canvas, so the
 * output equals the same rectangle of a whole-canvas avifImageYUVToRGB.  Plane pointers of
 * `canvas` address canvas sample (0,0); only samples inside the rectangle plus a one-chroma-sample
 * halo are read.  rect->x / rect->y must be even for subsampled formats (src/avif.c:335-337). */
AVIFHIP_API avifResult avifhipImageYUVToRGBRectAsync(const avifImage * canvas,
                                                     avifRGBImage * rgbCanvas,
                                                     const avifCropRect * rect,
                                                     void * hipStream);

/* One launch for `count` independent device-resident conversions (tile farm: count tiles of a
 * grid, or count frames of a sequence).  All images must share one kernel configuration
 * (format, depth, range, matrix, RGB format...); rects may be NULL (whole images). */
AVIFHIP_API avifResult avifhipImageYUVToRGBBatchAsync(uint32_t count,
                                                      const avifImage * const * images,
                                                      avifRGBImage * const * rgbs,
                                                      const avifCropRect * rects,
                                                      void * hipStream);

/* Decode-side tail of a grid image in one step: replaces, for a grid whose tiles were decoded into
 * separate device-resident images,
 *   avifDecoderDataCopyTileToImage x (rows * columns)   tile -> canvas copy, last column / row cropped (src/read.c:1823-1877)
 *   avifImageLimitedToFullAlpha                         limited-range alpha tiles of old files (src/read.c:6724-6764,6822)
 *   avifImageYUVToRGB(canvas, rgb)                      on the stitched canvas (src/reformat.c:1649)
 * without materialising the YUV canvas: tiles are converted where they lie, with the chroma filter reaching across tile
 * seams exactly as it would on the stitched canvas.  The result equals that three-step sequence byte for byte.
 * One kernel launch where every pixel goes through the tiled kernels and the tiles share their chroma pitches (the tiles are linked
 * to their neighbours and the kernels read the chroma sample beyond a seam from the neighbouring tile); otherwise, and for the
 * fp32 kernels on canvases above 32 megapixels, the tiles first and the pixels along the seams again in a second launch.
 * AVIFHIP_GRID_SEAM_PASS=1 / 0 in the environment forces either; same bytes.
 *   grid         rows, columns, outputWidth, outputHeight as in libavif's avifImageGrid (include/avif/internal.h)
 *   colorTiles   rows * columns images, row-major, all with the first tile's geometry and format (src/read.c:1832-1842);
 *                CICP, range and alphaPremultiplied are taken from colorTiles[0]
 *   alphaTiles   NULL, or rows * columns images whose alphaPlane holds the decoded alpha tile (same tiling)
 *   alphaIsLimitedRange   the alpha item is limited range (src/read.c:6822)
 * Buffers are device-resident; enqueued on `hipStream` without synchronising. */
typedef struct avifhipGrid
{
    uint32_t rows, columns;
    uint32_t outputWidth, outputHeight;
} avifhipGrid;
AVIFHIP_API avifResult avifhipGridYUVToRGBAsync(const avifhipGrid * grid,
                                                const avifImage * const * colorTiles,
                                                const avifImage * const * alphaTiles,
                                                avifBool alphaIsLimitedRange,
                                                avifRGBImage * rgbCanvas,
                                                void * hipStream);

/* The application-side pixel transforms libavif's tools apply to the converted RGB image, as ONE pass from `src` into
 * `dst` (device-resident, enqueued on `hipStream`): replaces avifApplyTransforms (apps/shared/avifutil.c:787-825), i.e.
 * clean-aperture crop (avifRGBImageSetViewRect, :667-682), then rotation by angle * 90 degrees anti-clockwise
 * (avifRGBImageRotate, :687-743), then mirroring about the horizontal (axis 0) or vertical (axis 1) axis
 * (avifRGBImageMirror, :745-785) -- the order MIAF 7.3.6.7 prescribes.  crop may be NULL (no 'clap'); pass rotate / mirror
 * = AVIF_FALSE for an absent 'irot' / 'imir'.  dst must have the format and depth of src and the transformed size
 * (crop size, swapped for angle 1 and 3).  The reference runs up to two full passes plus an allocation for this. */
AVIFHIP_API avifResult avifhipRGBImageTransformAsync(avifRGBImage * dst,
                                                     const avifRGBImage * src,
                                                     const avifCropRect * crop,
                                                     avifBool rotate,
                                                     uint8_t angle,
                                                     avifBool mirror,
                                                     uint8_t axis,
                                                     void * hipStream);

/* Gain-map application (tone mapping), the decode-side consumer of the conversion path: drop-ins for
 * avifRGBImageApplyGainMap / avifImageApplyGainMap (reference include/avif/avif.h:1727-1745, src/gainmap.c:73-355) with the
 * same arguments, result codes and diagnostics.  Output bytes are IDENTICAL to the reference's on this machine: every libm
 * transcendental of the reference is tabulated on the host with the host's libm (per sample code on the input side; as the
 * fp32 steps of the quantised output transfer function on the output side), the GPU does the IEEE arithmetic in between.
 * The only tolerance: clli->maxPALL, which the reference accumulates in fp32 pixel by pixel (order-dependent rounding) and
 * this library in fp64 partial sums -- it may differ by rounding of the last nit -- unless avifhipSetExactLightLevels(1) (or
 * AVIFHIP_EXACT_LIGHT_LEVELS=1) asks for the reference's own sum: the kernels then also leave every pixel's maximum, and the host adds them
 * up in one fp32 accumulator in raster order like src/gainmap.c:293 does (8 ms of host time and a 33 MB download for a 4K image: opt-in).  The gain map's own YUV -> RGB conversion
 * follows the library's arithmetic setting (default: what a libavif built with libyuv computes).
 * avifhipRGBImageApplyGainMap: host images; toneMappedImage->pixels is (re)allocated with malloc like the reference does -- except that a
 * buffer the struct already holds is kept when malloc_usable_size() says it has the size the reference would allocate (same bytes, same
 * ownership; a caller that tone-maps a sequence into one avifRGBImage is spared the release of pinned memory and the page faults of a
 * fresh buffer: 8.7 -> 2.3 ms per 4K call).  avifhipRGBImageComputeGainMap keeps gainMap->image's planes the same way.
 * ...Async: base pixels, gain-map planes and (pre-allocated) tone-mapped pixels are device-resident; the call enqueues on
 * `hipStream` and WAITS for it when its answer depends on the pixels: the CLLI values (clli != NULL), or the result code where the
 * curves' tables cannot rule a NaN out.  With clli == NULL and tables that prove no NaN can arise (4-channel integer pixels up to
 * 12 bits: the fast kernel's precondition) it returns with its work enqueued, like every other Async entry point.  The tone-mapped
 * pixels may lie on top of the base pixels (same layout): such a call keeps to the general kernel, one lane per pixel. */
AVIFHIP_API avifResult avifhipRGBImageApplyGainMap(const avifRGBImage * baseImage,
                                                   avifColorPrimaries baseColorPrimaries,
                                                   avifTransferCharacteristics baseTransferCharacteristics,
                                                   const avifGainMap * gainMap,
                                                   float hdrHeadroom,
                                                   avifColorPrimaries outputColorPrimaries,
                                                   avifTransferCharacteristics outputTransferCharacteristics,
                                                   avifRGBImage * toneMappedImage,
                                                   avifContentLightLevelInformationBox * clli,
                                                   avifDiagnostics * diag);
AVIFHIP_API avifResult avifhipRGBImageApplyGainMapAsync(const avifRGBImage * baseImage,
                                                        avifColorPrimaries baseColorPrimaries,
                                                        avifTransferCharacteristics baseTransferCharacteristics,
                                                        const avifGainMap * gainMap,
                                                        float hdrHeadroom,
                                                        avifColorPrimaries outputColorPrimaries,
                                                        avifTransferCharacteristics outputTransferCharacteristics,
                                                        avifRGBImage * toneMappedImage,
                                                        avifContentLightLevelInformationBox * clli,
                                                        avifDiagnostics * diag,
                                                        void * hipStream);
/* Measurement helper (bench.py): the apply kernel of that ...Async call alone, milliseconds per launch over `iters` back-to-back launches
 * (HIP events on the launch stream) after `warmup` untimed ones; the gain map's own conversion and the tables are prepared once.  < 0 on failure. */
AVIFHIP_API double avifhipTimeRGBImageApplyGainMap(const avifRGBImage * baseImage,
                                                   avifColorPrimaries baseColorPrimaries,
                                                   avifTransferCharacteristics baseTransferCharacteristics,
                                                   const avifGainMap * gainMap,
                                                   float hdrHeadroom,
                                                   avifColorPrimaries outputColorPrimaries,
                                                   avifTransferCharacteristics outputTransferCharacteristics,
                                                   avifRGBImage * toneMappedImage,
                                                   int warmup,
                                                   int iters,
                                                   void * hipStream);
AVIFHIP_API void avifhipSetExactLightLevels(int on);
/* avifhipRGBImageApplyGainMapAsync with clli != NULL (round 6): where the fast kernel serves the call (4-channel integer pixels, tables that
 * fit the LDS, no NaN possible: the usual case) the call returns with its work enqueued like every other Async entry point, and *clli is
 * filled by the calling thread's next avifhipSynchronize(hipStream) (the statistics travel into pinned memory behind the kernel;
 * synchronising the stream through the HIP runtime directly does not fill it) -- keep it alive until then; at most 8 such calls per
 * thread stay unsettled (the ninth settles the oldest, waiting for it).  Every other case (and exact light levels) waits for the stream
 * before it returns, as before. */
/* Gain-map computation (the encode side): drop-in for avifRGBImageComputeGainMap (reference include/avif/avif.h:1688-1722,
 * src/gainmap.c:535-843): host images in; the metadata fractions of `gainMap` and the (malloc'ed) planes of gainMap->image -- whose
 * width, height, depth, yuvFormat (range, matrix) carry the request, as in the reference -- out.  Byte-identical planes and
 * metadata: the kernels carry the exact fp32 ratio of every sample; log2f / powf, the outlier histogram's bucket index and the
 * final quantisation are monotone step functions of that ratio whose steps the host finds by bisection with its own libm. */
AVIFHIP_API avifResult avifhipRGBImageComputeGainMap(const avifRGBImage * baseRgbImage,
                                                     avifColorPrimaries baseColorPrimaries,
                                                     avifTransferCharacteristics baseTransferCharacteristics,
                                                     const avifRGBImage * altRgbImage,
                                                     avifColorPrimaries altColorPrimaries,
                                                     avifTransferCharacteristics altTransferCharacteristics,
                                                     avifGainMap * gainMap,
                                                     avifDiagnostics * diag);
/* ... with everything in device memory: the two renditions' pixels, and the planes of gainMap->image, which the caller allocates at the
 * requested size (width, height, depth, yuvFormat as in the host call; yuvRowBytes as allocated).  The metadata fractions are final when
 * the call returns -- they are functions of every pixel (channel minima, extreme ratios, the outlier histogram: src/gainmap.c:618-749), so the
 * call waits for its first passes on `hipStream` (NULL = the calling thread's library stream) -- the planes once that stream has drained.
 * Same bytes and metadata as avifhipRGBImageComputeGainMap on the downloaded copies.  avifhipTimeRGBImageComputeGainMap: `iters` such calls
 * back to back after `warmup` untimed ones, wall clock around them with the stream drained on both sides: milliseconds per call. */
AVIFHIP_API avifResult avifhipRGBImageComputeGainMapAsync(const avifRGBImage * baseRgbImage,
                                                          avifColorPrimaries baseColorPrimaries,
                                                          avifTransferCharacteristics baseTransferCharacteristics,
                                                          const avifRGBImage * altRgbImage,
                                                          avifColorPrimaries altColorPrimaries,
                                                          avifTransferCharacteristics altTransferCharacteristics,
                                                          avifGainMap * gainMap,
                                                          avifDiagnostics * diag,
                                                          void * hipStream);
AVIFHIP_API double avifhipTimeRGBImageComputeGainMap(const avifRGBImage * baseRgbImage,
                                                     avifColorPrimaries baseColorPrimaries,
                                                     avifTransferCharacteristics baseTransferCharacteristics,
                                                     const avifRGBImage * altRgbImage,
                                                     avifColorPrimaries altColorPrimaries,
                                                     avifTransferCharacteristics altTransferCharacteristics,
                                                     avifGainMap * gainMap,
                                                     int warmup,
                                                     int iters,
                                                     void * hipStream);
/* avifImageComputeGainMap (reference src/gainmap.c:843-912): both renditions as YUV images */
AVIFHIP_API avifResult avifhipImageComputeGainMap(const avifImage * baseImage, const avifImage * altImage, avifGainMap * gainMap, avifDiagnostics * diag);
AVIFHIP_API avifResult avifhipImageApplyGainMap(const avifImage * baseImage,
                                                const avifGainMap * gainMap,
                                                float hdrHeadroom,
                                                avifColorPrimaries outputColorPrimaries,
                                                avifTransferCharacteristics outputTransferCharacteristics,
                                                avifRGBImage * toneMappedImage,
                                                avifContentLightLevelInformationBox * clli,
                                                avifDiagnostics * diag);

/* Plane scaling.  Replaces avifImageScale, include/avif/avif.h:922, src/scale.c:23-201, which scales
 * every plane with the vendored libyuv scaler under kFilterBox (third_party/libyuv/source/scale*.c): results are byte-identical
 * (integer arithmetic).  avifhipImageScale works in place on a host-resident image exactly like the reference (new planes
 * are malloc'ed with tight rows, old ones freed if the image owned them).  avifhipImageScaleAsync scales the planes of a
 * device-resident `src` into the caller-allocated planes of `dst` (same depth and format; dst->width / height are the target
 * size; every plane present in src must be present in dst), enqueued on `hipStream`. */
AVIFHIP_API avifResult avifhipImageScale(avifImage * image, uint32_t dstWidth, uint32_t dstHeight);
AVIFHIP_API avifResult avifhipImageScaleAsync(const avifImage * src, avifImage * dst, void * hipStream);

/* Sample Transform derived image items.  Replaces avifImageApplyOperations, include/avif/internal.h:
 * 247-254, src/sampletransform.c:284-421: the postfix expression `tokens` is evaluated for every sample of the selected
 * planes in saturating 32-bit arithmetic (sum, difference, product, quotient, and / or / xor, pow, min, max, negation,
 * absolute value, not, bit-scan-reverse; src/sampletransform.c:199-277), the result clamped to the destination depth and
 * stored; dstImage may be one of the inputs.  All images device-resident, same plane sizes (AVIF_RESULT_BMFF_PARSE_FAILED
 * otherwise, like the reference); only AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_32 is implemented (NOT_IMPLEMENTED otherwise, like the
 * reference); an invalid expression answers AVIF_RESULT_INTERNAL_ERROR like the reference's release build; at most 64 tokens. */
AVIFHIP_API avifResult avifhipImageApplyOperationsAsync(avifImage * dstImage,
                                                        avifSampleTransformBitDepth bitDepth,
                                                        uint32_t numTokens,
                                                        const avifSampleTransformToken * tokens,
                                                        uint8_t numInputImageItems,
                                                        const avifImage * const * inputImageItems,
                                                        avifPlanesFlags planes,
                                                        void * hipStream);

/* ---- integer helpers (src/reformat.c:1778-1840; used on decoded alpha planes, src/read.c:6724) */
AVIFHIP_API int avifhipLimitedToFullY(uint32_t depth, int v);
AVIFHIP_API int avifhipLimitedToFullUV(uint32_t depth, int v);
AVIFHIP_API int avifhipFullToLimitedY(uint32_t depth, int v);
AVIFHIP_API int avifhipFullToLimitedUV(uint32_t depth, int v);

/* kr, kg, kb of an image's CICP (matrixCoefficients / colorPrimaries): replaces avifCalcYUVCoefficients,
 * include/avif/internal.h (src/colr.c:156-189). Host-only, needs no GPU. */
AVIFHIP_API void avifhipCalcYUVCoefficients(const avifImage * image, float * outR, float * outG, float * outB);

/* ---- library control ---------------------------------------------------------------------- */

typedef enum avifhipArithmetic
{
    AVIFHIP_ARITHMETIC_AUTO = 0,   /* what a libavif built with libyuv computes (default) */
    AVIFHIP_ARITHMETIC_FLOAT = 1,  /* what a libavif built without libyuv computes: fp32 everywhere */
    AVIFHIP_ARITHMETIC_LIBYUV = 2  /* AUTO, ignoring rgb->avoidLibYUV */
} avifhipArithmetic;

AVIFHIP_API void avifhipSetArithmetic(avifhipArithmetic mode);
AVIFHIP_API avifhipArithmetic avifhipGetArithmetic(void);

/* Diagnostics/tests: 0 routes every conversion through the universal one-lane-per-pixel kernels,
 * 1 (default) lets the bandwidth-tuned tiled kernels take the configurations they cover. */
AVIFHIP_API void avifhipSetTiledKernels(int enabled);

/* ---- the decode-side tail in one step (device-resident, asynchronous) ------------------------ */

/* YUV -> RGB with the application's transforms fused in: `rgb` receives what avifApplyTransforms (apps/shared/avifutil.c:787-825:
 * clean-aperture crop, avifRGBImageRotate by `angle` quarter turns anti-clockwise, avifRGBImageMirror about `axis`) makes of the
 * converted canvas -- byte for byte avifImageYUVToRGB followed by avifhipRGBImageTransformAsync, without the canvas-sized RGB
 * image in between.  `rgb` has the TRANSFORMED size (crop size, swapped for quarter turns) and carries the conversion
 * parameters (format, depth, chromaUpsampling, avoidLibYUV ...).  One launch where the tiled kernels store through the pixel
 * map (the integer path's 8-bit kernels); two passes through per-thread scratch elsewhere. */
AVIFHIP_API avifResult avifhipImageYUVToRGBTransformedAsync(const avifImage * image, avifRGBImage * rgb, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                                            avifBool mirror, uint8_t axis, void * hipStream);
/* The same for a grid of separately stored tiles (avifhipGridYUVToRGBAsync): tile -> canvas, limited -> full alpha, YUV -> RGB and
 * the transforms, straight from the tiles into the final image. */
AVIFHIP_API avifResult avifhipGridYUVToRGBTransformedAsync(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                                           avifBool alphaIsLimitedRange, avifRGBImage * rgb, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                                           avifBool mirror, uint8_t axis, void * hipStream);

/* ---- row packing for the file writers next to the path (device-resident, asynchronous) -------- */

/* The payload of a Y4M frame as y4mWrite emits it (apps/shared/y4m.c:603-618): planes Y, U, V (and A when `withAlpha`: 8-bit 4:4:4
 * only, like the reference), every row cut to its width, 16-bit samples little-endian as stored.  `frame`: device memory of
 * avifhipY4MFrameBytes(image, withAlpha) bytes -- 0 for every combination the packer refuses (depths other than 8 / 10 / 12, alpha
 * outside 8-bit 4:4:4): the two entry points agree on what a frame is.  The header line is the application's. */
AVIFHIP_API size_t avifhipY4MFrameBytes(const avifImage * image, avifBool withAlpha);
AVIFHIP_API avifResult avifhipImagePackY4MFrameAsync(const avifImage * image, avifBool withAlpha, uint8_t * frame, void * hipStream);
/* The row data avifPNGWrite hands to libpng (apps/shared/avifpng.c:865-880) in the byte order of the PNG stream: pixel rows
 * without padding (width * pixel bytes each), 16-bit samples swapped to big-endian -- png_set_swap done on the device, so the
 * application calls png_write_image on these rows WITHOUT png_set_swap.  `rows`: device memory of height * width * pixel bytes. */
AVIFHIP_API avifResult avifhipRGBImagePackPNGRowsAsync(const avifRGBImage * rgb, uint8_t * rows, void * hipStream);

/* Diagnostics/A-B measurements: bit mask of result-preserving performance knobs (plan.h TuningBits:
 * bit0 XCD-banded tile order, bit1 non-temporal RGB stores). Default: bit0. */
AVIFHIP_API void avifhipSetTuning(uint32_t bits);

/* Selects the HIP device used by the calling thread's context (default: current device). */
AVIFHIP_API avifResult avifhipSetDevice(int device);

/* ---- every GPU of the node behind the same calls --------------------------------------------------------------------
 * The device set: with two or more entries, the synchronous HOST-resident entry points -- avifhipImageYUVToRGB (and the hooks built on it:
 * ...ColorOnly, ...Hook), avifhipImageRGBToYUV, avifhipRGBImagePremultiplyAlpha / UnpremultiplyAlpha, avifhipRGBImageToF16 and
 * avifhipImageYUVToRGBRects -- cut an image of 4 megapixels or more into contiguous row shares (multiples of 32 rows, ~2 megapixels at
 * least; rectangles: contiguous blocks of the row-major job list, i.e. whole tile rows of a grid) and convert each share on its own
 * device: one persistent worker thread, pooled context, streams and staging per entry, every device moving its share over its own host
 * link.  This is the reference's own row-band fan-out (src/reformat.c:1695-1747) across devices instead of host threads.  Nothing is
 * exchanged between devices: the chroma filter's sample row above and below a share is uploaded with it.  Results are byte-identical to
 * the single-device call (every share is the same rectangle of the whole-image conversion).  An unmodified libavif over seam A / seam B
 * then uses every GPU of the node: AVIFHIP_DEVICES="all" or "0,1,2,3" in the environment selects the set without a code change
 * (avifhipSetDeviceSet overrides it).  A device may be named more than once ("0,0": two workers on one GPU -- how a one-GPU box
 * exercises the path).  count = 0 (the default): calls run on the calling thread's device as ever.  Device-resident (Async) entry
 * points are not affected: their buffers live on one device. */
AVIFHIP_API avifResult avifhipSetDeviceSet(const int * devices, uint32_t count);
/* The smallest share that is worth a device of its own, in pixels (0 = the default, 2^21: below ~2 megapixels a second device costs more
 * than its host link brings).  Tests lower it to send small images through the farm; shares stay multiples of 32 rows.
 * AVIFHIP_FARM_MIN_PIXELS=<n> in the environment (read with AVIFHIP_DEVICES) is the same knob for a process that cannot call this. */
AVIFHIP_API void avifhipSetFarmMinSharePixels(uint64_t pixels);
/* The current set: writes at most `capacity` entries, returns the set's size. */
AVIFHIP_API uint32_t avifhipGetDeviceSet(int * devices, uint32_t capacity);
/* Host-only (needs no GPU): the row shares a `workers`-entry set gives an image of this size -- bands[k] = { 0, first row, width, rows };
 * *count = 1 means "not farmed".  bands may be NULL (count only); capacity >= workers always suffices. */
AVIFHIP_API avifResult avifhipPlanFarmRows(uint32_t width, uint32_t height, uint32_t workers, avifCropRect * bands, uint32_t capacity, uint32_t * count);
/* The calling thread's last host-resident call: how many workers took part (0: it ran on the thread's own device), and for worker k its
 * device, its share (rows [begin, end) of the image; avifhipImageYUVToRGBRects: entries of the coalesced job list) and the bytes it moved
 * over its host link.  avifhipLastTransferBytes reports the sum. */
AVIFHIP_API uint32_t avifhipLastFarmWorkers(void);
AVIFHIP_API avifResult avifhipLastFarmTransferBytes(uint32_t worker, int * device, uint32_t * begin, uint32_t * end, uint64_t * bytesUp, uint64_t * bytesDown);
/* Number of visible HIP devices; 0 when no GPU / no driver. */
AVIFHIP_API int avifhipDeviceCount(void);
/* Extra HIP streams (returned as void*, usable as the `hipStream` argument of the Async entry points) so that
 * independent frames / tiles can overlap each other's head and tail; NULL on failure. */
AVIFHIP_API void * avifhipStreamCreate(void);
AVIFHIP_API void avifhipStreamDestroy(void * hipStream);
/* Blocks until the calling thread's library stream (or `hipStream`) is idle. */
AVIFHIP_API avifResult avifhipSynchronize(void * hipStream);
/* Text of the last HIP/runtime failure on this thread ("" if none). */
AVIFHIP_API const char * avifhipLastError(void);
/* Which kernel family served the last conversion on this thread (diagnostics/tests):
 * e.g. "yuv2rgb_tile<u8,420,bilinear,rgba8>" or "yuv2rgb_generic". */
AVIFHIP_API const char * avifhipLastKernel(void);
/* Host-only (needs no GPU): how the library would serve avifhipImageYUVToRGB(image, rgb) under the current arithmetic
 * setting, as text "arith=<fp32|libyuv> kernel=<tile|generic> native=<8|10|12> downshift=<n> bilinear=<0|1>
 * alpha=<keep|fill|plane-shift|plane-float> inloopmul=<n> postmul=<n> postmulfx=<0|1>" -- the plan layer's restatement of
 * src/reformat.c:1445-1593 and src/reformat_libyuv.c:544-1108, exposed for tests and diagnostics.  Returns the avifResult
 * the conversion would start with (argument / format errors), writes at most `size` bytes including the terminator. */
AVIFHIP_API avifResult avifhipExplainYUVToRGB(const avifImage * image, const avifRGBImage * rgb, char * text, size_t size);
/* Same for avifhipImageRGBToYUV: "arith=<fp32|libyuv> kernel=<tile|generic> mul=<n>". */
AVIFHIP_API avifResult avifhipExplainRGBToYUV(const avifImage * image, const avifRGBImage * rgb, char * text, size_t size);
/* Number of conversions this thread has enqueued on a GPU so far (tests: proves the HIP path, not a fallback, ran). */
AVIFHIP_API uint64_t avifhipLaunchCount(void);
/* Number of batch / grid descriptor tables this thread has sent to the device.  A batch or grid call whose buffers, geometry and settings
 * equal the previous one's (a decoder converting into the same tile buffers frame after frame) launches on the table the device still
 * holds and does not count (tests). */
AVIFHIP_API uint64_t avifhipTableUploadCount(void);
AVIFHIP_API const char * avifhipVersion(void);

/* Plain device-memory helpers so C callers (and the ctypes tests) need no HIP headers. */
AVIFHIP_API void * avifhipDeviceAlloc(size_t bytes);
AVIFHIP_API void avifhipDeviceFree(void * devicePtr);
AVIFHIP_API avifResult avifhipCopyToDevice(void * devicePtr, const void * hostPtr, size_t bytes);
AVIFHIP_API avifResult avifhipCopyToHost(void * hostPtr, const void * devicePtr, size_t bytes);
AVIFHIP_API avifResult avifhipDeviceMemset(void * devicePtr, int value, size_t bytes);

/* Kernel timing with HIP events on the stream the kernels are launched on (bench.py):
 * returns the average milliseconds per call of `iters` back-to-back
 * avifhipImageYUVToRGBAsync launches after `warmup` untimed ones, or a negative value on error. */
AVIFHIP_API double avifhipTimeYUVToRGB(const avifImage * image, avifRGBImage * rgb, int warmup, int iters, void * hipStream);
AVIFHIP_API double avifhipTimeRGBToYUV(avifImage * image, const avifRGBImage * rgb, int warmup, int iters, void * hipStream);
/* Same for launches that cycle over `count` distinct device-resident frames (launch k converts frame k % count), so
 * that a working set larger than the Infinity Cache makes every launch stream from and to HBM. */
AVIFHIP_API double avifhipTimeYUVToRGBCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream);
/* The chip's own ceiling for the byte movement of a conversion: the same timing for a kernel that reads every plane sample the
 * conversion reads once and writes every output byte once with NO arithmetic (kernels_bench.hip; the destination buffers receive
 * meaningless bytes), in the tiled kernels' own access shapes (4 samples per lane and plane row, 16 bytes of pixels per lane at
 * consecutive addresses, streaming stores).  Any plane layout (8-bit or 16-bit containers, 4:4:4 / 4:2:2 / 4:2:0 / 4:0:0, with or without
 * an alpha plane the conversion reads) into 4- or 8-byte pixels; all jobs of a call share one layout.  The fastest of the tile shapes /
 * orders / load policies the kernel knows (1024 x 2, 1024 x 4, 512 x 4 and 256 x 8 pixels in raster order, 256 x 16 and 256 x 32 in per-XCD bands;
 * plain and streaming loads) is reported;
 * avifhipLastKernel() then names it.  Negative for anything else (avifhipLastError() says why).
 *   avifhipTimeStreamCeiling          launch k moves job k % count (frames cycled, like avifhipTimeYUVToRGBCycle)
 *   avifhipTimeStreamCeilingRGBToYUV  the encode direction: pixels read, planes written (avifhipTimeRGBToYUVCycle)
 *   avifhipTimeStreamCeilingBatch     all `count` jobs in ONE launch (avifhipTimeYUVToRGBBatch; a grid: rgbs[k] = tile k's rectangle
 *                                     of the canvas) */
AVIFHIP_API double avifhipTimeStreamCeiling(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream);
AVIFHIP_API double avifhipTimeStreamCeilingRGBToYUV(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream);
/* ... of a plane scaling job (avifhipImageScaleAsync): every sample of `src`'s planes read once, every sample of `dst`'s planes written once, 16 bytes
 * per lane, nothing computed.  Planes and pitches must be multiples of 16 bytes. */
AVIFHIP_API double avifhipTimeStreamCeilingScale(const avifImage * src, avifImage * dst, int warmup, int iters, void * hipStream);
AVIFHIP_API double avifhipTimeStreamCeilingBatch(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream);
/* The same event timing for the encode direction over `count` cycled frames, for a batch call (avifhipImageYUVToRGBBatchAsync: milliseconds
 * per batch) and for a grid call (avifhipGridYUVToRGBAsync: milliseconds per canvas, every kernel the call launches included). */
AVIFHIP_API double avifhipTimeRGBToYUVCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream);
AVIFHIP_API double avifhipTimeRGBToYUVBatchCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup, int iters,
                                                 void * hipStream);
AVIFHIP_API double avifhipTimeStreamCeilingRGBToYUVBatchCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                              int iters, void * hipStream);
AVIFHIP_API double avifhipTimeYUVToRGBBatch(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, const avifCropRect * rects,
                                            int warmup, int iters, void * hipStream);
/* ... and for a sequence walked `perLaunch` frames at a time: launch k converts frames (k * perLaunch + j) % count, j < perLaunch, in ONE
 * launch (avifhipImageYUVToRGBBatchAsync, whole images).  Milliseconds per launch. */
AVIFHIP_API double avifhipTimeYUVToRGBBatchCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                 int iters, void * hipStream);
/* (its byte-movement ceiling: `count` must be a multiple of `perLaunch`) */
AVIFHIP_API double avifhipTimeStreamCeilingBatchCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                      int iters, void * hipStream);
AVIFHIP_API double avifhipTimeGridYUVToRGB(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                           avifBool alphaIsLimitedRange, avifRGBImage * rgbCanvas, int warmup, int iters, void * hipStream);

/* Synthetic planes for benchmarks/tests: xorshift32 stream
 * (x^=x<<13; x^=x>>17; x^=x<<5), one draw per sample, value = lo + draw % (hi-lo+1), written
 * as uint8 (bytesPerSample 1) or little-endian uint16 (2) rows. Returns the advanced state. */
AVIFHIP_API uint32_t avifhipSynthFill(uint32_t state,
                                      uint8_t * plane,
                                      uint32_t rowBytes,
                                      uint32_t width,
                                      uint32_t height,
                                      uint32_t bytesPerSample,
                                      uint32_t lo,
                                      uint32_t hi);

#ifdef __cplusplus
}
#endif
#endif /* AVIFHIP_H */
