/*
 * oracle_backend.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference composes its reformat calls from an optional accelerated backend (libyuv:
 * /root/reference/include/avif/internal.h:349-386) and its own fp32 code (src/reformat.c:1445-1593,
 * :264-272, src/alpha.c:163-166,350-353).  The restatement keeps that seam: reformat_oracle.c asks an
 * OracleBackend exactly where the reference asks libyuv, libyuv_oracle.c provides the backend that
 * restates libyuv's fixed-point arithmetic.  backend == NULL is a libavif built without libyuv.
 */
#ifndef AVIFHIP_ORACLE_BACKEND_H
#define AVIFHIP_ORACLE_BACKEND_H

#include "avifhip/avif_abi.h"

typedef struct OracleBackend
{
    /* avifImageYUVToRGBLibYUV, internal.h:349-363 */
    avifResult (*yuvToRgb)(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, avifBool * alphaReformatted);
    /* avifImageRGBToYUVLibYUV, internal.h:346 */
    avifResult (*rgbToYuv)(avifImage * image, const avifRGBImage * rgb);
    /* avifRGBImagePremultiplyAlphaLibYUV / avifRGBImageUnpremultiplyAlphaLibYUV, internal.h:369-378 */
    avifResult (*premultiply)(avifRGBImage * rgb);
    avifResult (*unpremultiply)(avifRGBImage * rgb);
} OracleBackend;

avifResult oracleImageYUVToRGBWithBackend(const avifImage * image, avifRGBImage * rgb, const OracleBackend * backend);
avifResult oracleImageRGBToYUVWithBackend(avifImage * image, const avifRGBImage * rgb, const OracleBackend * backend);
avifResult oracleAlphaPassWithBackend(avifRGBImage * rgb, int unmultiply, const OracleBackend * backend);

#endif
