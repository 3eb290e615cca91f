/*
 * libyuv_oracle.c -- TEST INFRASTRUCTURE (the checker), NOT PRODUCT CODE.
 * Fixed-point ("libyuv") arithmetic restated from SURVEY.md Appendix D.
 * Placeholder until the integer path lands: every entry declines.
 */
#include "reformat_oracle.h"

avifResult oracleLibyuvImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    (void)image;
    (void)rgb;
    return AVIF_RESULT_NOT_IMPLEMENTED;
}
avifResult oracleLibyuvImageRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    (void)image;
    (void)rgb;
    return AVIF_RESULT_NOT_IMPLEMENTED;
}
avifResult oracleLibyuvRGBImagePremultiplyAlpha(avifRGBImage * rgb)
{
    (void)rgb;
    return AVIF_RESULT_NOT_IMPLEMENTED;
}
avifResult oracleLibyuvRGBImageUnpremultiplyAlpha(avifRGBImage * rgb)
{
    (void)rgb;
    return AVIF_RESULT_NOT_IMPLEMENTED;
}
