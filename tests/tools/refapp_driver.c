/* refapp_driver.c -- test tool (not shipped).  A thin main() over the REFERENCE's own application code (apps/shared/y4m.c,
 * avifpng.c, avifutil.c -- compiled from where they lie under /root/reference by oracle/Makefile, never copied), linked twice:
 * against the reference built without a backend (oracle/_ref/libavif_ref.so) and against the reference built on top of the HIP
 * hooks (oracle/_ref/libavif_hipbackend.so).  What avifdec / avifenc do around the reformat path, without a codec:
 *   refapp y4m2png <in.y4m> <out prefix> <png depth: 8|16> [upsampling]
 *        every frame:  y4mRead (apps/shared/y4m.c:256) -> avifPNGWrite (apps/shared/avifpng.c:627: avifImageYUVToRGB at :688,
 *        avifApplyTransforms, libpng) into <out prefix>_<frame>.png
 *   refapp png2y4m <in.png> <out.y4m> <yuv format: 444|422|420|400> <yuv depth> [matrix coefficients] [range: full|limited]
 *        avifReadImage (apps/shared/avifutil.c:318: avifPNGRead -> avifImageRGBToYUV) -> y4mWrite (apps/shared/y4m.c:481)
 * Exit status 0 on success.  tests/test_gpu_refapps.py compares the files the two builds write. */
#include "avif/avif.h"

#include "avifpng.h"
#include "avifutil.h"
#include "y4m.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int y4m2png(const char * in, const char * prefix, int depth, avifChromaUpsampling upsampling)
{
    struct y4mFrameIterator * iter = NULL;
    int frame = 0;
    for (;;) {
        avifImage * image = avifImageCreateEmpty();
        if (!image)
            return 1;
        avifAppSourceTiming timing;
        if (!y4mRead(in, /*ignoreAlpha=*/AVIF_FALSE, /*imageSizeLimit=*/AVIF_DEFAULT_IMAGE_SIZE_LIMIT, image, &timing, &iter)) {
            avifImageDestroy(image);
            return 1;
        }
        char name[1024];
        snprintf(name, sizeof(name), "%s_%d.png", prefix, frame);
        const avifBool ok = avifPNGWrite(name, image, (uint32_t)depth, upsampling, /*compressionLevel=*/1);
        avifImageDestroy(image);
        if (!ok)
            return 1;
        ++frame;
        if (!iter)
            break; /* y4mRead closes the stream and clears the iterator after the last frame */
    }
    printf("frames=%d\n", frame);
    return 0;
}

static int png2y4m(const char * in, const char * out, const char * fmt, int depth, int mc, const char * range)
{
    avifPixelFormat format = AVIF_PIXEL_FORMAT_YUV444;
    if (!strcmp(fmt, "422"))
        format = AVIF_PIXEL_FORMAT_YUV422;
    else if (!strcmp(fmt, "420"))
        format = AVIF_PIXEL_FORMAT_YUV420;
    else if (!strcmp(fmt, "400"))
        format = AVIF_PIXEL_FORMAT_YUV400;
    avifImage * image = avifImageCreateEmpty();
    if (!image)
        return 1;
    /* what avifenc sets from its command line before it reads the input (apps/avifenc.c: --cicp, --range) */
    image->matrixCoefficients = (avifMatrixCoefficients)mc;
    image->yuvRange = !strcmp(range, "limited") ? AVIF_RANGE_LIMITED : AVIF_RANGE_FULL;
    uint32_t outDepth = 0;
    const avifAppFileFormat got = avifReadImage(in, AVIF_APP_FILE_FORMAT_UNKNOWN, format, depth, AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC,
                                                /*ignoreColorProfile=*/AVIF_TRUE, /*ignoreExif=*/AVIF_TRUE, /*ignoreXMP=*/AVIF_TRUE, /*ignoreAlpha=*/AVIF_FALSE,
                                                /*ignoreGainMap=*/AVIF_TRUE, AVIF_DEFAULT_IMAGE_SIZE_LIMIT, image, &outDepth, NULL, NULL);
    if (got != AVIF_APP_FILE_FORMAT_PNG) {
        avifImageDestroy(image);
        return 1;
    }
    const avifBool ok = y4mWrite(out, image);
    printf("png depth=%u yuv depth=%u alpha=%d\n", outDepth, image->depth, image->alphaPlane ? 1 : 0);
    avifImageDestroy(image);
    return ok ? 0 : 1;
}

int main(int argc, char ** argv)
{
    if (argc >= 5 && !strcmp(argv[1], "y4m2png")) {
        avifChromaUpsampling up = AVIF_CHROMA_UPSAMPLING_AUTOMATIC;
        if (argc >= 6)
            up = (avifChromaUpsampling)atoi(argv[5]);
        return y4m2png(argv[2], argv[3], atoi(argv[4]), up);
    }
    if (argc >= 6 && !strcmp(argv[1], "png2y4m"))
        return png2y4m(argv[2], argv[3], argv[4], atoi(argv[5]), argc >= 7 ? atoi(argv[6]) : AVIF_MATRIX_COEFFICIENTS_BT601, argc >= 8 ? argv[7] : "full");
    fprintf(stderr, "usage: refapp y4m2png <in.y4m> <out prefix> <8|16> [upsampling] | png2y4m <in.png> <out.y4m> <444|422|420|400> <depth> [mc] [full|limited]\n");
    return 2;
}
