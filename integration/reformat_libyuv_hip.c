/*
 * reformat_libyuv_hip.c -- drop-in replacement for libavif's src/reformat_libyuv.c.
 *
 * libavif already has a seam for an accelerated reformat backend: the six "LibYUV" hook functions declared in
 * include/avif/internal.h:349-386, whose "backend absent" stubs live in src/reformat_libyuv.c:6-41 and which
 * src/reformat.c:266,1423,1454 and src/alpha.c:163,350 call before their built-in CPU code.  This file defines
 * those hooks on top of libavifhip.so (the MI355X HIP kernels, include/avifhip.h).  Build libavif with this file
 * in place of src/reformat_libyuv.c and link libavifhip.so; avifenc / avifdec / every libavif consumer then runs
 * unchanged.
 *
 * Contract kept (include/avif/internal.h:349-386):
 *   AVIF_RESULT_OK               the hook did the work;
 *   AVIF_RESULT_NOT_IMPLEMENTED  libavif silently runs its built-in CPU code -- returned for everything that is not
 *                                worth a GPU round trip (small images), for a missing GPU, and for any HIP failure;
 *   anything else                is propagated to the caller: used only for the argument errors libavif itself
 *                                would report.
 * Arithmetic: by default (AVIFHIP_ARITHMETIC=auto) every result is byte-identical to a stock libavif built with
 * libyuv: the hooks compute libyuv's fixed-point arithmetic for the combinations src/reformat_libyuv.c hands to
 * libyuv, libavif's built-in fp32 arithmetic for the combinations libyuv declines (instead of declining too), and
 * decline the one case a hook cannot reproduce (a pending alpha multiply inside the built-in slow loop).  With
 * AVIFHIP_ARITHMETIC=float every hook computes the built-in fp32 arithmetic, i.e. a libavif built without libyuv.
 */
#include "avif/internal.h"

#include "avifhip.h" /* sees AVIF_AVIF_H: uses libavif's own struct definitions */

#include <stdlib.h>

/* Below this many pixels a conversion costs less on the CPU than the PCIe round trip (env override for tests). */
static uint32_t avifHipMinPixels(void)
{
    static int cached = -1;
    if (cached < 0) {
        const char * e = getenv("AVIFHIP_MIN_PIXELS");
        cached = e ? atoi(e) : 512 * 512;
        if (cached < 0)
            cached = 0;
    }
    return (uint32_t)cached;
}

static avifBool avifHipWorthIt(uint32_t width, uint32_t height)
{
    return ((uint64_t)width * height >= avifHipMinPixels()) && (avifhipDeviceCount() > 0);
}

/* HIP/runtime failures must not fail the user's call: the CPU path is still there. */
static avifResult avifHipOrFallback(avifResult r)
{
    return (r == AVIF_RESULT_UNKNOWN_ERROR || r == AVIF_RESULT_OUT_OF_MEMORY) ? AVIF_RESULT_NOT_IMPLEMENTED : r;
}

avifResult avifImageRGBToYUVLibYUV(avifImage * image, const avifRGBImage * rgb)
{
    /* called only without alpha (un)multiply and for non-gray sources (src/reformat.c:255,265); planes are allocated */
    if (!avifHipWorthIt(image->width, image->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipImageRGBToYUV(image, rgb));
}

avifResult avifImageYUVToRGBLibYUV(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, avifBool * alphaReformattedWithLibYUV)
{
    *alphaReformattedWithLibYUV = AVIF_FALSE; /* must be valid for OK and NOT_IMPLEMENTED (internal.h:361-363) */
    if (!avifHipWorthIt(image->width, image->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    const avifResult r = avifHipOrFallback(avifhipImageYUVToRGBColorOnly(image, rgb, reformatAlpha));
    if (r == AVIF_RESULT_OK && reformatAlpha)
        *alphaReformattedWithLibYUV = AVIF_TRUE; /* copied / rescaled from the alpha plane, or opaque fill */
    return r;
}

avifResult avifRGBImagePremultiplyAlphaLibYUV(avifRGBImage * rgb)
{
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImagePremultiplyAlpha(rgb));
}

avifResult avifRGBImageUnpremultiplyAlphaLibYUV(avifRGBImage * rgb)
{
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImageUnpremultiplyAlpha(rgb));
}

avifResult avifRGBImageToF16LibYUV(avifRGBImage * rgb)
{
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImageToF16(rgb));
}

/* printed by `avifenc --version` (apps/shared/avifutil.c:216) next to "libyuv"; 0 would read "not available" */
unsigned int avifLibYUVVersion(void)
{
    return 9500; /* gfx950 backend marker; libyuv's own versions are < 2000 */
}
