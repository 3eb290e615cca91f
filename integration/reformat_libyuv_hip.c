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
#include <string.h>

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

/* Folding libavif's follow-up steps into the colour hook (below) leans on two properties of libavif's own code: which steps
 * avifImageYUVToRGBImpl issues after its colour hook, and that it issues them right away on the same thread (src/reformat.c:1574-1590,
 * :1649-1678; src/alpha.c:151-166).  They were checked for libavif 1.4.x; in a libavif of another major.minor this file still works as a
 * backend, but the colour hook does only its own job (avifhipImageYUVToRGBColorOnly) and every follow-up call is a real pass.
 * AVIFHIP_FOLD=0 / 1 in the environment overrides (read at every call: cheap, and tests switch it). */
#if defined(AVIF_VERSION_MAJOR) && defined(AVIF_VERSION_MINOR) && AVIF_VERSION_MAJOR == 1 && AVIF_VERSION_MINOR == 4
#define AVIFHIP_FOLD_VALIDATED 1
#else
#define AVIFHIP_FOLD_VALIDATED 0
#endif
static avifBool avifHipFoldEnabled(void)
{
    const char * e = getenv("AVIFHIP_FOLD");
    if (e && *e)
        return atoi(e) != 0;
    /* (the headers this file was compiled against AND the library it runs in: a distribution may swap the shared object) */
    return AVIFHIP_FOLD_VALIDATED && strncmp(avifVersion(), "1.4.", 4) == 0;
}
/* One-shot note from the colour hook to the hooks libavif calls right after it on the same thread for the same pixels
 * (src/reformat.c:1574-1590: avifRGBImagePremultiplyAlpha / UnpremultiplyAlpha, then avifRGBImageToF16): avifhipImageYUVToRGBHook has
 * already produced the FINAL pixels, so those calls are answered without moving the image across the bus again.  A step is skipped only
 * if it is the very next hook call of this thread, for the same buffer and geometry, and a sample of the pixels still reads as the colour
 * hook left it; every other hook call drops the note.  (Between the two calls there is only libavif's own code: no application code runs
 * inside avifImageYUVToRGB, and a later, separate avifRGBImagePremultiplyAlpha of the application finds no note: libavif consumed it.
 * Should a libavif ever NOT issue the follow-up -- the version gate above is there so that this cannot happen silently -- the note dies with
 * the thread's next hook call.)
 * No clock takes part (until round 6 a note also expired after a second): the follow-up of a stalled thread -- SIGSTOP, a paused VM, swap --
 * arrives late but is still the follow-up, and answering it with a real pass would multiply pixels that are final already.  How long ago the
 * colour hook ran says nothing about whose call this is; the thread, the buffer, the geometry and the pixels do. */
typedef struct avifHipFoldNote
{
    const uint8_t * pixels;
    uint32_t width, height, rowBytes, depth;
    avifRGBFormat format;
    uint32_t steps; /* AVIFHIP_FOLDED_* still to be answered */
    uint64_t sample;
} avifHipFoldNote;
static _Thread_local avifHipFoldNote avifHipNote;

/* 64 probes of 8 bytes spread over the buffer (first and last row included): a cheap guard against a caller that changes the pixels
 * between two hook calls it makes itself */
static uint64_t avifHipSamplePixels(const avifRGBImage * rgb)
{
    const uint32_t pixelBytes = avifRGBImagePixelSize(rgb);
    const uint64_t rowPayload = (uint64_t)rgb->width * pixelBytes;
    uint64_t h = 0x9e3779b97f4a7c15ull;
    if (!rgb->pixels || rowPayload < 8 || !rgb->height)
        return h;
    for (uint32_t k = 0; k < 64; ++k) {
        const uint32_t row = (uint32_t)(((uint64_t)k * (rgb->height - 1)) / 63);
        const uint64_t col = ((uint64_t)k * 0x9e3779b1u) % (rowPayload - 7);
        uint64_t v;
        memcpy(&v, rgb->pixels + (size_t)row * rgb->rowBytes + col, 8);
        h = (h ^ v) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    return h;
}

/* AVIF_TRUE: `step` was folded into the colour hook's result for exactly these pixels -- consume it */
static avifBool avifHipTakeFoldedStep(const avifRGBImage * rgb, uint32_t step)
{
    avifHipFoldNote * n = &avifHipNote;
    const avifBool match = (n->steps & step) && n->pixels == rgb->pixels && n->width == rgb->width && n->height == rgb->height &&
                           n->rowBytes == rgb->rowBytes && n->depth == rgb->depth && n->format == rgb->format &&
                           n->sample == avifHipSamplePixels(rgb);
    if (!match) {
        n->steps = 0;
        return AVIF_FALSE;
    }
    n->steps &= ~step; /* (a remaining step sees the same pixels: nothing was written) */
    return AVIF_TRUE;
}

avifResult avifImageRGBToYUVLibYUV(avifImage * image, const avifRGBImage * rgb)
{
    avifHipNote.steps = 0;
    /* called only without alpha (un)multiply and for non-gray sources (src/reformat.c:255,265); planes are allocated */
    if (!avifHipWorthIt(image->width, image->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipImageRGBToYUV(image, rgb));
}

avifResult avifImageYUVToRGBLibYUV(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, avifBool * alphaReformattedWithLibYUV)
{
    avifHipNote.steps = 0;
    *alphaReformattedWithLibYUV = AVIF_FALSE; /* must be valid for OK and NOT_IMPLEMENTED (internal.h:361-363) */
    if (!avifHipWorthIt(image->width, image->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    uint32_t folded = 0;
    const avifResult r = avifHipOrFallback(avifHipFoldEnabled() ? avifhipImageYUVToRGBHook(image, rgb, reformatAlpha, &folded)
                                                                : avifhipImageYUVToRGBColorOnly(image, rgb, reformatAlpha));
    if (r == AVIF_RESULT_OK && reformatAlpha)
        *alphaReformattedWithLibYUV = AVIF_TRUE; /* copied / rescaled from the alpha plane, or opaque fill */
    if (r == AVIF_RESULT_OK && folded) {
        avifHipNote.pixels = rgb->pixels, avifHipNote.width = rgb->width, avifHipNote.height = rgb->height, avifHipNote.rowBytes = rgb->rowBytes;
        avifHipNote.depth = rgb->depth, avifHipNote.format = rgb->format;
        avifHipNote.sample = avifHipSamplePixels(rgb);
        avifHipNote.steps = folded;
    }
    return r;
}

avifResult avifRGBImagePremultiplyAlphaLibYUV(avifRGBImage * rgb)
{
    if (avifHipTakeFoldedStep(rgb, AVIFHIP_FOLDED_PREMULTIPLY))
        return AVIF_RESULT_OK;
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImagePremultiplyAlpha(rgb));
}

avifResult avifRGBImageUnpremultiplyAlphaLibYUV(avifRGBImage * rgb)
{
    if (avifHipTakeFoldedStep(rgb, AVIFHIP_FOLDED_UNPREMULTIPLY))
        return AVIF_RESULT_OK;
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImageUnpremultiplyAlpha(rgb));
}

avifResult avifRGBImageToF16LibYUV(avifRGBImage * rgb)
{
    if (avifHipTakeFoldedStep(rgb, AVIFHIP_FOLDED_TO_F16))
        return AVIF_RESULT_OK;
    if (!avifHipWorthIt(rgb->width, rgb->height))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    return avifHipOrFallback(avifhipRGBImageToF16(rgb));
}

/* printed by `avifenc --version` (apps/shared/avifutil.c:216) next to "libyuv"; 0 would read "not available" */
unsigned int avifLibYUVVersion(void)
{
    return 9500; /* gfx950 backend marker; libyuv's own versions are < 2000 */
}
