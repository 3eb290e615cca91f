/*
 * avif_preload_hip.c -- seam A: an LD_PRELOAD-able interposer for applications that link a SHARED libavif and cannot be
 * rebuilt.  It exports the four public reformat entry points of include/avif/avif.h:1031-1038
 *     avifImageYUVToRGB, avifImageRGBToYUV, avifRGBImagePremultiplyAlpha, avifRGBImageUnpremultiplyAlpha
 * and the gain-map entry points avifRGBImageApplyGainMap (:1736-1745) and avifRGBImageComputeGainMap (:1752-1759) with
 * libavif's own signatures, serves them from libavifhip.so (the MI355X HIP kernels), and forwards to the real
 * libavif (dlsym(RTLD_NEXT)) everything that is not worth a GPU round trip, that the GPU library declines, or that fails
 * on the accelerator -- the application's call never fails because of the interposer.
 *
 *     LD_PRELOAD=/path/libavifhip_preload.so avifdec in.avif out.png
 *
 * Arithmetic follows the libavif being interposed: if it was built with libyuv (avifLibYUVVersion() != 0) results equal
 * that build's (libavifhip's default, AVIFHIP_ARITHMETIC_AUTO); if it was built without, the fp32 arithmetic is pinned.
 * AVIFHIP_ARITHMETIC in the environment overrides.  Only the six symbols above are interposed; calls libavif makes
 * internally (e.g. avifImageYUVToRGB from its decoder helpers) are bound inside libavif and are not affected.
 *
 * The application's avifImage / avifRGBImage / avifGainMap are read through the struct mirror of include/avifhip/avif_abi.h,
 * which follows libavif AVIFHIP_MIRRORED_MAJOR.AVIFHIP_MIRRORED_MINOR.  The interposer therefore asks the interposed library
 * for avifVersion() first: with another major.minor every call is passed straight through (AVIFHIP_PRELOAD_FORCE=1 serves the
 * four reformat entry points anyway -- the fields they read have kept their places through 1.x -- but never the gain-map ones,
 * whose structs changed between the experimental 1.0/1.1 builds and 1.2).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#include "avifhip.h"

#define AVIF_EXPORT __attribute__((visibility("default")))

typedef avifResult (*YuvToRgbFn)(const avifImage *, avifRGBImage *);
typedef avifResult (*RgbToYuvFn)(avifImage *, const avifRGBImage *);
typedef avifResult (*AlphaFn)(avifRGBImage *);
typedef unsigned int (*VersionFn)(void);
typedef avifResult (*GainMapFn)(const avifRGBImage *, avifColorPrimaries, avifTransferCharacteristics, const avifGainMap *, float, avifColorPrimaries,
                                avifTransferCharacteristics, avifRGBImage *, avifContentLightLevelInformationBox *, avifDiagnostics *);

typedef avifResult (*ComputeGainMapFn)(const avifRGBImage *, avifColorPrimaries, avifTransferCharacteristics, const avifRGBImage *, avifColorPrimaries,
                                       avifTransferCharacteristics, avifGainMap *, avifDiagnostics *);

#define AVIFHIP_MIRRORED_MAJOR 1
#define AVIFHIP_MIRRORED_MINOR 4

typedef const char * (*VersionStringFn)(void);

static pthread_once_t gOnce = PTHREAD_ONCE_INIT;
static struct
{
    YuvToRgbFn yuvToRgb;
    RgbToYuvFn rgbToYuv;
    AlphaFn premultiply, unpremultiply;
    GainMapFn applyGainMap;
    ComputeGainMapFn computeGainMap;
    uint64_t minPixels;
    int gpu;      /* serve the reformat entry points from the GPU */
    int gpuGainMap; /* ... and the gain-map ones */
} g;

/* 1 when the interposed libavif reports the mirrored major.minor, 0 when it reports another one or none */
static int versionMatches(void)
{
    const VersionStringFn version = (VersionStringFn)dlsym(RTLD_NEXT, "avifVersion");
    const char * v = version ? version() : NULL;
    if (!v)
        return 0;
    char * end = NULL;
    const long major = strtol(v, &end, 10);
    if (!end || *end != '.')
        return 0;
    const long minor = strtol(end + 1, NULL, 10);
    return major == AVIFHIP_MIRRORED_MAJOR && minor == AVIFHIP_MIRRORED_MINOR;
}

static void resolveOnce(void)
{
    g.yuvToRgb = (YuvToRgbFn)dlsym(RTLD_NEXT, "avifImageYUVToRGB");
    g.rgbToYuv = (RgbToYuvFn)dlsym(RTLD_NEXT, "avifImageRGBToYUV");
    g.premultiply = (AlphaFn)dlsym(RTLD_NEXT, "avifRGBImagePremultiplyAlpha");
    g.unpremultiply = (AlphaFn)dlsym(RTLD_NEXT, "avifRGBImageUnpremultiplyAlpha");
    g.applyGainMap = (GainMapFn)dlsym(RTLD_NEXT, "avifRGBImageApplyGainMap");
    g.computeGainMap = (ComputeGainMapFn)dlsym(RTLD_NEXT, "avifRGBImageComputeGainMap");
    const char * e = getenv("AVIFHIP_MIN_PIXELS");
    const long v = e ? atol(e) : 512L * 512L;
    g.minPixels = v < 0 ? 0 : (uint64_t)v;
    const int sameVersion = versionMatches();
    const char * force = getenv("AVIFHIP_PRELOAD_FORCE");
    const int haveGpu = avifhipDeviceCount() > 0;
    g.gpuGainMap = haveGpu && sameVersion;
    g.gpu = haveGpu && (sameVersion || (force && force[0] == '1'));
    if (!getenv("AVIFHIP_ARITHMETIC")) {
        const VersionFn libyuvVersion = (VersionFn)dlsym(RTLD_NEXT, "avifLibYUVVersion");
        if (libyuvVersion && libyuvVersion() == 0)
            avifhipSetArithmetic(AVIFHIP_ARITHMETIC_FLOAT); /* the interposed libavif has no libyuv */
    }
}

/* every field of g is published by pthread_once before any caller reads it */
static void resolve(void)
{
    (void)pthread_once(&gOnce, resolveOnce);
}

static int worthIt(uint32_t width, uint32_t height)
{
    return g.gpu && (uint64_t)width * height >= g.minPixels;
}
/* results after which the real libavif should take the call */
static int declined(avifResult r)
{
    return r == AVIF_RESULT_NOT_IMPLEMENTED || r == AVIF_RESULT_UNKNOWN_ERROR || r == AVIF_RESULT_OUT_OF_MEMORY;
}

AVIF_EXPORT avifResult avifImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    resolve();
    if (image && rgb && worthIt(image->width, image->height)) {
        const avifResult r = avifhipImageYUVToRGB(image, rgb);
        if (!declined(r) || !g.yuvToRgb)
            return r;
    }
    return g.yuvToRgb ? g.yuvToRgb(image, rgb) : AVIF_RESULT_NOT_IMPLEMENTED;
}

AVIF_EXPORT avifResult avifImageRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    resolve();
    /* libsharpyuv downsampling is libavif's own (src/reformat_libsharpyuv.c): leave those calls alone */
    if (image && rgb && worthIt(image->width, image->height) && rgb->chromaDownsampling != AVIF_CHROMA_DOWNSAMPLING_SHARP_YUV) {
        const avifResult r = avifhipImageRGBToYUV(image, rgb);
        if (!declined(r) || !g.rgbToYuv)
            return r;
    }
    return g.rgbToYuv ? g.rgbToYuv(image, rgb) : AVIF_RESULT_NOT_IMPLEMENTED;
}

AVIF_EXPORT avifResult avifRGBImagePremultiplyAlpha(avifRGBImage * rgb)
{
    resolve();
    if (rgb && worthIt(rgb->width, rgb->height)) {
        const avifResult r = avifhipRGBImagePremultiplyAlpha(rgb);
        if (!declined(r) || !g.premultiply)
            return r;
    }
    return g.premultiply ? g.premultiply(rgb) : AVIF_RESULT_NOT_IMPLEMENTED;
}

AVIF_EXPORT avifResult avifRGBImageUnpremultiplyAlpha(avifRGBImage * rgb)
{
    resolve();
    if (rgb && worthIt(rgb->width, rgb->height)) {
        const avifResult r = avifhipRGBImageUnpremultiplyAlpha(rgb);
        if (!declined(r) || !g.unpremultiply)
            return r;
    }
    return g.unpremultiply ? g.unpremultiply(rgb) : AVIF_RESULT_NOT_IMPLEMENTED;
}

/* Tone mapping.  libavifhip (re)allocates toneMappedImage->pixels with malloc, libavif's avifRGBImageFreePixels releases
 * them with free (avifFree, src/mem.c): the two are interchangeable, also when the call is handed on after a decline. */
AVIF_EXPORT avifResult avifRGBImageApplyGainMap(const avifRGBImage * baseImage,
                                                avifColorPrimaries baseColorPrimaries,
                                                avifTransferCharacteristics baseTransferCharacteristics,
                                                const avifGainMap * gainMap,
                                                float hdrHeadroom,
                                                avifColorPrimaries outputColorPrimaries,
                                                avifTransferCharacteristics outputTransferCharacteristics,
                                                avifRGBImage * toneMappedImage,
                                                avifContentLightLevelInformationBox * clli,
                                                avifDiagnostics * diag)
{
    resolve();
    if (g.gpuGainMap && baseImage && gainMap && toneMappedImage && worthIt(baseImage->width, baseImage->height)) {
        const avifResult r = avifhipRGBImageApplyGainMap(baseImage, baseColorPrimaries, baseTransferCharacteristics, gainMap, hdrHeadroom,
                                                         outputColorPrimaries, outputTransferCharacteristics, toneMappedImage, clli, diag);
        if (!declined(r) || !g.applyGainMap)
            return r;
    }
    return g.applyGainMap ? g.applyGainMap(baseImage, baseColorPrimaries, baseTransferCharacteristics, gainMap, hdrHeadroom, outputColorPrimaries,
                                           outputTransferCharacteristics, toneMappedImage, clli, diag)
                          : AVIF_RESULT_NOT_IMPLEMENTED;
}

/* Gain-map computation.  libavifhip frees the planes gainMap->image owns and mallocs the new ones; libavif's
 * avifImageFreePlanes / avifImageDestroy release them with free: interchangeable, like the pixels above. */
AVIF_EXPORT avifResult avifRGBImageComputeGainMap(const avifRGBImage * baseRgbImage,
                                                  avifColorPrimaries baseColorPrimaries,
                                                  avifTransferCharacteristics baseTransferCharacteristics,
                                                  const avifRGBImage * altRgbImage,
                                                  avifColorPrimaries altColorPrimaries,
                                                  avifTransferCharacteristics altTransferCharacteristics,
                                                  avifGainMap * gainMap,
                                                  avifDiagnostics * diag)
{
    resolve();
    if (g.gpuGainMap && baseRgbImage && altRgbImage && gainMap && gainMap->image && worthIt(baseRgbImage->width, baseRgbImage->height)) {
        const avifImage request = *gainMap->image; /* the requested size / format, should the call be handed on */
        const avifResult r = avifhipRGBImageComputeGainMap(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries,
                                                           altTransferCharacteristics, gainMap, diag);
        if (!declined(r) || !g.computeGainMap)
            return r;
        gainMap->image->width = request.width, gainMap->image->height = request.height;
    }
    return g.computeGainMap ? g.computeGainMap(baseRgbImage, baseColorPrimaries, baseTransferCharacteristics, altRgbImage, altColorPrimaries,
                                               altTransferCharacteristics, gainMap, diag)
                            : AVIF_RESULT_NOT_IMPLEMENTED;
}
