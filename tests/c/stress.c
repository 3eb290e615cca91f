/* stress.c -- the C ABI's host side under concurrency, for the sanitizer builds of the library (libavif_amd/csrc/Makefile: `make tsan asan`).
 *
 *   ./stress[_tsan|_asan] [seconds [threads]]
 *
 * libavif calls its reformat hooks from up to 8 pthreads at once (src/reformat.c:1625-1638); the functions are re-entrant and keep no process
 * state (src/reformat.c:659-661).  The product keeps pooled contexts, a farm of worker threads, download helper threads and cached staging
 * behind the same contract.  This program runs `threads` threads (default 8) for `seconds` (default 10), every thread looping over
 *   YUV -> RGB | RGB -> YUV | premultiply in place | gain-map application (scaled gain map) | plane scaling
 * on host-resident images it allocates and frees anew every iteration, at sizes either side of the farm's threshold, while the main thread
 * switches the device set between { } (no farm), { 0, 0 } and { 0, 0, 0 } every 300 ms -- workers are retired and started under running calls,
 * contexts are leased and handed back by threads that come and go (half of the threads end and are replaced every second).
 * Every result is compared, byte for byte, with the result of the same call made once, single-threaded, before the threads start.
 * Exit code 0 = no mismatch and no failed call.  A sanitizer's findings go to stderr / its log file. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "avifhip.h"

static double nowS(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

enum { OP_Y2R, OP_R2Y, OP_PREMUL, OP_GAINMAP, OP_SCALE, OPS };
static const char * kOpNames[OPS] = { "yuv->rgb", "rgb->yuv", "premultiply", "gain map", "scale" };
/* sizes: small (never farmed), 2.4 MP (one share), 5.2 MP (two shares), 9.4 MP (three shares with three workers) */
static const uint32_t kSizes[][2] = { { 322, 130 }, { 1920, 1280 }, { 2560, 2048 }, { 3840, 2448 } };
enum { SIZES = 4 };

typedef struct Blob
{
    uint8_t * bytes;
    size_t size;
} Blob;
static Blob reference[OPS][SIZES];
static atomic_int failures, mismatches;
static atomic_long calls[OPS];
static atomic_int stopAll;

static void fill(uint8_t * p, size_t n, uint32_t seed)
{
    uint32_t x = seed * 2654435761u + 12345u;
    for (size_t k = 0; k < n; ++k) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;
        p[k] = (uint8_t)x;
    }
}

static void planesAlloc(avifImage * im, uint32_t w, uint32_t h, int format, int withAlpha, uint32_t seed)
{
    memset(im, 0, sizeof(*im));
    im->width = w, im->height = h, im->depth = 8, im->yuvFormat = (avifPixelFormat)format, im->yuvRange = AVIF_RANGE_LIMITED;
    im->colorPrimaries = AVIF_COLOR_PRIMARIES_BT709, im->transferCharacteristics = AVIF_TRANSFER_CHARACTERISTICS_SRGB;
    im->matrixCoefficients = AVIF_MATRIX_COEFFICIENTS_BT709;
    const uint32_t cw = (format == AVIF_PIXEL_FORMAT_YUV444) ? w : (w + 1) / 2, ch = (format == AVIF_PIXEL_FORMAT_YUV420) ? (h + 1) / 2 : h;
    for (int p = 0; p < (format == AVIF_PIXEL_FORMAT_YUV400 ? 1 : 3); ++p) {
        im->yuvRowBytes[p] = p ? cw : w;
        const size_t n = (size_t)im->yuvRowBytes[p] * (p ? ch : h);
        im->yuvPlanes[p] = (uint8_t *)malloc(n);
        fill(im->yuvPlanes[p], n, seed + (uint32_t)p);
        for (size_t k = 0; k < n; ++k) /* limited range: legal codes only */
            im->yuvPlanes[p][k] = (uint8_t)(16 + im->yuvPlanes[p][k] % (p ? 225 : 220));
    }
    im->imageOwnsYUVPlanes = AVIF_TRUE;
    if (withAlpha) {
        im->alphaRowBytes = w;
        im->alphaPlane = (uint8_t *)malloc((size_t)w * h);
        fill(im->alphaPlane, (size_t)w * h, seed + 7);
        im->imageOwnsAlphaPlane = AVIF_TRUE;
    }
}

static void planesFree(avifImage * im)
{
    for (int p = 0; p < 3; ++p)
        free(im->yuvPlanes[p]);
    free(im->alphaPlane);
}

static void pixelsAlloc(avifRGBImage * rgb, uint32_t w, uint32_t h, uint32_t seed, int fillThem)
{
    memset(rgb, 0, sizeof(*rgb));
    rgb->width = w, rgb->height = h, rgb->depth = 8, rgb->format = AVIF_RGB_FORMAT_RGBA;
    rgb->chromaUpsampling = AVIF_CHROMA_UPSAMPLING_BILINEAR, rgb->chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    rgb->maxThreads = 1;
    rgb->rowBytes = w * 4;
    rgb->pixels = (uint8_t *)malloc((size_t)rgb->rowBytes * h);
    if (fillThem)
        fill(rgb->pixels, (size_t)rgb->rowBytes * h, seed);
    else
        memset(rgb->pixels, 0xa5, (size_t)rgb->rowBytes * h);
}

/* one call of `op` at size `s`: the bytes it produced, malloc'ed (NULL: the call failed) */
static Blob runOp(int op, int s)
{
    Blob out = { NULL, 0 };
    const uint32_t w = kSizes[s][0], h = kSizes[s][1], seed = 0x1234u + (uint32_t)(op * 16 + s);
    avifResult r = AVIF_RESULT_OK;
    if (op == OP_Y2R) {
        avifImage im;
        avifRGBImage rgb;
        planesAlloc(&im, w, h, AVIF_PIXEL_FORMAT_YUV420, 0, seed);
        pixelsAlloc(&rgb, w, h, seed, 0);
        r = avifhipImageYUVToRGB(&im, &rgb);
        out.bytes = rgb.pixels, out.size = (size_t)rgb.rowBytes * h;
        planesFree(&im);
    } else if (op == OP_R2Y) {
        avifImage im;
        avifRGBImage rgb;
        planesAlloc(&im, w, h, AVIF_PIXEL_FORMAT_YUV420, 1, seed);
        pixelsAlloc(&rgb, w, h, seed, 1);
        rgb.avoidLibYUV = AVIF_TRUE;
        r = avifhipImageRGBToYUV(&im, &rgb);
        const size_t ny = (size_t)w * h, nc = (size_t)((w + 1) / 2) * ((h + 1) / 2);
        out.size = 2 * ny + 2 * nc, out.bytes = (uint8_t *)malloc(out.size);
        memcpy(out.bytes, im.yuvPlanes[0], ny), memcpy(out.bytes + ny, im.yuvPlanes[1], nc), memcpy(out.bytes + ny + nc, im.yuvPlanes[2], nc);
        memcpy(out.bytes + ny + 2 * nc, im.alphaPlane, ny);
        planesFree(&im);
        free(rgb.pixels);
    } else if (op == OP_PREMUL) {
        avifRGBImage rgb;
        pixelsAlloc(&rgb, w, h, seed, 1);
        rgb.avoidLibYUV = AVIF_TRUE;
        r = avifhipRGBImagePremultiplyAlpha(&rgb);
        out.bytes = rgb.pixels, out.size = (size_t)rgb.rowBytes * h;
    } else if (op == OP_GAINMAP) {
        avifRGBImage base, tone;
        avifImage gmImage;
        avifGainMap gm;
        pixelsAlloc(&base, w, h, seed, 1);
        planesAlloc(&gmImage, (w + 2) / 3, (h + 2) / 3, AVIF_PIXEL_FORMAT_YUV420, 0, seed + 3); /* scaled up by the call: the window scaling kernel */
        gmImage.yuvRange = AVIF_RANGE_FULL;
        memset(&gm, 0, sizeof(gm));
        gm.image = &gmImage;
        for (int c = 0; c < 3; ++c) {
            gm.gainMapMin[c].n = 0, gm.gainMapMin[c].d = 1, gm.gainMapMax[c].n = 3, gm.gainMapMax[c].d = 1;
            gm.gainMapGamma[c].n = 1, gm.gainMapGamma[c].d = 1;
            gm.baseOffset[c].n = 1, gm.baseOffset[c].d = 64, gm.alternateOffset[c].n = 1, gm.alternateOffset[c].d = 64;
        }
        gm.baseHdrHeadroom.n = 0, gm.baseHdrHeadroom.d = 1, gm.alternateHdrHeadroom.n = 3, gm.alternateHdrHeadroom.d = 1;
        gm.useBaseColorSpace = AVIF_TRUE;
        gm.altColorPrimaries = AVIF_COLOR_PRIMARIES_BT709, gm.altTransferCharacteristics = AVIF_TRANSFER_CHARACTERISTICS_PQ;
        memset(&tone, 0, sizeof(tone));
        tone.depth = 10, tone.format = AVIF_RGB_FORMAT_RGBA, tone.maxThreads = 1;
        avifContentLightLevelInformationBox clli = { 0, 0 };
        r = avifhipRGBImageApplyGainMap(&base, AVIF_COLOR_PRIMARIES_BT709, AVIF_TRANSFER_CHARACTERISTICS_SRGB, &gm, 2.0f, AVIF_COLOR_PRIMARIES_BT709,
                                        AVIF_TRANSFER_CHARACTERISTICS_PQ, &tone, &clli, NULL);
        if (r == AVIF_RESULT_OK) {
            out.size = (size_t)tone.rowBytes * tone.height + 4, out.bytes = (uint8_t *)malloc(out.size);
            memcpy(out.bytes, tone.pixels, out.size - 4);
            memcpy(out.bytes + out.size - 4, &clli.maxCLL, 2), memcpy(out.bytes + out.size - 2, &clli.maxCLL, 2); /* (maxPALL: fp64 partial sums, order-free) */
        }
        free(tone.pixels);
        free(base.pixels);
        planesFree(&gmImage);
    } else {
        avifImage im;
        planesAlloc(&im, w, h, AVIF_PIXEL_FORMAT_YUV420, 1, seed);
        const uint32_t dw = w * 2 / 3, dh = h * 2 / 3;
        r = avifhipImageScale(&im, dw, dh);
        if (r == AVIF_RESULT_OK) {
            const size_t ny = (size_t)im.yuvRowBytes[0] * dh, nc = (size_t)im.yuvRowBytes[1] * ((dh + 1) / 2), na = (size_t)im.alphaRowBytes * dh;
            out.size = ny + 2 * nc + na, out.bytes = (uint8_t *)malloc(out.size);
            memcpy(out.bytes, im.yuvPlanes[0], ny), memcpy(out.bytes + ny, im.yuvPlanes[1], nc), memcpy(out.bytes + ny + nc, im.yuvPlanes[2], nc);
            memcpy(out.bytes + ny + 2 * nc, im.alphaPlane, na);
        }
        planesFree(&im);
    }
    if (r != AVIF_RESULT_OK) {
        fprintf(stderr, "stress: %s at %ux%u failed with %d (%s)\n", kOpNames[op], w, h, (int)r, avifhipLastError());
        free(out.bytes);
        out.bytes = NULL, out.size = 0;
    }
    return out;
}

typedef struct Worker
{
    int index;
    double until;
} Worker;

static void * threadMain(void * arg)
{
    const Worker * me = (const Worker *)arg;
    uint32_t x = 0x9e3779b9u * (uint32_t)(me->index + 1);
    while (!atomic_load(&stopAll) && nowS() < me->until) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;
        const int op = (int)(x % OPS), s = (int)((x >> 8) % SIZES);
        Blob got = runOp(op, s);
        atomic_fetch_add(&calls[op], 1);
        if (!got.bytes) {
            atomic_fetch_add(&failures, 1);
        } else if (got.size != reference[op][s].size || memcmp(got.bytes, reference[op][s].bytes, got.size) != 0) {
            fprintf(stderr, "stress: %s at %ux%u differs from the single-threaded result\n", kOpNames[op], kSizes[s][0], kSizes[s][1]);
            atomic_fetch_add(&mismatches, 1);
        }
        free(got.bytes);
    }
    return NULL;
}

int main(int argc, char ** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    int threads = argc > 2 ? atoi(argv[2]) : 8;
    if (threads < 1 || threads > 64)
        threads = 8;
    if (avifhipDeviceCount() <= 0) {
        fprintf(stderr, "stress: no HIP device visible\n");
        return 2;
    }
    avifhipSetArithmetic(0);
    for (int op = 0; op < OPS; ++op)
        for (int s = 0; s < SIZES; ++s) {
            reference[op][s] = runOp(op, s);
            if (!reference[op][s].bytes)
                return 2;
        }
    printf("stress: %d threads for %.0f s over %d operations x %d sizes; device set switched every 300 ms\n", threads, seconds, OPS, SIZES);
    const double t0 = nowS(), tEnd = t0 + seconds;
    pthread_t tid[64];
    Worker w[64];
    for (int k = 0; k < threads; ++k) {
        /* odd threads end after a second and are replaced: contexts are handed back to the pool and leased again */
        w[k].index = k, w[k].until = (k & 1) ? t0 + 1.0 : tEnd;
        pthread_create(&tid[k], NULL, threadMain, &w[k]);
    }
    const int sets[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    int phase = 0, generation = 1;
    double nextSwitch = t0 + 0.3, nextRespawn = t0 + 1.0;
    while (nowS() < tEnd) {
        struct timespec nap = { 0, 20 * 1000 * 1000 };
        nanosleep(&nap, NULL);
        if (nowS() >= nextSwitch) {
            phase = (phase + 1) % 3;
            if (avifhipSetDeviceSet(phase ? sets[phase] : NULL, phase ? (uint32_t)(phase + 1) : 0u) != AVIF_RESULT_OK)
                atomic_fetch_add(&failures, 1);
            nextSwitch += 0.3;
        }
        if (nowS() >= nextRespawn && nowS() + 1.0 < tEnd) {
            for (int k = 1; k < threads; k += 2) {
                pthread_join(tid[k], NULL);
                w[k].index = k + 64 * generation, w[k].until = nowS() + 1.0;
                pthread_create(&tid[k], NULL, threadMain, &w[k]);
            }
            ++generation;
            nextRespawn += 1.0;
        }
    }
    atomic_store(&stopAll, 1);
    for (int k = 0; k < threads; ++k)
        pthread_join(tid[k], NULL);
    (void)avifhipSetDeviceSet(NULL, 0);
    long total = 0;
    for (int op = 0; op < OPS; ++op) {
        printf("  %-12s %ld calls\n", kOpNames[op], atomic_load(&calls[op]));
        total += atomic_load(&calls[op]);
    }
    printf("stress: %ld calls, %d failed, %d mismatched, %d generations of short-lived threads\n", total, atomic_load(&failures), atomic_load(&mismatches), generation);
    return (atomic_load(&failures) || atomic_load(&mismatches)) ? 1 : 0;
}
