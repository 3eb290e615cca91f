/* farm_check.c -- a C consumer of libavifhip.so (no Python anywhere): the in-process device farm behind the C ABI, checked against the oracle.
 *
 *   AVIFHIP_DEVICES=0,0 ./farm_check [width height depth [cfg2|cfg5|cfg4 [plane row padding in bytes]]]
 *
 * Builds a synthetic host-resident image, converts it with the product's synchronous entry point under the device set the environment (or
 * -d <list>) names, converts the same image with the oracle (liboracle.so: the CPU restatement of libavif's reformat path -- test
 * infrastructure, linked by this checker only) and compares every byte, row padding included.  Prints what each worker moved over its host
 * link and the host-to-host time per call.  Exit code 0 = byte-identical and farmed as announced (avifhipPlanFarmRows).
 *
 * Workloads (BASELINE.json): cfg2 = 8-bit 4:2:0 BT.709 limited -> RGBA8 bilinear, API defaults (the integer path; compared with the
 * libyuv-build oracle); cfg5 = 10-bit 4:2:0 -> RGBA at the image's depth, bilinear (the stitched canvas of the 8 x 8 grid; fp32 path);
 * cfg4 = RGBA8 -> 8-bit 4:2:0 BT.709 + alpha plane (avifImageRGBToYUV).
 * Build: tests/c/Makefile. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "avifhip.h"
#include "reformat_oracle.h"


static double nowMs(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int fail(const char * what)
{
    fprintf(stderr, "farm_check: %s (%s)\n", what, avifhipLastError());
    return 1;
}

static void fillPlanes(avifImage * image, uint32_t seed)
{
    const uint32_t bps = image->depth > 8 ? 2 : 1, maxv = (1u << image->depth) - 1;
    const uint32_t cw = (image->width + 1) / 2, ch = (image->height + 1) / 2;
    for (int p = 0; p < 4; ++p) {
        uint8_t * plane = p < 3 ? image->yuvPlanes[p] : image->alphaPlane;
        if (!plane)
            continue;
        const uint32_t w = (p == 1 || p == 2) ? cw : image->width, h = (p == 1 || p == 2) ? ch : image->height;
        seed = avifhipSynthFill(seed, plane, p < 3 ? image->yuvRowBytes[p] : image->alphaRowBytes, w, h, bps, 0, maxv);
    }
}

static int reportFarm(uint32_t width, uint32_t height, uint32_t expectWorkers)
{
    const uint32_t n = avifhipLastFarmWorkers();
    uint64_t up = 0, down = 0;
    avifhipLastTransferBytes(&up, &down);
    printf("  workers %u (announced %u), host link: %.1f MB up, %.1f MB down in total\n", n, expectWorkers, up / 1e6, down / 1e6);
    for (uint32_t k = 0; k < n; ++k) {
        int dev = -1;
        uint32_t b = 0, e = 0;
        uint64_t u = 0, d = 0;
        if (avifhipLastFarmTransferBytes(k, &dev, &b, &e, &u, &d) != AVIF_RESULT_OK)
            return fail("avifhipLastFarmTransferBytes");
        printf("    worker %u on device %d: rows [%u, %u) of %ux%u, %.1f MB up (%.1f %%), %.1f MB down (%.1f %%)\n", k, dev, b, e, width, height, u / 1e6,
               100.0 * u / (up ? up : 1), d / 1e6, 100.0 * d / (down ? down : 1));
    }
    if (n != (expectWorkers >= 2 ? expectWorkers : 0)) {
        fprintf(stderr, "farm_check: %u workers took part, avifhipPlanFarmRows announced %u\n", n, expectWorkers);
        return 1;
    }
    return 0;
}

int main(int argc, char ** argv)
{
    uint32_t width = 7680, height = 4320, depth = 8;
    const char * workload = "cfg2";
    if (argc >= 4)
        width = (uint32_t)atoi(argv[1]), height = (uint32_t)atoi(argv[2]), depth = (uint32_t)atoi(argv[3]);
    if (argc >= 5)
        workload = argv[4];
    const uint32_t pad = argc >= 6 ? (uint32_t)atoi(argv[5]) : 64u; /* rows padded by this much: the padding must come back untouched */
    if (!width || !height || (depth != 8 && depth != 10 && depth != 12))
        return fail("usage: farm_check [width height depth [cfg2|cfg5|cfg4]]");
    if (avifhipDeviceCount() <= 0)
        return fail("no HIP device visible: the HIP path is the only path");
    int set[64];
    const uint32_t workers = avifhipGetDeviceSet(set, 64);
    uint32_t shares = 1;
    if (avifhipPlanFarmRows(width, height, workers, NULL, 0, &shares) != AVIF_RESULT_OK)
        return fail("avifhipPlanFarmRows");
    printf("farm_check: %s %ux%u depth %u, device set of %u (AVIFHIP_DEVICES=%s): %u share(s)\n", workload, width, height, depth, workers,
           getenv("AVIFHIP_DEVICES") ? getenv("AVIFHIP_DEVICES") : "", shares);

    const uint32_t bps = depth > 8 ? 2 : 1, cw = (width + 1) / 2, ch = (height + 1) / 2;
    avifImage image;
    memset(&image, 0, sizeof(image));
    image.width = width, image.height = height, image.depth = depth;
    image.yuvFormat = AVIF_PIXEL_FORMAT_YUV420, image.yuvRange = AVIF_RANGE_LIMITED;
    image.colorPrimaries = AVIF_COLOR_PRIMARIES_BT709, image.transferCharacteristics = AVIF_TRANSFER_CHARACTERISTICS_BT709;
    image.matrixCoefficients = AVIF_MATRIX_COEFFICIENTS_BT709;
    const int encode = !strcmp(workload, "cfg4");
    const int wide = !strcmp(workload, "cfg5");
    for (int p = 0; p < 3; ++p) {
        image.yuvRowBytes[p] = (p ? cw : width) * bps + pad;
        image.yuvPlanes[p] = (uint8_t *)malloc((size_t)image.yuvRowBytes[p] * (p ? ch : height));
        if (!image.yuvPlanes[p])
            return fail("out of memory");
        memset(image.yuvPlanes[p], 0x5a, (size_t)image.yuvRowBytes[p] * (p ? ch : height));
    }
    if (encode) {
        image.alphaRowBytes = width * bps + pad;
        image.alphaPlane = (uint8_t *)malloc((size_t)image.alphaRowBytes * height);
        memset(image.alphaPlane, 0x5a, (size_t)image.alphaRowBytes * height);
    }
    avifRGBImage rgb;
    memset(&rgb, 0, sizeof(rgb));
    rgb.width = width, rgb.height = height, rgb.depth = wide ? depth : 8, rgb.format = AVIF_RGB_FORMAT_RGBA;
    rgb.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_BILINEAR, rgb.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    rgb.maxThreads = 1;
    rgb.rowBytes = width * 4 * (rgb.depth > 8 ? 2 : 1) + 32;
    const size_t pixelBytes = (size_t)rgb.rowBytes * height;
    uint8_t * got = (uint8_t *)malloc(pixelBytes), * want = (uint8_t *)malloc(pixelBytes);
    if (!got || !want)
        return fail("out of memory");
    int bad = 0;
    if (!encode) {
        fillPlanes(&image, 0x12345678u);
        memset(got, 0xa5, pixelBytes), memset(want, 0xa5, pixelBytes);
        rgb.pixels = want;
        const double c0 = nowMs();
        const avifResult ro = wide ? oracleImageYUVToRGB(&image, &rgb) : oracleLibyuvImageYUVToRGB(&image, &rgb);
        const double c1 = nowMs();
        if (ro != AVIF_RESULT_OK)
            return fail("the oracle refused the conversion");
        rgb.pixels = got;
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) { /* (the first call builds the workers' contexts and staging buffers) */
            const double t0 = nowMs();
            if (avifhipImageYUVToRGB(&image, &rgb) != AVIF_RESULT_OK)
                return fail("avifhipImageYUVToRGB");
            const double t1 = nowMs();
            if (rep && t1 - t0 < best)
                best = t1 - t0;
        }
        printf("  %s, host to host: %.2f ms per call (%.0f MP/s); oracle on one host core: %.0f ms\n", avifhipLastKernel(), best, width * (double)height / 1e3 / best, c1 - c0);
        if (reportFarm(width, height, shares))
            return 1;
        bad = memcmp(got, want, pixelBytes) != 0;
        if (bad) {
            size_t k = 0;
            while (got[k] == want[k])
                ++k;
            fprintf(stderr, "farm_check: first difference at row %zu byte %zu: got %u, oracle %u\n", k / rgb.rowBytes, k % rgb.rowBytes, got[k], want[k]);
        }
    } else {
        /* RGBA8 -> 4:2:0 + alpha: the pixels are the input; planes from the oracle and from the product */
        rgb.pixels = got;
        (void)avifhipSynthFill(0xcafebabeu, got, rgb.rowBytes, width * 4, height, 1, 0, 255);
        rgb.avoidLibYUV = AVIF_TRUE; /* BT.709: libyuv declines anyway (BT.601 only) */
        avifImage ref = image;
        for (int p = 0; p < 3; ++p) {
            ref.yuvPlanes[p] = (uint8_t *)malloc((size_t)image.yuvRowBytes[p] * (p ? ch : height));
            memset(ref.yuvPlanes[p], 0x5a, (size_t)image.yuvRowBytes[p] * (p ? ch : height));
        }
        ref.alphaPlane = (uint8_t *)malloc((size_t)image.alphaRowBytes * height);
        memset(ref.alphaPlane, 0x5a, (size_t)image.alphaRowBytes * height);
        if (oracleImageRGBToYUV(&ref, &rgb) != AVIF_RESULT_OK)
            return fail("the oracle refused the conversion");
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            const double t0 = nowMs();
            if (avifhipImageRGBToYUV(&image, &rgb) != AVIF_RESULT_OK)
                return fail("avifhipImageRGBToYUV");
            const double t1 = nowMs();
            if (rep && t1 - t0 < best)
                best = t1 - t0;
        }
        printf("  %s, host to host: %.2f ms per call (%.0f MP/s)\n", avifhipLastKernel(), best, width * (double)height / 1e3 / best);
        if (reportFarm(width, height, shares))
            return 1;
        for (int p = 0; p < 4 && !bad; ++p) {
            const uint8_t * a = p < 3 ? image.yuvPlanes[p] : image.alphaPlane, * b = p < 3 ? ref.yuvPlanes[p] : ref.alphaPlane;
            const size_t n = (size_t)(p < 3 ? image.yuvRowBytes[p] : image.alphaRowBytes) * ((p == 1 || p == 2) ? ch : height);
            if (memcmp(a, b, n)) {
                fprintf(stderr, "farm_check: plane %d differs from the oracle's\n", p);
                bad = 1;
            }
        }
    }
    printf(bad ? "farm_check: MISMATCH\n" : "farm_check: byte-identical to the oracle\n");
    return bad;
}
