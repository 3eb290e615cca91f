/* preload_probe.c -- test tool: a stand-in for an application linked against a SHARED libavif.  Converts one synthetic
 * image with avifImageYUVToRGB / avifImageRGBToYUV / premultiply (whatever the dynamic linker binds those names to),
 * writes the raw outputs to the file named on the command line and prints how many kernels libavifhip launched (0 when
 * the interposer is not loaded).  Run with and without LD_PRELOAD=libavifhip_preload.so (tests/test_gpu_preload.py). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "avifhip/avif_abi.h"

avifResult avifImageYUVToRGB(const avifImage * image, avifRGBImage * rgb);
avifResult avifImageRGBToYUV(avifImage * image, const avifRGBImage * rgb);
avifResult avifRGBImagePremultiplyAlpha(avifRGBImage * rgb);
avifResult avifRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries, avifTransferCharacteristics baseTransferCharacteristics,
                                    const avifGainMap * gainMap, float hdrHeadroom, avifColorPrimaries outputColorPrimaries,
                                    avifTransferCharacteristics outputTransferCharacteristics, avifRGBImage * toneMappedImage,
                                    avifContentLightLevelInformationBox * clli, avifDiagnostics * diag);

static uint32_t rnd(uint32_t * s)
{
    *s ^= *s << 13, *s ^= *s >> 17, *s ^= *s << 5;
    return *s;
}

int main(int argc, char ** argv)
{
    const uint32_t W = 640, H = 360;
    uint32_t seed = 0x12345678u;
    avifImage img;
    memset(&img, 0, sizeof(img));
    img.width = W, img.height = H, img.depth = 8, img.yuvFormat = AVIF_PIXEL_FORMAT_YUV420, img.yuvRange = AVIF_RANGE_LIMITED;
    img.matrixCoefficients = AVIF_MATRIX_COEFFICIENTS_BT709, img.colorPrimaries = 1, img.transferCharacteristics = 1;
    const uint32_t cw = W / 2, ch = H / 2;
    img.yuvPlanes[0] = malloc((size_t)W * H), img.yuvRowBytes[0] = W;
    img.yuvPlanes[1] = malloc((size_t)cw * ch), img.yuvRowBytes[1] = cw;
    img.yuvPlanes[2] = malloc((size_t)cw * ch), img.yuvRowBytes[2] = cw;
    for (size_t k = 0; k < (size_t)W * H; ++k)
        img.yuvPlanes[0][k] = (uint8_t)(16 + rnd(&seed) % 220);
    for (int p = 1; p <= 2; ++p)
        for (size_t k = 0; k < (size_t)cw * ch; ++k)
            img.yuvPlanes[p][k] = (uint8_t)(16 + rnd(&seed) % 225);
    avifRGBImage rgb;
    memset(&rgb, 0, sizeof(rgb));
    rgb.width = W, rgb.height = H, rgb.depth = 8, rgb.format = AVIF_RGB_FORMAT_RGBA, rgb.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_BILINEAR;
    rgb.maxThreads = 1, rgb.rowBytes = W * 4, rgb.pixels = malloc((size_t)W * H * 4);
    memset(rgb.pixels, 0xA5, (size_t)W * H * 4);
    const avifResult r1 = avifImageYUVToRGB(&img, &rgb);

    /* back to YUV 4:4:4 from the converted pixels, into fresh planes */
    avifImage back;
    memset(&back, 0, sizeof(back));
    back.width = W, back.height = H, back.depth = 8, back.yuvFormat = AVIF_PIXEL_FORMAT_YUV444, back.yuvRange = AVIF_RANGE_FULL;
    back.matrixCoefficients = AVIF_MATRIX_COEFFICIENTS_BT601;
    uint8_t * planes = malloc((size_t)W * H * 4);
    for (int p = 0; p < 3; ++p)
        back.yuvPlanes[p] = planes + (size_t)p * W * H, back.yuvRowBytes[p] = W;
    back.alphaPlane = planes + (size_t)3 * W * H, back.alphaRowBytes = W;
    const avifResult r2 = avifImageRGBToYUV(&back, &rgb);

    for (size_t k = 3; k < (size_t)W * H * 4; k += 4)
        rgb.pixels[k] = (uint8_t)(rnd(&seed) & 0xff);
    const avifResult r3 = avifRGBImagePremultiplyAlpha(&rgb);

    /* tone-map the premultiplied pixels with a half-size 4:2:0 gain map (a view on the first image's planes): sRGB -> 10-bit PQ BT.2020 */
    avifImage gainImage = img;
    gainImage.width = W / 2, gainImage.height = H / 2;
    gainImage.yuvRange = AVIF_RANGE_FULL, gainImage.matrixCoefficients = AVIF_MATRIX_COEFFICIENTS_BT601;
    avifGainMap gm;
    memset(&gm, 0, sizeof(gm));
    gm.image = &gainImage;
    for (int c = 0; c < 3; ++c) {
        gm.gainMapMin[c].n = 0, gm.gainMapMin[c].d = 1, gm.gainMapMax[c].n = 3, gm.gainMapMax[c].d = 1;
        gm.gainMapGamma[c].n = 1, gm.gainMapGamma[c].d = 1;
        gm.baseOffset[c].n = 1, gm.baseOffset[c].d = 64, gm.alternateOffset[c].n = 1, gm.alternateOffset[c].d = 64;
    }
    gm.baseHdrHeadroom.n = 0, gm.baseHdrHeadroom.d = 1, gm.alternateHdrHeadroom.n = 3, gm.alternateHdrHeadroom.d = 1;
    gm.useBaseColorSpace = AVIF_TRUE;
    avifRGBImage tone;
    memset(&tone, 0, sizeof(tone));
    tone.depth = 10, tone.format = AVIF_RGB_FORMAT_RGBA, tone.maxThreads = 1;
    avifContentLightLevelInformationBox clli = { 0, 0 };
    avifDiagnostics diag;
    const avifResult r4 = avifRGBImageApplyGainMap(&rgb, 1, 13, &gm, 2.0f, 9, 16, &tone, &clli, &diag);

    FILE * f = fopen(argc > 1 ? argv[1] : "/dev/null", "wb");
    if (!f)
        return 2;
    fwrite(rgb.pixels, 1, (size_t)W * H * 4, f);
    fwrite(planes, 1, (size_t)W * H * 4, f);
    if (r4 == AVIF_RESULT_OK)
        fwrite(tone.pixels, 1, (size_t)tone.rowBytes * tone.height, f);
    fwrite(&clli.maxCLL, 1, sizeof(clli.maxCLL), f);
    fclose(f);
    if (r3 == AVIF_RESULT_OK && r4 != AVIF_RESULT_OK)
        return 3;
    uint64_t (*launches)(void) = (uint64_t(*)(void))dlsym(RTLD_DEFAULT, "avifhipLaunchCount");
    printf("results %d %d %d launches %llu\n", (int)r1, (int)r2, (int)r3, launches ? (unsigned long long)launches() : 0ull);
    return 0;
}
