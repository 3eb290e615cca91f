/*
 * pack_oracle.c -- TEST INFRASTRUCTURE (never linked or imported by the product).  CPU restatement of the row packing the
 * reference's file writers perform next to the reformat path:
 *   oraclePackY4MFrame  the payload loop of y4mWrite, /root/reference/apps/shared/y4m.c:603-618: planes Y..V (and A for 8-bit
 *                       4:4:4 with alpha, :483-489, :517-523), each row cut to avifImagePlaneWidth << (depth > 8) bytes.
 *                       PINNED against the reference's own y4mWrite (oracle/_ref/libavifutil_ref.so, tests/test_pack.py).
 *   oraclePackPNGRows   the rows avifPNGWrite passes to png_write_image (apps/shared/avifpng.c:865-880) after png_set_swap
 *                       (:877) for depths above 8: every 16-bit sample byte-swapped.  libpng is a third-party dependency that is
 *                       not in the reference tree; png_set_swap is restated from its documented behaviour (libpng manual,
 *                       "png_set_swap(): swap the bytes of 16-bit samples to big-endian"), checked against numpy's byteswap.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "reformat_oracle.h"

static uint32_t packPlaneWidth(const avifImage * im, int p) /* avifImagePlaneWidth, src/avif.c:351-375 */
{
    if (p == 0 || p == 3)
        return im->width;
    if (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)
        return 0;
    return (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV444) ? im->width : (im->width + 1) >> 1;
}
static uint32_t packPlaneHeight(const avifImage * im, int p) /* avifImagePlaneHeight, src/avif.c:377-400 */
{
    if (p == 0 || p == 3)
        return im->height;
    if (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)
        return 0;
    return (im->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? (im->height + 1) >> 1 : im->height;
}

/* returns the number of bytes written (the size of the frame payload); out may be NULL to query it */
size_t oraclePackY4MFrame(const avifImage * im, int withAlpha, uint8_t * out)
{
    size_t n = 0;
    const int lastPlane = withAlpha ? 3 : 2;
    for (int p = 0; p <= lastPlane; ++p) {
        const uint8_t * row = (p < 3) ? im->yuvPlanes[p] : im->alphaPlane;
        const uint32_t rowBytes = (p < 3) ? im->yuvRowBytes[p] : im->alphaRowBytes;
        const uint32_t h = row ? packPlaneHeight(im, p) : 0;
        const uint32_t wb = packPlaneWidth(im, p) << (im->depth > 8);
        for (uint32_t y = 0; y < h; ++y) {
            if (out)
                memcpy(out + n, row, wb);
            n += wb;
            row += rowBytes;
        }
    }
    return n;
}

size_t oraclePackPNGRows(const avifRGBImage * rgb, uint32_t pixelBytes, uint8_t * out)
{
    const size_t wb = (size_t)rgb->width * pixelBytes;
    for (uint32_t y = 0; y < rgb->height; ++y) {
        const uint8_t * row = rgb->pixels + (size_t)y * rgb->rowBytes;
        uint8_t * dst = out + (size_t)y * wb;
        if (rgb->depth > 8) {
            for (size_t i = 0; i + 1 < wb; i += 2) /* png_set_swap */
                dst[i] = row[i + 1], dst[i + 1] = row[i];
        } else {
            memcpy(dst, row, wb);
        }
    }
    return wb * rgb->height;
}
