/*
 * avifutil_stubs.c -- TEST INFRASTRUCTURE.  apps/shared/avifutil.c (compiled from the reference tree into
 * _ref/libavifutil_ref.so to pin the pixel-transform oracle) refers to the JPEG / PNG file readers of the
 * reference's tools from avifReadImage(); those front ends need libjpeg / libpng and are never called by the tests.
 * These stubs only satisfy the linker.
 */
#include <stdlib.h>

int avifJPEGRead(void)
{
    abort();
}
int avifPNGRead(void)
{
    abort();
}
