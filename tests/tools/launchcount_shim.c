/* launchcount_shim.c -- test tool (not shipped).  LD_PRELOADed into an unmodified reference program (tests/avifyuv.c built against the
 * hip-backed libavif): at process exit it prints how many kernels libavifhip.so launched, so that a test can tell "ran on the GPU" from
 * "fell back to libavif's CPU code". */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>

__attribute__((destructor)) static void reportLaunches(void)
{
    uint64_t (*count)(void) = (uint64_t (*)(void))dlsym(RTLD_DEFAULT, "avifhipLaunchCount");
    fprintf(stderr, "avifhip launches=%lld\n", count ? (long long)count() : -1LL);
}
