// pkbench_wide.hip -- where do the microseconds of the packed kernel go on 64 tiles of 1920x1080 10-bit 4:2:0 -> RGBA8 (cfg5x64_8: 929 MB
// that cannot stay in the Infinity Cache)?  (run on the GPU box; not a test, not shipped)  Includes the product's tile_pk_impl.h, instantiates
// the batch kernel, times it with HIP events.  Variants: -DAVIFHIP_ABLATE_MATRIX / -DAVIFHIP_ABLATE_FILTER / -DAVIFHIP_ABLATE_STAGE /
// -DPKB_NSW=2|4 / -DPKB_BIL=false; arguments: name wavesXLog2 chunkRows.   tests/tools/pkbench.sh (with PKB_SRC=pkbench_wide.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "tile_fx_impl.h"
#ifndef PKB_NSW
#define PKB_NSW 4
#endif
#ifndef PKB_BIL
#define PKB_BIL true
#endif
#ifndef PKB_STREAM
#define PKB_STREAM false // -DPKB_STREAM=true: luma rows as streaming (non-temporal) loads
#endif
#ifndef PKB_TUNING
#define PKB_TUNING TUNE_DEFAULT
#endif
using namespace avifhip;
using namespace avifhip::tile;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char ** argv)
{
    const uint32_t W = 1920, H = 1080, NJ = 64;
    const char * name = argc > 1 ? argv[1] : "pkbench_wide";
    const uint32_t wavesXLog2 = argc > 2 ? (uint32_t)atoi(argv[2]) : 0, chunkRows = argc > 3 ? (uint32_t)atoi(argv[3]) : 1;
    std::vector<uint16_t> host((size_t)W * H + 20000);
    uint32_t x = 0x12345678u;
    for (size_t i = 0; i < host.size(); ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; host[i] = (uint16_t)(64 + x % 880); }
    std::vector<TileArgs> args(NJ);
    for (uint32_t k = 0; k < NJ; ++k) {
        uint8_t *y, *u, *v, *o;
        CK(hipMalloc(&y, (size_t)W * H * 2)); CK(hipMalloc(&u, (size_t)2048 * H / 2)); CK(hipMalloc(&v, (size_t)2048 * H / 2)); CK(hipMalloc(&o, (size_t)W * H * 4));
        CK(hipMemcpy(y, host.data() + k, (size_t)W * H * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(u, host.data() + 1000 + k, (size_t)2048 * H / 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(v, host.data() + 7777 + k, (size_t)2048 * H / 2, hipMemcpyHostToDevice));
        avifImage img; memset(&img, 0, sizeof(img));
        img.width = W; img.height = H; img.depth = 10; img.yuvFormat = AVIF_PIXEL_FORMAT_YUV420; img.yuvRange = AVIF_RANGE_LIMITED;
        img.matrixCoefficients = 1;
        img.yuvPlanes[0] = y; img.yuvPlanes[1] = u; img.yuvPlanes[2] = v;
        img.yuvRowBytes[0] = W * 2; img.yuvRowBytes[1] = 2048; img.yuvRowBytes[2] = 2048;
        avifRGBImage rgb; memset(&rgb, 0, sizeof(rgb));
        rgb.width = W; rgb.height = H; rgb.depth = 8; rgb.format = AVIF_RGB_FORMAT_RGBA;
        rgb.chromaUpsampling = PKB_BIL ? AVIF_CHROMA_UPSAMPLING_BILINEAR : AVIF_CHROMA_UPSAMPLING_NEAREST; rgb.avoidLibYUV = 0; rgb.maxThreads = 1;
        rgb.pixels = o; rgb.rowBytes = W * 4;
        YuvToRgbPlan plan;
        if (makeYuvToRgbPlan(&img, &rgb, nullptr, 0, PKB_TUNING, &plan) != AVIF_RESULT_OK || plan.arith != ARITH_LIBYUV || plan.fxDownshift) { printf("plan failed\n"); return 1; }
        args[k] = distillArgs(plan);
    }
    TileArgs * table; CK(hipMalloc(&table, sizeof(TileArgs) * NJ)); CK(hipMemcpy(table, args.data(), sizeof(TileArgs) * NJ, hipMemcpyHostToDevice));
    TileLaunch L; memset(&L, 0, sizeof(L));
    L.count = NJ; L.maxW4 = W; L.maxH2 = H; L.pkStrips = PKB_NSW; L.wavesXLog2 = wavesXLog2; L.chunkRows = chunkRows;
    uint32_t nsw, blocks; PkGeom g;
    pkGeometry(L, W, H, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock), grid(blocks, 1, NJ);
    const uint32_t ldsBytes = 4u * (uint32_t)PkLds<SUB_420, PKB_BIL, 4, PKB_NSW, false>::kPlain;
    auto launch = [&]() { hipLaunchKernelGGL((yuvToRgbPkBatchKernel<SUB_420, PKB_BIL, 4, false, PKB_NSW, false, WIDE_NATIVE, PKB_STREAM>), grid, block, ldsBytes, 0, table, g); };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 300; ++i) launch(); // clock ramp
    std::vector<float> t;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(b);
        CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms / 10 * 1000.0f);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s grid %5u x %u  %7.2f us  %5.2f TB/s\n", name, blocks, NJ, t[3], 64.0 * W * H * 7.0 / (t[3] * 1e-6) / 1e12);
    return 0;
}
