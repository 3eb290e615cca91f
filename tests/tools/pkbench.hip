// pkbench.hip -- where do the microseconds of the packed 16-bit cfg2 kernel go?  (run on the GPU box; not a test, not shipped)
// Includes the product's tile_pk_impl.h directly, instantiates the 4:2:0 bilinear RGBA8 kernel, times it with HIP events with
// 4 frames cycled (inputs fit the Infinity Cache) and 12 (nothing does).  Build variants with -DAVIFHIP_ABLATE_MATRIX /
// -DAVIFHIP_ABLATE_FILTER / -DPKB_NSW=2|4 and compare:  tests/tools/pkbench.sh
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
#ifndef PKB_TUNING
#define PKB_TUNING TUNE_DEFAULT
#endif
using namespace avifhip;
using namespace avifhip::tile;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void checksumKernel(const uint32_t * p, size_t n, unsigned long long * out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned)((i & 1023) + 1);
    atomicAdd(out, s);
}

int main(int argc, char ** argv)
{
    const uint32_t W = 7680, H = 4320;
    const int NB = 12;
    const char * name = argc > 1 ? argv[1] : "pkbench";
    const uint32_t wavesXLog2 = argc > 2 ? (uint32_t)atoi(argv[2]) : 0, chunkRows = argc > 3 ? (uint32_t)atoi(argv[3]) : 1;
    uint8_t *y[NB], *u[NB], *v[NB], *o[NB];
    std::vector<uint8_t> host((size_t)W * H);
    uint32_t x = 0x12345678u;
    for (int k = 0; k < NB; ++k) {
        CK(hipMalloc(&y[k], (size_t)W * H)); CK(hipMalloc(&u[k], (size_t)W * H / 4)); CK(hipMalloc(&v[k], (size_t)W * H / 4)); CK(hipMalloc(&o[k], (size_t)W * H * 4));
        if (k < 2) for (size_t i = 0; i < host.size(); ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; host[i] = 16 + x % 220; }
        CK(hipMemcpy(y[k], host.data(), (size_t)W * H, hipMemcpyHostToDevice));
        CK(hipMemcpy(u[k], host.data() + 1000 + k, (size_t)W * H / 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(v[k], host.data() + 7777 + k, (size_t)W * H / 4, hipMemcpyHostToDevice));
        CK(hipMemset(o[k], 0, (size_t)W * H * 4));
    }
    TileArgs args[NB];
    for (int k = 0; k < NB; ++k) {
        avifImage img; memset(&img, 0, sizeof(img));
        img.width = W; img.height = H; img.depth = 8; img.yuvFormat = AVIF_PIXEL_FORMAT_YUV420; img.yuvRange = AVIF_RANGE_LIMITED;
        img.matrixCoefficients = 1;
        img.yuvPlanes[0] = y[k]; img.yuvPlanes[1] = u[k]; img.yuvPlanes[2] = v[k];
        img.yuvRowBytes[0] = W; img.yuvRowBytes[1] = W / 2; img.yuvRowBytes[2] = W / 2;
        avifRGBImage rgb; memset(&rgb, 0, sizeof(rgb));
        rgb.width = W; rgb.height = H; rgb.depth = 8; rgb.format = AVIF_RGB_FORMAT_RGBA;
        rgb.chromaUpsampling = PKB_BIL ? AVIF_CHROMA_UPSAMPLING_BILINEAR : AVIF_CHROMA_UPSAMPLING_NEAREST; rgb.avoidLibYUV = 0; rgb.maxThreads = 1;
        rgb.pixels = o[k]; rgb.rowBytes = W * 4;
        YuvToRgbPlan plan;
        if (makeYuvToRgbPlan(&img, &rgb, nullptr, 0, PKB_TUNING, &plan) != AVIF_RESULT_OK || plan.arith != ARITH_LIBYUV) { printf("plan failed\n"); return 1; }
        args[k] = distillArgs(plan);
    }
    TileLaunch L; memset(&L, 0, sizeof(L));
    L.count = 1; L.maxW4 = W; L.maxH2 = H; L.pkStrips = PKB_NSW; L.wavesXLog2 = wavesXLog2; L.chunkRows = chunkRows;
    uint32_t nsw, blocks; PkGeom g;
    pkGeometry(L, W, H, &nsw, &g, &blocks);
    const dim3 block(kLanesX, kWavesPerBlock), grid(blocks);
    const uint32_t ldsBytes = 4u * (uint32_t)PkLds<SUB_420, PKB_BIL, 4, PKB_NSW, false>::kPlain;
    auto launch = [&](int k) { hipLaunchKernelGGL((yuvToRgbPkKernel<SUB_420, PKB_BIL, 4, false, PKB_NSW, false, WIDE_NONE>), grid, block, ldsBytes, 0, args[k], g); };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3000; ++i) launch(i % 4); // clock ramp
    float res[2];
    int m = 0;
    for (int n : { 4, NB }) {
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            for (int i = 0; i < 8; ++i) launch(i % n);
            hipEventRecord(a);
            for (int i = 0; i < 48; ++i) launch(i % n);
            hipEventRecord(b);
            CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            t.push_back(ms / 48 * 1000.0f);
        }
        std::sort(t.begin(), t.end());
        res[m++] = t[3];
    }
    unsigned long long * d; CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    checksumKernel<<<1024, 256>>>((const uint32_t *)o[0], (size_t)W * H, d);
    unsigned long long h = 0; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%-44s grid %5u  4 frames %6.2f us (%.3f)   %d frames %6.2f us (%.3f)   checksum %016llx\n", name, blocks, res[0], 22.8096 / res[0], NB, res[1], 22.8096 / res[1], h);
    return 0;
}
