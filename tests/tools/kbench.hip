// kbench.hip -- kernel-structure experiments for the cfg2 kernel (run on the GPU box; not a test, not shipped).
// Includes the product's tile_impl.h directly, instantiates ONE kernel, times it with HIP events (same frame and
// cycling over 4 frames), prints a checksum of the output so variants can be compared with each other.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -I include -I libavif_amd/csrc [-DKB_NS=2] [-DAVIFHIP_ABLATE_...] tests/tools/kbench.hip libavif_amd/csrc/plan.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "tile_impl.h"
#ifndef KB_NS
#define KB_NS 2
#endif
#ifndef KB_BIL
#define KB_BIL true
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
    const int NB = 4;
    uint32_t tuning = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 0) : TUNE_DEFAULT;
    uint8_t *y[NB], *u[NB], *v[NB], *o[NB];
    std::vector<uint8_t> host((size_t)W * H);
    uint32_t x = 0x12345678u;
    for (int k = 0; k < NB; ++k) {
        CK(hipMalloc(&y[k], (size_t)W * H)); CK(hipMalloc(&u[k], (size_t)W * H / 4)); CK(hipMalloc(&v[k], (size_t)W * H / 4)); CK(hipMalloc(&o[k], (size_t)W * H * 4));
        for (size_t i = 0; i < host.size(); ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; host[i] = 16 + x % 220; }
        CK(hipMemcpy(y[k], host.data(), (size_t)W * H, hipMemcpyHostToDevice));
        CK(hipMemcpy(u[k], host.data() + 1000, (size_t)W * H / 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(v[k], host.data() + 7777, (size_t)W * H / 4, hipMemcpyHostToDevice));
        CK(hipMemset(o[k], 0, (size_t)W * H * 4));
    }
    YuvToRgbPlan plans[NB];
    for (int k = 0; k < NB; ++k) {
        avifImage img; memset(&img, 0, sizeof(img));
        img.width = W; img.height = H; img.depth = 8; img.yuvFormat = AVIF_PIXEL_FORMAT_YUV420; img.yuvRange = AVIF_RANGE_LIMITED;
        img.matrixCoefficients = 1;
        img.yuvPlanes[0] = y[k]; img.yuvPlanes[1] = u[k]; img.yuvPlanes[2] = v[k];
        img.yuvRowBytes[0] = W; img.yuvRowBytes[1] = W / 2; img.yuvRowBytes[2] = W / 2;
        avifRGBImage rgb; memset(&rgb, 0, sizeof(rgb));
        rgb.width = W; rgb.height = H; rgb.depth = 8; rgb.format = AVIF_RGB_FORMAT_RGBA;
        rgb.chromaUpsampling = KB_BIL ? AVIF_CHROMA_UPSAMPLING_BILINEAR : AVIF_CHROMA_UPSAMPLING_NEAREST; rgb.avoidLibYUV = 1; rgb.maxThreads = 1;
        rgb.pixels = o[k]; rgb.rowBytes = W * 4;
        if (makeYuvToRgbPlan(&img, &rgb, nullptr, 1, tuning, &plans[k]) != AVIF_RESULT_OK) { printf("plan failed\n"); return 1; }
    }
    const dim3 block(kLanesX, kWavesPerBlock);
    const uint32_t run = argc > 3 ? (uint32_t)atoi(argv[3]) : 4;
    const uint32_t tilesY = (H + 8 * KB_NS - 1) / (8 * KB_NS);
    const dim3 grid(((W + 255) / 256) * ((tilesY + run - 1) / run));
    TileArgs args[NB];
    for (int k = 0; k < NB; ++k) args[k] = distillArgs(plans[k]);
    auto launch = [&](int k) { hipLaunchKernelGGL((yuvToRgbTileKernel<uint8_t, SUB_420, KB_BIL, uint8_t, 4, false, false, KB_NS>), grid, block, 0, 0, args[k], run); };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best[2] = { 1e9f, 1e9f };
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 5; ++rep) {
            for (int i = 0; i < 4; ++i) launch(mode ? i % NB : 0);
            hipEventRecord(a);
            for (int i = 0; i < 40; ++i) launch(mode ? i % NB : 0);
            hipEventRecord(b);
            CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms / 40 < best[mode]) best[mode] = ms / 40;
        }
    unsigned long long * d; CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    checksumKernel<<<1024, 256>>>((const uint32_t *)o[0], (size_t)W * H, d);
    unsigned long long h = 0; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    const double bytes = 5.5 * W * H;
    printf("%-40s same-frame %6.1f us (%4.1f%%)   cycling %6.1f us (%4.1f%%)   checksum %016llx\n", argc > 2 ? argv[2] : "kbench", best[0] * 1e3,
           bytes / (best[0] * 1e-3) / 8e12 * 100, best[1] * 1e3, bytes / (best[1] * 1e-3) / 8e12 * 100, h);
    return 0;
}
