// seqbench.hip -- sequence launches of the packed 16-bit cfg2 kernel (8-bit 4:2:0 bilinear -> RGBA8) over 12 cold 8K frames: which tile
// shape / order / load policy streams best when F frames share a launch?  (run on the GPU box; not a test, not shipped)
// Includes the product's tile_pk_impl.h directly.  -DSQB_NSW=1|2|4 (strips per wave) -DSQB_STREAM=true|false; argv: name, log2(waves side
// by side), tile rows per XCD chunk (0 = raster), frames per launch, tuning bits (TUNE_PRIVATE_HALO = 16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "tile_fx_impl.h"
#ifndef SQB_NSW
#define SQB_NSW 2
#endif
#ifndef SQB_STREAM
#define SQB_STREAM false
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
#ifndef SQB_W
#define SQB_W 7680
#define SQB_H 4320
#define SQB_NB 12
#endif
    const uint32_t W = SQB_W, H = SQB_H; // (-DSQB_W=3840 -DSQB_H=2160 -DSQB_NB=24: 4K frames, round 6)
    const int NB = SQB_NB;
    const char * name = argc > 1 ? argv[1] : "seqbench";
    const uint32_t wavesXLog2 = argc > 2 ? (uint32_t)atoi(argv[2]) : 0, chunkRows = argc > 3 ? (uint32_t)atoi(argv[3]) : 1;
    const uint32_t F = argc > 4 ? (uint32_t)atoi(argv[4]) : 4;
    const uint32_t tuning = argc > 5 ? (uint32_t)strtoul(argv[5], nullptr, 0) : (uint32_t)TUNE_DEFAULT;
    if (F == 0 || F > kSeqMaxFrames || NB % F) { printf("frames per launch must divide %d and be at most %u\n", NB, kSeqMaxFrames); return 1; }
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
        rgb.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_BILINEAR; rgb.avoidLibYUV = 0; rgb.maxThreads = 1;
        rgb.pixels = o[k]; rgb.rowBytes = W * 4;
        YuvToRgbPlan plan;
        if (makeYuvToRgbPlan(&img, &rgb, nullptr, 0, tuning, &plan) != AVIF_RESULT_OK || plan.arith != ARITH_LIBYUV) { printf("plan failed\n"); return 1; }
        args[k] = distillArgs(plan);
    }
    TileLaunch L; memset(&L, 0, sizeof(L));
    L.count = 1; L.maxW4 = W; L.maxH2 = H; L.pkStrips = SQB_NSW; L.wavesXLog2 = wavesXLog2; L.chunkRows = chunkRows;
    uint32_t nsw, blocks; PkGeom g;
    pkGeometry(L, W, H, &nsw, &g, &blocks, true);
    if (nsw != SQB_NSW) { printf("geometry chose %u strips\n", nsw); return 1; }
    const dim3 block(kLanesX, AVIFHIP_PK_WAVES), grid(blocks, 1, F);
    const uint32_t ldsBytes = 4u * (uint32_t)PkLds<SUB_420, true, 4, SQB_NSW, false>::kPlain;
    SeqFrames S[NB];
    for (int k = 0; k < NB; ++k)
        for (uint32_t f = 0; f < kSeqMaxFrames; ++f)
            seqSetFrame(S[k], f, args[(k * F + f) % NB]);
    auto launch = [&](int k) { hipLaunchKernelGGL((yuvToRgbPkKernel<SUB_420, true, 4, false, SQB_NSW, false, WIDE_NONE, SQB_STREAM>), grid, block, ldsBytes, 0, args[0], g, S[k % (NB / F)]); };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 1500; ++i) launch(i); // clock ramp
    std::vector<float> t;
    for (int rep = 0; rep < 7; ++rep) {
        for (int i = 0; i < 6; ++i) launch(i);
        hipEventRecord(a);
        for (int i = 0; i < 36; ++i) launch(i);
        hipEventRecord(b);
        CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms / 36 / F * 1000.0f);
    }
    std::sort(t.begin(), t.end());
    unsigned long long * d; CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    checksumKernel<<<1024, 256>>>((const uint32_t *)o[0], (size_t)W * H, d);
    unsigned long long h = 0; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%-52s grid %5u x %u  %d frames cycled: %6.2f us per frame (%.3f)   checksum %016llx\n", name, blocks, F, NB, t[3], (double)W * H * 5.5 / 8e6 / t[3], h);
    return 0;
}
