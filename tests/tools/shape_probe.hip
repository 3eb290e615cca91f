// shape_probe.hip -- measurement only: how the SHAPE of a wave's accesses changes what the chip sustains when frames stream from HBM
// (nothing cached).  Moves exactly the bytes of a 10-bit 4:2:0 -> RGBA8 conversion of 64 tiles of 1920x1080 (luma 2 B/px, two chroma
// planes 2 B per 4 px, 4 B/px out), no arithmetic: a lane owns PPL consecutive pixels of RPW rows; a workgroup is WX waves side by side,
// 4 / WX stacked; SPLIT: the lane's PPL pixels are two runs of PPL/2, one wave-width apart (every instruction stays fully coalesced).
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/shape_probe tests/tools/shape_probe.hip && /tmp/shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Job
{
    const uint8_t *y, *u, *v;
    uint8_t * rgb;
};
constexpr uint32_t W = 1920, H = 1080, YP = 3840, CP = 2048, RP = 7680;

template <int N>
struct Vec;
template <>
struct Vec<4>
{
    typedef unsigned T;
};
template <>
struct Vec<8>
{
    typedef u2 T;
};
template <>
struct Vec<16>
{
    typedef u4 T;
};
__device__ inline unsigned fold(unsigned a) { return a; }
__device__ inline unsigned fold(u2 a) { return a.x ^ a.y; }
__device__ inline unsigned fold(u4 a) { return a.x ^ a.y ^ a.z ^ a.w; }

// RUN pixels in a row per lane and run (4 or 8), NRUN runs one wave-width apart, RPW rows per wave
template <int RUN, int NRUN, int RPW, int WX>
__global__ __launch_bounds__(256) void shapeKernel(const Job * jobs, uint32_t tilesX)
{
    const Job j = jobs[blockIdx.z];
    constexpr uint32_t bandW = 64 * RUN * NRUN;
    constexpr int WY = 4 / WX;
    const uint32_t trow = blockIdx.x / tilesX, tcol = blockIdx.x - trow * tilesX;
    const uint32_t wave = threadIdx.y, wx = wave % WX, wy = wave / WX;
    const uint32_t X0 = (tcol * WX + wx) * bandW, Y0 = (trow * WY + wy) * RPW;
    if (X0 >= W || Y0 >= H)
        return;
    typedef typename Vec<RUN * 2>::T YV; // luma: 2 bytes per pixel
    typedef typename Vec<RUN>::T CV;     // chroma: 2 bytes per 2 pixels of a row
    YV yv[RPW][NRUN];
    CV cu[RPW / 2][NRUN], cv[RPW / 2][NRUN];
    bool ok[NRUN];
#pragma unroll
    for (int n = 0; n < NRUN; ++n) {
        const uint32_t X = X0 + n * 64 * RUN + threadIdx.x * RUN;
        ok[n] = X < W;
        const uint32_t Xc = ok[n] ? X : 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const uint32_t Y = Y0 + r < H ? Y0 + r : H - 1;
            yv[r][n] = *reinterpret_cast<const YV *>(j.y + (size_t)Y * YP + Xc * 2);
            if (!(r & 1)) {
                cu[r / 2][n] = *reinterpret_cast<const CV *>(j.u + (size_t)(Y >> 1) * CP + Xc);
                cv[r / 2][n] = *reinterpret_cast<const CV *>(j.v + (size_t)(Y >> 1) * CP + Xc);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NRUN; ++n) {
        const uint32_t X = X0 + n * 64 * RUN + threadIdx.x * RUN;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (Y0 + r >= H || !ok[n])
                continue;
            const unsigned c = fold(yv[r][n]) ^ fold(cu[r / 2][n]) ^ fold(cv[r / 2][n]);
#pragma unroll
            for (int q = 0; q < RUN / 4; ++q)
                __builtin_nontemporal_store((u4) { c, c + 1, c + 2, c + 3 + q }, reinterpret_cast<u4 *>(j.rgb + (size_t)(Y0 + r) * RP + (size_t)(X + 4 * q) * 4));
        }
    }
}

static Job * gJobs;
template <int RUN, int NRUN, int RPW, int WX>
static void run(const char * label)
{
    constexpr uint32_t bandW = 64 * RUN * NRUN;
    const uint32_t tilesX = (W + bandW * WX - 1) / (bandW * WX), tilesY = (H + (4 / WX) * RPW - 1) / ((4 / WX) * RPW);
    hipEvent_t t0, t1;
    hipEventCreate(&t0), hipEventCreate(&t1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(t0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((shapeKernel<RUN, NRUN, RPW, WX>), dim3(tilesX * tilesY, 1, 64), dim3(64, 4), 0, 0, gJobs, tilesX);
        hipEventRecord(t1);
        hipEventSynchronize(t1);
        float ms;
        hipEventElapsedTime(&ms, t0, t1);
        if (ms / 10 < best)
            best = ms / 10;
    }
    const double bytes = 64.0 * W * H * 7.0;
    printf("%-46s wg %4u x %2u px  %8.2f us  %5.2f TB/s\n", label, bandW * WX, (4 / WX) * RPW, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    Job h[64];
    for (int k = 0; k < 64; ++k) {
        uint8_t *y, *u, *v, *rgb;
        if (hipMalloc(&y, (size_t)YP * H) || hipMalloc(&u, (size_t)CP * H / 2) || hipMalloc(&v, (size_t)CP * H / 2) || hipMalloc(&rgb, (size_t)RP * H))
            return 1;
        hipMemset(y, k, (size_t)YP * H), hipMemset(u, k, (size_t)CP * H / 2), hipMemset(v, k, (size_t)CP * H / 2);
        h[k] = Job { y, u, v, rgb };
    }
    hipMalloc(&gJobs, sizeof(h));
    hipMemcpy(gJobs, h, sizeof(h), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    run<4, 1, 8, 1>("4 px/lane, 8 rows, waves stacked (as shipped)");
    run<4, 1, 8, 2>("4 px/lane, 8 rows, 2 side by side");
    run<4, 1, 8, 4>("4 px/lane, 8 rows, 4 side by side");
    run<4, 1, 4, 1>("4 px/lane, 4 rows, stacked");
    run<4, 1, 4, 2>("4 px/lane, 4 rows, 2 side by side");
    run<4, 1, 4, 4>("4 px/lane, 4 rows, 4 side by side");
    run<4, 1, 2, 4>("4 px/lane, 2 rows, 4 side by side");
    run<8, 1, 4, 1>("8 px/lane contiguous, 4 rows, stacked");
    run<8, 1, 4, 2>("8 px/lane contiguous, 4 rows, 2 side by side");
    run<8, 1, 4, 4>("8 px/lane contiguous, 4 rows, 4 side by side");
    run<8, 1, 2, 4>("8 px/lane contiguous, 2 rows, 4 side by side");
    run<8, 1, 8, 1>("8 px/lane contiguous, 8 rows, stacked");
    run<4, 2, 4, 1>("2 x 4 px/lane split, 4 rows, stacked");
    run<4, 2, 4, 2>("2 x 4 px/lane split, 4 rows, 2 side by side");
    run<4, 2, 4, 4>("2 x 4 px/lane split, 4 rows, 4 side by side");
    run<4, 2, 2, 4>("2 x 4 px/lane split, 2 rows, 4 side by side");
    run<4, 4, 2, 2>("4 x 4 px/lane split, 2 rows, 2 side by side");
    run<4, 4, 2, 1>("4 x 4 px/lane split, 2 rows, stacked");
    return 0;
}
