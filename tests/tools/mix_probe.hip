// mix_probe.hip -- measurement only: what the chip sustains for a read:write byte mix when nothing can stay in the Infinity Cache.
// Every wave reads RD KiB from one linear stream and writes WR KiB (non-temporal) to another, all loads issued before the stores,
// like a wave of the tiled kernels.  Each launch moves ~1 GB through fresh addresses of 3 GB + 3 GB buffers.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_probe tests/tools/mix_probe.hip && /tmp/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int RD, int WR>
__global__ __launch_bounds__(256) void mixKernel(const u4 * __restrict__ src, u4 * __restrict__ dst, size_t srcOff, size_t dstOff)
{
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.y;
    const u4 * s = src + srcOff + wave * (size_t)(RD * 64) + threadIdx.x;
    u4 * d = dst + dstOff + wave * (size_t)(WR * 64) + threadIdx.x;
    u4 acc = { 0, 0, 0, 0 };
    u4 r[RD > 0 ? RD : 1];
#pragma unroll
    for (int k = 0; k < RD; ++k)
        r[k] = s[k * 64];
#pragma unroll
    for (int k = 0; k < RD; ++k)
        acc ^= r[k];
    if (WR == 0) {
        if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) // never true for the zero-filled source: keeps the loads alive
            dst[0] = acc;
    }
#pragma unroll
    for (int k = 0; k < WR; ++k)
        __builtin_nontemporal_store(acc + (unsigned)k, d + k * 64);
}

template <int RD, int WR>
static void run(const char * label, u4 * src, u4 * dst, size_t bufBytes)
{
    const size_t perWave = (size_t)(RD + WR) * 1024;
    const size_t waves = ((size_t)930 << 20) / perWave / 4 * 4;
    const size_t rdBytes = waves * RD * 1024, wrBytes = waves * WR * 1024;
    hipEvent_t t0, t1;
    hipEventCreate(&t0), hipEventCreate(&t1);
    const int iters = 24;
    size_t so = 0, dof = 0;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(t0);
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL((mixKernel<RD, WR>), dim3((unsigned)(waves / 4)), dim3(64, 4), 0, 0, src, dst, so / 16, dof / 16);
            so += rdBytes, dof += wrBytes;
            if (so + rdBytes > bufBytes)
                so = 0;
            if (dof + wrBytes > bufBytes)
                dof = 0;
        }
        hipEventRecord(t1);
        hipEventSynchronize(t1);
        float ms;
        hipEventElapsedTime(&ms, t0, t1);
        if (ms / iters < best)
            best = ms / iters;
    }
    printf("%-28s read %7.1f MB  write %7.1f MB  %8.2f us  %6.2f TB/s\n", label, rdBytes / 1e6, wrBytes / 1e6, best * 1e3, (rdBytes + wrBytes) / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t bufBytes = (size_t)3 << 30;
    u4 *src, *dst;
    if (hipMalloc(&src, bufBytes) != hipSuccess || hipMalloc(&dst, bufBytes) != hipSuccess)
        return 1;
    hipMemset(src, 0, bufBytes), hipMemset(dst, 0, bufBytes);
    hipDeviceSynchronize();
    run<8, 0>("read only", src, dst, bufBytes);
    run<0, 8>("write only", src, dst, bufBytes);
    run<3, 4>("3:4  (10-bit 420 -> RGBA8)", src, dst, bufBytes);
    run<6, 8>("6:8  same mix, longer waves", src, dst, bufBytes);
    run<3, 8>("3:8  (10-bit 420 -> RGBA16)", src, dst, bufBytes);
    run<3, 11>("1.5:5.5 ~ 8-bit 420 -> RGBA8", src, dst, bufBytes);
    run<8, 8>("1:1 copy", src, dst, bufBytes);
    run<4, 1>("4:1 (RGBA8 -> 4:2:0 ~ 4:1.5)", src, dst, bufBytes);
    return 0;
}
