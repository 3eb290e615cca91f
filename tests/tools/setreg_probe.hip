// setreg_probe.hip -- what the rounding-mode switch around v_cvt_pk_u8_f32 blocks costs (run on the GPU box).
// Kernel variants convert the same number of floats; they differ in how many conversions share one s_setreg pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2000;

template <int PER> // conversions per s_setreg pair: 0 = no mode switch at all
__global__ __launch_bounds__(256) void k(unsigned * out, float a)
{
    float f[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) f[i] = a * (float)(threadIdx.x + i);
    unsigned w[6] = { 0, 0, 0, 0, 0, 0 };
    for (int it = 0; it < ITER; ++it) {
        if constexpr (PER == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(w[i / 4]) : "v"(f[i]), "n"(i & 3));
        } else {
#pragma unroll
            for (int g = 0; g < 24 / PER; ++g) {
                asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
#pragma unroll
                for (int i = g * PER; i < (g + 1) * PER; ++i) asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(w[i / 4]) : "v"(f[i]), "n"(i & 3));
                asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
            }
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) f[i] += 1.0f; // some ordinary fp32 work between the blocks, as in the kernels
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += w[i];
    if (s == 0x12345678u) out[0] = s;
}

template <int PER>
static int run(const char * name, unsigned * d, int cus, double ghz)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = cus * 8;
    k<PER><<<blocks, 256>>>(d, 1.0001f);
    hipEventRecord(a);
    k<PER><<<blocks, 256>>>(d, 1.0001f);
    hipEventRecord(b);
    CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    const double itersPerSimd = 8.0 * ITER; // 8 waves per SIMD
    printf("%-44s %8.3f ms   %7.1f cycles per iteration per SIMD (24 conversions + 24 adds)\n", name, ms, ms * 1e-3 * ghz * 1e9 / itersPerSimd);
    return 0;
}
int main()
{
    unsigned * d; CK(hipMalloc(&d, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate / 1e6;
    run<0>("no rounding-mode switch", d, cus, ghz);
    run<24>("1 s_setreg pair per 24 conversions", d, cus, ghz);
    run<12>("1 pair per 12 (packRgba8Row)", d, cus, ghz);
    run<4>("1 pair per 4 (packU8x4)", d, cus, ghz);
    run<0>("no rounding-mode switch (again)", d, cus, ghz);
    return 0;
}
