// valu_rate.hip -- issue rate of the VALU instructions the reformat kernels lean on (run on the GPU box).
// Each kernel runs ITER x 32 independent instructions per wave; reported: cycles per wave-instruction per SIMD
// with all SIMDs busy (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(float * out, float a, float b)
{
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = a * (threadIdx.x + i);
    f2 * r2 = reinterpret_cast<f2 *>(r);
    unsigned u[8];
    double dd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 77 + i, dd[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (OP == 0) { // v_fma_f32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            } else if constexpr (OP == 1) { // v_pk_fma_f32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r2[i]) : "v"(r2[(i + 1) & 7]));
            } else if constexpr (OP == 2) { // v_pk_mul_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r2[i]) : "v"(r2[(i + 1) & 7]));
            } else if constexpr (OP == 3) { // v_pk_add_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r2[i]) : "v"(r2[(i + 1) & 7]));
            } else if constexpr (OP == 4) { // v_mul_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            } else if constexpr (OP == 5) { // v_add_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            } else if constexpr (OP == 6) { // v_cvt_pk_u8_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(r[i]));
            } else if constexpr (OP == 7) { // v_cvt_f32_ubyte1
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(u[i]));
            } else if constexpr (OP == 8) { // v_floor_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
            } else if constexpr (OP == 9) { // v_rcp_f32 (transcendental rate reference)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            } else if constexpr (OP == 10) { // v_mov_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 7]));
            } else if constexpr (OP == 12) { // v_med3_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(r[i]));
            } else if constexpr (OP == 13) { // v_cvt_u32_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(r[i]));
            } else if constexpr (OP == 14) { // v_perm_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            } else if constexpr (OP == 15) { // v_lshl_or_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            } else if constexpr (OP == 16) { // v_mul_f64 (the gain-map kernel's primaries matrices)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) & 7]));
            } else if constexpr (OP == 17) { // v_add_f64
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) & 7]));
            } else if constexpr (OP == 18) { // v_fma_f64
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) & 7]));
            } else if constexpr (OP == 19) { // v_cvt_f64_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[i]) : "v"(r[i]));
            } else if constexpr (OP == 20) { // v_cvt_f32_f64
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(dd[i]));
            } else if constexpr (OP == 21) { // v_med3_i32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            } else if constexpr (OP == 22) { // v_lshlrev_b32_sdwa (byte operand)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            } else if constexpr (OP == 23) { // v_cmp_gt_u32 + v_addc_co_u32 (the pair the locator ends with)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            } else if constexpr (OP == 24) { // v_lshlrev_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
            } else if constexpr (OP == 25) { // v_max3_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)u[i] + (float)dd[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
static int run(const char * name, float * d, int cus, double ghz)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = cus * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1.0001f, 0.5f);
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 1.0001f, 0.5f);
    hipEventRecord(b);
    CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instrPerSimd = 8.0 * ITER * 32; // 8 waves per SIMD
    printf("%-28s %8.3f ms   %5.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / instrPerSimd, ghz);
    return 0;
}
int main()
{
    float * d; CK(hipMalloc(&d, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate / 1e6;
    printf("CUs %d clock %.2f GHz\n", cus, ghz);
    run<0>("v_fma_f32", d, cus, ghz); run<1>("v_pk_fma_f32", d, cus, ghz); run<2>("v_pk_mul_f32", d, cus, ghz); run<3>("v_pk_add_f32", d, cus, ghz);
    run<4>("v_mul_f32", d, cus, ghz); run<5>("v_add_f32", d, cus, ghz); run<6>("v_cvt_pk_u8_f32", d, cus, ghz); run<7>("v_cvt_f32_ubyte1", d, cus, ghz);
    run<8>("v_floor_f32", d, cus, ghz); run<9>("v_rcp_f32", d, cus, ghz); run<10>("v_mov_b32", d, cus, ghz); run<12>("v_med3_f32", d, cus, ghz);
    run<13>("v_cvt_u32_f32", d, cus, ghz); run<14>("v_perm_b32", d, cus, ghz); run<15>("v_lshl_or_b32", d, cus, ghz);
    run<16>("v_mul_f64", d, cus, ghz); run<17>("v_add_f64", d, cus, ghz); run<18>("v_fma_f64", d, cus, ghz); run<19>("v_cvt_f64_f32", d, cus, ghz);
    run<20>("v_cvt_f32_f64", d, cus, ghz); run<21>("v_med3_i32", d, cus, ghz); run<22>("v_lshlrev_b32_sdwa", d, cus, ghz);
    run<23>("v_cmp_gt_u32+v_addc_co_u32", d, cus, ghz); run<24>("v_lshlrev_b32", d, cus, ghz); run<25>("v_max3_f32", d, cus, ghz);
    return 0;
}
