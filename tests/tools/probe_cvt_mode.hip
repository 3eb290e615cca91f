// probe_cvt_mode.hip -- does v_cvt_pk_u8_f32 on gfx950 follow MODE.FP_ROUND?  (run on the GPU box)
// Sets the fp32 rounding mode to round-toward-zero around the conversion and compares with truncate+saturate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void probe(unsigned long long * bad, uint32_t lo, uint32_t hi)
{
    const uint64_t n = (uint64_t)hi - lo;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = lo + (uint32_t)k;
        const float x = __uint_as_float(bits);
        unsigned got;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 2\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_nop 2\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 2"
                     : "=v"(got)
                     : "v"(x));
        got &= 0xff;
        const float t = truncf(x);
        const unsigned wantTrunc = (t < 0.0f) ? 0u : (t > 255.0f ? 255u : (unsigned)t);
        const float r = rintf(x);
        const unsigned wantRne = (r < 0.0f) ? 0u : (r > 255.0f ? 255u : (unsigned)r);
        if (got != wantTrunc) atomicAdd(&bad[0], 1ull);
        if (got != wantRne) atomicAdd(&bad[1], 1ull);
    }
}
int main()
{
    unsigned long long * d; unsigned long long h[4] = {0,0,0,0};
    hipMalloc(&d, sizeof(h));
    const uint32_t ranges[2][2] = { { 0x00000000u, 0x44800000u }, { 0x80000000u, 0xC4800000u } };
    for (int k = 0; k < 2; ++k) {
        hipMemset(d, 0, sizeof(h));
        probe<<<4096, 256>>>(d, ranges[k][0], ranges[k][1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("RTZ mode, range %d: mismatches vs truncate+saturate: %llu, vs rne+saturate: %llu\n", k, h[0], h[1]);
    }
    return 0;
}
