// probe_cvt.hip -- empirical semantics of v_cvt_pk_u8_f32 on gfx950 (run on the GPU box):
// does it truncate, round to nearest even, and does it saturate on both sides?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void probe(unsigned long long * bad /*[4]*/, uint32_t lo, uint32_t hi)
{
    const uint64_t n = (uint64_t)hi - lo;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = lo + (uint32_t)k;
        const float x = __uint_as_float(bits);
        const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0, 0) & 0xff;
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
    // positive floats 0 .. 1024.0 (bits 0 .. 0x44800000) and negative -1024 .. -0
    const uint32_t ranges[2][2] = { { 0x00000000u, 0x44800000u }, { 0x80000000u, 0xC4800000u } };
    for (int k = 0; k < 2; ++k) {
        hipMemset(d, 0, sizeof(h));
        probe<<<4096, 256>>>(d, ranges[k][0], ranges[k][1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("range %d: mismatches vs truncate+saturate: %llu, vs rne+saturate: %llu\n", k, h[0], h[1]);
    }
    return 0;
}
