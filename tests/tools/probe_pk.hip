// probe_pk.hip -- semantics of the packed 16-bit instructions the integer tiled kernels rely on (run on the GPU box):
//   v_pk_mad_i16 ... clamp   : saturates the EXACT a*b+c to [-32768, 32767] per half?
//   v_sat_pk_u8_i16          : {sat_u8(lo), sat_u8(hi)} in bits 0..15 (bits 16..31 of the destination are not written)
//   v_perm_b32 selectors 8..11 (sign of bytes 1,3 of S1 / S0), 12 (0x00), 13 (0xff)
//   v_pk_ashrrev_i16, v_pk_add_u16 wrap
// Prints the number of mismatches per check (0 expected).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned pkMadClamp(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pkMad(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned satPk(unsigned a)
{
    unsigned d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ unsigned pkAshr(unsigned a, unsigned s)
{
    unsigned d;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "v"(s), "v"(a));
    return d;
}
__device__ int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ void probe(unsigned long long * bad)
{
    // (1) mad clamp: a in [-128,127] (lo) / a+3 (hi), b in a list, c over all 65536 values
    const int coeffs[] = { 128, 127, 120, 119, 115, 113, 107, 102, 101, 94, 90, -52, -46, -42, -37, -34, -30, -25, -22, -14, -12, -11, 1, -1, 300, -300 };
    const int nc = sizeof(coeffs) / sizeof(coeffs[0]);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    unsigned long long e1 = 0, e1b = 0, e2 = 0, e3 = 0, e4 = 0;
    for (unsigned idx = tid; idx < 256u * 65536u; idx += nthreads) {
        const int a = (int)(idx >> 16) - 128, c = (int)(int16_t)(idx & 0xffff);
        const int a2 = (a + 3 > 127) ? a - 250 : a + 3, c2 = (int)(int16_t)((idx * 7919u) & 0xffff);
        for (int k = 0; k < nc; ++k) {
            const int b = coeffs[k];
            const unsigned A = ((unsigned)a & 0xffffu) | ((unsigned)a2 << 16), B = ((unsigned)b & 0xffffu) | ((unsigned)b << 16), C = ((unsigned)c & 0xffffu) | ((unsigned)c2 << 16);
            const unsigned d = pkMadClamp(A, B, C);
            const int lo = sat16(a * b + c), hi = sat16(a2 * b + c2);
            if ((int)(int16_t)(d & 0xffff) != lo || (int)(int16_t)(d >> 16) != hi) ++e1;
            const unsigned w = pkMad(A, B, C);
            if ((w & 0xffff) != ((unsigned)(a * b + c) & 0xffff) || (w >> 16) != ((unsigned)(a2 * b + c2) & 0xffff)) ++e1b;
        }
    }
    // (2) sat_pk over all pairs (lo = every i16, hi = a hash of it), (3) ashr by 6
    for (unsigned idx = tid; idx < 65536u * 16u; idx += nthreads) {
        const unsigned lo = idx & 0xffff, hi = (idx * 40503u + (idx >> 16) * 977u) & 0xffff;
        const unsigned d = satPk(lo | (hi << 16)) & 0xffffu; // the instruction writes bits 0..15 only
        const unsigned want = (unsigned)sat8((int16_t)lo) | ((unsigned)sat8((int16_t)hi) << 8);
        if (d != want) ++e2;
        const unsigned s = pkAshr(lo | (hi << 16), 0x00060006u); // per-half shift amounts
        if ((int16_t)(s & 0xffff) != (int16_t)((int16_t)lo >> 6) || (int16_t)(s >> 16) != (int16_t)((int16_t)hi >> 6)) ++e3;
    }
    // (4) perm selectors
    for (unsigned idx = tid; idx < (1u << 22); idx += nthreads) {
        const unsigned s0 = idx * 2654435761u, s1 = (idx ^ 0x5bd1e995u) * 40503u + 12345u;
        const unsigned d = __builtin_amdgcn_perm(s0, s1, 0x0a050801u); // [S1.b1, sign(S1.b1), S0.b1, sign(S0.b1)]
        const unsigned d2 = __builtin_amdgcn_perm(s0, s1, 0x0b070903u); // [S1.b3, sign(S1.b3), S0.b3, sign(S0.b3)]
        const unsigned d3 = __builtin_amdgcn_perm(s0, s1, 0x0d0c0400u); // [S1.b0, S0.b0, 0, 0xff]
        auto se = [](unsigned b) { return (unsigned)(int)(int8_t)b & 0xffffu; };
        const unsigned w = se((s1 >> 8) & 0xff) | (se((s0 >> 8) & 0xff) << 16);
        const unsigned w2 = se(s1 >> 24) | (se(s0 >> 24) << 16);
        const unsigned w3 = (s1 & 0xff) | ((s0 & 0xff) << 8) | 0xff000000u;
        if (d != w || d2 != w2 || d3 != w3) ++e4;
    }
    atomicAdd(&bad[0], e1); atomicAdd(&bad[1], e1b); atomicAdd(&bad[2], e2); atomicAdd(&bad[3], e3); atomicAdd(&bad[4], e4);
}
int main()
{
    unsigned long long * bad, h[5];
    CK(hipMalloc(&bad, sizeof(h)));
    CK(hipMemset(bad, 0, sizeof(h)));
    probe<<<2048, 256>>>(bad);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("v_pk_mad_i16 clamp vs exact saturation: %llu mismatches\nv_pk_mad_i16 (wrap): %llu\nv_sat_pk_u8_i16: %llu\nv_pk_ashrrev_i16 6: %llu\nv_perm_b32 sign/const selectors: %llu\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
