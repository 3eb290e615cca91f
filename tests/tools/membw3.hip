// membw3.hip -- access-pattern study for cfg3's bytes: four 16-bit planes (Y,U,V,A 7680x4320) in, RGBA16 out, no
// arithmetic.  Variants differ in how a wave's lanes cover pixels:
//   A  4 pixels per lane and row: 8-byte loads per plane, two 16-byte stores at off, off+16 (the tiled kernel today)
//   B  2 pixels per lane and row: 4-byte loads per plane, one 16-byte store (1 KiB contiguous per wave instruction)
//   C  4 pixels per lane, stores re-distributed inside the wave so that each store instruction is contiguous
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int W = 7680, H = 4320;

template <int VAR, bool NT, int ROWS>
__global__ __launch_bounds__(256) void k(const uint8_t * __restrict__ y, const uint8_t * __restrict__ u, const uint8_t * __restrict__ v, const uint8_t * __restrict__ a, uint8_t * __restrict__ out)
{
    constexpr int PX = (VAR == 1) ? 2 : 4;
    constexpr int BANDW = 64 * PX;
    const int bands = W / BANDW;
    const int band = blockIdx.x % bands, chunk = blockIdx.x / bands;
    const int X = band * BANDW + PX * threadIdx.x;
    const int Y0 = (chunk * 4 + threadIdx.y) * ROWS;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const size_t off = ((size_t)(Y0 + r) * W + X) * 2;
        if constexpr (VAR == 1) {
            const unsigned wy = *reinterpret_cast<const unsigned *>(y + off), wu = *reinterpret_cast<const unsigned *>(u + off);
            const unsigned wv = *reinterpret_cast<const unsigned *>(v + off), wa = *reinterpret_cast<const unsigned *>(a + off);
            u4 o = { (wy & 0xffff) | (wu << 16), (wv & 0xffff) | (wa << 16), (wy >> 16) | (wu & 0xffff0000u), (wv >> 16) | (wa & 0xffff0000u) };
            u4 * dst = reinterpret_cast<u4 *>(out + ((size_t)(Y0 + r) * W + X) * 8);
            if (NT) __builtin_nontemporal_store(o, dst); else *dst = o;
        } else {
            const u2 wy = *reinterpret_cast<const u2 *>(y + off), wu = *reinterpret_cast<const u2 *>(u + off);
            const u2 wv = *reinterpret_cast<const u2 *>(v + off), wa = *reinterpret_cast<const u2 *>(a + off);
            u4 o0 = { (wy.x & 0xffff) | (wu.x << 16), (wv.x & 0xffff) | (wa.x << 16), (wy.x >> 16) | (wu.x & 0xffff0000u), (wv.x >> 16) | (wa.x & 0xffff0000u) };
            u4 o1 = { (wy.y & 0xffff) | (wu.y << 16), (wv.y & 0xffff) | (wa.y << 16), (wy.y >> 16) | (wu.y & 0xffff0000u), (wv.y >> 16) | (wa.y & 0xffff0000u) };
            uint8_t * row = out + ((size_t)(Y0 + r) * W + band * BANDW) * 8;
            if constexpr (VAR == 0) {
                u4 * dst = reinterpret_cast<u4 *>(row + 32 * threadIdx.x);
                if (NT) { __builtin_nontemporal_store(o0, dst); __builtin_nontemporal_store(o1, dst + 1); } else { dst[0] = o0; dst[1] = o1; }
            } else {
                // store instruction 1 covers bytes [0, 1024) of the wave's row segment, instruction 2 bytes [1024, 2048):
                // lane l writes 16 bytes at 16*l; the data of that slot belongs to lane (l>>1) (+32 for instr 2), half (l&1)
                const int l = threadIdx.x;
                u4 s0, s1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned lo0 = __shfl(o0[c], l >> 1), hi0 = __shfl(o1[c], l >> 1);
                    const unsigned lo1 = __shfl(o0[c], 32 + (l >> 1)), hi1 = __shfl(o1[c], 32 + (l >> 1));
                    s0[c] = (l & 1) ? hi0 : lo0;
                    s1[c] = (l & 1) ? hi1 : lo1;
                }
                u4 * d0 = reinterpret_cast<u4 *>(row + 16 * l), * d1 = reinterpret_cast<u4 *>(row + 1024 + 16 * l);
                if (NT) { __builtin_nontemporal_store(s0, d0); __builtin_nontemporal_store(s1, d1); } else { *d0 = s0; *d1 = s1; }
            }
        }
    }
}

template <int VAR, bool NT, int ROWS>
static int run(const char * label, uint8_t * const * pl, uint8_t * out)
{
    constexpr int PX = (VAR == 1) ? 2 : 4;
    const int blocks = (W / (64 * PX)) * (H / (4 * ROWS));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<VAR, NT, ROWS>), dim3(blocks), dim3(64, 4), 0, 0, pl[0], pl[1], pl[2], pl[3], out);
        CK(hipEventRecord(t0, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<VAR, NT, ROWS>), dim3(blocks), dim3(64, 4), 0, 0, pl[0], pl[1], pl[2], pl[3], out);
        CK(hipEventRecord(t1, 0));
        CK(hipEventSynchronize(t1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        best = ms / 20 < best ? ms / 20 : best;
    }
    const double bytes = 16.0 * W * H;
    printf("%-28s %8.1f us  %7.0f GB/s  %5.1f%% of 8 TB/s\n", label, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 80);
    return 0;
}

int main()
{
    uint8_t * pl[4], * out;
    for (int p = 0; p < 4; ++p) { CK(hipMalloc(&pl[p], (size_t)W * H * 2)); CK(hipMemset(pl[p], 17 * (p + 1), (size_t)W * H * 2)); }
    CK(hipMalloc(&out, (size_t)W * H * 8));
    run<0, false, 2>("A 4px 2x16B plain r2", pl, out);
    run<0, true, 2>("A 4px 2x16B nt r2", pl, out);
    run<0, true, 4>("A 4px 2x16B nt r4", pl, out);
    run<1, false, 2>("B 2px 1x16B plain r2", pl, out);
    run<1, true, 2>("B 2px 1x16B nt r2", pl, out);
    run<1, true, 4>("B 2px 1x16B nt r4", pl, out);
    run<1, true, 8>("B 2px 1x16B nt r8", pl, out);
    run<2, false, 2>("C 4px shuffled plain r2", pl, out);
    run<2, true, 2>("C 4px shuffled nt r2", pl, out);
    run<2, true, 4>("C 4px shuffled nt r4", pl, out);
    return 0;
}
