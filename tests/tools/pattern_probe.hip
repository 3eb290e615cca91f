// pattern_probe.hip -- which access pattern moves cfg2's bytes (Y + U/4 + V/4 in, RGBA8 out, no arithmetic) fastest when
// NOTHING is served by the 256 MB Infinity Cache (16 frames cycled, 2.9 GB) and when the inputs are (4 frames)?
// Variants: rows per wave, tile order, store / load cache policy, persistent grid.  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int W = 7680, H = 4320;
__device__ __forceinline__ uint32_t remap(uint32_t b, uint32_t n) { const uint32_t per = n >> 3, rem = n & 7, x = b & 7, s = b >> 3; return x * per + (x < rem ? x : rem) + s; }

enum { ORD_LINEAR = 0, ORD_BANDED = 1, ORD_COLUMN = 2 };
enum { ST_PLAIN = 0, ST_NT = 1, ST_SC1 = 2, ST_SC0SC1 = 3, ST_NTSC1 = 4 };

template <int ST>
__device__ __forceinline__ void store16(u4 * dst, u4 o)
{
    if (ST == ST_NT) __builtin_nontemporal_store(o, dst);
    else if (ST == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(o) : "memory");
    else if (ST == ST_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(o) : "memory");
    else if (ST == ST_NTSC1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(dst), "v"(o) : "memory");
    else *dst = o;
}
template <bool NTL, typename T>
__device__ __forceinline__ T ld(const T * p) { return NTL ? __builtin_nontemporal_load(p) : *p; }

// wave = 256 px wide x RPL rows per step; WAVES_X waves side by side in a workgroup (4 / WAVES_X stacked); a workgroup walks
// STEPS steps downwards (persistent-ish: fewer, longer workgroups)
template <int RPL, int WAVES_X, int ORD, int ST, bool NTL, int STEPS>
__global__ __launch_bounds__(256) void tileCopy(const uint8_t * __restrict__ y, const uint8_t * __restrict__ u, const uint8_t * __restrict__ v, uint8_t * __restrict__ rgba)
{
    constexpr int WAVES_Y = 4 / WAVES_X;
    constexpr int TW = 256 * WAVES_X, TH = RPL * WAVES_Y * STEPS;
    const int tilesX = W / TW, tilesY = (H + TH - 1) / TH;
    uint32_t tile = blockIdx.x;
    if (ORD == ORD_BANDED) tile = remap(tile, gridDim.x);
    int trow, tcol;
    if (ORD == ORD_COLUMN) { tcol = tile / tilesY; trow = tile - tcol * tilesY; } else { trow = tile / tilesX; tcol = tile - trow * tilesX; }
    const int wave = threadIdx.y, wx = wave % WAVES_X, wyv = wave / WAVES_X;
    const int X = tcol * TW + wx * 256 + 4 * threadIdx.x;
    for (int s = 0; s < STEPS; ++s) {
        const int Y0 = trow * TH + (s * WAVES_Y + wyv) * RPL;
        if (Y0 >= H) return;
        unsigned wy[RPL], cu[RPL], cv[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            wy[r] = ld<NTL>(reinterpret_cast<const unsigned *>(y + (size_t)(Y0 + r) * W + X));
            cu[r] = cv[r] = 0;
            if (!(r & 1)) {
                cu[r] = ld<NTL>(reinterpret_cast<const uint16_t *>(u + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1)));
                cv[r] = ld<NTL>(reinterpret_cast<const uint16_t *>(v + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1)));
            }
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const unsigned c = cu[r & ~1] | (cv[r & ~1] << 16);
            u4 o;
            o.x = (wy[r] & 0xff) | (c << 8);
            o.y = ((wy[r] >> 8) & 0xff) | (c << 8);
            o.z = ((wy[r] >> 16) & 0xff) | (c & 0xffffff00u);
            o.w = (wy[r] >> 24) | (c & 0xffffff00u);
            store16<ST>(reinterpret_cast<u4 *>(rgba + ((size_t)(Y0 + r) * W + X) * 4), o);
        }
    }
}

struct Frame { uint8_t *y, *u, *v, *o; };
template <int RPL, int WAVES_X, int ORD, int ST, bool NTL, int STEPS>
static void run(const char * name, const std::vector<Frame> & all)
{
    constexpr int TW = 256 * WAVES_X, TH = RPL * (4 / WAVES_X) * STEPS;
    const unsigned blocks = (W / TW) * ((H + TH - 1) / TH);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float res[2];
    int k = 0;
    for (int n : { 4, 16 }) {
        std::vector<float> t;
        for (int rep = 0; rep < 5; ++rep) {
            for (int i = 0; i < 8; ++i) tileCopy<RPL, WAVES_X, ORD, ST, NTL, STEPS><<<blocks, dim3(64, 4)>>>(all[i % n].y, all[i % n].u, all[i % n].v, all[i % n].o);
            hipEventRecord(a);
            for (int i = 0; i < 48; ++i) tileCopy<RPL, WAVES_X, ORD, ST, NTL, STEPS><<<blocks, dim3(64, 4)>>>(all[i % n].y, all[i % n].u, all[i % n].v, all[i % n].o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            t.push_back(ms / 48 * 1000.0f);
        }
        std::sort(t.begin(), t.end());
        res[k++] = t[2];
    }
    printf("%-52s wg %5u  tile %4dx%-3d | 4 frames %6.1f us (%.3f) | 16 frames %6.1f us (%.3f)\n", name, blocks, TW, TH, res[0], 22.8096 / res[0], res[1], 22.8096 / res[1]);
    fflush(stdout);
}
int main()
{
    const size_t ySize = (size_t)W * H, cSize = ySize / 4, oSize = ySize * 4;
    const int N = 16;
    std::vector<Frame> f(N);
    for (int k = 0; k < N; ++k) {
        CK(hipMalloc(&f[k].y, ySize)); CK(hipMalloc(&f[k].u, cSize)); CK(hipMalloc(&f[k].v, cSize)); CK(hipMalloc(&f[k].o, oSize));
        CK(hipMemset(f[k].y, 0x40 + k, ySize)); CK(hipMemset(f[k].u, 0x80, cSize)); CK(hipMemset(f[k].v, 0x81, cSize)); CK(hipMemset(f[k].o, 0, oSize));
    }
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 3; ++i) run<4, 1, ORD_BANDED, ST_NT, false, 1>("(clock ramp)", f);
    printf("---- (fraction of 8 TB/s in parentheses)\n");
    run<4, 1, ORD_BANDED, ST_NT, false, 1>("256x16 banded nt", f);
    run<4, 1, ORD_LINEAR, ST_NT, false, 1>("256x16 linear nt", f);
    run<4, 1, ORD_COLUMN, ST_NT, false, 1>("256x16 column nt", f);
    run<4, 1, ORD_BANDED, ST_PLAIN, false, 1>("256x16 banded plain", f);
    run<4, 1, ORD_BANDED, ST_SC1, false, 1>("256x16 banded sc1", f);
    run<4, 1, ORD_BANDED, ST_SC0SC1, false, 1>("256x16 banded sc0 sc1", f);
    run<4, 1, ORD_BANDED, ST_NTSC1, false, 1>("256x16 banded sc1 nt", f);
    run<4, 1, ORD_BANDED, ST_NT, true, 1>("256x16 banded nt stores + nt loads", f);
    run<4, 1, ORD_LINEAR, ST_NT, true, 1>("256x16 linear nt stores + nt loads", f);
    run<2, 1, ORD_BANDED, ST_NT, false, 1>("256x8 banded nt", f);
    run<8, 1, ORD_BANDED, ST_NT, false, 1>("256x32 banded nt", f);
    run<2, 4, ORD_BANDED, ST_NT, false, 1>("1024x2 banded nt", f);
    run<2, 4, ORD_LINEAR, ST_NT, false, 1>("1024x2 linear nt", f);
    run<4, 4, ORD_LINEAR, ST_NT, false, 1>("1024x4 linear nt", f);
    run<4, 4, ORD_BANDED, ST_NT, false, 1>("1024x4 banded nt", f);
    run<8, 4, ORD_LINEAR, ST_NT, false, 1>("1024x8 linear nt", f);
    run<4, 2, ORD_LINEAR, ST_NT, false, 1>("512x8 linear nt", f);
    run<4, 1, ORD_BANDED, ST_NT, false, 2>("256x16 x2 steps banded nt", f);
    run<4, 1, ORD_BANDED, ST_NT, false, 4>("256x16 x4 steps banded nt", f);
    run<4, 1, ORD_LINEAR, ST_NT, false, 4>("256x16 x4 steps linear nt", f);
    run<4, 4, ORD_LINEAR, ST_NT, false, 4>("1024x4 x4 steps linear nt", f);
    run<4, 4, ORD_LINEAR, ST_NT, false, 8>("1024x4 x8 steps linear nt (1020 wg)", f);
    run<4, 4, ORD_BANDED, ST_NT, false, 8>("1024x4 x8 steps banded nt (1020 wg)", f);
    run<4, 4, ORD_LINEAR, ST_NT, false, 16>("1024x4 x16 steps linear nt (510 wg)", f);
    run<2, 4, ORD_LINEAR, ST_NT, false, 16>("1024x2 x16 steps linear nt (1020 wg)", f);
    run<4, 4, ORD_COLUMN, ST_NT, false, 8>("1024x4 x8 steps column nt", f);
    return 0;
}
