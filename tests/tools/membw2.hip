// membw2.hip -- search for the access pattern that moves cfg2's bytes (33 MB Y + 2 x 8.3 MB chroma in, 133 MB RGBA
// out, no arithmetic) fastest when frames stream from HBM (4 frames cycled, working set 730 MB > Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int W = 7680, H = 4320;

__device__ __forceinline__ uint32_t remap(uint32_t b, uint32_t n) { const uint32_t per = n >> 3, rem = n & 7, x = b & 7, s = b >> 3; return x * per + (x < rem ? x : rem) + s; }

// MODE bit0: XCD-banded order; bit1: nt stores; bit2: nt loads; bit3: sc1 (write-through) stores; bit4: column-major tile order
template <int TW_WAVES, int ROWS_PER_LANE, int MODE>
__global__ __launch_bounds__(256) void tileCopy(const uint8_t * __restrict__ y, const uint8_t * __restrict__ u, const uint8_t * __restrict__ v, uint8_t * __restrict__ rgba)
{
    // block = 4 waves arranged TW_WAVES wide x (4/TW_WAVES) high; each wave covers 256 px x (2*ROWS_PER_LANE/2...) rows
    constexpr int WAVES_Y = 4 / TW_WAVES;
    constexpr int TW = 256 * TW_WAVES, TH = ROWS_PER_LANE * WAVES_Y;
    const int tilesX = W / TW, tilesY = H / TH;
    uint32_t tile = blockIdx.x;
    if (MODE & 1) tile = remap(tile, gridDim.x);
    int trow, tcol;
    if (MODE & 16) { tcol = tile / tilesY; trow = tile - tcol * tilesY; } else { trow = tile / tilesX; tcol = tile - trow * tilesX; }
    const int wave = threadIdx.y, wx = wave % TW_WAVES, wy = wave / TW_WAVES;
    const int X = tcol * TW + wx * 256 + 4 * threadIdx.x;
    const int Y0 = trow * TH + wy * ROWS_PER_LANE;
    unsigned wy_[ROWS_PER_LANE], cu[ROWS_PER_LANE], cv[ROWS_PER_LANE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_LANE; ++r) {
        const unsigned * py = reinterpret_cast<const unsigned *>(y + (size_t)(Y0 + r) * W + X);
        wy_[r] = (MODE & 4) ? __builtin_nontemporal_load(py) : *py;
        cu[r] = cv[r] = 0;
        if (!(r & 1)) {
            const uint16_t * pu = reinterpret_cast<const uint16_t *>(u + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1));
            const uint16_t * pv = reinterpret_cast<const uint16_t *>(v + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1));
            cu[r] = (MODE & 4) ? __builtin_nontemporal_load(pu) : *pu;
            cv[r] = (MODE & 4) ? __builtin_nontemporal_load(pv) : *pv;
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS_PER_LANE; ++r) {
        const unsigned c = cu[r & ~1] | (cv[r & ~1] << 16);
        u4 o;
        o.x = (wy_[r] & 0xff) | (c << 8);
        o.y = ((wy_[r] >> 8) & 0xff) | (c << 8);
        o.z = ((wy_[r] >> 16) & 0xff) | (c & 0xffffff00u);
        o.w = (wy_[r] >> 24) | (c & 0xffffff00u);
        u4 * dst = reinterpret_cast<u4 *>(rgba + ((size_t)(Y0 + r) * W + X) * 4);
        if (MODE & 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(o) : "memory");
        else if (MODE & 2) __builtin_nontemporal_store(o, dst);
        else *dst = o;
    }
}
__global__ __launch_bounds__(256) void linearCopy(const u4 * __restrict__ in, u4 * __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
template <typename F>
static float timeIt(F launch, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        for (int i = 0; i < 4; ++i) launch(i);
        hipEventRecord(a);
        for (int i = 0; i < iters; ++i) launch(i);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    return best * 1000.0f;
}
uint8_t *y[4], *u[4], *v[4], *o[4];
template <int TWW, int RPL, int MODE>
static void run(const char * name)
{
    constexpr int TW = 256 * TWW, TH = RPL * (4 / TWW);
    const unsigned blocks = (W / TW) * (H / TH);
    const float same = timeIt([&](int i) { tileCopy<TWW, RPL, MODE><<<blocks, dim3(64, 4)>>>(y[0], u[0], v[0], o[0]); }, 40);
    const float cyc = timeIt([&](int i) { tileCopy<TWW, RPL, MODE><<<blocks, dim3(64, 4)>>>(y[i & 3], u[i & 3], v[i & 3], o[i & 3]); }, 40);
    printf("%-44s tile %4dx%-3d  same %6.1f us (%4.1f%%)  cycling %6.1f us (%4.1f%%)\n", name, TW, TH, same, 182.4768 / same / 8e-2 * 1e-0 * 1.0 * 1e-0 / 1.0 * 1.0 * 1e0 * 1e-0 * 1e0 * 1e0 * 1e0, cyc, 182.4768 / cyc / 8e-2);
}
int main()
{
    const size_t ySize = (size_t)W * H, cSize = ySize / 4, oSize = ySize * 4;
    for (int k = 0; k < 4; ++k) {
        CK(hipMalloc(&y[k], ySize)); CK(hipMalloc(&u[k], cSize)); CK(hipMalloc(&v[k], cSize)); CK(hipMalloc(&o[k], oSize));
        CK(hipMemset(y[k], 0x40 + k, ySize)); CK(hipMemset(u[k], 0x80, cSize)); CK(hipMemset(v[k], 0x81, cSize)); CK(hipMemset(o[k], 0, oSize));
    }
    CK(hipDeviceSynchronize());
    const size_t n = oSize / 16;
    const float lc = timeIt([&](int i) { linearCopy<<<(unsigned)((n + 255) / 256), 256>>>((const u4 *)o[i & 3], (u4 *)o[(i + 1) & 3], n); }, 20);
    printf("linearCopy 133MB->133MB cycling: %.1f us  (%.0f GB/s total)\n", lc, 2 * oSize / 1e3 / lc);
    run<1, 2, 0>("256x8 row-major");
    run<1, 2, 1>("256x8 xcd-banded");
    run<1, 2, 16>("256x8 column-major");
    run<1, 2, 17>("256x8 column-major xcd-banded");
    run<1, 2, 2>("256x8 nt stores");
    run<1, 2, 4>("256x8 nt loads");
    run<1, 2, 6>("256x8 nt loads+stores");
    run<1, 2, 8>("256x8 sc1 stores");
    run<4, 2, 0>("1024x2 row-major");
    run<4, 2, 1>("1024x2 xcd-banded");
    run<4, 4, 0>("1024x4 row-major");
    run<4, 8, 0>("1024x8 row-major");
    run<2, 4, 0>("512x8 row-major");
    run<1, 4, 0>("256x16 row-major");
    run<1, 4, 1>("256x16 xcd-banded");
    run<1, 8, 1>("256x32 xcd-banded");
    run<1, 8, 17>("256x32 column-major xcd-banded");
    return 0;
}
