// transpose_probe.hip -- what a quarter-turn store pattern costs when fused into a row-oriented conversion kernel: a lane holds
// 4 columns x RPL rows of pixels; it writes, for each of its 4 source columns, RPL consecutive destination pixels (RPL * 4 bytes)
// of one destination row straight from registers.  Four stacked waves of a workgroup complete 4 * RPL * 4 bytes of each row.
// Compared: plain vs non-temporal stores, RPL 8 / 16, and the row-preserving store of the same tile as the baseline.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int W = 7680, H = 4320;

template <int RPL, int MODE> // MODE 0: row-preserving nt; 1: transposed nt; 2: transposed plain
__global__ __launch_bounds__(256) void k(const uint8_t * __restrict__ y, uint8_t * __restrict__ out)
{
    const int tilesX = W / 256;
    const int trow = blockIdx.x / tilesX, tcol = blockIdx.x - trow * tilesX;
    const int X = tcol * 256 + 4 * threadIdx.x;
    const int Y0 = (trow * 4 + threadIdx.y) * RPL;
    if (Y0 >= H) return;
    unsigned wy[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) wy[r] = *reinterpret_cast<const unsigned *>(y + (size_t)(Y0 + r) * W + X);
    unsigned px[RPL][4];
#pragma unroll
    for (int r = 0; r < RPL; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) px[r][c] = ((wy[r] >> (8 * c)) & 0xff) * 0x010101u | 0xff000000u;
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r)
            __builtin_nontemporal_store((u4) { px[r][0], px[r][1], px[r][2], px[r][3] }, reinterpret_cast<u4 *>(out + ((size_t)(Y0 + r) * W + X) * 4));
    } else {
        // quarter turn: source (i, j) -> destination row i, column j (mirroring only flips orders); destination pitch = H pixels
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < RPL / 4; ++q) {
                const u4 v = { px[4 * q][c], px[4 * q + 1][c], px[4 * q + 2][c], px[4 * q + 3][c] };
                u4 * dst = reinterpret_cast<u4 *>(out + ((size_t)(X + c) * H + Y0 + 4 * q) * 4);
                if (MODE == 1) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
    }
}
template <int RPL, int MODE>
static void run(const char * name, uint8_t * const * y, uint8_t * const * o, int n)
{
    const unsigned blocks = (W / 256) * ((H + 4 * RPL - 1) / (4 * RPL));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 8; ++i) k<RPL, MODE><<<blocks, dim3(64, 4)>>>(y[i % n], o[i % n]);
        hipEventRecord(a);
        for (int i = 0; i < 40; ++i) k<RPL, MODE><<<blocks, dim3(64, 4)>>>(y[i % n], o[i % n]);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms / 40 * 1000);
    }
    std::sort(t.begin(), t.end());
    printf("%-40s %6.1f us  (%.0f GB/s of 165.9 MB)\n", name, t[2], 165.888e3 / t[2]);
}
int main()
{
    uint8_t *y[4], *o[4];
    for (int k2 = 0; k2 < 4; ++k2) { CK(hipMalloc(&y[k2], (size_t)W * H)); CK(hipMalloc(&o[k2], (size_t)W * H * 4)); CK(hipMemset(y[k2], 0x40 + k2, (size_t)W * H)); }
    for (int i = 0; i < 20; ++i) run<8, 0>("(ramp)", y, o, 4);
    run<8, 0>("rows, 8 rows/wave, nt", y, o, 4);
    run<8, 1>("transposed 32 B/lane/row, nt", y, o, 4);
    run<8, 2>("transposed 32 B/lane/row, plain", y, o, 4);
    run<16, 0>("rows, 16 rows/wave, nt", y, o, 4);
    run<16, 1>("transposed 64 B/lane/row, nt", y, o, 4);
    run<16, 2>("transposed 64 B/lane/row, plain", y, o, 4);
    return 0;
}
