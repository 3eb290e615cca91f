// ws_probe.hip -- how does the streaming rate of cfg2's byte movement depend on the working set (frames cycled), on the
// allocation (separate hipMallocs vs one slab) and on the direction (read only / write only / both)?  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int W = 7680, H = 4320;
__device__ __forceinline__ uint32_t remap(uint32_t b, uint32_t n) { const uint32_t per = n >> 3, rem = n & 7, x = b & 7, s = b >> 3; return x * per + (x < rem ? x : rem) + s; }

// MODE bit0: read planes; bit1: write rgba (nt); bit2: plain stores instead of nt
template <int MODE>
__global__ __launch_bounds__(256) void tileCopy(const uint8_t * __restrict__ y, const uint8_t * __restrict__ u, const uint8_t * __restrict__ v, uint8_t * __restrict__ rgba, unsigned * sink)
{
    constexpr int RPL = 4; // rows per wave: tile 256 x 16
    const int tilesX = W / 256;
    const uint32_t tile = remap(blockIdx.x, gridDim.x);
    const int trow = tile / tilesX, tcol = tile - trow * tilesX;
    const int X = tcol * 256 + 4 * threadIdx.x;
    const int Y0 = trow * 16 + threadIdx.y * RPL;
    unsigned wy[RPL], cu[RPL], cv[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        wy[r] = 0x40414243u, cu[r] = cv[r] = 0x80;
        if (MODE & 1) {
            wy[r] = *reinterpret_cast<const unsigned *>(y + (size_t)(Y0 + r) * W + X);
            if (!(r & 1)) {
                cu[r] = *reinterpret_cast<const uint16_t *>(u + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1));
                cv[r] = *reinterpret_cast<const uint16_t *>(v + (size_t)((Y0 + r) >> 1) * (W / 2) + (X >> 1));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const unsigned c = cu[r & ~1] | (cv[r & ~1] << 16);
        u4 o;
        o.x = (wy[r] & 0xff) | (c << 8);
        o.y = ((wy[r] >> 8) & 0xff) | (c << 8);
        o.z = ((wy[r] >> 16) & 0xff) | (c & 0xffffff00u);
        o.w = (wy[r] >> 24) | (c & 0xffffff00u);
        acc += o.x + o.y + o.z + o.w;
        if (MODE & 2) {
            u4 * dst = reinterpret_cast<u4 *>(rgba + ((size_t)(Y0 + r) * W + X) * 4);
            if (MODE & 4) *dst = o; else __builtin_nontemporal_store(o, dst);
        }
    }
    if (!(MODE & 2) && acc == 0x12345u) *sink = acc;
}

struct Frame { uint8_t *y, *u, *v, *o; };
static unsigned * sink;
template <int MODE>
static float timeIt(const std::vector<Frame> & f, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned blocks = (W / 256) * (H / 16);
    const int n = (int)f.size();
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 8; ++i) tileCopy<MODE><<<blocks, dim3(64, 4)>>>(f[i % n].y, f[i % n].u, f[i % n].v, f[i % n].o, sink);
        hipEventRecord(a);
        for (int i = 0; i < iters; ++i) tileCopy<MODE><<<blocks, dim3(64, 4)>>>(f[i % n].y, f[i % n].u, f[i % n].v, f[i % n].o, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms / iters * 1000.0f);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}
int main()
{
    const size_t ySize = (size_t)W * H, cSize = ySize / 4, oSize = ySize * 4;
    const size_t frameBytes = ySize + 2 * cSize + oSize; // 182.5 MB
    CK(hipMalloc(&sink, 4));
    const int NMAX = 24;
    std::vector<Frame> sep(NMAX), slab(NMAX);
    for (int k = 0; k < NMAX; ++k) {
        CK(hipMalloc(&sep[k].y, ySize)); CK(hipMalloc(&sep[k].u, cSize)); CK(hipMalloc(&sep[k].v, cSize)); CK(hipMalloc(&sep[k].o, oSize));
        CK(hipMemset(sep[k].y, 0x40 + k, ySize)); CK(hipMemset(sep[k].u, 0x80, cSize)); CK(hipMemset(sep[k].v, 0x81, cSize)); CK(hipMemset(sep[k].o, 0, oSize));
    }
    uint8_t * big;
    const size_t stride = (frameBytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    CK(hipMalloc(&big, stride * NMAX));
    CK(hipMemset(big, 0x55, stride * NMAX));
    for (int k = 0; k < NMAX; ++k) {
        uint8_t * p = big + stride * k;
        slab[k].y = p; slab[k].u = p + ySize; slab[k].v = p + ySize + cSize; slab[k].o = p + ySize + 2 * cSize;
    }
    CK(hipDeviceSynchronize());
    // clock ramp: ~0.3 s of work
    { std::vector<Frame> f(sep.begin(), sep.begin() + 4); for (int i = 0; i < 3; ++i) timeIt<3>(f, 400); }
    printf("frames  set(MB)  | separate allocations: r+w(nt)  r+w(plain)  read-only  write-only(nt) | one slab: r+w(nt)  read-only  write-only(nt)   [us per pass; r+w moves 182.5 MB, read 49.8 MB, write 132.7 MB]\n");
    for (int n : { 1, 2, 4, 6, 8, 12, 16, 24 }) {
        std::vector<Frame> a(sep.begin(), sep.begin() + n), b(slab.begin(), slab.begin() + n);
        const float t1 = timeIt<3>(a, 48), t2 = timeIt<7>(a, 48), t3 = timeIt<1>(a, 48), t4 = timeIt<2>(a, 48);
        const float s1 = timeIt<3>(b, 48), s3 = timeIt<1>(b, 48), s4 = timeIt<2>(b, 48);
        printf("%5d  %7.0f  | %7.1f (%.2f)  %7.1f  %7.1f (%4.0f GB/s)  %7.1f (%4.0f GB/s) | %7.1f (%.2f)  %7.1f  %7.1f\n", n, n * frameBytes / 1e6, t1, 182.4768 / t1 / 8e-3 * 1e-3 * 1e0, t2, t3,
               49.7664e3 / t3, t4, 132.7104e3 / t4, s1, 182.4768 / s1 / 8e-3 * 1e-3, s3, s4);
        fflush(stdout);
    }
    return 0;
}
