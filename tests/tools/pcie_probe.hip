// pcie_probe.hip -- what the host <-> HBM legs of a drop-in call cost on this box (run on the GPU box): pageable and pinned
// copies each way, both ways at once, hipHostRegister, and the CPU-side memcpy between pageable and pinned memory with 1..8
// threads.  Sizes are cfg2's: 49.8 MB of planes in, 132.7 MB of pixels out.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static double best(F f, int n = 5) { double b = 1e9; for (int i = 0; i < n; ++i) { const double t0 = now(); f(); const double t = now() - t0; if (t < b) b = t; } return b; }

static void parCopy(uint8_t * dst, const uint8_t * src, size_t n, int threads)
{
    if (threads <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t o = (size_t)t * chunk;
        if (o >= n) break;
        const size_t len = (o + chunk > n) ? n - o : chunk;
        th.emplace_back([=] { memcpy(dst + o, src + o, len); });
    }
    for (auto & t : th) t.join();
}

int main()
{
    const size_t IN = 49766400, OUT = 132710400;
    uint8_t *dIn, *dOut, *pIn, *pOut;
    uint8_t * hIn = (uint8_t *)malloc(IN), * hOut = (uint8_t *)malloc(OUT);
    memset(hIn, 1, IN); memset(hOut, 2, OUT);
    CK(hipMalloc(&dIn, IN)); CK(hipMalloc(&dOut, OUT));
    CK(hipHostMalloc(&pIn, IN, hipHostMallocDefault)); CK(hipHostMalloc(&pOut, OUT, hipHostMallocDefault));
    memset(pIn, 1, IN); memset(pOut, 2, OUT);
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    double t;
    t = best([&] { hipMemcpy(dIn, hIn, IN, hipMemcpyHostToDevice); }); printf("pageable H2D  %6.1f MB: %6.2f ms  %5.1f GB/s\n", IN / 1e6, t * 1e3, IN / t / 1e9);
    t = best([&] { hipMemcpy(hOut, dOut, OUT, hipMemcpyDeviceToHost); }); printf("pageable D2H  %6.1f MB: %6.2f ms  %5.1f GB/s\n", OUT / 1e6, t * 1e3, OUT / t / 1e9);
    t = best([&] { hipMemcpyAsync(dIn, pIn, IN, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }); printf("pinned   H2D  %6.1f MB: %6.2f ms  %5.1f GB/s\n", IN / 1e6, t * 1e3, IN / t / 1e9);
    t = best([&] { hipMemcpyAsync(pOut, dOut, OUT, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); }); printf("pinned   D2H  %6.1f MB: %6.2f ms  %5.1f GB/s\n", OUT / 1e6, t * 1e3, OUT / t / 1e9);
    t = best([&] { hipMemcpyAsync(dIn, pIn, IN, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pOut, dOut, OUT, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
    printf("pinned both ways at once: %6.2f ms  (%5.1f GB/s out, %5.1f GB/s total)\n", t * 1e3, OUT / t / 1e9, (IN + OUT) / t / 1e9);
    for (size_t chunkMB : { 1, 4, 16 }) {
        const size_t c = chunkMB << 20;
        t = best([&] { for (size_t o = 0; o < OUT; o += c) hipMemcpyAsync(pOut + o, dOut + o, (o + c > OUT) ? OUT - o : c, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); });
        printf("pinned D2H in %2zu MB chunks: %6.2f ms  %5.1f GB/s\n", chunkMB, t * 1e3, OUT / t / 1e9);
    }
    // pageable memory, asynchronous calls: do the two directions overlap when issued from ONE thread?  from two?
    t = best([&] { hipMemcpyAsync(dIn, hIn, IN, hipMemcpyHostToDevice, s1); hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
    printf("pageable both ways, async calls from one thread: %6.2f ms (serial sum would be ~3.25, full overlap ~2.5)\n", t * 1e3);
    t = best([&] {
        std::thread up([&] { hipMemcpyAsync(dIn, hIn, IN, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); });
        hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2);
        up.join();
    });
    printf("pageable both ways, one thread per direction:     %6.2f ms\n", t * 1e3);
    {
        double t0 = now(); hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, s2); double t1 = now(); hipStreamSynchronize(s2); double t2 = now();
        printf("pageable D2H async call returns after %6.2f ms, sync adds %6.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
        t0 = now(); hipMemcpyAsync(dIn, hIn, IN, hipMemcpyHostToDevice, s1); t1 = now(); hipStreamSynchronize(s1); t2 = now();
        printf("pageable H2D async call returns after %6.2f ms, sync adds %6.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    }
    for (int chunks : { 2, 4, 8 }) {
        const size_t ci = IN / chunks, co = OUT / chunks;
        t = best([&] {
            for (int c = 0; c < chunks; ++c) { hipMemcpyAsync(dIn + c * ci, hIn + c * ci, ci, hipMemcpyHostToDevice, s1); hipMemcpyAsync(hOut + c * co, dOut + c * co, co, hipMemcpyDeviceToHost, s2); }
            hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
        printf("pageable both ways in %d chunks, one thread:      %6.2f ms\n", chunks, t * 1e3);
        t = best([&] {
            std::thread up([&] { for (int c = 0; c < chunks; ++c) hipMemcpyAsync(dIn + c * ci, hIn + c * ci, ci, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); });
            for (int c = 0; c < chunks; ++c) hipMemcpyAsync(hOut + c * co, dOut + c * co, co, hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s2); up.join(); });
        printf("pageable both ways in %d chunks, two threads:     %6.2f ms\n", chunks, t * 1e3);
    }
    for (int th : { 1, 2, 4, 8, 16 }) {
        const double a = best([&] { parCopy(pIn, hIn, IN, th); }), b = best([&] { parCopy(hOut, pOut, OUT, th); });
        printf("CPU memcpy %2d thread(s): pageable->pinned %5.1f GB/s (%5.2f ms)   pinned->pageable %5.1f GB/s (%5.2f ms)\n", th, IN / a / 1e9, a * 1e3, OUT / b / 1e9, b * 1e3);
    }
    {
        const double t0 = now(); CK(hipHostRegister(hOut, OUT, hipHostRegisterDefault)); const double t1 = now();
        const double c = best([&] { hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); });
        const double t2 = now(); CK(hipHostUnregister(hOut)); const double t3 = now();
        printf("hipHostRegister %6.1f MB: %6.2f ms, unregister %6.2f ms; D2H into registered memory %5.1f GB/s\n", OUT / 1e6, (t1 - t0) * 1e3, (t3 - t2) * 1e3, OUT / c / 1e9);
        const double t4 = now(); CK(hipHostRegister(hOut, OUT, hipHostRegisterDefault)); const double t5 = now(); CK(hipHostUnregister(hOut));
        printf("hipHostRegister again (pages touched): %6.2f ms\n", (t5 - t4) * 1e3);
    }
    return 0;
}
