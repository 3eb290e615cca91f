// api.cpp -- the C ABI of libavifhip.so (include/avifhip.h): argument checks and error codes of
// libavif's entry points, pointer classification (host vs HBM), staging through device scratch for
// host-resident images, kernel selection, and the per-thread stream/scratch context.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "avifhip.h"
#include "kernels.h"
#include "plan.h"
#include "gainmap_plan.h"
#include "scale_plan.h"

using namespace avifhip;

namespace {

std::atomic<int> gArithmetic { -1 }; // -1: not decided yet (environment, then AUTO)
std::atomic<int> gTiledKernels { 1 };
std::atomic<uint32_t> gTuning { TUNE_DEFAULT };

struct Scratch
{
    void * ptr = nullptr;
    size_t capacity = 0;
};

// what tls.scaleTable currently holds (avifhipImageScaleAsync)
struct ScaleTableCache
{
    bool valid = false;
    uint32_t key[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    size_t offset[4] = { 0, 0, 0, 0 };
    int mode[4] = { 0, 0, 0, 0 };
    ScaleStaging staging[4]; // row-staged kernel
    ScaleStaging window[4];  // window kernel
};

// what tls.gainMap[2] currently holds (the tables of avifhipRGBImageApplyGainMap): rebuilt only when a parameter changes
struct GainMapTableCache
{
    bool valid = false;
    struct Key
    {
        uint32_t baseTC, baseDepth, baseFloat, outTC, outDepth, outFloat, gainDepth, applyGain;
        float gammaInv[3], minLog2[3], maxLog2[3], weight;
        uint64_t stream;
    } key;
    size_t baseLutOffset = 0, gainLutOffset = 0, stepsOffset = 0, guideOffset = 0; // in floats
    uint32_t maxCode = 0, nanCode = 0, stepEntries = 0;
};

// One context per calling thread: libavif's reformat functions are re-entrant and may be called
// concurrently from up to 8 threads (src/reformat.c:1709-1735); nothing here is shared.
struct Context
{
    int device = -1;
    hipStream_t stream = nullptr;
    Scratch planes[4]; // Y, U, V, A staging
    Scratch pixels;    // interleaved RGB staging
    Scratch table;     // batch descriptor table (device)
    Scratch gridTable; // tile table of a grid conversion (device)
    Scratch scaleTable; // schedules of a plane scale (device)
    ScaleTableCache scaleCache; // ... and which geometry they belong to
    Scratch satoTable;  // input plane tables of a sample transform (device)
    Scratch gainMap[11]; // gain maps: [0] output pixels, [1] gain map as RGB, [2] tables, [3] statistics / partials, [4] scaled planes, [5] base pixels;
                         // computation: [6] tables, [7] ratios, [8] histograms, [9] alternate pixels, [10] gain-map planes
    GainMapTableCache gainMapCache; // what gainMap[2] holds
    void * pinnedTable = nullptr;
    size_t pinnedTableCapacity = 0;
    hipEvent_t tableCopied = nullptr;
    void * pinnedUpload = nullptr; // staging for small host tables (grid tile tables, scale schedules)
    size_t pinnedUploadCapacity = 0;
    hipEvent_t uploadCopied = nullptr;
    char lastError[512] = { 0 };
    const char * lastKernel = "";
    uint64_t launches = 0; // kernels enqueued by this thread

    ~Context()
    {
        // Best effort: the runtime may already be shutting down at thread/process exit.
        for (Scratch & s : planes)
            if (s.ptr)
                (void)hipFree(s.ptr);
        if (pixels.ptr)
            (void)hipFree(pixels.ptr);
        if (table.ptr)
            (void)hipFree(table.ptr);
        if (gridTable.ptr)
            (void)hipFree(gridTable.ptr);
        if (scaleTable.ptr)
            (void)hipFree(scaleTable.ptr);
        if (satoTable.ptr)
            (void)hipFree(satoTable.ptr);
        for (Scratch & g : gainMap)
            if (g.ptr)
                (void)hipFree(g.ptr);
        if (pinnedTable)
            (void)hipHostFree(pinnedTable);
        if (tableCopied)
            (void)hipEventDestroy(tableCopied);
        if (pinnedUpload)
            (void)hipHostFree(pinnedUpload);
        if (uploadCopied)
            (void)hipEventDestroy(uploadCopied);
        if (stream)
            (void)hipStreamDestroy(stream);
    }
};

thread_local Context tls;

void setError(const char * fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls.lastError, sizeof(tls.lastError), fmt, ap);
    va_end(ap);
}

// HIP failure -> avifResult.  The message is kept for avifhipLastError(); the sticky HIP error is cleared.
avifResult hipFailed(hipError_t e, const char * what)
{
    setError("%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? AVIF_RESULT_OUT_OF_MEMORY : AVIF_RESULT_UNKNOWN_ERROR;
}

#define HIP_TRY(expr)                          \
    do {                                       \
        const hipError_t hipTryErr_ = (expr);  \
        if (hipTryErr_ != hipSuccess)          \
            return hipFailed(hipTryErr_, #expr); \
    } while (0)

avifResult ensureContext()
{
    if (tls.stream)
        return AVIF_RESULT_OK;
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        setError("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return AVIF_RESULT_UNKNOWN_ERROR;
    }
    if (tls.device >= 0)
        HIP_TRY(hipSetDevice(tls.device));
    else
        HIP_TRY(hipGetDevice(&tls.device));
    HIP_TRY(hipStreamCreateWithFlags(&tls.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&tls.tableCopied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&tls.uploadCopied, hipEventDisableTiming));
    return AVIF_RESULT_OK;
}

// Enqueues a copy of a small host table to device memory.  An asynchronous copy from pageable memory may still be reading
// its source after the call returns, so the bytes go through a pinned per-thread staging buffer first; the buffer is reused
// only after the previous upload has left it.
avifResult uploadTableAsync(void * deviceDst, const void * hostSrc, size_t bytes, hipStream_t stream)
{
    if (tls.pinnedUpload)
        HIP_TRY(hipEventSynchronize(tls.uploadCopied));
    if (bytes > tls.pinnedUploadCapacity) {
        if (tls.pinnedUpload)
            HIP_TRY(hipHostFree(tls.pinnedUpload));
        tls.pinnedUpload = nullptr, tls.pinnedUploadCapacity = 0;
        const size_t rounded = (bytes + 65535) & ~(size_t)65535;
        HIP_TRY(hipHostMalloc(&tls.pinnedUpload, rounded, hipHostMallocDefault));
        tls.pinnedUploadCapacity = rounded;
    }
    memcpy(tls.pinnedUpload, hostSrc, bytes);
    HIP_TRY(hipMemcpyAsync(deviceDst, tls.pinnedUpload, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(tls.uploadCopied, stream));
    return AVIF_RESULT_OK;
}

avifResult reserve(Scratch & s, size_t bytes)
{
    if (bytes <= s.capacity)
        return AVIF_RESULT_OK;
    if (s.ptr) {
        HIP_TRY(hipStreamSynchronize(tls.stream));
        HIP_TRY(hipFree(s.ptr));
        s.ptr = nullptr;
        s.capacity = 0;
    }
    const size_t rounded = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipMalloc(&s.ptr, rounded));
    s.capacity = rounded;
    return AVIF_RESULT_OK;
}

bool isDevicePointer(const void * p)
{
    if (!p)
        return false;
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory on older runtimes
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

inline uint32_t alignUp(uint32_t v, uint32_t a)
{
    return (v + a - 1) / a * a;
}

// Initial arithmetic family from the environment (AVIFHIP_ARITHMETIC=auto|float|libyuv), so that an unmodified
// application over the seam-B build can choose; avifhipSetArithmetic() overrides.
int arithmeticFromEnvironment()
{
    const char * e = getenv("AVIFHIP_ARITHMETIC");
    if (e && !strcmp(e, "float"))
        return AVIFHIP_ARITHMETIC_FLOAT;
    if (e && !strcmp(e, "libyuv"))
        return AVIFHIP_ARITHMETIC_LIBYUV;
    return AVIFHIP_ARITHMETIC_AUTO;
}

int effectiveArithmetic()
{
    int a = gArithmetic.load(std::memory_order_relaxed);
    if (a < 0) {
        a = arithmeticFromEnvironment();
        gArithmetic.store(a, std::memory_order_relaxed);
    }
    return a;
}

// ---- kernel selection -----------------------------------------------------------------------

avifResult enqueueYuvToRgb(const YuvToRgbPlan & plan, hipStream_t stream)
{
    hipError_t e;
    if (gTiledKernels.load(std::memory_order_relaxed) && tileYuvToRgbSupported(plan)) {
        e = launchYuvToRgbTile(plan, stream, &tls.lastKernel);
    } else {
        tls.lastKernel = (plan.arith == ARITH_LIBYUV) ? "yuv2rgb_fixed_generic" : "yuv2rgb_generic";
        e = launchYuvToRgbGeneric(plan, stream);
    }
    if (e != hipSuccess)
        return hipFailed(e, "YUV->RGB kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

avifResult enqueueRgbToYuv(const RgbToYuvPlan & plan, hipStream_t stream)
{
    hipError_t e;
    if (gTiledKernels.load(std::memory_order_relaxed) && tileRgbToYuvSupported(plan)) {
        e = launchRgbToYuvTile(plan, stream, &tls.lastKernel);
    } else {
        tls.lastKernel = (plan.arith == ARITH_LIBYUV) ? "rgb2yuv_fixed_generic" : "rgb2yuv_generic";
        e = launchRgbToYuvGeneric(plan, stream);
    }
    if (e != hipSuccess)
        return hipFailed(e, "RGB->YUV kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

avifResult enqueueAlphaMul(const AlphaMulPlan & plan, hipStream_t stream)
{
    tls.lastKernel = (plan.arith == ARITH_LIBYUV) ? (plan.unmultiply ? "unattenuate_fixed_generic" : "attenuate_fixed_generic")
                                                  : (plan.unmultiply ? "unpremultiply_generic" : "premultiply_generic");
    const hipError_t e = launchAlphaMulGeneric(plan, stream);
    if (e != hipSuccess)
        return hipFailed(e, "alpha multiply kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// ---- staging of host-resident buffers -----------------------------------------------------------

struct PlaneGeometry
{
    uint32_t widthBytes[4];
    uint32_t rows[4];
};

PlaneGeometry planeGeometry(const avifImage * image)
{
    PlaneGeometry g;
    const uint32_t bps = (image->depth > 8) ? 2 : 1;
    const int sx = (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV444) ? 0 : 1;
    const int sy = (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420 || image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 1 : 0;
    const uint32_t cw = (uint32_t)(((uint64_t)image->width + sx) >> sx);
    const uint32_t ch = (uint32_t)(((uint64_t)image->height + sy) >> sy);
    g.widthBytes[0] = g.widthBytes[3] = image->width * bps;
    g.rows[0] = g.rows[3] = image->height;
    g.widthBytes[1] = g.widthBytes[2] = cw * bps;
    g.rows[1] = g.rows[2] = ch;
    return g;
}

// Replaces host plane pointers of `view` (a shallow copy of the caller's image) with device copies.
// upload=false only reserves the device planes (RGB->YUV destinations).
avifResult stagePlanes(avifImage * view, bool upload, bool mirrorRowBytes)
{
    const PlaneGeometry g = planeGeometry(view);
    for (int p = 0; p < 4; ++p) {
        uint8_t * host = (p < 3) ? view->yuvPlanes[p] : view->alphaPlane;
        const uint32_t hostRowBytes = (p < 3) ? view->yuvRowBytes[p] : view->alphaRowBytes;
        if (!host || !hostRowBytes || isDevicePointer(host))
            continue;
        const uint32_t pitch = mirrorRowBytes ? hostRowBytes : alignUp(g.widthBytes[p], 256);
        const avifResult r = reserve(tls.planes[p], (size_t)pitch * g.rows[p]);
        if (r != AVIF_RESULT_OK)
            return r;
        if (upload)
            HIP_TRY(hipMemcpy2DAsync(tls.planes[p].ptr, pitch, host, hostRowBytes, g.widthBytes[p], g.rows[p], hipMemcpyHostToDevice, tls.stream));
        if (p < 3) {
            view->yuvPlanes[p] = (uint8_t *)tls.planes[p].ptr;
            view->yuvRowBytes[p] = pitch;
        } else {
            view->alphaPlane = (uint8_t *)tls.planes[p].ptr;
            view->alphaRowBytes = pitch;
        }
    }
    return AVIF_RESULT_OK;
}

uint32_t rgbPixelBytes(const avifRGBImage * rgb)
{
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return 2;
    return (uint32_t)rgbFormatChannelCount((int)rgb->format) * ((rgb->depth > 8) ? 2 : 1);
}

avifResult stagePixels(avifRGBImage * view, bool upload)
{
    const uint32_t widthBytes = view->width * rgbPixelBytes(view);
    const uint32_t pitch = alignUp(widthBytes, 256);
    const avifResult r = reserve(tls.pixels, (size_t)pitch * view->height);
    if (r != AVIF_RESULT_OK)
        return r;
    if (upload)
        HIP_TRY(hipMemcpy2DAsync(tls.pixels.ptr, pitch, view->pixels, view->rowBytes, widthBytes, view->height, hipMemcpyHostToDevice, tls.stream));
    view->pixels = (uint8_t *)tls.pixels.ptr;
    view->rowBytes = pitch;
    return AVIF_RESULT_OK;
}

hipStream_t pickStream(void * hipStream)
{
    return hipStream ? (hipStream_t)hipStream : tls.stream;
}

// malloc-backed avifImageAllocatePlanes, reference src/avif.c:431-490
avifResult allocateHostPlanes(avifImage * image, bool withAlpha)
{
    if (image->width == 0 || image->height == 0 || image->depth == 0 || image->depth > 16)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const size_t bps = (image->depth > 8) ? 2 : 1;
    const size_t fullRow = bps * image->width;
    if (image->yuvFormat != AVIF_PIXEL_FORMAT_NONE) {
        image->imageOwnsYUVPlanes = AVIF_TRUE;
        if (!image->yuvPlanes[0]) {
            image->yuvPlanes[0] = (uint8_t *)malloc(fullRow * image->height);
            if (!image->yuvPlanes[0])
                return AVIF_RESULT_OUT_OF_MEMORY;
            image->yuvRowBytes[0] = (uint32_t)fullRow;
        }
        if (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400) {
            const PlaneGeometry g = planeGeometry(image);
            for (int p = 1; p <= 2; ++p) {
                if (!image->yuvPlanes[p]) {
                    image->yuvPlanes[p] = (uint8_t *)malloc((size_t)g.widthBytes[p] * g.rows[p]);
                    if (!image->yuvPlanes[p])
                        return AVIF_RESULT_OUT_OF_MEMORY;
                    image->yuvRowBytes[p] = g.widthBytes[p];
                }
            }
        }
    }
    if (withAlpha) {
        image->imageOwnsAlphaPlane = AVIF_TRUE;
        if (!image->alphaPlane) {
            image->alphaPlane = (uint8_t *)malloc(fullRow * image->height);
            if (!image->alphaPlane)
                return AVIF_RESULT_OUT_OF_MEMORY;
            image->alphaRowBytes = (uint32_t)fullRow;
        }
    }
    return AVIF_RESULT_OK;
}

// Completes a RgbToYuvPlan once the destination planes exist: alpha plane source, src/reformat.c:545-569
void finishRgbToYuvPlan(const avifImage * image, const avifRGBImage * rgb, RgbToYuvPlan * plan)
{
    for (int p = 0; p < 3; ++p) {
        plan->yuv.plane[p] = image->yuvPlanes[p];
        plan->yuv.rowBytes[p] = image->yuvRowBytes[p];
    }
    plan->yuv.alpha = image->alphaPlane;
    plan->yuv.alphaRowBytes = image->alphaRowBytes;
    plan->rgb.pixels = rgb->pixels;
    plan->rgb.rowBytes = rgb->rowBytes;
    plan->alphaSource = ALPHA_KEEP;
    if (image->alphaPlane && image->alphaRowBytes)
        plan->alphaSource = (plan->rgb.hasAlpha && !rgb->ignoreAlpha) ? ALPHA_PLANE : ALPHA_FILL;
}

bool sharpYuvRequested(const avifImage * image, const avifRGBImage * rgb)
{
    return !rgbFormatIsGray((int)rgb->format) && rgb->chromaDownsampling == AVIF_CHROMA_DOWNSAMPLING_SHARP_YUV &&
           image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420;
}

} // namespace

// =================================================================================================
// YUV -> RGB
// =================================================================================================

extern "C" avifResult avifhipImageYUVToRGBRectAsync(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * rect, void * hipStream)
{
    if (!canvas || !rgbCanvas)
        return AVIF_RESULT_INVALID_ARGUMENT;
    YuvToRgbPlan plan;
    const avifResult pr = makeYuvToRgbPlan(canvas, rgbCanvas, rect, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan);
    if (pr != AVIF_RESULT_OK)
        return pr;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    return enqueueYuvToRgb(plan, pickStream(hipStream));
}

extern "C" avifResult avifhipImageYUVToRGBAsync(const avifImage * image, avifRGBImage * rgb, void * hipStream)
{
    return avifhipImageYUVToRGBRectAsync(image, rgb, nullptr, hipStream);
}

static avifResult yuvToRgbSync(const avifImage * image, avifRGBImage * rgb, bool colorOnly, bool reformatAlpha)
{
    if (!image || !rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    // Validate exactly like the reference before touching the device (error-code matrix,
    // tests/gtest/avif_fuzztest_yuvrgb.cc:36-46).
    YuvToRgbPlan probe;
    const avifResult pr = makeYuvToRgbPlan(image, rgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &probe, colorOnly, reformatAlpha);
    if (pr != AVIF_RESULT_OK)
        return pr;
    if (!rgb->pixels) {
        setError("avifhipImageYUVToRGB: rgb->pixels is NULL");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;

    avifImage imageView;
    memcpy(&imageView, image, sizeof(avifImage));
    avifRGBImage rgbView = *rgb;
    avifResult r = stagePlanes(&imageView, /*upload=*/true, /*mirrorRowBytes=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    const bool pixelsOnHost = !isDevicePointer(rgb->pixels);
    if (pixelsOnHost) {
        // destination bytes the kernel does not define (alpha kept as is) must survive the round trip
        const bool keepsBytes = probe.rgb.hasAlpha && probe.alphaSource == ALPHA_KEEP;
        r = stagePixels(&rgbView, /*upload=*/keepsBytes);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    YuvToRgbPlan plan;
    r = makeYuvToRgbPlan(&imageView, &rgbView, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan, colorOnly, reformatAlpha);
    if (r != AVIF_RESULT_OK)
        return r;
    r = enqueueYuvToRgb(plan, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;
    if (pixelsOnHost) {
        const uint32_t widthBytes = rgb->width * rgbPixelBytes(rgb);
        HIP_TRY(hipMemcpy2DAsync(rgb->pixels, rgb->rowBytes, rgbView.pixels, rgbView.rowBytes, widthBytes, rgb->height, hipMemcpyDeviceToHost, tls.stream));
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    return yuvToRgbSync(image, rgb, false, false);
}

extern "C" avifResult avifhipImageYUVToRGBColorOnly(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha)
{
    return yuvToRgbSync(image, rgb, true, reformatAlpha != AVIF_FALSE);
}

namespace {
// Per-job overrides of a batch: the chroma window (cwinX0, cwinX1, cwinY0, cwinY1) and the limited-range alpha flag
struct JobOverride
{
    int32_t window[4];
    bool alphaLimited;
};
} // namespace

static avifResult batchAsyncImpl(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, const avifCropRect * rects,
                                 const JobOverride * overrides, void * hipStream)
{
    if (count == 0)
        return AVIF_RESULT_OK;
    if (!images || !rgbs)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    // pinned staging: [tile descriptors][plans: whole jobs, or the leftover right strips][leftover bottom rows]
    const size_t tileBytes = (tileBatchTableBytes(count) + 255) & ~(size_t)255;
    const size_t planBytes = (size_t)count * sizeof(YuvToRgbPlan);
    const size_t bytes = tileBytes + 2 * planBytes;
    if (bytes > tls.pinnedTableCapacity) {
        if (tls.pinnedTable) {
            HIP_TRY(hipEventSynchronize(tls.tableCopied));
            HIP_TRY(hipHostFree(tls.pinnedTable));
            tls.pinnedTable = nullptr;
            tls.pinnedTableCapacity = 0;
        }
        HIP_TRY(hipHostMalloc(&tls.pinnedTable, bytes, hipHostMallocDefault));
        tls.pinnedTableCapacity = bytes;
    } else {
        HIP_TRY(hipEventSynchronize(tls.tableCopied)); // previous upload must have consumed the table
    }
    uint8_t * pinned = (uint8_t *)tls.pinnedTable;
    YuvToRgbPlan * plansA = (YuvToRgbPlan *)(pinned + tileBytes);
    YuvToRgbPlan * plansB = plansA + count;
    uint32_t maxW = 0, maxH = 0;
    bool allTiled = gTiledKernels.load(std::memory_order_relaxed) != 0;
    int variant = -2;
    for (uint32_t k = 0; k < count; ++k) {
        if (!images[k] || !rgbs[k])
            return AVIF_RESULT_INVALID_ARGUMENT;
        const avifResult pr = makeYuvToRgbPlan(images[k], rgbs[k], rects ? &rects[k] : nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plansA[k]);
        if (pr != AVIF_RESULT_OK)
            return pr;
        if (overrides) {
            plansA[k].cwinX0 = overrides[k].window[0], plansA[k].cwinX1 = overrides[k].window[1];
            plansA[k].cwinY0 = overrides[k].window[2], plansA[k].cwinY1 = overrides[k].window[3];
            plansA[k].yuv.alphaLimited = overrides[k].alphaLimited ? 1 : 0;
        }
        maxW = plansA[k].w > maxW ? plansA[k].w : maxW;
        maxH = plansA[k].h > maxH ? plansA[k].h : maxH;
        // one launch serves the whole batch only if every job maps to the same tiled kernel
        const int v = tileYuvToRgbVariant(plansA[k]);
        if (variant == -2)
            variant = v;
        if (v < 0 || v != variant)
            allTiled = false;
    }
    const avifResult rr = reserve(tls.table, bytes);
    if (rr != AVIF_RESULT_OK)
        return rr;
    hipStream_t stream = pickStream(hipStream);
    uint8_t * dev = (uint8_t *)tls.table.ptr;
    hipError_t e = hipSuccess;
    if (allTiled) {
        const YuvToRgbPlan representative = plansA[0];
        fillTileBatchTable(plansA, count, pinned);
        // leftovers that do not fill a 4x2 pixel group: right strips (in place of the whole jobs) and bottom rows
        uint32_t restW = 0, restH = 0, restMaxH = 0, restMaxW = 0;
        for (uint32_t k = 0; k < count; ++k) {
            const YuvToRgbPlan whole = plansA[k];
            const uint32_t w4 = whole.w & ~3u, h2 = whole.h & ~1u;
            plansB[k] = whole;
            plansB[k].y0 = whole.y0 + h2, plansB[k].h = whole.h - h2, plansB[k].w = w4;
            plansA[k].x0 = whole.x0 + w4, plansA[k].w = whole.w - w4;
            restW = plansA[k].w > restW ? plansA[k].w : restW;
            restMaxH = whole.h > restMaxH ? whole.h : restMaxH;
            restH = plansB[k].h > restH ? plansB[k].h : restH;
            restMaxW = w4 > restMaxW ? w4 : restMaxW;
        }
        HIP_TRY(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(tls.tableCopied, stream));
        e = launchYuvToRgbTileBatch(dev, representative, count, maxW, maxH, stream, &tls.lastKernel);
        if (e == hipSuccess && restW)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes), count, restW, restMaxH, stream);
        if (e == hipSuccess && restH)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes) + count, count, restMaxW, restH, stream);
    } else {
        HIP_TRY(hipMemcpyAsync(dev + tileBytes, plansA, planBytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(tls.tableCopied, stream));
        tls.lastKernel = "yuv2rgb_generic_batch";
        e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes), count, maxW, maxH, stream);
    }
    if (e != hipSuccess)
        return hipFailed(e, "YUV->RGB batch kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipImageYUVToRGBBatchAsync(uint32_t count,
                                                     const avifImage * const * images,
                                                     avifRGBImage * const * rgbs,
                                                     const avifCropRect * rects,
                                                     void * hipStream)
{
    return batchAsyncImpl(count, images, rgbs, rects, nullptr, hipStream);
}

// Grid canvases: tiles converted where they lie (a batch of rectangle jobs over "virtual canvases" whose plane pointers are
// shifted so that canvas coordinates address the tile's own memory, each confined to its own chroma samples), then the
// pixels next to interior seams redone with samples fetched from both sides (kernels_generic.hip: GridReader).
extern "C" avifResult avifhipGridYUVToRGBAsync(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                               avifBool alphaIsLimitedRange, avifRGBImage * rgbCanvas, void * hipStream)
{
    if (!grid || !colorTiles || !rgbCanvas || !grid->rows || !grid->columns || !grid->outputWidth || !grid->outputHeight)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t count = grid->rows * grid->columns;
    const avifImage * first = colorTiles[0];
    if (!first || !first->width || !first->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t tw = first->width, th = first->height;
    // the grid must cover the output and no tile may lie entirely outside it (ISO/IEC 23008-12 6.6.2.3.1, src/read.c:1538-1560)
    if ((uint64_t)tw * grid->columns < grid->outputWidth || (uint64_t)th * grid->rows < grid->outputHeight ||
        (uint64_t)tw * (grid->columns - 1) >= grid->outputWidth || (uint64_t)th * (grid->rows - 1) >= grid->outputHeight)
        return AVIF_RESULT_INVALID_IMAGE_GRID;
    const int sx = (first->yuvFormat == AVIF_PIXEL_FORMAT_YUV444) ? 0 : 1;
    const int sy = (first->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    const bool subsampled = first->yuvFormat == AVIF_PIXEL_FORMAT_YUV420 || first->yuvFormat == AVIF_PIXEL_FORMAT_YUV422;
    if (count > 1 && subsampled && ((tw & 1) || (sy && (th & 1))))
        return AVIF_RESULT_INVALID_IMAGE_GRID; // odd tile sizes cannot tile a subsampled canvas (src/read.c:1562-1580)
    const uint32_t bps = (first->depth > 8) ? 2 : 1;

    std::vector<avifImage> views(count);
    std::vector<const avifImage *> viewPtrs(count);
    std::vector<avifRGBImage *> rgbPtrs(count, rgbCanvas);
    std::vector<avifCropRect> rects(count);
    std::vector<JobOverride> overrides(count);
    std::vector<GridTile> tiles(count);
    for (uint32_t t = 0; t < count; ++t) {
        const avifImage * tile = colorTiles[t];
        if (!tile || !tile->yuvPlanes[0])
            return AVIF_RESULT_INVALID_ARGUMENT;
        // "All tiles in a grid image should match the first tile", src/read.c:1832-1842
        if (tile->width != tw || tile->height != th || tile->depth != first->depth || tile->yuvFormat != first->yuvFormat ||
            tile->yuvRange != first->yuvRange || tile->colorPrimaries != first->colorPrimaries ||
            tile->transferCharacteristics != first->transferCharacteristics || tile->matrixCoefficients != first->matrixCoefficients)
            return AVIF_RESULT_INVALID_IMAGE_GRID;
        const avifImage * atile = alphaTiles ? alphaTiles[t] : nullptr;
        if (alphaTiles && (!atile || !atile->alphaPlane || atile->width != tw || atile->height != th || atile->depth != first->depth))
            return AVIF_RESULT_INVALID_IMAGE_GRID;
        const uint32_t col = t % grid->columns, row = t / grid->columns;
        const uint32_t X0 = col * tw, Y0 = row * th;
        avifCropRect & r = rects[t];
        r.x = X0, r.y = Y0;
        r.width = (X0 + tw > grid->outputWidth) ? grid->outputWidth - X0 : tw;   // src/read.c:1863-1868
        r.height = (Y0 + th > grid->outputHeight) ? grid->outputHeight - Y0 : th;
        avifImage & v = views[t];
        memcpy(&v, first, sizeof(avifImage)); // CICP, range, alphaPremultiplied: the canvas takes the first tile's
        v.width = grid->outputWidth, v.height = grid->outputHeight;
        GridTile & gt = tiles[t];
        memset(&gt, 0, sizeof(gt));
        for (int p = 0; p < 3; ++p) {
            const bool chroma = p > 0;
            gt.plane[p] = tile->yuvPlanes[p], gt.rowBytes[p] = tile->yuvRowBytes[p];
            v.yuvRowBytes[p] = tile->yuvRowBytes[p];
            v.yuvPlanes[p] = nullptr;
            if (tile->yuvPlanes[p]) {
                const uint64_t ox = chroma ? (X0 >> sx) : X0, oy = chroma ? (Y0 >> sy) : Y0;
                v.yuvPlanes[p] = tile->yuvPlanes[p] - (oy * tile->yuvRowBytes[p] + ox * bps); // canvas sample (0,0), virtually
            }
        }
        v.alphaPlane = nullptr, v.alphaRowBytes = 0;
        if (atile) {
            gt.alpha = atile->alphaPlane, gt.alphaRowBytes = atile->alphaRowBytes;
            v.alphaRowBytes = atile->alphaRowBytes;
            v.alphaPlane = atile->alphaPlane - ((uint64_t)Y0 * atile->alphaRowBytes + (uint64_t)X0 * bps);
            v.alphaPremultiplied = first->alphaPremultiplied;
        }
        viewPtrs[t] = &v;
        JobOverride & o = overrides[t];
        o.window[0] = (int32_t)(X0 >> sx), o.window[1] = (int32_t)((X0 >> sx) + ((r.width + sx) >> sx) - 1);
        o.window[2] = (int32_t)(Y0 >> sy), o.window[3] = (int32_t)((Y0 >> sy) + ((r.height + sy) >> sy) - 1);
        o.alphaLimited = atile && alphaIsLimitedRange;
    }
    avifResult r = batchAsyncImpl(count, viewPtrs.data(), rgbPtrs.data(), rects.data(), overrides.data(), hipStream);
    if (r != AVIF_RESULT_OK)
        return r;
    if (count == 1)
        return AVIF_RESULT_OK;
    // seams: only a filtering chroma upsampler looks across them
    YuvToRgbPlan canvasPlan;
    r = makeYuvToRgbPlan(viewPtrs[0], rgbCanvas, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &canvasPlan);
    if (r != AVIF_RESULT_OK)
        return r;
    canvasPlan.yuv.alphaLimited = (alphaTiles && alphaIsLimitedRange) ? 1 : 0;
    const bool filters = canvasPlan.bilinear && canvasPlan.yuv.hasColor && subsampled;
    if (!filters)
        return AVIF_RESULT_OK;
    const size_t tableBytes = tiles.size() * sizeof(GridTile);
    r = reserve(tls.gridTable, tableBytes);
    if (r != AVIF_RESULT_OK)
        return r;
    hipStream_t stream = pickStream(hipStream);
    r = uploadTableAsync(tls.gridTable.ptr, tiles.data(), tableBytes, stream);
    if (r != AVIF_RESULT_OK)
        return r;
    GridGeometry g;
    g.columns = grid->columns, g.rows = grid->rows, g.tileW = tw, g.tileH = th, g.tileCW = tw >> sx, g.tileCH = th >> sy;
    const hipError_t e = launchYuvToRgbGridSeams(canvasPlan, g, (const GridTile *)tls.gridTable.ptr, grid->columns > 1, sy && grid->rows > 1, stream);
    if (e != hipSuccess)
        return hipFailed(e, "grid seam kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// =================================================================================================
// RGB -> YUV
// =================================================================================================

extern "C" avifResult avifhipImageRGBToYUVAsync(avifImage * image, const avifRGBImage * rgb, void * hipStream)
{
    if (!image || !rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    RgbToYuvPlan plan;
    const avifResult pr = makeRgbToYuvPlan(image, rgb, effectiveArithmetic(), &plan);
    if (pr != AVIF_RESULT_OK)
        return pr;
    if (sharpYuvRequested(image, rgb))
        return AVIF_RESULT_NOT_IMPLEMENTED; // libsharpyuv is out of scope, like src/reformat_libsharpyuv.c:77-84
    const bool needAlpha = plan.rgb.hasAlpha && !rgb->ignoreAlpha;
    if (!image->yuvPlanes[0] || (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400 && (!image->yuvPlanes[1] || !image->yuvPlanes[2])) ||
        (needAlpha && !image->alphaPlane)) {
        setError("avifhipImageRGBToYUVAsync: destination planes must be allocated by the caller");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    finishRgbToYuvPlan(image, rgb, &plan);
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    return enqueueRgbToYuv(plan, pickStream(hipStream));
}

extern "C" avifResult avifhipImageRGBToYUV(avifImage * image, const avifRGBImage * rgb)
{
    if (!image || !rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    RgbToYuvPlan plan;
    avifResult r = makeRgbToYuvPlan(image, rgb, effectiveArithmetic(), &plan);
    if (r != AVIF_RESULT_OK)
        return r;
    const bool hasAlpha = plan.rgb.hasAlpha && !rgb->ignoreAlpha;
    const bool pixelsOnHost = !isDevicePointer(rgb->pixels);
    if (pixelsOnHost || !image->yuvPlanes[0]) {
        r = allocateHostPlanes(image, hasAlpha); // src/reformat.c:236-240
        if (r != AVIF_RESULT_OK)
            return r;
    }
    if (sharpYuvRequested(image, rgb))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;

    avifImage imageView;
    memcpy(&imageView, image, sizeof(avifImage));
    avifRGBImage rgbView = *rgb;
    const bool gray = rgbFormatIsGray((int)rgb->format);
    // the gray path sets whole chroma rows (padding included) to the half value: keep the caller's pitch there
    r = stagePlanes(&imageView, /*upload=*/false, /*mirrorRowBytes=*/gray);
    if (r != AVIF_RESULT_OK)
        return r;
    if (pixelsOnHost) {
        r = stagePixels(&rgbView, /*upload=*/true);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    r = makeRgbToYuvPlan(&imageView, &rgbView, effectiveArithmetic(), &plan);
    if (r != AVIF_RESULT_OK)
        return r;
    finishRgbToYuvPlan(&imageView, &rgbView, &plan);
    r = enqueueRgbToYuv(plan, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;

    const PlaneGeometry g = planeGeometry(image);
    for (int p = 0; p < 4; ++p) {
        uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
        const uint8_t * dev = (p < 3) ? imageView.yuvPlanes[p] : imageView.alphaPlane;
        const uint32_t devRowBytes = (p < 3) ? imageView.yuvRowBytes[p] : imageView.alphaRowBytes;
        if (!host || !hostRowBytes || host == dev)
            continue; // absent, or already device-resident
        if (gray && (p == 1 || p == 2)) {
            HIP_TRY(hipMemcpyAsync(host, dev, (size_t)hostRowBytes * g.rows[p], hipMemcpyDeviceToHost, tls.stream));
        } else if (p == 1 || p == 2) {
            if (image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)
                continue; // colour source into 4:0:0: chroma untouched
            HIP_TRY(hipMemcpy2DAsync(host, hostRowBytes, dev, devRowBytes, g.widthBytes[p], g.rows[p], hipMemcpyDeviceToHost, tls.stream));
        } else {
            HIP_TRY(hipMemcpy2DAsync(host, hostRowBytes, dev, devRowBytes, g.widthBytes[p], g.rows[p], hipMemcpyDeviceToHost, tls.stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    return AVIF_RESULT_OK;
}

// =================================================================================================
// premultiply / unpremultiply
// =================================================================================================

static avifResult alphaMulAsync(avifRGBImage * rgb, bool unmultiply, void * hipStream)
{
    if (!rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    AlphaMulPlan plan;
    const avifResult pr = makeAlphaMulPlan(rgb, unmultiply, effectiveArithmetic(), &plan);
    if (pr != AVIF_RESULT_OK)
        return pr;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    return enqueueAlphaMul(plan, pickStream(hipStream));
}

static avifResult alphaMulSync(avifRGBImage * rgb, bool unmultiply)
{
    if (!rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    AlphaMulPlan plan;
    avifResult r = makeAlphaMulPlan(rgb, unmultiply, effectiveArithmetic(), &plan);
    if (r != AVIF_RESULT_OK)
        return r;
    r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    avifRGBImage view = *rgb;
    const bool onHost = !isDevicePointer(rgb->pixels);
    if (onHost) {
        r = stagePixels(&view, /*upload=*/true);
        if (r != AVIF_RESULT_OK)
            return r;
        r = makeAlphaMulPlan(&view, unmultiply, effectiveArithmetic(), &plan);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    r = enqueueAlphaMul(plan, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;
    if (onHost) {
        const uint32_t widthBytes = rgb->width * rgbPixelBytes(rgb);
        HIP_TRY(hipMemcpy2DAsync(rgb->pixels, rgb->rowBytes, view.pixels, view.rowBytes, widthBytes, rgb->height, hipMemcpyDeviceToHost, tls.stream));
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    return AVIF_RESULT_OK;
}

// in-place integer -> half float, src/reformat.c:1419-1443
extern "C" avifResult avifhipRGBImageToF16(avifRGBImage * rgb)
{
    if (!rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (!rgb->isFloat || rgb->depth != 16 || !rgb->pixels || !rgb->rowBytes || rgb->format == AVIF_RGB_FORMAT_RGB_565)
        return AVIF_RESULT_NOT_IMPLEMENTED;
    avifResult r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    avifRGBImage view = *rgb;
    const bool onHost = !isDevicePointer(rgb->pixels);
    if (onHost) {
        r = stagePixels(&view, /*upload=*/true);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    const uint32_t channels = (uint32_t)rgbFormatChannelCount((int)rgb->format);
    const float multiplier = 1.9259299444e-34f * (1.0f / 65535.0f); // src/reformat.c:1411,1429-1430
    tls.lastKernel = "to_f16_generic";
    const hipError_t e = launchToF16Generic(view.pixels, view.rowBytes, view.width * channels, view.height, multiplier, tls.stream);
    if (e != hipSuccess)
        return hipFailed(e, "half-float kernel launch");
    ++tls.launches;
    if (onHost)
        HIP_TRY(hipMemcpy2DAsync(rgb->pixels, rgb->rowBytes, view.pixels, view.rowBytes, (size_t)rgb->width * channels * 2, rgb->height, hipMemcpyDeviceToHost, tls.stream));
    HIP_TRY(hipStreamSynchronize(tls.stream));
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipRGBImagePremultiplyAlpha(avifRGBImage * rgb)
{
    return alphaMulSync(rgb, false);
}
extern "C" avifResult avifhipRGBImageUnpremultiplyAlpha(avifRGBImage * rgb)
{
    return alphaMulSync(rgb, true);
}
extern "C" avifResult avifhipRGBImagePremultiplyAlphaAsync(avifRGBImage * rgb, void * hipStream)
{
    return alphaMulAsync(rgb, false, hipStream);
}
extern "C" avifResult avifhipRGBImageUnpremultiplyAlphaAsync(avifRGBImage * rgb, void * hipStream)
{
    return alphaMulAsync(rgb, true, hipStream);
}

// plane sizes of an image (avifImagePlaneWidth / Height, reference src/avif.c:351-400)
namespace {

struct PlaneDims
{
    int w[4], h[4];
};
PlaneDims planeDims(uint32_t width, uint32_t height, int yuvFormat)
{
    const int sx = (yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 0 : 1;
    const int sy = (yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    PlaneDims d;
    d.w[0] = d.w[3] = (int)width, d.h[0] = d.h[3] = (int)height;
    d.w[1] = d.w[2] = (int)((width + sx) >> sx), d.h[1] = d.h[2] = (int)((height + sy) >> sy);
    return d;
}

} // namespace

// =================================================================================================
// Sample Transform derived image items, reference src/sampletransform.c
// =================================================================================================

extern "C" avifResult avifhipImageApplyOperationsAsync(avifImage * dstImage, avifSampleTransformBitDepth bitDepth, uint32_t numTokens,
                                                       const avifSampleTransformToken * tokens, uint8_t numInputImageItems,
                                                       const avifImage * const * inputImageItems, avifPlanesFlags planes, void * hipStream)
{
    if (!dstImage || !tokens || !inputImageItems)
        return AVIF_RESULT_INVALID_ARGUMENT;
    // avifSampleTransformExpressionIsValid, src/sampletransform.c:13-40 (AVIF_ASSERT_OR_RETURN: INTERNAL_ERROR in release builds)
    if (numTokens == 0 || numTokens > (uint32_t)kSatoMaxTokens || numInputImageItems > kSatoMaxInputs)
        return (numTokens == 0) ? AVIF_RESULT_INTERNAL_ERROR : AVIF_RESULT_NOT_IMPLEMENTED;
    uint32_t depthOfStack = 0;
    for (uint32_t t = 0; t < numTokens; ++t) {
        const int type = (int)tokens[t].type;
        if (type >= AVIF_SAMPLE_TRANSFORM_RESERVED)
            return AVIF_RESULT_INTERNAL_ERROR;
        if (type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX && (tokens[t].inputImageItemIndex == 0 || tokens[t].inputImageItemIndex > numInputImageItems))
            return AVIF_RESULT_INTERNAL_ERROR;
        if (type < AVIF_SAMPLE_TRANSFORM_FIRST_UNARY_OPERATOR) {
            ++depthOfStack;
        } else if (type < AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR) {
            if (depthOfStack < 1)
                return AVIF_RESULT_INTERNAL_ERROR;
        } else {
            if (depthOfStack < 2)
                return AVIF_RESULT_INTERNAL_ERROR;
            --depthOfStack;
        }
    }
    if (depthOfStack != 1)
        return AVIF_RESULT_INTERNAL_ERROR;
    const bool skipColor = !(planes & AVIF_PLANES_YUV), skipAlpha = !(planes & AVIF_PLANES_A);
    const PlaneDims dd = planeDims(dstImage->width, dstImage->height, (int)dstImage->yuvFormat);
    auto planeW = [&](const avifImage * im, int c) { // avifImagePlaneWidth / Height, src/avif.c:351-400: 0 when the plane is absent
        const PlaneDims d = planeDims(im->width, im->height, (int)im->yuvFormat);
        const bool present = (c < 3) ? (im->yuvPlanes[c] && !((c == 1 || c == 2) && im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)) : im->alphaPlane != nullptr;
        return present ? d.w[c] : 0;
    };
    auto planeH = [&](const avifImage * im, int c) {
        const PlaneDims d = planeDims(im->width, im->height, (int)im->yuvFormat);
        const bool present = (c < 3) ? (im->yuvPlanes[c] && !((c == 1 || c == 2) && im->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)) : im->alphaPlane != nullptr;
        return present ? d.h[c] : 0;
    };
    for (int c = 0; c < 4; ++c) { // :371-384
        if ((skipColor && c < 3) || (skipAlpha && c == 3))
            continue;
        for (uint32_t i = 0; i < numInputImageItems; ++i) {
            if (!inputImageItems[i])
                return AVIF_RESULT_INVALID_ARGUMENT;
            if (planeW(inputImageItems[i], c) != planeW(dstImage, c) || planeH(inputImageItems[i], c) != planeH(dstImage, c))
                return AVIF_RESULT_BMFF_PARSE_FAILED;
        }
    }
    if (bitDepth != AVIF_SAMPLE_TRANSFORM_BIT_DEPTH_32)
        return AVIF_RESULT_NOT_IMPLEMENTED; // :386-395
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    // input plane tables of the (up to four) planes, one upload
    SatoInputs tables[4];
    memset(tables, 0, sizeof(tables));
    bool run[4] = { false, false, false, false };
    for (int c = 0; c < 4; ++c) {
        if ((skipColor && c < 3) || (skipAlpha && c == 3) || planeW(dstImage, c) == 0 || planeH(dstImage, c) == 0)
            continue;
        run[c] = true;
        for (uint32_t i = 0; i < numInputImageItems; ++i) {
            const avifImage * im = inputImageItems[i];
            tables[c].plane[i] = (c < 3) ? im->yuvPlanes[c] : im->alphaPlane;
            tables[c].pitch[i] = (c < 3) ? im->yuvRowBytes[c] : im->alphaRowBytes;
            tables[c].wide[i] = im->depth > 8;
        }
    }
    avifResult r = reserve(tls.satoTable, sizeof(tables));
    if (r != AVIF_RESULT_OK)
        return r;
    r = uploadTableAsync(tls.satoTable.ptr, tables, sizeof(tables), stream);
    if (r != AVIF_RESULT_OK)
        return r;
    SatoArgs A;
    memset(&A, 0, sizeof(A));
    A.numTokens = (int32_t)numTokens;
    for (uint32_t t = 0; t < numTokens; ++t) {
        A.tokens[t].type = (int32_t)tokens[t].type;
        A.tokens[t].value = (tokens[t].type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX) ? (int32_t)tokens[t].inputImageItemIndex - 1 : tokens[t].constant;
    }
    A.maxValue = (1 << dstImage->depth) - 1;
    A.dstWide = dstImage->depth > 8;
    for (int c = 0; c < 4; ++c) {
        if (!run[c])
            continue;
        A.dst = (c < 3) ? dstImage->yuvPlanes[c] : dstImage->alphaPlane;
        A.dstPitch = (c < 3) ? dstImage->yuvRowBytes[c] : dstImage->alphaRowBytes;
        A.width = dd.w[c], A.height = dd.h[c];
        const hipError_t e = launchSato(A, (const SatoInputs *)tls.satoTable.ptr + c, stream);
        if (e != hipSuccess)
            return hipFailed(e, "sample transform kernel launch");
    }
    tls.lastKernel = "sample_transform";
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// =================================================================================================
// gain-map application, reference src/gainmap.c:73-355
// =================================================================================================

namespace {

void diagClear(avifDiagnostics * diag)
{
    if (diag)
        diag->error[0] = '\0';
}
void diagPrintf(avifDiagnostics * diag, const char * fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    char text[AVIF_DIAGNOSTICS_ERROR_BUFFER_SIZE];
    vsnprintf(text, sizeof(text), fmt, ap);
    va_end(ap);
    if (diag)
        memcpy(diag->error, text, sizeof(text));
    setError("%s", text);
}

inline float fractionToFloat(avifSignedFraction f) // src/gainmap.c:32-38
{
    return f.d == 0 ? 0.0f : (float)f.n / f.d;
}
inline float fractionToFloat(avifUnsignedFraction f) // :40-46
{
    return f.d == 0 ? 0.0f : (float)f.n / f.d;
}

avifResult gainMapValidateMetadata(const avifGainMap * gainMap, avifDiagnostics * diag) // :430-457
{
    for (int i = 0; i < 3; ++i) {
        if (gainMap->gainMapMin[i].d == 0 || gainMap->gainMapMax[i].d == 0 || gainMap->gainMapGamma[i].d == 0 || gainMap->baseOffset[i].d == 0 ||
            gainMap->alternateOffset[i].d == 0) {
            diagPrintf(diag, "Per-channel denominator is 0 in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        if ((int64_t)gainMap->gainMapMax[i].n * gainMap->gainMapMin[i].d < (int64_t)gainMap->gainMapMin[i].n * gainMap->gainMapMax[i].d) {
            diagPrintf(diag, "Per-channel max is less than per-channel min in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        if (gainMap->gainMapGamma[i].n == 0) {
            diagPrintf(diag, "Per-channel gamma is 0 in gain map metadata");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
    }
    if (gainMap->baseHdrHeadroom.d == 0 || gainMap->alternateHdrHeadroom.d == 0) {
        diagPrintf(diag, "Headroom denominator is 0 in gain map metadata");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    if (gainMap->useBaseColorSpace != 0 && gainMap->useBaseColorSpace != 1) {
        diagPrintf(diag, "useBaseColorSpace is %d in gain map metadata", gainMap->useBaseColorSpace);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    return AVIF_RESULT_OK;
}

float gainMapWeight(float hdrHeadroom, const avifGainMap * gainMap) // avifGetGainMapWeight, :52-63
{
    const float base = fractionToFloat(gainMap->baseHdrHeadroom), alternate = fractionToFloat(gainMap->alternateHdrHeadroom);
    if (base == alternate)
        return 0.0f;
    const float r = (hdrHeadroom - base) / (alternate - base);
    const float w = (r < 0.0f) ? 0.0f : ((1.0f < r) ? 1.0f : r);
    return (alternate < base) ? -w : w;
}

bool gainMapLayout(const avifRGBImage * rgb, GainMapPixelLayout * L) // avifGetRGBColorSpaceInfo, src/reformat.c:32-117
{
    if (rgb->depth != 8 && rgb->depth != 10 && rgb->depth != 12 && rgb->depth != 16)
        return false;
    if ((rgb->isFloat && rgb->depth != 16) || (rgb->format == AVIF_RGB_FORMAT_RGB_565 && rgb->depth != 8))
        return false;
    memset(L, 0, sizeof(*L));
    const uint32_t cb = (rgb->depth > 8) ? 2 : 1;
    L->channelBytes = cb;
    uint32_t n = 0;
    switch (rgb->format) {
        case AVIF_RGB_FORMAT_RGB: L->offR = 0, L->offG = cb, L->offB = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_RGBA: L->offR = 0, L->offG = cb, L->offB = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ARGB: L->offA = 0, L->offR = cb, L->offG = 2 * cb, L->offB = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_BGR: L->offB = 0, L->offG = cb, L->offR = 2 * cb, n = 3; break;
        case AVIF_RGB_FORMAT_BGRA: L->offB = 0, L->offG = cb, L->offR = 2 * cb, L->offA = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_ABGR: L->offA = 0, L->offB = cb, L->offG = 2 * cb, L->offR = 3 * cb, n = 4; break;
        case AVIF_RGB_FORMAT_RGB_565: L->is565 = 1, n = 2; break;
        default: return false; // gray layouts have no R, G, B offsets for the tone-mapping loop to index
    }
    L->pixelBytes = L->is565 ? 2 : n * cb;
    L->hasAlpha = (n == 4 && !L->is565) ? 1 : 0;
    L->isFloat = rgb->isFloat ? 1 : 0;
    L->depth = rgb->depth;
    L->maxF = (float)((1u << rgb->depth) - 1);
    return true;
}

// The tone-mapping of device-resident images.  `gainImage`: gainMap->image with device plane pointers.  The tone-mapped
// image must already own device pixels of the base image's size.  Waits for the stream: the result code and the CLLI
// values depend on the pixels.
avifResult applyGainMapOnDevice(const avifRGBImage * base, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC, const avifGainMap * gainMap,
                                const avifImage * gainImage, float weight, avifColorPrimaries outPrimaries, avifTransferCharacteristics outTC,
                                avifRGBImage * out, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag, hipStream_t stream)
{
    const uint32_t width = base->width, height = base->height;
    GainMapArgs A;
    memset(&A, 0, sizeof(A));
    if (!gainMapLayout(base, &A.baseL) || !gainMapLayout(out, &A.outL)) {
        diagPrintf(diag, "Unsupported RGB color space");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    A.base = base->pixels, A.basePitch = base->rowBytes, A.out = out->pixels, A.outPitch = out->rowBytes;
    A.width = width, A.height = height;

    const avifColorPrimaries mathPrimaries =
        (gainMap->useBaseColorSpace || (gainMap->altColorPrimaries == AVIF_COLOR_PRIMARIES_UNSPECIFIED)) ? basePrimaries : gainMap->altColorPrimaries;
    const bool applyGain = weight != 0.0f;
    if (!applyGain) { // "Just convert from one rgb format to another", src/gainmap.c:142-170
        const bool primariesDiffer = basePrimaries != outPrimaries;
        if (primariesDiffer && !gainMapPrimariesMatrix(basePrimaries, outPrimaries, A.inM)) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
        A.inConv = primariesDiffer ? 1 : 0;
        A.convert = (outTC != baseTC || primariesDiffer) ? 1 : 0;
    } else {
        A.convert = 1;
        A.inConv = (basePrimaries != mathPrimaries) ? 1 : 0, A.outConv = (mathPrimaries != outPrimaries) ? 1 : 0;
        if ((A.inConv && !gainMapPrimariesMatrix(basePrimaries, mathPrimaries, A.inM)) ||
            (A.outConv && !gainMapPrimariesMatrix(mathPrimaries, outPrimaries, A.outM))) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
    }

    // ---- the gain map as RGB at the base image's size, :185-212 ----
    uint32_t gainDepth = 8;
    if (applyGain) {
        avifImage gm;
        memcpy(&gm, gainImage, sizeof(avifImage));
        if (gm.width != width || gm.height != height) {
            avifImage scaled;
            memcpy(&scaled, &gm, sizeof(avifImage));
            scaled.width = width, scaled.height = height;
            const PlaneDims dd = planeDims(width, height, (int)gm.yuvFormat);
            const size_t bps = (gm.depth > 8) ? 2 : 1;
            size_t offset[4] = { 0, 0, 0, 0 }, total = 0;
            uint32_t pitch[4] = { 0, 0, 0, 0 };
            for (int p = 0; p < 4; ++p) {
                const uint8_t * sp = (p < 3) ? gm.yuvPlanes[p] : gm.alphaPlane;
                if (!sp || ((p == 1 || p == 2) && gm.yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
                    continue;
                pitch[p] = alignUp((uint32_t)(dd.w[p] * bps), 256);
                offset[p] = total, total += (size_t)pitch[p] * dd.h[p];
            }
            const avifResult rr = reserve(tls.gainMap[4], total ? total : 1);
            if (rr != AVIF_RESULT_OK)
                return rr;
            for (int p = 0; p < 4; ++p) {
                uint8_t * dp = pitch[p] ? (uint8_t *)tls.gainMap[4].ptr + offset[p] : nullptr;
                if (p < 3)
                    scaled.yuvPlanes[p] = dp, scaled.yuvRowBytes[p] = pitch[p];
                else
                    scaled.alphaPlane = dp, scaled.alphaRowBytes = pitch[p];
            }
            const avifResult sr = avifhipImageScaleAsync(&gm, &scaled, stream);
            if (sr != AVIF_RESULT_OK)
                return sr;
            memcpy(&gm, &scaled, sizeof(avifImage));
        }
        avifRGBImage rgbGain; // avifRGBImageSetDefaults, src/avif.c:700-717
        memset(&rgbGain, 0, sizeof(rgbGain));
        rgbGain.width = width, rgbGain.height = height, rgbGain.depth = gm.depth, rgbGain.format = AVIF_RGB_FORMAT_RGBA;
        rgbGain.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgbGain.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
        rgbGain.maxThreads = 1;
        rgbGain.rowBytes = alignUp(width * 4 * ((gm.depth > 8) ? 2 : 1), 256);
        const avifResult rr = reserve(tls.gainMap[1], (size_t)rgbGain.rowBytes * height);
        if (rr != AVIF_RESULT_OK)
            return rr;
        rgbGain.pixels = (uint8_t *)tls.gainMap[1].ptr;
        const avifResult cr = avifhipImageYUVToRGBAsync(&gm, &rgbGain, stream);
        if (cr != AVIF_RESULT_OK)
            return cr;
        A.gain = rgbGain.pixels, A.gainPitch = rgbGain.rowBytes, A.gainDepth = gainDepth = gm.depth;
        for (int c = 0; c < 3; ++c)
            A.baseOffset[c] = fractionToFloat(gainMap->baseOffset[c]), A.altOffset[c] = fractionToFloat(gainMap->alternateOffset[c]);
    }

    // ---- tables (kept while the parameters stay the same: sequences of frames, tiles) ----
    if (A.convert) {
        GainMapTableCache & cache = tls.gainMapCache;
        GainMapTableCache::Key key;
        memset(&key, 0, sizeof(key));
        key.baseTC = baseTC, key.baseDepth = base->depth, key.baseFloat = base->isFloat ? 1 : 0;
        key.outTC = outTC, key.outDepth = A.outL.is565 ? 8 : out->depth, key.outFloat = out->isFloat ? 1 : 0;
        key.gainDepth = gainDepth, key.applyGain = applyGain ? 1 : 0, key.stream = (uint64_t)(uintptr_t)stream;
        if (applyGain) {
            for (int c = 0; c < 3; ++c) {
                key.gammaInv[c] = 1.0f / fractionToFloat(gainMap->gainMapGamma[c]);
                key.minLog2[c] = fractionToFloat(gainMap->gainMapMin[c]), key.maxLog2[c] = fractionToFloat(gainMap->gainMapMax[c]);
            }
            key.weight = weight;
        }
        if (!cache.valid || memcmp(&cache.key, &key, sizeof(key)) != 0) {
            cache.valid = false;
            std::vector<float> tables = gainMapLinearLut(baseTC, base->depth, base->isFloat != 0);
            cache.baseLutOffset = 0, cache.gainLutOffset = tables.size();
            if (applyGain) {
                for (int c = 0; c < 3; ++c) {
                    const std::vector<float> g = gainMapGainLut(gainDepth, key.gammaInv[c], key.minLog2[c], key.maxLog2[c], weight);
                    tables.insert(tables.end(), g.begin(), g.end());
                }
            }
            cache.stepsOffset = tables.size();
            const GainMapSteps & S = gainMapOutputSteps(outTC, key.outDepth, out->isFloat != 0);
            tables.insert(tables.end(), S.steps.begin(), S.steps.end());
            cache.guideOffset = tables.size();
            tables.resize(tables.size() + (S.guide.size() + 1) / 2, 0.0f); // the 16-bit guide entries ride in float slots
            memcpy(tables.data() + cache.guideOffset, S.guide.data(), S.guide.size() * sizeof(uint16_t));
            cache.maxCode = S.maxCode, cache.stepEntries = S.pieceEntries;
            // what the reference computes for a NaN input (the weight-0 path can meet one in a half-float base image)
            const float nanGamma = fminf(1.0f, fmaxf(0.0f, gainMapToGamma(outTC, NAN)));
            cache.nanCode = out->isFloat ? ((uint32_t)0) : (uint32_t)(0.5f + nanGamma * (float)((1u << key.outDepth) - 1));
            if (out->isFloat) {
                const float f = nanGamma * 1.9259299444e-34f;
                uint32_t u;
                memcpy(&u, &f, 4);
                cache.nanCode = (u >> 13) & 0xffffu;
            }
            const avifResult rr = reserve(tls.gainMap[2], tables.size() * sizeof(float));
            if (rr != AVIF_RESULT_OK)
                return rr;
            const avifResult ur = uploadTableAsync(tls.gainMap[2].ptr, tables.data(), tables.size() * sizeof(float), stream);
            if (ur != AVIF_RESULT_OK)
                return ur;
            cache.key = key;
            cache.valid = true;
        }
        const float * t = (const float *)tls.gainMap[2].ptr;
        A.baseLut = t + cache.baseLutOffset, A.gainLut = t + cache.gainLutOffset, A.steps = t + cache.stepsOffset;
        A.maxCode = cache.maxCode, A.nanCode = cache.nanCode, A.stepEntries = cache.stepEntries;
        A.guide = (const uint16_t *)(t + cache.guideOffset);
        A.guideFirstBits = kGainMapGuideFirstBits, A.guideShift = kGainMapGuideShift, A.guideBuckets = kGainMapGuideBuckets;
        // the kernel keeps the tables in LDS when all of them fit (up to 12-bit images; 16-bit and half-float tables stay in
        // global memory)
        const size_t stepsEntries = 2 * (size_t)cache.stepEntries, baseEntries = cache.gainLutOffset - cache.baseLutOffset,
                     gainEntries = cache.stepsOffset - cache.gainLutOffset;
        if ((stepsEntries + baseEntries + gainEntries) * sizeof(float) + (kGainMapGuideBuckets + 2) * sizeof(uint16_t) <= 64 * 1024)
            A.ldsSteps = (uint32_t)stepsEntries, A.ldsBaseLut = (uint32_t)baseEntries, A.ldsGainLut = (uint32_t)gainEntries;
    }

    const size_t partials = kGainMapMaxGroups;
    const avifResult sr = reserve(tls.gainMap[3], 64 + partials * (sizeof(double) + sizeof(float)));
    if (sr != AVIF_RESULT_OK)
        return sr;
    A.stats = (GainMapStats *)tls.gainMap[3].ptr;
    A.blockSum = (double *)((uint8_t *)tls.gainMap[3].ptr + 64);
    A.blockMax = (float *)(A.blockSum + partials);
    HIP_TRY(hipMemsetAsync(A.stats, 0, sizeof(GainMapStats), stream));
    const hipError_t e = launchGainMapApply(A, stream);
    if (e != hipSuccess)
        return hipFailed(e, "gain map kernel launch");
    tls.lastKernel = applyGain ? "gainmap_apply" : (A.convert ? "gainmap_convert" : "gainmap_requantise");
    ++tls.launches;
    GainMapStats stats;
    HIP_TRY(hipMemcpyAsync(&stats, A.stats, sizeof(stats), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (applyGain && stats.nan) {
        diagPrintf(diag, "Degenerate gain map parameters produce NaN");
        return AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE;
    }
    if (applyGain && clli) { // src/gainmap.c:292-302 (the reference sums in fp32 pixel by pixel; here fp64 partial sums)
        float rgbMaxLinear;
        memcpy(&rgbMaxLinear, &stats.maxBits, 4);
        const float kSdrWhiteNits = 203.0f;
        auto toNits = [&](float v) -> uint16_t {
            const float r = floorf(v * kSdrWhiteNits + 0.5f);
            return (uint16_t)((r < 0.0f) ? 0.0f : ((65535.0f < r) ? 65535.0f : r));
        };
        clli->maxCLL = toNits(rgbMaxLinear);
        clli->maxPALL = toNits((float)stats.sum / (float)((size_t)width * height));
    }
    return AVIF_RESULT_OK;
}

// argument checks shared by the entry points, src/gainmap.c:86-94
avifResult gainMapCheckArguments(const avifRGBImage * base, const avifGainMap * gainMap, float hdrHeadroom, const avifRGBImage * out, avifDiagnostics * diag)
{
    diagClear(diag);
    if (hdrHeadroom < 0.0f) {
        diagPrintf(diag, "hdrHeadroom should be >= 0, got %f", hdrHeadroom);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    if (base == NULL || gainMap == NULL || out == NULL) {
        diagPrintf(diag, "NULL input image");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    return gainMapValidateMetadata(gainMap, diag);
}

bool gainMapIsPlainCopy(const avifRGBImage * base, avifColorPrimaries basePrimaries, avifTransferCharacteristics baseTC, float weight,
                        avifColorPrimaries outPrimaries, avifTransferCharacteristics outTC, const avifRGBImage * out) // :120-128
{
    return weight == 0.0f && outTC == baseTC && outPrimaries == basePrimaries && base->format == out->format && base->depth == out->depth &&
           base->isFloat == out->isFloat && base->rowBytes == out->rowBytes;
}

} // namespace

extern "C" avifResult avifhipRGBImageApplyGainMapAsync(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                                       avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap,
                                                       float hdrHeadroom, avifColorPrimaries outputColorPrimaries,
                                                       avifTransferCharacteristics outputTransferCharacteristics, avifRGBImage * toneMappedImage,
                                                       avifContentLightLevelInformationBox * clli, avifDiagnostics * diag, void * hipStream)
{
    const avifResult ar = gainMapCheckArguments(baseImage, gainMap, hdrHeadroom, toneMappedImage, diag);
    if (ar != AVIF_RESULT_OK)
        return ar;
    if (!baseImage->pixels || !toneMappedImage->pixels || !toneMappedImage->rowBytes || !gainMap->image) {
        diagPrintf(diag, "avifhipRGBImageApplyGainMapAsync: device-resident base, gain map and tone-mapped pixels are required");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    toneMappedImage->width = baseImage->width, toneMappedImage->height = baseImage->height;
    const float weight = gainMapWeight(hdrHeadroom, gainMap);
    if (gainMapIsPlainCopy(baseImage, baseColorPrimaries, baseTransferCharacteristics, weight, outputColorPrimaries, outputTransferCharacteristics,
                           toneMappedImage)) {
        HIP_TRY(hipMemcpyAsync(toneMappedImage->pixels, baseImage->pixels, (size_t)baseImage->rowBytes * baseImage->height, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return AVIF_RESULT_OK;
    }
    return applyGainMapOnDevice(baseImage, baseColorPrimaries, baseTransferCharacteristics, gainMap, gainMap->image, weight, outputColorPrimaries,
                                outputTransferCharacteristics, toneMappedImage, clli, diag, stream);
}

// host-resident images, like the reference: the tone-mapped image's pixels are (re)allocated with malloc (src/gainmap.c:112-114)
extern "C" avifResult avifhipRGBImageApplyGainMap(const avifRGBImage * baseImage, avifColorPrimaries baseColorPrimaries,
                                                  avifTransferCharacteristics baseTransferCharacteristics, const avifGainMap * gainMap, float hdrHeadroom,
                                                  avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                                  avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag)
{
    const avifResult ar = gainMapCheckArguments(baseImage, gainMap, hdrHeadroom, toneMappedImage, diag);
    if (ar != AVIF_RESULT_OK)
        return ar;
    const uint32_t width = baseImage->width, height = baseImage->height;
    toneMappedImage->width = width, toneMappedImage->height = height;
    // avifRGBImageAllocatePixels, src/avif.c:719-737
    free(toneMappedImage->pixels);
    toneMappedImage->pixels = NULL, toneMappedImage->rowBytes = 0;
    const uint32_t outPixelBytes = rgbPixelBytes(toneMappedImage);
    if (!width || !height || width > UINT32_MAX / outPixelBytes)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t outRowBytes = width * outPixelBytes;
    toneMappedImage->pixels = (uint8_t *)malloc((size_t)outRowBytes * height);
    if (!toneMappedImage->pixels)
        return AVIF_RESULT_OUT_OF_MEMORY;
    toneMappedImage->rowBytes = outRowBytes;

    const float weight = gainMapWeight(hdrHeadroom, gainMap);
    if (gainMapIsPlainCopy(baseImage, baseColorPrimaries, baseTransferCharacteristics, weight, outputColorPrimaries, outputTransferCharacteristics,
                           toneMappedImage)) {
        memcpy(toneMappedImage->pixels, baseImage->pixels, (size_t)baseImage->rowBytes * baseImage->height); // "Copy the base image", :124-127
        return AVIF_RESULT_OK;
    }
    if (!baseImage->pixels || (weight != 0.0f && !gainMap->image))
        return AVIF_RESULT_INVALID_ARGUMENT;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    // device copies: base pixels, gain map planes, tone-mapped pixels
    avifRGBImage baseView, outView;
    memcpy(&baseView, baseImage, sizeof(avifRGBImage));
    memcpy(&outView, toneMappedImage, sizeof(avifRGBImage));
    const uint32_t baseWidthBytes = width * rgbPixelBytes(baseImage);
    baseView.rowBytes = alignUp(baseWidthBytes, 256);
    avifResult r = reserve(tls.gainMap[5], (size_t)baseView.rowBytes * height);
    if (r != AVIF_RESULT_OK)
        return r;
    baseView.pixels = (uint8_t *)tls.gainMap[5].ptr;
    HIP_TRY(hipMemcpy2DAsync(baseView.pixels, baseView.rowBytes, baseImage->pixels, baseImage->rowBytes, baseWidthBytes, height, hipMemcpyHostToDevice, tls.stream));
    outView.rowBytes = alignUp(outRowBytes, 256);
    r = reserve(tls.gainMap[0], (size_t)outView.rowBytes * height);
    if (r != AVIF_RESULT_OK)
        return r;
    outView.pixels = (uint8_t *)tls.gainMap[0].ptr;
    avifImage gainView;
    memset(&gainView, 0, sizeof(gainView));
    if (weight != 0.0f) {
        memcpy(&gainView, gainMap->image, sizeof(avifImage));
        r = stagePlanes(&gainView, true, false);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    r = applyGainMapOnDevice(&baseView, baseColorPrimaries, baseTransferCharacteristics, gainMap, &gainView, weight, outputColorPrimaries,
                             outputTransferCharacteristics, &outView, clli, diag, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;
    HIP_TRY(hipMemcpy2DAsync(toneMappedImage->pixels, outRowBytes, outView.pixels, outView.rowBytes, outRowBytes, height, hipMemcpyDeviceToHost, tls.stream));
    HIP_TRY(hipStreamSynchronize(tls.stream));
    return AVIF_RESULT_OK;
}

// avifImageApplyGainMap, src/gainmap.c:317-355: the base image arrives as YUV
extern "C" avifResult avifhipImageApplyGainMap(const avifImage * baseImage, const avifGainMap * gainMap, float hdrHeadroom,
                                               avifColorPrimaries outputColorPrimaries, avifTransferCharacteristics outputTransferCharacteristics,
                                               avifRGBImage * toneMappedImage, avifContentLightLevelInformationBox * clli, avifDiagnostics * diag)
{
    diagClear(diag);
    if (!baseImage || !gainMap)
        return AVIF_RESULT_INVALID_ARGUMENT;
    // (ICC profiles, :328-331, live in the part of avifImage / avifGainMap this library does not read: the caller checks them)
    avifRGBImage baseRgb; // avifRGBImageSetDefaults + avifRGBImageAllocatePixels, :333-335
    memset(&baseRgb, 0, sizeof(baseRgb));
    baseRgb.width = baseImage->width, baseRgb.height = baseImage->height, baseRgb.depth = baseImage->depth, baseRgb.format = AVIF_RGB_FORMAT_RGBA;
    baseRgb.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, baseRgb.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    baseRgb.maxThreads = 1;
    const uint32_t pixelBytes = rgbPixelBytes(&baseRgb);
    if (!baseRgb.width || !baseRgb.height || baseRgb.width > UINT32_MAX / pixelBytes)
        return AVIF_RESULT_INVALID_ARGUMENT;
    baseRgb.rowBytes = baseRgb.width * pixelBytes;
    baseRgb.pixels = (uint8_t *)malloc((size_t)baseRgb.rowBytes * baseRgb.height);
    if (!baseRgb.pixels)
        return AVIF_RESULT_OUT_OF_MEMORY;
    avifResult r = avifhipImageYUVToRGB(baseImage, &baseRgb);
    if (r == AVIF_RESULT_OK)
        r = avifhipRGBImageApplyGainMap(&baseRgb, baseImage->colorPrimaries, baseImage->transferCharacteristics, gainMap, hdrHeadroom, outputColorPrimaries,
                                        outputTransferCharacteristics, toneMappedImage, clli, diag);
    free(baseRgb.pixels);
    return r;
}

// ---- gain-map computation (the encode side), reference src/gainmap.c:535-843 ----

namespace {

// device planes (Y, U, V, A) of a wxh image in one scratch buffer, 256-byte row pitch
avifResult deviceGainMapPlanes(avifImage * view, uint32_t width, uint32_t height, Scratch & scratch)
{
    view->width = width, view->height = height;
    const PlaneDims d = planeDims(width, height, (int)view->yuvFormat);
    const size_t bps = (view->depth > 8) ? 2 : 1;
    size_t offset[4], total = 0;
    uint32_t pitch[4];
    for (int p = 0; p < 4; ++p) {
        const bool present = !((p == 1 || p == 2) && view->yuvFormat == AVIF_PIXEL_FORMAT_YUV400);
        pitch[p] = present ? alignUp((uint32_t)(d.w[p] * bps), 256) : 0;
        offset[p] = total, total += (size_t)pitch[p] * d.h[p];
    }
    const avifResult r = reserve(scratch, total);
    if (r != AVIF_RESULT_OK)
        return r;
    for (int p = 0; p < 4; ++p) {
        uint8_t * ptr = pitch[p] ? (uint8_t *)scratch.ptr + offset[p] : nullptr;
        if (p < 3)
            view->yuvPlanes[p] = ptr, view->yuvRowBytes[p] = pitch[p];
        else
            view->alphaPlane = ptr, view->alphaRowBytes = pitch[p];
    }
    return AVIF_RESULT_OK;
}

void freeHostPlanes(avifImage * image) // avifImageFreePlanes(AVIF_PLANES_ALL), src/avif.c:492-517
{
    if (image->imageOwnsYUVPlanes)
        for (int p = 0; p < 3; ++p)
            free(image->yuvPlanes[p]);
    for (int p = 0; p < 3; ++p)
        image->yuvPlanes[p] = NULL, image->yuvRowBytes[p] = 0;
    image->imageOwnsYUVPlanes = AVIF_FALSE;
    if (image->imageOwnsAlphaPlane)
        free(image->alphaPlane);
    image->alphaPlane = NULL, image->alphaRowBytes = 0, image->imageOwnsAlphaPlane = AVIF_FALSE;
}

} // namespace

// Host images in, gain-map metadata and (malloc'ed) gain-map planes out, like the reference.
extern "C" avifResult avifhipRGBImageComputeGainMap(const avifRGBImage * baseRgbImage, avifColorPrimaries baseColorPrimaries,
                                                    avifTransferCharacteristics baseTransferCharacteristics, const avifRGBImage * altRgbImage,
                                                    avifColorPrimaries altColorPrimaries, avifTransferCharacteristics altTransferCharacteristics,
                                                    avifGainMap * gainMap, avifDiagnostics * diag)
{
    diagClear(diag);
    if (baseRgbImage == NULL || altRgbImage == NULL || gainMap == NULL || gainMap->image == NULL)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (baseRgbImage->width != altRgbImage->width || baseRgbImage->height != altRgbImage->height) {
        diagPrintf(diag, "Both images should have the same dimensions");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    avifImage * gmImage = gainMap->image;
    if (gmImage->width == 0 || gmImage->height == 0 || gmImage->depth == 0 || (int)gmImage->yuvFormat <= (int)AVIF_PIXEL_FORMAT_NONE ||
        (int)gmImage->yuvFormat > (int)AVIF_PIXEL_FORMAT_YUV400) {
        diagPrintf(diag, "gainMap->image should be non null with desired width, height, depth and yuvFormat set");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const bool colorSpacesDiffer = baseColorPrimaries != altColorPrimaries;
    int mathPrimaries = 0;
    if (!gainMapChooseMathPrimaries(baseColorPrimaries, altColorPrimaries, &mathPrimaries))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    const uint32_t width = baseRgbImage->width, height = baseRgbImage->height;
    GainMapComputeArgs A;
    memset(&A, 0, sizeof(A));
    if (!gainMapLayout(baseRgbImage, &A.baseL) || !gainMapLayout(altRgbImage, &A.altL)) {
        diagPrintf(diag, "Unsupported RGB color space");
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
    if (!width || !height || !baseRgbImage->pixels || !altRgbImage->pixels || gmImage->depth > 16)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const size_t numPixels = (size_t)width * height;
    const bool singleChannel = gmImage->yuvFormat == AVIF_PIXEL_FORMAT_YUV400;
    const int channels = singleChannel ? 1 : 3;
    avifResult r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    hipStream_t stream = tls.stream;
    tls.gainMapCache.valid = false; // (the apply path's tables are not touched, but keep the two paths independent of call order)

    // avifGainMapSetEncodingDefaults, :18-30
    for (int i = 0; i < 3; ++i) {
        gainMap->gainMapMin[i].n = 1, gainMap->gainMapMin[i].d = 1, gainMap->gainMapMax[i].n = 1, gainMap->gainMapMax[i].d = 1;
        gainMap->baseOffset[i].n = 1, gainMap->baseOffset[i].d = 64, gainMap->alternateOffset[i].n = 1, gainMap->alternateOffset[i].d = 64;
        gainMap->gainMapGamma[i].n = 1, gainMap->gainMapGamma[i].d = 1;
    }
    gainMap->baseHdrHeadroom.n = 0, gainMap->baseHdrHeadroom.d = 1, gainMap->alternateHdrHeadroom.n = 1, gainMap->alternateHdrHeadroom.d = 1;
    gainMap->useBaseColorSpace = (mathPrimaries == (int)baseColorPrimaries) ? AVIF_TRUE : AVIF_FALSE;

    if (colorSpacesDiffer) {
        const bool ok = gainMap->useBaseColorSpace ? gainMapPrimariesMatrix(altColorPrimaries, baseColorPrimaries, A.M)
                                                   : gainMapPrimariesMatrix(baseColorPrimaries, altColorPrimaries, A.M);
        if (!ok) {
            diagPrintf(diag, "Unsupported RGB color space conversion");
            return AVIF_RESULT_NOT_IMPLEMENTED;
        }
        A.convertAlt = gainMap->useBaseColorSpace ? 1 : 0, A.convertBase = gainMap->useBaseColorSpace ? 0 : 1;
    }
    A.singleChannel = singleChannel ? 1 : 0;
    gainMapYCoefficients(mathPrimaries, A.yCoeffs);
    float baseOffset[3], altOffset[3];
    for (int c = 0; c < 3; ++c)
        baseOffset[c] = fractionToFloat(gainMap->baseOffset[c]), altOffset[c] = fractionToFloat(gainMap->alternateOffset[c]);

    // ---- device copies of the two images, lookup tables ----
    A.width = width, A.height = height;
    const uint32_t baseWidthBytes = width * rgbPixelBytes(baseRgbImage), altWidthBytes = width * rgbPixelBytes(altRgbImage);
    A.basePitch = alignUp(baseWidthBytes, 256), A.altPitch = alignUp(altWidthBytes, 256);
    if ((r = reserve(tls.gainMap[5], (size_t)A.basePitch * height)) != AVIF_RESULT_OK || (r = reserve(tls.gainMap[9], (size_t)A.altPitch * height)) != AVIF_RESULT_OK)
        return r;
    A.base = (const uint8_t *)tls.gainMap[5].ptr, A.alt = (const uint8_t *)tls.gainMap[9].ptr;
    HIP_TRY(hipMemcpy2DAsync(tls.gainMap[5].ptr, A.basePitch, baseRgbImage->pixels, baseRgbImage->rowBytes, baseWidthBytes, height, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpy2DAsync(tls.gainMap[9].ptr, A.altPitch, altRgbImage->pixels, altRgbImage->rowBytes, altWidthBytes, height, hipMemcpyHostToDevice, stream));
    std::vector<float> tables = gainMapLinearLut(baseTransferCharacteristics, baseRgbImage->depth, baseRgbImage->isFloat != 0);
    const size_t altLutOffset = tables.size();
    {
        const std::vector<float> alt = gainMapLinearLut(altTransferCharacteristics, altRgbImage->depth, altRgbImage->isFloat != 0);
        tables.insert(tables.end(), alt.begin(), alt.end());
    }
    // room for the step tables that follow (3 channels x at most 65536 entries)
    const size_t stepsOffset = (tables.size() + 3) & ~(size_t)3, stepsCapacity = (size_t)3 * 65536;
    if ((r = reserve(tls.gainMap[6], (stepsOffset + stepsCapacity) * sizeof(float))) != AVIF_RESULT_OK)
        return r;
    if ((r = uploadTableAsync(tls.gainMap[6].ptr, tables.data(), tables.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
        return r;
    float * deviceTables = (float *)tls.gainMap[6].ptr;
    A.baseLut = deviceTables, A.altLut = deviceTables + altLutOffset;
    if ((r = reserve(tls.gainMap[7], (size_t)channels * numPixels * sizeof(float))) != AVIF_RESULT_OK ||
        (r = reserve(tls.gainMap[3], (size_t)kGainMapMaxGroups * 8 * sizeof(float))) != AVIF_RESULT_OK)
        return r;
    A.ratios = (float *)tls.gainMap[7].ptr, A.partials = (float *)tls.gainMap[3].ptr;
    const uint32_t tiles = ((width + 63) / 64) * ((height + 3) / 4);
    const uint32_t groups = tiles < kGainMapMaxGroups ? tiles : kGainMapMaxGroups;
    std::vector<float> partials((size_t)groups * 8);

    // ---- pass 0: offsets that keep the converted side's channels positive, :618-660 ----
    if (colorSpacesDiffer) {
        hipError_t e = launchGainMapChannelMin(A, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map channel-minimum kernel launch");
        HIP_TRY(hipMemcpyAsync(partials.data(), A.partials, partials.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float channelMin[3] = { 0.0f, 0.0f, 0.0f };
        for (uint32_t g = 0; g < groups; ++g)
            for (int c = 0; c < 3; ++c)
                channelMin[c] = (channelMin[c] < partials[(size_t)g * 8 + c]) ? channelMin[c] : partials[(size_t)g * 8 + c];
        for (int c = 0; c < 3; ++c) {
            const float maxOffset = 0.1f;
            if (channelMin[c] < -1e-10f) {
                if (gainMap->useBaseColorSpace) {
                    const float o = altOffset[c] - channelMin[c];
                    altOffset[c] = (o < maxOffset) ? o : maxOffset;
                } else {
                    const float o = baseOffset[c] - channelMin[c];
                    baseOffset[c] = (o < maxOffset) ? o : maxOffset;
                }
            }
        }
    }
    for (int c = 0; c < 3; ++c)
        A.baseOffset[c] = baseOffset[c], A.altOffset[c] = altOffset[c];

    // ---- pass 1: ratios, maxima, extreme ratios, :662-715 ----
    {
        hipError_t e = launchGainMapRatios(A, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map ratio kernel launch");
        HIP_TRY(hipMemcpyAsync(partials.data(), A.partials, partials.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    float baseMax = 1.0f, altMax = 1.0f, minRatio[3] = { INFINITY, INFINITY, INFINITY }, maxRatio[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t g = 0; g < groups; ++g) {
        const float * p = &partials[(size_t)g * 8];
        baseMax = fmaxf(baseMax, p[0]), altMax = fmaxf(altMax, p[1]);
        for (int c = 0; c < channels; ++c)
            minRatio[c] = fminf(minRatio[c], p[2 + c]), maxRatio[c] = fmaxf(maxRatio[c], p[5 + c]);
    }
    const float kEps = 1e-10f;
    const double baseHeadroom = log2f(baseMax > kEps ? baseMax : kEps), alternateHeadroom = log2f(altMax > kEps ? altMax : kEps);
    if (!gainMapDoubleToUnsignedFraction(baseHeadroom, &gainMap->baseHdrHeadroom.n, &gainMap->baseHdrHeadroom.d) ||
        !gainMapDoubleToUnsignedFraction(alternateHeadroom, &gainMap->alternateHdrHeadroom.n, &gainMap->alternateHdrHeadroom.d))
        return AVIF_RESULT_INVALID_ARGUMENT;
    const float sign = (alternateHeadroom < baseHeadroom) ? -1.0f : 1.0f; // :728-739

    // ---- pass 2: range without outliers, :741-749 ----
    GainMapChannelRange ranges[3];
    GainMapStepTable stepTables[3];
    memset(stepTables, 0, sizeof(stepTables));
    float minLog2[3] = { 0.0f, 0.0f, 0.0f }, maxLog2[3] = { 0.0f, 0.0f, 0.0f };
    bool anyHistogram = false;
    std::vector<float> hostSteps;
    size_t histogramOffset[3] = { 0, 0, 0 }, histogramTotal = 0;
    for (int c = 0; c < channels; ++c) {
        ranges[c] = gainMapChannelRange(sign, minRatio[c], maxRatio[c], numPixels);
        minLog2[c] = ranges[c].lo, maxLog2[c] = ranges[c].hi;
        if (ranges[c].numBuckets > 0) {
            uint32_t entries = 0;
            const std::vector<float> steps = gainMapBucketSteps(ranges[c], &entries);
            stepTables[c].steps = deviceTables + stepsOffset + hostSteps.size();
            stepTables[c].entries = entries, stepTables[c].flipped = sign < 0 ? 1 : 0, stepTables[c].flip = (uint32_t)ranges[c].numBuckets - 1;
            hostSteps.insert(hostSteps.end(), steps.begin(), steps.end());
            histogramOffset[c] = histogramTotal, histogramTotal += (size_t)ranges[c].numBuckets;
            anyHistogram = true;
        }
    }
    if (anyHistogram) {
        if ((r = reserve(tls.gainMap[8], histogramTotal * sizeof(uint32_t))) != AVIF_RESULT_OK)
            return r;
        if ((r = uploadTableAsync(deviceTables + stepsOffset, hostSteps.data(), hostSteps.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
            return r;
        HIP_TRY(hipMemsetAsync(tls.gainMap[8].ptr, 0, histogramTotal * sizeof(uint32_t), stream));
        uint32_t * histograms[3];
        for (int c = 0; c < 3; ++c)
            histograms[c] = (uint32_t *)tls.gainMap[8].ptr + histogramOffset[c];
        const hipError_t e = launchGainMapHistogram(A.ratios, numPixels, channels, stepTables, histograms, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map histogram kernel launch");
        std::vector<uint32_t> hostHistograms(histogramTotal);
        HIP_TRY(hipMemcpyAsync(hostHistograms.data(), tls.gainMap[8].ptr, histogramTotal * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int c = 0; c < channels; ++c)
            if (ranges[c].numBuckets > 0)
                gainMapRangeWithoutOutliers(ranges[c], hostHistograms.data() + histogramOffset[c], &minLog2[c], &maxLog2[c]);
    }
    for (int c = 0; c < 3; ++c) { // metadata, :751-760
        const int k = singleChannel ? 0 : c;
        if (!gainMapDoubleToFraction(minLog2[k], &gainMap->gainMapMin[c].n, &gainMap->gainMapMin[c].d) ||
            !gainMapDoubleToFraction(maxLog2[k], &gainMap->gainMapMax[c].n, &gainMap->gainMapMax[c].d) ||
            !gainMapDoubleToFraction(altOffset[c], &gainMap->alternateOffset[c].n, &gainMap->alternateOffset[c].d) ||
            !gainMapDoubleToFraction(baseOffset[c], &gainMap->baseOffset[c].n, &gainMap->baseOffset[c].d))
            return AVIF_RESULT_INVALID_ARGUMENT;
    }

    // ---- pass 3: [min, max] -> codes -> RGBA -> YUV (-> requested size), :762-829 ----
    hostSteps.clear();
    memset(stepTables, 0, sizeof(stepTables));
    const uint32_t depth = gmImage->depth;
    for (int c = 0; c < channels; ++c) {
        const float range = (maxLog2[c] - minLog2[c] > 0.0f) ? maxLog2[c] - minLog2[c] : 0.0f;
        if (range == 0.0f)
            continue; // every value becomes 0, :766-773
        const std::vector<float> steps = gainMapCodeSteps(ranges[c], minLog2[c], maxLog2[c], fractionToFloat(gainMap->gainMapGamma[c]), depth);
        stepTables[c].steps = deviceTables + stepsOffset + hostSteps.size();
        stepTables[c].entries = (uint32_t)steps.size(), stepTables[c].flipped = sign < 0 ? 1 : 0, stepTables[c].flip = (1u << depth) - 1;
        hostSteps.insert(hostSteps.end(), steps.begin(), steps.end());
    }
    if (!hostSteps.empty() && (r = uploadTableAsync(deviceTables + stepsOffset, hostSteps.data(), hostSteps.size() * sizeof(float), stream)) != AVIF_RESULT_OK)
        return r;
    avifRGBImage rgbGain; // avifRGBImageSetDefaults, src/avif.c:700-717
    memset(&rgbGain, 0, sizeof(rgbGain));
    rgbGain.width = width, rgbGain.height = height, rgbGain.depth = depth, rgbGain.format = AVIF_RGB_FORMAT_RGBA;
    rgbGain.chromaUpsampling = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, rgbGain.chromaDownsampling = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC;
    rgbGain.maxThreads = 1;
    rgbGain.rowBytes = alignUp(width * 4 * ((depth > 8) ? 2 : 1), 256);
    if ((r = reserve(tls.gainMap[1], (size_t)rgbGain.rowBytes * height)) != AVIF_RESULT_OK)
        return r;
    rgbGain.pixels = (uint8_t *)tls.gainMap[1].ptr;
    {
        const hipError_t e = launchGainMapQuantise(A.ratios, width, height, channels, stepTables, rgbGain.pixels, rgbGain.rowBytes, depth, stream);
        if (e != hipSuccess)
            return hipFailed(e, "gain map quantisation kernel launch");
    }
    const uint32_t requestedWidth = gmImage->width, requestedHeight = gmImage->height;
    freeHostPlanes(gmImage);
    avifImage deviceGain;
    memcpy(&deviceGain, gmImage, sizeof(avifImage));
    if ((r = deviceGainMapPlanes(&deviceGain, width, height, tls.gainMap[10])) != AVIF_RESULT_OK)
        return r;
    if ((r = avifhipImageRGBToYUVAsync(&deviceGain, &rgbGain, stream)) != AVIF_RESULT_OK)
        return r;
    avifImage deviceFinal;
    memcpy(&deviceFinal, &deviceGain, sizeof(avifImage));
    if (requestedWidth != width || requestedHeight != height) {
        if ((r = deviceGainMapPlanes(&deviceFinal, requestedWidth, requestedHeight, tls.gainMap[4])) != AVIF_RESULT_OK)
            return r;
        if ((r = avifhipImageScaleAsync(&deviceGain, &deviceFinal, stream)) != AVIF_RESULT_OK)
            return r;
    }
    gmImage->width = deviceFinal.width, gmImage->height = deviceFinal.height;
    if ((r = allocateHostPlanes(gmImage, true)) != AVIF_RESULT_OK) {
        freeHostPlanes(gmImage);
        return r;
    }
    const PlaneGeometry g = planeGeometry(gmImage);
    for (int p = 0; p < 4; ++p) {
        uint8_t * host = (p < 3) ? gmImage->yuvPlanes[p] : gmImage->alphaPlane;
        const uint8_t * dev = (p < 3) ? deviceFinal.yuvPlanes[p] : deviceFinal.alphaPlane;
        if (!host || !dev)
            continue;
        HIP_TRY(hipMemcpy2DAsync(host, (p < 3) ? gmImage->yuvRowBytes[p] : gmImage->alphaRowBytes, dev, (p < 3) ? deviceFinal.yuvRowBytes[p] : deviceFinal.alphaRowBytes,
                                 g.widthBytes[p], g.rows[p], hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    tls.lastKernel = "gainmap_compute";
    return AVIF_RESULT_OK;
}

// =================================================================================================
// plane scaling, reference src/scale.c:23-201
// =================================================================================================

namespace {

inline size_t colTablePad(size_t n)
{
    return ((n + 15) & ~(size_t)15) + 16;
}

// second source column of destination column i (the last source column of a box)
inline int scaleSecondColumn(const ScaleSchedule & S, size_t i, int srcW)
{
    const int a = S.colA[i];
    return (S.mode == SCALE_UP2) ? S.colB[i] : (S.mode == SCALE_BOX) ? a + S.colB[i] - 1 : (S.mode == SCALE_POINT) ? a : (a + 1 < srcW ? a + 1 : srcW - 1);
}

// Parameters of an LDS-staged kernel whose waves own `cols` destination columns (kernels.h: ScaleStaging; 256: the row-staged
// kernel, 1024: the window kernel); rowsPerWave = 0 when the block a wave must stage cannot fit.
ScaleStaging scaleStagedPlan(const ScaleSchedule & S, int srcW, bool wide, size_t cols)
{
    ScaleStaging none, st;
    const int bps = wide ? 2 : 1;
    const size_t n = S.colA.size(), rows = S.rowA.size();
    int64_t segBytes = 0;
    for (size_t i0 = 0; i0 < n; i0 += cols) {
        int lo = INT32_MAX, hi = 0;
        for (size_t i = i0; i < n && i < i0 + cols; ++i) {
            const int a = S.colA[i], b = scaleSecondColumn(S, i, srcW);
            lo = a < lo ? a : lo, lo = b < lo ? b : lo;
            hi = a > hi ? a : hi, hi = b > hi ? b : hi;
        }
        const int64_t bytes = (int64_t)(hi - lo + 1) * bps;
        segBytes = bytes > segBytes ? bytes : segBytes;
    }
    if (segBytes > kScaleStageBytes / 2)
        return none;
    st.segPitch = (uint32_t)((segBytes + 15 + 15) & ~(int64_t)15) + 16;
    if (S.mode == SCALE_BOX && !wide) // 8-bit box rows are summed in 16 bits by the reference: no wrap up to 257 rows
        for (int rb : S.rowB)
            if (rb > 257)
                return none;
    const int maxRows = (int)(kScaleStageBytes / st.segPitch);
    const size_t segs = (n + cols - 1) / cols;
    // the largest rows-per-wave whose staged block fits and that still leaves >= ~1024 workgroups (or 1)
    for (int rpw = 16; rpw >= 1; rpw >>= 1) {
        if (rpw > 1 && segs * ((rows + 4 * rpw - 1) / (4 * rpw)) < 1024)
            continue;
        int cap = 0;
        for (size_t j0 = 0; j0 < rows; j0 += rpw) {
            int lo = INT32_MAX, hi = 0;
            for (size_t j = j0; j < rows && j < j0 + rpw; ++j) {
                const int a = S.rowA[j];
                const int b = (S.mode == SCALE_BOX) ? a + S.rowB[j] - 1 : (S.mode == SCALE_POINT) ? a : S.rowB[j];
                lo = a < lo ? a : lo, lo = b < lo ? b : lo;
                hi = a > hi ? a : hi, hi = b > hi ? b : hi;
            }
            cap = (hi - lo + 1) > cap ? (hi - lo + 1) : cap;
        }
        if (cap <= maxRows) {
            st.rowsPerWave = rpw, st.rowsCap = cap;
            return st;
        }
    }
    return none;
}

// the window kernel's extra conditions: 8-bit samples, not a box, every aligned group of 4 destination columns reads within 8
// source columns (groups past the last column repeat it, like the padded column tables)
bool scaleWindowCovers(const ScaleSchedule & S, int srcW, bool wide)
{
    if (wide || S.mode == SCALE_BOX || srcW < 8)
        return false;
    const size_t n = S.colA.size();
    for (size_t i0 = 0; i0 < n; i0 += 4) {
        int lo = INT32_MAX, hi = 0;
        for (size_t k = 0; k < 4; ++k) {
            const size_t i = (i0 + k < n) ? i0 + k : n - 1;
            const int a = S.colA[i], b = scaleSecondColumn(S, i, srcW);
            lo = a < lo ? a : lo, lo = b < lo ? b : lo;
            hi = a > hi ? a : hi, hi = b > hi ? b : hi;
        }
        if (hi - lo + 1 > 8)
            return false;
    }
    return true;
}

} // namespace

extern "C" avifResult avifhipImageScaleAsync(const avifImage * src, avifImage * dst, void * hipStream)
{
    if (!src || !dst || !dst->width || !dst->height)
        return AVIF_RESULT_INVALID_ARGUMENT; // src/scale.c:35-38
    if (src->depth != dst->depth || src->yuvFormat != dst->yuvFormat)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if ((src->yuvPlanes[0] || src->alphaPlane) && (src->width > 16384 || src->height > 16384))
        return AVIF_RESULT_NOT_IMPLEMENTED; // "invalid width/height scale for libyuv", src/scale.c:66-80
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    const bool wide = src->depth > 8;
    const PlaneDims sd = planeDims(src->width, src->height, (int)src->yuvFormat), dd = planeDims(dst->width, dst->height, (int)dst->yuvFormat);
    // The schedules of every plane live in one per-thread device table (successive calls of one thread are ordered by the
    // stream they share, like the grid table).  Building and uploading them is O(width + height) host work plus one small
    // copy -- as long as one kernel -- so the table of the last geometry is kept: a sequence of frames, or the tiles of a
    // grid, scaled to the same size pay for it once.
    bool present[4] = { false, false, false, false };
    for (int p = 0; p < 4; ++p) {
        const uint8_t * sp = (p < 3) ? src->yuvPlanes[p] : src->alphaPlane;
        uint8_t * dp = (p < 3) ? dst->yuvPlanes[p] : dst->alphaPlane;
        if (!sp || ((p == 1 || p == 2) && src->yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
            continue;
        if (!dp) {
            setError("avifhipImageScaleAsync: destination plane %d is missing", p);
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        present[p] = true;
    }
    if (!present[0] && !present[1] && !present[2] && !present[3])
        return AVIF_RESULT_OK;
    ScaleTableCache & cache = tls.scaleCache;
    // (the stream is part of the key: a table uploaded on one stream is only ordered before kernels of that stream)
    const uint32_t key[9] = { src->width, src->height, dst->width, dst->height, (uint32_t)src->yuvFormat, wide ? 1u : 0u,
                              (uint32_t)(present[0] | (present[1] << 1) | (present[2] << 2) | (present[3] << 3)),
                              (uint32_t)(uintptr_t)stream, (uint32_t)((uint64_t)(uintptr_t)stream >> 32) };
    if (!cache.valid || memcmp(cache.key, key, sizeof(key)) != 0) {
        cache.valid = false;
        std::vector<int32_t> tables;
        for (int p = 0; p < 4; ++p) {
            if (!present[p])
                continue;
            const ScaleSchedule sched = makeScaleSchedule(sd.w[p], sd.h[p], dd.w[p], dd.h[p], wide);
            cache.offset[p] = tables.size();
            cache.mode[p] = sched.mode;
            cache.staging[p] = scaleStagedPlan(sched, sd.w[p], wide, 256);
            cache.window[p] = ScaleStaging();
            if (scaleWindowCovers(sched, sd.w[p], wide)) { // no staging: rows per wave only amortise the prologue
                int rpw = 16;
                while (rpw > 4 && ((size_t)dd.w[p] + 255) / 256 * (((size_t)dd.h[p] + 4 * rpw - 1) / (4 * rpw)) < 2048)
                    rpw >>= 1;
                cache.window[p].rowsPerWave = rpw;
            }
            // column tables: padded to a multiple of 16 entries + 16 with copies of the last entry (a lane of the window kernel
            // reads the entries of its 16 columns unclamped); row tables: to a multiple of 4; every table starts 16-byte aligned
            int which = 0;
            for (const std::vector<int32_t> * v : { &sched.colA, &sched.colB, &sched.rowA, &sched.rowB, &sched.rowF }) {
                tables.insert(tables.end(), v->begin(), v->end());
                const size_t padded = (which < 2) ? colTablePad(v->size()) : ((v->size() + 3) & ~(size_t)3);
                tables.insert(tables.end(), padded - v->size(), (which < 2 && !v->empty()) ? v->back() : 0);
                ++which;
            }
        }
        const avifResult rr = reserve(tls.scaleTable, tables.size() * sizeof(int32_t));
        if (rr != AVIF_RESULT_OK)
            return rr;
        const avifResult ur = uploadTableAsync(tls.scaleTable.ptr, tables.data(), tables.size() * sizeof(int32_t), stream);
        if (ur != AVIF_RESULT_OK)
            return ur;
        memcpy(cache.key, key, sizeof(key));
        cache.valid = true;
    }
    const int32_t * dev = (const int32_t *)tls.scaleTable.ptr;
    const bool staged = gTiledKernels.load(std::memory_order_relaxed) != 0;
    ScaleStagedLaunch L, W; // planes served by the row-staged kernel / by the window kernel
    L.count = W.count = 0;
    for (int p = 0; p < 4; ++p) {
        if (!present[p])
            continue;
        ScaleArgs A;
        A.src = (p < 3) ? src->yuvPlanes[p] : src->alphaPlane;
        A.dst = (p < 3) ? dst->yuvPlanes[p] : dst->alphaPlane;
        A.srcPitch = (p < 3) ? src->yuvRowBytes[p] : src->alphaRowBytes;
        A.dstPitch = (p < 3) ? dst->yuvRowBytes[p] : dst->alphaRowBytes;
        A.srcW = sd.w[p], A.srcH = sd.h[p], A.dstW = dd.w[p], A.dstH = dd.h[p];
        A.mode = cache.mode[p];
        const int32_t * t = dev + cache.offset[p];
        const size_t wPad = colTablePad((size_t)dd.w[p]), hPad = ((size_t)dd.h[p] + 3) & ~(size_t)3;
        A.colA = t, A.colB = t + wPad, A.rowA = t + 2 * wPad, A.rowB = A.rowA + hPad, A.rowF = A.rowB + hPad;
        if (staged && cache.window[p].rowsPerWave > 0) {
            W.plane[W.count] = A, W.staging[W.count] = cache.window[p];
            ++W.count;
            continue;
        }
        if (staged && cache.staging[p].rowsPerWave > 0) {
            L.plane[L.count] = A, L.staging[L.count] = cache.staging[p];
            ++L.count;
            continue;
        }
        const hipError_t e = launchScalePlane(A, wide, stream);
        if (e != hipSuccess)
            return hipFailed(e, "plane scaling kernel launch");
    }
    hipError_t le = launchScalePlanesStaged(W, wide, true, stream);
    if (le == hipSuccess)
        le = launchScalePlanesStaged(L, wide, false, stream);
    if (le != hipSuccess)
        return hipFailed(le, "plane scaling kernel launch");
    // mode of the first plane [kernel family that served it]
    static const char * names[3][5] = { { "scale_point[gather]", "scale_down[gather]", "scale_up[gather]", "scale_box[gather]", "scale_up2[gather]" },
                                        { "scale_point[staged]", "scale_down[staged]", "scale_up[staged]", "scale_box[staged]", "scale_up2[staged]" },
                                        { "scale_point[window]", "scale_down[window]", "scale_up[window]", "scale_box[window]", "scale_up2[window]" } };
    const int first = present[0] ? 0 : 3;
    const int family = !staged ? 0 : cache.window[first].rowsPerWave > 0 ? 2 : cache.staging[first].rowsPerWave > 0 ? 1 : 0;
    tls.lastKernel = names[family][cache.mode[first]];
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// in place on a host-resident image, like the reference
extern "C" avifResult avifhipImageScale(avifImage * image, uint32_t dstWidth, uint32_t dstHeight)
{
    if (!image)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (image->width == dstWidth && image->height == dstHeight)
        return AVIF_RESULT_OK; // "Nothing to do", src/scale.c:30-33
    if (!dstWidth || !dstHeight)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if ((image->yuvPlanes[0] || image->alphaPlane) && (image->width > 16384 || image->height > 16384))
        return AVIF_RESULT_NOT_IMPLEMENTED;
    avifResult r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    const size_t bps = (image->depth > 8) ? 2 : 1;
    const PlaneDims sd = planeDims(image->width, image->height, (int)image->yuvFormat), dd = planeDims(dstWidth, dstHeight, (int)image->yuvFormat);
    avifImage srcView, dstView;
    memcpy(&srcView, image, sizeof(avifImage));
    memcpy(&dstView, image, sizeof(avifImage));
    dstView.width = dstWidth, dstView.height = dstHeight;
    // stage the source planes, reserve the destination planes (one device buffer: [sources][destinations])
    size_t srcOff[4], dstOff[4], total = 0;
    uint32_t srcPitch[4], dstPitch[4];
    bool present[4];
    for (int p = 0; p < 4; ++p) {
        const uint8_t * sp = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        present[p] = sp && !((p == 1 || p == 2) && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400);
        srcOff[p] = dstOff[p] = 0, srcPitch[p] = dstPitch[p] = 0;
        if (!present[p])
            continue;
        srcPitch[p] = alignUp((uint32_t)(sd.w[p] * bps), 256), dstPitch[p] = alignUp((uint32_t)(dd.w[p] * bps), 256);
        srcOff[p] = total, total += (size_t)srcPitch[p] * sd.h[p];
        dstOff[p] = total, total += (size_t)dstPitch[p] * dd.h[p];
    }
    if (total == 0) {
        image->width = dstWidth, image->height = dstHeight;
        return AVIF_RESULT_OK;
    }
    r = reserve(tls.pixels, total);
    if (r != AVIF_RESULT_OK)
        return r;
    uint8_t * base = (uint8_t *)tls.pixels.ptr;
    for (int p = 0; p < 4; ++p) {
        uint8_t ** sv = (p < 3) ? &srcView.yuvPlanes[p] : &srcView.alphaPlane;
        uint8_t ** dv = (p < 3) ? &dstView.yuvPlanes[p] : &dstView.alphaPlane;
        uint32_t * svp = (p < 3) ? &srcView.yuvRowBytes[p] : &srcView.alphaRowBytes;
        uint32_t * dvp = (p < 3) ? &dstView.yuvRowBytes[p] : &dstView.alphaRowBytes;
        if (!present[p]) {
            *sv = *dv = nullptr;
            continue;
        }
        const uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        const uint32_t hostPitch = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
        HIP_TRY(hipMemcpy2DAsync(base + srcOff[p], srcPitch[p], host, hostPitch, sd.w[p] * bps, sd.h[p], hipMemcpyHostToDevice, tls.stream));
        *sv = base + srcOff[p], *svp = srcPitch[p];
        *dv = base + dstOff[p], *dvp = dstPitch[p];
    }
    r = avifhipImageScaleAsync(&srcView, &dstView, tls.stream);
    if (r != AVIF_RESULT_OK)
        return r;
    // new planes: malloc'ed with tight rows like avifImageAllocatePlanes (src/avif.c:431-490)
    uint8_t * fresh[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int p = 0; p < 4; ++p) {
        if (!present[p])
            continue;
        fresh[p] = (uint8_t *)malloc((size_t)dd.w[p] * bps * dd.h[p]);
        if (!fresh[p]) {
            for (int q = 0; q < p; ++q)
                free(fresh[q]);
            (void)hipStreamSynchronize(tls.stream);
            return AVIF_RESULT_OUT_OF_MEMORY;
        }
        HIP_TRY(hipMemcpy2DAsync(fresh[p], dd.w[p] * bps, base + dstOff[p], dstPitch[p], dd.w[p] * bps, dd.h[p], hipMemcpyDeviceToHost, tls.stream));
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    for (int p = 0; p < 4; ++p) {
        if (!present[p])
            continue;
        uint8_t ** plane = (p < 3) ? &image->yuvPlanes[p] : &image->alphaPlane;
        uint32_t * pitch = (p < 3) ? &image->yuvRowBytes[p] : &image->alphaRowBytes;
        const bool owned = (p < 3) ? image->imageOwnsYUVPlanes : image->imageOwnsAlphaPlane;
        if (owned)
            free(*plane); // src/scale.c:186-193 (avifFree is free, src/mem.c)
        *plane = fresh[p], *pitch = (uint32_t)(dd.w[p] * bps);
    }
    if (image->yuvPlanes[0])
        image->imageOwnsYUVPlanes = AVIF_TRUE;
    if (image->alphaPlane)
        image->imageOwnsAlphaPlane = AVIF_TRUE;
    image->width = dstWidth, image->height = dstHeight;
    return AVIF_RESULT_OK;
}

// =================================================================================================
// application-side pixel transforms, reference apps/shared/avifutil.c:667-825
// =================================================================================================

extern "C" avifResult avifhipRGBImageTransformAsync(avifRGBImage * dst, const avifRGBImage * src, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                                    avifBool mirror, uint8_t axis, void * hipStream)
{
    if (!dst || !src || !dst->pixels || !src->pixels)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if ((rotate && angle > 3) || (mirror && axis > 1))
        return AVIF_RESULT_INVALID_ARGUMENT; // "Invalid angle." / "Invalid axis value.", apps/shared/avifutil.c:741,781
    if (dst->format != src->format || dst->depth != src->depth)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifCropRect whole = { 0, 0, src->width, src->height };
    const avifCropRect & r = crop ? *crop : whole;
    if (r.width > src->width || r.height > src->height || r.x > src->width - r.width || r.y > src->height - r.height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    TransformArgs A;
    memset(&A, 0, sizeof(A));
    const uint32_t px = rgbPixelBytes(src);
    A.angle = (rotate && angle != 0) ? angle : 0; // :805
    A.mirror = mirror ? (int32_t)axis : -1;
    A.cw = r.width, A.ch = r.height;
    A.dw = (A.angle & 1) ? r.height : r.width, A.dh = (A.angle & 1) ? r.width : r.height; // :692-693
    if (dst->width != A.dw || dst->height != A.dh || (uint64_t)dst->rowBytes < (uint64_t)A.dw * px)
        return AVIF_RESULT_INVALID_ARGUMENT;
    A.src = src->pixels + (size_t)r.y * src->rowBytes + (size_t)r.x * px; // avifRGBImageSetViewRect, :677-680
    A.dst = dst->pixels;
    A.srcPitch = src->rowBytes, A.dstPitch = dst->rowBytes;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    tls.lastKernel = (A.angle & 1) ? "rgb_transform_transpose" : "rgb_transform_rows";
    const hipError_t e = launchRgbTransform(A, px, pickStream(hipStream));
    if (e != hipSuccess)
        return hipFailed(e, "pixel transform kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// =================================================================================================
// integer range helpers, reference src/reformat.c:1750-1840
// =================================================================================================

namespace {
struct RangeRow
{
    int lo, hiY, hiUV, full;
};
const RangeRow * rangeRow(uint32_t depth)
{
    static const RangeRow rows[3] = { { 16, 235, 240, 255 }, { 64, 940, 960, 1023 }, { 256, 3760, 3840, 4095 } };
    switch (depth) {
        case 8: return &rows[0];
        case 10: return &rows[1];
        case 12: return &rows[2];
        default: return nullptr;
    }
}
int clampHost(int v, int lo, int hi)
{
    return v < lo ? lo : (hi < v ? hi : v);
}
int limitedToFull(int v, int lo, int hi, int full)
{
    return clampHost((((v - lo) * full) + ((hi - lo) / 2)) / (hi - lo), 0, full);
}
int fullToLimited(int v, int lo, int hi, int full)
{
    return clampHost((((v * (hi - lo)) + (full / 2)) / full) + lo, lo, hi);
}
} // namespace

extern "C" int avifhipLimitedToFullY(uint32_t depth, int v)
{
    const RangeRow * r = rangeRow(depth);
    return r ? limitedToFull(v, r->lo, r->hiY, r->full) : v;
}
extern "C" int avifhipLimitedToFullUV(uint32_t depth, int v)
{
    const RangeRow * r = rangeRow(depth);
    return r ? limitedToFull(v, r->lo, r->hiUV, r->full) : v;
}
extern "C" int avifhipFullToLimitedY(uint32_t depth, int v)
{
    const RangeRow * r = rangeRow(depth);
    return r ? fullToLimited(v, r->lo, r->hiY, r->full) : v;
}
extern "C" int avifhipFullToLimitedUV(uint32_t depth, int v)
{
    const RangeRow * r = rangeRow(depth);
    return r ? fullToLimited(v, r->lo, r->hiUV, r->full) : v;
}

// =================================================================================================
// library control, device memory helpers, timing
// =================================================================================================

extern "C" void avifhipCalcYUVCoefficients(const avifImage * image, float * outR, float * outG, float * outB)
{
    calcYuvCoefficients(image, outR, outG, outB);
}

extern "C" void avifhipSetArithmetic(avifhipArithmetic mode)
{
    gArithmetic.store((int)mode, std::memory_order_relaxed);
}
extern "C" avifhipArithmetic avifhipGetArithmetic(void)
{
    return (avifhipArithmetic)effectiveArithmetic();
}
extern "C" void avifhipSetTuning(uint32_t bits)
{
    gTuning.store(bits, std::memory_order_relaxed);
}
extern "C" void avifhipSetTiledKernels(int enabled)
{
    gTiledKernels.store(enabled ? 1 : 0, std::memory_order_relaxed);
}

extern "C" int avifhipDeviceCount(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

extern "C" avifResult avifhipSetDevice(int device)
{
    if (tls.stream && tls.device != device) {
        setError("avifhipSetDevice: this thread's context is already bound to device %d", tls.device);
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    HIP_TRY(hipSetDevice(device));
    tls.device = device;
    return AVIF_RESULT_OK;
}

extern "C" void * avifhipStreamCreate(void)
{
    if (ensureContext() != AVIF_RESULT_OK)
        return nullptr;
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        hipFailed(e, "hipStreamCreateWithFlags");
        return nullptr;
    }
    return (void *)s;
}
extern "C" void avifhipStreamDestroy(void * hipStream)
{
    if (hipStream)
        (void)hipStreamDestroy((hipStream_t)hipStream);
}

extern "C" avifResult avifhipSynchronize(void * hipStream)
{
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    HIP_TRY(hipStreamSynchronize(pickStream(hipStream)));
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipExplainYUVToRGB(const avifImage * image, const avifRGBImage * rgb, char * text, size_t size)
{
    if (!image || !rgb || !text || !size)
        return AVIF_RESULT_INVALID_ARGUMENT;
    text[0] = 0;
    YuvToRgbPlan p;
    const avifResult r = makeYuvToRgbPlan(image, rgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &p);
    if (r != AVIF_RESULT_OK)
        return r;
    const char * alpha = "keep";
    if (p.alphaSource == ALPHA_FILL)
        alpha = "fill";
    else if (p.alphaSource == ALPHA_PLANE)
        alpha = (p.arith == ARITH_LIBYUV && p.fxAlpha == FXA_SHIFT) ? "plane-shift" : "plane-float";
    const bool tiled = gTiledKernels.load(std::memory_order_relaxed) && tileYuvToRgbSupported(p);
    snprintf(text, size, "arith=%s kernel=%s native=%d downshift=%d bilinear=%d alpha=%s inloopmul=%d postmul=%d postmulfx=%d",
             p.arith == ARITH_LIBYUV ? "libyuv" : "fp32", tiled ? "tile" : "generic", p.arith == ARITH_LIBYUV ? p.fxNative : 0,
             p.arith == ARITH_LIBYUV ? p.fxDownshift : 0, p.bilinear, alpha, p.inLoopMul, p.postMul, p.postMulFx);
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipExplainRGBToYUV(const avifImage * image, const avifRGBImage * rgb, char * text, size_t size)
{
    if (!image || !rgb || !text || !size)
        return AVIF_RESULT_INVALID_ARGUMENT;
    text[0] = 0;
    RgbToYuvPlan p;
    const avifResult r = makeRgbToYuvPlan(image, rgb, effectiveArithmetic(), &p);
    if (r != AVIF_RESULT_OK)
        return r;
    finishRgbToYuvPlan(image, rgb, &p);
    const bool tiled = gTiledKernels.load(std::memory_order_relaxed) && tileRgbToYuvSupported(p);
    snprintf(text, size, "arith=%s kernel=%s mul=%d", p.arith == ARITH_LIBYUV ? "libyuv" : "fp32", tiled ? "tile" : "generic", p.mul);
    return AVIF_RESULT_OK;
}

extern "C" const char * avifhipLastError(void)
{
    return tls.lastError;
}
extern "C" const char * avifhipLastKernel(void)
{
    return tls.lastKernel;
}
extern "C" uint64_t avifhipLaunchCount(void)
{
    return tls.launches;
}
extern "C" const char * avifhipVersion(void)
{
    return "avifhip 0.1.0 (gfx950; mirrors libavif 1.4.2 reformat path)";
}

extern "C" void * avifhipDeviceAlloc(size_t bytes)
{
    if (ensureContext() != AVIF_RESULT_OK)
        return nullptr;
    void * p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        hipFailed(e, "hipMalloc");
        return nullptr;
    }
    return p;
}
extern "C" void avifhipDeviceFree(void * devicePtr)
{
    if (devicePtr)
        (void)hipFree(devicePtr);
}
extern "C" avifResult avifhipCopyToDevice(void * devicePtr, const void * hostPtr, size_t bytes)
{
    HIP_TRY(hipMemcpy(devicePtr, hostPtr, bytes, hipMemcpyHostToDevice));
    return AVIF_RESULT_OK;
}
extern "C" avifResult avifhipCopyToHost(void * hostPtr, const void * devicePtr, size_t bytes)
{
    HIP_TRY(hipMemcpy(hostPtr, devicePtr, bytes, hipMemcpyDeviceToHost));
    return AVIF_RESULT_OK;
}
extern "C" avifResult avifhipDeviceMemset(void * devicePtr, int value, size_t bytes)
{
    HIP_TRY(hipMemset(devicePtr, value, bytes));
    return AVIF_RESULT_OK;
}

extern "C" double avifhipTimeYUVToRGB(const avifImage * image, avifRGBImage * rgb, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (avifhipImageYUVToRGBAsync(image, rgb, stream) != AVIF_RESULT_OK)
            return -1.0;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)
        return -1.0;
    (void)hipEventRecord(t0, stream);
    for (int k = 0; k < iters; ++k)
        if (avifhipImageYUVToRGBAsync(image, rgb, stream) != AVIF_RESULT_OK)
            return -1.0;
    (void)hipEventRecord(t1, stream);
    float ms = -1.0f;
    if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
        ms = -1.0f;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return ms < 0 ? -1.0 : (double)ms / iters;
}

extern "C" double avifhipTimeYUVToRGBCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || count == 0 || !images || !rgbs || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (avifhipImageYUVToRGBAsync(images[k % count], rgbs[k % count], stream) != AVIF_RESULT_OK)
            return -1.0;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)
        return -1.0;
    (void)hipEventRecord(t0, stream);
    for (int k = 0; k < iters; ++k)
        if (avifhipImageYUVToRGBAsync(images[k % count], rgbs[k % count], stream) != AVIF_RESULT_OK)
            return -1.0;
    (void)hipEventRecord(t1, stream);
    float ms = -1.0f;
    if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
        ms = -1.0f;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return ms < 0 ? -1.0 : (double)ms / iters;
}

extern "C" double avifhipTimeRGBToYUV(avifImage * image, const avifRGBImage * rgb, int warmup, int iters, void * hipStream)
{
    if (iters <= 0 || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (avifhipImageRGBToYUVAsync(image, rgb, stream) != AVIF_RESULT_OK)
            return -1.0;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)
        return -1.0;
    (void)hipEventRecord(t0, stream);
    for (int k = 0; k < iters; ++k)
        if (avifhipImageRGBToYUVAsync(image, rgb, stream) != AVIF_RESULT_OK)
            return -1.0;
    (void)hipEventRecord(t1, stream);
    float ms = -1.0f;
    if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
        ms = -1.0f;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return ms < 0 ? -1.0 : (double)ms / iters;
}

// Synthetic planes, BASELINE.md section 3 (xorshift32, one draw per sample, row-major)
extern "C" uint32_t avifhipSynthFill(uint32_t state, uint8_t * plane, uint32_t rowBytes, uint32_t width, uint32_t height,
                                     uint32_t bytesPerSample, uint32_t lo, uint32_t hi)
{
    uint32_t x = state ? state : 0x12345678u;
    const uint32_t span = hi - lo + 1;
    for (uint32_t j = 0; j < height; ++j) {
        uint8_t * row = plane + (size_t)j * rowBytes;
        for (uint32_t i = 0; i < width; ++i) {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            const uint32_t v = lo + (span ? x % span : x);
            if (bytesPerSample == 1) {
                row[i] = (uint8_t)v;
            } else {
                const uint16_t w = (uint16_t)v;
                memcpy(row + 2 * (size_t)i, &w, 2);
            }
        }
    }
    return x;
}
