// api.cpp -- the C ABI of libavifhip.so (include/avifhip.h): argument checks and error codes of
// libavif's entry points, pointer classification (host vs HBM), staging through device scratch for
// host-resident images, kernel selection, and the per-thread stream/scratch context.
#include "api_internal.h"

using namespace avifhip;
using namespace avifhip::api;

namespace avifhip {
namespace api {

std::atomic<int> gArithmetic { -1 }; // -1: not decided yet (environment, then AUTO)
std::atomic<int> gTiledKernels { 1 };
std::atomic<uint32_t> gTuning { TUNE_DEFAULT };

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


avifResult ensureContext()
{
    if (tls.stream) {
        // The context's stream and scratch belong to tls.device.  A host application that shares the thread (torch, anything
        // driving several GPUs) may have made another device current since the last call: allocations would then land on that
        // device while the kernels run on this one.  The thread is switched back (and stays there: avifhip.h, avifhipSetDevice).
        int current = -1;
        if (hipGetDevice(&current) != hipSuccess || current != tls.device)
            HIP_TRY(hipSetDevice(tls.device));
        return AVIF_RESULT_OK;
    }
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
    HIP_TRY(hipEventCreateWithFlags(&tls.scratchUsed, hipEventDisableTiming));
    return AVIF_RESULT_OK;
}

ScratchScope::ScratchScope(hipStream_t s) : stream(s), result(AVIF_RESULT_OK)
{
    if (tls.scratchPending && tls.scratchStream != s) {
        const hipError_t e = hipStreamWaitEvent(s, tls.scratchUsed, 0);
        if (e != hipSuccess)
            result = hipFailed(e, "hipStreamWaitEvent(scratch)");
    }
}

ScratchScope::~ScratchScope()
{
    if (tls.scratchUsed && hipEventRecord(tls.scratchUsed, stream) == hipSuccess) {
        tls.scratchStream = stream;
        tls.scratchPending = true;
    } else {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream); // could not mark the use: make sure nothing is pending instead
        tls.scratchPending = false;
    }
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

} // namespace api
} // namespace avifhip

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
    ScratchScope scratch(stream); // tls.table may still be read by a batch enqueued on another stream
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
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
    ScratchScope scratch(stream);
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
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
        // token types in the gaps of the enumeration (2..63, 68..127): the reference's validity check counts them as operands /
        // unary operators, but its evaluator takes every type it does not know down the binary-operator path
        // (src/sampletransform.c:313-336), whose assertions end the call with AVIF_RESULT_INTERNAL_ERROR at the latest; the kernel
        // has no such path, so they are refused here
        const bool known = type == AVIF_SAMPLE_TRANSFORM_CONSTANT || type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX ||
                           (type >= AVIF_SAMPLE_TRANSFORM_FIRST_UNARY_OPERATOR && type <= AVIF_SAMPLE_TRANSFORM_BSR) ||
                           (type >= AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR && type <= AVIF_SAMPLE_TRANSFORM_MAX);
        if (!known)
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
    ScratchScope scratch(stream);
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
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
