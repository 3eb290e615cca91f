// api.cpp -- the C ABI of libavifhip.so (include/avifhip.h): argument checks and error codes of
// libavif's entry points, pointer classification (host vs HBM), staging through device scratch for
// host-resident images, kernel selection, and the per-thread stream/scratch context.
#include "api_internal.h"

#include <algorithm>

#include <unistd.h>

using namespace avifhip;
using namespace avifhip::api;

namespace avifhip {
namespace api {

std::atomic<int> gArithmetic { -1 }; // -1: not decided yet (environment, then AUTO)
std::atomic<int> gTiledKernels { 1 };
std::atomic<uint32_t> gTuning { TUNE_DEFAULT };

// ---- download helper thread ----
CopyWorker::CopyWorker(int device, hipStream_t stream) : device_(device), stream_(stream), thread_([this] { run(); }) {}

CopyWorker::~CopyWorker()
{
    {
        std::lock_guard<std::mutex> lock(mutex_);
        stop_ = true;
    }
    wake_.notify_all();
    if (thread_.joinable())
        thread_.join();
}

void CopyWorker::post(const Job & job)
{
    {
        std::lock_guard<std::mutex> lock(mutex_);
        queue_.push_back(job);
        ++pending_;
    }
    wake_.notify_one();
}

hipError_t CopyWorker::drain()
{
    std::unique_lock<std::mutex> lock(mutex_);
    idle_.wait(lock, [this] { return pending_ == 0; });
    const hipError_t e = error_;
    error_ = hipSuccess;
    return e;
}

void CopyWorker::run()
{
    (void)hipSetDevice(device_);
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            wake_.wait(lock, [this] { return stop_ || !queue_.empty(); });
            if (queue_.empty())
                return; // stop requested and nothing left
            job = queue_.front();
            queue_.pop_front();
        }
        hipError_t e = hipStreamWaitEvent(stream_, job.after, 0);
        if (e == hipSuccess) // pageable destination: returns when the bytes have arrived
            e = hipMemcpy2DAsync(job.dst, job.dstPitch, job.src, job.srcPitch, job.widthBytes, job.rows, hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream_);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            if (e != hipSuccess && error_ == hipSuccess)
                error_ = e;
            --pending_;
        }
        idle_.notify_all();
    }
}

// ---- context pool ----
namespace {
struct ContextPool
{
    std::mutex mutex;
    std::vector<Context *> idle;
};
// never destroyed: at process exit the HIP runtime may already be gone, and the driver reclaims everything anyway
ContextPool & contextPool()
{
    static ContextPool * pool = new ContextPool;
    return *pool;
}
struct ContextLease
{
    Context * context = nullptr;
    ~ContextLease()
    {
        if (!context)
            return;
        ContextPool & pool = contextPool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        pool.idle.push_back(context); // work still pending on its streams stays ordered: the next holder uses the same streams
    }
};
thread_local ContextLease lease;

// an idle context bound to `device` (or not bound yet), else a new one
Context * acquireContext(int device)
{
    ContextPool & pool = contextPool();
    {
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (size_t k = pool.idle.size(); k-- > 0;) {
            Context * c = pool.idle[k];
            if (c->device == device || c->device < 0 || device < 0) {
                pool.idle.erase(pool.idle.begin() + (long)k);
                return c;
            }
        }
    }
    return new Context;
}
} // namespace

Context & currentContext()
{
    if (!lease.context) {
        int device = -1;
        if (hipGetDevice(&device) != hipSuccess) {
            (void)hipGetLastError();
            device = -1;
        }
        lease.context = acquireContext(device);
        lease.context->lastError[0] = 0;
        lease.context->lastKernel = "";
    }
    return *lease.context;
}

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


namespace {
// destroys whatever a failed first-time ensureContext() managed to create, so that the next call starts from scratch
void discardPartialContext()
{
    auto dropStream = [](hipStream_t & s) {
        if (s)
            (void)hipStreamDestroy(s);
        s = nullptr;
    };
    auto dropEvent = [](hipEvent_t & e) {
        if (e)
            (void)hipEventDestroy(e);
        e = nullptr;
    };
    dropStream(tls.stream), dropStream(tls.upStream), dropStream(tls.downStream);
    for (int b = 0; b < Context::kMaxBands; ++b)
        dropEvent(tls.bandUp[b]), dropEvent(tls.bandDone[b]);
    for (int k = 0; k < Context::kTableRing; ++k)
        dropEvent(tls.tableCopied[k]), dropEvent(tls.tableConsumed[k]), dropEvent(tls.uploadCopied[k]);
    dropEvent(tls.scratchUsed);
    (void)hipGetLastError();
    tls.ready = false;
}

avifResult buildContext()
{
    HIP_TRY(hipStreamCreateWithFlags(&tls.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&tls.upStream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&tls.downStream, hipStreamNonBlocking));
    for (int b = 0; b < Context::kMaxBands; ++b) {
        HIP_TRY(hipEventCreateWithFlags(&tls.bandUp[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&tls.bandDone[b], hipEventDisableTiming));
    }
    for (int k = 0; k < Context::kTableRing; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&tls.tableCopied[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&tls.tableConsumed[k], hipEventDisableTiming));
    }
    for (int k = 0; k < Context::kTableRing; ++k)
        HIP_TRY(hipEventCreateWithFlags(&tls.uploadCopied[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&tls.scratchUsed, hipEventDisableTiming));
    return AVIF_RESULT_OK;
}
} // namespace

avifResult ensureContext()
{
    if (tls.ready && tls.ownerPid != (int)getpid()) {
        // a fork()ed child (Python multiprocessing's default start method) inherits the parent's context: its streams belong to a runtime
        // the child does not have, and its download helper thread does not exist here -- waiting for it would block forever.  Nothing of it
        // can be released from this side; the child starts over with a context of its own.
        lease.context = new Context;
    }
    if (tls.ready) {
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
    // all or nothing: contexts are pooled and recycled to other threads for the life of the process, so a context whose creation failed
    // half way (an event that could not be created) must not look initialised -- what exists is destroyed and the next call retries
    const avifResult built = buildContext();
    if (built != AVIF_RESULT_OK) {
        char keep[sizeof(tls.lastError)];
        memcpy(keep, tls.lastError, sizeof(keep));
        discardPartialContext();
        memcpy(tls.lastError, keep, sizeof(keep));
        return built;
    }
    tls.ownerPid = (int)getpid();
    tls.ready = true;
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
    constexpr int kRing = Context::kTableRing;
    if (bytes > tls.pinnedUploadCapacity) {
        if (tls.pinnedUpload) {
            for (int k = 0; k < kRing; ++k)
                HIP_TRY(hipEventSynchronize(tls.uploadCopied[k]));
            HIP_TRY(hipHostFree(tls.pinnedUpload));
        }
        tls.pinnedUpload = nullptr, tls.pinnedUploadCapacity = 0;
        const size_t rounded = (bytes + 16383) & ~(size_t)16383;
        HIP_TRY(hipHostMalloc(&tls.pinnedUpload, rounded * kRing, hipHostMallocDefault));
        tls.pinnedUploadCapacity = rounded;
    }
    const uint32_t slot = tls.uploadSlot++ % (uint32_t)kRing;
    HIP_TRY(hipEventSynchronize(tls.uploadCopied[slot])); // the upload of kRing calls ago has left this slot
    uint8_t * pinned = (uint8_t *)tls.pinnedUpload + (size_t)slot * tls.pinnedUploadCapacity;
    memcpy(pinned, hostSrc, bytes);
    HIP_TRY(hipMemcpyAsync(deviceDst, pinned, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(tls.uploadCopied[slot], stream));
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

// Row bands of a host-resident conversion: band b's upload (upStream), its kernel (stream) and its download (downStream) are
// chained by events, so the download of one band, the kernel of the next and the upload of the one after run at the same time
// -- PCIe is full duplex (tests/tools/pcie_probe.hip: 56 GB/s each way alone, 53 + 20 GB/s together).  Bands start on multiples
// of 32 rows (whole tiles of the tiled kernels), at least ~2 megapixels each, at most Context::kMaxBands.
namespace {
// leaves no download running into the caller's memory when a banded call returns early
struct DrainOnExit
{
    CopyWorker * worker;
    ~DrainOnExit()
    {
        if (worker)
            (void)worker->drain();
    }
};
} // namespace

static uint32_t bandRowsFor(uint32_t width, uint32_t height)
{
    const uint64_t pixels = (uint64_t)width * height;
    uint32_t bands = (uint32_t)(pixels >> 21);
    bands = bands < 1 ? 1 : (bands > 8 ? 8 : bands);
    uint32_t rows = (height + bands - 1) / bands;
    rows = (rows + 31u) & ~31u;
    return rows < 64 ? 64 : rows;
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
    // device twins of the host-resident buffers (reserved here, filled band by band below)
    bool planeOnHost[4];
    for (int p = 0; p < 4; ++p) {
        const uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
        planeOnHost[p] = host && hostRowBytes && !isDevicePointer(host);
    }
    avifResult r = stagePlanes(&imageView, /*upload=*/false, /*mirrorRowBytes=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    const bool pixelsOnHost = !isDevicePointer(rgb->pixels);
    // destination bytes the kernel does not define (alpha kept as is) must survive the round trip
    const bool keepsBytes = probe.rgb.hasAlpha && probe.alphaSource == ALPHA_KEEP;
    if (pixelsOnHost) {
        r = stagePixels(&rgbView, /*upload=*/false);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    const PlaneGeometry g = planeGeometry(image);
    const bool subY = image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420;
    const uint32_t pixelRowBytes = rgb->width * rgbPixelBytes(rgb);
    const uint32_t bandRows = bandRowsFor(image->width, image->height);
    const bool banded = pixelsOnHost && bandRows < image->height;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    uint32_t chromaUploaded = 0; // chroma rows [0, chromaUploaded) are on the device (or on their way, on upStream)
    int band = 0;
    for (uint32_t y0 = 0; y0 < image->height; y0 += bandRows, ++band) {
        const uint32_t y1 = (y0 + bandRows < image->height) ? y0 + bandRows : image->height;
        const int e = band % Context::kMaxBands;
        // ---- up: luma / alpha rows [y0, y1); chroma rows up to the one below the band's last (the 4:2:0 filter's lower
        //      neighbour; the upper one arrived with the previous band): every row crosses the bus exactly once ----
        bool uploaded = false;
        for (int p = 0; p < 4; ++p) {
            if (!planeOnHost[p])
                continue;
            const uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
            const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
            uint8_t * dev = (p < 3) ? imageView.yuvPlanes[p] : imageView.alphaPlane;
            const uint32_t devRowBytes = (p < 3) ? imageView.yuvRowBytes[p] : imageView.alphaRowBytes;
            uint32_t r0 = y0, r1 = y1;
            if (p == 1 || p == 2) {
                r0 = chromaUploaded;
                r1 = subY ? ((y1 - 1) >> 1) + 2 : y1;
                r1 = (r1 > g.rows[p] || y1 == image->height) ? g.rows[p] : r1;
            }
            if (r1 > r0) {
                HIP_TRY(hipMemcpy2DAsync(dev + (size_t)r0 * devRowBytes, devRowBytes, host + (size_t)r0 * hostRowBytes, hostRowBytes, g.widthBytes[p], r1 - r0,
                                         hipMemcpyHostToDevice, tls.upStream));
                uploaded = true;
            }
            if (p == 2 || (p == 1 && !planeOnHost[2]))
                chromaUploaded = r1 > chromaUploaded ? r1 : chromaUploaded;
        }
        if (pixelsOnHost && keepsBytes) {
            HIP_TRY(hipMemcpy2DAsync(rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes,
                                     y1 - y0, hipMemcpyHostToDevice, tls.upStream));
            uploaded = true;
        }
        if (uploaded) {
            HIP_TRY(hipEventRecord(tls.bandUp[e], tls.upStream));
            HIP_TRY(hipStreamWaitEvent(tls.stream, tls.bandUp[e], 0));
        }
        // ---- convert the band: a rectangle of the canvas (edge rules against the whole image) ----
        avifCropRect rect;
        rect.x = 0, rect.y = y0, rect.width = image->width, rect.height = y1 - y0;
        YuvToRgbPlan plan;
        r = makeYuvToRgbPlan(&imageView, &rgbView, (y0 == 0 && y1 == image->height) ? nullptr : &rect, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan,
                             colorOnly, reformatAlpha);
        if (r == AVIF_RESULT_OK)
            r = enqueueYuvToRgb(plan, tls.stream);
        if (r != AVIF_RESULT_OK) {
            (void)hipStreamSynchronize(tls.upStream);
            (void)hipStreamSynchronize(tls.stream);
            return r; // (drainOnExit waits for the downloads already posted)
        }
        // ---- down: by the helper thread (a pageable download blocks its caller), or right here when there is one band only ----
        if (pixelsOnHost) {
            HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
            const CopyWorker::Job job = { tls.bandDone[e], rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes,
                                          pixelRowBytes, y1 - y0 };
            if (banded) {
                tls.downloader->post(job);
            } else {
                HIP_TRY(hipStreamWaitEvent(tls.downStream, job.after, 0));
                HIP_TRY(hipMemcpy2DAsync(job.dst, job.dstPitch, job.src, job.srcPitch, job.widthBytes, job.rows, hipMemcpyDeviceToHost, tls.downStream));
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    if (pixelsOnHost) {
        if (banded) {
            const hipError_t de = tls.downloader->drain();
            if (de != hipSuccess)
                return hipFailed(de, "download of converted rows");
        } else {
            HIP_TRY(hipStreamSynchronize(tls.downStream));
        }
    }
    return AVIF_RESULT_OK;
}

// ---- rectangles of a host-resident canvas (the tile farm's per-rank primitive) ----
namespace {
// what one rectangle moves over the host link: windows of the planes (own samples plus the 1-sample chroma halo of the bilinear
// filter, clamped to the plane) and the pixel rectangle
struct RectWindows
{
    uint32_t x0[4], y0[4], w[4], h[4]; // per plane (Y, U, V, A), in samples; w == 0: nothing to move
};

RectWindows rectWindows(const avifImage * canvas, const YuvToRgbPlan & plan, const avifCropRect & r)
{
    RectWindows W;
    memset(&W, 0, sizeof(W));
    const int sx = (canvas->yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || canvas->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 0 : 1;
    const int sy = (canvas->yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    const uint32_t cw = (canvas->width + (uint32_t)sx) >> sx, ch = (canvas->height + (uint32_t)sy) >> sy;
    const uint32_t halo = (plan.bilinear && plan.yuv.hasColor) ? 1u : 0u; // src/reformat.c:760-800: neighbours of the 4-tap filter
    for (int p = 0; p < 4; ++p) {
        const uint8_t * plane = (p < 3) ? canvas->yuvPlanes[p] : canvas->alphaPlane;
        if (!plane || ((p == 1 || p == 2) && canvas->yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
            continue;
        if (p == 3 && !(plan.alphaSource == ALPHA_PLANE || plan.inLoopMul != MUL_NONE || plan.postMul != MUL_NONE))
            continue; // the conversion does not read the alpha plane
        if (p == 0 || p == 3) {
            W.x0[p] = r.x, W.y0[p] = r.y, W.w[p] = r.width, W.h[p] = r.height;
            continue;
        }
        const uint32_t hx = sx ? halo : 0, hy = sy ? halo : 0;
        const uint32_t cx0 = r.x >> sx, cx1 = (r.x + r.width - 1) >> sx, cy0 = r.y >> sy, cy1 = (r.y + r.height - 1) >> sy;
        const uint32_t ax0 = cx0 >= hx ? cx0 - hx : 0, ay0 = cy0 >= hy ? cy0 - hy : 0;
        const uint32_t ax1 = (cx1 + hx < cw) ? cx1 + hx : cw - 1, ay1 = (cy1 + hy < ch) ? cy1 + hy : ch - 1;
        W.x0[p] = ax0, W.y0[p] = ay0, W.w[p] = ax1 - ax0 + 1, W.h[p] = ay1 - ay0 + 1;
    }
    return W;
}
} // namespace

// Horizontally adjacent rectangles of one tile row are converted as one wider rectangle (the same bytes: edge rules are the
// canvas's): copies between pageable memory and the device move long rows far faster than short ones (64 tiles of 1920 x 1080
// one by one: 46 ms per 15360 x 8640 canvas; as 8 full-width bands: 21 ms -- tests/tools/e2e_bench.py)
static std::vector<avifCropRect> coalesceRects(const avifCropRect * rects, uint32_t count)
{
    std::vector<avifCropRect> jobs(rects, rects + count);
    if (jobs.empty())
        return jobs;
    std::sort(jobs.begin(), jobs.end(), [](const avifCropRect & a, const avifCropRect & b) { return a.y != b.y ? a.y < b.y : a.x < b.x; });
    size_t n = 0;
    for (size_t k = 1; k < jobs.size(); ++k) {
        avifCropRect & cur = jobs[n];
        if (jobs[k].y == cur.y && jobs[k].height == cur.height && jobs[k].x == cur.x + cur.width)
            cur.width += jobs[k].width;
        else
            jobs[++n] = jobs[k];
    }
    jobs.resize(n + 1);
    return jobs;
}

extern "C" avifResult avifhipPlanRectTransfers(const avifImage * canvas, const avifRGBImage * rgbCanvas, const avifCropRect * rects, uint32_t count, uint64_t * bytesUp,
                                               uint64_t * bytesDown)
{
    if (!canvas || !rgbCanvas || (count && !rects))
        return AVIF_RESULT_INVALID_ARGUMENT;
    uint64_t up = 0, down = 0;
    const uint32_t bps = (canvas->depth > 8) ? 2 : 1;
    for (uint32_t k = 0; k < count; ++k) {
        YuvToRgbPlan probe;
        const avifResult pr = makeYuvToRgbPlan(canvas, rgbCanvas, &rects[k], effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &probe);
        if (pr != AVIF_RESULT_OK)
            return pr;
    }
    for (const avifCropRect & rc : coalesceRects(rects, count)) {
        YuvToRgbPlan plan;
        const avifResult pr = makeYuvToRgbPlan(canvas, rgbCanvas, &rc, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan);
        if (pr != AVIF_RESULT_OK)
            return pr;
        const RectWindows W = rectWindows(canvas, plan, rc);
        for (int p = 0; p < 4; ++p)
            up += (uint64_t)W.w[p] * W.h[p] * bps;
        const uint64_t px = (uint64_t)rc.width * rc.height * rgbPixelBytes(rgbCanvas);
        if (plan.rgb.hasAlpha && plan.alphaSource == ALPHA_KEEP)
            up += px; // destination bytes the kernel leaves alone must make the round trip
        down += px;
    }
    if (bytesUp)
        *bytesUp = up;
    if (bytesDown)
        *bytesDown = down;
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipImageYUVToRGBRects(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * rects, uint32_t count)
{
    if (!canvas || !rgbCanvas || (count && !rects))
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (count == 0)
        return AVIF_RESULT_OK;
    // every rectangle is validated before the device is touched (error codes of the whole-image call, plus the rectangle rules)
    for (uint32_t k = 0; k < count; ++k) {
        YuvToRgbPlan probe;
        const avifResult pr = makeYuvToRgbPlan(canvas, rgbCanvas, &rects[k], effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &probe);
        if (pr != AVIF_RESULT_OK)
            return pr;
    }
    if (!rgbCanvas->pixels) {
        setError("avifhipImageYUVToRGBRects: rgb->pixels is NULL");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    for (int p = 0; p < 4; ++p) {
        const uint8_t * plane = (p < 3) ? canvas->yuvPlanes[p] : canvas->alphaPlane;
        if (plane && isDevicePointer(plane)) {
            setError("avifhipImageYUVToRGBRects: host-resident canvases only (device-resident ones: avifhipImageYUVToRGBBatchAsync)");
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
    }
    if (isDevicePointer(rgbCanvas->pixels)) {
        setError("avifhipImageYUVToRGBRects: host-resident canvases only (device-resident ones: avifhipImageYUVToRGBBatchAsync)");
        return AVIF_RESULT_INVALID_ARGUMENT;
    }
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    // canvas-sized device twins (reserved, not filled: only the rectangles' windows are uploaded)
    avifImage view;
    memcpy(&view, canvas, sizeof(avifImage));
    avifRGBImage rgbView = *rgbCanvas;
    avifResult r = stagePlanes(&view, /*upload=*/false, /*mirrorRowBytes=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    r = stagePixels(&rgbView, /*upload=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    if (!tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    DrainOnExit drainOnExit = { tls.downloader };
    const uint32_t bps = (canvas->depth > 8) ? 2 : 1, px = rgbPixelBytes(rgbCanvas);
    tls.bytesUp = tls.bytesDown = 0;
    const std::vector<avifCropRect> jobs = coalesceRects(rects, count);
    count = (uint32_t)jobs.size();
    for (uint32_t k = 0; k < count; ++k) {
        const avifCropRect & rc = jobs[k];
        const int e = (int)(k % Context::kMaxBands);
        YuvToRgbPlan plan;
        r = makeYuvToRgbPlan(&view, &rgbView, &rc, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan);
        if (r != AVIF_RESULT_OK)
            return r;
        const RectWindows W = rectWindows(canvas, plan, rc);
        for (int p = 0; p < 4; ++p) {
            if (!W.w[p])
                continue;
            const uint8_t * host = (p < 3) ? canvas->yuvPlanes[p] : canvas->alphaPlane;
            const uint32_t hostPitch = (p < 3) ? canvas->yuvRowBytes[p] : canvas->alphaRowBytes;
            uint8_t * dev = (p < 3) ? view.yuvPlanes[p] : view.alphaPlane;
            const uint32_t devPitch = (p < 3) ? view.yuvRowBytes[p] : view.alphaRowBytes;
            HIP_TRY(hipMemcpy2DAsync(dev + (size_t)W.y0[p] * devPitch + (size_t)W.x0[p] * bps, devPitch, host + (size_t)W.y0[p] * hostPitch + (size_t)W.x0[p] * bps, hostPitch,
                                     (size_t)W.w[p] * bps, W.h[p], hipMemcpyHostToDevice, tls.upStream));
            tls.bytesUp += (uint64_t)W.w[p] * W.h[p] * bps;
        }
        uint8_t * hostPx = rgbCanvas->pixels + (size_t)rc.y * rgbCanvas->rowBytes + (size_t)rc.x * px;
        uint8_t * devPx = rgbView.pixels + (size_t)rc.y * rgbView.rowBytes + (size_t)rc.x * px;
        if (plan.rgb.hasAlpha && plan.alphaSource == ALPHA_KEEP) {
            HIP_TRY(hipMemcpy2DAsync(devPx, rgbView.rowBytes, hostPx, rgbCanvas->rowBytes, (size_t)rc.width * px, rc.height, hipMemcpyHostToDevice, tls.upStream));
            tls.bytesUp += (uint64_t)rc.width * rc.height * px;
        }
        HIP_TRY(hipEventRecord(tls.bandUp[e], tls.upStream));
        HIP_TRY(hipStreamWaitEvent(tls.stream, tls.bandUp[e], 0));
        r = enqueueYuvToRgb(plan, tls.stream);
        if (r != AVIF_RESULT_OK) {
            (void)hipStreamSynchronize(tls.upStream);
            (void)hipStreamSynchronize(tls.stream);
            return r;
        }
        HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
        tls.downloader->post({ tls.bandDone[e], hostPx, rgbCanvas->rowBytes, devPx, rgbView.rowBytes, (size_t)rc.width * px, rc.height });
        tls.bytesDown += (uint64_t)rc.width * rc.height * px;
        // (the 16 events are reused round-robin: a download whose wait is enqueued after its event was recorded again simply waits
        //  for a LATER kernel of the same in-order stream -- still after its own)
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    const hipError_t de = tls.downloader->drain();
    if (de != hipSuccess)
        return hipFailed(de, "download of converted rectangles");
    return AVIF_RESULT_OK;
}

extern "C" void avifhipLastTransferBytes(uint64_t * bytesUp, uint64_t * bytesDown)
{
    if (bytesUp)
        *bytesUp = tls.bytesUp;
    if (bytesDown)
        *bytesDown = tls.bytesDown;
}

extern "C" avifResult avifhipImageYUVToRGB(const avifImage * image, avifRGBImage * rgb)
{
    return yuvToRgbSync(image, rgb, false, false);
}

extern "C" avifResult avifhipImageYUVToRGBColorOnly(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha)
{
    return yuvToRgbSync(image, rgb, true, reformatAlpha != AVIF_FALSE);
}

// The colour hook with what libavif does NEXT folded in.  After AVIF_RESULT_OK from avifImageYUVToRGBLibYUV, avifImageYUVToRGBImpl runs
// avifRGBImagePremultiplyAlpha / UnpremultiplyAlpha on the same pixels when an alpha (un)multiply is pending and avifRGBImageToF16 when
// rgb->isFloat (src/reformat.c:1574-1590) -- each of them another hook call that stages a host-resident image across the bus both ways
// (8K RGBA16: 265 MB each way per call).  The whole-call plan computes exactly that sequence in one pass (the integer post-pass after the
// conversion is what a libyuv-backed libavif runs too), so the hook can hand back the FINAL pixels and tell its caller which follow-up
// calls to answer with AVIF_RESULT_OK without touching the pixels again.
extern "C" avifResult avifhipImageYUVToRGBHook(const avifImage * image, avifRGBImage * rgb, avifBool reformatAlpha, uint32_t * folded)
{
    if (folded)
        *folded = 0;
    if (!image || !rgb)
        return AVIF_RESULT_INVALID_ARGUMENT;
    YuvToRgbPlan hook, whole;
    const avifResult hr = makeYuvToRgbPlan(image, rgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &hook, true, reformatAlpha != AVIF_FALSE);
    if (hr != AVIF_RESULT_OK)
        return hr; // (declines exactly what avifhipImageYUVToRGBColorOnly declines)
    const bool pending = hook.mulOfTheCall != MUL_NONE || rgb->isFloat;
    if (!folded || !pending)
        return yuvToRgbSync(image, rgb, true, reformatAlpha != AVIF_FALSE);
    // the whole call must be the hook's job plus post-passes: same arithmetic family, same alpha channel, the multiply as a post-pass
    const avifResult wr = makeYuvToRgbPlan(image, rgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &whole, false, false);
    const bool sameJob = wr == AVIF_RESULT_OK && whole.arith == hook.arith && whole.alphaSource == hook.alphaSource && whole.inLoopMul == MUL_NONE &&
                         whole.postMul == hook.mulOfTheCall && whole.bilinear == hook.bilinear && whole.identityCopy == hook.identityCopy;
    if (!sameJob)
        return yuvToRgbSync(image, rgb, true, reformatAlpha != AVIF_FALSE);
    const avifResult r = yuvToRgbSync(image, rgb, false, false);
    if (r == AVIF_RESULT_OK)
        *folded = (hook.mulOfTheCall == MUL_MULTIPLY ? AVIFHIP_FOLDED_PREMULTIPLY : hook.mulOfTheCall == MUL_UNMULTIPLY ? AVIFHIP_FOLDED_UNPREMULTIPLY : 0u) |
                  (rgb->isFloat ? AVIFHIP_FOLDED_TO_F16 : 0u);
    return r;
}

namespace {
// Per-job overrides of a batch: the chroma window (cwinX0, cwinX1, cwinY0, cwinY1) and the limited-range alpha flag
struct JobOverride
{
    int32_t window[4];
    bool alphaLimited;
};
} // namespace

// `extra` (optional): a host table of the caller that rides in the same upload (the grid's tile table for the seam kernel); its device
// address comes back in *extraDevice, the ring slot in *slotOut -- the caller records tls.tableConsumed[slot] again after ITS kernels.
static avifResult batchAsyncImpl(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, const avifCropRect * rects,
                                 const JobOverride * overrides, void * hipStream, const PixelMap * map = nullptr, const void * extra = nullptr, size_t extraBytes = 0,
                                 const void ** extraDevice = nullptr, uint32_t * slotOut = nullptr)
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
    const size_t extraOffset = (tileBytes + 2 * planBytes + 255) & ~(size_t)255;
    const size_t bytes = extra ? extraOffset + extraBytes : tileBytes + 2 * planBytes;
    constexpr int kRing = Context::kTableRing;
    if (bytes > tls.pinnedTableCapacity) {
        if (tls.pinnedTable) {
            for (int k = 0; k < kRing; ++k)
                HIP_TRY(hipEventSynchronize(tls.tableCopied[k]));
            HIP_TRY(hipHostFree(tls.pinnedTable));
            tls.pinnedTable = nullptr;
            tls.pinnedTableCapacity = 0;
            HIP_TRY(hipDeviceSynchronize()); // the slots' device slices move as well: no batch may still be reading the old ones
        }
        const size_t slotBytes = (bytes + 4095) & ~(size_t)4095;
        HIP_TRY(hipHostMalloc(&tls.pinnedTable, slotBytes * kRing, hipHostMallocDefault));
        tls.pinnedTableCapacity = slotBytes;
    }
    const uint32_t slot = tls.tableSlot++ % (uint32_t)kRing;
    HIP_TRY(hipEventSynchronize(tls.tableCopied[slot])); // the upload of kRing batches ago has left this slot's pinned memory
    uint8_t * pinned = (uint8_t *)tls.pinnedTable + (size_t)slot * tls.pinnedTableCapacity;
    if (extra)
        memcpy(pinned + extraOffset, extra, extraBytes);
    if (slotOut)
        *slotOut = slot;
    YuvToRgbPlan * plansA = (YuvToRgbPlan *)(pinned + tileBytes);
    YuvToRgbPlan * plansB = plansA + count;
    uint32_t maxW = 0, maxH = 0;
    bool allTiled = gTiledKernels.load(std::memory_order_relaxed) != 0;
    int variant = -2;
    const int arithmetic = effectiveArithmetic();
    const uint32_t tuning = gTuning.load(std::memory_order_relaxed);
    YuvToRgbPlan firstPlan;
    for (uint32_t k = 0; k < count; ++k) {
        if (!images[k] || !rgbs[k])
            return AVIF_RESULT_INVALID_ARGUMENT;
        // tiles of a grid / frames of a sequence share everything a plan is derived from: derive once, re-bind the buffers
        avifResult pr = AVIF_RESULT_OK;
        if (k == 0 || !rebindYuvToRgbPlan(firstPlan, images[0], rgbs[0], images[k], rgbs[k], rects ? &rects[k] : nullptr, &plansA[k], &pr))
            pr = makeYuvToRgbPlan(images[k], rgbs[k], rects ? &rects[k] : nullptr, arithmetic, tuning, &plansA[k]);
        if (pr != AVIF_RESULT_OK)
            return pr;
        if (k == 0)
            firstPlan = plansA[0]; // (before the per-job overrides below)
        if (map)
            plansA[k].rgb.map = *map; // fused crop / rotate / mirror: every job stores through the canvas's map
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
    const avifResult rr = reserve(tls.table, tls.pinnedTableCapacity * kRing); // (growing it waits for the device: nothing reads the old one then)
    if (rr != AVIF_RESULT_OK)
        return rr;
    hipStream_t stream = pickStream(hipStream);
    uint8_t * dev = (uint8_t *)tls.table.ptr + (size_t)slot * tls.pinnedTableCapacity;
    if (extraDevice)
        *extraDevice = dev + extraOffset;
    // The table crosses the link on `upStream` while earlier batches compute on `stream`: the upload waits only for the kernels that read
    // this slot's device slice kRing batches ago (on whichever stream they ran), the batch's kernels wait for the upload.
    HIP_TRY(hipStreamWaitEvent(tls.upStream, tls.tableConsumed[slot], 0));
    struct MarkConsumed
    {
        hipEvent_t ev;
        hipStream_t s;
        ~MarkConsumed() { (void)hipEventRecord(ev, s); }
    } markConsumed = { tls.tableConsumed[slot], stream };
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
        HIP_TRY(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, tls.upStream));
        HIP_TRY(hipEventRecord(tls.tableCopied[slot], tls.upStream));
        HIP_TRY(hipStreamWaitEvent(stream, tls.tableCopied[slot], 0));
        e = launchYuvToRgbTileBatch(dev, representative, count, maxW, maxH, stream, &tls.lastKernel);
        if (e == hipSuccess && restW)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes), count, restW, restMaxH, stream);
        if (e == hipSuccess && restH)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes) + count, count, restMaxW, restH, stream);
    } else {
        HIP_TRY(hipMemcpyAsync(dev + tileBytes, plansA, extra ? bytes - tileBytes : planBytes, hipMemcpyHostToDevice, tls.upStream));
        HIP_TRY(hipEventRecord(tls.tableCopied[slot], tls.upStream));
        HIP_TRY(hipStreamWaitEvent(stream, tls.tableCopied[slot], 0));
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
static avifResult gridYuvToRgbImpl(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles, avifBool alphaIsLimitedRange,
                                   avifRGBImage * rgbCanvas, void * hipStream, const PixelMap * map)
{
    if (!grid || !colorTiles || !rgbCanvas || !grid->rows || !grid->columns || !grid->outputWidth || !grid->outputHeight)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const uint32_t count = grid->rows * grid->columns;
    const avifImage * first = colorTiles[0];
    if (!first || !first->width || !first->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (grid->outputWidth >= 65536u || grid->outputHeight >= 65536u) {
        // the seam kernel divides canvas coordinates by the tile size with a 32-bit multiply-high (kernels_generic.hip GridReader::divBy),
        // exact only below 65536; libavif's own default limits (16384^2 pixels, 32768 per side) are far inside
        setError("grid canvases of 65536 pixels or more per side are not supported (%u x %u)", grid->outputWidth, grid->outputHeight);
        return AVIF_RESULT_NOT_IMPLEMENTED;
    }
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
    // the seam kernel's tile table rides in the batch's descriptor upload (its own copy on the compute stream cost 5 us plus two gaps)
    const void * deviceTiles = nullptr;
    uint32_t tableSlot = 0;
    avifResult r = batchAsyncImpl(count, viewPtrs.data(), rgbPtrs.data(), rects.data(), overrides.data(), hipStream, map, tiles.data(), tiles.size() * sizeof(GridTile),
                                  &deviceTiles, &tableSlot);
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
    if (map)
        canvasPlan.rgb.map = *map;
    const bool filters = canvasPlan.bilinear && canvasPlan.yuv.hasColor && subsampled;
    if (!filters)
        return AVIF_RESULT_OK;
    hipStream_t stream = pickStream(hipStream);
    GridGeometry g;
    memset(&g, 0, sizeof(g));
    g.columns = grid->columns, g.rows = grid->rows, g.tileW = tw, g.tileH = th, g.tileCW = tw >> sx, g.tileCH = th >> sy;
    const hipError_t e = launchYuvToRgbGridSeams(canvasPlan, g, (const GridTile *)deviceTiles, grid->columns > 1, sy && grid->rows > 1, stream);
    (void)hipEventRecord(tls.tableConsumed[tableSlot], stream); // the slot is free again after the seam kernel, not after the batch
    if (e != hipSuccess)
        return hipFailed(e, "grid seam kernel launch");
    ++tls.launches;
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipGridYUVToRGBAsync(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                               avifBool alphaIsLimitedRange, avifRGBImage * rgbCanvas, void * hipStream)
{
    return gridYuvToRgbImpl(grid, colorTiles, alphaTiles, alphaIsLimitedRange, rgbCanvas, hipStream, nullptr);
}

// ---- the decode-side tail in one step: tiles -> canvas (src/read.c:1823-1877), limited -> full alpha
//      (:6724-6764), YUV -> RGB, and the application's avifApplyTransforms (apps/shared/avifutil.c:787-825) ----
namespace {
// validates crop / angle / axis like avifhipRGBImageTransformAsync and derives the destination size
avifResult transformGeometry(uint32_t canvasW, uint32_t canvasH, const avifCropRect * crop, avifBool rotate, uint8_t angle, avifBool mirror, uint8_t axis, avifCropRect * r,
                             int * quarterTurns, int * mirrorAxis, uint32_t * dw, uint32_t * dh)
{
    if ((rotate && angle > 3) || (mirror && axis > 1))
        return AVIF_RESULT_INVALID_ARGUMENT; // "Invalid angle." / "Invalid axis value.", apps/shared/avifutil.c:741,781
    const avifCropRect whole = { 0, 0, canvasW, canvasH };
    *r = crop ? *crop : whole;
    if (!r->width || !r->height || r->width > canvasW || r->height > canvasH || r->x > canvasW - r->width || r->y > canvasH - r->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    *quarterTurns = (rotate && angle != 0) ? angle : 0; // :805
    *mirrorAxis = mirror ? (int)axis : -1;
    *dw = (*quarterTurns & 1) ? r->height : r->width, *dh = (*quarterTurns & 1) ? r->width : r->height; // :692-693
    return AVIF_RESULT_OK;
}

// conversion parameters of `out` on a canvas-sized buffer
avifRGBImage canvasLike(const avifRGBImage * out, uint32_t w, uint32_t h, uint8_t * pixels, uint32_t rowBytes)
{
    avifRGBImage v = *out;
    v.width = w, v.height = h, v.pixels = pixels, v.rowBytes = rowBytes;
    return v;
}
} // namespace

extern "C" avifResult avifhipGridYUVToRGBTransformedAsync(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                                          avifBool alphaIsLimitedRange, avifRGBImage * rgb, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                                          avifBool mirror, uint8_t axis, void * hipStream)
{
    if (!grid || !colorTiles || !colorTiles[0] || !rgb || !rgb->pixels)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifCropRect r;
    int turns, mirrorAxis;
    uint32_t dw, dh;
    const avifResult gr = transformGeometry(grid->outputWidth, grid->outputHeight, crop, rotate, angle, mirror, axis, &r, &turns, &mirrorAxis, &dw, &dh);
    if (gr != AVIF_RESULT_OK)
        return gr;
    const uint32_t px = rgbPixelBytes(rgb);
    if (rgb->width != dw || rgb->height != dh || (uint64_t)rgb->rowBytes < (uint64_t)dw * px)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    // Fused when the conversion's tiled kernels can store through a map (today: the packed 16-bit integer kernels); otherwise
    // two passes: conversion into a canvas-sized scratch buffer, then the permutation pass of avifhipRGBImageTransformAsync
    const PixelMap map = makePixelMap(r.x, r.y, r.width, r.height, turns, mirrorAxis);
    avifImage probeImage;
    memcpy(&probeImage, colorTiles[0], sizeof(avifImage));
    if (alphaTiles && alphaTiles[0])
        probeImage.alphaPlane = alphaTiles[0]->alphaPlane, probeImage.alphaRowBytes = alphaTiles[0]->alphaRowBytes;
    avifRGBImage probeRgb = canvasLike(rgb, probeImage.width, probeImage.height, rgb->pixels, rgb->rowBytes);
    YuvToRgbPlan probe;
    const avifResult pr = makeYuvToRgbPlan(&probeImage, &probeRgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &probe);
    if (pr != AVIF_RESULT_OK)
        return pr;
    probe.rgb.map = map;
    probe.yuv.alphaLimited = (alphaTiles && alphaIsLimitedRange) ? 1 : 0;
    const bool fused = gTiledKernels.load(std::memory_order_relaxed) && tileYuvToRgbSupported(probe);
    if (fused) {
        avifRGBImage canvasRgb = canvasLike(rgb, grid->outputWidth, grid->outputHeight, rgb->pixels, rgb->rowBytes);
        return gridYuvToRgbImpl(grid, colorTiles, alphaTiles, alphaIsLimitedRange, &canvasRgb, hipStream, &map);
    }
    hipStream_t stream = pickStream(hipStream);
    ScratchScope scratch(stream);
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    const uint32_t pitch = alignUp(grid->outputWidth * px, 256);
    const avifResult rr = reserve(tls.xformCanvas, (size_t)pitch * grid->outputHeight);
    if (rr != AVIF_RESULT_OK)
        return rr;
    avifRGBImage canvasRgb = canvasLike(rgb, grid->outputWidth, grid->outputHeight, (uint8_t *)tls.xformCanvas.ptr, pitch);
    const avifResult g1 = gridYuvToRgbImpl(grid, colorTiles, alphaTiles, alphaIsLimitedRange, &canvasRgb, stream, nullptr);
    if (g1 != AVIF_RESULT_OK)
        return g1;
    return avifhipRGBImageTransformAsync(rgb, &canvasRgb, &r, rotate, angle, mirror, axis, stream);
}

extern "C" avifResult avifhipImageYUVToRGBTransformedAsync(const avifImage * image, avifRGBImage * rgb, const avifCropRect * crop, avifBool rotate, uint8_t angle,
                                                           avifBool mirror, uint8_t axis, void * hipStream)
{
    if (!image || !rgb || !rgb->pixels)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifCropRect r;
    int turns, mirrorAxis;
    uint32_t dw, dh;
    const avifResult gr = transformGeometry(image->width, image->height, crop, rotate, angle, mirror, axis, &r, &turns, &mirrorAxis, &dw, &dh);
    if (gr != AVIF_RESULT_OK)
        return gr;
    const uint32_t px = rgbPixelBytes(rgb);
    if (rgb->width != dw || rgb->height != dh || (uint64_t)rgb->rowBytes < (uint64_t)dw * px)
        return AVIF_RESULT_INVALID_ARGUMENT;
    avifRGBImage canvasRgb = canvasLike(rgb, image->width, image->height, rgb->pixels, rgb->rowBytes);
    YuvToRgbPlan plan;
    const avifResult pr = makeYuvToRgbPlan(image, &canvasRgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan);
    if (pr != AVIF_RESULT_OK)
        return pr;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    plan.rgb.map = makePixelMap(r.x, r.y, r.width, r.height, turns, mirrorAxis);
    if (gTiledKernels.load(std::memory_order_relaxed) && tileYuvToRgbSupported(plan))
        return enqueueYuvToRgb(plan, stream); // one launch (plus the universal kernel on the <= 3 columns / 1 row of leftovers)
    ScratchScope scratch(stream);
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    const uint32_t pitch = alignUp(image->width * px, 256);
    const avifResult rr = reserve(tls.xformCanvas, (size_t)pitch * image->height);
    if (rr != AVIF_RESULT_OK)
        return rr;
    canvasRgb.pixels = (uint8_t *)tls.xformCanvas.ptr, canvasRgb.rowBytes = pitch;
    const avifResult c1 = avifhipImageYUVToRGBAsync(image, &canvasRgb, stream);
    if (c1 != AVIF_RESULT_OK)
        return c1;
    return avifhipRGBImageTransformAsync(rgb, &canvasRgb, &r, rotate, angle, mirror, axis, stream);
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
        r = stagePixels(&rgbView, /*upload=*/false);
        if (r != AVIF_RESULT_OK)
            return r;
    }
    const PlaneGeometry g = planeGeometry(image);
    const bool subY = image->yuvFormat == AVIF_PIXEL_FORMAT_YUV420;
    const uint32_t pixelRowBytes = rgb->width * rgbPixelBytes(rgb);
    // Row bands, like yuvToRgbSync: a band of RGB rows is an independent sub-image of this direction (2 x 2 blocks never cross an
    // even row), so band b is converted as an image of its own rows while band b+1 uploads and band b-1 downloads.
    // Gray sources keep the single pass (their chroma planes are filled pitch-wide).
    const uint32_t bandRows = gray ? image->height : bandRowsFor(image->width, image->height);
    const bool banded = bandRows < image->height;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    int band = 0;
    for (uint32_t y0 = 0; y0 < image->height; y0 += bandRows, ++band) {
        const uint32_t y1 = (y0 + bandRows < image->height) ? y0 + bandRows : image->height;
        const int e = band % Context::kMaxBands;
        const uint32_t c0 = subY ? (y0 >> 1) : y0, c1 = subY ? ((y1 + 1) >> 1) : y1; // chroma rows of the band
        if (pixelsOnHost) {
            HIP_TRY(hipMemcpy2DAsync(rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes,
                                     y1 - y0, hipMemcpyHostToDevice, tls.upStream));
            HIP_TRY(hipEventRecord(tls.bandUp[e], tls.upStream));
            HIP_TRY(hipStreamWaitEvent(tls.stream, tls.bandUp[e], 0));
        }
        avifImage subImage;
        memcpy(&subImage, &imageView, sizeof(avifImage));
        avifRGBImage subRgb = rgbView;
        subImage.height = subRgb.height = y1 - y0;
        subRgb.pixels = rgbView.pixels + (size_t)y0 * rgbView.rowBytes;
        for (int p = 0; p < 3; ++p)
            if (subImage.yuvPlanes[p])
                subImage.yuvPlanes[p] += (size_t)((p == 0) ? y0 : c0) * subImage.yuvRowBytes[p];
        if (subImage.alphaPlane)
            subImage.alphaPlane += (size_t)y0 * subImage.alphaRowBytes;
        r = makeRgbToYuvPlan(&subImage, &subRgb, effectiveArithmetic(), &plan);
        if (r == AVIF_RESULT_OK) {
            finishRgbToYuvPlan(&subImage, &subRgb, &plan);
            r = enqueueRgbToYuv(plan, tls.stream);
        }
        if (r != AVIF_RESULT_OK) {
            (void)hipStreamSynchronize(tls.upStream);
            (void)hipStreamSynchronize(tls.stream);
            (void)hipStreamSynchronize(tls.downStream);
            return r;
        }
        HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
        if (!banded)
            HIP_TRY(hipStreamWaitEvent(tls.downStream, tls.bandDone[e], 0));
        for (int p = 0; p < 4; ++p) {
            uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
            const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
            const uint8_t * dev = (p < 3) ? imageView.yuvPlanes[p] : imageView.alphaPlane;
            const uint32_t devRowBytes = (p < 3) ? imageView.yuvRowBytes[p] : imageView.alphaRowBytes;
            if (!host || !hostRowBytes || host == dev)
                continue; // absent, or already device-resident
            const bool chroma = p == 1 || p == 2;
            const uint32_t r0 = chroma ? c0 : y0, r1 = chroma ? c1 : y1;
            if (gray && chroma) {
                HIP_TRY(hipMemcpyAsync(host, dev, (size_t)hostRowBytes * g.rows[p], hipMemcpyDeviceToHost, tls.downStream));
            } else if (chroma && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) {
                continue; // colour source into 4:0:0: chroma untouched
            } else if (banded) {
                tls.downloader->post({ tls.bandDone[e], host + (size_t)r0 * hostRowBytes, hostRowBytes, dev + (size_t)r0 * devRowBytes, devRowBytes, g.widthBytes[p], r1 - r0 });
            } else {
                HIP_TRY(hipMemcpy2DAsync(host + (size_t)r0 * hostRowBytes, hostRowBytes, dev + (size_t)r0 * devRowBytes, devRowBytes, g.widthBytes[p], r1 - r0, hipMemcpyDeviceToHost,
                                         tls.downStream));
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    if (banded) {
        const hipError_t de = tls.downloader->drain();
        if (de != hipSuccess)
            return hipFailed(de, "download of converted rows");
    } else {
        HIP_TRY(hipStreamSynchronize(tls.downStream));
    }
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

// In-place passes over host-resident pixels (premultiply / unpremultiply, half float): rows go up, through the kernel and back in bands,
// so that both directions of the link and the kernel overlap (the same three streams and helper thread as yuvToRgbSync).
// `launch(view, y0, rows, stream)` enqueues the pass on rows [y0, y0 + rows) of the device copy.
template <class Launch>
static avifResult inPlaceBanded(avifRGBImage * rgb, uint32_t pixelRowBytes, Launch launch)
{
    avifRGBImage view = *rgb;
    avifResult r = stagePixels(&view, /*upload=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    const uint32_t bandRows = bandRowsFor(rgb->width, rgb->height);
    const bool banded = bandRows < rgb->height;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    int band = 0;
    for (uint32_t y0 = 0; y0 < rgb->height; y0 += bandRows, ++band) {
        const uint32_t rows = (y0 + bandRows < rgb->height) ? bandRows : rgb->height - y0;
        const int e = band % Context::kMaxBands;
        HIP_TRY(hipMemcpy2DAsync(view.pixels + (size_t)y0 * view.rowBytes, view.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes, rows,
                                 hipMemcpyHostToDevice, tls.upStream));
        HIP_TRY(hipEventRecord(tls.bandUp[e], tls.upStream));
        HIP_TRY(hipStreamWaitEvent(tls.stream, tls.bandUp[e], 0));
        r = launch(view, y0, rows, tls.stream);
        if (r != AVIF_RESULT_OK) {
            (void)hipStreamSynchronize(tls.upStream);
            (void)hipStreamSynchronize(tls.stream);
            return r;
        }
        HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
        const CopyWorker::Job job = { tls.bandDone[e], rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, view.pixels + (size_t)y0 * view.rowBytes, view.rowBytes,
                                      pixelRowBytes, rows };
        if (banded) {
            tls.downloader->post(job);
        } else {
            HIP_TRY(hipStreamWaitEvent(tls.downStream, job.after, 0));
            HIP_TRY(hipMemcpy2DAsync(job.dst, job.dstPitch, job.src, job.srcPitch, job.widthBytes, job.rows, hipMemcpyDeviceToHost, tls.downStream));
        }
    }
    HIP_TRY(hipStreamSynchronize(tls.stream));
    if (banded) {
        const hipError_t de = tls.downloader->drain();
        if (de != hipSuccess)
            return hipFailed(de, "download of processed rows");
    } else {
        HIP_TRY(hipStreamSynchronize(tls.downStream));
    }
    return AVIF_RESULT_OK;
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
    if (isDevicePointer(rgb->pixels)) {
        r = enqueueAlphaMul(plan, tls.stream);
        if (r != AVIF_RESULT_OK)
            return r;
        HIP_TRY(hipStreamSynchronize(tls.stream));
        return AVIF_RESULT_OK;
    }
    return inPlaceBanded(rgb, rgb->width * rgbPixelBytes(rgb), [&](const avifRGBImage & view, uint32_t y0, uint32_t rows, hipStream_t stream) -> avifResult {
        avifRGBImage bandView = view;
        bandView.pixels = view.pixels + (size_t)y0 * view.rowBytes;
        bandView.height = rows;
        AlphaMulPlan bandPlan;
        const avifResult pr = makeAlphaMulPlan(&bandView, unmultiply, effectiveArithmetic(), &bandPlan);
        return pr != AVIF_RESULT_OK ? pr : enqueueAlphaMul(bandPlan, stream);
    });
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
    const uint32_t channels = (uint32_t)rgbFormatChannelCount((int)rgb->format);
    const float multiplier = 1.9259299444e-34f * (1.0f / 65535.0f); // src/reformat.c:1411,1429-1430
    tls.lastKernel = "to_f16_generic";
    if (isDevicePointer(rgb->pixels)) {
        const hipError_t e = launchToF16Generic(rgb->pixels, rgb->rowBytes, rgb->width * channels, rgb->height, multiplier, tls.stream);
        if (e != hipSuccess)
            return hipFailed(e, "half-float kernel launch");
        ++tls.launches;
        HIP_TRY(hipStreamSynchronize(tls.stream));
        return AVIF_RESULT_OK;
    }
    return inPlaceBanded(rgb, rgb->width * channels * 2, [&](const avifRGBImage & view, uint32_t y0, uint32_t rows, hipStream_t stream) -> avifResult {
        const hipError_t e = launchToF16Generic(view.pixels + (size_t)y0 * view.rowBytes, view.rowBytes, view.width * channels, rows, multiplier, stream);
        if (e != hipSuccess)
            return hipFailed(e, "half-float kernel launch");
        ++tls.launches;
        return AVIF_RESULT_OK;
    });
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

// =================================================================================================
// row packing for the file writers: Y4M frame payload, PNG rows
// =================================================================================================

extern "C" size_t avifhipY4MFrameBytes(const avifImage * image, avifBool withAlpha)
{
    if (!image)
        return 0;
    // the frame avifhipImagePackY4MFrameAsync would write: no frame (0) for what it refuses -- depths y4mWrite does not support, alpha
    // outside 8-bit 4:4:4 (apps/shared/y4m.c:487-489, :570-572)
    if (image->depth != 8 && image->depth != 10 && image->depth != 12)
        return 0;
    if (withAlpha && (!image->alphaPlane || !image->alphaRowBytes || image->depth != 8 || image->yuvFormat != AVIF_PIXEL_FORMAT_YUV444))
        return 0;
    const PlaneGeometry g = planeGeometry(image);
    size_t total = 0;
    for (int p = 0; p < 4; ++p) {
        if ((p == 3 && !withAlpha) || ((p == 1 || p == 2) && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
            continue;
        const uint8_t * plane = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        if (plane)
            total += (size_t)g.widthBytes[p] * g.rows[p];
    }
    return total;
}

// y4mWrite's payload loop, apps/shared/y4m.c:603-618: planes Y..V (..A), each row cut to its width
extern "C" avifResult avifhipImagePackY4MFrameAsync(const avifImage * image, avifBool withAlpha, uint8_t * frame, void * hipStream)
{
    if (!image || !frame || !image->yuvPlanes[0])
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (image->depth != 8 && image->depth != 10 && image->depth != 12)
        return AVIF_RESULT_NOT_IMPLEMENTED; // "y4mWrite unsupported depth", y4m.c:570-572
    if (withAlpha && (!image->alphaPlane || !image->alphaRowBytes || image->depth != 8 || image->yuvFormat != AVIF_PIXEL_FORMAT_YUV444))
        return AVIF_RESULT_NOT_IMPLEMENTED; // "writing alpha is currently only supported in 8bpc YUV444", y4m.c:487-489
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    const PlaneGeometry g = planeGeometry(image);
    size_t offset = 0;
    for (int p = 0; p < 4; ++p) {
        if ((p == 3 && !withAlpha) || ((p == 1 || p == 2) && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400))
            continue;
        const uint8_t * plane = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        if (!plane)
            continue;
        PackArgs A;
        A.src = plane, A.dst = frame + offset;
        A.srcPitch = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
        A.dstPitch = A.widthBytes = g.widthBytes[p];
        A.rows = g.rows[p];
        A.swap16 = 0; // Y4M stores 16-bit samples little-endian, as libavif does
        const hipError_t e = launchPackRows(A, stream);
        if (e != hipSuccess)
            return hipFailed(e, "row packing kernel launch");
        offset += (size_t)A.widthBytes * A.rows;
    }
    tls.lastKernel = "pack_rows";
    ++tls.launches;
    return AVIF_RESULT_OK;
}

// what avifPNGWrite hands to libpng, apps/shared/avifpng.c:865-880: the pixel rows, and png_set_swap for depths above 8
extern "C" avifResult avifhipRGBImagePackPNGRowsAsync(const avifRGBImage * rgb, uint8_t * rows, void * hipStream)
{
    if (!rgb || !rgb->pixels || !rows || !rgb->width || !rgb->height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    if (rgb->format == AVIF_RGB_FORMAT_RGB_565 || rgb->isFloat)
        return AVIF_RESULT_NOT_IMPLEMENTED; // the PNG writer asks for 8- or 16-bit integer RGB(A) / gray, avifpng.c:640-690
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    PackArgs A;
    A.src = rgb->pixels, A.dst = rows;
    A.srcPitch = rgb->rowBytes;
    A.dstPitch = A.widthBytes = rgb->width * rgbPixelBytes(rgb);
    A.rows = rgb->height;
    A.swap16 = rgb->depth > 8;
    const hipError_t e = launchPackRows(A, pickStream(hipStream));
    if (e != hipSuccess)
        return hipFailed(e, "row packing kernel launch");
    tls.lastKernel = "pack_rows";
    ++tls.launches;
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipSetDevice(int device)
{
    if (tls.ready && tls.device != device) {
        // the thread's context lives on another device: hand it back and lease one of `device`
        HIP_TRY(hipStreamSynchronize(tls.stream));
        {
            ContextPool & pool = contextPool();
            std::lock_guard<std::mutex> lock(pool.mutex);
            pool.idle.push_back(lease.context);
        }
        lease.context = acquireContext(device);
        lease.context->lastError[0] = 0;
        lease.context->lastKernel = "";
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
    // hipMemset on device memory returns before the fill has run (it is enqueued on the null stream), and the library's own
    // streams are non-blocking, i.e. NOT ordered behind the null stream: complete the fill here, as the name of a plain helper
    // promises (tests/test_pack.py caught a kernel's output being overwritten by a late fill)
    HIP_TRY(hipMemset(devicePtr, value, bytes));
    HIP_TRY(hipStreamSynchronize(nullptr));
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
