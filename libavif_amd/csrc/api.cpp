// api.cpp -- behind the C ABI of libavifhip.so (include/avifhip.h), the part every entry point shares: the per-thread context (streams,
// events, scratch, descriptor rings), its pool and what a fork() does to it, error text, pointer classification (host vs HBM), staging of
// host-resident images through device scratch, kernel selection for a plan; and the library's control / diagnostics entry points.
// The conversions themselves: api_decode.cpp (YUV -> RGB, one image), api_batch.cpp (batches, grids, the fused tail), api_encode.cpp
// (RGB -> YUV, alpha), api_apps.cpp (Sample Transform, crop / rotate / mirror, row packing), api_gainmap.cpp, api_scale.cpp.
#include "api_internal.h"

#include <map>

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
        // the thread's last use of the device scratch is marked now: the stream it ran on may be the thread's own, gone before the
        // context's next holder asks (ScratchScope)
        if (context->scratchPending && !context->scratchMarked && context->scratchUsed) {
            if (context->scratchGeneration && ownedStreamGeneration(context->scratchStream) == context->scratchGeneration &&
                hipEventRecord(context->scratchUsed, context->scratchStream) == hipSuccess) {
                context->scratchMarked = true;
            } else {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
                context->scratchPending = false;
            }
        }
        for (int k = 0; k < Context::kTableRing; ++k) { // likewise the batch tables' slots (api_batch.cpp batchAsyncImpl)
            if (!context->tableUnmarked[k] || !context->tableConsumed[k])
                continue;
            if (ownedStreamGeneration(context->tableLastStream[k]) != context->tableLastGeneration[k] ||
                hipEventRecord(context->tableConsumed[k], context->tableLastStream[k]) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
            context->tableUnmarked[k] = false;
        }
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
    // (a caller's own stream -- generation 0 -- was marked when it was noted, and is waited for even when the handle LOOKS like the one asking now:
    //  a stream destroyed behind the library's back may have handed its address to a new one, which is ordered behind nothing; when it is the
    //  same stream after all, waiting for its own event costs next to nothing)
    if (tls.scratchPending && (tls.scratchStream != s || (tls.scratchGeneration == 0 && tls.scratchMarked))) {
        // The previous user's stream is marked only now that somebody on another stream needs to wait for it: an event recorded here
        // covers everything that stream was given before, and calls that stay on one stream (nearly all) pay for no event at all -- a
        // record behind every launch kept the next kernel waiting for the signal (plane scaling: 24 us per call around an 18 us kernel).
        // (only a stream the library owns, still of the generation noted, is trusted with the deferred record)
        hipError_t e = tls.scratchMarked ? hipSuccess
                       : (tls.scratchGeneration && ownedStreamGeneration(tls.scratchStream) == tls.scratchGeneration)
                           ? hipEventRecord(tls.scratchUsed, tls.scratchStream)
                           : hipErrorInvalidHandle;
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s, tls.scratchUsed, 0);
        if (e != hipSuccess) {
            // (that stream is gone: everything the device was given finishes first, then)
            (void)hipGetLastError();
            e = hipDeviceSynchronize();
        }
        if (e != hipSuccess)
            result = hipFailed(e, "ordering the device scratch between streams");
    }
}

ScratchScope::~ScratchScope()
{
    tls.scratchStream = stream;
    tls.scratchGeneration = ownedStreamGeneration(stream);
    tls.scratchPending = true;
    tls.scratchMarked = false;
    // a stream of the caller's own may be destroyed (and its address reused) before anybody asks: its use is marked while the handle is
    // known to be good; the library's own streams wait until a call on another stream needs the mark (ScratchScope::ScratchScope)
    if (!tls.scratchGeneration && tls.scratchUsed && hipEventRecord(tls.scratchUsed, stream) == hipSuccess)
        tls.scratchMarked = true;
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
            AVIFHIP_HOST_MEMORY_FREED(tls.pinnedUpload);
            HIP_TRY(hipHostFree(tls.pinnedUpload));
        }
        tls.pinnedUpload = nullptr, tls.pinnedUploadCapacity = 0;
        const size_t rounded = (bytes + 16383) & ~(size_t)16383;
        HIP_TRY(hipHostMalloc(&tls.pinnedUpload, rounded * kRing, hipHostMallocDefault));
        AVIFHIP_NEW_HOST_MEMORY(tls.pinnedUpload, rounded * kRing);
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
    // AVIFHIP_POISON_SCRATCH=1 (the GPU test tier sets it, tests/conftest.py): new scratch starts as 0xA5 bytes instead of whatever the
    // allocator hands out -- zeros in a young process, another context's pixels later.  A kernel that reads scratch nobody wrote (a table
    // past its padding, a staging row past the upload) then misbehaves in EVERY process, not in two of three after the right history.
    static const bool poison = [] {
        const char * e = getenv("AVIFHIP_POISON_SCRATCH");
        return e && *e && strcmp(e, "0") != 0;
    }();
    if (poison) {
        HIP_TRY(hipMemsetAsync(s.ptr, 0xA5, rounded, tls.stream));
        HIP_TRY(hipStreamSynchronize(tls.stream));
    }
    return AVIF_RESULT_OK;
}

bool hostRowsWantOneBlock(const void * host, size_t hostPitch, size_t widthBytes, size_t rows)
{
    if ((((uintptr_t)host | hostPitch | widthBytes) & 3u) == 0)
        return false; // the 2-D copy runs at link speed
    return rows >= 8 && hostPitch >= widthBytes && hostPitch <= 2 * widthBytes + 256;
}

avifResult uploadRows(Scratch & raw, uint8_t * dev, size_t devPitch, const uint8_t * host, size_t hostPitch, size_t widthBytes, size_t rows, hipStream_t stream)
{
    if (!rows || !widthBytes)
        return AVIF_RESULT_OK;
    if ((devPitch & 3u) || ((uintptr_t)dev & 3u) || !hostRowsWantOneBlock(host, hostPitch, widthBytes, rows) || (hostPitch >> 32) || (devPitch >> 32) || (widthBytes >> 32) || (rows >> 32)) {
        HIP_TRY(hipMemcpy2DAsync(dev, devPitch, host, hostPitch, widthBytes, rows, hipMemcpyHostToDevice, stream));
        return AVIF_RESULT_OK;
    }
    const size_t bytes = (rows - 1) * hostPitch + widthBytes;
    if (bytes > raw.capacity) {
        // (grows with the first band of a call at most: the kernel that read the previous band's block runs on this very stream)
        HIP_TRY(hipStreamSynchronize(stream));
        if (raw.ptr)
            HIP_TRY(hipFree(raw.ptr));
        raw.ptr = nullptr, raw.capacity = 0;
        const size_t rounded = ((bytes + bytes / 8) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        HIP_TRY(hipMalloc(&raw.ptr, rounded));
        raw.capacity = rounded;
    }
    HIP_TRY(hipMemcpyAsync(raw.ptr, host, bytes, hipMemcpyHostToDevice, stream));
    const PackArgs a = { (const uint8_t *)raw.ptr, dev, (uint32_t)hostPitch, (uint32_t)devPitch, (uint32_t)widthBytes, (uint32_t)rows, 0 };
    const hipError_t e = launchUnpackRows(a, stream);
    if (e != hipSuccess)
        return hipFailed(e, "re-pitch of uploaded rows");
    return AVIF_RESULT_OK;
}

bool packRowsForDownload(Scratch & raw, size_t rawOffset, const uint8_t * dev, size_t devPitch, uint8_t * host, size_t hostPitch, size_t widthBytes, size_t rows,
                         hipStream_t stream, hipEvent_t after, CopyWorker::Job * job, avifResult * result)
{
    *result = AVIF_RESULT_OK;
    if (hostPitch != widthBytes || !hostRowsWantOneBlock(host, hostPitch, widthBytes, rows) || !raw.ptr || rawOffset + widthBytes * rows > raw.capacity ||
        (devPitch >> 32) || (widthBytes >> 32) || (rows >> 32))
        return false;
    uint8_t * block = (uint8_t *)raw.ptr + rawOffset;
    const PackArgs a = { dev, block, (uint32_t)devPitch, (uint32_t)widthBytes, (uint32_t)widthBytes, (uint32_t)rows, 0 };
    const hipError_t e = launchPackRows(a, stream);
    if (e != hipSuccess) {
        *result = hipFailed(e, "packing of rows for download");
        return true;
    }
    *job = { after, host, widthBytes * rows, block, widthBytes * rows, widthBytes * rows, 1 }; // one "row": the whole block
    return true;
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
        e = launchRgbToYuvTile(plan, stream, &tls.lastKernel, gTuning.load(std::memory_order_relaxed));
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

namespace {
struct OwnedStreams
{
    std::mutex mutex;
    std::map<hipStream_t, uint64_t> generation;
    uint64_t next = 2; // (1: a context's own stream)
};
OwnedStreams & ownedStreams()
{
    static OwnedStreams * set = new OwnedStreams; // never destroyed, like the context pool
    return *set;
}
} // namespace

uint64_t ownedStreamGeneration(hipStream_t stream)
{
    if (!stream)
        return 0;
    if (stream == tls.stream || stream == tls.upStream || stream == tls.downStream)
        return 1; // lives as long as the context that remembers it
    OwnedStreams & set = ownedStreams();
    std::lock_guard<std::mutex> lock(set.mutex);
    auto it = set.generation.find(stream);
    return it == set.generation.end() ? 0 : it->second;
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

uint32_t bandRowsFor(uint32_t width, uint32_t height)
{
    const uint64_t pixels = (uint64_t)width * height;
    uint32_t bands = (uint32_t)(pixels >> 21);
    bands = bands < 1 ? 1 : (bands > 8 ? 8 : bands);
    uint32_t rows = (height + bands - 1) / bands;
    rows = (rows + 31u) & ~31u;
    return rows < 64 ? 64 : rows;
}

} // namespace api
} // namespace avifhip

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
    OwnedStreams & set = ownedStreams();
    std::lock_guard<std::mutex> lock(set.mutex);
    set.generation[s] = set.next++;
    return (void *)s;
}
extern "C" void avifhipStreamDestroy(void * hipStream)
{
    if (!hipStream)
        return;
    if (tls.scratchPending && tls.scratchStream == (hipStream_t)hipStream) {
        // the stream that last used this thread's device scratch: a later call on another stream could no longer wait for it
        (void)hipStreamSynchronize((hipStream_t)hipStream);
        tls.scratchPending = false;
    }
    for (int k = 0; k < Context::kTableRing; ++k) { // ... or off a resident batch table
        if (tls.tableUnmarked[k] && tls.tableLastStream[k] == (hipStream_t)hipStream) {
            (void)hipStreamSynchronize((hipStream_t)hipStream);
            tls.tableUnmarked[k] = false;
        }
    }
    {
        // (from here on another thread's deferred record on this handle finds no generation and waits for the device instead)
        OwnedStreams & set = ownedStreams();
        std::lock_guard<std::mutex> lock(set.mutex);
        set.generation.erase((hipStream_t)hipStream);
    }
    (void)hipStreamDestroy((hipStream_t)hipStream);
}

extern "C" avifResult avifhipSynchronize(void * hipStream)
{
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    HIP_TRY(hipStreamSynchronize(stream));
    return settleLightLevels(stream, false, true); // (asynchronous gain-map applications that asked for light levels: their clli are filled here)
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
extern "C" uint64_t avifhipTableUploadCount(void)
{
    return tls.tableUploads;
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

// the remaining timing helpers share one shape: `warmup` untimed calls, then `iters` calls between two events on the launch stream
template <typename Call>
static double timeCalls(void * hipStream, int warmup, int iters, Call call)
{
    if (iters <= 0 || ensureContext() != AVIF_RESULT_OK)
        return -1.0;
    hipStream_t stream = pickStream(hipStream);
    for (int k = 0; k < warmup; ++k)
        if (call(k, stream) != AVIF_RESULT_OK)
            return -1.0;
    hipEvent_t t0, t1;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)
        return -1.0;
    (void)hipEventRecord(t0, stream);
    bool ok = true;
    for (int k = 0; k < iters && ok; ++k)
        ok = call(k, stream) == AVIF_RESULT_OK;
    (void)hipEventRecord(t1, stream);
    float ms = -1.0f;
    if (hipEventSynchronize(t1) != hipSuccess || hipEventElapsedTime(&ms, t0, t1) != hipSuccess)
        ms = -1.0f;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return (!ok || ms < 0) ? -1.0 : (double)ms / iters;
}

extern "C" double avifhipTimeRGBToYUVCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, int warmup, int iters, void * hipStream)
{
    if (count == 0 || !images || !rgbs)
        return -1.0;
    return timeCalls(hipStream, warmup, iters, [&](int k, hipStream_t s) { return avifhipImageRGBToYUVAsync(images[k % count], rgbs[k % count], s); });
}

// ... walked `perLaunch` frames at a time through avifhipImageRGBToYUVBatchAsync: milliseconds per LAUNCH
extern "C" double avifhipTimeRGBToYUVBatchCycle(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup, int iters,
                                                void * hipStream)
{
    if (count == 0 || perLaunch == 0 || perLaunch > count || !images || !rgbs)
        return -1.0;
    std::vector<avifImage *> im(perLaunch);
    std::vector<const avifRGBImage *> px(perLaunch);
    return timeCalls(hipStream, warmup, iters, [&](int k, hipStream_t s) {
        for (uint32_t j = 0; j < perLaunch; ++j) {
            const uint32_t f = (uint32_t)(((uint64_t)k * perLaunch + j) % count);
            im[j] = images[f], px[j] = rgbs[f];
        }
        return avifhipImageRGBToYUVBatchAsync(perLaunch, im.data(), px.data(), s);
    });
}

extern "C" double avifhipTimeYUVToRGBBatch(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, const avifCropRect * rects, int warmup,
                                           int iters, void * hipStream)
{
    return timeCalls(hipStream, warmup, iters, [&](int, hipStream_t s) { return avifhipImageYUVToRGBBatchAsync(count, images, rgbs, rects, s); });
}

// launch k converts frames (k * perLaunch + j) % count, j < perLaunch, in ONE launch: a sequence walked `perLaunch` frames at a time, over a
// working set (count frames) the caches cannot hold.  Milliseconds per LAUNCH.
extern "C" double avifhipTimeYUVToRGBBatchCycle(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, uint32_t perLaunch, int warmup,
                                                int iters, void * hipStream)
{
    if (count == 0 || perLaunch == 0 || perLaunch > count || !images || !rgbs)
        return -1.0;
    std::vector<const avifImage *> im(perLaunch);
    std::vector<avifRGBImage *> px(perLaunch);
    return timeCalls(hipStream, warmup, iters, [&](int k, hipStream_t s) {
        for (uint32_t j = 0; j < perLaunch; ++j) {
            const uint32_t f = (uint32_t)(((uint64_t)k * perLaunch + j) % count);
            im[j] = images[f], px[j] = rgbs[f];
        }
        return avifhipImageYUVToRGBBatchAsync(perLaunch, im.data(), px.data(), nullptr, s);
    });
}

extern "C" double avifhipTimeGridYUVToRGB(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles,
                                          avifBool alphaIsLimitedRange, avifRGBImage * rgbCanvas, int warmup, int iters, void * hipStream)
{
    return timeCalls(hipStream, warmup, iters,
                     [&](int, hipStream_t s) { return avifhipGridYUVToRGBAsync(grid, colorTiles, alphaTiles, alphaIsLimitedRange, rgbCanvas, s); });
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

