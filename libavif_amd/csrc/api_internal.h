// api_internal.h -- what the translation units behind the C ABI share (api.cpp: context and control; api_decode / _batch / _encode / _apps.cpp:
// the conversion path; api_gainmap.cpp: gain maps; api_scale.cpp: plane scaling): the per-thread context (stream, device scratch, table caches), error plumbing, staging helpers.
// Internal to libavifhip.so (hidden visibility).
#pragma once

#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "avifhip.h"
#include "gainmap_plan.h"
#include "kernels.h"
#include "plan.h"
#include "scale_plan.h"

// Pinned host memory comes from the (uninstrumented) HIP runtime: when it hands a block one thread's context freed to another thread's context,
// ThreadSanitizer has not seen the allocator's own synchronisation between the free and the allocation, still holds the previous owner's
// accesses for those addresses, and reports the new owner's first write as a race (profiles/r06_tsan.txt: two contexts' upload rings, the
// smaller of which had just been replaced by a larger one).  The `make tsan` build says what an allocator it could see would have said:
// everything before the free happens before everything after the allocation that returns the same address.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
extern "C" void AnnotateHappensBefore(const char * file, int line, const volatile void * address);
extern "C" void AnnotateHappensAfter(const char * file, int line, const volatile void * address);
#define AVIFHIP_HOST_MEMORY_FREED(p) AnnotateHappensBefore(__FILE__, __LINE__, (p))
#define AVIFHIP_NEW_HOST_MEMORY(p, n) AnnotateHappensAfter(__FILE__, __LINE__, (p))
#endif
#endif
#ifndef AVIFHIP_NEW_HOST_MEMORY
#define AVIFHIP_HOST_MEMORY_FREED(p) ((void)0)
#define AVIFHIP_NEW_HOST_MEMORY(p, n) ((void)0)
#endif

namespace avifhip {
namespace api {

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
    int exactBox[4] = { 0, 0, 0, 0 };                  // exact N x N boxes: the box kernel when the buffers' alignment allows
    bool doubling[4] = { false, false, false, false }; // 2x on both axes: the doubling kernel when the buffers' alignment allows
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
    size_t baseLutOffset = 0, gainLutOffset = 0, stepsOffset = 0, guideOffset = 0, locOffset = 0, alphaOffset = 0; // in floats
    uint32_t maxCode = 0, nanCode = 0, stepEntries = 0;
    uint32_t locBuckets = 0, locFirstBits = 0, locShift = 0; // GainMapSteps::locator (0 buckets: none for this curve / depth)
    float baseMax = 0.0f, gainMax[3] = { 0.0f, 0.0f, 0.0f };    // largest magnitudes in the base / gain tables (infinity: a NaN or inf entry)
};

// Copies between pageable host memory and the device run at full PCIe rate on this platform, but "asynchronous" ones block the
// CALLING THREAD until they are done (tests/tools/pcie_probe.hip: hipMemcpyAsync from/to malloc'ed memory returns after the
// whole transfer; two directions issued by one thread take the sum of their times, by two threads the maximum).  To use both
// directions of the link at once, a host-resident call hands its downloads to this helper thread while its own thread keeps
// uploading and launching.  One helper per context, started at the first banded call, parked on a condition variable between
// calls.
class CopyWorker
{
public:
    struct Job
    {
        hipEvent_t after; // the copy may start once this event has completed
        void * dst;
        size_t dstPitch;
        const void * src;
        size_t srcPitch, widthBytes, rows;
    };
    CopyWorker(int device, hipStream_t stream);
    ~CopyWorker();
    void post(const Job & job);
    hipError_t drain(); // blocks until every posted job is done; the first error since the last drain (hipSuccess if none)

private:
    void run();
    int device_;
    hipStream_t stream_;
    std::mutex mutex_;
    std::condition_variable wake_, idle_;
    std::deque<Job> queue_;
    int pending_ = 0;
    bool stop_ = false;
    hipError_t error_ = hipSuccess;
    std::thread thread_;
};

// One context per calling thread at a time: libavif's reformat functions are re-entrant and may be called concurrently from up
// to 8 threads (src/reformat.c:1709-1735); nothing here is shared between threads that run at the same time.  Contexts are
// LEASED from a process-wide pool (currentContext()): libavif creates its worker threads anew for every call
// (src/reformat.c:1625-1638), and a thread that ends hands its context -- streams, events, device scratch, pinned staging -- back
// for the next one instead of destroying it.
struct Context
{
    int device = -1;
    bool ready = false; // streams and events all exist (ensureContext commits only a completely built context)
    int ownerPid = 0;   // the process that built it: a fork()ed child must not use the parent's streams or wait for its helper thread
    hipStream_t stream = nullptr;
    // host-resident calls split large images into row bands: uploads and downloads of different bands run on these two while
    // `stream` computes (api_decode.cpp: yuvToRgbSync)
    hipStream_t upStream = nullptr, downStream = nullptr;
    static constexpr int kMaxBands = 16;
    hipEvent_t bandUp[kMaxBands] = {}, bandDone[kMaxBands] = {};
    CopyWorker * downloader = nullptr; // issues the downloads of banded host-resident calls (created on first use)
    Scratch planes[4]; // Y, U, V, A staging
    Scratch pixels;    // interleaved RGB staging
    // rows whose host pitch, width or address is not a multiple of 4 cross the link as ONE block per band and are re-pitched on the device
    // (uploadRows / packRowsForDownload): [0..3] planes, [4] pixels
    Scratch rawUp[5], rawDown[5];
    Scratch table;     // batch descriptor table (device)
    Scratch scaleTable; // schedules of a plane scale (device)
    ScaleTableCache scaleCache; // ... and which geometry they belong to
    Scratch satoTable;  // input plane tables of a sample transform (device)
    Scratch xformCanvas; // canvas-sized RGB of the two-pass route of avifhip*TransformedAsync (device)
    Scratch gainMap[12]; // gain maps: [0] output pixels, [1] gain map as RGB, [2] tables, [3] statistics / partials, [4] scaled planes, [5] base pixels;
                         // computation: [6] tables, [7] ratios, [8] histograms, [9] alternate pixels, [10] gain-map planes
    GainMapTableCache gainMapCache; // what gainMap[2] holds
    int gainMapTimeWarmup = 0, gainMapTimeIters = 0; // avifhipTimeRGBImageApplyGainMap in progress on this thread: repeat the apply kernel
    double gainMapTimedMs = -1.0;                    // ... and what it measured
    void * gainMapPartials = nullptr; // apply: the statistics as the workgroups leave them (pinned host memory, kGainMapMaxGroups partials)
    // ... of asynchronous calls that asked for light levels (avifhipRGBImageApplyGainMapAsync with clli): a ring of pinned slots, each filled by a
    // copy behind the kernel (the event says when) and turned into the caller's clli by the thread's next avifhipSynchronize on that stream
    // (api.cpp settleLightLevels; a host function on the stream cost 30 us per call)
    static constexpr int kLightSlots = 8;
    struct PendingLight
    {
        bool pending = false;
        uint32_t count = 0;   // partials in the slot
        size_t pixels = 0;
        void * clli = nullptr; // avifContentLightLevelInformationBox * of the caller
        hipStream_t stream = nullptr;
    };
    void * lightPinned[kLightSlots] = {};
    hipEvent_t lightCopied[kLightSlots] = {};
    PendingLight lightPending[kLightSlots];
    uint32_t lightSlot = 0;
    // batch descriptor tables travel through a ring of kTableRing slots (pinned host memory + the matching slice of `table`), uploaded on
    // `upStream`: the host prepares batch n + 1 and its table crosses the link while batch n computes (api_batch.cpp: batchAsyncImpl)
    static constexpr int kTableRing = 4;
    void * pinnedTable = nullptr;
    size_t pinnedTableCapacity = 0; // bytes per slot
    uint32_t tableSlot = 0;
    hipEvent_t tableCopied[kTableRing] = {};   // the slot's upload has left the pinned memory (and the device slice holds it)
    hipEvent_t tableConsumed[kTableRing] = {}; // the kernels reading the slot's device slice are done
    bool tableUnmarked[kTableRing] = {};        // the slot's last readers left no tableConsumed record (they ran off the resident copy) ...
    hipStream_t tableLastStream[kTableRing] = {}; // ... on this stream: recorded when an upload wants the slot, or at the context's hand-back
    uint64_t tableLastGeneration[kTableRing] = {}; // ... which was an owned stream of this generation (ownedStreamGeneration)
    bool tableSlotIdle[kTableRing] = { true, true, true, true }; // no upload out of the slot's pinned memory since the thread last waited for tableCopied
    // the most recent upload, if small: a batch whose table is byte-identical launches on the copy the device still holds (batchAsyncImpl)
    static constexpr size_t kResidentTableMax = 256 * 1024;
    int residentSlot = -1;
    size_t residentBytes = 0;
    hipStream_t residentStream = nullptr;
    uint64_t residentGeneration = 0;
    bool residentAllTiled = false;
    void * pinnedUpload = nullptr; // staging for small host tables (grid tile tables, scale schedules): a ring of kTableRing slots, so that the
    size_t pinnedUploadCapacity = 0; // calling thread does not wait for the previous call's upload (which sits behind that call's kernels); bytes per slot
    uint32_t uploadSlot = 0;
    hipEvent_t uploadCopied[kTableRing] = {};
    // last asynchronous user of the device scratch above (ScratchScope)
    hipEvent_t scratchUsed = nullptr;
    hipStream_t scratchStream = nullptr;
    uint64_t scratchGeneration = 0; // of scratchStream when it was noted (0: a caller's own stream -- its use was marked right away)
    bool scratchPending = false;
    bool scratchMarked = false; // scratchUsed already covers the last use (recorded when the thread handed the context back)
    char lastError[512] = { 0 };
    const char * lastKernel = "";
    uint64_t launches = 0; // kernels enqueued by this thread
    uint64_t tableUploads = 0; // batch tables sent to the device by this thread
    uint64_t bytesUp = 0, bytesDown = 0; // host link traffic of the thread's last host-resident conversion (avifhipLastTransferBytes)
    char lastKernelText[128] = { 0 };    // a farmed call's kernel name, copied from the worker that ran it (lastKernel may point here)
    // the last farmed call of this thread: one entry per worker that took part (avifhipLastFarmWorkers / avifhipLastFarmTransferBytes)
    struct FarmReport
    {
        int device;
        uint32_t rowBegin, rowEnd; // the worker's share (rows of the image; rectangles: indices of the coalesced job list)
        uint64_t bytesUp, bytesDown;
    };
    std::vector<FarmReport> farmReports;

    ~Context()
    {
        // Best effort: the runtime may already be shutting down at thread/process exit.
        for (Scratch & s : planes)
            if (s.ptr)
                (void)hipFree(s.ptr);
        if (pixels.ptr)
            (void)hipFree(pixels.ptr);
        for (Scratch & s : rawUp)
            if (s.ptr)
                (void)hipFree(s.ptr);
        for (Scratch & s : rawDown)
            if (s.ptr)
                (void)hipFree(s.ptr);
        if (table.ptr)
            (void)hipFree(table.ptr);
        if (scaleTable.ptr)
            (void)hipFree(scaleTable.ptr);
        if (satoTable.ptr)
            (void)hipFree(satoTable.ptr);
        if (xformCanvas.ptr)
            (void)hipFree(xformCanvas.ptr);
        for (Scratch & g : gainMap)
            if (g.ptr)
                (void)hipFree(g.ptr);
        if (gainMapPartials)
            AVIFHIP_HOST_MEMORY_FREED(gainMapPartials), (void)hipHostFree(gainMapPartials);
        for (int k = 0; k < kLightSlots; ++k) { // (pending light levels of a thread that ends without synchronising are dropped: its clli may be gone)
            if (lightCopied[k]) {
                (void)hipEventSynchronize(lightCopied[k]);
                (void)hipEventDestroy(lightCopied[k]);
            }
            if (lightPinned[k])
                AVIFHIP_HOST_MEMORY_FREED(lightPinned[k]), (void)hipHostFree(lightPinned[k]);
        }
        if (pinnedTable)
            AVIFHIP_HOST_MEMORY_FREED(pinnedTable), (void)hipHostFree(pinnedTable);
        for (int k = 0; k < kTableRing; ++k) {
            if (tableCopied[k])
                (void)hipEventDestroy(tableCopied[k]);
            if (tableConsumed[k])
                (void)hipEventDestroy(tableConsumed[k]);
        }
        if (pinnedUpload)
            AVIFHIP_HOST_MEMORY_FREED(pinnedUpload), (void)hipHostFree(pinnedUpload);
        for (int k = 0; k < kTableRing; ++k)
            if (uploadCopied[k])
                (void)hipEventDestroy(uploadCopied[k]);
        if (scratchUsed)
            (void)hipEventDestroy(scratchUsed);
        delete downloader;
        for (int b = 0; b < kMaxBands; ++b) {
            if (bandUp[b])
                (void)hipEventDestroy(bandUp[b]);
            if (bandDone[b])
                (void)hipEventDestroy(bandDone[b]);
        }
        if (upStream)
            (void)hipStreamDestroy(upStream);
        if (downStream)
            (void)hipStreamDestroy(downStream);
        if (stream)
            (void)hipStreamDestroy(stream);
    }
};

// the calling thread's context (leased from the pool at the thread's first call, returned when the thread ends)
Context & currentContext();
#define tls (::avifhip::api::currentContext())
extern std::atomic<int> gTiledKernels;

void setError(const char * fmt, ...);
// HIP failure -> avifResult.  The message is kept for avifhipLastError(); the sticky HIP error is cleared.
avifResult hipFailed(hipError_t e, const char * what);

#define HIP_TRY(expr)                          \
    do {                                       \
        const hipError_t hipTryErr_ = (expr);  \
        if (hipTryErr_ != hipSuccess)          \
            return ::avifhip::api::hipFailed(hipTryErr_, #expr); \
    } while (0)

avifResult ensureContext();
// The per-thread device scratch (descriptor tables, schedules, gain-map work buffers) is shared by every asynchronous entry point
// that needs any, WHATEVER stream the caller passes: without ordering, a call on stream B could rewrite a table that a kernel
// enqueued earlier on stream A is still reading.  An entry point that touches scratch opens a ScratchScope on its stream: the
// constructor makes that stream wait for the scratch's previous user when that was a different stream (the event is recorded on the
// previous user's stream at that moment), the destructor notes the stream as the new last user.  (Entry points that use no scratch --
// single-image conversions -- pay nothing; calls that stay on one stream no event.)
struct ScratchScope
{
    hipStream_t stream;
    avifResult result;
    explicit ScratchScope(hipStream_t s);
    ~ScratchScope();
    ScratchScope(const ScratchScope &) = delete;
    ScratchScope & operator=(const ScratchScope &) = delete;
};
// Enqueues a copy of a small host table to device memory through a pinned per-thread staging buffer (api.cpp: uploadTableAsync)
avifResult uploadTableAsync(void * deviceDst, const void * hostSrc, size_t bytes, hipStream_t stream);
avifResult reserve(Scratch & s, size_t bytes);
// hipMemcpy2DAsync between pageable host memory and the device degenerates into one copy per row -- ~9 us each -- when the host pitch, the row
// width or the host address is not a multiple of 4 (tests/c/farm_check, 12 megapixels host to host: 1.1 ms at 4096 pixels per row, 56 ms at
// 4098, 29 ms at 4100 whose chroma rows are 2050 bytes): every libavif image of odd width, whose planes avifImageAllocatePlanes packs tight.
// Such rows cross the link as ONE block per band instead (host rows are contiguous in memory, pitch by pitch) and change their pitch on the
// device.  true: this block of rows wants that treatment (an unfriendly alignment, enough rows to matter, and a pitch that does not drag
// more than twice the rows' own bytes along -- a narrow view into a wide canvas keeps the 2-D copy).
bool hostRowsWantOneBlock(const void * host, size_t hostPitch, size_t widthBytes, size_t rows);
// host rows -> device rows at `devPitch` (a multiple of 4), enqueued on `stream`: a 2-D copy, or one block into `raw` and a re-pitch kernel
avifResult uploadRows(Scratch & raw, uint8_t * dev, size_t devPitch, const uint8_t * host, size_t hostPitch, size_t widthBytes, size_t rows, hipStream_t stream);
// The way back for TIGHT host rows (pitch == width: nothing between the rows that a block copy could overwrite): enqueues, on `stream`, the
// packing of `rows` device rows into `raw` at byte offset `rawOffset` and fills `job` with the one-block download of them; false (nothing
// enqueued): use the 2-D copy.  `raw` must have been reserved for the whole call before its first band (reserve()).
bool packRowsForDownload(Scratch & raw, size_t rawOffset, const uint8_t * dev, size_t devPitch, uint8_t * host, size_t hostPitch, size_t widthBytes, size_t rows,
                         hipStream_t stream, hipEvent_t after, CopyWorker::Job * job, avifResult * result);
bool isDevicePointer(const void * p);
inline uint32_t alignUp(uint32_t v, uint32_t a)
{
    return (v + a - 1) / a * a;
}
hipStream_t pickStream(void * hipStream);
// Streams the library made -- the context's own, and those of avifhipStreamCreate until avifhipStreamDestroy -- carry a generation number
// (never reused); 0 for any other handle (a hipStream_t of the caller's, which may be destroyed, and its address reused, behind the
// library's back).  Only owned streams may be remembered past a call: the deferred event records of the device scratch and of the batch
// tables, and the "this stream has already waited for the resident table" shortcut, check the generation before they trust a handle.
uint64_t ownedStreamGeneration(hipStream_t stream);

struct PlaneGeometry
{
    uint32_t widthBytes[4];
    uint32_t rows[4];
};
PlaneGeometry planeGeometry(const avifImage * image);
// Replaces host plane pointers of `view` (a shallow copy of the caller's image) with device copies (tls.planes)
avifResult stagePlanes(avifImage * view, bool upload, bool mirrorRowBytes);
uint32_t rgbPixelBytes(const avifRGBImage * rgb);
avifResult stagePixels(avifRGBImage * view, bool upload);
// malloc'ed planes like avifImageAllocatePlanes (src/avif.c:431-490): only the missing ones
avifResult allocateHostPlanes(avifImage * image, bool withAlpha);

// plane sizes of an image (avifImagePlaneWidth / Height, reference src/avif.c:351-400)
struct PlaneDims
{
    int w[4], h[4];
};
inline PlaneDims planeDims(uint32_t width, uint32_t height, int yuvFormat)
{
    const int sx = (yuvFormat == AVIF_PIXEL_FORMAT_YUV444 || yuvFormat == AVIF_PIXEL_FORMAT_YUV400) ? 0 : 1;
    const int sy = (yuvFormat == AVIF_PIXEL_FORMAT_YUV420) ? 1 : 0;
    PlaneDims d;
    d.w[0] = d.w[3] = (int)width, d.h[0] = d.h[3] = (int)height;
    d.w[1] = d.w[2] = (int)((width + sx) >> sx), d.h[1] = d.h[2] = (int)((height + sy) >> sy);
    return d;
}

// library-wide settings (avifhipSetArithmetic / SetTiledKernels / SetTuning)
extern std::atomic<int> gArithmetic;
extern std::atomic<int> gTiledKernels;
extern std::atomic<uint32_t> gTuning;
int effectiveArithmetic();
// launches of the planned conversions on `stream` (tiled kernels where the plan allows, the universal kernel otherwise)
avifResult enqueueYuvToRgb(const YuvToRgbPlan & plan, hipStream_t stream);
// fills the clli of the calling thread's asynchronous gain-map applications whose copies have arrived (`stream`: only that stream's; nullptr:
// every stream's), waiting for them when `wait` (api_gainmap.cpp)
avifResult settleLightLevels(hipStream_t stream, bool everyStream, bool wait);
avifResult enqueueRgbToYuv(const RgbToYuvPlan & plan, hipStream_t stream);
avifResult enqueueAlphaMul(const AlphaMulPlan & plan, hipStream_t stream);
void finishRgbToYuvPlan(const avifImage * image, const avifRGBImage * rgb, RgbToYuvPlan * plan);
bool sharpYuvRequested(const avifImage * image, const avifRGBImage * rgb);
// row bands of a host-resident call (api_decode.cpp: yuvToRgbSync): rows per band -- multiples of 32 rows, at least ~2 megapixels each
uint32_t bandRowsFor(uint32_t width, uint32_t height);

// ---- the in-process device farm (api_farm.cpp; SURVEY.md 8e: one host thread + streams + staging per GPU) ----
// A host-resident call whose image is large enough is cut into contiguous row shares, one per worker of the device set
// (avifhipSetDeviceSet / AVIFHIP_DEVICES); every worker is a persistent thread with a pooled context of its own on its device, uploads only
// what its share needs (its rows plus the chroma filter's halo row either side -- uploaded, not exchanged), converts with the ordinary banded
// path and downloads its rows over its own host link.  No device talks to another.
struct FarmShare
{
    uint32_t begin, end; // rows [begin, end) of the image, or entries of a job list
};
// contiguous shares of `height` rows for at most `workers` workers: multiples of 32 rows (whole tiles of the tiled kernels, even rows for
// subsampled chroma), at least ~2 megapixels each (below that a second device costs more than it brings); one share = not farmed
std::vector<FarmShare> planFarmRows(uint32_t width, uint32_t height, uint32_t workers);
// contiguous blocks of a list of `count` jobs (libavif_amd/farm.py: shard -- the first count % workers blocks are one longer)
std::vector<FarmShare> planFarmJobs(uint32_t count, uint32_t workers);
// number of workers of the current device set (0 or 1: calls run on the calling thread's own device as ever)
uint32_t farmWorkers();
// smallest share worth a device of its own, in pixels (avifhipSetFarmMinSharePixels / AVIFHIP_FARM_MIN_PIXELS; never 0)
uint64_t farmMinSharePixels();
// Runs job(k, shares[k]) on worker k for every share and waits for all of them.  The result is the first failure in share order (its error text
// becomes the calling thread's), AVIF_RESULT_OK otherwise; the calling thread's launch count, transfer bytes, kernel name and farm report are
// updated from the workers'.
avifResult farmRun(const std::vector<FarmShare> & shares, avifResult (*job)(void * arg, uint32_t worker, FarmShare share), void * arg);
// Host-resident entry points: however a call ends -- an error half way included -- nothing it enqueued may still be reading (uploads) or writing
// (downloads) the caller's memory when it returns.  An "asynchronous" copy from pageable memory blocks its caller on this platform
// UNLESS the runtime still has that memory pinned from an earlier copy: then it returns at once, and a call that gave up after its uploads were
// enqueued (an unsupported colour space found later, say) handed the caller back memory the GPU was still reading -- a "Memory access fault by
// GPU" once the caller freed it.  Declared
// before the first upload of every host-resident entry point; the streams of a call that ended normally are idle already.
struct QuiesceOnExit
{
    ~QuiesceOnExit()
    {
        Context & c = tls;
        if (c.upStream)
            (void)hipStreamSynchronize(c.upStream);
        if (c.stream)
            (void)hipStreamSynchronize(c.stream);
        if (c.downStream)
            (void)hipStreamSynchronize(c.downStream);
    }
};

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

} // namespace api
} // namespace avifhip
