// api_decode.cpp -- YUV -> RGB of one image: device-resident (async), host-resident in row bands over the three streams, dirty rectangles,
// and the entry points libavif's hooks bind (include/avifhip.h).
#include "api_internal.h"

#include <algorithm>

using namespace avifhip;
using namespace avifhip::api;

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

// Rows [rowBegin, rowEnd) of the conversion on the calling thread's device (the whole image: 0, image->height; a farm worker: its share --
// rowBegin a multiple of 32, so that shares are whole tiles and start on even rows).  Everything has been validated by the caller.
static avifResult yuvToRgbRows(const avifImage * image, avifRGBImage * rgb, bool colorOnly, bool reformatAlpha, uint32_t rowBegin, uint32_t rowEnd)
{
    YuvToRgbPlan probe;
    const avifResult pr = makeYuvToRgbPlan(image, rgb, nullptr, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &probe, colorOnly, reformatAlpha);
    if (pr != AVIF_RESULT_OK)
        return pr;
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
    const uint32_t bandRows = bandRowsFor(image->width, rowEnd - rowBegin);
    const bool banded = pixelsOnHost && bandRows < rowEnd - rowBegin;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    QuiesceOnExit quiesceOnExit; // (destroyed after drainOnExit: the helper thread's downloads first, then the streams)
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    if (pixelsOnHost && rgb->rowBytes == pixelRowBytes && hostRowsWantOneBlock(rgb->pixels, rgb->rowBytes, pixelRowBytes, rowEnd - rowBegin)) {
        r = reserve(tls.rawDown[4], (size_t)pixelRowBytes * (rowEnd - rowBegin)); // (api_internal.h: packRowsForDownload)
        if (r != AVIF_RESULT_OK)
            return r;
    }
    // chroma rows [.., chromaUploaded) that this call needs are on the device (or on their way, on upStream): a share that starts inside the
    // image begins one chroma row above its first (the 4:2:0 filter's upper neighbour)
    uint32_t chromaUploaded = subY ? ((rowBegin >> 1) ? (rowBegin >> 1) - 1 : 0) : rowBegin;
    tls.bytesUp = tls.bytesDown = 0;
    int band = 0;
    for (uint32_t y0 = rowBegin; y0 < rowEnd; y0 += bandRows, ++band) {
        const uint32_t y1 = (y0 + bandRows < rowEnd) ? y0 + bandRows : rowEnd;
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
                r = uploadRows(tls.rawUp[p], dev + (size_t)r0 * devRowBytes, devRowBytes, host + (size_t)r0 * hostRowBytes, hostRowBytes, g.widthBytes[p], r1 - r0, tls.upStream);
                if (r != AVIF_RESULT_OK)
                    return r;
                tls.bytesUp += (uint64_t)g.widthBytes[p] * (r1 - r0);
                uploaded = true;
            }
            if (p == 2 || (p == 1 && !planeOnHost[2]))
                chromaUploaded = r1 > chromaUploaded ? r1 : chromaUploaded;
        }
        if (pixelsOnHost && keepsBytes) {
            r = uploadRows(tls.rawUp[4], rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes,
                           y1 - y0, tls.upStream);
            if (r != AVIF_RESULT_OK)
                return r;
            tls.bytesUp += (uint64_t)pixelRowBytes * (y1 - y0);
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
            CopyWorker::Job job = { tls.bandDone[e], rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes,
                                    pixelRowBytes, y1 - y0 };
            // (tight rows of an unfriendly width -- RGB of odd width: packed on the device, one block back)
            if (packRowsForDownload(tls.rawDown[4], (size_t)(y0 - rowBegin) * pixelRowBytes, rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes,
                                    rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes, y1 - y0, tls.stream, tls.bandDone[e], &job, &r) && r != AVIF_RESULT_OK)
                return r;
            HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
            tls.bytesDown += (uint64_t)pixelRowBytes * (y1 - y0);
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

static bool planesAndPixelsOnHost(const avifImage * image, const avifRGBImage * rgb)
{
    for (int p = 0; p < 4; ++p) {
        const uint8_t * plane = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        if (plane && isDevicePointer(plane))
            return false;
    }
    return !isDevicePointer(rgb->pixels);
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
    // A device set of two or more workers (avifhipSetDeviceSet / AVIFHIP_DEVICES) takes host-resident images of 4 megapixels and more in row
    // shares, one per device (api_farm.cpp).  Shares start on multiples of 32 rows: every share is the same rectangle of the whole-image
    // conversion, byte for byte (chroma edge rules against the image; the halo row either side is uploaded with the share).
    const uint32_t workers = farmWorkers();
    if (workers >= 2) {
        const std::vector<FarmShare> shares = planFarmRows(image->width, image->height, workers);
        if (shares.size() >= 2 && planesAndPixelsOnHost(image, rgb)) {
            struct Call
            {
                const avifImage * image;
                avifRGBImage * rgb;
                bool colorOnly, reformatAlpha;
            } call = { image, rgb, colorOnly, reformatAlpha };
            return farmRun(shares, [](void * arg, uint32_t, FarmShare share) -> avifResult {
                const Call & c = *static_cast<const Call *>(arg);
                return yuvToRgbRows(c.image, c.rgb, c.colorOnly, c.reformatAlpha, share.begin, share.end);
            }, &call);
        }
    }
    tls.farmReports.clear();
    return yuvToRgbRows(image, rgb, colorOnly, reformatAlpha, 0, image->height);
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

// ... and a tall job is cut into pieces of ~2 megapixels (bandRowsFor: multiples of 32 rows), so that the upload of one piece, the kernel of
// the previous and the download of the one before overlap inside a job too (a farm worker often has ONE job: its tile row)
// -- but only as far as the call (or the farm worker's share of it) lacks jobs to overlap: with four jobs or more the pipeline is full as it
// is, and more, smaller copies only cost (cfg5's canvas on one device, 8 full-width jobs: 20.8 ms; cut into 56 pieces: 27.1 ms)
static std::vector<avifCropRect> pipelinePieces(const avifCropRect * jobs, uint32_t count)
{
    std::vector<avifCropRect> pieces;
    const uint32_t wanted = count ? (4u + count - 1) / count : 1u; // pieces per job that bring the call to about four
    for (uint32_t k = 0; k < count; ++k) {
        const avifCropRect & rc = jobs[k];
        const uint32_t finest = bandRowsFor(rc.width, rc.height); // no piece below ~2 megapixels
        uint32_t rows = (rc.height + wanted - 1) / wanted;
        rows = (rows + 31u) & ~31u;
        rows = rows < finest ? finest : rows;
        for (uint32_t y = 0; y < rc.height; y += rows)
            pieces.push_back({ rc.x, rc.y + y, rc.width, (rc.height - y > rows) ? rows : rc.height - y });
    }
    return pieces;
}

// number of farm workers a list of rectangle jobs is worth: the device set's, capped like a whole image's row shares (planFarmRows) at one worker
// per gFarmMinShare pixels of the jobs together -- two small rectangles are not worth a second device, whose worker stages device twins of the
// whole canvas
static uint32_t rectFarmWorkers(const std::vector<avifCropRect> & jobs)
{
    uint64_t pixels = 0;
    for (const avifCropRect & rc : jobs)
        pixels += (uint64_t)rc.width * rc.height;
    const uint32_t workers = farmWorkers(); // (first: reads the environment's share size with its device set)
    const uint64_t byPixels = pixels / farmMinSharePixels();
    return (uint64_t)workers > byPixels ? (uint32_t)byPixels : workers;
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
    // (the pieces are cut per farm worker when a device set is active: the same shares avifhipImageYUVToRGBRects will hand out)
    const std::vector<avifCropRect> coalesced = coalesceRects(rects, count);
    std::vector<avifCropRect> pieces;
    const uint32_t workers = rectFarmWorkers(coalesced);
    if (workers >= 2 && coalesced.size() >= 2) {
        for (const FarmShare & share : planFarmJobs((uint32_t)coalesced.size(), workers)) {
            const std::vector<avifCropRect> part = pipelinePieces(coalesced.data() + share.begin, share.end - share.begin);
            pieces.insert(pieces.end(), part.begin(), part.end());
        }
    } else {
        pieces = pipelinePieces(coalesced.data(), (uint32_t)coalesced.size());
    }
    for (const avifCropRect & rc : pieces) {
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

static avifResult rectJobsOnThisDevice(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * jobs, uint32_t count);

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
    // a device set of two or more workers: contiguous blocks of the (coalesced, row-major) job list, one per device -- whole tile rows when the
    // rectangles are the tiles of a grid (libavif_amd/farm.py: shard)
    const std::vector<avifCropRect> jobs = coalesceRects(rects, count);
    const uint32_t workers = rectFarmWorkers(jobs);
    if (workers >= 2 && jobs.size() >= 2) {
        const std::vector<FarmShare> shares = planFarmJobs((uint32_t)jobs.size(), workers);
        struct Call
        {
            const avifImage * canvas;
            avifRGBImage * rgbCanvas;
            const std::vector<avifCropRect> * jobs;
        } call = { canvas, rgbCanvas, &jobs };
        return farmRun(shares, [](void * arg, uint32_t, FarmShare share) -> avifResult {
            const Call & c = *static_cast<const Call *>(arg);
            return rectJobsOnThisDevice(c.canvas, c.rgbCanvas, c.jobs->data() + share.begin, share.end - share.begin);
        }, &call);
    }
    tls.farmReports.clear();
    return rectJobsOnThisDevice(canvas, rgbCanvas, jobs.data(), (uint32_t)jobs.size());
}

// the coalesced jobs `jobs[0 .. count)` of avifhipImageYUVToRGBRects on the calling thread's device
static avifResult rectJobsOnThisDevice(const avifImage * canvas, avifRGBImage * rgbCanvas, const avifCropRect * coalesced, uint32_t coalescedCount)
{
    const std::vector<avifCropRect> jobs = pipelinePieces(coalesced, coalescedCount);
    const uint32_t count = (uint32_t)jobs.size();
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
    QuiesceOnExit quiesceOnExit; // (destroyed after drainOnExit: the helper thread's downloads first, then the streams)
    DrainOnExit drainOnExit = { tls.downloader };
    const uint32_t bps = (canvas->depth > 8) ? 2 : 1, px = rgbPixelBytes(rgbCanvas);
    tls.bytesUp = tls.bytesDown = 0;
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
            r = uploadRows(tls.rawUp[p], dev + (size_t)W.y0[p] * devPitch + (size_t)W.x0[p] * bps, devPitch, host + (size_t)W.y0[p] * hostPitch + (size_t)W.x0[p] * bps, hostPitch,
                           (size_t)W.w[p] * bps, W.h[p], tls.upStream);
            if (r != AVIF_RESULT_OK)
                return r;
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

