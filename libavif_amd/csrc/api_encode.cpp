// api_encode.cpp -- RGB -> YUV, alpha (un)premultiply and the half-float conversion: device-resident (async) and host-resident in row bands.
#include "api_internal.h"

#include <algorithm>

using namespace avifhip;
using namespace avifhip::api;

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

// `count` device-resident frames in as few launches as their configurations allow (include/avifhip.h): frames that differ in their buffers
// only -- an image sequence on its way into an encoder -- share launches of the single-image kernel, up to 8 per launch (kernels.h
// launchRgbToYuvTileSequence); anything else is converted frame by frame, exactly like `count` calls of avifhipImageRGBToYUVAsync.
extern "C" avifResult avifhipImageRGBToYUVBatchAsync(uint32_t count, avifImage * const * images, const avifRGBImage * const * rgbs, void * hipStream)
{
    if (count == 0)
        return AVIF_RESULT_OK;
    if (!images || !rgbs)
        return AVIF_RESULT_INVALID_ARGUMENT;
    std::vector<RgbToYuvPlan> plans(count);
    const uint32_t tuning = gTuning.load(std::memory_order_relaxed);
    bool sequence = gTiledKernels.load(std::memory_order_relaxed) != 0;
    for (uint32_t k = 0; k < count; ++k) {
        avifImage * image = images[k];
        const avifRGBImage * rgb = rgbs[k];
        if (!image || !rgb)
            return AVIF_RESULT_INVALID_ARGUMENT;
        const avifResult pr = makeRgbToYuvPlan(image, rgb, effectiveArithmetic(), &plans[k]);
        if (pr != AVIF_RESULT_OK)
            return pr;
        if (sharpYuvRequested(image, rgb))
            return AVIF_RESULT_NOT_IMPLEMENTED; // (like avifhipImageRGBToYUVAsync)
        const bool needAlpha = plans[k].rgb.hasAlpha && !rgb->ignoreAlpha;
        if (!image->yuvPlanes[0] || (image->yuvFormat != AVIF_PIXEL_FORMAT_YUV400 && (!image->yuvPlanes[1] || !image->yuvPlanes[2])) ||
            (needAlpha && !image->alphaPlane)) {
            setError("avifhipImageRGBToYUVBatchAsync: destination planes must be allocated by the caller (frame %u)", k);
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
        finishRgbToYuvPlan(image, rgb, &plans[k]);
        sequence = sequence && tileRgbToYuvSequenceCompatible(plans[0], plans[k], tuning);
    }
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    if (!sequence) {
        for (uint32_t k = 0; k < count; ++k) {
            const avifResult r = enqueueRgbToYuv(plans[k], stream);
            if (r != AVIF_RESULT_OK)
                return r;
        }
        return AVIF_RESULT_OK;
    }
    for (uint32_t first = 0; first < count; first += kRgbToYuvSequenceMax) {
        const uint32_t n = count - first < kRgbToYuvSequenceMax ? count - first : kRgbToYuvSequenceMax;
        const hipError_t e = launchRgbToYuvTileSequence(plans.data() + first, n, stream, &tls.lastKernel, tuning);
        if (e != hipSuccess)
            return hipFailed(e, "RGB->YUV sequence kernel launch");
        ++tls.launches;
    }
    return AVIF_RESULT_OK;
}

// Rows [rowBegin, rowEnd) of the conversion on the calling thread's device (the whole image, or a farm worker's share: rowBegin even, so
// that no 2 x 2 block is cut).  Arguments validated and destination planes allocated by the caller.
static avifResult rgbToYuvRows(avifImage * image, const avifRGBImage * rgb, uint32_t rowBegin, uint32_t rowEnd)
{
    RgbToYuvPlan plan;
    avifResult r = makeRgbToYuvPlan(image, rgb, effectiveArithmetic(), &plan);
    if (r != AVIF_RESULT_OK)
        return r;
    const bool pixelsOnHost = !isDevicePointer(rgb->pixels);
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
    const uint32_t bandRows = gray ? image->height : bandRowsFor(image->width, rowEnd - rowBegin);
    const bool banded = bandRows < rowEnd - rowBegin;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    QuiesceOnExit quiesceOnExit; // (destroyed after drainOnExit: the helper thread's downloads first, then the streams)
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    tls.bytesUp = tls.bytesDown = 0;
    for (int p = 0; p < 4 && !gray; ++p) {
        uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
        const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
        const bool chroma = p == 1 || p == 2;
        const uint32_t rows = chroma && subY ? ((rowEnd + 1) >> 1) - (rowBegin >> 1) : rowEnd - rowBegin;
        if (host && hostRowBytes == g.widthBytes[p] && !isDevicePointer(host) && hostRowsWantOneBlock(host, hostRowBytes, g.widthBytes[p], rows)) {
            r = reserve(tls.rawDown[p], (size_t)g.widthBytes[p] * rows);
            if (r != AVIF_RESULT_OK)
                return r;
        }
    }
    int band = 0;
    for (uint32_t y0 = rowBegin; y0 < rowEnd; y0 += bandRows, ++band) {
        const uint32_t y1 = (y0 + bandRows < rowEnd) ? y0 + bandRows : rowEnd;
        const int e = band % Context::kMaxBands;
        const uint32_t c0 = subY ? (y0 >> 1) : y0, c1 = subY ? ((y1 + 1) >> 1) : y1; // chroma rows of the band
        if (pixelsOnHost) {
            r = uploadRows(tls.rawUp[4], rgbView.pixels + (size_t)y0 * rgbView.rowBytes, rgbView.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes,
                           y1 - y0, tls.upStream);
            if (r != AVIF_RESULT_OK)
                return r;
            tls.bytesUp += (uint64_t)pixelRowBytes * (y1 - y0);
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
        // the downloads of the band's plane rows: 2-D copies, or -- tight rows of an unfriendly width (every image of odd width whose planes
        // avifImageAllocatePlanes made) -- packed on the device behind the conversion and fetched as one block (api_internal.h)
        CopyWorker::Job planeJob[4];
        for (int p = 0; p < 4; ++p) {
            uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
            const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
            const uint8_t * dev = (p < 3) ? imageView.yuvPlanes[p] : imageView.alphaPlane;
            const uint32_t devRowBytes = (p < 3) ? imageView.yuvRowBytes[p] : imageView.alphaRowBytes;
            const bool chroma = p == 1 || p == 2;
            if (!host || !hostRowBytes || host == dev || (chroma && (gray || image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400)))
                continue;
            const uint32_t r0 = chroma ? c0 : y0, r1 = chroma ? c1 : y1, first = chroma ? (subY ? rowBegin >> 1 : rowBegin) : rowBegin;
            planeJob[p] = { tls.bandDone[e], host + (size_t)r0 * hostRowBytes, hostRowBytes, dev + (size_t)r0 * devRowBytes, devRowBytes, g.widthBytes[p], r1 - r0 };
            if (packRowsForDownload(tls.rawDown[p], (size_t)(r0 - first) * g.widthBytes[p], dev + (size_t)r0 * devRowBytes, devRowBytes, host + (size_t)r0 * hostRowBytes, hostRowBytes,
                                    g.widthBytes[p], r1 - r0, tls.stream, tls.bandDone[e], &planeJob[p], &r) && r != AVIF_RESULT_OK)
                return r;
        }
        HIP_TRY(hipEventRecord(tls.bandDone[e], tls.stream));
        if (!banded)
            HIP_TRY(hipStreamWaitEvent(tls.downStream, tls.bandDone[e], 0));
        for (int p = 0; p < 4; ++p) {
            uint8_t * host = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
            const uint32_t hostRowBytes = (p < 3) ? image->yuvRowBytes[p] : image->alphaRowBytes;
            const uint8_t * dev = (p < 3) ? imageView.yuvPlanes[p] : imageView.alphaPlane;
            if (!host || !hostRowBytes || host == dev)
                continue; // absent, or already device-resident
            const bool chroma = p == 1 || p == 2;
            const uint32_t r0 = chroma ? c0 : y0, r1 = chroma ? c1 : y1;
            if (gray && chroma) {
                HIP_TRY(hipMemcpyAsync(host, dev, (size_t)hostRowBytes * g.rows[p], hipMemcpyDeviceToHost, tls.downStream));
                tls.bytesDown += (uint64_t)hostRowBytes * g.rows[p];
                continue;
            } else if (chroma && image->yuvFormat == AVIF_PIXEL_FORMAT_YUV400) {
                continue; // colour source into 4:0:0: chroma untouched
            }
            tls.bytesDown += (uint64_t)g.widthBytes[p] * (r1 - r0);
            const CopyWorker::Job & job = planeJob[p];
            if (banded) {
                tls.downloader->post(job);
            } else {
                HIP_TRY(hipMemcpy2DAsync(job.dst, job.dstPitch, job.src, job.srcPitch, job.widthBytes, job.rows, hipMemcpyDeviceToHost, tls.downStream));
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
    // a device set of two or more workers takes host-resident images of 4 megapixels and more in row shares (api_farm.cpp); gray sources
    // keep their single pass (their chroma planes are filled pitch-wide by one launch)
    const uint32_t workers = farmWorkers();
    if (workers >= 2 && pixelsOnHost && !rgbFormatIsGray((int)rgb->format)) {
        bool planesOnHost = true;
        for (int p = 0; p < 4; ++p) {
            const uint8_t * plane = (p < 3) ? image->yuvPlanes[p] : image->alphaPlane;
            planesOnHost = planesOnHost && !(plane && isDevicePointer(plane));
        }
        const std::vector<FarmShare> shares = planFarmRows(image->width, image->height, workers);
        if (planesOnHost && shares.size() >= 2) {
            struct Call
            {
                avifImage * image;
                const avifRGBImage * rgb;
            } call = { image, rgb };
            return farmRun(shares, [](void * arg, uint32_t, FarmShare share) -> avifResult {
                const Call & c = *static_cast<const Call *>(arg);
                return rgbToYuvRows(c.image, c.rgb, share.begin, share.end);
            }, &call);
        }
    }
    tls.farmReports.clear();
    return rgbToYuvRows(image, rgb, 0, image->height);
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
static avifResult inPlaceBandedRows(avifRGBImage * rgb, uint32_t pixelRowBytes, Launch & launch, uint32_t rowBegin, uint32_t rowEnd)
{
    avifResult r = ensureContext();
    if (r != AVIF_RESULT_OK)
        return r;
    avifRGBImage view = *rgb;
    r = stagePixels(&view, /*upload=*/false);
    if (r != AVIF_RESULT_OK)
        return r;
    const uint32_t bandRows = bandRowsFor(rgb->width, rowEnd - rowBegin);
    const bool banded = bandRows < rowEnd - rowBegin;
    if (banded && !tls.downloader)
        tls.downloader = new CopyWorker(tls.device, tls.downStream);
    QuiesceOnExit quiesceOnExit; // (destroyed after drainOnExit: the helper thread's downloads first, then the streams)
    DrainOnExit drainOnExit = { banded ? tls.downloader : nullptr };
    tls.bytesUp = tls.bytesDown = 0;
    int band = 0;
    for (uint32_t y0 = rowBegin; y0 < rowEnd; y0 += bandRows, ++band) {
        const uint32_t rows = (y0 + bandRows < rowEnd) ? bandRows : rowEnd - y0;
        tls.bytesUp += (uint64_t)pixelRowBytes * rows, tls.bytesDown += (uint64_t)pixelRowBytes * rows;
        const int e = band % Context::kMaxBands;
        r = uploadRows(tls.rawUp[4], view.pixels + (size_t)y0 * view.rowBytes, view.rowBytes, rgb->pixels + (size_t)y0 * rgb->rowBytes, rgb->rowBytes, pixelRowBytes, rows, tls.upStream);
        if (r != AVIF_RESULT_OK)
            return r;
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

// ... over the whole image: on the calling thread's device, or in row shares over the device set (api_farm.cpp; every row is independent)
template <class Launch>
static avifResult inPlaceBanded(avifRGBImage * rgb, uint32_t pixelRowBytes, Launch launch)
{
    const uint32_t workers = farmWorkers();
    if (workers >= 2) {
        const std::vector<FarmShare> shares = planFarmRows(rgb->width, rgb->height, workers);
        if (shares.size() >= 2) {
            struct Call
            {
                avifRGBImage * rgb;
                uint32_t pixelRowBytes;
                Launch * launch;
            } call = { rgb, pixelRowBytes, &launch };
            return farmRun(shares, [](void * arg, uint32_t, FarmShare share) -> avifResult {
                const Call & c = *static_cast<const Call *>(arg);
                return inPlaceBandedRows(c.rgb, c.pixelRowBytes, *c.launch, share.begin, share.end);
            }, &call);
        }
    }
    tls.farmReports.clear();
    return inPlaceBandedRows(rgb, pixelRowBytes, launch, 0, rgb->height);
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
        tls.lastKernel = "to_f16_generic"; // (a farm worker's own context: the caller's is filled in from it)
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


