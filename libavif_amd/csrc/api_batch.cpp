// api_batch.cpp -- YUV -> RGB of many jobs in one launch: batches of frames, grids of tiles converted where they lie, and the fused
// decode-side tail (conversion + crop / rotate / mirror); the descriptor tables' ring and the table the device still holds.
#include "api_internal.h"

#include <algorithm>

using namespace avifhip;
using namespace avifhip::api;

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
                                 const void ** extraDevice = nullptr, uint32_t * slotOut = nullptr, bool * residentOut = nullptr,
                                 const TileNeighbours * neighbours = nullptr, bool * seamsDone = nullptr, int linkForced = -1, uint32_t canvasColumns = 0)
{
    if (seamsDone)
        *seamsDone = false;
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
            AVIFHIP_HOST_MEMORY_FREED(tls.pinnedTable);
            HIP_TRY(hipHostFree(tls.pinnedTable));
            tls.pinnedTable = nullptr;
            tls.pinnedTableCapacity = 0;
            tls.residentSlot = -1;
            HIP_TRY(hipDeviceSynchronize()); // the slots' device slices move as well: no batch may still be reading the old ones
        }
        const size_t slotBytes = (bytes + 4095) & ~(size_t)4095;
        HIP_TRY(hipHostMalloc(&tls.pinnedTable, slotBytes * kRing, hipHostMallocDefault));
        AVIFHIP_NEW_HOST_MEMORY(tls.pinnedTable, slotBytes * kRing);
        tls.pinnedTableCapacity = slotBytes;
    }
    uint32_t slot = tls.tableSlot++ % (uint32_t)kRing;
    if (!tls.tableSlotIdle[slot]) {
        HIP_TRY(hipEventSynchronize(tls.tableCopied[slot])); // the upload of kRing batches ago has left this slot's pinned memory
        tls.tableSlotIdle[slot] = true;
    }
    if ((int)slot == tls.residentSlot)
        tls.residentSlot = -1; // (its pinned copy changes now)
    uint8_t * pinned = (uint8_t *)tls.pinnedTable + (size_t)slot * tls.pinnedTableCapacity;
    // A small table is compared with the previous upload before it travels: a caller converting the same buffers again (a decoder's tile
    // buffers, frame after frame) finds its descriptors still on the device and skips the upload and its four stream / event calls --
    // half of what a grid of small tiles costs on the host (photo_grid).  Every byte of the slot is defined for that.
    const bool comparable = bytes <= Context::kResidentTableMax;
    if (comparable)
        memset(pinned, 0, bytes);
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
    // A plain batch whose jobs are the tiles of ONE canvas in row-major order (avifhipImageYUVToRGBBatchAsync with one RGB canvas and its tile
    // rectangles: what a rank of the tile farm converts) may walk along the canvas rows like a grid call does (launchYuvToRgbTileBatch):
    // `canvasColumns` = the number of leading jobs that share the first job's canvas rows, if the rest repeats that pattern.
    if (!canvasColumns && !map && rects && count > 1) {
        uint32_t cols = 1;
        while (cols < count && rgbs[cols]->pixels == rgbs[0]->pixels && rgbs[cols]->rowBytes == rgbs[0]->rowBytes && rects[cols].y == rects[0].y &&
               rects[cols].x == rects[cols - 1].x + rects[cols - 1].width)
            ++cols;
        bool regular = cols > 1 && count % cols == 0;
        for (uint32_t k = cols; regular && k < count; ++k)
            regular = rgbs[k]->pixels == rgbs[0]->pixels && rgbs[k]->rowBytes == rgbs[0]->rowBytes && rects[k].x == rects[k % cols].x &&
                      rects[k].y == rects[k - cols].y + rects[k - cols].height && (k % cols == 0 || rects[k].y == rects[k - 1].y);
        if (regular)
            canvasColumns = cols;
    }
    const avifResult rr = reserve(tls.table, tls.pinnedTableCapacity * kRing); // (growing it waits for the device: nothing reads the old one then)
    if (rr != AVIF_RESULT_OK)
        return rr;
    hipStream_t stream = pickStream(hipStream);
    YuvToRgbPlan representative;
    uint32_t restW = 0, restH = 0, restMaxH = 0, restMaxW = 0;
    if (allTiled) {
        representative = plansA[0];
        fillTileBatchTable(plansA, count, pinned);
        // leftovers that do not fill a 4x2 pixel group: right strips (in place of the whole jobs) and bottom rows
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
    }
    // Tiles of one canvas whose every pixel goes through the tiled kernels: the jobs are linked to their neighbours and the seam-aware build
    // of the family filters chroma across the seams in the same launch (tile_impl.h TILE_SEAMS).  With leftover columns / rows -- converted
    // by the universal kernel from each job's own window -- the caller's seam pass still runs.
    const bool linked = neighbours && allTiled && !restW && !restH && tileBatchLinksNeighbours(representative, count, maxW, maxH, linkForced, canvasColumns);
    if (linked) {
        for (uint32_t k = 0; k < count; ++k)
            linkTileBatchHalo(pinned, k, neighbours[k]);
        if (seamsDone)
            *seamsDone = true;
    }
    // (same stream: that stream waited for the resident slot's upload when it ran the batch that brought it)
    // (... which holds only for a stream the library owns: a caller's handle may be a new stream at a recycled address)
    const uint64_t streamGeneration = ownedStreamGeneration(stream);
    const bool resident = comparable && streamGeneration && tls.residentSlot >= 0 && tls.residentBytes == bytes && tls.residentStream == stream &&
                          tls.residentGeneration == streamGeneration && tls.residentAllTiled == allTiled &&
                          memcmp((const uint8_t *)tls.pinnedTable + (size_t)tls.residentSlot * tls.pinnedTableCapacity, pinned, bytes) == 0;
    if (resident) {
        --tls.tableSlot; // the slot just filled was not used
        slot = (uint32_t)tls.residentSlot;
        if (slotOut)
            *slotOut = slot;
    }
    uint8_t * dev = (uint8_t *)tls.table.ptr + (size_t)slot * tls.pinnedTableCapacity;
    if (extraDevice)
        *extraDevice = dev + extraOffset;
    // The table crosses the link on `upStream` while earlier batches compute on `stream`: the upload waits only for the kernels that read
    // this slot's device slice kRing batches ago (on whichever stream they ran), the batch's kernels wait for the upload.
    if (residentOut)
        *residentOut = resident;
    if (!resident) {
        if (tls.tableUnmarked[slot]) { // its last readers ran off the resident copy and left no event (below): marked now
            if (ownedStreamGeneration(tls.tableLastStream[slot]) != tls.tableLastGeneration[slot] ||
                hipEventRecord(tls.tableConsumed[slot], tls.tableLastStream[slot]) != hipSuccess) {
                (void)hipGetLastError();
                HIP_TRY(hipDeviceSynchronize()); // (a stream destroyed since)
            }
            tls.tableUnmarked[slot] = false;
        }
        HIP_TRY(hipStreamWaitEvent(tls.upStream, tls.tableConsumed[slot], 0));
    }
    // A batch that launches on the resident table records no event behind its kernels: the next kernel would wait for the event's signal
    // (~5 us per call of a small grid), and the event is only needed once an upload wants the slot again -- it is recorded then, on the
    // stream noted here, and covers everything that stream was given before.
    struct MarkConsumed
    {
        hipEvent_t ev;
        hipStream_t s;
        bool armed, lazy;
        uint32_t slot;
        uint64_t generation;
        ~MarkConsumed()
        {
            if (!armed)
                return;
            if (lazy)
                tls.tableUnmarked[slot] = true, tls.tableLastStream[slot] = s, tls.tableLastGeneration[slot] = generation;
            else
                (void)hipEventRecord(ev, s);
        }
    } markConsumed = { tls.tableConsumed[slot], stream, true, resident, slot, streamGeneration }; // (resident implies an owned stream)
    auto upload = [&](void * to, const void * from, size_t n) -> hipError_t {
        if (resident)
            return hipSuccess;
        tls.tableSlotIdle[slot] = false;
        tls.residentSlot = -1;
        ++tls.tableUploads;
        hipError_t ue = hipMemcpyAsync(to, from, n, hipMemcpyHostToDevice, tls.upStream);
        if (ue == hipSuccess)
            ue = hipEventRecord(tls.tableCopied[slot], tls.upStream);
        if (ue == hipSuccess)
            ue = hipStreamWaitEvent(stream, tls.tableCopied[slot], 0);
        if (ue == hipSuccess && comparable)
            tls.residentSlot = (int)slot, tls.residentBytes = bytes, tls.residentStream = stream, tls.residentGeneration = streamGeneration,
            tls.residentAllTiled = allTiled;
        return ue;
    };
    hipError_t e = hipSuccess;
    if (allTiled) {
        HIP_TRY(upload(dev, pinned, bytes));
        e = launchYuvToRgbTileBatch(dev, representative, count, maxW, maxH, stream, &tls.lastKernel, linked, canvasColumns);
        if (e == hipSuccess && restW)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes), count, restW, restMaxH, stream);
        if (e == hipSuccess && restH)
            e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes) + count, count, restMaxW, restH, stream);
    } else {
        HIP_TRY(upload(dev + tileBytes, plansA, extra ? bytes - tileBytes : planBytes));
        tls.lastKernel = "yuv2rgb_generic_batch";
        e = launchYuvToRgbGenericBatch((const YuvToRgbPlan *)(dev + tileBytes), count, maxW, maxH, stream);
    }
    if (e != hipSuccess)
        return hipFailed(e, "YUV->RGB batch kernel launch");
    ++tls.launches;
    if (slotOut)
        markConsumed.armed = false; // the caller's kernels read the slot too: it records the event after them
    return AVIF_RESULT_OK;
}

// A batch of large frames that differ in their buffers only -- an image sequence -- runs through the single-image kernels, up to 8 frames per
// launch, the frames' addresses in the kernel arguments (kernels.h launchYuvToRgbTileSequence): nothing is uploaded and no event sits between
// consecutive launches, which costs a table batch of two 8K frames more than the second frame's ramp and tail save it.  *taken = false:
// not a sequence (or a kernel family without sequence kernels) and nothing was launched.  AVIFHIP_SEQUENCE=0 keeps the table batches.
static avifResult sequenceAsync(uint32_t count, const avifImage * const * images, avifRGBImage * const * rgbs, const avifCropRect * rects, void * hipStream, bool * taken)
{
    *taken = false;
    static const bool off = [] {
        const char * e = getenv("AVIFHIP_SEQUENCE");
        return e && e[0] == '0' && !e[1];
    }();
    if (off || count == 0 || !images || !rgbs || !gTiledKernels.load(std::memory_order_relaxed))
        return AVIF_RESULT_OK;
    for (uint32_t k = 0; k < count; ++k)
        if (!images[k] || !rgbs[k] || (uint64_t)images[k]->width * images[k]->height < ((uint64_t)2 << 20))
            return AVIF_RESULT_OK; // (cheap rejection of grids of small tiles before anything is planned)
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    const int arithmetic = effectiveArithmetic();
    const uint32_t tuning = gTuning.load(std::memory_order_relaxed);
    std::vector<YuvToRgbPlan> plans(count);
    for (uint32_t k = 0; k < count; ++k) {
        avifResult pr = AVIF_RESULT_OK;
        if (k == 0 || !rebindYuvToRgbPlan(plans[0], images[0], rgbs[0], images[k], rgbs[k], rects ? &rects[k] : nullptr, &plans[k], &pr))
            pr = makeYuvToRgbPlan(images[k], rgbs[k], rects ? &rects[k] : nullptr, arithmetic, tuning, &plans[k]);
        if (pr != AVIF_RESULT_OK)
            return AVIF_RESULT_OK; // (the table path reports it, with its own message)
        if (!tileSequenceCompatible(plans[0], plans[k]))
            return AVIF_RESULT_OK;
    }
    hipStream_t stream = pickStream(hipStream);
    for (uint32_t first = 0; first < count; first += kTileSequenceMax) {
        const uint32_t n = count - first < kTileSequenceMax ? count - first : kTileSequenceMax;
        const hipError_t e = launchYuvToRgbTileSequence(plans.data() + first, n, stream, &tls.lastKernel);
        if (e == hipErrorNotSupported && first == 0) {
            (void)hipGetLastError();
            return AVIF_RESULT_OK;
        }
        if (e != hipSuccess)
            return hipFailed(e, "YUV->RGB sequence kernel launch");
        ++tls.launches;
    }
    *taken = true;
    return AVIF_RESULT_OK;
}

extern "C" avifResult avifhipImageYUVToRGBBatchAsync(uint32_t count,
                                                     const avifImage * const * images,
                                                     avifRGBImage * const * rgbs,
                                                     const avifCropRect * rects,
                                                     void * hipStream)
{
    bool taken = false;
    const avifResult sr = sequenceAsync(count, images, rgbs, rects, hipStream, &taken);
    if (sr != AVIF_RESULT_OK || taken)
        return sr;
    return batchAsyncImpl(count, images, rgbs, rects, nullptr, hipStream);
}

// Grid canvases: tiles converted where they lie (a batch of rectangle jobs over "virtual canvases" whose plane pointers are
// shifted so that canvas coordinates address the tile's own memory, each confined to its own chroma samples), then the
// pixels next to interior seams redone with samples fetched from both sides (kernels_generic.hip: GridReader).
// `only`: the part of the canvas somebody will look at (the fused tail's crop, grown to the tile kernels' origin rule by coverOfCrop);
// tiles are converted inside it, tiles outside it not at all
static avifResult gridYuvToRgbImpl(const avifhipGrid * grid, const avifImage * const * colorTiles, const avifImage * const * alphaTiles, avifBool alphaIsLimitedRange,
                                   avifRGBImage * rgbCanvas, void * hipStream, const PixelMap * map, const avifCropRect * only = nullptr)
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
    // one launch for tiles AND seams (kernels.h TileNeighbours) needs one chroma pitch per plane over the whole grid: a neighbour's sample is
    // addressed with the job's own offset.  Whether it is used: tileBatchLinksNeighbours; AVIFHIP_GRID_SEAM_PASS=1 always keeps the second
    // pass, =0 never does where the grid can be linked (A/B measurements, tests of the seam kernels)
    // (exactly "1" or "0": anything else -- "yes", an empty string -- is ignored rather than read as 0)
    const char * seamPassEnv = getenv("AVIFHIP_GRID_SEAM_PASS");
    const int linkForced = (seamPassEnv && (seamPassEnv[0] == '0' || seamPassEnv[0] == '1') && !seamPassEnv[1]) ? (seamPassEnv[0] == '1' ? 0 : 1) : -1;
    bool linkable = subsampled && count > 1 && linkForced != 0;
    for (uint32_t t = 0; t < count; ++t) {
        const avifImage * tile = colorTiles[t];
        if (!tile || !tile->yuvPlanes[0])
            return AVIF_RESULT_INVALID_ARGUMENT;
        // "All tiles in a grid image should match the first tile", src/read.c:1832-1842
        if (tile->width != tw || tile->height != th || tile->depth != first->depth || tile->yuvFormat != first->yuvFormat ||
            tile->yuvRange != first->yuvRange || tile->colorPrimaries != first->colorPrimaries ||
            tile->transferCharacteristics != first->transferCharacteristics || tile->matrixCoefficients != first->matrixCoefficients)
            return AVIF_RESULT_INVALID_IMAGE_GRID;
        if (tile->yuvRowBytes[1] != first->yuvRowBytes[1] || tile->yuvRowBytes[2] != first->yuvRowBytes[2] || !tile->yuvPlanes[1] || !tile->yuvPlanes[2])
            linkable = false;
        const avifImage * atile = alphaTiles ? alphaTiles[t] : nullptr;
        if (alphaTiles && (!atile || !atile->alphaPlane || atile->width != tw || atile->height != th || atile->depth != first->depth))
            return AVIF_RESULT_INVALID_IMAGE_GRID;
        const uint32_t col = t % grid->columns, row = t / grid->columns;
        const uint32_t X0 = col * tw, Y0 = row * th;
        avifCropRect & r = rects[t];
        r.x = X0, r.y = Y0;
        r.width = (X0 + tw > grid->outputWidth) ? grid->outputWidth - X0 : tw;   // src/read.c:1863-1868
        r.height = (Y0 + th > grid->outputHeight) ? grid->outputHeight - Y0 : th;
        const uint32_t seenW = r.width, seenH = r.height; // the tile's part of the canvas: what the chroma window below spans
        if (only) {
            const uint32_t xa = std::max(r.x, only->x), xb = std::min(r.x + r.width, only->x + only->width);
            const uint32_t ya = std::max(r.y, only->y), yb = std::min(r.y + r.height, only->y + only->height);
            r.x = xa, r.y = ya, r.width = xb > xa ? xb - xa : 0, r.height = yb > ya ? yb - ya : 0; // (empty: dropped below)
        }
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
        o.window[0] = (int32_t)(X0 >> sx), o.window[1] = (int32_t)((X0 >> sx) + ((seenW + sx) >> sx) - 1);
        o.window[2] = (int32_t)(Y0 >> sy), o.window[3] = (int32_t)((Y0 >> sy) + ((seenH + sy) >> sy) - 1);
        o.alphaLimited = atile && alphaIsLimitedRange;
    }
    // every tile's neighbours in the grid, by the virtual plane pointers made above (a tile the crop drops from the batch still lends its samples)
    std::vector<TileNeighbours> neighbours(linkable ? count : 0);
    for (uint32_t t = 0; linkable && t < count; ++t) {
        const int col = (int)(t % grid->columns), row = (int)(t / grid->columns);
        TileNeighbours & n = neighbours[t];
        n.above = row > 0, n.below = row + 1 < (int)grid->rows, n.left = col > 0, n.right = col + 1 < (int)grid->columns;
        for (int d = 0; d < 9; ++d) {
            const int dr = (d / 3 == 1) ? -1 : (d / 3 == 2 ? 1 : 0), dc = (d % 3 == 1) ? -1 : (d % 3 == 2 ? 1 : 0);
            const int r2 = row + dr, c2 = col + dc;
            const bool there = r2 >= 0 && r2 < (int)grid->rows && c2 >= 0 && c2 < (int)grid->columns;
            const avifImage & nv = views[there ? (uint32_t)r2 * grid->columns + (uint32_t)c2 : t];
            n.plane1[d] = nv.yuvPlanes[1], n.plane2[d] = nv.yuvPlanes[2];
        }
    }
    // the seam kernel's tile table rides in the batch's descriptor upload (its own copy on the compute stream cost 5 us plus two gaps)
    const void * deviceTiles = nullptr;
    uint32_t tableSlot = 0;
    uint32_t jobs = count;
    if (only) { // (the tile table stays whole: the seam kernel finds tiles by their grid position)
        jobs = 0;
        for (uint32_t t = 0; t < count; ++t) {
            if (!rects[t].width || !rects[t].height)
                continue;
            viewPtrs[jobs] = viewPtrs[t], rects[jobs] = rects[t], overrides[jobs] = overrides[t];
            if (linkable)
                neighbours[jobs] = neighbours[t];
            ++jobs;
        }
    }
    bool residentTable = false, seamsDone = false;
    avifResult r = batchAsyncImpl(jobs, viewPtrs.data(), rgbPtrs.data(), rects.data(), overrides.data(), hipStream, map, tiles.data(), tiles.size() * sizeof(GridTile),
                                  &deviceTiles, &tableSlot, &residentTable, linkable ? neighbours.data() : nullptr, &seamsDone, linkForced,
                                  // (every tile a job, row-major, into one canvas: the kernels may walk along the canvas rows)
                                  (!only && !map && jobs == count) ? grid->columns : 0u);
    if (r != AVIF_RESULT_OK)
        return r;
    hipStream_t stream = pickStream(hipStream);
    struct SlotRead // the table's slot is free again after the last kernel that reads it: the seam kernel if there is one, the batch otherwise
    {               // (a launch on the resident table: noted, not recorded -- batchAsyncImpl)
        hipEvent_t ev;
        hipStream_t s;
        bool lazy;
        uint32_t slot;
        ~SlotRead()
        {
            if (lazy) // (a resident table implies a stream the library owns: batchAsyncImpl)
                tls.tableUnmarked[slot] = true, tls.tableLastStream[slot] = s, tls.tableLastGeneration[slot] = ownedStreamGeneration(s);
            else
                (void)hipEventRecord(ev, s);
        }
    } slotRead = { tls.tableConsumed[tableSlot], stream, residentTable, tableSlot };
    static const bool traceGrids = getenv("AVIFHIP_GRID_TRACE") != nullptr; // (read once: a debugging aid, not a switch to flip while running)
    if (traceGrids)
        fprintf(stderr, "avifhip grid %ux%u: %s, seams %s (%s)\n", grid->columns, grid->rows, linkable ? "linkable" : "not linkable",
                seamsDone ? "in the tile kernels" : "in a second pass", tls.lastKernel ? tls.lastKernel : "?");
    if (count == 1 || seamsDone)
        return AVIF_RESULT_OK; // (one tile, or the tiled kernels read across the seams themselves)
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
    GridGeometry g;
    memset(&g, 0, sizeof(g));
    g.columns = grid->columns, g.rows = grid->rows, g.tileW = tw, g.tileH = th, g.tileCW = tw >> sx, g.tileCH = th >> sy;
    const hipError_t e = launchYuvToRgbGridSeams(canvasPlan, g, (const GridTile *)deviceTiles, grid->columns > 1, sy && grid->rows > 1, stream);
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
        const avifCropRect cover = coverOfCrop(r, map, (uintptr_t)rgb->pixels, px);
        return gridYuvToRgbImpl(grid, colorTiles, alphaTiles, alphaIsLimitedRange, &canvasRgb, hipStream, &map, &cover);
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
    const PixelMap map = makePixelMap(r.x, r.y, r.width, r.height, turns, mirrorAxis);
    const avifCropRect cover = coverOfCrop(r, map, (uintptr_t)rgb->pixels, px);
    const avifResult pr = makeYuvToRgbPlan(image, &canvasRgb, &cover, effectiveArithmetic(), gTuning.load(std::memory_order_relaxed), &plan);
    if (pr != AVIF_RESULT_OK)
        return pr;
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    plan.rgb.map = map;
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

