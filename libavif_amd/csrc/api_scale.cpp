// api_scale.cpp -- C ABI entry points for plane scaling (include/avifhip.h): avifhipImageScale[Async], reference src/scale.c.
// Schedules come from scale_plan.cpp, kernels from kernels_scale.hip.
#include "api_internal.h"

using namespace avifhip;
using namespace avifhip::api;

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
            if (S.mode == SCALE_BOX && !wide && cols == 256 && n > 0) { // uniform boxes of 4 / 8 columns: the dword path
                const int w = S.colB[0];
                bool uniform = (w == 4 || w == 8) && (n % 4 == 0); // (a box filter is only chosen beyond 2x, scale_plan.cpp)
                for (size_t i = 0; i < n && uniform; ++i)
                    uniform = S.colB[i] == w && S.colA[i] == S.colA[0] + (int)i * w;
                // (the lane's 4 boxes must lie inside the row: guaranteed since the reference reads them; n % 4: no clamped tail lanes)
                st.boxWidth = uniform ? w : 0;
            }
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
    if (dst->width > 32768u || dst->height > 32768u || (uint64_t)dst->width * dst->height > (uint64_t)16384 * 16384)
        return AVIF_RESULT_NOT_IMPLEMENTED; // avifDimensionsTooLarge with the default limits, src/scale.c:39-42
    const avifResult cr = ensureContext();
    if (cr != AVIF_RESULT_OK)
        return cr;
    hipStream_t stream = pickStream(hipStream);
    ScratchScope scratch(stream); // the schedule table may still be read by a scale enqueued on another stream
    if (scratch.result != AVIF_RESULT_OK)
        return scratch.result;
    const bool wide = src->depth > 8;
    const PlaneDims sd = planeDims(src->width, src->height, (int)src->yuvFormat), dd = planeDims(dst->width, dst->height, (int)dst->yuvFormat);
    // The schedules of every plane live in one per-thread device table (successive calls of one thread are ordered by the
    // stream they share, or through ScratchScope when the stream changes).  Building and uploading them is O(width + height) host work plus one small
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
            cache.doubling[p] = sched.doubling && !wide;
            cache.exactBox[p] = wide ? 0 : sched.exactBox;
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
    ScaleStagedLaunch L, W, D, B; // planes served by the row-staged kernel / the window kernel / the doubling kernel / the exact-box kernel
    L.count = W.count = D.count = B.count = 0;
    bool firstDoubled = false, firstBoxed = false;
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
        if (staged && cache.doubling[p] && scaleDoublingCovers(A)) {
            D.plane[D.count] = A, D.staging[D.count] = ScaleStaging();
            ++D.count;
            firstDoubled = firstDoubled || p == (present[0] ? 0 : 3);
            continue;
        }
        if (staged && cache.exactBox[p] && scaleExactBoxCovers(A)) {
            B.plane[B.count] = A, B.staging[B.count] = ScaleStaging();
            B.staging[B.count].boxWidth = cache.exactBox[p];
            ++B.count;
            firstBoxed = firstBoxed || p == (present[0] ? 0 : 3);
            continue;
        }
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
    hipError_t le = launchScalePlanesDoubling(D, stream);
    if (le == hipSuccess)
        le = launchScalePlanesExactBox(B, stream);
    if (le == hipSuccess)
        le = launchScalePlanesStaged(W, wide, true, stream);
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
    tls.lastKernel = firstDoubled ? "scale_up2[doubling]" : firstBoxed ? "scale_box[exact]" : names[family][cache.mode[first]];
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
    // avifDimensionsTooLarge(dstWidth, dstHeight, AVIF_DEFAULT_IMAGE_SIZE_LIMIT, AVIF_DEFAULT_IMAGE_DIMENSION_LIMIT), src/scale.c:39-42
    // (16384 * 16384 pixels, 32768 per dimension: src/avif.c avifDimensionsTooLarge, include/avif/avif.h:88-95)
    if (dstWidth > 32768u || dstHeight > 32768u || (uint64_t)dstWidth * dstHeight > (uint64_t)16384 * 16384)
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
    QuiesceOnExit quiesceOnExit; // (the scaler may still refuse the job after the planes' uploads were enqueued)
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
        const hipError_t ce = hipMemcpy2DAsync(fresh[p], dd.w[p] * bps, base + dstOff[p], dstPitch[p], dd.w[p] * bps, dd.h[p], hipMemcpyDeviceToHost, tls.stream);
        if (ce != hipSuccess) {
            (void)hipStreamSynchronize(tls.stream);
            for (int q = 0; q <= p; ++q)
                free(fresh[q]);
            return hipFailed(ce, "hipMemcpy2DAsync(scaled plane)");
        }
    }
    {
        const hipError_t se = hipStreamSynchronize(tls.stream);
        if (se != hipSuccess) {
            for (int q = 0; q < 4; ++q)
                free(fresh[q]);
            return hipFailed(se, "hipStreamSynchronize(scaled planes)");
        }
    }
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

