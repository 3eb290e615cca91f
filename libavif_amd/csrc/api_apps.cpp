// api_apps.cpp -- what sits either side of the conversions: Sample Transform items, the applications' crop / rotate / mirror, and the
// row packing of the file writers (Y4M frames, PNG rows).
#include "api_internal.h"

#include <algorithm>

using namespace avifhip;
using namespace avifhip::api;

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
