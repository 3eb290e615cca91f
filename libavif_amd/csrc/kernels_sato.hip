// kernels_sato.hip -- Sample Transform derived image items: a postfix expression over the samples of up to 32 input image
// items, evaluated per sample in saturating 32-bit arithmetic (reference src/sampletransform.c:199-277 operators, :284-352
// evaluation loop).  One lane per sample; the expression (at most 64 tokens) travels in the kernel arguments, the operand
// stack (at most 33 entries) lives in registers / scratch.
#include <hip/hip_runtime.h>

#include "avifhip/avif_abi.h"
#include "kernels.h"

namespace avifhip {

namespace {

__device__ __forceinline__ int32_t clamp32(int64_t v) // avifSampleTransformClamp32b, :194-197
{
    return v <= INT32_MIN ? INT32_MIN : (v >= INT32_MAX ? INT32_MAX : (int32_t)v);
}

__device__ __forceinline__ int32_t unaryOp(int32_t a, int type) // :199-224
{
    switch (type) {
        case AVIF_SAMPLE_TRANSFORM_NEGATION: return clamp32(-(int64_t)a);
        case AVIF_SAMPLE_TRANSFORM_ABSOLUTE: return a >= 0 ? a : clamp32(-(int64_t)a);
        case AVIF_SAMPLE_TRANSFORM_NOT: return ~a;
        default: // BSR: a <= 0 ? 0 : floor(log2(a))
            return a <= 0 ? 0 : 31 - __clz(a);
    }
}

__device__ __forceinline__ int32_t binaryOp(int32_t l, int32_t r, int type) // :226-277
{
    switch (type) {
        case AVIF_SAMPLE_TRANSFORM_SUM: return clamp32((int64_t)l + r);
        case AVIF_SAMPLE_TRANSFORM_DIFFERENCE: return clamp32((int64_t)l - r);
        case AVIF_SAMPLE_TRANSFORM_PRODUCT: return clamp32((int64_t)l * r);
        case AVIF_SAMPLE_TRANSFORM_QUOTIENT:
            // clamp32((int64_t)l / r).  The only quotient of two 32-bit values that leaves the 32-bit range is INT32_MIN / -1;
            // it is spelled out because the device compiler narrows the sign-extended 64-bit division to a 32-bit one, for
            // which that case overflows (tests/tools/sato_probe: the narrowed division returned INT32_MIN).
            if (r == 0)
                return l;
            if (l == INT32_MIN && r == -1)
                return INT32_MAX;
            return l / r;
        case AVIF_SAMPLE_TRANSFORM_AND: return l & r;
        case AVIF_SAMPLE_TRANSFORM_OR: return l | r;
        case AVIF_SAMPLE_TRANSFORM_XOR: return l ^ r;
        case AVIF_SAMPLE_TRANSFORM_POW: {
            if (l == 0 || l == 1)
                return l;
            if (l == -1)
                return (r % 2 == 0) ? 1 : -1;
            if (r == 0)
                return 1;
            if (r == 1)
                return l;
            if (r < 0)
                return 0;
            int64_t result = l;
            for (int32_t i = 1; i < r; ++i) { // |l| >= 2: leaves the 32-bit range within 31 steps
                result *= l;
                if (result < INT32_MIN || result > INT32_MAX)
                    return (l > 0 || r % 2 == 0) ? INT32_MAX : INT32_MIN;
            }
            return (int32_t)result;
        }
        case AVIF_SAMPLE_TRANSFORM_MIN: return l <= r ? l : r;
        default: return l <= r ? r : l; // MAX
    }
}

__global__ __launch_bounds__(256) void satoKernel(SatoArgs A, const SatoInputs * __restrict__ in)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.width || y >= A.height)
        return;
    int32_t stack[kSatoMaxTokens / 2 + 1];
    int n = 0;
    for (int t = 0; t < A.numTokens; ++t) {
        const int type = A.tokens[t].type;
        if (type == AVIF_SAMPLE_TRANSFORM_CONSTANT) {
            stack[n++] = A.tokens[t].value;
        } else if (type == AVIF_SAMPLE_TRANSFORM_INPUT_IMAGE_ITEM_INDEX) {
            const int k = A.tokens[t].value;
            const uint8_t * row = in->plane[k] + (size_t)y * in->pitch[k];
            stack[n++] = in->wide[k] ? (int32_t)reinterpret_cast<const uint16_t *>(row)[x] : (int32_t)row[x];
        } else if (type < AVIF_SAMPLE_TRANSFORM_FIRST_BINARY_OPERATOR) {
            stack[n - 1] = unaryOp(stack[n - 1], type);
        } else {
            stack[n - 2] = binaryOp(stack[n - 2], stack[n - 1], type);
            --n;
        }
    }
    // "Fit to the range defined by the PixelInformationProperty", :340-342
    const int32_t v = min(max(stack[0], 0), A.maxValue);
    uint8_t * d = A.dst + (size_t)y * A.dstPitch;
    if (A.dstWide)
        reinterpret_cast<uint16_t *>(d)[x] = (uint16_t)v;
    else
        d[x] = (uint8_t)v;
}

} // namespace

hipError_t launchSato(const SatoArgs & args, const SatoInputs * deviceInputs, hipStream_t stream)
{
    if (args.width <= 0 || args.height <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(satoKernel, dim3((args.width + 63) / 64, (args.height + 3) / 4), dim3(64, 4), 0, stream, args, deviceInputs);
    return hipGetLastError();
}

} // namespace avifhip
