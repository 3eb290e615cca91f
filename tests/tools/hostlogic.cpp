// hostlogic.cpp -- test tool: C wrappers over the PRODUCT's host-only logic (libavif_amd/csrc/scale_plan.cpp, gainmap_plan.cpp:
// no HIP, no GPU), compiled with g++ into tests/tools/libhostlogic.so by tests/test_host_plans.py and compared with the oracle on
// the CPU.  What the kernels do with these tables is the GPU tests' business; that the tables are right is checked here.
#include <string.h>

#include "gainmap_plan.h"
#include "scale_plan.h"

using namespace avifhip;

extern "C" {

int hostScaleSchedule(int srcW, int srcH, int dstW, int dstH, int wide, int * colA, int * colB, int * rowA, int * rowB, int * rowF)
{
    const ScaleSchedule S = makeScaleSchedule(srcW, srcH, dstW, dstH, wide != 0);
    memcpy(colA, S.colA.data(), S.colA.size() * sizeof(int)), memcpy(colB, S.colB.data(), S.colB.size() * sizeof(int));
    memcpy(rowA, S.rowA.data(), S.rowA.size() * sizeof(int)), memcpy(rowB, S.rowB.data(), S.rowB.size() * sizeof(int));
    memcpy(rowF, S.rowF.data(), S.rowF.size() * sizeof(int));
    return S.mode;
}

float hostTransferFunction(int tc, int direction, float v)
{
    return direction ? gainMapToGamma(tc, v) : gainMapToLinear(tc, v);
}

int hostPrimariesMatrix(int src, int dst, double coeffs[9])
{
    return gainMapPrimariesMatrix(src, dst, coeffs) ? 1 : 0;
}

int hostDoubleToSignedFraction(double v, int32_t * n, uint32_t * d)
{
    return gainMapDoubleToFraction(v, n, d) ? 1 : 0;
}
int hostDoubleToUnsignedFraction(double v, uint32_t * n, uint32_t * d)
{
    return gainMapDoubleToUnsignedFraction(v, n, d) ? 1 : 0;
}

// output steps of a transfer function: copies up to `capacity` floats, returns the entries per piece; *maxCode receives the last code
uint32_t hostOutputSteps(int tc, uint32_t depth, int isFloat, float * steps, uint32_t capacity, uint32_t * maxCode)
{
    const GainMapSteps & S = gainMapOutputSteps(tc, depth, isFloat != 0);
    const size_t n = S.steps.size() < capacity ? S.steps.size() : capacity;
    memcpy(steps, S.steps.data(), n * sizeof(float));
    *maxCode = S.maxCode;
    return S.pieceEntries;
}

int hostChooseMathPrimaries(int basePrimaries, int altPrimaries)
{
    int out = -1;
    return gainMapChooseMathPrimaries(basePrimaries, altPrimaries, &out) ? out : -1;
}

} // extern "C"
