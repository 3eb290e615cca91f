// kernels.h -- launch entry points of the HIP kernels (kernels_generic.hip, kernels_tile.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace avifhip {

// universal one-lane-per-pixel kernels (every combination the reference accepts)
hipError_t launchYuvToRgbGeneric(const YuvToRgbPlan & plan, hipStream_t stream);
hipError_t launchYuvToRgbGenericBatch(const YuvToRgbPlan * deviceTable, uint32_t count, uint32_t maxW, uint32_t maxH, hipStream_t stream);
hipError_t launchRgbToYuvGeneric(const RgbToYuvPlan & plan, hipStream_t stream);
hipError_t launchAlphaMulGeneric(const AlphaMulPlan & plan, hipStream_t stream);

// bandwidth-tuned tiled kernels; return false from the *Supported predicates when a plan is not covered
bool tileYuvToRgbSupported(const YuvToRgbPlan & plan);
// id of the tiled-kernel instantiation serving `plan`, or -1 when only the generic kernel can
int tileYuvToRgbVariant(const YuvToRgbPlan & plan);
hipError_t launchYuvToRgbTile(const YuvToRgbPlan & plan, hipStream_t stream, const char ** kernelName);
hipError_t launchYuvToRgbTileBatch(const YuvToRgbPlan * deviceTable, const YuvToRgbPlan & representative, uint32_t count,
                                   uint32_t maxW, uint32_t maxH, hipStream_t stream, const char ** kernelName);
bool tileRgbToYuvSupported(const RgbToYuvPlan & plan);
hipError_t launchRgbToYuvTile(const RgbToYuvPlan & plan, hipStream_t stream, const char ** kernelName);

} // namespace avifhip
