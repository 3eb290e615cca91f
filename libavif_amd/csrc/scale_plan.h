// scale_plan.h -- host-side description of one plane scale (avifImageScale, reference src/scale.c:23-201, which runs the
// vendored libyuv scaler under kFilterBox: third_party/libyuv/source/scale.c:829-1007).
//
// The reference scales a plane row by row through temporary row buffers; here a scale is a MODE plus one schedule entry
// per destination column and per destination row, derived on the host (O(width + height) integer work that reproduces
// the reference's 16.16 stepping, clamps and row-buffer bookkeeping), after which every destination sample is an
// independent function of at most 2 x 2 source samples (or one box): the kernel is one lane per sample.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace avifhip {

enum ScaleMode : int {
    SCALE_POINT = 0, // CopyPlane, ScalePlaneSimple: src[rowA][colA]
    SCALE_DOWN = 1,  // ScalePlaneBilinearDown, ScalePlaneVertical: rows blended first (rounded), then columns
    SCALE_UP = 2,    // ScalePlaneBilinearUp: columns blended first (rounded), then rows
    SCALE_BOX = 3,   // ScalePlaneBox: box average with the reference's fixed-point reciprocal
    SCALE_UP2 = 4    // ScalePlaneUp2_Linear / _Bilinear: 9:3:3:1 with duplicated neighbours at the edges
};

struct ScaleSchedule
{
    int mode = SCALE_POINT;
    int exactBox = 0;      // BOX with every box exactly N x N on the N-grid, N in {4, 8} (exact 4x / 8x thumbnails): what the box kernel serves
    bool doubling = false; // UP2 on both axes (ScalePlaneUp2_Bilinear and twins): what the doubling kernel serves without the tables
    // per destination column.  POINT/DOWN/UP: source column, 16.16 fraction; UP2: near, far column; BOX: first column, width
    std::vector<int32_t> colA, colB;
    // per destination row.  DOWN/UP/UP2: first and second source row, 8-bit fraction; BOX: first row, height
    std::vector<int32_t> rowA, rowB, rowF;
};

// (srcW x srcH) -> (dstW x dstH) for 8-bit (wide = false: ScalePlane) or 16-bit samples (wide = true: ScalePlane_12)
ScaleSchedule makeScaleSchedule(int srcW, int srcH, int dstW, int dstH, bool wide);

} // namespace avifhip
