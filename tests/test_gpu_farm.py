"""The tile farm's product converter on the GPU: rectangles of a stitched canvas converted by batched launches
(avifhipImageYUVToRGBBatchAsync) must reproduce the whole-canvas conversion of the oracle byte for byte, seams included;
and the single-rectangle entry point (avifhipImageYUVToRGBRectAsync) must agree with the oracle's rectangle function."""
import ctypes as C

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi, device, farm, native

pytestmark = pytest.mark.gpu

CASES = [
    # cfg5 in miniature: 10-bit 4:2:0 limited BT.709 -> RGBA(10) bilinear, 3 x 3 grid with cropped last column/row
    (H.Y2RCase(1100, 150, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=10, upsampling=4), (512, 64)),
    (H.Y2RCase(1100, 150, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4), (512, 64)),
    (H.Y2RCase(777, 66, yuv_depth=8, yuv_format=2, yuv_range=1, matrix=6, rgb_depth=8, rgb_format=0, upsampling=4), (256, 32)),
    (H.Y2RCase(640, 48, yuv_depth=12, yuv_format=1, yuv_range=0, matrix=9, rgb_depth=16, alpha=True, rgb_premultiplied=True), (320, 16)),
    (H.Y2RCase(300, 40, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, rgb_format=9, upsampling=3), (64, 8)),  # generic path
]


@pytest.mark.parametrize("world", [1, 2, 8])
def test_grid_farm_equals_whole_canvas(hip, world):
    conv = farm.HipRectConverter()
    for case, (tw, th) in CASES:
        res, whole = H.run_y2r(H.oracle_backend(), case)
        assert res == 0
        canvas = H.make_y2r_inputs(case)
        rects = farm.grid_rects(case.w, case.h, tw, th)
        px = abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
        union = H.make_y2r_output(case)
        for rank in range(world):  # the ranks of a node, one after the other on this one GPU
            out = H.make_y2r_output(case)
            mine = farm.convert_shard(canvas, out, rects, rank, world, conv)
            for t in mine:
                x, y, w, h = rects[t]
                union.pixels[y:y + h, x * px:(x + w) * px] = out.pixels[y:y + h, x * px:(x + w) * px]
        wb = case.w * px
        assert np.array_equal(union.pixels[:, :wb], whole[:, :wb]), (case.ident(), native.last_kernel(), H.describe_diff(whole[:, :wb], union.pixels[:, :wb]))


def test_rect_entry_point_matches_oracle_rect(hip):
    o = oracle_lib.oracle()
    for case, _ in CASES:
        canvas = H.make_y2r_inputs(case)
        for rect in [(0, 0, case.w, case.h), (8, 2, 264, 20), (256, 16, case.w - 256, case.h - 16), (16, 4, 67, 7)]:
            x, y, w, h = rect
            want = H.make_y2r_output(case)
            r = abi.avifCropRect(x, y, w, h)
            assert o.oracleImageYUVToRGBRect(canvas.struct, want.struct, C.byref(r)) == 0
            got = H.make_y2r_output(case)
            dimg, drgb = device.DeviceYUV(canvas), device.DeviceRGB(got, upload=True)
            native.check(hip.avifhipImageYUVToRGBRectAsync(dimg.struct, drgb.struct, C.byref(r), None), "rect")
            native.check(hip.avifhipSynchronize(None), "sync")
            drgb.download_into_host()
            wb = case.w * abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
            assert np.array_equal(got.pixels[:, :wb], want.pixels[:, :wb]), (case.ident(), rect, native.last_kernel(), H.describe_diff(want.pixels[:, :wb], got.pixels[:, :wb]))
