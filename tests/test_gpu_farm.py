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


def test_farm_moves_only_its_share_over_the_host_link(hip):
    """avifhipImageYUVToRGBRects uploads the rectangles' plane windows (+ chroma halo) and downloads the rectangles: per rank
    ~1/N of the canvas, exactly what avifhipPlanRectTransfers announces."""
    conv = farm.HipRectConverter()
    case, (tw, th) = CASES[0]
    canvas = H.make_y2r_inputs(case)
    rects = farm.grid_rects(case.w, case.h, tw, th)
    px = abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
    world = 4
    ups, downs = [], []
    for rank in range(world):
        out = H.make_y2r_output(case)
        mine = farm.convert_shard(canvas, out, rects, rank, world, conv)
        assert (conv.bytes_up, conv.bytes_down) == farm.planned_transfers(canvas, out, [rects[t] for t in mine])
        ups.append(conv.bytes_up), downs.append(conv.bytes_down)
    assert sum(downs) == case.w * case.h * px
    canvas_in = case.w * case.h * 2 + 2 * ((case.w + 1) // 2) * ((case.h + 1) // 2) * 2
    assert canvas_in <= sum(ups) <= 1.1 * canvas_in
    assert max(ups) <= 0.45 * sum(ups)  # 9 tiles over 4 ranks: 3 + 2 + 2 + 2


def test_rects_entry_point_error_codes(hip):
    case, _ = CASES[0]
    canvas = H.make_y2r_inputs(case)
    out = H.make_y2r_output(case)
    bad = (abi.avifCropRect * 1)(abi.avifCropRect(1, 0, 64, 32))  # x off the chroma grid
    assert hip.avifhipImageYUVToRGBRects(canvas.struct, out.struct, bad, 1) != 0
    outside = (abi.avifCropRect * 1)(abi.avifCropRect(case.w - 8, 0, 64, 32))
    assert hip.avifhipImageYUVToRGBRects(canvas.struct, out.struct, outside, 1) != 0
    assert (out.pixels == H.FILL_BYTE).all()
    assert hip.avifhipImageYUVToRGBRects(canvas.struct, out.struct, None, 0) == 0


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
