"""The in-process device farm (include/avifhip.h: avifhipSetDeviceSet; libavif_amd/csrc/api_farm.cpp) on the one GPU a test box has: a device
set that names device 0 two or three times gives two or three worker threads with contexts of their own -- the multi-device code path with
real kernels and real transfers.  Every farmed call must produce the bytes of the oracle (and therefore of the single-device call), row
padding included; every worker must move about 1/N of the image plus the chroma halo row of its seams."""
import ctypes as C
import dataclasses
import os

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, farm, native, synth

pytestmark = pytest.mark.gpu

# heights of 64 rows and more (shares are multiples of 32 rows): odd sizes, partial tiles right and below, leftovers of the 4 x 2 groups
FARMED = [(300, 100), (777, 131), (1027, 70), (64, 97), (512, 256)]


@pytest.fixture()
def farm3(hip):
    """Three workers on device 0 and shares from 64 x 32 pixels up, so that the sweeps' small images are farmed."""
    assert hip.avifhipSetDeviceSet((C.c_int * 3)(0, 0, 0), 3) == 0
    hip.avifhipSetFarmMinSharePixels(64 * 32)
    yield hip
    hip.avifhipSetFarmMinSharePixels(0)
    assert hip.avifhipSetDeviceSet(None, 0) == 0


@pytest.fixture()
def farm2(hip):
    """Two workers on device 0, default share size: images of 4 megapixels and more are farmed."""
    assert hip.avifhipSetDeviceSet((C.c_int * 2)(0, 0), 2) == 0
    yield hip
    assert hip.avifhipSetDeviceSet(None, 0) == 0


def _flush(hip):
    """AVIFHIP_TEST_FLUSH=1 (tests/tools/fault_hunt.sh): small farmed calls right after a large one, while the large one's buffers are still alive."""
    if os.environ.get("AVIFHIP_TEST_FLUSH"):
        hip.avifhipSetFarmMinSharePixels(64 * 32)
        for k in range(4):
            H.run_y2r(H.hip_host_backend(), H.Y2RCase(300, 128 + 32 * k, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, upsampling=4))
        hip.avifhipSetFarmMinSharePixels(0)


def reports(lib):
    out = []
    for k in range(lib.avifhipLastFarmWorkers()):
        dev, b, e, up, down = C.c_int(-1), C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        assert lib.avifhipLastFarmTransferBytes(k, C.byref(dev), C.byref(b), C.byref(e), C.byref(up), C.byref(down)) == 0
        out.append((dev.value, b.value, e.value, up.value, down.value))
    return out


def test_yuv_to_rgb_sweep_through_three_workers(farm3):
    farm3.avifhipSetTiledKernels(1)
    be, oracle = H.hip_host_backend(), H.oracle_backend()
    cases = H.y2r_sweep(FARMED, n_random=260, seed=41)
    bad, farmed, launches0 = [], 0, farm3.avifhipLaunchCount()
    for c in cases:
        ro, po = H.run_y2r(oracle, c)
        rh, ph = H.run_y2r(be, c)
        if ro == 0 and farm3.avifhipLastFarmWorkers() >= 2:
            farmed += 1
            rows = [(b, e) for (_, b, e, _, _) in reports(farm3)]
            assert rows[0][0] == 0 and rows[-1][1] == c.h and all(rows[k][1] == rows[k + 1][0] for k in range(len(rows) - 1)), rows
        if ro != rh or not np.array_equal(po, ph):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    assert farmed >= len(cases) // 2, (farmed, len(cases))
    assert farm3.avifhipLaunchCount() > launches0  # the workers' launches are counted on the calling thread


def test_integer_path_sweep_through_three_workers(farm3):
    farm3.avifhipSetArithmetic(0)
    try:
        be, oracle = H.hip_host_backend(), H.oracle_libyuv_backend()
        cases = H.libyuv_y2r_cases(FARMED, n_random=200, seed=43)
        bad = []
        for c in cases:
            ro, po = H.run_y2r(oracle, c)
            rh, ph = H.run_y2r(be, c)
            if ro != rh or not np.array_equal(po, ph):
                bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
        assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    finally:
        farm3.avifhipSetArithmetic(1)


def test_rgb_to_yuv_sweep_through_three_workers(farm3):
    be, oracle = H.hip_host_backend(), H.oracle_backend()
    cases = H.r2y_sweep(FARMED, n_random=200, seed=47)
    bad, farmed = [], 0
    for c in cases:
        ro, io = H.run_r2y(oracle, c)
        rh, ih = H.run_r2y(be, c)
        farmed += farm3.avifhipLastFarmWorkers() >= 2
        diff = None if ro != 0 else H.planes_equal(io, ih)
        if ro != rh or diff:
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh} {diff or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    assert farmed >= len(cases) // 3, (farmed, len(cases))


def test_alpha_passes_in_place_through_three_workers(farm3):
    be, oracle = H.hip_host_backend(), H.oracle_backend()
    for depth in (8, 10, 16):
        for fmt in (abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_ABGR):
            for fn_name in ("premultiply", "unpremultiply"):
                imgs = []
                for b in (oracle, be):
                    rgb = abi.make_rgb(333, 150, depth, fmt, avoid_libyuv=True, row_pad=6, fill=0x5A)
                    synth.fill_rgb(rgb, 0x77 + depth)
                    if depth == 10:
                        rgb.pixels.view(np.uint16)[...] &= 1023
                    assert getattr(b, fn_name)(rgb.struct) == 0
                    imgs.append(rgb.pixels.copy())
                assert farm3.avifhipLastFarmWorkers() == 3
                assert np.array_equal(imgs[0], imgs[1]), (depth, fmt, fn_name, H.describe_diff(imgs[0], imgs[1]))


def test_headline_frame_and_grid_canvas_on_two_workers(farm2):
    """BASELINE cfg2 (8K 8-bit 4:2:0 -> RGBA8 bilinear, both arithmetics) and cfg5's stitched canvas (15360 x 8640 10-bit 4:2:0 -> RGBA(10))
    from host memory to host memory over two workers: the oracle's bytes, and each worker moves about half of them."""
    cfg2 = H.Y2RCase(7680, 4320, yuv_depth=8, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1,
                     upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=True)
    cfg5 = H.Y2RCase(15360, 8640, yuv_depth=10, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1, rgb_depth=10,
                     upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
    be = H.hip_host_backend()
    for case, oracle, arith in ((cfg2, H.oracle_backend(), 1), (dataclasses.replace(cfg2, avoid_libyuv=False), H.oracle_libyuv_backend(), 0), (cfg5, H.oracle_backend(), 0)):
        farm2.avifhipSetArithmetic(arith)
        try:
            ro, po = H.run_y2r(oracle, case)
            rh, ph = H.run_y2r(be, case)
        finally:
            farm2.avifhipSetArithmetic(1)
        assert ro == 0 and rh == 0, (case.ident(), ro, rh, farm2.avifhipLastError())
        rep = reports(farm2)
        assert len(rep) == 2 and rep[0][1:3] == (0, case.h // 2 if (case.h // 2) % 32 == 0 else ((case.h // 2 + 31) // 32) * 32), rep
        assert np.array_equal(po, ph), (case.ident(), native.last_kernel(), H.describe_diff(po, ph))
        bps = 2 if case.yuv_depth > 8 else 1
        plane_bytes = case.w * case.h * bps * 3 // 2
        pixel_bytes = case.w * case.h * abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
        ups, downs = [r[3] for r in rep], [r[4] for r in rep]
        assert sum(downs) == pixel_bytes
        # every plane sample once, plus the chroma row either side of the one seam (two planes, two rows)
        assert plane_bytes <= sum(ups) <= plane_bytes + 4 * (case.w // 2) * bps, (sum(ups), plane_bytes)
        for up, down in zip(ups, downs):
            assert 0.45 * plane_bytes <= up <= 0.55 * plane_bytes and 0.45 * pixel_bytes <= down <= 0.55 * pixel_bytes, rep
        total_up, total_down = C.c_uint64(0), C.c_uint64(0)
        farm2.avifhipLastTransferBytes(C.byref(total_up), C.byref(total_down))
        assert (total_up.value, total_down.value) == (sum(ups), sum(downs))
        _flush(farm2)


def test_encode_direction_4k_on_two_workers(farm2):
    """BASELINE cfg4 (3840 x 2160 RGBA8 -> 8-bit 4:2:0 BT.709 + alpha) from host to host over two workers."""
    case = H.R2YCase(3840, 2160, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, yuv_depth=8, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420,
                     yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1, avoid_libyuv=True)
    ro, io = H.run_r2y(H.oracle_backend(), case)
    rh, ih = H.run_r2y(H.hip_host_backend(), case)
    assert ro == 0 and rh == 0
    assert farm2.avifhipLastFarmWorkers() == 2
    assert H.planes_equal(io, ih) is None
    rep = reports(farm2)
    assert sum(r[3] for r in rep) == 3840 * 2160 * 4 and sum(r[4] for r in rep) == 3840 * 2160 * 5 // 2


def test_rectangles_of_a_canvas_over_two_workers(farm3):
    """avifhipImageYUVToRGBRects under a device set: contiguous blocks of the coalesced job list per worker, same bytes as the whole canvas."""
    case = H.Y2RCase(1100, 300, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=10, upsampling=4)
    res, whole = H.run_y2r(H.oracle_backend(), case)
    assert res == 0
    canvas, out = H.make_y2r_inputs(case), H.make_y2r_output(case)
    rects = farm.grid_rects(case.w, case.h, 512, 64)
    crops = (abi.avifCropRect * len(rects))(*[abi.avifCropRect(*r) for r in rects])
    native.check(farm3.avifhipImageYUVToRGBRects(canvas.struct, out.struct, crops, len(rects)), "avifhipImageYUVToRGBRects")
    rep = reports(farm3)
    assert len(rep) == 3 and rep[0][1] == 0 and rep[-1][2] == 5  # five tile rows (coalesced), blocks of 2 + 2 + 1
    wb = case.w * abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
    assert np.array_equal(out.pixels[:, :wb], whole[:, :wb]), H.describe_diff(whole[:, :wb], out.pixels[:, :wb])
    up, down = C.c_uint64(0), C.c_uint64(0)
    planned_up, planned_down = C.c_uint64(0), C.c_uint64(0)
    farm3.avifhipLastTransferBytes(C.byref(up), C.byref(down))
    native.check(farm3.avifhipPlanRectTransfers(canvas.struct, out.struct, crops, len(rects), C.byref(planned_up), C.byref(planned_down)), "avifhipPlanRectTransfers")
    assert (up.value, down.value) == (planned_up.value, planned_down.value)


def test_a_set_with_a_device_that_does_not_exist_is_refused(hip):
    n = hip.avifhipDeviceCount()
    assert hip.avifhipSetDeviceSet((C.c_int * 2)(0, n), 2) != 0
    assert b"not one of" in hip.avifhipLastError()
    have = (C.c_int * 4)()
    assert hip.avifhipGetDeviceSet(have, 4) == 0  # the refused set left nothing behind


def test_single_device_calls_report_no_farm(hip):
    case = H.Y2RCase(512, 128, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, upsampling=4)
    res, _ = H.run_y2r(H.hip_host_backend(), case)
    assert res == 0 and hip.avifhipLastFarmWorkers() == 0
    up, down = C.c_uint64(0), C.c_uint64(0)
    hip.avifhipLastTransferBytes(C.byref(up), C.byref(down))
    assert down.value == 512 * 128 * 4 and up.value >= 512 * 128 * 3 // 2


def test_gain_maps_between_farmed_calls(hip):
    """Round 5's open fault, as a test: host-resident gain-map applications (scaled gain maps: the window scaling kernel) after the device farm
    has been used in the same process, then the farm again.  The reference's functions are stateless (src/gainmap.c:73, src/reformat.c:1709-1735):
    what ran before -- worker threads come and gone, their contexts pooled, their device buffers freed and recycled -- must not matter."""
    import gainmap_cases as G
    import oracle_lib
    import test_gainmap as TG

    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    big = H.Y2RCase(2560, 2560, yuv_depth=8, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1,
                    upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=True)
    want_r, want = H.run_y2r(H.oracle_backend(), big)
    maps = [G.GainMapCase(66, 64, gm_w=23, gm_h=24, gm_format=abi.AVIF_PIXEL_FORMAT_YUV422, seed=3),
            G.GainMapCase(300, 200, gm_w=77, gm_h=51, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420, out_depth=10, out_tc=16, seed=4),
            G.GainMapCase(640, 360, gm_w=320, gm_h=180, gm_format=abi.AVIF_PIXEL_FORMAT_YUV444, seed=5),
            G.GainMapCase(129, 65, gm_w=64, gm_h=33, gm_format=abi.AVIF_PIXEL_FORMAT_YUV400, gm_depth=10, seed=6)]
    for round_ in range(3):
        assert hip.avifhipSetDeviceSet((C.c_int * 3)(0, 0, 0), 3) == 0
        try:
            got_r, got = H.run_y2r(H.hip_host_backend(), big)
            assert hip.avifhipLastFarmWorkers() >= 2
        finally:
            assert hip.avifhipSetDeviceSet(None, 0) == 0  # the workers end: their contexts go back to the pool
        assert got_r == want_r == 0 and np.array_equal(got, want), H.describe_diff(want, got)
        hip.avifhipSetArithmetic(0)
        try:
            for c in maps:
                ra, pa, ca = TG.run(o.oracleRGBImageApplyGainMap, c, 1)
                rb, pb, cb = TG.run(hip.avifhipRGBImageApplyGainMap, c, C.byref(diag))
                assert ra == rb == 0 and np.array_equal(pa, pb) and ca[0] == cb[0] and abs(ca[1] - cb[1]) <= 1, (round_, c.ident(), native.last_kernel())
        finally:
            hip.avifhipSetArithmetic(1)
