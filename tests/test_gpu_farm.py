"""The tile farm's product converter on the GPU: rectangles of a stitched canvas converted by batched launches
(avifhipImageYUVToRGBBatchAsync) must reproduce the whole-canvas conversion of the oracle byte for byte, seams included;
and the single-rectangle entry point (avifhipImageYUVToRGBRectAsync) must agree with the oracle's rectangle function."""
import ctypes as C
import dataclasses

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


def test_batches_of_canvas_tiles_along_the_canvas_rows(hip):
    """A batch whose jobs are the row-major tiles of one canvas (avifhipImageYUVToRGBBatchAsync: one device-resident canvas, its tile
    rectangles) walks along the canvas rows when it streams (round 5; api_batch.cpp infers the columns from the rectangles).
    TUNE_CANVAS_ORDER sends these small canvases the same way, in the packed and in the wave-private fp32 kernels, tall and short tiles."""
    try:
        for arith, oracle in ((1, H.oracle_backend()), (0, H.oracle_libyuv_backend())):
            hip.avifhipSetArithmetic(arith)
            for tuning in (0x2000001, 0x2000401, 0x2000009):
                hip.avifhipSetTuning(tuning)
                for case, (tw, th) in CASES:
                    case = dataclasses.replace(case, avoid_libyuv=(arith == 1))
                    res, whole = H.run_y2r(oracle, case)
                    assert res == 0
                    canvas, out = H.make_y2r_inputs(case), H.make_y2r_output(case)
                    dimg, drgb = device.DeviceYUV(canvas), device.DeviceRGB(out, upload=True)
                    rects = farm.grid_rects(case.w, case.h, tw, th)
                    n = len(rects)
                    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(dimg.struct)] * n)
                    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(drgb.struct)] * n)
                    crops = (abi.avifCropRect * n)(*[abi.avifCropRect(*r) for r in rects])
                    native.check(hip.avifhipImageYUVToRGBBatchAsync(n, imgs, rgbs, crops, None), "avifhipImageYUVToRGBBatchAsync")
                    native.check(hip.avifhipSynchronize(None), "sync")
                    drgb.download_into_host()
                    wb = case.w * abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
                    assert np.array_equal(out.pixels[:, :wb], whole[:, :wb]), (arith, hex(tuning), case.ident(), native.last_kernel(), H.describe_diff(whole[:, :wb], out.pixels[:, :wb]))
    finally:
        hip.avifhipSetTuning(1)
        hip.avifhipSetArithmetic(1)


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


# ---- two PROCESSES with the product converter: the multi-process HIP path itself, on the one GPU a test box has ----

def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _hip_rank(rank: int, world: int, port: int, out_dir: str) -> None:
    """One rank of the farm as bench.py --gpus N runs it, except that every rank uses device 0 and the ranks meet over gloo (RCCL
    wants one GPU per rank): own process, own HIP context and streams, HipRectConverter on this rank's block of tiles."""
    import os
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    import harness as Hh
    from libavif_amd import farm as F, native as N

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = N.load()
        assert lib.avifhipDeviceCount() >= 1
        N.check(lib.avifhipSetDevice(0), "avifhipSetDevice")
        lib.avifhipSetArithmetic(0)
        conv = F.HipRectConverter()
        for k, (case, (tw, th)) in enumerate(CASES[:3]):
            canvas = Hh.make_y2r_inputs(case)  # the same decoded canvas on every rank
            out = Hh.make_y2r_output(case)
            rects = F.grid_rects(case.w, case.h, tw, th)
            launches = lib.avifhipLaunchCount()
            elapsed = F.timed_region(lambda: F.convert_shard(canvas, out, rects, rank, world, conv), lambda: N.check(lib.avifhipSynchronize(None)), dist)
            assert lib.avifhipLaunchCount() > launches, "the HIP path did not run"
            t = torch.tensor([conv.bytes_up, conv.bytes_down], dtype=torch.int64)
            total = t.clone()
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
            np.save(os.path.join(out_dir, f"rgb_{k}_{rank}.npy"), out.pixels)
            np.save(os.path.join(out_dir, f"meta_{k}_{rank}.npy"), np.array([conv.bytes_up, conv.bytes_down, int(total[0]), int(total[1]), int(elapsed * 1e9)]))
    finally:
        dist.destroy_process_group()


def test_two_processes_share_one_gpu(hip, tmp_path):
    """The first 8-GPU run must not be the multi-process HIP path's first run: two ranks (torch.distributed over gloo), both on device 0, each
    converting its block of tiles through avifhipImageYUVToRGBRects; the union equals the oracle's whole-canvas conversion, nobody writes into
    the other's tiles, each rank moves about half of the canvas over the host link and the timed region reduces to one MAX."""
    import os

    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_hip_rank, args=(world, _free_port(), os.fspath(tmp_path)), nprocs=world, join=True)
    for k, (case, (tw, th)) in enumerate(CASES[:3]):
        res, whole = H.run_y2r(H.oracle_backend() if case.avoid_libyuv else H.oracle_libyuv_backend(), case)
        assert res == 0
        rects = farm.grid_rects(case.w, case.h, tw, th)
        px = abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
        union = np.full_like(whole, H.FILL_BYTE)
        metas = [np.load(tmp_path / f"meta_{k}_{r}.npy").tolist() for r in range(world)]
        for r in range(world):
            got = np.load(tmp_path / f"rgb_{k}_{r}.npy")
            mine = farm.shard(len(rects), r, world)
            for t, (x, y, w, h) in enumerate(rects):
                tile = got[y:y + h, x * px:(x + w) * px]
                if t in mine:
                    union[y:y + h, x * px:(x + w) * px] = tile
                else:
                    assert (tile == H.FILL_BYTE).all(), f"rank {r} wrote into tile {t} of the other rank"
        wb = case.w * px
        assert np.array_equal(union[:, :wb], whole[:, :wb]), (case.ident(), H.describe_diff(whole[:, :wb], union[:, :wb]))
        assert metas[0][2:4] == metas[1][2:4] and metas[0][4] == metas[1][4]  # the all-reduced totals and the MAX of the timed region
        total_up, total_down = metas[0][2], metas[0][3]
        assert total_down == case.w * case.h * px
        canvas, out = H.make_y2r_inputs(case), H.make_y2r_output(case)
        for r, (up, down, *_) in enumerate(metas):
            # exactly what the library announces for this rank's block of tiles (the cropped last column / row make the blocks unequal), and
            # never the whole canvas
            assert (up, down) == farm.planned_transfers(canvas, out, [rects[t] for t in farm.shard(len(rects), r, world)]), (case.ident(), r, metas)
            assert 0 < up < total_up and 0 < down < total_down, metas
