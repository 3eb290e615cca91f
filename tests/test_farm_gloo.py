"""The N > 1 path on CPU: two processes (torch.distributed, gloo, 127.0.0.1) farm the tiles of a grid canvas with
libavif_amd.farm; the rectangle converter injected here is the ORACLE's canvas-rectangle entry point (test
infrastructure -- the product converter, HipRectConverter, needs a GPU and is exercised by tests/test_gpu_farm.py).
Checks: the shards partition the tile list, no rank touches another rank's tiles, the union of the shards equals the
whole-canvas conversion byte for byte (seams included: 4:2:0 bilinear), and the timed region reduces with MAX."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C

    import torch.distributed as dist

    import harness as H
    import oracle_lib
    from libavif_amd import abi, farm

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle_lib.oracle()
        case = H.Y2RCase(300, 86, yuv_depth=10, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1,
                         rgb_depth=10, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR)
        canvas = H.make_y2r_inputs(case)  # same seed on every rank: every rank holds the decoded canvas
        rgb = H.make_y2r_output(case)
        rects = farm.grid_rects(case.w, case.h, 64, 32)  # 5 x 3 tiles, cropped last column/row

        def convert_rects(cv, out, rs):
            for (x, y, w, h) in rs:
                r = abi.avifCropRect(x, y, w, h)
                assert o.oracleImageYUVToRGBRect(cv.struct, out.struct, C.byref(r)) == 0

        elapsed = farm.timed_region(lambda: farm.convert_shard(canvas, rgb, rects, rank, world, convert_rects), lambda: None, dist)
        mine = farm.shard(len(rects), rank, world)
        # every rank sees the same MAX
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(float(g.item()) == elapsed for g in gathered)
        np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb.pixels)
        np.save(os.path.join(out_dir, f"mine_{rank}.npy"), np.array(mine))
        # what the PRODUCT converter (avifhipImageYUVToRGBRects) would move over this rank's host link for its tiles: planned on
        # the host by the library itself (no GPU needed), summed over the ranks by a gloo all-reduce
        up, down = farm.planned_transfers(canvas, rgb, [rects[t] for t in mine])
        mine_t = torch.tensor([up, down], dtype=torch.int64)
        total_t = mine_t.clone()
        dist.all_reduce(total_t, op=dist.ReduceOp.SUM)
        np.save(os.path.join(out_dir, f"bytes_{rank}.npy"), np.array([up, down, int(total_t[0]), int(total_t[1])]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_farm_a_grid(tmp_path):
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "tests"))
    import harness as H
    from libavif_amd import abi, farm

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), os.fspath(tmp_path)), nprocs=world, join=True)

    case = H.Y2RCase(300, 86, yuv_depth=10, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1,
                     rgb_depth=10, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR)
    res, whole = H.run_y2r(H.oracle_backend(), case)
    assert res == 0
    rects = farm.grid_rects(case.w, case.h, 64, 32)
    px = abi.rgb_pixel_size(case.rgb_format, case.rgb_depth)
    shards = [np.load(tmp_path / f"mine_{r}.npy").tolist() for r in range(world)]
    assert sorted(shards[0] + shards[1]) == list(range(len(rects))) and not set(shards[0]) & set(shards[1])
    union = np.full_like(whole, H.FILL_BYTE)
    for r in range(world):
        got = np.load(tmp_path / f"rgb_{r}.npy")
        for t, (x, y, w, h) in enumerate(rects):
            tile = got[y:y + h, x * px:(x + w) * px]
            if t in shards[r]:
                union[y:y + h, x * px:(x + w) * px] = tile
            else:
                assert (tile == H.FILL_BYTE).all(), f"rank {r} wrote into tile {t} of another rank"
    assert np.array_equal(union[:, : case.w * px], whole[:, : case.w * px]), H.describe_diff(whole, union)
    # host-link traffic of the product converter: every rank moves about 1/N of the canvas, not the whole of it
    bytes_ = [np.load(tmp_path / f"bytes_{r}.npy").tolist() for r in range(world)]
    canvas_in = case.w * case.h * 2 + 2 * ((case.w + 1) // 2) * ((case.h + 1) // 2) * 2  # 10-bit 4:2:0 planes
    canvas_out = case.w * case.h * px
    assert bytes_[0][2:] == bytes_[1][2:]  # the all-reduced totals
    total_up, total_down = bytes_[0][2], bytes_[0][3]
    assert total_down == canvas_out  # every pixel comes back exactly once
    assert canvas_in <= total_up <= 1.35 * canvas_in  # every sample once, plus the one-sample chroma halo around 64 x 32 tiles
    for up, down, _, _ in bytes_:
        assert 0.35 * total_up <= up <= 0.65 * total_up and 0.35 * total_down <= down <= 0.65 * total_down, bytes_


def test_shard_and_grid_helpers():
    sys.path.insert(0, str(ROOT))
    from libavif_amd import abi, farm

    assert farm.grid_rects(10, 5, 4, 4) == [(0, 0, 4, 4), (4, 0, 4, 4), (8, 0, 2, 4), (0, 4, 4, 1), (4, 4, 4, 1), (8, 4, 2, 1)]
    for n, world in ((64, 8), (7, 3), (1, 4), (0, 2)):
        parts = [farm.shard(n, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        farm.validate_rects([(1, 0, 4, 4)], abi.AVIF_PIXEL_FORMAT_YUV420)
    with pytest.raises(ValueError):
        farm.shard(4, 2, 2)
    farm.validate_rects([(1, 1, 4, 4)], abi.AVIF_PIXEL_FORMAT_YUV444)
