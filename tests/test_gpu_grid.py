"""The decode-side tail of a grid image on the GPU (avifhipGridYUVToRGBAsync): tiles stored separately in HBM are
converted where they lie -- no YUV canvas is materialised -- and the result must equal, byte for byte, what the
reference does in three steps (tile -> canvas copies, limited-range alpha conversion, avifImageYUVToRGB on the canvas),
seams included.  Expected values: oracleGridYUVToRGB, pinned against the reference's own functions by
tests/test_grid_oracle.py."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, device, native
from test_grid_oracle import oracle_grid

pytestmark = pytest.mark.gpu
A = abi


def cases(avoid_libyuv):
    base = dict(avoid_libyuv=avoid_libyuv)
    return [
        # cfg5 in miniature: 10-bit 4:2:0 limited BT.709 -> RGBA(10) / RGBA8 bilinear, cropped last column and row
        H.GridCase(3, 3, 512, 64, 1100, 150, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=10, upsampling=4, **base)),
        H.GridCase(3, 3, 512, 64, 1100, 150, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4, **base)),
        H.GridCase(2, 3, 256, 32, 701, 61, H.Y2RCase(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, alpha=True, **base)),
        H.GridCase(2, 2, 320, 40, 639, 79, H.Y2RCase(0, 0, yuv_format=2, yuv_range=1, matrix=6, rgb_format=A.AVIF_RGB_FORMAT_RGB, upsampling=4, **base)),
        H.GridCase(2, 2, 320, 16, 640, 32, H.Y2RCase(0, 0, yuv_depth=12, yuv_format=1, yuv_range=0, matrix=9, rgb_depth=16, alpha=True, rgb_premultiplied=True, **base)),
        H.GridCase(2, 4, 128, 24, 500, 41, H.Y2RCase(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=3, alpha=True, **base), alpha_limited=True),
        H.GridCase(3, 2, 64, 64, 127, 190, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, rgb_depth=16, upsampling=4, alpha=True, rgb_premultiplied=True, **base),
                   alpha_limited=True),
        H.GridCase(1, 1, 300, 22, 300, 22, H.Y2RCase(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, **base)),
        # no leftover columns / rows anywhere (every job entirely in the tiled kernels; seams across 2 to 4 tiles per side)
        H.GridCase(2, 3, 256, 32, 704, 60, H.Y2RCase(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, alpha=True, **base)),
        H.GridCase(3, 3, 256, 32, 768, 96, H.Y2RCase(0, 0, yuv_depth=12, yuv_format=3, yuv_range=1, matrix=9, rgb_depth=8, upsampling=4, **base)),
        H.GridCase(2, 2, 320, 40, 640, 80, H.Y2RCase(0, 0, yuv_format=2, yuv_range=1, matrix=6, rgb_format=A.AVIF_RGB_FORMAT_RGB, upsampling=4, **base)),
        H.GridCase(3, 2, 64, 64, 128, 192, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, rgb_depth=16, upsampling=4, alpha=True, rgb_premultiplied=True, **base),
                   alpha_limited=True),
        H.GridCase(4, 5, 64, 16, 316, 62, H.Y2RCase(0, 0, yuv_format=3, yuv_range=0, matrix=5, rgb_format=A.AVIF_RGB_FORMAT_BGRA, upsampling=4, **base)),
        H.GridCase(2, 2, 64, 16, 100, 30, H.Y2RCase(0, 0, yuv_format=3, yuv_range=1, matrix=6, rgb_format=A.AVIF_RGB_FORMAT_RGB_565, upsampling=4, **base)),  # universal kernel
    ]


def linked_cases(avoid_libyuv):
    """Grids every pixel of which goes through the tiled kernels (tile and canvas sizes in whole 4 x 2 pixel groups): the jobs are linked to
    their neighbours and ONE launch converts tiles and seams (the seam-aware builds, tile_impl.h TILE_SEAMS).  Every kernel flavour that
    stages a chroma neighbourhood, tiles narrower / wider than a 256-pixel band, shorter / taller than a kernel tile, canvases that end
    inside the last tile."""
    base = dict(avoid_libyuv=avoid_libyuv)
    Y = H.Y2RCase
    return [
        H.GridCase(3, 3, 512, 64, 1536, 192, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, **base)),
        H.GridCase(3, 3, 512, 64, 1100, 150, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, alpha=True, **base)),
        H.GridCase(2, 3, 320, 96, 960, 190, Y(0, 0, yuv_format=3, yuv_range=1, matrix=6, rgb_format=A.AVIF_RGB_FORMAT_BGRA, upsampling=4, **base)),
        H.GridCase(4, 2, 64, 32, 128, 126, Y(0, 0, yuv_format=3, yuv_range=0, matrix=5, rgb_format=A.AVIF_RGB_FORMAT_RGB, upsampling=4, **base)),
        H.GridCase(2, 2, 320, 40, 640, 80, Y(0, 0, yuv_format=2, yuv_range=1, matrix=6, upsampling=4, **base)),
        H.GridCase(2, 3, 256, 32, 704, 60, Y(0, 0, yuv_format=2, yuv_range=0, matrix=1, rgb_format=A.AVIF_RGB_FORMAT_ARGB, upsampling=4, alpha=True, rgb_premultiplied=True, **base)),
        # cfg5 in miniature, RGBA(10) and RGBA8, then 12-bit 4:2:2
        H.GridCase(3, 3, 512, 64, 1536, 192, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=10, upsampling=4, **base)),
        H.GridCase(3, 3, 512, 64, 1100, 150, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4, **base)),
        H.GridCase(2, 2, 768, 48, 1280, 96, Y(0, 0, yuv_depth=12, yuv_format=2, yuv_range=1, matrix=9, rgb_depth=16, upsampling=4, alpha=True, **base)),
        H.GridCase(3, 2, 64, 64, 128, 192, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, rgb_depth=16, upsampling=4, alpha=True, rgb_premultiplied=True, **base),
                   alpha_limited=True),
        H.GridCase(2, 2, 256, 64, 512, 128, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4, alpha=True, rgb_premultiplied=True, **base)),
        H.GridCase(2, 2, 256, 64, 512, 128, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, alpha=True, image_premultiplied=True, **base)),
        H.GridCase(2, 2, 320, 64, 600, 100, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, rgb_depth=16, rgb_format=A.AVIF_RGB_FORMAT_BGR, upsampling=4, **base)),
        # a canvas large enough for the wave-private fp32 kernels (8-bit planes from 6 megapixels) and tall tiles
        H.GridCase(2, 3, 1280, 1024, 3700, 1900, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, **base)),
    ]


def run_grid(lib, g, launches=None):
    tiles = H.make_grid_tiles(g)
    want = H.grid_output(g)
    assert oracle_grid(g, tiles, want, libyuv_build=not g.conv.avoid_libyuv) == 0
    dtiles = [device.DeviceYUV(t) for t in tiles]
    out = H.grid_output(g)
    drgb = device.DeviceRGB(out, upload=True)
    n = g.rows * g.columns
    P = C.POINTER(abi.avifImage)
    colour = (P * n)(*[C.pointer(d.struct) for d in dtiles])
    alpha = (P * n)(*[C.pointer(d.struct) for d in dtiles]) if g.conv.alpha else None
    grid = native.avifhipGrid(g.rows, g.columns, g.out_w, g.out_h)
    before = lib.avifhipLaunchCount()
    native.check(lib.avifhipGridYUVToRGBAsync(C.byref(grid), colour, alpha, int(g.alpha_limited), drgb.struct, None), "avifhipGridYUVToRGBAsync")
    native.check(lib.avifhipSynchronize(None), "sync")
    drgb.download_into_host()
    wb = g.out_w * abi.rgb_pixel_size(g.conv.rgb_format, g.conv.rgb_depth)
    assert np.array_equal(out.pixels[:, :wb], want.pixels[:, :wb]), (g.ident(), native.last_kernel(), H.describe_diff(want.pixels[:, :wb], out.pixels[:, :wb]))
    if launches is not None:
        assert lib.avifhipLaunchCount() - before == launches, (g.ident(), native.last_kernel())


@pytest.mark.parametrize("g", linked_cases(True), ids=lambda g: g.ident())
def test_linked_grid_is_one_launch_fp32_path(hip, g):
    run_grid(hip, g, launches=1)


@pytest.mark.parametrize("g", linked_cases(False), ids=lambda g: g.ident())
def test_linked_grid_is_one_launch_default_arithmetic(hip_auto_arithmetic, g):
    run_grid(hip_auto_arithmetic, g, launches=1)


def test_linked_grids_with_the_seam_pass_forced(hip, monkeypatch):
    """AVIFHIP_GRID_SEAM_PASS=1: the same grids through the tile batch + seam kernel of rounds 1-3 (two launches), same bytes."""
    monkeypatch.setenv("AVIFHIP_GRID_SEAM_PASS", "1")
    try:
        for avoid in (True, False):
            hip.avifhipSetArithmetic(1 if avoid else 0)
            for g in linked_cases(avoid)[:8]:
                run_grid(hip, g, launches=2)
    finally:
        hip.avifhipSetArithmetic(1)


def test_grids_along_the_canvas_rows(hip):
    """Round 5: a grid that streams (cfg5's 64 tiles: more than the Infinity Cache holds) is walked along the rows of the CANVAS by the
    wave-private kernels (tile_geom.h PkGeom::canvasColumns) instead of job by job.  TUNE_CANVAS_ORDER (plan.h) sends small grids the same
    way: every grid case of this file, linked or not, both arithmetics, tall and short tiles -- the oracle's bytes."""
    try:
        for avoid in (True, False):
            hip.avifhipSetArithmetic(1 if avoid else 0)
            for strips in (0, 2, 4):
                hip.avifhipSetTuning(0x2000001 | (strips << 8))
                for g in (linked_cases(avoid) + cases(avoid))[:: (1 if strips == 0 else 3)]:
                    run_grid(hip, g)
    finally:
        hip.avifhipSetTuning(1)
        hip.avifhipSetArithmetic(1)


@pytest.mark.parametrize("g", cases(True), ids=lambda g: g.ident())
def test_grid_fp32_path(hip, g):
    run_grid(hip, g)


@pytest.mark.parametrize("g", cases(False), ids=lambda g: g.ident())
def test_grid_default_arithmetic(hip_auto_arithmetic, g):
    run_grid(hip_auto_arithmetic, g)


@pytest.mark.parametrize("pairs", ["0", "1"])
def test_both_seam_kernels(hip, pairs, monkeypatch):
    """Seams are redone by one lane per pixel (small grids) or one lane per pair of pixels either side of a seam sharing the chroma quad
    (large ones); AVIFHIP_SEAM_PAIRS forces either, so that the small test grids go through both, in both arithmetics."""
    monkeypatch.setenv("AVIFHIP_SEAM_PAIRS", pairs)
    monkeypatch.setenv("AVIFHIP_GRID_SEAM_PASS", "1")  # (grids without leftovers would not reach the seam kernels otherwise)
    try:
        for avoid in (True, False):
            hip.avifhipSetArithmetic(1 if avoid else 0)
            for g in cases(avoid):
                run_grid(hip, g)
    finally:
        hip.avifhipSetArithmetic(1)


def test_grid_argument_checks(hip):
    g = cases(True)[0]
    tiles = H.make_grid_tiles(g)
    dtiles = [device.DeviceYUV(t) for t in tiles]
    out = H.grid_output(g)
    drgb = device.DeviceRGB(out)
    n = g.rows * g.columns
    P = C.POINTER(abi.avifImage)
    colour = (P * n)(*[C.pointer(d.struct) for d in dtiles])
    bad = native.avifhipGrid(g.rows, g.columns, g.out_w + 2 * g.tile_w, g.out_h)  # the tiles do not cover the output
    assert hip.avifhipGridYUVToRGBAsync(C.byref(bad), colour, None, 0, drgb.struct, None) == 18  # AVIF_RESULT_INVALID_IMAGE_GRID
    dtiles[3].struct.depth = 8  # a tile that does not match the first one, src/read.c:1832-1842
    ok = native.avifhipGrid(g.rows, g.columns, g.out_w, g.out_h)
    assert hip.avifhipGridYUVToRGBAsync(C.byref(ok), colour, None, 0, drgb.struct, None) == 18


def test_repeated_grid_call_reuses_the_table_on_the_device(hip):
    """A decoder converts into the same tile buffers frame after frame: the second call's descriptor table equals the first one's and is
    not sent again; new PIXELS in the same buffers are picked up (nothing derived from pixels is kept), other buffers send a new table."""
    g = cases(True)[2]
    tiles = H.make_grid_tiles(g)
    dtiles = [device.DeviceYUV(t) for t in tiles]
    n = g.rows * g.columns
    P = C.POINTER(abi.avifImage)
    colour = (P * n)(*[C.pointer(d.struct) for d in dtiles])
    grid = native.avifhipGrid(g.rows, g.columns, g.out_w, g.out_h)
    wb = g.out_w * abi.rgb_pixel_size(g.conv.rgb_format, g.conv.rgb_depth)

    def convert(into):
        native.check(hip.avifhipGridYUVToRGBAsync(C.byref(grid), colour, colour, int(g.alpha_limited), into.struct, None), "avifhipGridYUVToRGBAsync")
        native.check(hip.avifhipSynchronize(None), "sync")
        into.download_into_host()

    def expect(ts):
        want = H.grid_output(g)
        assert oracle_grid(g, ts, want, libyuv_build=False) == 0
        return want.pixels[:, :wb]

    out = H.grid_output(g)
    drgb = device.DeviceRGB(out, upload=True)
    before = hip.avifhipTableUploadCount()
    convert(drgb)
    assert hip.avifhipTableUploadCount() == before + 1
    assert np.array_equal(out.pixels[:, :wb], expect(tiles))
    out.pixels[:] = 0
    drgb.upload()
    convert(drgb)
    assert hip.avifhipTableUploadCount() == before + 1  # same buffers, same geometry: the table is already there
    assert np.array_equal(out.pixels[:, :wb], expect(tiles))
    # the next frame in the same buffers
    rng = np.random.default_rng(99)
    for t, d in zip(tiles, dtiles):
        for plane in t.planes:
            if plane is not None:
                plane[:] = rng.integers(0, 256, plane.shape, dtype=plane.dtype)
        if t.alpha is not None:
            t.alpha[:] = rng.integers(0, 256, t.alpha.shape, dtype=t.alpha.dtype)
        d.upload()
    convert(drgb)
    assert hip.avifhipTableUploadCount() == before + 1
    assert np.array_equal(out.pixels[:, :wb], expect(tiles))
    # another destination: another table
    out2 = H.grid_output(g)
    drgb2 = device.DeviceRGB(out2, upload=True)
    convert(drgb2)
    assert hip.avifhipTableUploadCount() == before + 2
    assert np.array_equal(out2.pixels[:, :wb], expect(tiles))
    # ... and back: the first table is no longer the resident one
    convert(drgb)
    assert hip.avifhipTableUploadCount() == before + 3
    assert np.array_equal(out.pixels[:, :wb], expect(tiles))
