"""What the one-launch grid kernels rely on (tile_shared.h TileHalo), checked on the CPU with the pinned oracles: a tile of a
grid canvas converted from its OWN chroma samples plus ONE sample row / column of every neighbouring tile -- and clamped where the canvas
ends, the reference's border rule -- gives exactly the pixels the reference computes for that tile on the stitched canvas
(avifDecoderDataCopyTileToImage + avifImageYUVToRGB, src/read.c:1823-1877, src/reformat.c:766-816), corners, cropped last columns / rows
and odd canvas sizes included.  The GPU tests (tests/test_gpu_grid.py) check the kernels' bytes; this one checks the rule they implement,
where no GPU is needed: the expected canvas is oracleGridYUVToRGB (pinned against the reference's own functions by tests/test_grid_oracle.py),
the tile is converted by oracleImageYUVToRGB / the libyuv oracle (pinned by tests/test_oracle_vs_ref.py, tests/test_libyuv_oracle.py)."""
from dataclasses import replace

import numpy as np
import pytest

import harness as H
from libavif_amd import abi
from test_grid_oracle import oracle_grid

A = abi


def cases():
    Y = H.Y2RCase
    out = []
    for avoid in (True, False):
        base = dict(avoid_libyuv=avoid)
        out += [
            H.GridCase(3, 3, 64, 32, 192, 96, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, **base)),
            H.GridCase(3, 3, 64, 32, 150, 70, Y(0, 0, yuv_format=3, yuv_range=1, matrix=6, upsampling=4, rgb_format=A.AVIF_RGB_FORMAT_BGRA, **base)),
            H.GridCase(2, 3, 48, 16, 141, 31, Y(0, 0, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, rgb_format=A.AVIF_RGB_FORMAT_RGB, **base)),  # odd canvas
            H.GridCase(2, 2, 40, 12, 80, 24, Y(0, 0, yuv_format=2, yuv_range=1, matrix=6, upsampling=4, **base)),  # 4:2:2: columns only
            H.GridCase(4, 2, 32, 8, 62, 30, Y(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, rgb_depth=8 if not avoid else 10, upsampling=4, **base)),
        ]
    return out


def _samples(img, p):
    return img.plane_samples(p)


@pytest.mark.parametrize("g", cases(), ids=lambda g: g.ident())
def test_own_samples_plus_one_from_each_neighbour_is_the_stitched_canvas(g):
    tiles = H.make_grid_tiles(g)
    want = H.grid_output(g)
    assert oracle_grid(g, tiles, want, libyuv_build=not g.conv.avoid_libyuv) == 0
    c = g.conv
    sx = 1
    sy = 1 if c.yuv_format == A.AVIF_PIXEL_FORMAT_YUV420 else 0
    dt = np.uint16 if c.yuv_depth > 8 else np.uint8
    cw_all, ch_all = (g.out_w + sx) >> sx, (g.out_h + sy) >> sy
    # the stitched canvas (what libavif would build): every tile's part that is inside the output
    Yc = np.zeros((g.out_h, g.out_w), dt)
    Uc, Vc = np.zeros((ch_all, cw_all), dt), np.zeros((ch_all, cw_all), dt)
    for t, tile in enumerate(tiles):
        r, col = divmod(t, g.columns)
        X0, Y0 = col * g.tile_w, r * g.tile_h
        w, h = min(g.tile_w, g.out_w - X0), min(g.tile_h, g.out_h - Y0)
        Yc[Y0:Y0 + h, X0:X0 + w] = _samples(tile, 0)[:h, :w]
        cw, chh = (w + sx) >> sx, (h + sy) >> sy
        Uc[Y0 >> sy:(Y0 >> sy) + chh, X0 >> sx:(X0 >> sx) + cw] = _samples(tile, 1)[:chh, :cw]
        Vc[Y0 >> sy:(Y0 >> sy) + chh, X0 >> sx:(X0 >> sx) + cw] = _samples(tile, 2)[:chh, :cw]
    backend = H.oracle_backend() if c.avoid_libyuv else H.oracle_libyuv_backend()
    px = abi.rgb_pixel_size(c.rgb_format, c.rgb_depth)
    for t in range(g.rows * g.columns):
        r, col = divmod(t, g.columns)
        X0, Y0 = col * g.tile_w, r * g.tile_h
        w, h = min(g.tile_w, g.out_w - X0), min(g.tile_h, g.out_h - Y0)
        # the tile's chroma window and ONE sample beyond it on every side that has a neighbour
        cx0, cx1 = X0 >> sx, (X0 >> sx) + ((w + sx) >> sx) - 1
        cy0, cy1 = Y0 >> sy, (Y0 >> sy) + ((h + sy) >> sy) - 1
        ex0, ex1 = cx0 - (1 if col > 0 else 0), cx1 + (1 if col + 1 < g.columns else 0)
        ey0, ey1 = cy0 - (1 if (sy and r > 0) else 0), cy1 + (1 if (sy and r + 1 < g.rows) else 0)
        # ... and the luma that goes with those chroma samples (its values outside the tile do not reach the tile's pixels)
        lx0, lx1 = ex0 << sx, min(g.out_w, (ex1 + 1) << sx)
        ly0, ly1 = ey0 << sy, min(g.out_h, (ey1 + 1) << sy)
        ew, eh = lx1 - lx0, ly1 - ly0
        img = abi.make_yuv(ew, eh, c.yuv_depth, c.yuv_format, c.yuv_range, c.matrix, color_primaries=c.color_primaries)
        _samples(img, 0)[...] = Yc[ly0:ly1, lx0:lx1]
        _samples(img, 1)[...] = Uc[ey0:ey1 + 1, ex0:ex1 + 1]
        _samples(img, 2)[...] = Vc[ey0:ey1 + 1, ex0:ex1 + 1]
        rgb = H.make_y2r_output(replace(c, w=ew, h=eh))
        assert backend.yuv_to_rgb(img.struct, rgb.struct) == 0
        got = rgb.pixels[Y0 - ly0:Y0 - ly0 + h, (X0 - lx0) * px:(X0 - lx0 + w) * px]
        exp = want.pixels[Y0:Y0 + h, X0 * px:(X0 + w) * px]
        assert np.array_equal(got, exp), (g.ident(), (r, col), H.describe_diff(exp, got))
