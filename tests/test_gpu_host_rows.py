"""Host rows of unfriendly alignment (libavif_amd/csrc/api.cpp uploadRows / packRowsForDownload): hipMemcpy2DAsync between pageable memory and
the device falls back to one copy per row when the host pitch, the row width or the address is not a multiple of 4 -- every image of odd
width, whose planes avifImageAllocatePlanes packs tight.  Such rows cross the link as one block per band and change their pitch on the
device.  Same bytes (row padding untouched), and a 12-megapixel image of odd width now moves at the link's pace."""
import time

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, native

pytestmark = pytest.mark.gpu

# tight rows, padded rows (1 and 3 bytes: still unfriendly; 64: friendly pitch, odd width), 8- and 16-bit containers, every chroma layout
SIZES = [(1027, 70), (4099, 301), (2050, 132), (333, 9), (259, 8)]


def test_odd_widths_and_pitches_decode_direction(hip):
    be, oracle = H.hip_host_backend(), H.oracle_backend()
    bad = []
    for (w, h) in SIZES:
        for depth, yf, fmt, pad in ((8, 3, abi.AVIF_RGB_FORMAT_RGBA, 0), (8, 3, abi.AVIF_RGB_FORMAT_RGB, 0), (8, 2, abi.AVIF_RGB_FORMAT_BGR, 1), (10, 1, abi.AVIF_RGB_FORMAT_RGB, 0),
                                    (12, 3, abi.AVIF_RGB_FORMAT_RGBA, 3), (8, 4, abi.AVIF_RGB_FORMAT_RGB, 0), (8, 3, abi.AVIF_RGB_FORMAT_RGB_565, 0), (8, 1, abi.AVIF_RGB_FORMAT_ARGB, 64)):
            c = H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, rgb_format=fmt, rgb_depth=8 if fmt == abi.AVIF_RGB_FORMAT_RGB_565 else depth, alpha=(fmt == abi.AVIF_RGB_FORMAT_ARGB),
                          upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, row_pad=pad, seed=w * 7 + h + depth)
            if not H.valid_y2r(c):
                continue
            ro, po = H.run_y2r(oracle, c)
            rh, ph = H.run_y2r(be, c)
            if ro != rh or not np.array_equal(po, ph):
                bad.append(f"{c.ident()} pad {pad} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, "\n".join(bad[:20])


def test_odd_widths_and_pitches_encode_direction(hip):
    be, oracle = H.hip_host_backend(), H.oracle_backend()
    bad = []
    for (w, h) in SIZES:
        for depth, yf, fmt, pad in ((8, 3, abi.AVIF_RGB_FORMAT_RGBA, 0), (8, 3, abi.AVIF_RGB_FORMAT_RGB, 0), (8, 2, abi.AVIF_RGB_FORMAT_BGR, 1), (10, 1, abi.AVIF_RGB_FORMAT_RGB, 0),
                                    (12, 3, abi.AVIF_RGB_FORMAT_RGBA, 3), (8, 4, abi.AVIF_RGB_FORMAT_RGB, 0), (8, 1, abi.AVIF_RGB_FORMAT_ARGB, 64)):
            c = H.R2YCase(w, h, rgb_depth=depth, rgb_format=fmt, yuv_depth=depth, yuv_format=yf, row_pad=pad, seed=w * 5 + h + depth)
            ro, io = H.run_r2y(oracle, c)
            rh, ih = H.run_r2y(be, c)
            diff = None if ro != 0 else H.planes_equal(io, ih)
            if ro != rh or diff:
                bad.append(f"{c.ident()} pad {pad} [{native.last_kernel()}]: results {ro}/{rh} {diff or ''}")
    assert not bad, "\n".join(bad[:20])


def test_an_image_of_odd_width_moves_at_the_links_pace(hip_auto_arithmetic):
    """4099 x 3001 8-bit 4:2:0 with tight planes -> RGBA8, host to host: 56 ms through per-row copies (rounds 1-4), ~1.2 ms as blocks.  The bound is
    generous (boxes differ, the first call builds staging): 8 ms still tells the two apart."""
    c = H.Y2RCase(4099, 3001, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
    img, rgb = H.make_y2r_inputs(c), H.make_y2r_output(c)
    lib = hip_auto_arithmetic
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        assert lib.avifhipImageYUVToRGB(img.struct, rgb.struct) == 0
        best = min(best, time.perf_counter() - t0)
    ro, po = H.run_y2r(H.oracle_libyuv_backend(), c)
    assert ro == 0 and np.array_equal(po, rgb.pixels)
    assert best < 8e-3, f"{best * 1e3:.1f} ms for a 12-megapixel image"
    # ... and the way back: RGB8 (3-byte pixels, 12297-byte rows) -> tight planes of odd width
    e = H.R2YCase(4099, 3001, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGB, yuv_depth=8, yuv_format=3)
    src, dst = H.make_r2y_inputs(e), H.make_r2y_output(e)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        assert lib.avifhipImageRGBToYUV(dst.struct, src.struct) == 0
        best = min(best, time.perf_counter() - t0)
    assert best < 8e-3, f"{best * 1e3:.1f} ms for a 12-megapixel image (encode direction)"
