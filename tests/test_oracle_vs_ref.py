"""Pins the oracle: the plain-C restatement (oracle/reformat_oracle.c) must be byte-identical to the reference
compiled from its own sources (oracle/_ref/libavif_ref.so, built by oracle/Makefile with libyuv OFF) on every
entry point, over the configuration sweep.  CPU only.  Skipped (not failed) where the reference build is absent."""
import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi

SIZES = [(37, 21), (1, 1), (2, 2), (1, 6), (6, 1), (3, 5), (127, 10), (64, 33)]

pytestmark = pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def backends():
    return H.oracle_backend(), H.libavif_backend(oracle_lib.ref(), "reference")


def test_yuv_to_rgb_sweep(backends):
    o, r = backends
    cases = H.y2r_sweep(SIZES, n_random=1500)
    assert len(cases) > 1500
    bad = []
    for c in cases:
        ro, po = H.run_y2r(o, c)
        rr, pr = H.run_y2r(r, c)
        if ro != rr or not np.array_equal(po, pr):
            bad.append(f"{c.ident()}: results {ro}/{rr}" + ("" if ro != rr else " " + H.describe_diff(po, pr)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


def test_rgb_to_yuv_sweep(backends):
    o, r = backends
    cases = H.r2y_sweep(SIZES, n_random=1000)
    bad = []
    for c in cases:
        ro, io = H.run_r2y(o, c)
        rr, ir = H.run_r2y(r, c)
        d = None if ro != rr else H.planes_equal(io, ir)
        if ro != rr or d:
            bad.append(f"{c.ident()}: results {ro}/{rr} {d or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.parametrize("depth", [8, 10, 12, 16])
@pytest.mark.parametrize("fmt", list(range(10)))
def test_premultiply_unpremultiply(backends, fmt, depth):
    o, r = backends
    if fmt == abi.AVIF_RGB_FORMAT_RGB_565 and depth != 8:
        pytest.skip("565 is 8-bit only")
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(61, 17, depth, fmt, row_pad=6, fill=0x5A)
        from libavif_amd import synth
        synth.fill_rgb(a, 0xBEEF + fmt + depth)
        if depth in (10, 12):
            # channels are stored in 16-bit containers: keep them within the depth, plus a few out-of-range alphas
            ch = a.pixels.view(np.uint16)
            ch &= (1 << depth) - 1
        b = abi.make_rgb(61, 17, depth, fmt, row_pad=6)
        b.pixels[...] = a.pixels
        ra = getattr(o, which)(a.struct)
        rb = getattr(r, which)(b.struct)
        assert ra == rb, (which, fmt, depth)
        assert np.array_equal(a.pixels, b.pixels), (which, fmt, depth, H.describe_diff(a.pixels, b.pixels))


def test_exhaustive_alpha_pairs_8bit(backends):
    """All 65,536 (colour, alpha) pairs, both directions (SURVEY.md appendix D.4 domain, float arithmetic)."""
    o, r = backends
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(256, 256, 8, abi.AVIF_RGB_FORMAT_RGBA)
        ch = a.channels()
        ch[:, :, 0] = np.arange(256)[None, :]
        ch[:, :, 1] = 255 - np.arange(256)[None, :]
        ch[:, :, 2] = (np.arange(256)[None, :] * 7) % 256
        ch[:, :, 3] = np.arange(256)[:, None]
        b = abi.make_rgb(256, 256, 8, abi.AVIF_RGB_FORMAT_RGBA)
        b.pixels[...] = a.pixels
        assert getattr(o, which)(a.struct) == getattr(r, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), which


@pytest.mark.parametrize("depth,fmt", [(10, abi.AVIF_RGB_FORMAT_RGBA), (10, abi.AVIF_RGB_FORMAT_ARGB), (12, abi.AVIF_RGB_FORMAT_BGRA)])
def test_exhaustive_alpha_pairs_10_and_12_bit(backends, depth, fmt):
    """Every (colour, alpha) pair of a 10- / 12-bit channel, both directions: what tests/test_gpu_parity.py holds the product's
    integer un-premultiply against is itself the reference's result."""
    o, r = backends
    n = 1 << depth
    a_first = fmt == abi.AVIF_RGB_FORMAT_ARGB
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(n, n, depth, fmt)
        ch = a.channels()
        cols = [k for k in range(4) if k != (0 if a_first else 3)]
        ch[:, :, cols[0]] = np.arange(n)[None, :]
        ch[:, :, cols[1]] = (n - 1) - np.arange(n)[None, :]
        ch[:, :, cols[2]] = (np.arange(n)[None, :] * 7) % n
        ch[:, :, 0 if a_first else 3] = np.arange(n)[:, None]
        b = abi.make_rgb(n, n, depth, fmt)
        b.pixels[...] = a.pixels
        assert getattr(o, which)(a.struct) == getattr(r, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), which


def test_limited_full_helpers():
    o, r = oracle_lib.oracle(), oracle_lib.ref()
    for depth in (8, 10, 12, 9):
        for v in range(-3, (1 << min(depth, 12)) + 3):
            assert o.oracleLimitedToFullY(depth, v) == r.avifLimitedToFullY(depth, v)
            assert o.oracleLimitedToFullUV(depth, v) == r.avifLimitedToFullUV(depth, v)
            assert o.oracleFullToLimitedY(depth, v) == r.avifFullToLimitedY(depth, v)
            assert o.oracleFullToLimitedUV(depth, v) == r.avifFullToLimitedUV(depth, v)


def test_error_codes_match(backends):
    """Argument/format errors of the reference (pinned by tests/gtest/avif_fuzztest_yuvrgb.cc:36-46)."""
    o, r = backends
    bad_cases = [
        H.Y2RCase(8, 8, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565, rgb_depth=10),
        H.Y2RCase(8, 8, rgb_depth=8, is_float=True),
        H.Y2RCase(8, 8, matrix=3),
        H.Y2RCase(8, 8, matrix=8, yuv_range=abi.AVIF_RANGE_LIMITED),
        H.Y2RCase(8, 8, matrix=0, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=1),
        H.Y2RCase(8, 8, matrix=10), H.Y2RCase(8, 8, matrix=11), H.Y2RCase(8, 8, matrix=13), H.Y2RCase(8, 8, matrix=14),
        H.Y2RCase(8, 8, matrix=18), H.Y2RCase(8, 8, matrix=16, yuv_depth=10, rgb_depth=10, yuv_range=1),
        H.Y2RCase(8, 8, matrix=17, yuv_depth=10, rgb_depth=8, yuv_range=1), H.Y2RCase(8, 8, rgb_depth=9),
    ]
    for c in bad_cases:
        ro, _ = H.run_y2r(o, c)
        rr, _ = H.run_y2r(r, c)
        assert ro == rr == abi.AVIF_RESULT_REFORMAT_FAILED, c.ident()
    # maxThreads < 0
    img = H.make_y2r_inputs(H.Y2RCase(8, 8))
    for be in (o, r):
        rgb = H.make_y2r_output(H.Y2RCase(8, 8))
        rgb.struct.maxThreads = -1
        assert be.yuv_to_rgb(img.struct, rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
    # RGB -> YUV: 565 fails, float is not implemented
    for be in (o, r):
        c = H.R2YCase(8, 8)
        rgb = H.make_r2y_inputs(c)
        out = H.make_r2y_output(c)
        rgb.struct.format = abi.AVIF_RGB_FORMAT_RGB_565
        assert be.rgb_to_yuv(out.struct, rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
        rgb.struct.format = abi.AVIF_RGB_FORMAT_RGBA
        rgb.struct.depth = 16
        rgb.struct.isFloat = 1
        assert be.rgb_to_yuv(out.struct, rgb.struct) == abi.AVIF_RESULT_NOT_IMPLEMENTED
    # premultiply on a format without alpha: INVALID_ARGUMENT vs REFORMAT_FAILED (src/alpha.c:159-161, :346-348)
    for be in (o, r):
        rgb = abi.make_rgb(4, 4, 8, abi.AVIF_RGB_FORMAT_RGB)
        assert be.premultiply(rgb.struct) == abi.AVIF_RESULT_INVALID_ARGUMENT
        assert be.unpremultiply(rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
        rgb = abi.make_rgb(4, 4, 8, abi.AVIF_RGB_FORMAT_RGBA, allocate=False)
        assert be.premultiply(rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
        assert be.unpremultiply(rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
