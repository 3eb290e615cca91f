"""Pins the INTEGER-path oracle (oracle/libyuv_oracle.c): a libavif built with libyuv, restated.

libyuv's source is not under /root/reference (third-party, pinned 1949); the only libyuv-enabled libavif available
offline is Pillow's bundled binary (libavif 1.4.1 + libyuv 1922).  These tests call that binary's PUBLIC entry
points next to the restatement on identical inputs and require identical bytes and identical avifResult codes.
CPU only.  Where the Pillow binary is absent (the GPU box) they are skipped; the golden fixtures generated from it
(tests/golden/yuvlib_*.npz, tests/test_golden.py) still run.
"""
import itertools
import random
from dataclasses import replace

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi, synth

SIZES = [(37, 21), (1, 1), (2, 2), (1, 6), (6, 1), (3, 5), (127, 10), (64, 33), (200, 127)]

pytestmark = pytest.mark.skipif(oracle_lib.pillow() is None, reason="no libyuv-enabled libavif binary on this machine")


@pytest.fixture(scope="module")
def backends():
    return H.oracle_libyuv_backend(), H.libavif_backend(oracle_lib.pillow(), "pillow-libavif+libyuv"), H.oracle_backend()


def _compare_y2r(o, p, f, cases):
    bad, integer = [], 0
    for c in cases:
        ro, po = H.run_y2r(o, c)
        rp, pp = H.run_y2r(p, c)
        if ro != rp or not np.array_equal(po, pp):
            bad.append(f"{c.ident()}: results {ro}/{rp}" + ("" if ro != rp else " " + H.describe_diff(po, pp)))
            continue
        _, pf = H.run_y2r(f, c)
        integer += int(not np.array_equal(po, pf))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    return integer


def test_yuv_to_rgb_general_sweep(backends):
    """The whole configuration space with the default avoidLibYUV=0 (plus a slice with avoidLibYUV=1, where only the
    (un)premultiply post-pass may still go to libyuv, src/alpha.c:163)."""
    o, p, f = backends
    cases = [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SIZES, n_random=1200, seed=101)]
    cases += H.y2r_sweep(SIZES[:3], n_random=200, seed=5)
    integer = _compare_y2r(o, p, f, cases)
    assert integer > 100  # the sweep really reaches the fixed-point path


def test_yuv_to_rgb_libyuv_domain(backends):
    o, p, f = backends
    cases = H.libyuv_y2r_cases(SIZES)
    integer = _compare_y2r(o, p, f, cases)
    assert integer > 0.45 * len(cases)  # (about half of the domain takes the fixed-point path: 1 122-1 190 of 2 250 over the seed rotations)


def test_rgb_to_yuv(backends):
    o, p, f = backends
    cases = [replace(c, avoid_libyuv=False) for c in H.r2y_sweep(SIZES, n_random=600, seed=103)] + H.libyuv_r2y_cases(SIZES)
    bad, integer = [], 0
    for c in cases:
        ro, io = H.run_r2y(o, c)
        rp, ip = H.run_r2y(p, c)
        d = None if ro != rp else H.planes_equal(io, ip)
        if ro != rp or d:
            bad.append(f"{c.ident()}: results {ro}/{rp} {d or ''}")
            continue
        _, i_f = H.run_r2y(f, c)
        integer += int(H.planes_equal(io, i_f) is not None)
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    assert integer > 500


def test_rgb_to_yuv_exhaustive_luma(backends):
    """Every (R,G,B) through the BT.601 luma formulas, limited and full."""
    o, p, _ = backends
    for yr in (0, 1):
        for r0 in range(0, 256, 16):
            c = H.R2YCase(4096, 256, rgb_depth=8, yuv_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGB, matrix=6, yuv_range=yr,
                          yuv_format=abi.AVIF_PIXEL_FORMAT_YUV444, avoid_libyuv=False)
            outs = []
            for be in (o, p):
                rgb = H.make_r2y_inputs(c)
                ch = rgb.channels()
                g, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
                for k in range(16):
                    ch[:, 256 * k:256 * (k + 1), 0] = r0 + k
                    ch[:, 256 * k:256 * (k + 1), 1] = g
                    ch[:, 256 * k:256 * (k + 1), 2] = b
                img = H.make_r2y_output(c)
                assert be.rgb_to_yuv(img.struct, rgb.struct) == 0
                outs.append(img)
            assert H.planes_equal(outs[0], outs[1]) is None, (yr, r0)


@pytest.mark.parametrize("fmt", [abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGRA, abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_ABGR])
def test_exhaustive_alpha_pairs_8bit(backends, fmt):
    """All 65,536 (colour, alpha) pairs, both directions (ARGBAttenuate / ARGBUnattenuate for RGBA and BGRA, the fp32
    path for the layouts libyuv is not asked about)."""
    o, p, _ = backends
    a_first = fmt in (abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_ABGR)
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(256, 256, 8, fmt)
        ch = a.channels()
        cols = [k for k in range(4) if k != (0 if a_first else 3)]
        ch[:, :, cols[0]] = np.arange(256)[None, :]
        ch[:, :, cols[1]] = 255 - np.arange(256)[None, :]
        ch[:, :, cols[2]] = (np.arange(256)[None, :] * 7) % 256
        ch[:, :, 0 if a_first else 3] = np.arange(256)[:, None]
        b = abi.make_rgb(256, 256, 8, fmt)
        b.pixels[...] = a.pixels
        assert getattr(o, which)(a.struct) == getattr(p, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), (which, H.describe_diff(a.pixels, b.pixels))


@pytest.mark.parametrize("depth", [8, 10, 12, 16])
def test_premultiply_other_depths_and_errors(backends, depth):
    o, p, _ = backends
    for fmt in range(10):
        if fmt == abi.AVIF_RGB_FORMAT_RGB_565 and depth != 8:
            continue
        for which in ("premultiply", "unpremultiply"):
            a = abi.make_rgb(61, 17, depth, fmt, row_pad=6, fill=0x5A)
            synth.fill_rgb(a, 0xBEEF + fmt + depth)
            if depth in (10, 12):
                a.pixels.view(np.uint16)[...] &= (1 << depth) - 1
            b = abi.make_rgb(61, 17, depth, fmt, row_pad=6)
            b.pixels[...] = a.pixels
            assert getattr(o, which)(a.struct) == getattr(p, which)(b.struct), (which, fmt, depth)
            assert np.array_equal(a.pixels, b.pixels), (which, fmt, depth)


def test_half_float_pass_is_the_builtin_one(backends):
    """HalfFloatPlane (src/reformat_libyuv.c:1163-1179) and the built-in avifRGBImageToF16 (src/reformat.c:1419-1443)
    agree on every 16-bit value, so the restatement keeps a single half-float pass."""
    o, p, _ = backends
    c = H.Y2RCase(256, 256, yuv_depth=12, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV444, rgb_depth=16, is_float=True, rgb_format=abi.AVIF_RGB_FORMAT_RGB,
                  avoid_libyuv=False, pattern="gradient", yuv_range=abi.AVIF_RANGE_FULL)
    ro, po = H.run_y2r(o, c)
    rp, pp = H.run_y2r(p, c)
    assert ro == rp == 0 and np.array_equal(po, pp)
