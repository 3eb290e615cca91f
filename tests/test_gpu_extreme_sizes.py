"""The largest images libavif accepts by default (AVIF_DEFAULT_IMAGE_DIMENSION_LIMIT = 32768 per side, AVIF_DEFAULT_IMAGE_SIZE_LIMIT =
16384 x 16384 pixels, include/avif/avif.h) and the smallest, through the C ABI on the GPU, byte for byte against the oracles: sizes at
which 32-bit offsets, tile counts and grid dimensions are largest."""
from dataclasses import replace

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, native

pytestmark = pytest.mark.gpu


def _check(be, oracle, c, expect=None):
    ro, po = H.run_y2r(oracle, c)
    rh, ph = H.run_y2r(be, c)
    assert ro == rh == abi.AVIF_RESULT_OK, (c.ident(), ro, rh)
    assert np.array_equal(po, ph), (c.ident(), native.last_kernel(), H.describe_diff(po, ph))
    if expect:
        assert native.last_kernel().startswith(expect), (c.ident(), native.last_kernel())


@pytest.mark.parametrize("w,h", [(32768, 34), (34, 32768), (32767, 3), (3, 32767), (32768, 2)])
def test_longest_sides(hip_auto_arithmetic, w, h):
    """One side at the dimension limit: 128 bands of 256 pixels / 8192 tile rows; odd leftovers on both axes."""
    for c in (H.Y2RCase(w, h, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, avoid_libyuv=False),
              H.Y2RCase(w, h, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=9, upsampling=4, avoid_libyuv=False, alpha=True),
              H.Y2RCase(w, h, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=9, rgb_depth=16, alpha=True, rgb_premultiplied=True, avoid_libyuv=False),
              H.Y2RCase(w, h, yuv_format=2, yuv_range=1, matrix=6, rgb_format=abi.AVIF_RGB_FORMAT_RGB, upsampling=4, avoid_libyuv=False)):
        for be in (H.HipDeviceBackend(), H.hip_host_backend()):
            _check(be, H.oracle_libyuv_backend(), c)
    r = H.R2YCase(w, h, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1)
    ro, io = H.run_r2y(H.oracle_libyuv_backend(), r)
    rh, ih = H.run_r2y(H.HipDeviceBackend(), r)
    assert ro == rh == 0 and H.planes_equal(io, ih, padding=False) is None, (r.ident(), native.last_kernel())


def test_largest_default_image(hip_auto_arithmetic):
    """16384 x 16384 (268 megapixels, the default size limit): 8-bit 4:2:0 -> RGBA8 with the API defaults (1.07 GB of pixels, byte offsets up
    to 2^30) and 10-bit 4:4:4 + alpha -> premultiplied RGBA16 on the upper half (offsets up to 2^30 as well), device-resident."""
    be = H.HipDeviceBackend()
    _check(be, H.oracle_libyuv_backend(), H.Y2RCase(16384, 16384, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, avoid_libyuv=False, pattern="gradient"),
           "yuv2rgb_fixed_tile<u8,420,bilinear,rgba8,pk16")
    _check(be, H.oracle_libyuv_backend(), H.Y2RCase(16384, 8192, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=9, rgb_depth=16, alpha=True, rgb_premultiplied=True,
                                                    avoid_libyuv=False, pattern="gradient"), "yuv2rgb_tile<u16,444")


def test_smallest_images(hip_auto_arithmetic):
    for w, h in ((1, 1), (2, 1), (1, 2), (3, 3), (4, 2), (5, 2), (63, 2), (64, 2), (65, 3)):
        for yf, up in ((1, 3), (2, 4), (3, 4), (4, 3)):
            for be in (H.HipDeviceBackend(), H.hip_host_backend()):
                _check(be, H.oracle_libyuv_backend(), H.Y2RCase(w, h, yuv_format=yf, yuv_range=0, matrix=1, upsampling=up, avoid_libyuv=False))
                _check(be, H.oracle_libyuv_backend(), H.Y2RCase(w, h, yuv_depth=12, yuv_format=yf, yuv_range=1, matrix=9, rgb_depth=8, upsampling=up, avoid_libyuv=False, alpha=True))
