"""BASELINE.json's five configurations at their full sizes, on the GPU, compared byte for byte with the oracles (the C
oracles convert an 8K frame in well under a second, so no sampling is needed), through both boundaries: the synchronous
C ABI on host buffers and the device-resident Async entry points.  Plus the size-independent properties the domain offers
at those sizes (lossless identity / YCgCo-Re round trips, opaque alpha == no alpha)."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, native
from test_gpu_grid import run_grid

pytestmark = pytest.mark.gpu

BILINEAR = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR


def _y2r(be, oracle, c):
    ro, po = H.run_y2r(oracle, c)
    rh, ph = H.run_y2r(be, c)
    assert ro == rh == abi.AVIF_RESULT_OK, (c.ident(), ro, rh)
    assert np.array_equal(po, ph), (c.ident(), native.last_kernel(), H.describe_diff(po, ph))
    return native.last_kernel()


def _r2y(be, oracle, c):
    ro, io = H.run_r2y(oracle, c)
    rh, ih = H.run_r2y(be, c)
    assert ro == rh == abi.AVIF_RESULT_OK, (c.ident(), ro, rh)
    assert H.planes_equal(io, ih) is None, (c.ident(), native.last_kernel(), H.planes_equal(io, ih))
    return native.last_kernel()


def _backends():
    return [("host", H.hip_host_backend()), ("device", H.HipDeviceBackend())]


# configs[0]: single 256x256 8-bit YUV420 BT.601 -> RGBA8 (the reference's CPU-runnable case, tests/avifyuv.c)
CFG1 = H.Y2RCase(256, 256, yuv_depth=8, yuv_format=3, yuv_range=1, matrix=6, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA)
# configs[1]: 7680x4320 8-bit YUV420 BT.709 limited -> RGBA8, bilinear (the headline)
CFG2 = H.Y2RCase(7680, 4320, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA,
                 upsampling=BILINEAR)
# configs[2]: 7680x4320 10-bit YUV444 BT.2020 full -> RGBA16 + alpha premultiply
CFG3 = H.Y2RCase(7680, 4320, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=9, alpha=True, rgb_depth=16,
                 rgb_format=abi.AVIF_RGB_FORMAT_RGBA, rgb_premultiplied=True)
# configs[3]: RGBA8 -> YUV420 8-bit BT.709, 3840x2160
CFG4 = H.R2YCase(3840, 2160, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1)


def test_cfg1_both_arithmetics(hip):
    for arithmetic, oracle, avoid in ((1, H.oracle_backend(), True), (0, H.oracle_libyuv_backend(), False)):
        hip.avifhipSetArithmetic(arithmetic)
        try:
            for _, be in _backends():
                for up in (0, 1, 2, 3, 4):
                    _y2r(be, oracle, replace(CFG1, avoid_libyuv=avoid, upsampling=up))
        finally:
            hip.avifhipSetArithmetic(1)


def test_cfg2_fp32(hip):
    for name, be in _backends():
        k = _y2r(be, H.oracle_backend(), CFG2)
        assert k.startswith("yuv2rgb_tile<u8,420,bilinear,rgba8"), k


def test_cfg2_default_arithmetic(hip_auto_arithmetic):
    for name, be in _backends():
        k = _y2r(be, H.oracle_libyuv_backend(), replace(CFG2, avoid_libyuv=False))
        assert k.startswith("yuv2rgb_fixed_tile<u8,420,bilinear,rgba8"), k  # the integer tiled kernel served the headline configuration


def test_cfg2_nearest_and_padded_rows(hip_auto_arithmetic):
    be = H.HipDeviceBackend()
    _y2r(be, H.oracle_libyuv_backend(), replace(CFG2, avoid_libyuv=False, upsampling=abi.AVIF_CHROMA_UPSAMPLING_NEAREST))
    _y2r(be, H.oracle_libyuv_backend(), replace(CFG2, avoid_libyuv=False, row_pad=64, seed=99))


def test_8k_frames_of_the_other_chroma_layouts_default_arithmetic(hip_auto_arithmetic):
    """The headline's frame size in 4:2:2 and 4:4:4, 8- and 10-bit planes, bilinear: the launch geometry a frame of this size selects is not the
    one the parity sweeps' small images see (round 5: 4:2:2 with four strips per wave staged two chroma rows too few -- wrong rows in every
    4:2:2 image of ~8 megapixels and more since round 2, found only when another rule forced that geometry on a small grid)."""
    be = H.HipDeviceBackend()
    for depth, yf, fmt in ((8, 2, abi.AVIF_RGB_FORMAT_RGBA), (10, 2, abi.AVIF_RGB_FORMAT_BGRA), (8, 1, abi.AVIF_RGB_FORMAT_RGB), (12, 3, abi.AVIF_RGB_FORMAT_RGBA)):
        k = _y2r(be, H.oracle_libyuv_backend(), replace(CFG2, yuv_depth=depth, yuv_format=yf, rgb_format=fmt, avoid_libyuv=False, seed=depth * 31 + yf))
        assert "tile<" in k, k


def test_cfg3(hip):
    for name, be in _backends():
        _y2r(be, H.oracle_backend(), CFG3)


def test_cfg3_default_arithmetic_is_the_same_path(hip_auto_arithmetic):
    # 16-bit RGB is outside libyuv's domain (src/reformat_libyuv.c:237-247): the default build computes fp32 too
    _y2r(H.HipDeviceBackend(), H.oracle_libyuv_backend(), replace(CFG3, avoid_libyuv=False))


def test_cfg4_fp32(hip):
    for name, be in _backends():
        _r2y(be, H.oracle_backend(), CFG4)
        _r2y(be, H.oracle_backend(), replace(CFG4, yuv_range=1))


def test_cfg4_default_arithmetic(hip_auto_arithmetic):
    o = H.oracle_libyuv_backend()
    for name, be in _backends():
        _r2y(be, o, replace(CFG4, avoid_libyuv=False))                        # BT.709: libyuv has no matrix, fp32 path
        _r2y(be, o, replace(CFG4, avoid_libyuv=False, matrix=6))              # BT.601 limited: libyuv's ARGBToI420
        _r2y(be, o, replace(CFG4, avoid_libyuv=False, matrix=6, yuv_range=1))  # BT.601 full: ARGBToJ420


# configs[4]: 8x8 grid of 1920x1080 10-bit YUV420 tiles -> RGB canvas (here the whole grid on one GPU)
def _cfg5(rgb_depth, avoid):
    return H.GridCase(8, 8, 1920, 1080, 8 * 1920, 8 * 1080,
                      H.Y2RCase(0, 0, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=rgb_depth, upsampling=BILINEAR,
                                avoid_libyuv=avoid))


def test_cfg5_grid_rgba8_default_arithmetic(hip_auto_arithmetic):
    run_grid(hip_auto_arithmetic, _cfg5(8, False))


def test_cfg5_grid_rgba10_fp32(hip):
    run_grid(hip, _cfg5(10, True))


def test_large_batches_of_other_layouts(hip_auto_arithmetic):
    """Batches whose planes exceed the Infinity Cache take the streaming-load instantiations of every kernel family: 4:4:4 + alpha tiles
    (wave-private fp32 kernels), 12-bit 4:2:0 through the reduction to 8 bits and 10-bit 4:2:2 (packed kernels), nearest upsampling."""
    for g in (H.GridCase(6, 6, 1920, 1080, 6 * 1920, 6 * 1080, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=9, rgb_depth=16, alpha=True,
                                                                         rgb_premultiplied=True, avoid_libyuv=False)),
              H.GridCase(7, 7, 1920, 1080, 7 * 1920, 7 * 1080, H.Y2RCase(0, 0, yuv_depth=12, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=BILINEAR,
                                                                         avoid_libyuv=False)),
              H.GridCase(6, 6, 1920, 1080, 6 * 1920, 6 * 1080, H.Y2RCase(0, 0, yuv_depth=10, yuv_format=2, yuv_range=0, matrix=9, rgb_depth=8, upsampling=3,
                                                                         rgb_format=abi.AVIF_RGB_FORMAT_RGB, avoid_libyuv=False))):
        run_grid(hip_auto_arithmetic, g)


# ---------------------------------------------------------------------------------------------------
# size-independent properties at the full sizes


def _round_trip(be, w, h, rgb_depth, yuv_depth, yuv_format, matrix, avoid):
    src = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=avoid)
    dst = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=avoid)
    img = abi.make_yuv(w, h, yuv_depth, yuv_format, abi.AVIF_RANGE_FULL, matrix, with_alpha=True)
    rng = np.random.default_rng(5)
    ch = src.channels()
    ch[...] = rng.integers(0, 1 << rgb_depth, size=ch.shape, dtype=np.uint16 if rgb_depth > 8 else np.uint8)
    if yuv_format == abi.AVIF_PIXEL_FORMAT_YUV400:
        ch[:, :, 1] = ch[:, :, 0]
        ch[:, :, 2] = ch[:, :, 0]
    assert be.rgb_to_yuv(img.struct, src.struct) == abi.AVIF_RESULT_OK
    assert be.yuv_to_rgb(img.struct, dst.struct) == abi.AVIF_RESULT_OK
    return src, dst


@pytest.mark.parametrize("rgb_depth,yuv_depth,yuv_format,matrix", [(8, 8, 1, 0), (8, 12, 1, 0), (10, 10, 1, 0), (12, 12, 1, 0), (16, 16, 1, 0),
                                                                   (8, 10, 1, 16), (10, 12, 1, 16), (8, 8, 4, 6), (12, 12, 4, 6)])
@pytest.mark.parametrize("arithmetic", [0, 1], ids=["default", "fp32"])
def test_8k_lossless_round_trips(hip, arithmetic, rgb_depth, yuv_depth, yuv_format, matrix):
    """Identity, YCgCo-Re and gray->monochrome RGB -> YUV -> RGB round trips are lossless (the reference's Identity*,
    YCgCo_Re8b and MonochromeLossless* suites, tests/gtest/avifrgbtoyuvtest.cc:598-726), here on 7680x4320 frames."""
    hip.avifhipSetArithmetic(arithmetic)
    try:
        src, dst = _round_trip(H.hip_host_backend(), 7680, 4320, rgb_depth, yuv_depth, yuv_format, matrix, avoid=False)
    finally:
        hip.avifhipSetArithmetic(1)
    assert np.array_equal(src.pixels, dst.pixels)


def test_8k_opaque_alpha_equals_no_alpha(hip_auto_arithmetic):
    """tests/gtest/avifalphapremtest.cc:15-62 at 8K, 4:2:0 bilinear."""
    c = replace(CFG2, avoid_libyuv=False, alpha=True, image_premultiplied=True, rgb_format=abi.AVIF_RGB_FORMAT_RGB)
    img = H.make_y2r_inputs(c)
    img.alpha[...] = 255
    out_a, out_b = H.make_y2r_output(c), H.make_y2r_output(c)
    assert hip_auto_arithmetic.avifhipImageYUVToRGB(img.struct, out_a.struct) == abi.AVIF_RESULT_OK
    img.struct.alphaPlane = None
    img.struct.alphaRowBytes = 0
    assert hip_auto_arithmetic.avifhipImageYUVToRGB(img.struct, out_b.struct) == abi.AVIF_RESULT_OK
    assert np.array_equal(out_a.pixels, out_b.pixels)
