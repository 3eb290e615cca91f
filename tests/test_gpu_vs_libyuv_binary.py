"""The product's default arithmetic against the REAL integer path: a libavif built with libyuv (Pillow's bundled libavif
1.4.1 + libyuv 1922, the only libyuv-enabled binary available offline -- it ships with the image, also on the GPU box), called
through its public avifImageYUVToRGB / avifImageRGBToYUV / premultiply entry points.  No oracle in between: every byte of the
GPU's output must equal the binary's, from the configuration sweep of the libyuv domain up to BASELINE.json's full sizes."""
from dataclasses import replace

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi, native

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(oracle_lib.pillow() is None, reason="Pillow's bundled libavif not present")]

SIZES = [(37, 21), (2, 2), (1, 6), (127, 10), (512, 16), (300, 21), (777, 35)]


def _binary():
    return H.libavif_backend(oracle_lib.pillow(), "libavif+libyuv")


def test_yuv_to_rgb_libyuv_domain(hip_auto_arithmetic):
    ref, be = _binary(), H.hip_host_backend()
    bad = []
    cases = H.libyuv_y2r_cases(SIZES, n_random=500, seed=101)
    for c in cases:
        ra, pa = H.run_y2r(ref, c)
        rb, pb = H.run_y2r(be, c)
        if ra != rb or not np.array_equal(pa, pb):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ra}/{rb}")
    assert not bad, f"{len(bad)} of {len(cases)} differ:\n" + "\n".join(bad[:20])


def test_yuv_to_rgb_whole_space_api_defaults(hip_auto_arithmetic):
    """avoidLibYUV = 0 everywhere: the binary decides per configuration between libyuv and its built-in loops, and so must we."""
    ref, be = _binary(), H.hip_host_backend()
    bad = []
    cases = [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SIZES[:4], n_random=400, seed=303)]
    for c in cases:
        ra, pa = H.run_y2r(ref, c)
        rb, pb = H.run_y2r(be, c)
        if ra != rb or not np.array_equal(pa, pb):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ra}/{rb}")
    assert not bad, f"{len(bad)} of {len(cases)} differ:\n" + "\n".join(bad[:20])


def test_rgb_to_yuv_libyuv_domain(hip_auto_arithmetic):
    ref, be = _binary(), H.hip_host_backend()
    bad = []
    cases = H.libyuv_r2y_cases(SIZES, n_random=300, seed=102)
    for c in cases:
        ra, ia = H.run_r2y(ref, c)
        rb, ib = H.run_r2y(be, c)
        if ra != rb or (ra == 0 and H.planes_equal(ia, ib, padding=False) is not None):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ra}/{rb}")
    assert not bad, f"{len(bad)} of {len(cases)} differ:\n" + "\n".join(bad[:20])


@pytest.mark.parametrize("up", [abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, abi.AVIF_CHROMA_UPSAMPLING_NEAREST, abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC])
def test_headline_configuration_at_8k(hip_auto_arithmetic, up):
    """BASELINE configs[1]: 7680x4320 8-bit 4:2:0 BT.709 limited -> RGBA8 with the API defaults, host and device-resident."""
    c = H.Y2RCase(7680, 4320, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, upsampling=up,
                  avoid_libyuv=False)
    ra, pa = H.run_y2r(_binary(), c)
    for be in (H.hip_host_backend(), H.HipDeviceBackend()):
        rb, pb = H.run_y2r(be, c)
        assert ra == rb == 0
        assert np.array_equal(pa, pb), (be.name, native.last_kernel(), H.describe_diff(pa, pb))
        assert "fixed" in native.last_kernel(), native.last_kernel()


def test_other_baseline_configurations_at_full_size(hip_auto_arithmetic):
    ref, be = _binary(), H.HipDeviceBackend()
    cfg1 = H.Y2RCase(256, 256, yuv_depth=8, yuv_format=3, yuv_range=1, matrix=6, avoid_libyuv=False)
    cfg5_tile = H.Y2RCase(1920, 1080, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
    cfg5_tile16 = replace(cfg5_tile, rgb_depth=10)  # API default depth: outside libyuv's domain, the binary's fp32 loops
    for c in (cfg1, cfg5_tile, cfg5_tile16):
        ra, pa = H.run_y2r(ref, c)
        rb, pb = H.run_y2r(be, c)
        assert ra == rb == 0 and np.array_equal(pa, pb), (c.ident(), native.last_kernel())
    cfg4_601 = H.R2YCase(3840, 2160, rgb_depth=8, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=6, avoid_libyuv=False)
    for c in (cfg4_601, replace(cfg4_601, matrix=1)):
        ra, ia = H.run_r2y(ref, c)
        rb, ib = H.run_r2y(be, c)
        assert ra == rb == 0 and H.planes_equal(ia, ib, padding=False) is None, (c.ident(), native.last_kernel())
