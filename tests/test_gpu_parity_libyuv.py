"""Parity tests proper (-m gpu) for the reference's INTEGER path: the HIP library in its default arithmetic
(AVIFHIP_ARITHMETIC_AUTO = what a libavif built with libyuv computes) against the integer-path oracle
(oracle/libyuv_oracle.c, pinned against the libyuv-enabled binary by tests/test_libyuv_oracle.py and the
tests/golden/yuvlib_* fixtures).  Bit-exact on every byte of every output buffer, through the C ABI on host buffers
and on device-resident buffers."""
from dataclasses import replace

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, farm, native

pytestmark = pytest.mark.gpu

SMALL = [(37, 21), (1, 1), (2, 2), (1, 6), (6, 1), (3, 5), (127, 10), (64, 33)]
TILED = [(512, 16), (300, 21), (256, 8), (777, 35), (1027, 18)]


def _compare_y2r(be, oracle, cases):
    bad, kernels = [], {}
    for c in cases:
        ro, po = H.run_y2r(oracle, c)
        rh, ph = H.run_y2r(be, c)
        k = native.last_kernel().split("<")[0]
        kernels[k] = kernels.get(k, 0) + 1
        if ro != rh or not np.array_equal(po, ph):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    return kernels


def test_yuv_to_rgb_libyuv_domain_host(hip_auto_arithmetic):
    kernels = _compare_y2r(H.hip_host_backend(), H.oracle_libyuv_backend(), H.libyuv_y2r_cases(SMALL + TILED, n_random=900))
    assert sum(v for k, v in kernels.items() if "fixed" in k) > 600, kernels


def test_yuv_to_rgb_libyuv_domain_device(hip_auto_arithmetic):
    kernels = _compare_y2r(H.HipDeviceBackend(), H.oracle_libyuv_backend(), H.libyuv_y2r_cases(SMALL + TILED, n_random=400, seed=77))
    assert sum(v for k, v in kernels.items() if "fixed" in k) > 300, kernels


def _wide_cases():
    import itertools
    cases = []
    for (w, h), depth, yf, up, fmt in itertools.product(TILED, (10, 12), (1, 2, 3, 4), (3, 4), (0, 1, 2, 4)):
        for alpha in ((False, True) if fmt != 0 else (False,)):
            cases.append(H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, upsampling=up, rgb_format=fmt, rgb_depth=8, alpha=alpha, avoid_libyuv=False,
                                   matrix=(1, 6, 9)[(w + depth + yf) % 3], yuv_range=(w + yf + fmt) % 2, row_pad=64 if (h + fmt) % 2 else 0,
                                   seed=(w * 31 + depth * 7 + yf * 5 + up * 3 + fmt) | 1))
    return cases


def test_wide_planes_run_in_the_packed_kernels(hip_auto_arithmetic):
    """10- and 12-bit planes to 8-bit RGB through libyuv's high-bit-depth entries (I010 / I210 / I410 / I012 and their alpha twins) or through
    libavif's reduction to 8 bits (src/reformat_libyuv.c:714-772, :906-930): every chroma layout, both upsamplings, 3- and 4-byte pixels, alpha
    from the plane -- byte-exact, and served by the packed 16-bit kernels unless an (un)premultiply follows."""
    cases = _wide_cases()
    kernels = {}
    bad = []
    o = H.oracle_libyuv_backend()
    for be in (H.HipDeviceBackend(), H.hip_host_backend()):
        for c in cases:
            ro, po = H.run_y2r(o, c)
            rh, ph = H.run_y2r(be, c)
            k = native.last_kernel()
            kernels[k] = kernels.get(k, 0) + 1
            if ro != rh or not np.array_equal(po, ph):
                bad.append(f"{c.ident()} [{k}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {2 * len(cases)} cases differ:\n" + "\n".join(bad[:25])
    packed = sum(v for k, v in kernels.items() if "u16" in k and "pk16" in k)
    assert packed > 0.6 * 2 * len(cases), kernels  # the rest: layouts libyuv has no entry for (fp32 kernels), small leftovers
    assert not [k for k in kernels if k.startswith("yuv2rgb_fixed_tile<u16") and "alphamul" not in k and "pk16" not in k], kernels


def test_packed_kernels_at_every_tile_height(hip_auto_arithmetic):
    """The packed kernels' launch geometry is chosen by image size (tile_geom.h pkGeometry: 2 or 4 strips per wave, waves stacked or side by
    side, per-XCD chunks or raster order), so the parity sweeps' small images only ever see one of the geometries.  Every geometry is forced
    here (plan.h TuningBits) on images small enough for the oracle: all chroma layouts, nearest and bilinear, 8- / 10- / 12-bit planes, with
    and without alpha, RGBA and RGB.  Round 5 found the 4:2:2 bilinear kernel with four strips per wave -- what 8-megapixel images select --
    staging two chroma rows too few (rows 6 and 7 of every wave wrong): this test fails on that build."""
    import itertools
    cases = []
    for (w, h), depth, yf, up, fmt in itertools.product([(777, 70), (512, 64), (1027, 35)], (8, 10, 12), (1, 2, 3, 4), (3, 4), (abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_RGB)):
        cases.append(H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, upsampling=up, rgb_format=fmt, rgb_depth=8, alpha=(w + depth) % 3 == 0, avoid_libyuv=False,
                               matrix=(1, 6, 9)[(w + depth + yf) % 3], yuv_range=(w + yf + fmt) % 2, seed=(w * 23 + depth * 7 + yf * 5 + up * 3 + fmt) | 1))
    o = H.oracle_libyuv_backend()
    want = [H.run_y2r(o, c) for c in cases]
    be = H.HipDeviceBackend()
    bad = []
    try:
        tunings = [(f"strips {st} waves-x code {wx} bands {b}", b | (st << 8) | (wx << 16)) for st, wx, b in itertools.product((2, 4), (1, 2, 3), (0, 1))]
        # ... and round 1's cooperative 32-bit kernels of the 10- / 12-bit family (plan.h TUNE_COOPERATIVE), 1 or 2 strips, runs of 1 or 3 tiles
        tunings += [(f"cooperative, strips {st} run {run} bands {b}", 0x4 | b | (st << 8) | (run << 12)) for st, run, b in itertools.product((1, 2), (1, 3), (0, 1))]
        for label, tuning in tunings:
            strips, waves_x, bands = label, "", ""
            hip_auto_arithmetic.avifhipSetTuning(tuning)
            for c, (ro, po) in zip(cases, want):
                rh, ph = H.run_y2r(be, c)
                if ro != rh or not np.array_equal(po, ph):
                    bad.append(f"{label}: {c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" +
                               ("" if ro != rh else " " + H.describe_diff(po, ph)))
    finally:
        hip_auto_arithmetic.avifhipSetTuning(1)
    assert not bad, f"{len(bad)} differ:\n" + "\n".join(bad[:25])


def test_premultiplied_outputs_fuse_the_attenuate_pass(hip_auto_arithmetic):
    """Images with an alpha plane into premultiplied RGBA / BGRA (Android's bitmaps): libyuv's conversion followed by ARGBAttenuate
    (src/reformat.c:1574-1585 -> src/alpha.c:163), in one pass of the packed kernels -- 8-, 10- and 12-bit planes, every chroma layout."""
    import itertools
    cases = []
    for (w, h), depth, yf, up, fmt in itertools.product(TILED, (8, 10, 12), (1, 2, 3, 4), (3, 4), (1, 4)):
        cases.append(H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, upsampling=up, rgb_format=fmt, rgb_depth=8, alpha=True, rgb_premultiplied=True, avoid_libyuv=False,
                               matrix=(1, 6, 9)[(w + depth + yf) % 3], yuv_range=(w + yf + fmt) % 2, row_pad=64 if (h + fmt) % 2 else 0,
                               seed=(w * 17 + depth * 11 + yf * 5 + up * 3 + fmt) | 1))
    kernels = {}
    bad = []
    o = H.oracle_libyuv_backend()
    for be in (H.HipDeviceBackend(), H.hip_host_backend()):
        for c in cases:
            ro, po = H.run_y2r(o, c)
            rh, ph = H.run_y2r(be, c)
            k = native.last_kernel()
            kernels[k] = kernels.get(k, 0) + 1
            if ro != rh or not np.array_equal(po, ph):
                bad.append(f"{c.ident()} [{k}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {2 * len(cases)} cases differ:\n" + "\n".join(bad[:25])
    fused = sum(v for k, v in kernels.items() if "alphamul,pk16" in k)
    assert fused > 0.6 * 2 * len(cases), kernels


def test_premultiplied_images_fuse_the_unattenuate_pass(hip_auto_arithmetic):
    """Images stored PREMULTIPLIED into straight-alpha RGBA / BGRA: libyuv's conversion followed by ARGBUnattenuate (src/reformat.c:1574-1585 ->
    src/alpha.c:350 -> src/reformat_libyuv.c:1138-1161), in one pass of the packed kernels (round 5; round 1's cooperative 32-bit kernel before)."""
    import itertools
    cases = []
    for (w, h), depth, yf, up, fmt in itertools.product(TILED, (8, 10, 12), (1, 2, 3, 4), (3, 4), (1, 4)):
        cases.append(H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, upsampling=up, rgb_format=fmt, rgb_depth=8, alpha=True, image_premultiplied=True, avoid_libyuv=False,
                               matrix=(1, 6, 9)[(w + depth + yf) % 3], yuv_range=(w + yf + fmt) % 2, row_pad=64 if (h + fmt) % 2 else 0,
                               seed=(w * 19 + depth * 13 + yf * 5 + up * 3 + fmt) | 1))
    kernels = {}
    bad = []
    o = H.oracle_libyuv_backend()
    for be in (H.HipDeviceBackend(), H.hip_host_backend()):
        for c in cases:
            ro, po = H.run_y2r(o, c)
            rh, ph = H.run_y2r(be, c)
            k = native.last_kernel()
            kernels[k] = kernels.get(k, 0) + 1
            if ro != rh or not np.array_equal(po, ph):
                bad.append(f"{c.ident()} [{k}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {2 * len(cases)} cases differ:\n" + "\n".join(bad[:25])
    fused = sum(v for k, v in kernels.items() if "alphamul,pk16" in k)
    assert fused > 0.6 * 2 * len(cases), kernels


def test_unattenuate_in_the_packed_kernel_every_colour_alpha_pair(hip_auto_arithmetic):
    """All 65,536 (colour byte, alpha byte) pairs through the conversion + ARGBUnattenuate of the packed kernel, the a == 1, c >= 128 -> 0
    artefact of libyuv's signed saturating pack included: a gray 4:4:4 image whose luma runs over every code in x and whose alpha runs over
    every code in y (full range, identity-like: R = G = B = Y for chroma 128), 8-bit planes -> RGBA8, image premultiplied."""
    w, h = 256, 256
    img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 6, with_alpha=True, alpha_premultiplied=True)
    img.planes[0][:h, :w] = np.arange(256, dtype=np.uint8)[None, :]
    img.planes[1][:h, :w] = 128
    img.planes[2][:h, :w] = 128
    img.alpha[:h, :w] = np.arange(256, dtype=np.uint8)[:, None]
    outs = []
    for be_name in ("oracle", "hip"):
        rgb = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False, fill=0x5A)
        be = H.oracle_libyuv_backend() if be_name == "oracle" else H.hip_host_backend()
        assert be.yuv_to_rgb(img.struct, rgb.struct) == 0
        outs.append(rgb.pixels.copy())
    assert "alphamul,pk16" in native.last_kernel(), native.last_kernel()
    px = outs[0].reshape(h, -1)[:, : 4 * w].reshape(h, w, 4)
    assert (px[1, 128:, 0] == 0).all() and px[1, 127, 0] == 255  # the artefact is in the oracle's picture: a == 1, c >= 128 -> 0
    assert len(np.unique(px[:, :, 0])) == 256
    assert np.array_equal(outs[0], outs[1]), H.describe_diff(outs[0], outs[1])


def test_wide_planes_cooperative_kernels_still_exact(hip_auto_arithmetic):
    """The round-1 kernels of the 10/12-bit integer family stay selectable for A/B runs (plan.h TUNE_COOPERATIVE): same bytes."""
    hip_auto_arithmetic.avifhipSetTuning(5)
    try:
        kernels = _compare_y2r(H.HipDeviceBackend(), H.oracle_libyuv_backend(), _wide_cases()[::7])
        assert "yuv2rgb_fixed_tile" in kernels, kernels
    finally:
        hip_auto_arithmetic.avifhipSetTuning(1)


def test_yuv_to_rgb_general_sweep_default_arithmetic(hip_auto_arithmetic):
    """The whole configuration space with avoidLibYUV = 0 (the API default) and a slice with avoidLibYUV = 1."""
    cases = [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SMALL + TILED[:2], n_random=700, seed=201)]
    cases += H.y2r_sweep(SMALL[:3] + TILED[:1], n_random=200, seed=9)
    _compare_y2r(H.hip_host_backend(), H.oracle_libyuv_backend(), cases)


def test_yuv_to_rgb_generic_kernels_only(hip_auto_arithmetic):
    hip_auto_arithmetic.avifhipSetTiledKernels(0)
    try:
        _compare_y2r(H.hip_host_backend(), H.oracle_libyuv_backend(), H.libyuv_y2r_cases(TILED[:3], n_random=300, seed=5))
    finally:
        hip_auto_arithmetic.avifhipSetTiledKernels(1)


def test_forced_libyuv_arithmetic_ignores_avoid_flag(hip):
    hip.avifhipSetArithmetic(2)
    try:
        for c in H.libyuv_y2r_cases([(300, 21)], n_random=60, seed=3)[:200]:
            ro, po = H.run_y2r(H.oracle_libyuv_backend(), c)
            rh, ph = H.run_y2r(H.hip_host_backend(), replace(c, avoid_libyuv=True))
            assert ro == rh and np.array_equal(po, ph), c.ident()
    finally:
        hip.avifhipSetArithmetic(1)


def _compare_r2y(be, oracle, cases, padding=True):
    bad, fixed = [], 0
    for c in cases:
        ro, io = H.run_r2y(oracle, c)
        rh, ih = H.run_r2y(be, c)
        fixed += "fixed" in native.last_kernel()
        d = None if ro != rh else H.planes_equal(io, ih, padding=padding)
        if ro != rh or d:
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh} {d or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    return fixed


def test_rgb_to_yuv_host(hip_auto_arithmetic):
    cases = H.libyuv_r2y_cases(SMALL + TILED[:3], n_random=600) + [replace(c, avoid_libyuv=False) for c in H.r2y_sweep(SMALL, n_random=200, seed=61)]
    assert _compare_r2y(H.hip_host_backend(), H.oracle_libyuv_backend(), cases) > 400


def test_rgb_to_yuv_device(hip_auto_arithmetic):
    cases = H.libyuv_r2y_cases(SMALL + TILED[:2], n_random=300, seed=31)
    assert _compare_r2y(H.HipDeviceBackend(), H.oracle_libyuv_backend(), cases, padding=False) > 200


@pytest.mark.parametrize("fmt", [abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGRA, abi.AVIF_RGB_FORMAT_ARGB])
def test_exhaustive_alpha_pairs_8bit(hip_auto_arithmetic, fmt):
    """All 65,536 (colour, alpha) pairs: ARGBAttenuate / ARGBUnattenuate for RGBA and BGRA, fp32 for ARGB."""
    o = H.oracle_libyuv_backend()
    a_first = fmt == abi.AVIF_RGB_FORMAT_ARGB
    for be in (H.hip_host_backend(), H.HipDeviceBackend()):
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
            if isinstance(be, H.HipDeviceBackend):
                be.bind_host(b.struct, b)
            assert getattr(o, which)(a.struct) == getattr(be, which)(b.struct) == 0
            assert np.array_equal(a.pixels, b.pixels), (be.name, which, H.describe_diff(a.pixels, b.pixels))
            assert ("fixed" in native.last_kernel()) == (fmt != abi.AVIF_RGB_FORMAT_ARGB)


def test_grid_farm_integer_path_equals_whole_canvas(hip_auto_arithmetic):
    """Tiles of a stitched canvas converted independently (chroma edge rules against the canvas) reproduce the
    whole-canvas conversion of the integer path, seams included."""
    conv = farm.HipRectConverter()
    for case, (tw, th) in [
        (H.Y2RCase(1100, 150, yuv_depth=10, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4, avoid_libyuv=False), (512, 64)),
        (H.Y2RCase(777, 66, yuv_depth=8, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4, avoid_libyuv=False), (256, 32)),
        (H.Y2RCase(640, 49, yuv_depth=8, yuv_format=2, yuv_range=1, matrix=6, rgb_depth=8, rgb_format=0, upsampling=4, avoid_libyuv=False), (320, 16)),
    ]:
        res, whole = H.run_y2r(H.oracle_libyuv_backend(), case)
        assert res == 0
        canvas = H.make_y2r_inputs(case)
        rects = farm.grid_rects(case.w, case.h, tw, th)
        out = H.make_y2r_output(case)
        farm.convert_shard(canvas, out, rects, 0, 1, conv)
        assert np.array_equal(out.pixels, whole), (case.ident(), H.describe_diff(whole, out.pixels))


def test_rgb565_through_the_packed_kernels(hip_auto_arithmetic):
    """RGB565 (Android's bitmap format): libyuv's I420ToRGB565Matrix / I422ToRGB565Matrix in the packed 16-bit tiled kernels, byte for
    byte like the integer-path oracle; formats libyuv has no 565 entry for stay with the universal kernels."""
    seen = set()
    for fmt in (abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_PIXEL_FORMAT_YUV422, abi.AVIF_PIXEL_FORMAT_YUV444):
        for (w, h) in ((1024, 64), (777, 35), (2052, 130)):
            for rng in (abi.AVIF_RANGE_LIMITED, abi.AVIF_RANGE_FULL):
                for mc in (1, 6, 9):
                    c = H.Y2RCase(w, h, yuv_format=fmt, yuv_range=rng, matrix=mc, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565, avoid_libyuv=False,
                                  upsampling=abi.AVIF_CHROMA_UPSAMPLING_FASTEST)
                    for be in (H.hip_host_backend(), H.HipDeviceBackend()):
                        ro, po = H.run_y2r(H.oracle_libyuv_backend(), c)
                        rh, ph = H.run_y2r(be, c)
                        assert ro == rh and np.array_equal(po, ph), (c.ident(), native.last_kernel(), H.describe_diff(po, ph))
                        seen.add((fmt, native.last_kernel()))
    assert (abi.AVIF_PIXEL_FORMAT_YUV420, "yuv2rgb_fixed_tile<u8,420,nearest,rgb565_8,pk16>") in seen, seen
    assert (abi.AVIF_PIXEL_FORMAT_YUV422, "yuv2rgb_fixed_tile<u8,422,nearest,rgb565_8,pk16>") in seen, seen
