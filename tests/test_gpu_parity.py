"""Parity tests proper (-m gpu): the HIP library, called through its C ABI, against the oracle on the same
seeded inputs.  Bit-exact: every byte of every output buffer, row padding included.

Three routes are exercised for every case family:
  * hip-host   : synchronous libavif-style calls on host buffers (staged through HBM by the library);
  * hip-device : the Async entry points on device-resident buffers;
  * generic    : the same with the tiled kernels disabled (avifhipSetTiledKernels(0)), so that the universal
                 kernels and the bandwidth-tuned ones are each compared with the oracle, not with each other.
"""
import numpy as np
import pytest

import harness as H
from libavif_amd import abi, native

pytestmark = pytest.mark.gpu

SMALL = [(37, 21), (1, 1), (2, 2), (1, 6), (6, 1), (3, 5), (127, 10), (64, 33)]
# large enough for full 256x8 tiles plus partial right/bottom tiles
TILED = [(512, 16), (300, 21), (256, 8), (777, 35), (1027, 18)]


def _compare_y2r(be, oracle, cases, expect_kernel=None):
    bad, kernels = [], set()
    for c in cases:
        ro, po = H.run_y2r(oracle, c)
        rh, ph = H.run_y2r(be, c)
        kernels.add(native.last_kernel().split("<")[0])
        if ro != rh or not np.array_equal(po, ph):
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    if expect_kernel:
        assert expect_kernel in kernels, kernels
    return kernels


def test_yuv_to_rgb_small_sweep_host(hip):
    hip.avifhipSetTiledKernels(1)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), H.y2r_sweep(SMALL, n_random=700))


def test_yuv_to_rgb_small_sweep_device(hip):
    hip.avifhipSetTiledKernels(1)
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), H.y2r_sweep(SMALL, n_random=300, seed=23))


def test_yuv_to_rgb_tiled_sweep_host(hip):
    hip.avifhipSetTiledKernels(1)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), H.y2r_sweep(TILED, n_random=500, seed=5), "yuv2rgb_tile")
    # (until round 6 part of this sweep went through the universal kernel -- RGB565 on 2-byte-aligned rows, with an alpha plane to multiply in, from
    #  10- / 12-bit planes in the integer arithmetic: test_universal_kernels_serve_only_the_enumerated_rest holds what is left)


def test_fp32_tiles_at_every_launch_geometry(hip):
    """The fp32 tiles' launch geometry follows the image size too (wave-private kernels with 2 or 4 strips per wave, waves stacked or side by
    side, per-XCD chunks or raster order; the cooperative runs with 1 or 2 strips and runs of 1-3 tiles): every geometry forced on images the
    oracle converts in milliseconds (the twin of test_gpu_parity_libyuv.py::test_packed_kernels_at_every_tile_height, which found a geometry
    that only 8-megapixel 4:2:2 images select computing wrong rows)."""
    import itertools
    cases = []
    for (w, h), depth, yf, up in itertools.product([(777, 70), (512, 64), (1027, 35)], (8, 10, 12), (1, 2, 3, 4), (3, 4)):
        rd = (8, depth, 16)[(w + yf + up) % 3]
        cases.append(H.Y2RCase(w, h, yuv_depth=depth, yuv_format=yf, upsampling=up, rgb_format=(abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGR)[(w + depth) % 2], rgb_depth=rd,
                               alpha=(w + depth) % 3 == 0, rgb_premultiplied=(w + depth + yf) % 6 == 0, matrix=(1, 6, 9)[(w + depth + yf) % 3], yuv_range=(w + yf) % 2,
                               seed=(w * 29 + depth * 7 + yf * 5 + up * 3) | 1))
    cases = [c for c in cases if H.valid_y2r(c)]
    o = H.oracle_backend()
    want = [H.run_y2r(o, c) for c in cases]
    be = H.HipDeviceBackend()
    bad = []
    try:
        hip.avifhipSetTiledKernels(1)
        tunings = [b | (st << 8) | (wx << 16) | 0x8 for st, wx, b in itertools.product((2, 4), (1, 2, 3), (0, 1))]   # wave-private kernels for every family
        tunings += [0x4 | b | (st << 8) | (run << 12) for st, run, b in itertools.product((1, 2), (1, 3), (0, 1))]     # cooperative runs
        for t in tunings:
            hip.avifhipSetTuning(t)
            for c, (ro, po) in zip(cases, want):
                rh, ph = H.run_y2r(be, c)
                if ro != rh or not np.array_equal(po, ph):
                    bad.append(f"tuning {t:#x}: {c.ident()} [{native.last_kernel()}]: results {ro}/{rh}" + ("" if ro != rh else " " + H.describe_diff(po, ph)))
    finally:
        hip.avifhipSetTuning(1)
    assert not bad, f"{len(bad)} differ:\n" + "\n".join(bad[:25])


def test_universal_kernels_serve_only_the_enumerated_rest(hip):
    """What still reaches the one-lane-per-pixel kernels once a conversion fills the tiled kernels' 4 x 2 pixel groups (VERDICT r04 #5, r05 #8;
    tests/tools/list_generic.py prints the census): since round 6 NOTHING of this sweep, in either arithmetic and under every seed rotation
    tried (profiles/r06_generic_rest.txt: 0 of 1082-1093 over rotations 0-3; round 5: 18 / 28 of 1089).  The rest that had been enumerated:
    the pending alpha (un)multiplies that mix the two arithmetics -- on ARGB / ABGR libyuv attenuates nothing (RGBA / BGRA only), so the
    reference runs its fp32 post-pass over libyuv's bytes; on RGBA / BGRA converted by the fp32 loops the reference runs libyuv's
    ARGBAttenuate / ARGBUnattenuate over fp32's bytes (`postMulFx`) -- now post-passes of the tiled kernels that converted the pixels; and
    RGB565 on rows that are 2-byte aligned only (odd widths, padded pitches: 8-byte stores at the alignment the format has), with an alpha
    plane to multiply in inside the loop (src/reformat.c:1503-1511: the fp32 kernels with alpha arithmetic), from 10- / 12-bit planes in the
    integer arithmetic (Convert16To8Plane in the packed kernels' front end).  What the universal kernels keep is outside this sweep: images
    below 64 x 2 pixels, bases / pitches of planes or of 3- / 4-channel pixels off their vector alignment, matrices whose divisor is off the
    verified list (exactdiv.h), the <= 3 leftover columns and <= 1 leftover row of every image."""
    from dataclasses import replace
    try:
        for arith, avoid in ((1, True), (0, False)):
            hip.avifhipSetArithmetic(arith)
            hip.avifhipSetTiledKernels(1)
            be = H.HipDeviceBackend()
            generic, total = [], 0
            for c in H.y2r_sweep(TILED, n_random=600, seed=5):
                c = replace(c, avoid_libyuv=avoid)
                res, _ = H.run_y2r(be, c)
                if res != 0:
                    continue
                total += 1
                if "generic" in native.last_kernel():
                    generic.append(c)
            assert total > 900
            assert not generic, (arith, len(generic), total, [c.ident() for c in generic[:10]])
    finally:
        hip.avifhipSetArithmetic(1)


def test_half_float_and_identity_copy_use_the_tiled_kernels(hip):
    """Half-float RGB(A) outputs (avifRGBImageToF16 fused, src/reformat.c:1419-1443) and the 8-bit identity byte shuffle
    (src/reformat.c:1278-1309) are served by the bandwidth-tuned kernels, byte-exact."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for fmt, alpha, yf, up in ((1, True, 1, 3), (1, False, 3, 4), (0, False, 2, 3), (5, True, 3, 3), (2, True, 4, 3)):
            cases.append(H.Y2RCase(w, h, rgb_format=fmt, rgb_depth=16, is_float=True, alpha=alpha, yuv_depth=10, yuv_format=yf, upsampling=up,
                                   matrix=1, yuv_range=0, row_pad=6))
        cases.append(H.Y2RCase(w, h, rgb_format=1, rgb_depth=16, is_float=True, alpha=True, yuv_depth=12, yuv_format=3, matrix=9, yuv_range=1,
                               rgb_premultiplied=True))
        for fmt, alpha in ((1, False), (1, True), (0, False), (4, True), (5, False), (3, False)):
            cases.append(H.Y2RCase(w, h, rgb_format=fmt, rgb_depth=8, yuv_depth=8, yuv_format=1, matrix=0, yuv_range=1, alpha=alpha, row_pad=64))
        # ... and with the integer alpha (un)multiply that follows the copy for premultiplied lossless images (src/reformat.c:1574-1585; the
        # identity transform as arithmetic reproduces every code: round 5, found by --seed-rotation 1)
        for fmt, image_pm, rgb_pm in ((1, True, False), (1, False, True), (4, True, False), (2, False, True)):
            cases.append(H.Y2RCase(w, h, rgb_format=fmt, rgb_depth=8, yuv_depth=8, yuv_format=1, matrix=0, yuv_range=1, alpha=True, image_premultiplied=image_pm,
                                   rgb_premultiplied=rgb_pm, row_pad=64))
    for c in cases:
        H.run_y2r(H.HipDeviceBackend(), c)
        assert native.last_kernel().startswith("yuv2rgb_tile"), (c.ident(), native.last_kernel())
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), cases)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), cases)


def test_identity_matrix_at_any_depth_uses_the_tiled_kernels(hip):
    """GBR planes beyond the 8-bit full-range byte shuffle: limited range, 10 / 12 bits, other output depths (src/reformat.c:855-858)."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for yd, rd, rng, fmt, alpha, yf in ((8, 8, 0, 1, False, 1), (10, 10, 1, 1, True, 1), (12, 16, 1, 0, False, 1), (10, 8, 1, 4, True, 1), (8, 16, 1, 5, False, 1),
                                          (10, 16, 0, 1, False, 4)):
            cases.append(H.Y2RCase(w, h, yuv_depth=yd, rgb_depth=rd, yuv_range=rng, rgb_format=fmt, alpha=alpha, yuv_format=yf, matrix=0, row_pad=64))
    for c in cases:
        H.run_y2r(H.HipDeviceBackend(), c)
        assert native.last_kernel().startswith("yuv2rgb_tile"), (c.ident(), native.last_kernel())
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), cases)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), cases)


def test_gray_outputs_use_the_tiled_kernels(hip):
    """GRAY / GRAYA / AGRAY outputs (clamp01(Y) through the alpha multiply and the quantiser, src/reformat.c:886-961) read luma and alpha
    only and are served by the 4:0:0 instantiations of the bandwidth-tuned kernels, whatever the image's chroma layout."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for fmt, yf, yd, rd, alpha, prem, fl in ((7, 1, 8, 8, False, False, False), (7, 3, 10, 16, True, False, False), (8, 3, 8, 8, True, False, False),
                                                 (9, 2, 12, 12, True, True, False), (8, 4, 10, 8, False, False, False), (9, 1, 8, 16, True, False, False),
                                                 (8, 3, 10, 16, True, False, True), (7, 3, 8, 16, False, False, True)):
            cases.append(H.Y2RCase(w, h, rgb_format=fmt, rgb_depth=rd, yuv_depth=yd, yuv_format=yf, alpha=alpha, rgb_premultiplied=prem, is_float=fl,
                                   matrix=(1, 6, 9)[(w + fmt) % 3], yuv_range=(w + yd) % 2, upsampling=(3, 4)[(h + fmt) % 2]))
    for c in cases:
        H.run_y2r(H.HipDeviceBackend(), c)
        assert native.last_kernel().startswith("yuv2rgb_tile") and "gray" in native.last_kernel(), (c.ident(), native.last_kernel())
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), cases)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), cases)


def test_ignore_alpha_keeps_the_destination_alpha_in_the_tiled_kernels(hip, monkeypatch):
    """rgb->ignoreAlpha on a format that has an alpha channel, fp32 arithmetic: nothing writes the destination's alpha samples
    (src/reformat.c:1449-1450) -- whatever the buffer held stays, and the half-float pass still runs over it (:1419-1443).  The fp32 tiles
    read a pixel's alpha back from the destination and store it with the colours (TileArgs::alphaKeep): every 4-channel order, 8- and
    16-bit containers, gray + alpha, half floats, with and without an alpha plane in the image, every chroma layout."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for fmt, yf, yd, rd, alpha, fl, up in ((1, 3, 8, 8, True, False, 4), (2, 3, 8, 8, False, False, 4), (4, 2, 8, 8, True, False, 3), (5, 1, 8, 8, False, False, 4),
                                               (1, 3, 10, 10, True, False, 4), (2, 2, 12, 16, False, False, 4), (4, 1, 10, 16, True, False, 3), (5, 3, 8, 16, False, False, 4),
                                               (1, 3, 10, 8, True, False, 4), (8, 3, 8, 8, True, False, 4), (9, 1, 10, 16, False, False, 4), (8, 4, 12, 12, True, False, 3),
                                               (1, 3, 10, 16, True, True, 4), (2, 1, 8, 16, False, True, 4), (8, 3, 10, 16, True, True, 4)):
            cases.append(H.Y2RCase(w, h, rgb_format=fmt, rgb_depth=rd, yuv_depth=yd, yuv_format=yf, alpha=alpha, ignore_alpha=True, is_float=fl,
                                   image_premultiplied=alpha and (w + fmt) % 2 == 0, matrix=(1, 6, 9)[(w + fmt) % 3], yuv_range=(w + yd) % 2, upsampling=up))
    # a destination whose every byte differs from its neighbours': a pixel must get back ITS alpha, not its neighbour's
    plain_output = H.make_y2r_output

    def patterned_output(c):
        rgb = plain_output(c)
        flat = rgb.pixels.reshape(-1)
        flat[:] = (np.arange(flat.size, dtype=np.uint64) * 0x9E3779B1 >> 7).astype(np.uint8)
        return rgb

    monkeypatch.setattr(H, "make_y2r_output", patterned_output)
    for c in cases:
        H.run_y2r(H.HipDeviceBackend(), c)
        assert native.last_kernel().startswith("yuv2rgb_tile"), (c.ident(), native.last_kernel())
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), cases)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), cases)


def test_rgb565_from_the_fp32_arithmetic_uses_the_tiled_kernels(hip):
    """RGB565 where libyuv has no entry (10- / 12-bit planes -- an HDR image into an Android RGB_565 bitmap --, filtered chroma, 4:4:4, gray,
    avoidLibYUV): the fp32 tiles quantise to 8 bits and pack b >> 3 | (g >> 2) << 5 | (r >> 3) << 11 (src/reformat.c:619-626), the identity
    copy and the YCgCo family included; a pending alpha (un)multiply stays with the universal kernel."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for yf, yd, mc, rng, up in ((3, 8, 1, 0, 4), (3, 10, 9, 0, 4), (2, 12, 6, 1, 4), (3, 10, 1, 0, 3), (1, 8, 0, 1, 4), (1, 10, 0, 1, 4), (1, 8, 8, 1, 4),
                                    (4, 12, 1, 1, 4), (2, 8, 5, 0, 3), (1, 12, 9, 0, 4)):
            cases.append(H.Y2RCase(w, h, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565, rgb_depth=8, yuv_depth=yd, yuv_format=yf, matrix=mc, yuv_range=rng, upsampling=up))
    for c in cases:
        H.run_y2r(H.HipDeviceBackend(), c)
        assert native.last_kernel().startswith("yuv2rgb_tile") and "rgb565" in native.last_kernel(), (c.ident(), native.last_kernel())
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), cases)
    _compare_y2r(H.hip_host_backend(), H.oracle_backend(), cases)
    pending = H.Y2RCase(512, 16, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565, rgb_depth=8, yuv_depth=10, yuv_format=3, alpha=True, image_premultiplied=True)
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), [pending])


def test_yuv_to_rgb_tiled_sweep_device(hip):
    hip.avifhipSetTiledKernels(1)
    _compare_y2r(H.HipDeviceBackend(), H.oracle_backend(), H.y2r_sweep(TILED, n_random=300, seed=17), "yuv2rgb_tile")


def test_yuv_to_rgb_generic_kernels_on_tiled_sizes(hip):
    hip.avifhipSetTiledKernels(0)
    try:
        kernels = _compare_y2r(H.hip_host_backend(), H.oracle_backend(), H.y2r_sweep(TILED[:3], n_random=200, seed=31))
        assert kernels == {"yuv2rgb_generic"}
    finally:
        hip.avifhipSetTiledKernels(1)


def _compare_r2y(be, oracle, cases, padding=True):
    bad, kernels = [], {}
    for c in cases:
        ro, io = H.run_r2y(oracle, c)
        rh, ih = H.run_r2y(be, c)
        k = native.last_kernel().split("<")[0]
        kernels[k] = kernels.get(k, 0) + 1
        d = None if ro != rh else H.planes_equal(io, ih, padding=padding)
        if ro != rh or d:
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ro}/{rh} {d or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:25])
    return kernels


def test_rgb_to_yuv_tiled_sweep_host(hip):
    hip.avifhipSetTiledKernels(1)
    kernels = _compare_r2y(H.hip_host_backend(), H.oracle_backend(), H.r2y_sweep(TILED, n_random=500, seed=71))
    assert kernels.get("rgb2yuv_tile", 0) > 200, kernels
    assert "rgb2yuv_generic" in kernels  # unaligned buffers, divisors off the verified lists


def test_identity_encode_uses_the_tiled_kernels(hip):
    """Lossless encodes (identity matrix: GBR planes, src/reformat.c:361-365) in the bandwidth-tuned encode kernels, full and limited range."""
    hip.avifhipSetTiledKernels(1)
    cases = []
    for (w, h) in TILED:
        for fmt, rd, yd, rng, yf in ((1, 8, 8, 1, 1), (0, 8, 8, 1, 1), (4, 8, 8, 0, 1), (1, 16, 10, 1, 1), (0, 16, 12, 1, 1), (1, 8, 8, 1, 4), (1, 10, 10, 1, 1)):
            cases.append(H.R2YCase(w, h, rgb_depth=rd, rgb_format=fmt, yuv_depth=yd, yuv_format=yf, yuv_range=rng, matrix=0))
    kernels = _compare_r2y(H.HipDeviceBackend(), H.oracle_backend(), cases, padding=False)
    assert set(kernels) == {"rgb2yuv_tile"}, kernels
    _compare_r2y(H.hip_host_backend(), H.oracle_backend(), cases)


def test_rgb_to_yuv_tiled_sweep_device(hip):
    hip.avifhipSetTiledKernels(1)
    kernels = _compare_r2y(H.HipDeviceBackend(), H.oracle_backend(), H.r2y_sweep(TILED, n_random=300, seed=73), padding=False)
    assert kernels.get("rgb2yuv_tile", 0) > 100, kernels


def test_rgb_to_yuv_tiles_at_every_strip_count(hip, monkeypatch):
    """The encode tiles take 1, 2 or 4 strips per wave by image size (kernels_r2y_tile.hip: 2 from 8K frames up) and per-XCD bands or raster
    order: each forced on the sweep's small images (AVIFHIP_R2Y_SPW, plan.h TUNE_R2Y_RASTER), both arithmetics' kernels -- the encode-side
    twin of test_fp32_tiles_at_every_launch_geometry."""
    from dataclasses import replace
    hip.avifhipSetTiledKernels(1)
    cases = H.r2y_sweep([(777, 70), (512, 64), (1027, 35)], n_random=120, seed=83)
    try:
        for spw in ("1", "2", "4"):
            monkeypatch.setenv("AVIFHIP_R2Y_SPW", spw)
            for tuning in (0x1, 0x81):
                hip.avifhipSetTuning(tuning)
                kernels = _compare_r2y(H.HipDeviceBackend(), H.oracle_backend(), cases, padding=False)
                assert kernels.get("rgb2yuv_tile", 0) > 40, (spw, tuning, kernels)
        hip.avifhipSetArithmetic(0)  # libyuv's fixed-point encode kernel (BT.601) under the same knobs
        fx = [replace(c, avoid_libyuv=False) for c in H.libyuv_r2y_cases([(777, 70), (512, 64)], n_random=80, seed=89)]
        for spw in ("1", "2", "4"):
            monkeypatch.setenv("AVIFHIP_R2Y_SPW", spw)
            _compare_r2y(H.HipDeviceBackend(), H.oracle_libyuv_backend(), fx, padding=False)
    finally:
        hip.avifhipSetTuning(1)
        hip.avifhipSetArithmetic(1)


def test_rgb_to_yuv_generic_kernels_on_tiled_sizes(hip):
    hip.avifhipSetTiledKernels(0)
    try:
        kernels = _compare_r2y(H.hip_host_backend(), H.oracle_backend(), H.r2y_sweep(TILED[:3], n_random=150, seed=79))
        assert set(kernels) == {"rgb2yuv_generic"}
    finally:
        hip.avifhipSetTiledKernels(1)


def test_rgb_to_yuv_sweep_host(hip):
    _compare_r2y(H.hip_host_backend(), H.oracle_backend(), H.r2y_sweep(SMALL + [(300, 21), (512, 16)], n_random=500))


def test_rgb_to_yuv_sweep_device(hip):
    # device planes have their own row pitch: the gray path's whole-row chroma fill (src/reformat.c:520-542) lands in
    # the device rows' padding, which the test does not copy back -- compare the samples only
    _compare_r2y(H.HipDeviceBackend(), H.oracle_backend(), H.r2y_sweep(SMALL + [(300, 21)], n_random=200, seed=3), padding=False)


@pytest.mark.parametrize("depth", [8, 10, 12, 16])
@pytest.mark.parametrize("fmt", [1, 2, 4, 5, 8, 9])
def test_premultiply_unpremultiply(hip, fmt, depth):
    from libavif_amd import synth

    o = H.oracle_backend()
    for be in (H.hip_host_backend(), H.HipDeviceBackend()):
        for which in ("premultiply", "unpremultiply"):
            a = abi.make_rgb(261, 19, depth, fmt, row_pad=6, fill=0x5A)
            synth.fill_rgb(a, 0xBEEF + fmt + depth)
            if depth in (10, 12):
                a.pixels.view(np.uint16)[...] &= (1 << depth) - 1
            b = abi.make_rgb(261, 19, depth, fmt, row_pad=6)
            b.pixels[...] = a.pixels
            if isinstance(be, H.HipDeviceBackend):
                be.bind_host(b.struct, b)
            assert getattr(o, which)(a.struct) == getattr(be, which)(b.struct) == 0
            # the device route stages tight rows: compare the pixel bytes, the host route must keep padding as is
            wb = a.struct.width * abi.rgb_pixel_size(fmt, depth)
            assert np.array_equal(a.pixels[:, :wb], b.pixels[:, :wb]), (be.name, which, H.describe_diff(a.pixels[:, :wb], b.pixels[:, :wb]))
            if be.name == "hip-host":
                assert np.array_equal(a.pixels, b.pixels)


def test_exhaustive_alpha_pairs_8bit(hip):
    o, be = H.oracle_backend(), H.hip_host_backend()
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(256, 256, 8, abi.AVIF_RGB_FORMAT_RGBA)
        ch = a.channels()
        ch[:, :, 0] = np.arange(256)[None, :]
        ch[:, :, 1] = 255 - np.arange(256)[None, :]
        ch[:, :, 2] = (np.arange(256)[None, :] * 7) % 256
        ch[:, :, 3] = np.arange(256)[:, None]
        b = abi.make_rgb(256, 256, 8, abi.AVIF_RGB_FORMAT_RGBA)
        b.pixels[...] = a.pixels
        assert getattr(o, which)(a.struct) == getattr(be, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), which


@pytest.mark.parametrize("depth,fmt", [(10, abi.AVIF_RGB_FORMAT_RGBA), (10, abi.AVIF_RGB_FORMAT_ARGB), (12, abi.AVIF_RGB_FORMAT_BGRA)])
def test_exhaustive_alpha_pairs_10_and_12_bit(hip, depth, fmt):
    """Every (colour, alpha) pair of a 10- / 12-bit channel, both directions: the un-premultiply direction runs on integers
    (pixel_math.h: unpremulRcp, the shared exact reciprocal), which has to equal the reference's float expression everywhere."""
    o, be = H.oracle_backend(), H.HipDeviceBackend()
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
        be.bind_host(b.struct, b)
        assert getattr(o, which)(a.struct) == getattr(be, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), (which, H.describe_diff(a.pixels, b.pixels))


def test_error_codes(hip):
    """Same error-code matrix as the reference (tests/gtest/avif_fuzztest_yuvrgb.cc:36-46, src/alpha.c:154-161,341-348)."""
    o, be = H.oracle_backend(), H.hip_host_backend()
    bad_cases = [
        H.Y2RCase(8, 8, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565, rgb_depth=10), H.Y2RCase(8, 8, rgb_depth=8, is_float=True),
        H.Y2RCase(8, 8, matrix=3), H.Y2RCase(8, 8, matrix=8, yuv_range=abi.AVIF_RANGE_LIMITED),
        H.Y2RCase(8, 8, matrix=0, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=1), H.Y2RCase(8, 8, matrix=10),
        H.Y2RCase(8, 8, matrix=18), H.Y2RCase(8, 8, matrix=16, yuv_depth=10, rgb_depth=10, yuv_range=1), H.Y2RCase(8, 8, rgb_depth=9),
    ]
    for c in bad_cases:
        ro, _ = H.run_y2r(o, c)
        rh, ph = H.run_y2r(be, c)
        assert ro == rh == abi.AVIF_RESULT_REFORMAT_FAILED, c.ident()
        assert (ph == H.FILL_BYTE).all()
    rgb = abi.make_rgb(4, 4, 8, abi.AVIF_RGB_FORMAT_RGB)
    assert be.premultiply(rgb.struct) == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert be.unpremultiply(rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
    c = H.R2YCase(8, 8)
    rgbi, out = H.make_r2y_inputs(c), H.make_r2y_output(c)
    rgbi.struct.format = abi.AVIF_RGB_FORMAT_RGB_565
    assert be.rgb_to_yuv(out.struct, rgbi.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
    rgbi.struct.format, rgbi.struct.depth, rgbi.struct.isFloat = abi.AVIF_RGB_FORMAT_RGBA, 16, 1
    assert be.rgb_to_yuv(out.struct, rgbi.struct) == abi.AVIF_RESULT_NOT_IMPLEMENTED


def test_rgb_to_yuv_allocates_missing_planes(hip):
    """avifImageRGBToYUV allocates planes that are NULL (src/reformat.c:236-240, src/avif.c:431-490)."""
    import ctypes as C

    c = H.R2YCase(33, 9, rgb_format=abi.AVIF_RGB_FORMAT_RGBA)
    rgb = H.make_r2y_inputs(c)
    want = H.make_r2y_output(c)
    assert H.oracle_backend().rgb_to_yuv(want.struct, rgb.struct) == 0
    img = abi.make_yuv(33, 9, 8, c.yuv_format, c.yuv_range, c.matrix, allocate=False)
    assert hip.avifhipImageRGBToYUV(img.struct, rgb.struct) == 0
    assert img.struct.imageOwnsYUVPlanes and img.struct.imageOwnsAlphaPlane and img.struct.alphaPlane
    y = np.ctypeslib.as_array(C.cast(img.struct.yuvPlanes[0], C.POINTER(C.c_uint8)), shape=(9, img.struct.yuvRowBytes[0]))
    assert img.struct.yuvRowBytes[0] == 33
    assert np.array_equal(y, want.planes[0][:, :33])
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for p in range(3):
        libc.free(img.struct.yuvPlanes[p])
    libc.free(img.struct.alphaPlane)
