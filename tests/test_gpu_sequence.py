"""Image sequences through avifhipImageYUVToRGBBatchAsync (round 6): frames that differ in their buffers only run through the single-image
kernels, up to 8 per launch, the frames' addresses in the kernel arguments (api_batch.cpp sequenceAsync, kernels_tile.hip
launchYuvToRgbTileSequence) -- no descriptor table is uploaded.  Every frame must equal the oracle's conversion of that frame
(src/reformat.c:1625-1747 is stateless: a sequence is N independent avifImageYUVToRGB calls), leftover columns / rows included; batches that
are not sequences (small tiles, mixed sizes, kernel families without sequence kernels) still take the table path and still match."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, device, native

pytestmark = pytest.mark.gpu

# >= 2 megapixels each (the sequence rule), sizes that leave 1-3 columns and a row to the universal kernel
SEQUENCES = [
    # the headline's layout in both arithmetics: 8-bit 4:2:0 limited BT.709 -> RGBA8, bilinear
    H.Y2RCase(2051, 1031, yuv_format=3, yuv_range=0, matrix=1, upsampling=4),
    H.Y2RCase(2048, 1032, yuv_format=3, yuv_range=0, matrix=1, upsampling=3, rgb_format=abi.AVIF_RGB_FORMAT_BGR),
    H.Y2RCase(2304, 912, yuv_format=2, yuv_range=1, matrix=6, upsampling=4, rgb_format=abi.AVIF_RGB_FORMAT_BGRA, alpha=True),
    H.Y2RCase(2050, 1030, yuv_depth=10, yuv_format=1, yuv_range=0, matrix=9, rgb_depth=16, alpha=True),
    H.Y2RCase(2052, 1026, yuv_depth=12, yuv_format=3, yuv_range=0, matrix=1, rgb_depth=8, upsampling=4),
    H.Y2RCase(2056, 1024, yuv_format=4, yuv_range=1, matrix=6, rgb_format=abi.AVIF_RGB_FORMAT_RGB),
    # 6.3 megapixels: the size from which the fp32 arithmetic filters 8-bit chroma in its wave-private kernels
    H.Y2RCase(3842, 1642, yuv_format=3, yuv_range=0, matrix=1, upsampling=4),
]


def _run_batch(hip, cases, stream=None):
    hosts = [(H.make_y2r_inputs(c), H.make_y2r_output(c)) for c in cases]
    devs = [(device.DeviceYUV(i), device.DeviceRGB(o, upload=True)) for i, o in hosts]
    n = len(cases)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(d[0].struct) for d in devs])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(d[1].struct) for d in devs])
    uploads0 = hip.avifhipTableUploadCount()
    native.check(hip.avifhipImageYUVToRGBBatchAsync(n, imgs, rgbs, None, stream), "avifhipImageYUVToRGBBatchAsync")
    native.check(hip.avifhipSynchronize(stream), "sync")
    kernel = native.last_kernel()
    for d in devs:
        d[1].download_into_host()
    return [o for _, o in hosts], hip.avifhipTableUploadCount() - uploads0, kernel


def _check(cases, outs, oracle, what):
    for k, (c, out) in enumerate(zip(cases, outs)):
        res, want = H.run_y2r(oracle, c)
        assert res == 0
        wb = c.w * abi.rgb_pixel_size(c.rgb_format, c.rgb_depth)
        assert np.array_equal(out.pixels[:, :wb], want[:, :wb]), (what, k, c.ident(), H.describe_diff(want[:, :wb], out.pixels[:, :wb]))
        if out.pixels.shape[1] > wb:
            assert (out.pixels[:, wb:] == H.FILL_BYTE).all(), (what, k, "row padding written")


@pytest.mark.parametrize("frames", [1, 2, 3, 9])
@pytest.mark.parametrize("arith", [1, 0], ids=["fp32", "integer"])
def test_sequences_equal_frame_by_frame_conversions(hip, arith, frames):
    oracle = H.oracle_backend() if arith == 1 else H.oracle_libyuv_backend()
    hip.avifhipSetArithmetic(arith)
    try:
        for base in SEQUENCES if frames <= 3 else SEQUENCES[:2]:
            cases = [dataclasses.replace(base, avoid_libyuv=(arith == 1), seed=0x1234 + 977 * k) for k in range(frames)]
            outs, uploads, kernel = _run_batch(hip, cases)
            _check(cases, outs, oracle, ("sequence", arith, frames, kernel))
            # the packed integer kernels and the wave-private fp32 kernels have sequence kernels: nothing was uploaded for them
            # (kernels_tile.hip soloPays: unfiltered layouts always, filtered 8-bit planes from 6 megapixels up)
            packed = ",pk16" in kernel
            solo_fp32 = kernel.startswith("yuv2rgb_tile<") and ("nearest" in kernel or ("<u8" in kernel and base.w * base.h >= 6 << 20))
            assert uploads == (0 if packed or solo_fp32 else 1), (kernel, uploads)
    finally:
        hip.avifhipSetArithmetic(1)


def test_sequences_on_a_caller_stream_and_below_the_size_rule(hip):
    """The same frames on a caller's stream; and the table path for them (what AVIFHIP_SEQUENCE=0 selects is read once per process, so the
    table path is reached here the way a caller reaches it: frames below the size rule)."""
    base = SEQUENCES[0]
    cases = [dataclasses.replace(base, avoid_libyuv=False, seed=77 + k) for k in range(3)]
    hip.avifhipSetArithmetic(0)
    try:
        stream = hip.avifhipStreamCreate()
        assert stream
        try:
            outs, uploads, kernel = _run_batch(hip, cases, stream)
            assert uploads == 0 and ",pk16" in kernel
            _check(cases, outs, H.oracle_libyuv_backend(), "caller stream")
        finally:
            hip.avifhipStreamDestroy(stream)
        small = [dataclasses.replace(c, w=1027, h=515) for c in cases]
        outs, uploads, kernel = _run_batch(hip, small)
        assert uploads == 1, (kernel, uploads)
        _check(small, outs, H.oracle_libyuv_backend(), "table batch")
    finally:
        hip.avifhipSetArithmetic(1)


def test_batches_that_are_not_sequences_keep_the_table_path(hip):
    """Frames of different sizes, or of different layouts, are not a sequence: one table batch (or the universal batch kernel) as before."""
    hip.avifhipSetArithmetic(1)
    a = dataclasses.replace(SEQUENCES[0], seed=5)
    b = dataclasses.replace(SEQUENCES[0], w=2064, h=1040, seed=6)
    outs, uploads, kernel = _run_batch(hip, [a, b])
    assert uploads == 1
    _check([a, b], outs, H.oracle_backend(), ("mixed sizes", kernel))
    # premultiplied output of the integer path: the attenuate kernels have no sequence variant -- refused before anything is launched
    hip.avifhipSetArithmetic(0)
    try:
        c = [dataclasses.replace(SEQUENCES[2], avoid_libyuv=False, rgb_premultiplied=True, seed=9 + k) for k in range(2)]
        outs, uploads, kernel = _run_batch(hip, c)
        assert uploads == 1, kernel
        _check(c, outs, H.oracle_libyuv_backend(), ("attenuate", kernel))
    finally:
        hip.avifhipSetArithmetic(1)


# ---- the encode direction: avifhipImageRGBToYUVBatchAsync (src/reformat.c:161-519 per frame) ----

ENCODE_SEQUENCES = [
    # cfg4's layout: RGBA8 -> 8-bit 4:2:0 BT.709 limited + alpha plane, both arithmetics (BT.601 is the one libyuv serves)
    H.R2YCase(2051, 1031, yuv_format=3, yuv_range=0, matrix=1),
    H.R2YCase(2052, 1030, yuv_format=3, yuv_range=0, matrix=6),
    H.R2YCase(2048, 1026, rgb_format=abi.AVIF_RGB_FORMAT_BGR, yuv_format=2, yuv_range=1, matrix=6),
    H.R2YCase(2050, 1028, rgb_depth=16, yuv_depth=12, yuv_format=1, yuv_range=1, matrix=9, rgb_premultiplied=False),
    H.R2YCase(2304, 914, rgb_depth=8, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=16),   # YCgCo-Re: one of the rare modes
    H.R2YCase(2056, 1024, yuv_format=1, yuv_range=1, matrix=0),                               # identity (lossless)
]


def _run_encode_batch(hip, cases, stream=None):
    hosts = [(H.make_r2y_output(c), H.make_r2y_inputs(c)) for c in cases]
    devs = [(device.DeviceYUV(i, upload=True), device.DeviceRGB(o, upload=True)) for i, o in hosts]
    n = len(cases)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(d[0].struct) for d in devs])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(d[1].struct) for d in devs])
    launches0 = hip.avifhipLaunchCount()
    native.check(hip.avifhipImageRGBToYUVBatchAsync(n, imgs, rgbs, stream), "avifhipImageRGBToYUVBatchAsync")
    native.check(hip.avifhipSynchronize(stream), "sync")
    kernel = native.last_kernel()
    for d in devs:
        d[0].download_into_host()
    return [i for i, _ in hosts], hip.avifhipLaunchCount() - launches0, kernel


@pytest.mark.parametrize("frames", [1, 2, 3, 9])
@pytest.mark.parametrize("arith", [1, 0], ids=["fp32", "integer"])
def test_encode_sequences_equal_frame_by_frame_conversions(hip, arith, frames):
    oracle = H.oracle_backend() if arith == 1 else H.oracle_libyuv_backend()
    hip.avifhipSetArithmetic(arith)
    try:
        for base in ENCODE_SEQUENCES if frames <= 3 else ENCODE_SEQUENCES[:2]:
            cases = [dataclasses.replace(base, avoid_libyuv=(arith == 1), seed=0x4321 + 613 * k) for k in range(frames)]
            outs, launches, kernel = _run_encode_batch(hip, cases)
            for k, (c, got) in enumerate(zip(cases, outs)):
                res, want = H.run_r2y(oracle, c)
                assert res == 0
                assert H.planes_equal(want, got, padding=False) is None, (arith, frames, k, c.ident(), kernel, H.planes_equal(want, got, padding=False))
            # one launch per 8 frames wherever the tiled kernels serve the layout (all but what the universal kernel keeps)
            assert launches == ((frames + 7) // 8 if "tile" in kernel else frames), (kernel, launches)
            if base is ENCODE_SEQUENCES[0] or base is ENCODE_SEQUENCES[1]:
                assert "tile" in kernel, kernel
    finally:
        hip.avifhipSetArithmetic(1)


def test_encode_batches_that_are_not_sequences_run_frame_by_frame(hip):
    hip.avifhipSetArithmetic(1)
    mixed = [dataclasses.replace(ENCODE_SEQUENCES[0], seed=3), dataclasses.replace(ENCODE_SEQUENCES[0], w=1030, h=514, seed=4),
             dataclasses.replace(ENCODE_SEQUENCES[2], seed=5)]
    outs, launches, kernel = _run_encode_batch(hip, mixed)
    assert launches == 3
    for c, got in zip(mixed, outs):
        res, want = H.run_r2y(H.oracle_backend(), c)
        assert res == 0 and H.planes_equal(want, got, padding=False) is None, (c.ident(), kernel)
    # a missing destination plane is the caller's error, like in the single-frame entry point
    img, rgb = H.make_r2y_output(ENCODE_SEQUENCES[0]), H.make_r2y_inputs(ENCODE_SEQUENCES[0])
    dimg, drgb = device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)
    dimg.struct.yuvPlanes[1] = None
    imgs = (C.POINTER(abi.avifImage) * 1)(C.pointer(dimg.struct))
    rgbs = (C.POINTER(abi.avifRGBImage) * 1)(C.pointer(drgb.struct))
    assert hip.avifhipImageRGBToYUVBatchAsync(1, imgs, rgbs, None) == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert hip.avifhipImageRGBToYUVBatchAsync(0, None, None, None) == abi.AVIF_RESULT_OK
