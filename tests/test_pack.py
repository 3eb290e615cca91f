"""Row packing for the file writers next to the reformat path (second half): the Y4M frame payload of
apps/shared/y4m.c:603-618 and the PNG row data of apps/shared/avifpng.c:865-880 (png_set_swap).
  CPU: the oracle's Y4M frame equals the payload of the file the reference's own y4mWrite writes (oracle/_ref/libavifutil_ref.so);
       the oracle's PNG rows equal numpy's byteswap of the tight rows (png_set_swap's definition).
  GPU: avifhipImagePackY4MFrameAsync / avifhipRGBImagePackPNGRowsAsync equal the oracle byte for byte, aligned and unaligned."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
from libavif_amd import abi, synth

SIZES = [(64, 16), (300, 21), (37, 9), (1027, 18), (256, 2), (1, 1), (130, 5)]


def _yuv_cases():
    for (w, h) in SIZES:
        for depth in (8, 10, 12):
            for fmt in (abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_PIXEL_FORMAT_YUV422, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_PIXEL_FORMAT_YUV400):
                for pad in (0, 6):
                    alpha = depth == 8 and fmt == abi.AVIF_PIXEL_FORMAT_YUV444 and pad == 6
                    yield w, h, depth, fmt, pad, alpha


def _make_yuv(w, h, depth, fmt, pad, alpha, seed):
    img = abi.make_yuv(w, h, depth, fmt, abi.AVIF_RANGE_FULL, 1, with_alpha=alpha, row_pad=pad)
    for buf in img.planes + [img.alpha]:
        if buf is not None:
            buf[...] = 0xA5
    synth.fill_yuv(img, seed)
    return img


def _oracle_frame(img, alpha):
    o = oracle_lib.oracle()
    n = o.oraclePackY4MFrame(img.struct, int(alpha), None)
    out = np.zeros(n, dtype=np.uint8)
    assert o.oraclePackY4MFrame(img.struct, int(alpha), out.ctypes.data) == n
    return out


@pytest.mark.skipif(oracle_lib.util_ref() is None or not hasattr(oracle_lib.util_ref(), "y4mWrite"), reason="oracle/_ref/libavifutil_ref.so (with y4m.c) not built")
def test_oracle_y4m_frame_equals_the_reference_writer(tmp_path):
    ref = oracle_lib.util_ref()
    for k, (w, h, depth, fmt, pad, alpha) in enumerate(_yuv_cases()):
        img = _make_yuv(w, h, depth, fmt, pad, alpha, 0x1000 + k)
        path = tmp_path / f"f{k}.y4m"
        assert ref.y4mWrite(os.fspath(path).encode(), img.struct) == 1
        data = path.read_bytes()
        marker = data.index(b"\nFRAME\n") + len(b"\nFRAME\n")
        payload = np.frombuffer(data[marker:], dtype=np.uint8)
        want = _oracle_frame(img, alpha)
        assert payload.size == want.size and np.array_equal(payload, want), (w, h, depth, fmt, pad, alpha)


def _rgb_cases():
    for (w, h) in SIZES:
        for depth in (8, 16):
            for fmt in (abi.AVIF_RGB_FORMAT_RGB, abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_GRAY, abi.AVIF_RGB_FORMAT_GRAYA):
                for pad in (0, 10):
                    yield w, h, depth, fmt, pad


def _make_rgb(w, h, depth, fmt, pad, seed):
    rgb = abi.make_rgb(w, h, depth, fmt, row_pad=pad, fill=0x3C)
    rng = np.random.default_rng(seed)
    rgb.pixels[...] = rng.integers(0, 256, size=rgb.pixels.shape, dtype=np.uint8)
    return rgb


def test_oracle_png_rows_are_the_byteswapped_tight_rows():
    o = oracle_lib.oracle()
    for k, (w, h, depth, fmt, pad) in enumerate(_rgb_cases()):
        rgb = _make_rgb(w, h, depth, fmt, pad, k)
        px = abi.rgb_pixel_size(fmt, depth)
        out = np.zeros(w * px * h, dtype=np.uint8)
        assert o.oraclePackPNGRows(rgb.struct, px, out.ctypes.data) == out.size
        tight = np.ascontiguousarray(rgb.pixels[:, : w * px])
        want = tight.view("<u2").byteswap().view(np.uint8).reshape(-1) if depth > 8 else tight.reshape(-1)
        assert np.array_equal(out, want), (w, h, depth, fmt, pad)


@pytest.mark.gpu
def test_gpu_y4m_frame(hip):
    from libavif_amd import device, native

    for k, (w, h, depth, fmt, pad, alpha) in enumerate(_yuv_cases()):
        img = _make_yuv(w, h, depth, fmt, pad, alpha, 0x2000 + k)
        want = _oracle_frame(img, alpha)
        for tight in (False, True):  # device planes with 256-byte pitches, or with the tight (often unaligned) pitches
            dimg = device.DeviceYUV(img, tight=tight)
            n = hip.avifhipY4MFrameBytes(dimg.struct, int(alpha))
            assert n == want.size
            buf = device.DeviceBuffer(n + 64)
            buf.memset(0xEE)
            native.check(hip.avifhipImagePackY4MFrameAsync(dimg.struct, int(alpha), buf.ptr, None), "pack y4m")
            native.check(hip.avifhipSynchronize(None), "sync")
            got = buf.download()
            bad = np.flatnonzero(got[:n] != want)
            assert bad.size == 0, (w, h, depth, fmt, pad, alpha, tight, "first mismatches at", bad[:8].tolist(), "of", bad.size, got[bad[:8]].tolist(), want[bad[:8]].tolist())
            assert (got[n:] == 0xEE).all()
    # error codes of the writer: alpha only for 8-bit 4:4:4, depths 8 / 10 / 12 only
    img = _make_yuv(64, 16, 10, abi.AVIF_PIXEL_FORMAT_YUV444, 0, True, 1)
    dimg = device.DeviceYUV(img)
    buf = device.DeviceBuffer(1 << 16)
    assert hip.avifhipImagePackY4MFrameAsync(dimg.struct, 1, buf.ptr, None) == abi.AVIF_RESULT_NOT_IMPLEMENTED
    assert hip.avifhipY4MFrameBytes(dimg.struct, 1) == 0  # ... and the size query agrees: no such frame
    assert hip.avifhipY4MFrameBytes(dimg.struct, 0) == 64 * 16 * 2 * 3
    assert hip.avifhipImagePackY4MFrameAsync(None, 0, buf.ptr, None) == abi.AVIF_RESULT_INVALID_ARGUMENT


@pytest.mark.gpu
def test_gpu_png_rows(hip):
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    for k, (w, h, depth, fmt, pad) in enumerate(_rgb_cases()):
        rgb = _make_rgb(w, h, depth, fmt, pad, 77 + k)
        px = abi.rgb_pixel_size(fmt, depth)
        want = np.zeros(w * px * h, dtype=np.uint8)
        o.oraclePackPNGRows(rgb.struct, px, want.ctypes.data)
        drgb = device.DeviceRGB(rgb, upload=True)
        buf = device.DeviceBuffer(want.size + 64)
        buf.memset(0xEE)
        native.check(hip.avifhipRGBImagePackPNGRowsAsync(drgb.struct, buf.ptr, None), "pack png")
        native.check(hip.avifhipSynchronize(None), "sync")
        got = buf.download()
        assert np.array_equal(got[: want.size], want), (w, h, depth, fmt, pad)
        assert (got[want.size:] == 0xEE).all()
    # the 16-byte fast path at full size: 8K RGBA16
    rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA)
    rgb.pixels[...] = np.random.default_rng(5).integers(0, 256, size=rgb.pixels.shape, dtype=np.uint8)
    drgb = device.DeviceRGB(rgb, upload=True)
    buf = device.DeviceBuffer(rgb.pixels.size)
    native.check(hip.avifhipRGBImagePackPNGRowsAsync(drgb.struct, buf.ptr, None), "pack png")
    native.check(hip.avifhipSynchronize(None), "sync")
    assert np.array_equal(buf.download(), rgb.pixels.reshape(-1).view("<u2").byteswap().view(np.uint8))
