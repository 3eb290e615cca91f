"""Crop + rotate + mirror of the converted RGB image (avifApplyTransforms, apps/shared/avifutil.c:787-825).

CPU: the oracle's restatement (oracleRGBImageTransform) against the reference's own avifRGBImageSetViewRect /
avifRGBImageRotate / avifRGBImageMirror, compiled from apps/shared/avifutil.c (oracle/_ref/libavifutil_ref.so), called in
the order avifApplyTransforms calls them.  GPU (-m gpu): avifhipRGBImageTransformAsync, one pass, against the oracle."""
import ctypes as C
import itertools

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi, synth

A = abi
FORMATS = [(A.AVIF_RGB_FORMAT_RGBA, 8), (A.AVIF_RGB_FORMAT_RGB, 8), (A.AVIF_RGB_FORMAT_RGBA, 16), (A.AVIF_RGB_FORMAT_BGR, 12), (A.AVIF_RGB_FORMAT_RGB_565, 8),
           (A.AVIF_RGB_FORMAT_GRAY, 8), (A.AVIF_RGB_FORMAT_GRAYA, 16)]
SIZES = [(67, 41), (128, 64), (33, 100), (1, 7), (300, 37)]


def combos():
    out = []
    for (fmt, depth), (w, h), angle, mirror in itertools.product(FORMATS, SIZES, (None, 0, 1, 2, 3), (None, 0, 1)):
        crop = None if (w * h) % 3 == 0 else (w // 5, h // 7, max(1, w - w // 3), max(1, h - h // 2))
        out.append((fmt, depth, w, h, crop, angle, mirror))
    return out


def source(fmt, depth, w, h, row_pad=0):
    rgb = abi.make_rgb(w, h, depth, fmt, row_pad=row_pad, fill=0x5A)
    synth.fill_rgb(rgb, 0xC0FFEE + w * 131 + h)
    return rgb


def out_dims(w, h, crop, angle):
    cw, ch = (crop[2], crop[3]) if crop else (w, h)
    return (ch, cw) if angle in (1, 3) else (cw, ch)


def call_oracle(src, dst, crop, angle, mirror):
    o = oracle_lib.oracle()
    rect = abi.avifCropRect(*crop) if crop else None
    return o.oracleRGBImageTransform(dst.struct, src.struct, C.byref(rect) if rect else None, int(angle is not None), angle or 0, int(mirror is not None), mirror or 0)


@pytest.mark.skipif(oracle_lib.util_ref() is None, reason="oracle/_ref/libavifutil_ref.so not built (needs /root/reference)")
def test_oracle_equals_the_references_helpers():
    u, ref = oracle_lib.util_ref(), oracle_lib.ref()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for fmt, depth, w, h, crop, angle, mirror in combos():
        src = source(fmt, depth, w, h, row_pad=6)
        dw, dh = out_dims(w, h, crop, angle)
        got = abi.make_rgb(dw, dh, depth, fmt, fill=0x11)
        assert call_oracle(src, got, crop, angle, mirror) == 0
        # the reference, step by step as in avifApplyTransforms (:793-824)
        work = source(fmt, depth, w, h, row_pad=6)  # mirroring is in place: work on a copy
        view = abi.avifRGBImage()
        C.memmove(C.byref(view), C.byref(work.struct), C.sizeof(view))
        if crop:
            rect = abi.avifCropRect(*crop)
            u.avifRGBImageSetViewRect(C.byref(view), C.byref(work.struct), C.byref(rect))
        rotated = None
        if angle:  # "irot.angle != 0"
            rotated = abi.avifRGBImage()
            ang = C.c_uint8(angle)
            assert u.avifRGBImageRotate(C.byref(rotated), C.byref(view), C.byref(ang)) == 0
            view = rotated
        if mirror is not None:
            ax = C.c_uint8(mirror)
            assert u.avifRGBImageMirror(C.byref(view), C.byref(ax)) == 0
        px = abi.rgb_pixel_size(fmt, depth)
        assert (view.width, view.height) == (dw, dh)
        want = np.ctypeslib.as_array(C.cast(view.pixels, C.POINTER(C.c_uint8)), shape=(dh, view.rowBytes))[:, : dw * px]
        assert np.array_equal(got.pixels[:, : dw * px], want), (fmt, depth, w, h, crop, angle, mirror)
        if rotated is not None:
            libc.free(rotated.pixels)  # avifRGBImageAllocatePixels -> avifAlloc -> malloc (src/mem.c)


@pytest.mark.gpu
def test_gpu_transform_equals_the_oracle(hip):
    from libavif_amd import device, native

    kernels = set()
    for fmt, depth, w, h, crop, angle, mirror in combos() + [(A.AVIF_RGB_FORMAT_RGBA, 8, 1030, 517, (6, 2, 1001, 500), a, m) for a in (None, 1, 2, 3) for m in (None, 0, 1)]:
        src = source(fmt, depth, w, h)
        dw, dh = out_dims(w, h, crop, angle)
        want = abi.make_rgb(dw, dh, depth, fmt, fill=0x11)
        assert call_oracle(src, want, crop, angle, mirror) == 0
        got = abi.make_rgb(dw, dh, depth, fmt, fill=0x22)
        dsrc, ddst = device.DeviceRGB(src, upload=True, tight=(w % 2 == 1)), device.DeviceRGB(got, upload=True, tight=(h % 2 == 1))
        rect = abi.avifCropRect(*crop) if crop else None
        native.check(hip.avifhipRGBImageTransformAsync(ddst.struct, dsrc.struct, C.byref(rect) if rect else None, int(angle is not None), angle or 0,
                                                       int(mirror is not None), mirror or 0, None), "avifhipRGBImageTransformAsync")
        native.check(hip.avifhipSynchronize(None), "sync")
        kernels.add(native.last_kernel())
        ddst.download_into_host()
        px = abi.rgb_pixel_size(fmt, depth)
        assert np.array_equal(got.pixels[:, : dw * px], want.pixels[:, : dw * px]), (fmt, depth, w, h, crop, angle, mirror, native.last_kernel())
    assert kernels == {"rgb_transform_rows", "rgb_transform_transpose"}
    # argument errors, apps/shared/avifutil.c:741,781
    src = source(A.AVIF_RGB_FORMAT_RGBA, 8, 16, 16)
    d = device.DeviceRGB(src, upload=True)
    assert hip.avifhipRGBImageTransformAsync(d.struct, d.struct, None, 1, 4, 0, 0, None) == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert hip.avifhipRGBImageTransformAsync(d.struct, d.struct, None, 0, 0, 1, 2, None) == abi.AVIF_RESULT_INVALID_ARGUMENT
