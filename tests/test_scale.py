"""Plane scaling (avifImageScale, src/scale.c:23-201 over the vendored libyuv scaler).

CPU: the oracle's per-sample restatement (oracle/scale_oracle.c) against avifImageScale of the reference compiled from its
own sources -- every plane of every case byte-identical.  GPU (-m gpu): avifhipImageScale (host image, in place, like the
reference) and avifhipImageScaleAsync (device-resident) against the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi

SIZES = [(64, 48), (33, 17), (1, 1), (5, 9), (100, 3), (2, 2), (17, 64), (128, 128), (7, 1), (1, 7), (255, 31)]
RATIOS = [0.1, 0.25, 0.3, 1 / 3, 0.5, 0.6, 1, 1.5, 2, 3, 4.7]
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def scale_cases(n, seed):
    rnd = random.Random(seed)
    out = [(H.Y2RCase(600, 700, yuv_depth=8, yuv_format=1, alpha=True, yuv_range=1), 9, 2),     # 350-row boxes: the 8-bit row sums wrap
           (H.Y2RCase(600, 700, yuv_depth=10, yuv_format=3, yuv_range=1), 9, 2),
           (H.Y2RCase(320, 200, yuv_depth=8, yuv_format=3, yuv_range=1), 640, 400),               # exact 2x (ScalePlaneUp2_Bilinear)
           (H.Y2RCase(320, 200, yuv_depth=12, yuv_format=2, yuv_range=1, alpha=True), 639, 399),
           (H.Y2RCase(1920, 1080, yuv_depth=8, yuv_format=3, yuv_range=1), 480, 270),            # a thumbnail
           (H.Y2RCase(480, 270, yuv_depth=10, yuv_format=3, yuv_range=1), 1920, 1080),
           # several 256-column segments per row, odd widths, padded (unaligned) rows: every kernel family's tails
           (H.Y2RCase(1001, 77, yuv_depth=8, yuv_format=1, yuv_range=1, row_pad=6), 1333, 91),     # bilinear up
           (H.Y2RCase(1001, 77, yuv_depth=8, yuv_format=3, yuv_range=1, alpha=True), 2002, 154),   # 2x
           (H.Y2RCase(1501, 90, yuv_depth=8, yuv_format=2, yuv_range=1, row_pad=6), 1003, 61),     # bilinear down 1.5x
           (H.Y2RCase(1501, 90, yuv_depth=8, yuv_format=1, yuv_range=1), 801, 47),                 # down 1.87x: windows too wide -> staged
           (H.Y2RCase(2100, 130, yuv_depth=8, yuv_format=3, yuv_range=1, row_pad=6), 519, 31),     # box
           (H.Y2RCase(1300, 64, yuv_depth=12, yuv_format=1, yuv_range=1, row_pad=6), 1733, 81),    # 16-bit samples: staged
           (H.Y2RCase(1300, 64, yuv_depth=10, yuv_format=3, yuv_range=1), 433, 21),
           (H.Y2RCase(5000, 40, yuv_depth=8, yuv_format=1, yuv_range=1), 280, 33),                # 17.9x: segments too long -> gather
           (H.Y2RCase(2048, 96, yuv_depth=8, yuv_format=3, yuv_range=1), 512, 24),                # exact 4x boxes: the dword (v_sad_u8) path
           (H.Y2RCase(2048, 128, yuv_depth=8, yuv_format=1, yuv_range=1, alpha=True), 256, 16),   # exact 8x
           (H.Y2RCase(2052, 96, yuv_depth=8, yuv_format=1, yuv_range=1, row_pad=6), 513, 24)]     # 4x, but rows and tails off the dword grid                 # 17.9x: segments too long -> gather
    for _ in range(n):
        sw, sh = rnd.choice(SIZES)
        if rnd.random() < 0.5:
            dw, dh = max(1, int(sw * rnd.choice(RATIOS))), max(1, int(sh * rnd.choice(RATIOS)))
        else:
            dw, dh = rnd.randint(1, 160), rnd.randint(1, 140)
        if rnd.random() < 0.2:
            dw, dh = max(1, 2 * sw - rnd.choice([0, 1])), max(1, 2 * sh - rnd.choice([0, 1]))
        c = H.Y2RCase(sw, sh, yuv_depth=rnd.choice([8, 10, 12]), yuv_format=rnd.choice([1, 2, 3, 4]), alpha=rnd.random() < 0.4,
                      seed=rnd.getrandbits(30) | 1, yuv_range=1)
        out.append((c, dw, dh))
    return out


def planes_of(st):
    """The (cropped) planes of an avifImage struct whose buffers may have been replaced by the callee."""
    bps = 2 if st.depth > 8 else 1
    cw, ch = abi.chroma_dims(st.width, st.height, st.yuvFormat)
    out = []
    for p in range(4):
        ptr, rb = (st.yuvPlanes[p], st.yuvRowBytes[p]) if p < 3 else (st.alphaPlane, st.alphaRowBytes)
        if not ptr:
            out.append(None)
            continue
        w, h = (st.width, st.height) if p in (0, 3) else (cw, ch)
        out.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, rb))[:, : w * bps].copy())
    return out


def free_owned(st):
    """avifImageScale leaves malloc'ed planes behind (src/avif.c:431-490)."""
    if st.imageOwnsYUVPlanes:
        for p in range(3):
            if st.yuvPlanes[p]:
                libc.free(C.cast(st.yuvPlanes[p], C.c_void_p))
    if st.imageOwnsAlphaPlane and st.alphaPlane:
        libc.free(C.cast(st.alphaPlane, C.c_void_p))


def compare(a_planes, b_planes, what):
    for p, (x, y) in enumerate(zip(a_planes, b_planes)):
        assert (x is None) == (y is None), (what, p)
        if x is not None:
            assert x.shape == y.shape and np.array_equal(x, y), (what, p, int((x != y).sum()) if x.shape == y.shape else "shape")


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_oracle_equals_reference_avifImageScale():
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    diag = C.create_string_buffer(512)
    for c, dw, dh in scale_cases(900, seed=1):
        a, b = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
        ra, rb = ref.avifImageScale(a.struct, dw, dh, diag), o.oracleImageScale(b.struct, dw, dh)
        assert ra == rb, (c.ident(), dw, dh)
        if ra == 0:
            assert (a.struct.width, a.struct.height) == (b.struct.width, b.struct.height) == (dw, dh)
            compare(planes_of(a.struct), planes_of(b.struct), (c.ident(), dw, dh))
        if (c.w, c.h) != (dw, dh):
            free_owned(a.struct)
            free_owned(b.struct)
    img = H.make_y2r_inputs(H.Y2RCase(8, 8))
    assert o.oracleImageScale(img.struct, 0, 4) == ref.avifImageScale(img.struct, 0, 4, diag) == abi.AVIF_RESULT_INVALID_ARGUMENT


@pytest.mark.gpu
def test_gpu_scale_in_place_equals_the_oracle(hip):
    from libavif_amd import native

    o = oracle_lib.oracle()
    kernels = set()
    for c, dw, dh in scale_cases(350, seed=2):
        a, b = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
        ra, rb = o.oracleImageScale(a.struct, dw, dh), hip.avifhipImageScale(b.struct, dw, dh)
        assert ra == rb == 0, (c.ident(), dw, dh, hip.avifhipLastError())
        kernels.add(native.last_kernel())
        assert (b.struct.width, b.struct.height) == (dw, dh)
        compare(planes_of(a.struct), planes_of(b.struct), (c.ident(), dw, dh, native.last_kernel()))
        if (c.w, c.h) != (dw, dh):
            free_owned(a.struct)
            free_owned(b.struct)
    assert {"scale_down", "scale_up", "scale_box", "scale_up2", "scale_point"} <= {k.split("[")[0] for k in kernels}, kernels
    assert {"gather", "staged", "window"} <= {k.split("[")[1].rstrip("]") for k in kernels}, kernels
    img = H.make_y2r_inputs(H.Y2RCase(8, 8))
    assert hip.avifhipImageScale(img.struct, 0, 4) == abi.AVIF_RESULT_INVALID_ARGUMENT


@pytest.mark.gpu
def test_gpu_scale_gather_kernel_only(hip):
    """The one-lane-per-sample kernel (what serves geometries the staged and window kernels decline) on every case."""
    from libavif_amd import native

    o = oracle_lib.oracle()
    hip.avifhipSetTiledKernels(0)
    try:
        for c, dw, dh in scale_cases(120, seed=4):
            a, b = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
            ra, rb = o.oracleImageScale(a.struct, dw, dh), hip.avifhipImageScale(b.struct, dw, dh)
            assert ra == rb == 0, (c.ident(), dw, dh, hip.avifhipLastError())
            assert native.last_kernel().endswith("[gather]") or (c.w, c.h) == (dw, dh), native.last_kernel()
            compare(planes_of(a.struct), planes_of(b.struct), (c.ident(), dw, dh, native.last_kernel()))
            if (c.w, c.h) != (dw, dh):
                free_owned(a.struct)
                free_owned(b.struct)
    finally:
        hip.avifhipSetTiledKernels(1)


@pytest.mark.gpu
def test_gpu_scale_device_resident_equals_the_oracle(hip):
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    for idx, (c, dw, dh) in enumerate(scale_cases(60, seed=3)[:48]):
        tight = idx % 2 == 1  # rows at any byte alignment: the kernels' unaligned-row store paths
        a, src = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
        assert o.oracleImageScale(a.struct, dw, dh) == 0
        dst = H.make_y2r_inputs(H.Y2RCase(dw, dh, yuv_depth=c.yuv_depth, yuv_format=c.yuv_format, alpha=c.alpha, yuv_range=1))
        dsrc, ddst = device.DeviceYUV(src, tight=tight), device.DeviceYUV(dst, tight=tight)
        native.check(hip.avifhipImageScaleAsync(dsrc.struct, ddst.struct, None), "avifhipImageScaleAsync")
        native.check(hip.avifhipSynchronize(None), "sync")
        ddst.download_into_host()
        compare(planes_of(a.struct), planes_of(dst.struct), (c.ident(), dw, dh, tight, native.last_kernel()))
        if (c.w, c.h) != (dw, dh):
            free_owned(a.struct)


@pytest.mark.gpu
def test_gpu_scale_doubling_kernel(hip):
    """2x on both axes (ScalePlaneUp2_Bilinear) in the doubling kernel: even and odd destination sizes on either axis, widths off the 8-column
    and 512-column grids, every 8-bit layout, alpha; device-resident with 256-byte row pitches (the kernel's alignment needs), and tight rows
    (which it declines: the window kernel must serve those with the same bytes)."""
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    seen = set()
    for (w, h), (ox, oy), yf, alpha, tight in [((520, 37), (0, 0), 3, False, False), ((520, 37), (1, 0), 3, True, False), ((521, 36), (0, 1), 1, False, False),
                                               ((1031, 19), (1, 1), 2, False, False), ((64, 64), (0, 0), 4, True, False), ((9, 5), (1, 1), 1, False, False),
                                               ((777, 130), (0, 0), 3, False, True), ((2048, 16), (1, 0), 3, False, False), ((8, 8), (0, 0), 1, True, False)]:
        c = H.Y2RCase(w, h, yuv_depth=8, yuv_format=yf, alpha=alpha, yuv_range=1)
        dw, dh = 2 * w - ox, 2 * h - oy
        a, src = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
        assert o.oracleImageScale(a.struct, dw, dh) == 0
        dst = H.make_y2r_inputs(H.Y2RCase(dw, dh, yuv_depth=8, yuv_format=yf, alpha=alpha, yuv_range=1))
        dsrc, ddst = device.DeviceYUV(src, tight=tight), device.DeviceYUV(dst, tight=tight)
        native.check(hip.avifhipImageScaleAsync(dsrc.struct, ddst.struct, None), "avifhipImageScaleAsync")
        native.check(hip.avifhipSynchronize(None), "sync")
        ddst.download_into_host()
        seen.add((tight, native.last_kernel()))
        compare(planes_of(a.struct), planes_of(dst.struct), (c.ident(), dw, dh, tight, native.last_kernel()))
        free_owned(a.struct)
    assert (False, "scale_up2[doubling]") in seen and not any(t and k == "scale_up2[doubling]" for t, k in seen), seen


@pytest.mark.gpu
def test_gpu_scale_exact_box_kernel(hip):
    """Thumbnails at exactly 1/4 and 1/8 (every box N x N on the N-grid) in the exact-box kernel: destination widths on and off the 4-sample
    grid, every 8-bit layout, alpha; tight rows (unaligned: declined, the row-staged kernel serves them with the same bytes)."""
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    seen = set()
    for (dw, dh), n, yf, alpha, tight in [((260, 33), 4, 1, False, False), ((512, 20), 4, 3, True, False), ((131, 17), 8, 1, False, False), ((258, 9), 4, 4, False, False),
                                          ((64, 64), 8, 3, False, False), ((480, 270), 4, 3, False, False), ((130, 50), 4, 1, True, True), ((6, 6), 4, 1, False, False)]:
        mult = n * (2 if yf == 3 else 1)  # subsampled planes must reduce by the same exact factor
        w, h = dw * n, dh * n
        if yf in (2, 3) and (dw % 2 or (yf == 3 and dh % 2)):
            dw, dh = dw + dw % 2, dh + (dh % 2 if yf == 3 else 0)
            w, h = dw * n, dh * n
        c = H.Y2RCase(w, h, yuv_depth=8, yuv_format=yf, alpha=alpha, yuv_range=1)
        a, src = H.make_y2r_inputs(c), H.make_y2r_inputs(c)
        assert o.oracleImageScale(a.struct, dw, dh) == 0
        dst = H.make_y2r_inputs(H.Y2RCase(dw, dh, yuv_depth=8, yuv_format=yf, alpha=alpha, yuv_range=1))
        dsrc, ddst = device.DeviceYUV(src, tight=tight), device.DeviceYUV(dst, tight=tight)
        native.check(hip.avifhipImageScaleAsync(dsrc.struct, ddst.struct, None), "avifhipImageScaleAsync")
        native.check(hip.avifhipSynchronize(None), "sync")
        ddst.download_into_host()
        seen.add((tight, native.last_kernel()))
        compare(planes_of(a.struct), planes_of(dst.struct), (c.ident(), dw, dh, n, tight, native.last_kernel()))
        free_owned(a.struct)
    assert (False, "scale_box[exact]") in seen, seen
