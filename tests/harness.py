"""Shared machinery of the parity tests: case descriptions, deterministic inputs, and runners that push one
case through a backend (oracle restatement, compiled reference, Pillow's libyuv build, or the HIP library).

A backend is an object with four callables taking ctypes structs and returning an avifResult:
    yuv_to_rgb(image, rgb), rgb_to_yuv(image, rgb), premultiply(rgb), unpremultiply(rgb)
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import random
from dataclasses import dataclass, field, replace
from typing import Callable, Optional

import numpy as np

from libavif_amd import abi, synth

FILL_BYTE = 0xA5  # destination prefill: bytes a conversion must not touch keep this value


@dataclass(frozen=True)
class Y2RCase:
    """One avifImageYUVToRGB configuration."""

    w: int
    h: int
    yuv_depth: int = 8
    yuv_format: int = abi.AVIF_PIXEL_FORMAT_YUV420
    yuv_range: int = abi.AVIF_RANGE_FULL
    matrix: int = abi.AVIF_MATRIX_COEFFICIENTS_BT601
    alpha: bool = False
    image_premultiplied: bool = False
    rgb_depth: int = 8
    rgb_format: int = abi.AVIF_RGB_FORMAT_RGBA
    upsampling: int = abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC
    rgb_premultiplied: bool = False
    ignore_alpha: bool = False
    is_float: bool = False
    avoid_libyuv: bool = True
    row_pad: int = 0
    seed: int = 0x12345678
    color_primaries: int = 1
    pattern: str = "random"  # random | gradient

    def ident(self) -> str:
        fmt = {1: "444", 2: "422", 3: "420", 4: "400"}[self.yuv_format]
        return (f"{self.w}x{self.h}-y{self.yuv_depth}-{fmt}-{'lim' if self.yuv_range == 0 else 'full'}-mc{self.matrix}"
                f"{'-A' if self.alpha else ''}{'p' if self.image_premultiplied else ''}"
                f"-{abi.RGB_FORMAT_NAMES[self.rgb_format]}{self.rgb_depth}-up{self.upsampling}"
                f"{'-pm' if self.rgb_premultiplied else ''}{'-ia' if self.ignore_alpha else ''}{'-f16' if self.is_float else ''}"
                f"{'' if self.avoid_libyuv else '-yuvlib'}{'-pad' if self.row_pad else ''}{'-grad' if self.pattern != 'random' else ''}")


@dataclass(frozen=True)
class R2YCase:
    """One avifImageRGBToYUV configuration."""

    w: int
    h: int
    rgb_depth: int = 8
    rgb_format: int = abi.AVIF_RGB_FORMAT_RGBA
    rgb_premultiplied: bool = False
    ignore_alpha: bool = False
    opaque: bool = False
    yuv_depth: int = 8
    yuv_format: int = abi.AVIF_PIXEL_FORMAT_YUV420
    yuv_range: int = abi.AVIF_RANGE_LIMITED
    matrix: int = abi.AVIF_MATRIX_COEFFICIENTS_BT709
    image_premultiplied: bool = False
    avoid_libyuv: bool = True
    row_pad: int = 0
    seed: int = 0xCAFEBABE
    color_primaries: int = 1

    def ident(self) -> str:
        fmt = {1: "444", 2: "422", 3: "420", 4: "400"}[self.yuv_format]
        return (f"{self.w}x{self.h}-{abi.RGB_FORMAT_NAMES[self.rgb_format]}{self.rgb_depth}"
                f"{'-pm' if self.rgb_premultiplied else ''}{'-ia' if self.ignore_alpha else ''}{'-op' if self.opaque else ''}"
                f"-y{self.yuv_depth}-{fmt}-{'lim' if self.yuv_range == 0 else 'full'}-mc{self.matrix}"
                f"{'-ip' if self.image_premultiplied else ''}{'' if self.avoid_libyuv else '-yuvlib'}{'-pad' if self.row_pad else ''}")


# ---------------------------------------------------------------------------------------------------
# inputs


def make_y2r_inputs(c: Y2RCase):
    img = abi.make_yuv(c.w, c.h, c.yuv_depth, c.yuv_format, c.yuv_range, c.matrix, with_alpha=c.alpha,
                       alpha_premultiplied=c.image_premultiplied, row_pad=c.row_pad, color_primaries=c.color_primaries)
    for buf in img.planes + [img.alpha]:
        if buf is not None:
            buf[...] = 0x5A  # row padding content is irrelevant but deterministic
    if c.pattern == "gradient":
        synth.gradient_planes(img)
    else:
        synth.fill_yuv(img, c.seed)
    return img


def make_y2r_output(c: Y2RCase) -> abi.HostRGB:
    return abi.make_rgb(c.w, c.h, c.rgb_depth, c.rgb_format, upsampling=c.upsampling, avoid_libyuv=c.avoid_libyuv,
                        ignore_alpha=c.ignore_alpha, alpha_premultiplied=c.rgb_premultiplied, is_float=c.is_float,
                        row_pad=c.row_pad, fill=FILL_BYTE)


def make_r2y_inputs(c: R2YCase) -> abi.HostRGB:
    rgb = abi.make_rgb(c.w, c.h, c.rgb_depth, c.rgb_format, avoid_libyuv=c.avoid_libyuv, ignore_alpha=c.ignore_alpha,
                       alpha_premultiplied=c.rgb_premultiplied, row_pad=c.row_pad, fill=0x5A)
    synth.fill_rgb(rgb, c.seed, opaque=c.opaque)
    if c.rgb_premultiplied and abi.rgb_format_has_alpha(c.rgb_format):
        # make the data a legal premultiplied image (colour <= alpha) for half of the pixels; the other half
        # keeps arbitrary values to exercise the clamp in the unmultiply path
        ch = rgb.channels()
        nch = ch.shape[2]
        a_first = c.rgb_format in (abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_ABGR, abi.AVIF_RGB_FORMAT_AGRAY)
        a = ch[:, :, 0 if a_first else nch - 1].astype(np.int64)
        for k in range(nch):
            if k == (0 if a_first else nch - 1):
                continue
            col = ch[:, ::2, k]
            col[...] = np.minimum(col, a[:, ::2]).astype(ch.dtype)
    return rgb


def make_r2y_output(c: R2YCase) -> abi.HostYUV:
    want_alpha = abi.rgb_format_has_alpha(c.rgb_format) and not c.ignore_alpha
    img = abi.make_yuv(c.w, c.h, c.yuv_depth, c.yuv_format, c.yuv_range, c.matrix, with_alpha=want_alpha,
                       alpha_premultiplied=c.image_premultiplied, row_pad=c.row_pad, color_primaries=c.color_primaries)
    for buf in img.planes + [img.alpha]:
        if buf is not None:
            buf[...] = FILL_BYTE
    return img


# ---------------------------------------------------------------------------------------------------
# backends


class Backend:
    def __init__(self, name, yuv_to_rgb, rgb_to_yuv, premultiply, unpremultiply):
        self.name = name
        self.yuv_to_rgb, self.rgb_to_yuv = yuv_to_rgb, rgb_to_yuv
        self.premultiply, self.unpremultiply = premultiply, unpremultiply


def oracle_backend() -> Backend:
    import oracle_lib

    o = oracle_lib.oracle()
    return Backend("oracle", o.oracleImageYUVToRGB, o.oracleImageRGBToYUV, o.oracleRGBImagePremultiplyAlpha,
                   o.oracleRGBImageUnpremultiplyAlpha)


def oracle_libyuv_backend() -> Backend:
    import oracle_lib

    o = oracle_lib.oracle()
    return Backend("oracle-libyuv", o.oracleLibyuvImageYUVToRGB, o.oracleLibyuvImageRGBToYUV,
                   o.oracleLibyuvRGBImagePremultiplyAlpha, o.oracleLibyuvRGBImageUnpremultiplyAlpha)


def libavif_backend(lib, name: str) -> Backend:
    return Backend(name, lib.avifImageYUVToRGB, lib.avifImageRGBToYUV, lib.avifRGBImagePremultiplyAlpha,
                   lib.avifRGBImageUnpremultiplyAlpha)


def hip_host_backend() -> Backend:
    """The product's synchronous entry points on host buffers (staged through HBM by the library)."""
    from libavif_amd import native

    lib = native.load()
    return Backend("hip-host", lib.avifhipImageYUVToRGB, lib.avifhipImageRGBToYUV, lib.avifhipRGBImagePremultiplyAlpha,
                   lib.avifhipRGBImageUnpremultiplyAlpha)


class HipDeviceBackend(Backend):
    """The product's device-resident (Async) entry points: inputs uploaded by the test, kernels run in place."""

    def __init__(self):
        from libavif_amd import device, native

        self.lib = native.load()
        self.device = device
        self.native = native
        super().__init__("hip-device", self._y2r, self._r2y, self._pre, self._unpre)
        self._host_images: dict = {}

    def bind_host(self, struct, host_obj) -> None:
        self._host_images[C.addressof(struct)] = host_obj

    def _y2r(self, image_struct, rgb_struct):
        img = self._host_images[C.addressof(image_struct)]
        rgb = self._host_images[C.addressof(rgb_struct)]
        dimg = self.device.DeviceYUV(img)
        drgb = self.device.DeviceRGB(rgb, upload=True)
        r = self.lib.avifhipImageYUVToRGBAsync(dimg.struct, drgb.struct, None)
        if r == abi.AVIF_RESULT_OK:
            self.native.check(self.lib.avifhipSynchronize(None), "sync")
            drgb.download_into_host()
        return r

    def _r2y(self, image_struct, rgb_struct):
        img = self._host_images[C.addressof(image_struct)]
        rgb = self._host_images[C.addressof(rgb_struct)]
        dimg = self.device.DeviceYUV(img)  # destination planes keep the prefill
        drgb = self.device.DeviceRGB(rgb, upload=True)
        r = self.lib.avifhipImageRGBToYUVAsync(dimg.struct, drgb.struct, None)
        if r == abi.AVIF_RESULT_OK:
            self.native.check(self.lib.avifhipSynchronize(None), "sync")
            dimg.download_into_host()
        return r

    def _alpha(self, rgb_struct, fn):
        rgb = self._host_images[C.addressof(rgb_struct)]
        drgb = self.device.DeviceRGB(rgb, upload=True)
        r = fn(drgb.struct, None)
        if r == abi.AVIF_RESULT_OK:
            self.native.check(self.lib.avifhipSynchronize(None), "sync")
            drgb.download_into_host()
        return r

    def _pre(self, rgb_struct):
        return self._alpha(rgb_struct, self.lib.avifhipRGBImagePremultiplyAlphaAsync)

    def _unpre(self, rgb_struct):
        return self._alpha(rgb_struct, self.lib.avifhipRGBImageUnpremultiplyAlphaAsync)


# ---------------------------------------------------------------------------------------------------
# runners


_KEPT = []  # AVIFHIP_TEST_KEEP=1 (tests/tools/fault_hunt.sh): no buffer a conversion has seen is ever freed


def run_y2r(backend: Backend, c: Y2RCase):
    img = make_y2r_inputs(c)
    rgb = make_y2r_output(c)
    if os.environ.get("AVIFHIP_TEST_KEEP"):
        _KEPT.append((img, rgb))
    if isinstance(backend, HipDeviceBackend):
        backend.bind_host(img.struct, img)
        backend.bind_host(rgb.struct, rgb)
    res = backend.yuv_to_rgb(img.struct, rgb.struct)
    return res, rgb.pixels


def run_r2y(backend: Backend, c: R2YCase):
    rgb = make_r2y_inputs(c)
    img = make_r2y_output(c)
    if isinstance(backend, HipDeviceBackend):
        backend.bind_host(img.struct, img)
        backend.bind_host(rgb.struct, rgb)
    res = backend.rgb_to_yuv(img.struct, rgb.struct)
    return res, img


def planes_equal(a: abi.HostYUV, b: abi.HostYUV, padding: bool = True) -> Optional[str]:
    """None if every plane byte (row padding included unless padding=False) matches, else the first difference."""
    for p, (x, y) in enumerate(zip(a.planes + [a.alpha], b.planes + [b.alpha])):
        if (x is None) != (y is None):
            return f"plane {p}: presence differs"
        if x is None:
            continue
        if not padding:
            x, y = a.plane_samples(p), b.plane_samples(p)
        if not np.array_equal(x, y):
            d = np.argwhere(x != y)[0]
            return f"plane {p}: first difference at row {d[0]} byte {d[1]}: {x[tuple(d)]} != {y[tuple(d)]} ({(x != y).sum()} bytes differ)"
    return None


def describe_diff(a: np.ndarray, b: np.ndarray) -> str:
    diff = a != b
    d = np.argwhere(diff)[0]
    return (f"{diff.sum()} bytes differ, first at row {d[0]} byte {d[1]}: {a[tuple(d)]} != {b[tuple(d)]}, "
            f"max |delta| {np.abs(a.astype(np.int32) - b.astype(np.int32)).max()}")


# ---------------------------------------------------------------------------------------------------
# case sweeps

ALL_RGB_FORMATS = list(range(10))
COLOR_MATRICES = [1, 2, 4, 5, 6, 7, 9, 12]


def valid_y2r(c: Y2RCase) -> bool:
    """Combinations the reference accepts (src/reformat.c:32-194) -- invalid ones are covered by the error tests."""
    if c.is_float and c.rgb_depth != 16:
        return False
    if c.rgb_format == abi.AVIF_RGB_FORMAT_RGB_565 and c.rgb_depth != 8:
        return False
    if c.matrix in (8, 16, 17) and c.yuv_range == abi.AVIF_RANGE_LIMITED:
        return False
    if c.matrix == 0 and c.yuv_format not in (abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_PIXEL_FORMAT_YUV400):
        return False
    if c.matrix == 16 and c.yuv_depth - 2 != c.rgb_depth:
        return False
    if c.matrix == 17 and c.yuv_depth - 1 != c.rgb_depth:
        return False
    return True


# Seed rotation ("a nightly-style --seed rotation would cost nothing"): `pytest --seed-rotation N` (or AVIFHIP_TEST_SEED_ROTATION=N)
# moves the RANDOM part of every sweep to another stream; 0, the default, is the fixed set every committed figure and fixture was made with.
# The structured part of a sweep (every format x depth x matrix ... combination it enumerates) does not depend on it.
def sweep_rng(seed: int) -> random.Random:
    rotation = int(os.environ.get("AVIFHIP_TEST_SEED_ROTATION", "0") or "0")
    return random.Random(seed if rotation == 0 else seed + 1000003 * rotation)


def y2r_sweep(sizes, n_random: int, seed: int = 7) -> list:
    """Structured + seeded-random sample of the configuration space, deduplicated, all valid."""
    rnd = sweep_rng(seed)
    cases: list = []
    w0, h0 = sizes[0]
    # every RGB format x container, default everything else
    for fmt, depth in itertools.product(ALL_RGB_FORMATS, (8, 10, 12, 16)):
        cases.append(Y2RCase(w0, h0, rgb_format=fmt, rgb_depth=depth, yuv_range=abi.AVIF_RANGE_LIMITED, matrix=1))
    # every yuv format x depth x range x upsampling
    for yf, yd, yr, up in itertools.product((1, 2, 3, 4), (8, 10, 12), (0, 1), (0, 1, 2, 3, 4)):
        cases.append(Y2RCase(w0, h0, yuv_format=yf, yuv_depth=yd, yuv_range=yr, upsampling=up, matrix=9, rgb_depth=rnd.choice((8, 16))))
    # every matrix
    for mc in COLOR_MATRICES + [0, 8]:
        yf = rnd.choice((1, 4)) if mc == 0 else rnd.choice((1, 2, 3))
        cases.append(Y2RCase(w0, h0, yuv_format=yf, matrix=mc, yuv_range=1 if mc in (0, 8) else rnd.choice((0, 1)),
                             color_primaries=rnd.choice((1, 9, 12))))
        cases.append(Y2RCase(w0, h0, yuv_format=yf, yuv_depth=10, rgb_depth=16, matrix=mc, yuv_range=1 if mc in (0, 8) else 0))
    for yf in (1, 2, 3):
        cases.append(Y2RCase(w0, h0, yuv_format=yf, yuv_depth=10, rgb_depth=8, matrix=16, yuv_range=1))
        cases.append(Y2RCase(w0, h0, yuv_format=yf, yuv_depth=12, rgb_depth=10, matrix=16, yuv_range=1))
        cases.append(Y2RCase(w0, h0, yuv_format=yf, yuv_depth=16, rgb_depth=16, matrix=8, yuv_range=1))
    # alpha handling matrix
    for fmt, ip, rp, ia, yf, up in itertools.product((0, 1, 2, 4, 5, 7, 8, 9), (False, True), (False, True), (False, True), (1, 3), (3, 4)):
        cases.append(Y2RCase(w0, h0, alpha=True, image_premultiplied=ip, rgb_premultiplied=rp, ignore_alpha=ia, rgb_format=fmt,
                             yuv_format=yf, upsampling=up, matrix=1, yuv_range=0, yuv_depth=rnd.choice((8, 10)),
                             rgb_depth=rnd.choice((8, 16))))
    # half float
    for fmt in (0, 1, 2, 7, 8):
        cases.append(Y2RCase(w0, h0, rgb_format=fmt, rgb_depth=16, is_float=True, alpha=fmt in (1, 2, 8), yuv_depth=10,
                             rgb_premultiplied=(fmt == 1)))
    cases.append(Y2RCase(w0, h0, rgb_format=1, rgb_depth=16, is_float=True, ignore_alpha=True))
    # sizes / strides
    for (w, h), yf, up, pad in itertools.product(sizes, (1, 2, 3, 4), (3, 4), (0, 64)):
        cases.append(Y2RCase(w, h, yuv_format=yf, upsampling=up, row_pad=pad, matrix=1, yuv_range=0, alpha=rnd.random() < 0.3,
                             rgb_format=rnd.choice((0, 1, 4)), yuv_depth=rnd.choice((8, 10)), rgb_depth=rnd.choice((8, 16))))
    # seeded random points of the full space
    for _ in range(n_random):
        w, h = rnd.choice(sizes)
        mc = rnd.choice(COLOR_MATRICES + [0, 8, 16, 17, 1, 1, 6, 9])
        yd = rnd.choice((8, 10, 12))
        c = Y2RCase(w, h, yuv_depth=yd, yuv_format=rnd.choice((1, 2, 3, 4)), yuv_range=rnd.choice((0, 1)), matrix=mc,
                    alpha=rnd.random() < 0.5, image_premultiplied=rnd.random() < 0.3,
                    rgb_depth=rnd.choice((8, 10, 12, 16)), rgb_format=rnd.choice(ALL_RGB_FORMATS), upsampling=rnd.choice((0, 1, 2, 3, 4)),
                    rgb_premultiplied=rnd.random() < 0.3, ignore_alpha=rnd.random() < 0.2, is_float=rnd.random() < 0.1,
                    row_pad=rnd.choice((0, 0, 64, 6)), seed=rnd.getrandbits(31) | 1, color_primaries=rnd.choice((1, 5, 9, 12, 2)),
                    pattern=rnd.choice(("random", "random", "gradient")))
        if c.matrix in (0, 8, 16, 17):
            c = replace(c, yuv_range=1)
        if c.matrix == 0 and c.yuv_format in (2, 3):
            c = replace(c, yuv_format=1)
        if c.matrix == 16:
            c = replace(c, yuv_depth=rnd.choice((10, 12)))
            c = replace(c, rgb_depth=c.yuv_depth - 2)
        if c.matrix == 17:
            c = replace(c, yuv_depth=12, rgb_depth=8)  # no legal pair exists (depth-1 is odd): keeps the error path honest
        if c.is_float:
            c = replace(c, rgb_depth=16)
        if c.rgb_format == abi.AVIF_RGB_FORMAT_RGB_565:
            c = replace(c, rgb_depth=8, is_float=False)
        cases.append(c)
    seen, out = set(), []
    for c in cases:
        if c in seen or not valid_y2r(c):
            continue
        seen.add(c)
        out.append(c)
    return out


def r2y_sweep(sizes, n_random: int, seed: int = 11) -> list:
    rnd = sweep_rng(seed)
    cases: list = []
    w0, h0 = sizes[0]
    rgb_formats = [f for f in ALL_RGB_FORMATS if f != abi.AVIF_RGB_FORMAT_RGB_565]
    for fmt, depth in itertools.product(rgb_formats, (8, 10, 12, 16)):
        cases.append(R2YCase(w0, h0, rgb_format=fmt, rgb_depth=depth, yuv_depth=rnd.choice((8, 10, 12))))
    for yf, yd, yr in itertools.product((1, 2, 3, 4), (8, 10, 12), (0, 1)):
        cases.append(R2YCase(w0, h0, yuv_format=yf, yuv_depth=yd, yuv_range=yr, rgb_depth=rnd.choice((8, 16)), matrix=rnd.choice((1, 6, 9))))
    for mc in COLOR_MATRICES + [0, 8]:
        yf = rnd.choice((1, 4)) if mc == 0 else rnd.choice((1, 2, 3))
        cases.append(R2YCase(w0, h0, yuv_format=yf, matrix=mc, yuv_range=1 if mc in (0, 8) else rnd.choice((0, 1)), color_primaries=rnd.choice((1, 9, 12))))
    for yf in (1, 2, 3):
        cases.append(R2YCase(w0, h0, yuv_format=yf, rgb_depth=8, yuv_depth=10, matrix=16, yuv_range=1))
        cases.append(R2YCase(w0, h0, yuv_format=yf, rgb_depth=10, yuv_depth=12, matrix=16, yuv_range=1, rgb_format=0))
    for fmt, rp, ip, ia in itertools.product((1, 2, 4, 5, 8, 9), (False, True), (False, True), (False, True)):
        cases.append(R2YCase(w0, h0, rgb_format=fmt, rgb_premultiplied=rp, image_premultiplied=ip, ignore_alpha=ia,
                             rgb_depth=rnd.choice((8, 16)), yuv_depth=rnd.choice((8, 10)), yuv_format=rnd.choice((1, 2, 3))))
    for (w, h), yf, pad in itertools.product(sizes, (1, 2, 3, 4), (0, 64)):
        cases.append(R2YCase(w, h, yuv_format=yf, row_pad=pad, rgb_format=rnd.choice((0, 1, 4, 7)), opaque=rnd.random() < 0.5))
    for _ in range(n_random):
        w, h = rnd.choice(sizes)
        mc = rnd.choice(COLOR_MATRICES + [0, 8, 16, 1, 6, 9])
        c = R2YCase(w, h, rgb_depth=rnd.choice((8, 10, 12, 16)), rgb_format=rnd.choice(rgb_formats), rgb_premultiplied=rnd.random() < 0.3,
                    ignore_alpha=rnd.random() < 0.2, opaque=rnd.random() < 0.3, yuv_depth=rnd.choice((8, 10, 12)),
                    yuv_format=rnd.choice((1, 2, 3, 4)), yuv_range=rnd.choice((0, 1)), matrix=mc, image_premultiplied=rnd.random() < 0.3,
                    row_pad=rnd.choice((0, 0, 64, 6)), seed=rnd.getrandbits(31) | 1, color_primaries=rnd.choice((1, 5, 9, 12)))
        if c.matrix in (0, 8, 16):
            c = replace(c, yuv_range=1)
        if c.matrix == 0 and c.yuv_format in (2, 3):
            c = replace(c, yuv_format=1)
        if c.matrix == 16:
            c = replace(c, yuv_depth=rnd.choice((10, 12)))
            c = replace(c, rgb_depth=c.yuv_depth - 2)
        cases.append(c)
    seen, out = set(), []
    for c in cases:
        if c in seen:
            continue
        seen.add(c)
        out.append(c)
    return out


# ---------------------------------------------------------------------------------------------------
# the sub-space libavif hands to libyuv (the reference's integer path)

def libyuv_y2r_cases(sizes, n_random=1500, seed=23):
    """Dense sample of the sub-space libavif hands to libyuv: 8-bit RGB outputs."""
    rnd = sweep_rng(seed)
    cases = []
    for fmt, yf, yd, up in itertools.product(range(7), (1, 2, 3, 4), (8, 10, 12), (0, 1, 2, 3, 4)):
        cases.append(Y2RCase(38, 11, rgb_format=fmt, yuv_format=yf, yuv_depth=yd, upsampling=up, matrix=rnd.choice((1, 5, 6, 2, 9)),
                               yuv_range=rnd.choice((0, 1)), alpha=rnd.random() < 0.4, avoid_libyuv=False, seed=rnd.getrandbits(31) | 1))
    for mc, cp, yr, yf in itertools.product((0, 1, 2, 4, 5, 6, 7, 8, 9, 12), (1, 2, 5, 6, 9, 12), (0, 1), (1, 3, 4)):
        c = Y2RCase(21, 9, matrix=mc, color_primaries=cp, yuv_range=yr, yuv_format=yf, avoid_libyuv=False, rgb_format=rnd.choice((0, 1, 4)))
        if valid_y2r(c):
            cases.append(c)
    for _ in range(n_random):
        w, h = rnd.choice(sizes)
        cases.append(Y2RCase(w, h, yuv_depth=rnd.choice((8, 8, 10, 12)), yuv_format=rnd.choice((1, 2, 3, 3, 4)), yuv_range=rnd.choice((0, 1)),
                               matrix=rnd.choice((1, 5, 6, 2, 9, 12)), color_primaries=rnd.choice((1, 2, 5, 6, 9)), alpha=rnd.random() < 0.5,
                               image_premultiplied=rnd.random() < 0.3, rgb_depth=8, rgb_format=rnd.choice(range(7)),
                               upsampling=rnd.choice((0, 1, 2, 3, 4)), rgb_premultiplied=rnd.random() < 0.3, ignore_alpha=rnd.random() < 0.2,
                               avoid_libyuv=False, row_pad=rnd.choice((0, 0, 6, 64)), seed=rnd.getrandbits(31) | 1,
                               pattern=rnd.choice(("random", "random", "gradient"))))
    seen, out = set(), []
    for c in cases:
        if c not in seen and valid_y2r(c):
            seen.add(c)
            out.append(c)
    return out



def libyuv_r2y_cases(sizes, n_random=1200, seed=29):
    rnd = sweep_rng(seed)
    cases = []
    for fmt, yf, yr, mc in itertools.product((0, 1, 2, 3, 4, 5, 7, 8, 9), (1, 2, 3, 4), (0, 1), (5, 6, 2, 1)):
        cases.append(R2YCase(23, 7, rgb_depth=8, yuv_depth=8, rgb_format=fmt, yuv_format=yf, yuv_range=yr, matrix=mc, avoid_libyuv=False,
                               seed=rnd.getrandbits(31) | 1))
    for _ in range(n_random):
        w, h = rnd.choice(sizes)
        cases.append(R2YCase(w, h, rgb_depth=8, yuv_depth=8, rgb_format=rnd.choice((0, 1, 2, 3, 4, 5)), matrix=rnd.choice((5, 6)),
                               yuv_range=rnd.choice((0, 1)), yuv_format=rnd.choice((1, 2, 3, 3, 4)), avoid_libyuv=False,
                               opaque=rnd.random() < 0.3, ignore_alpha=rnd.random() < 0.3, rgb_premultiplied=rnd.random() < 0.15,
                               row_pad=rnd.choice((0, 6, 64)), seed=rnd.getrandbits(31) | 1))
    return cases




# ---------------------------------------------------------------------------------------------------
# grid images: separately stored tiles (the decode-side tail)


@dataclass(frozen=True)
class GridCase:
    """A grid of `rows` x `columns` tiles of tile_w x tile_h covering out_w x out_h, converted like Y2RCase `conv`
    (whose w, h are ignored)."""

    rows: int
    columns: int
    tile_w: int
    tile_h: int
    out_w: int
    out_h: int
    conv: Y2RCase
    alpha_limited: bool = False

    def ident(self) -> str:
        return f"grid{self.rows}x{self.columns}-tile{self.tile_w}x{self.tile_h}-out{self.out_w}x{self.out_h}{'-alim' if self.alpha_limited else ''}-" + self.conv.ident()


def make_grid_tiles(g: GridCase):
    """Independent random tiles (tile t seeded with conv.seed + t), alpha in each tile image's alpha plane."""
    tiles = []
    for t in range(g.rows * g.columns):
        c = replace(g.conv, w=g.tile_w, h=g.tile_h, seed=(g.conv.seed + 7919 * t) & 0x7FFFFFFF | 1)
        img = make_y2r_inputs(c)
        if g.alpha_limited and img.alpha is not None:
            d = c.yuv_depth
            a = img.plane_samples(3)
            lo, hi = 16 << (d - 8), 235 << (d - 8)
            a[...] = (lo + a.astype(np.int64) % (hi - lo + 1) if t % 2 else a.astype(np.int64) % ((1 << d))).astype(a.dtype)  # some tiles out of range on purpose
        tiles.append(img)
    return tiles


def grid_output(g: GridCase) -> abi.HostRGB:
    return make_y2r_output(replace(g.conv, w=g.out_w, h=g.out_h))
