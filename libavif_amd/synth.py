"""Synthetic inputs of BASELINE.md section 3 / SURVEY.md 8d.

xorshift32 (x^=x<<13; x^=x>>17; x^=x<<5), one draw per sample in plane order Y,U,V,A, row-major; limited
range draws in the legal interval [16,235]/[16,240]<<(d-8), full range and alpha in [0, 2^d-1]; value =
lo + draw % (hi-lo+1).  The generator itself is C (avifhipSynthFill) so 8K planes fill in a fraction of a second.
"""
from __future__ import annotations

import numpy as np

from . import abi, native


def fill_yuv(img: abi.HostYUV, seed: int = 0x12345678) -> int:
    """Fills every allocated plane of `img`; returns the advanced generator state."""
    lib = native.load()
    st = img.struct
    d = st.depth
    bps = 2 if d > 8 else 1
    full = (0, (1 << d) - 1)
    if st.yuvRange == abi.AVIF_RANGE_LIMITED:
        ry = (16 << (d - 8), 235 << (d - 8))
        ruv = (16 << (d - 8), 240 << (d - 8))
    else:
        ry = ruv = full
    cw, ch = abi.chroma_dims(st.width, st.height, st.yuvFormat)
    state = seed
    specs = [(img.planes[0], st.yuvRowBytes[0], st.width, st.height, ry),
             (img.planes[1], st.yuvRowBytes[1], cw, ch, ruv),
             (img.planes[2], st.yuvRowBytes[2], cw, ch, ruv),
             (img.alpha, st.alphaRowBytes, st.width, st.height, full)]
    for buf, rb, w, h, (lo, hi) in specs:
        if buf is None:
            continue
        state = lib.avifhipSynthFill(state, buf.ctypes.data, rb, w, h, bps, lo, hi)
    return state


def fill_rgb(rgb: abi.HostRGB, seed: int = 0xCAFEBABE, opaque: bool = False) -> int:
    """Random interleaved pixels over the full channel range (one draw per channel); opaque forces A = max."""
    lib = native.load()
    st = rgb.struct
    if st.format == abi.AVIF_RGB_FORMAT_RGB_565:
        return lib.avifhipSynthFill(seed, rgb.pixels.ctypes.data, st.rowBytes, st.width, st.height, 2, 0, 0xFFFF)
    nch = abi.rgb_format_channel_count(st.format)
    bps = 2 if st.depth > 8 else 1
    state = lib.avifhipSynthFill(seed, rgb.pixels.ctypes.data, st.rowBytes, st.width * nch, st.height, bps, 0, (1 << st.depth) - 1)
    if opaque and abi.rgb_format_has_alpha(st.format):
        ch = rgb.channels()
        a_first = st.format in (abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_ABGR, abi.AVIF_RGB_FORMAT_AGRAY)
        ch[:, :, 0 if a_first else nch - 1] = (1 << st.depth) - 1
    return state


def constant_planes(img: abi.HostYUV, y: int, u: int, v: int, a: int | None = None) -> None:
    vals = [y, u, v]
    for p in range(3):
        if img.planes[p] is not None:
            img.plane_samples(p)[...] = vals[p]
    if img.alpha is not None and a is not None:
        img.plane_samples(3)[...] = a


def gradient_planes(img: abi.HostYUV) -> None:
    """Deterministic ramps touching the full legal range (in the spirit of FillImageGradient,
    tests/gtest/aviftest_helpers.cc:150-186)."""
    st = img.struct
    d = st.depth
    maxv = (1 << d) - 1
    lo_y, hi_y, lo_c, hi_c = 0, maxv, 0, maxv
    if st.yuvRange == abi.AVIF_RANGE_LIMITED:
        lo_y, hi_y, lo_c, hi_c = 16 << (d - 8), 235 << (d - 8), 16 << (d - 8), 240 << (d - 8)
    for p in range(3):
        if img.planes[p] is None:
            continue
        s = img.plane_samples(p)
        h, w = s.shape
        lo, hi = (lo_y, hi_y) if p == 0 else (lo_c, hi_c)
        xs = np.arange(w, dtype=np.int64)[None, :]
        ys = np.arange(h, dtype=np.int64)[:, None]
        ramp = (xs * (3 + p) + ys * (5 - p)) % (hi - lo + 1) + lo
        s[...] = ramp.astype(s.dtype)
    if img.alpha is not None:
        s = img.plane_samples(3)
        h, w = s.shape
        xs = np.arange(w, dtype=np.int64)[None, :]
        ys = np.arange(h, dtype=np.int64)[:, None]
        s[...] = ((xs * 7 + ys * 3) % (maxv + 1)).astype(s.dtype)
