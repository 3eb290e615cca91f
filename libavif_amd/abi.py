"""ctypes mirror of the libavif boundary types (avifImage / avifRGBImage).

Field names, enum names and values follow libavif's public header
(reference include/avif/avif.h:777-851, :948-1018); the C-side twin is
include/avifhip/avif_abi.h.  Planes and pixels are plain numpy buffers (host)
or raw device addresses (see libavif_amd.device), never torch tensors.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

# --- avifResult (avif.h:163-205) -------------------------------------------------
AVIF_RESULT_OK = 0
AVIF_RESULT_UNKNOWN_ERROR = 1
AVIF_RESULT_REFORMAT_FAILED = 5
AVIF_RESULT_INVALID_ARGUMENT = 24
AVIF_RESULT_NOT_IMPLEMENTED = 25
AVIF_RESULT_OUT_OF_MEMORY = 26
AVIF_RESULT_INTERNAL_ERROR = 29

# --- avifPixelFormat (avif.h:280-289) --------------------------------------------
AVIF_PIXEL_FORMAT_NONE = 0
AVIF_PIXEL_FORMAT_YUV444 = 1
AVIF_PIXEL_FORMAT_YUV422 = 2
AVIF_PIXEL_FORMAT_YUV420 = 3
AVIF_PIXEL_FORMAT_YUV400 = 4

AVIF_RANGE_LIMITED = 0
AVIF_RANGE_FULL = 1

# --- avifMatrixCoefficients (avif.h:394-413) -------------------------------------
AVIF_MATRIX_COEFFICIENTS_IDENTITY = 0
AVIF_MATRIX_COEFFICIENTS_BT709 = 1
AVIF_MATRIX_COEFFICIENTS_UNSPECIFIED = 2
AVIF_MATRIX_COEFFICIENTS_FCC = 4
AVIF_MATRIX_COEFFICIENTS_BT470BG = 5
AVIF_MATRIX_COEFFICIENTS_BT601 = 6
AVIF_MATRIX_COEFFICIENTS_SMPTE240 = 7
AVIF_MATRIX_COEFFICIENTS_YCGCO = 8
AVIF_MATRIX_COEFFICIENTS_BT2020_NCL = 9
AVIF_MATRIX_COEFFICIENTS_BT2020_CL = 10
AVIF_MATRIX_COEFFICIENTS_SMPTE2085 = 11
AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_NCL = 12
AVIF_MATRIX_COEFFICIENTS_CHROMA_DERIVED_CL = 13
AVIF_MATRIX_COEFFICIENTS_ICTCP = 14
AVIF_MATRIX_COEFFICIENTS_YCGCO_RE = 16
AVIF_MATRIX_COEFFICIENTS_YCGCO_RO = 17

# --- avifRGBFormat (avif.h:948-971) ----------------------------------------------
AVIF_RGB_FORMAT_RGB = 0
AVIF_RGB_FORMAT_RGBA = 1
AVIF_RGB_FORMAT_ARGB = 2
AVIF_RGB_FORMAT_BGR = 3
AVIF_RGB_FORMAT_BGRA = 4
AVIF_RGB_FORMAT_ABGR = 5
AVIF_RGB_FORMAT_RGB_565 = 6
AVIF_RGB_FORMAT_GRAY = 7
AVIF_RGB_FORMAT_GRAYA = 8
AVIF_RGB_FORMAT_AGRAY = 9
RGB_FORMAT_NAMES = ["RGB", "RGBA", "ARGB", "BGR", "BGRA", "ABGR", "RGB_565", "GRAY", "GRAYA", "AGRAY"]

# --- avifChromaUpsampling / Downsampling (avif.h:975-992) ------------------------
AVIF_CHROMA_UPSAMPLING_AUTOMATIC = 0
AVIF_CHROMA_UPSAMPLING_FASTEST = 1
AVIF_CHROMA_UPSAMPLING_BEST_QUALITY = 2
AVIF_CHROMA_UPSAMPLING_NEAREST = 3
AVIF_CHROMA_UPSAMPLING_BILINEAR = 4
AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC = 0
AVIF_CHROMA_DOWNSAMPLING_SHARP_YUV = 4


class avifCropRect(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class avifImage(C.Structure):
    """Leading part of libavif's avifImage + opaque tail (sizeof == 224)."""

    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("depth", C.c_uint32),
        ("yuvFormat", C.c_int),
        ("yuvRange", C.c_int),
        ("yuvChromaSamplePosition", C.c_int),
        ("yuvPlanes", C.c_void_p * 3),
        ("yuvRowBytes", C.c_uint32 * 3),
        ("imageOwnsYUVPlanes", C.c_int),
        ("alphaPlane", C.c_void_p),
        ("alphaRowBytes", C.c_uint32),
        ("imageOwnsAlphaPlane", C.c_int),
        ("alphaPremultiplied", C.c_int),
        ("_icc", C.c_uint64 * 2),
        ("colorPrimaries", C.c_uint16),
        ("transferCharacteristics", C.c_uint16),
        ("matrixCoefficients", C.c_uint16),
        ("_tail", C.c_uint8 * (224 - 110)),
    ]


class avifRGBImage(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("depth", C.c_uint32),
        ("format", C.c_int),
        ("chromaUpsampling", C.c_int),
        ("chromaDownsampling", C.c_int),
        ("avoidLibYUV", C.c_int),
        ("ignoreAlpha", C.c_int),
        ("alphaPremultiplied", C.c_int),
        ("isFloat", C.c_int),
        ("maxThreads", C.c_int),
        ("pixels", C.c_void_p),
        ("rowBytes", C.c_uint32),
    ]


assert C.sizeof(avifImage) == 224 and avifImage.matrixCoefficients.offset == 108
assert C.sizeof(avifRGBImage) == 64 and avifRGBImage.pixels.offset == 48


# --- gain maps (avif.h:419-453, :582-610, :630-711) -------------------------------
AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE = 32


class avifSignedFraction(C.Structure):
    _fields_ = [("n", C.c_int32), ("d", C.c_uint32)]


class avifUnsignedFraction(C.Structure):
    _fields_ = [("n", C.c_uint32), ("d", C.c_uint32)]


class avifContentLightLevelInformationBox(C.Structure):
    _fields_ = [("maxCLL", C.c_uint16), ("maxPALL", C.c_uint16)]


class avifDiagnostics(C.Structure):
    _fields_ = [("error", C.c_char * 256)]


class avifGainMap(C.Structure):
    _fields_ = [
        ("image", C.POINTER(avifImage)),
        ("gainMapMin", avifSignedFraction * 3),
        ("gainMapMax", avifSignedFraction * 3),
        ("gainMapGamma", avifUnsignedFraction * 3),
        ("baseOffset", avifSignedFraction * 3),
        ("alternateOffset", avifSignedFraction * 3),
        ("baseHdrHeadroom", avifUnsignedFraction),
        ("alternateHdrHeadroom", avifUnsignedFraction),
        ("useBaseColorSpace", C.c_int),
        ("altICC_data", C.c_void_p),
        ("altICC_size", C.c_size_t),
        ("altColorPrimaries", C.c_uint16),
        ("altTransferCharacteristics", C.c_uint16),
        ("altMatrixCoefficients", C.c_uint16),
        ("altYUVRange", C.c_int),
        ("altDepth", C.c_uint32),
        ("altPlaneCount", C.c_uint32),
        ("altCLLI", avifContentLightLevelInformationBox),
    ]


assert C.sizeof(avifGainMap) == 192 and avifGainMap.altColorPrimaries.offset == 168 and avifGainMap.useBaseColorSpace.offset == 144


def rgb_format_has_alpha(fmt: int) -> bool:  # src/avif.c:675-679
    return fmt in (AVIF_RGB_FORMAT_RGBA, AVIF_RGB_FORMAT_ARGB, AVIF_RGB_FORMAT_BGRA, AVIF_RGB_FORMAT_ABGR,
                   AVIF_RGB_FORMAT_GRAYA, AVIF_RGB_FORMAT_AGRAY)


def rgb_format_is_gray(fmt: int) -> bool:  # src/avif.c:670-673
    return fmt in (AVIF_RGB_FORMAT_GRAY, AVIF_RGB_FORMAT_GRAYA, AVIF_RGB_FORMAT_AGRAY)


def rgb_format_channel_count(fmt: int) -> int:  # src/avif.c:681-690
    if fmt == AVIF_RGB_FORMAT_GRAY:
        return 1
    if fmt in (AVIF_RGB_FORMAT_GRAYA, AVIF_RGB_FORMAT_AGRAY):
        return 2
    return 4 if rgb_format_has_alpha(fmt) else 3


def rgb_pixel_size(fmt: int, depth: int) -> int:  # src/avif.c:692-698
    if fmt == AVIF_RGB_FORMAT_RGB_565:
        return 2
    return rgb_format_channel_count(fmt) * (2 if depth > 8 else 1)


def chroma_shifts(yuv_format: int) -> tuple[int, int]:  # src/avif.c:39-72
    return {AVIF_PIXEL_FORMAT_YUV444: (0, 0), AVIF_PIXEL_FORMAT_YUV422: (1, 0),
            AVIF_PIXEL_FORMAT_YUV420: (1, 1), AVIF_PIXEL_FORMAT_YUV400: (1, 1)}[yuv_format]


def chroma_dims(width: int, height: int, yuv_format: int) -> tuple[int, int]:  # src/avif.c:455-456
    sx, sy = chroma_shifts(yuv_format)
    return (width + sx) >> sx, (height + sy) >> sy


@dataclass
class HostYUV:
    """Host-side planar image: numpy planes + the avifImage struct pointing at them."""

    struct: avifImage
    planes: list  # [Y, U, V] numpy 2-D arrays (uint8, rows = rowBytes) or None
    alpha: Optional[np.ndarray]

    def plane_samples(self, idx: int) -> np.ndarray:
        """Plane `idx` (0..2, 3=alpha) as a (rows, cols) array of samples, padding stripped."""
        buf = self.alpha if idx == 3 else self.planes[idx]
        st = self.struct
        if idx in (0, 3):
            w, h = st.width, st.height
        else:
            w, h = chroma_dims(st.width, st.height, st.yuvFormat)
        if st.depth > 8:
            return buf.view(np.uint16)[:h, :w]
        return buf[:h, :w]


def make_yuv(width: int, height: int, depth: int, yuv_format: int, yuv_range: int = AVIF_RANGE_FULL,
             matrix: int = AVIF_MATRIX_COEFFICIENTS_BT601, with_alpha: bool = False, alpha_premultiplied: bool = False,
             row_pad: int = 0, color_primaries: int = 2, allocate: bool = True) -> HostYUV:
    """Host avifImage with tight (or padded) rows; mirrors avifImageCreate + avifImageAllocatePlanes
    (src/avif.c:137, :431-490)."""
    st = avifImage()
    st.width, st.height, st.depth = width, height, depth
    st.yuvFormat, st.yuvRange = yuv_format, yuv_range
    st.matrixCoefficients = matrix
    st.colorPrimaries = color_primaries
    st.transferCharacteristics = 2
    st.alphaPremultiplied = int(alpha_premultiplied)
    bps = 2 if depth > 8 else 1
    planes: list = [None, None, None]
    alpha = None
    if allocate:
        cw, ch = chroma_dims(width, height, yuv_format)
        dims = [(width, height), (cw, ch), (cw, ch)]
        for p in range(3):
            if p > 0 and yuv_format == AVIF_PIXEL_FORMAT_YUV400:
                continue
            w, h = dims[p]
            rb = w * bps + row_pad
            planes[p] = np.zeros((h, rb), dtype=np.uint8)
            st.yuvPlanes[p] = planes[p].ctypes.data
            st.yuvRowBytes[p] = rb
        if with_alpha:
            rb = width * bps + row_pad
            alpha = np.zeros((height, rb), dtype=np.uint8)
            st.alphaPlane = alpha.ctypes.data
            st.alphaRowBytes = rb
    return HostYUV(st, planes, alpha)


@dataclass
class HostRGB:
    struct: avifRGBImage
    pixels: Optional[np.ndarray]  # (height, rowBytes) uint8

    def channels(self) -> np.ndarray:
        """(height, width, nch) view of the interleaved channels (not for RGB_565)."""
        st = self.struct
        nch = rgb_format_channel_count(st.format)
        if st.depth > 8:
            return self.pixels.view(np.uint16)[:, : st.width * nch].reshape(st.height, st.width, nch)
        return self.pixels[:, : st.width * nch].reshape(st.height, st.width, nch)


def make_rgb(width: int, height: int, depth: int = 8, fmt: int = AVIF_RGB_FORMAT_RGBA,
             upsampling: int = AVIF_CHROMA_UPSAMPLING_AUTOMATIC, downsampling: int = AVIF_CHROMA_DOWNSAMPLING_AUTOMATIC,
             avoid_libyuv: bool = True, ignore_alpha: bool = False, alpha_premultiplied: bool = False,
             is_float: bool = False, max_threads: int = 1, row_pad: int = 0, fill: int = 0,
             allocate: bool = True) -> HostRGB:
    """avifRGBImageSetDefaults + avifRGBImageAllocatePixels (src/avif.c:700-737).  avoid_libyuv defaults to
    True here because parity is stated against the reference's built-in (float) path unless a test says otherwise."""
    st = avifRGBImage()
    st.width, st.height, st.depth, st.format = width, height, depth, fmt
    st.chromaUpsampling, st.chromaDownsampling = upsampling, downsampling
    st.avoidLibYUV = int(avoid_libyuv)
    st.ignoreAlpha = int(ignore_alpha)
    st.alphaPremultiplied = int(alpha_premultiplied)
    st.isFloat = int(is_float)
    st.maxThreads = max_threads
    pixels = None
    if allocate:
        rb = width * rgb_pixel_size(fmt, depth) + row_pad
        pixels = np.full((height, rb), fill, dtype=np.uint8)
        st.pixels = pixels.ctypes.data
        st.rowBytes = rb
    return HostRGB(st, pixels)
