"""Test-only access to the checkers under oracle/ (never imported by the product package).

  oracle()  -> ctypes handle of oracle/liboracle.so (plain-C restatement, built by oracle/Makefile)
  ref()     -> ctypes handle of oracle/_ref/libavif_ref.so (the reference compiled from its own sources),
               or None when it has not been built
  pillow()  -> ctypes handle of Pillow's bundled libavif (1.4.1 + libyuv 1922), the only libyuv-enabled
               binary available offline, or None
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess
from pathlib import Path

from libavif_amd.abi import (avifContentLightLevelInformationBox, avifCropRect, avifDiagnostics, avifGainMap, avifImage,
                              avifRGBImage)

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"

_P_IMG, _P_RGB, _P_RECT = C.POINTER(avifImage), C.POINTER(avifRGBImage), C.POINTER(avifCropRect)
_cache: dict = {}


def _ensure_built() -> None:
    so = ORACLE_DIR / "liboracle.so"
    srcs = [ORACLE_DIR / "reformat_oracle.c", ORACLE_DIR / "libyuv_oracle.c", ORACLE_DIR / "scale_oracle.c", ORACLE_DIR / "gainmap_oracle.c", ORACLE_DIR / "pack_oracle.c",
            ORACLE_DIR / "reformat_oracle.h", ORACLE_DIR / "oracle_backend.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", os.fspath(ORACLE_DIR), "liboracle.so"], check=True, capture_output=True)


def oracle() -> C.CDLL:
    if "oracle" not in _cache:
        _ensure_built()
        lib = C.CDLL(os.fspath(ORACLE_DIR / "liboracle.so"))
        for name in ("oracleImageYUVToRGB", "oracleLibyuvImageYUVToRGB"):
            getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [_P_IMG, _P_RGB]
        for name in ("oracleImageRGBToYUV", "oracleLibyuvImageRGBToYUV"):
            getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [_P_IMG, _P_RGB]
        for name in ("oracleRGBImagePremultiplyAlpha", "oracleRGBImageUnpremultiplyAlpha",
                     "oracleLibyuvRGBImagePremultiplyAlpha", "oracleLibyuvRGBImageUnpremultiplyAlpha"):
            getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [_P_RGB]
        lib.oracleLibyuvHookYUVToRGB.restype, lib.oracleLibyuvHookYUVToRGB.argtypes = C.c_int, [_P_IMG, _P_RGB, C.c_int, C.POINTER(C.c_int)]
        lib.oracleLibyuvHookRGBToYUV.restype, lib.oracleLibyuvHookRGBToYUV.argtypes = C.c_int, [_P_IMG, _P_RGB]
        lib.oracleImageApplyOperations.restype = C.c_int
        lib.oracleImageApplyOperations.argtypes = [_P_IMG, C.c_int, C.c_uint32, C.c_void_p, C.c_uint8, C.POINTER(_P_IMG), C.c_uint32]
        lib.oracleImageScale.restype, lib.oracleImageScale.argtypes = C.c_int, [_P_IMG, C.c_uint32, C.c_uint32]
        lib.oracleRGBImageTransform.restype = C.c_int
        lib.oracleRGBImageTransform.argtypes = [_P_RGB, _P_RGB, _P_RECT, C.c_int, C.c_uint8, C.c_int, C.c_uint8]
        lib.oracleGridYUVToRGB.restype = C.c_int
        lib.oracleGridYUVToRGB.argtypes = [C.c_void_p, C.POINTER(_P_IMG), C.POINTER(_P_IMG), C.c_int, _P_RGB, C.c_int]
        lib.oracleImageYUVToRGBRect.restype, lib.oracleImageYUVToRGBRect.argtypes = C.c_int, [_P_IMG, _P_RGB, _P_RECT]
        for name in ("oracleLimitedToFullY", "oracleLimitedToFullUV", "oracleFullToLimitedY", "oracleFullToLimitedUV"):
            getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [C.c_uint32, C.c_int]
        lib.oracleRGBImageApplyGainMap.restype = C.c_int
        lib.oracleRGBImageApplyGainMap.argtypes = [_P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, _P_RGB,
                                                   C.POINTER(avifContentLightLevelInformationBox), C.c_int]
        lib.oracleRGBImageComputeGainMap.restype = C.c_int
        lib.oracleRGBImageComputeGainMap.argtypes = [_P_RGB, C.c_uint16, C.c_uint16, _P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_int]
        lib.oraclePackY4MFrame.restype, lib.oraclePackY4MFrame.argtypes = C.c_size_t, [_P_IMG, C.c_int, C.c_void_p]
        lib.oraclePackPNGRows.restype, lib.oraclePackPNGRows.argtypes = C.c_size_t, [_P_RGB, C.c_uint32, C.c_void_p]
        lib.oracleTransferFunction.restype, lib.oracleTransferFunction.argtypes = C.c_float, [C.c_int, C.c_int, C.c_float]
        lib.oracleColorPrimariesComputeRGBToRGBMatrix.restype = C.c_int
        lib.oracleColorPrimariesComputeRGBToRGBMatrix.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double * 9)]
        _cache["oracle"] = lib
    return _cache["oracle"]


def _bind_libavif(lib: C.CDLL) -> C.CDLL:
    lib.avifImageYUVToRGB.restype, lib.avifImageYUVToRGB.argtypes = C.c_int, [_P_IMG, _P_RGB]
    lib.avifImageRGBToYUV.restype, lib.avifImageRGBToYUV.argtypes = C.c_int, [_P_IMG, _P_RGB]
    lib.avifRGBImagePremultiplyAlpha.restype, lib.avifRGBImagePremultiplyAlpha.argtypes = C.c_int, [_P_RGB]
    lib.avifRGBImageUnpremultiplyAlpha.restype, lib.avifRGBImageUnpremultiplyAlpha.argtypes = C.c_int, [_P_RGB]
    for name in ("avifLimitedToFullY", "avifLimitedToFullUV", "avifFullToLimitedY", "avifFullToLimitedUV"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [C.c_uint32, C.c_int]
    lib.avifLibYUVVersion.restype = C.c_uint
    # internal.h functions: exported by the from-source build (default visibility), not by a packaged shared libavif
    if hasattr(lib, "avifImageSetViewRect"):
        lib.avifImageSetViewRect.restype, lib.avifImageSetViewRect.argtypes = C.c_int, [_P_IMG, _P_IMG, _P_RECT]
    if hasattr(lib, "avifImageApplyOperations"):
        lib.avifImageApplyOperations.restype = C.c_int
        lib.avifImageApplyOperations.argtypes = [_P_IMG, C.c_int, C.c_uint32, C.c_void_p, C.c_uint8, C.POINTER(_P_IMG), C.c_uint32]
    if hasattr(lib, "avifImageScale"):
        lib.avifImageScale.restype, lib.avifImageScale.argtypes = C.c_int, [_P_IMG, C.c_uint32, C.c_uint32, C.c_void_p]
    if hasattr(lib, "avifRGBImageApplyGainMap"):
        lib.avifRGBImageApplyGainMap.restype = C.c_int
        lib.avifRGBImageApplyGainMap.argtypes = [_P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, _P_RGB,
                                                 C.POINTER(avifContentLightLevelInformationBox), C.POINTER(avifDiagnostics)]
    if hasattr(lib, "avifRGBImageComputeGainMap"):
        lib.avifRGBImageComputeGainMap.restype = C.c_int
        lib.avifRGBImageComputeGainMap.argtypes = [_P_RGB, C.c_uint16, C.c_uint16, _P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap),
                                                   C.POINTER(avifDiagnostics)]
    if hasattr(lib, "avifColorPrimariesComputeRGBToRGBMatrix"):
        lib.avifColorPrimariesComputeRGBToRGBMatrix.restype = C.c_int
        lib.avifColorPrimariesComputeRGBToRGBMatrix.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(C.c_double * 9)]
    if hasattr(lib, "avifImageCopySamples"):
        lib.avifImageCopySamples.restype, lib.avifImageCopySamples.argtypes = None, [_P_IMG, _P_IMG, C.c_uint32]
    return lib


def ref():
    if "ref" not in _cache:
        path = ORACLE_DIR / "_ref" / "libavif_ref.so"
        _cache["ref"] = _bind_libavif(C.CDLL(os.fspath(path), mode=os.RTLD_LOCAL)) if path.exists() else None
    return _cache["ref"]


def util_ref():
    """apps/shared/avifutil.c of the reference, compiled from where it lies (oracle/Makefile target utilref), or None."""
    if "util_ref" not in _cache:
        path = ORACLE_DIR / "_ref" / "libavifutil_ref.so"
        lib = None
        if path.exists():
            lib = C.CDLL(os.fspath(path), mode=os.RTLD_LOCAL)
            lib.avifRGBImageSetViewRect.restype, lib.avifRGBImageSetViewRect.argtypes = None, [_P_RGB, _P_RGB, _P_RECT]
            lib.avifRGBImageRotate.restype, lib.avifRGBImageRotate.argtypes = C.c_int, [_P_RGB, _P_RGB, C.POINTER(C.c_uint8)]
            lib.avifRGBImageMirror.restype, lib.avifRGBImageMirror.argtypes = C.c_int, [_P_RGB, C.POINTER(C.c_uint8)]
            if hasattr(lib, "y4mWrite"):  # apps/shared/y4m.c, in the same library since round 2
                lib.y4mWrite.restype, lib.y4mWrite.argtypes = C.c_int, [C.c_char_p, _P_IMG]
        _cache["util_ref"] = lib
    return _cache["util_ref"]


def pillow():
    """The libyuv-enabled libavif binary the integer-path oracle is pinned against: Pillow's bundled libavif 1.4.1 + libyuv 1922 -- or, when
    AVIFHIP_LIBYUV_BINARY names one, a libavif built with another libyuv (tests/tools/repin_libyuv.sh builds the reference with the pinned
    1949 where a libyuv checkout exists: the re-pin is then `AVIFHIP_LIBYUV_BINARY=... pytest tests/test_libyuv_oracle.py`)."""
    if "pillow" not in _cache:
        override = os.environ.get("AVIFHIP_LIBYUV_BINARY")
        hits = [override] if override else sorted(glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libavif-*.so*"))
        lib = None
        if hits:
            try:
                lib = _bind_libavif(C.CDLL(hits[0], mode=os.RTLD_LOCAL))
                if lib.avifLibYUVVersion() == 0:
                    lib = None
            except OSError:
                lib = None
        _cache["pillow"] = lib
    return _cache["pillow"]
