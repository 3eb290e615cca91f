"""ctypes binding of libavifhip.so (include/avifhip.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing the
binding raises; if no GPU is present the conversion entry points return an error code and
``check`` raises with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .abi import (AVIF_RESULT_OK, avifContentLightLevelInformationBox, avifCropRect, avifDiagnostics, avifGainMap, avifImage,
                  avifRGBImage)

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libavifhip.so"

# every symbol include/avifhip.h declares (tests assert the library exports each of them)
EXPORTED_SYMBOLS = [
    "avifhipImageYUVToRGB", "avifhipImageRGBToYUV", "avifhipRGBImagePremultiplyAlpha", "avifhipRGBImageUnpremultiplyAlpha",
    "avifhipImageYUVToRGBAsync", "avifhipImageRGBToYUVAsync", "avifhipRGBImagePremultiplyAlphaAsync",
    "avifhipRGBImageUnpremultiplyAlphaAsync", "avifhipImageYUVToRGBRectAsync", "avifhipImageYUVToRGBBatchAsync", "avifhipImageRGBToYUVBatchAsync",
    "avifhipTimeRGBToYUVBatchCycle", "avifhipTimeStreamCeilingRGBToYUVBatchCycle",
    "avifhipLimitedToFullY", "avifhipLimitedToFullUV", "avifhipFullToLimitedY", "avifhipFullToLimitedUV",
    "avifhipSetArithmetic", "avifhipGetArithmetic", "avifhipSetTiledKernels", "avifhipSetDevice", "avifhipDeviceCount",
    "avifhipSynchronize", "avifhipLastError", "avifhipLastKernel", "avifhipVersion", "avifhipDeviceAlloc", "avifhipDeviceFree",
    "avifhipCopyToDevice", "avifhipCopyToHost", "avifhipDeviceMemset", "avifhipTimeYUVToRGB", "avifhipTimeRGBToYUV",
    "avifhipSynthFill", "avifhipStreamCreate", "avifhipStreamDestroy", "avifhipSetTuning", "avifhipTimeYUVToRGBCycle", "avifhipTimeStreamCeiling", "avifhipTimeStreamCeilingRGBToYUV", "avifhipTimeStreamCeilingBatch", "avifhipTimeStreamCeilingScale", "avifhipTimeRGBToYUVCycle", "avifhipTimeYUVToRGBBatch", "avifhipTimeYUVToRGBBatchCycle", "avifhipTimeStreamCeilingBatchCycle", "avifhipTimeGridYUVToRGB", "avifhipImageYUVToRGBTransformedAsync", "avifhipGridYUVToRGBTransformedAsync", "avifhipY4MFrameBytes", "avifhipImagePackY4MFrameAsync", "avifhipRGBImagePackPNGRowsAsync", "avifhipImageYUVToRGBRects", "avifhipPlanRectTransfers", "avifhipLastTransferBytes", "avifhipImageYUVToRGBColorOnly", "avifhipImageYUVToRGBHook", "avifhipRGBImageToF16", "avifhipLaunchCount", "avifhipTableUploadCount", "avifhipCalcYUVCoefficients",
    "avifhipExplainYUVToRGB", "avifhipExplainRGBToYUV", "avifhipGridYUVToRGBAsync", "avifhipRGBImageTransformAsync", "avifhipImageScale", "avifhipImageScaleAsync", "avifhipImageApplyOperationsAsync",
    "avifhipSetDeviceSet", "avifhipSetFarmMinSharePixels", "avifhipGetDeviceSet", "avifhipPlanFarmRows", "avifhipLastFarmWorkers", "avifhipLastFarmTransferBytes",
    "avifhipRGBImageApplyGainMap", "avifhipRGBImageApplyGainMapAsync", "avifhipTimeRGBImageApplyGainMap", "avifhipImageApplyGainMap", "avifhipRGBImageComputeGainMap", "avifhipImageComputeGainMap", "avifhipRGBImageComputeGainMapAsync", "avifhipTimeRGBImageComputeGainMap", "avifhipSetExactLightLevels",
]

class avifSampleTransformToken(C.Structure):
    """include/avif/internal.h:222-228"""
    _fields_ = [("type", C.c_int), ("constant", C.c_int32), ("inputImageItemIndex", C.c_uint8)]


class avifhipGrid(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("columns", C.c_uint32), ("outputWidth", C.c_uint32), ("outputHeight", C.c_uint32)]


_lib = None


class AvifHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads libavifhip.so (once). Raises if it has not been built: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise AvifHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C {CSRC}`; there is no CPU fallback")
    lib = C.CDLL(os.fspath(LIB_PATH))
    P_IMG, P_RGB, P_RECT = C.POINTER(avifImage), C.POINTER(avifRGBImage), C.POINTER(avifCropRect)
    vp, i32, u32 = C.c_void_p, C.c_int, C.c_uint32
    sigs = {
        "avifhipImageYUVToRGB": (i32, [P_IMG, P_RGB]),
        "avifhipImageRGBToYUV": (i32, [P_IMG, P_RGB]),
        "avifhipRGBImagePremultiplyAlpha": (i32, [P_RGB]),
        "avifhipRGBImageUnpremultiplyAlpha": (i32, [P_RGB]),
        "avifhipImageYUVToRGBAsync": (i32, [P_IMG, P_RGB, vp]),
        "avifhipImageRGBToYUVAsync": (i32, [P_IMG, P_RGB, vp]),
        "avifhipRGBImagePremultiplyAlphaAsync": (i32, [P_RGB, vp]),
        "avifhipRGBImageUnpremultiplyAlphaAsync": (i32, [P_RGB, vp]),
        "avifhipImageYUVToRGBRectAsync": (i32, [P_IMG, P_RGB, P_RECT, vp]),
        "avifhipImageYUVToRGBBatchAsync": (i32, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), P_RECT, vp]),
        "avifhipLimitedToFullY": (i32, [u32, i32]),
        "avifhipLimitedToFullUV": (i32, [u32, i32]),
        "avifhipFullToLimitedY": (i32, [u32, i32]),
        "avifhipFullToLimitedUV": (i32, [u32, i32]),
        "avifhipSetArithmetic": (None, [i32]),
        "avifhipGetArithmetic": (i32, []),
        "avifhipSetTiledKernels": (None, [i32]),
        "avifhipSetDevice": (i32, [i32]),
        "avifhipDeviceCount": (i32, []),
        "avifhipSynchronize": (i32, [vp]),
        "avifhipLastError": (C.c_char_p, []),
        "avifhipLastKernel": (C.c_char_p, []),
        "avifhipVersion": (C.c_char_p, []),
        "avifhipDeviceAlloc": (vp, [C.c_size_t]),
        "avifhipDeviceFree": (None, [vp]),
        "avifhipCopyToDevice": (i32, [vp, vp, C.c_size_t]),
        "avifhipCopyToHost": (i32, [vp, vp, C.c_size_t]),
        "avifhipDeviceMemset": (i32, [vp, i32, C.c_size_t]),
        "avifhipTimeYUVToRGB": (C.c_double, [P_IMG, P_RGB, i32, i32, vp]),
        "avifhipTimeRGBToYUV": (C.c_double, [P_IMG, P_RGB, i32, i32, vp]),
        "avifhipTimeRGBToYUVCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), i32, i32, vp]),
        "avifhipImageRGBToYUVBatchAsync": (i32, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), vp]),
        "avifhipTimeRGBToYUVBatchCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), u32, i32, i32, vp]),
        "avifhipTimeStreamCeilingRGBToYUVBatchCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), u32, i32, i32, vp]),
        "avifhipTimeYUVToRGBBatch": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), P_RECT, i32, i32, vp]),
        "avifhipTimeYUVToRGBBatchCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), u32, i32, i32, vp]),
        "avifhipTimeStreamCeilingBatchCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), u32, i32, i32, vp]),
        "avifhipTimeGridYUVToRGB": (C.c_double, [C.POINTER(avifhipGrid), C.POINTER(P_IMG), C.POINTER(P_IMG), i32, P_RGB, i32, i32, vp]),
        "avifhipSynthFill": (u32, [u32, vp, u32, u32, u32, u32, u32, u32]),
        "avifhipStreamCreate": (vp, []),
        "avifhipStreamDestroy": (None, [vp]),
        "avifhipSetTuning": (None, [u32]),
        "avifhipImageYUVToRGBColorOnly": (i32, [P_IMG, P_RGB, i32]),
        "avifhipImageYUVToRGBHook": (i32, [P_IMG, P_RGB, i32, C.POINTER(C.c_uint32)]),
        "avifhipRGBImageToF16": (i32, [P_RGB]),
        "avifhipLaunchCount": (C.c_uint64, []),
        "avifhipTableUploadCount": (C.c_uint64, []),
        "avifhipTimeYUVToRGBCycle": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), i32, i32, vp]),
        "avifhipImageYUVToRGBTransformedAsync": (i32, [P_IMG, P_RGB, P_RECT, i32, C.c_uint8, i32, C.c_uint8, vp]),
        "avifhipGridYUVToRGBTransformedAsync": (i32, [C.POINTER(avifhipGrid), C.POINTER(P_IMG), C.POINTER(P_IMG), i32, P_RGB, P_RECT, i32, C.c_uint8, i32, C.c_uint8, vp]),
        "avifhipY4MFrameBytes": (C.c_size_t, [P_IMG, i32]),
        "avifhipImagePackY4MFrameAsync": (i32, [P_IMG, i32, vp, vp]),
        "avifhipRGBImagePackPNGRowsAsync": (i32, [P_RGB, vp, vp]),
        "avifhipImageYUVToRGBRects": (i32, [P_IMG, P_RGB, P_RECT, u32]),
        "avifhipPlanRectTransfers": (i32, [P_IMG, P_RGB, P_RECT, u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "avifhipLastTransferBytes": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "avifhipTimeStreamCeiling": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), i32, i32, vp]),
        "avifhipTimeStreamCeilingRGBToYUV": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), i32, i32, vp]),
        "avifhipTimeStreamCeilingBatch": (C.c_double, [u32, C.POINTER(P_IMG), C.POINTER(P_RGB), i32, i32, vp]),
        "avifhipTimeStreamCeilingScale": (C.c_double, [P_IMG, P_IMG, i32, i32, vp]),
        "avifhipSetDeviceSet": (i32, [C.POINTER(C.c_int), u32]),
        "avifhipGetDeviceSet": (u32, [C.POINTER(C.c_int), u32]),
        "avifhipSetFarmMinSharePixels": (None, [C.c_uint64]),
        "avifhipPlanFarmRows": (i32, [u32, u32, u32, P_RECT, u32, C.POINTER(u32)]),
        "avifhipLastFarmWorkers": (u32, []),
        "avifhipLastFarmTransferBytes": (i32, [u32, C.POINTER(C.c_int), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "avifhipExplainYUVToRGB": (i32, [P_IMG, P_RGB, C.c_char_p, C.c_size_t]),
        "avifhipExplainRGBToYUV": (i32, [P_IMG, P_RGB, C.c_char_p, C.c_size_t]),
        "avifhipRGBImageTransformAsync": (i32, [P_RGB, P_RGB, P_RECT, i32, C.c_uint8, i32, C.c_uint8, vp]),
        "avifhipImageApplyOperationsAsync": (i32, [P_IMG, i32, u32, C.POINTER(avifSampleTransformToken), C.c_uint8, C.POINTER(P_IMG), u32, vp]),
        "avifhipImageScale": (i32, [P_IMG, u32, u32]),
        "avifhipImageScaleAsync": (i32, [P_IMG, P_IMG, vp]),
        "avifhipGridYUVToRGBAsync": (i32, [C.POINTER(avifhipGrid), C.POINTER(P_IMG), C.POINTER(P_IMG), i32, P_RGB, vp]),
        "avifhipRGBImageApplyGainMap": (i32, [P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, P_RGB,
                                              C.POINTER(avifContentLightLevelInformationBox), C.POINTER(avifDiagnostics)]),
        "avifhipRGBImageApplyGainMapAsync": (i32, [P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, P_RGB,
                                                   C.POINTER(avifContentLightLevelInformationBox), C.POINTER(avifDiagnostics), vp]),
        "avifhipTimeRGBImageApplyGainMap": (C.c_double, [P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, P_RGB,
                                                         i32, i32, vp]),
        "avifhipRGBImageComputeGainMap": (i32, [P_RGB, C.c_uint16, C.c_uint16, P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.POINTER(avifDiagnostics)]),
        "avifhipRGBImageComputeGainMapAsync": (i32, [P_RGB, C.c_uint16, C.c_uint16, P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), C.POINTER(avifDiagnostics), vp]),
        "avifhipSetExactLightLevels": (None, [i32]),
        "avifhipTimeRGBImageComputeGainMap": (C.c_double, [P_RGB, C.c_uint16, C.c_uint16, P_RGB, C.c_uint16, C.c_uint16, C.POINTER(avifGainMap), i32, i32, vp]),
        "avifhipImageComputeGainMap": (i32, [P_IMG, P_IMG, C.POINTER(avifGainMap), C.POINTER(avifDiagnostics)]),
        "avifhipImageApplyGainMap": (i32, [P_IMG, C.POINTER(avifGainMap), C.c_float, C.c_uint16, C.c_uint16, P_RGB,
                                           C.POINTER(avifContentLightLevelInformationBox), C.POINTER(avifDiagnostics)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(result: int, what: str = "avifhip call") -> None:
    if result != AVIF_RESULT_OK:
        msg = load().avifhipLastError().decode() or "no detail"
        raise AvifHipError(f"{what} failed with avifResult {result}: {msg}")


def last_kernel() -> str:
    return load().avifhipLastKernel().decode()


def explain_y2r(image_struct, rgb_struct):
    """(avifResult, {key: value}) of avifhipExplainYUVToRGB: host-only view of the plan layer's decisions."""
    buf = C.create_string_buffer(256)
    res = load().avifhipExplainYUVToRGB(image_struct, rgb_struct, buf, len(buf))
    return res, dict(kv.split("=", 1) for kv in buf.value.decode().split())


def explain_r2y(image_struct, rgb_struct):
    buf = C.create_string_buffer(256)
    res = load().avifhipExplainRGBToYUV(image_struct, rgb_struct, buf, len(buf))
    return res, dict(kv.split("=", 1) for kv in buf.value.decode().split())


def device_count() -> int:
    return int(load().avifhipDeviceCount())
