"""The drop-in boundary without a GPU: libavifhip.so loads and exports every symbol include/avifhip.h declares, the
struct mirrors have libavif's layout, host-only entry points work, and conversions fail loudly (never fall back to a
CPU path) when no GPU is visible."""
import ctypes as C
import re
from pathlib import Path

from libavif_amd import abi, native

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "avifhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"AVIFHIP_API\s+[^;(]*?\b(avifhip\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = native.load()
    declared = _declared_symbols()
    assert len(declared) >= 35, declared
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(native.EXPORTED_SYMBOLS) == declared, set(native.EXPORTED_SYMBOLS) ^ set(declared)


def test_struct_layout_matches_libavif():
    # include/avif/avif.h:777-851, :996-1018 (offsets probed against the reference header)
    assert C.sizeof(abi.avifRGBImage) == 64
    assert abi.avifRGBImage.pixels.offset == 48 and abi.avifRGBImage.rowBytes.offset == 56
    assert abi.avifImage.yuvPlanes.offset == 24 and abi.avifImage.yuvRowBytes.offset == 48
    assert abi.avifImage.alphaPlane.offset == 64 and abi.avifImage.matrixCoefficients.offset == 108
    assert C.sizeof(abi.avifImage) >= 112


def test_host_only_entry_points():
    lib = native.load()
    assert b"avifhip" in lib.avifhipVersion()
    # src/reformat.c:1778-1840 on a few points (the full table is pinned in test_oracle_vs_ref.py)
    assert lib.avifhipLimitedToFullY(8, 16) == 0 and lib.avifhipLimitedToFullY(8, 235) == 255
    assert lib.avifhipFullToLimitedUV(10, 1023) == 960 and lib.avifhipFullToLimitedY(12, 0) == 256
    img = abi.make_yuv(4, 4, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
    kr, kg, kb = C.c_float(), C.c_float(), C.c_float()
    lib.avifhipCalcYUVCoefficients.argtypes = [C.POINTER(abi.avifImage), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.avifhipCalcYUVCoefficients(img.struct, C.byref(kr), C.byref(kg), C.byref(kb))
    assert abs(kr.value - 0.2126) < 1e-7 and abs(kb.value - 0.0722) < 1e-7 and abs(kg.value - 0.7152) < 1e-6


def test_limited_full_exports_over_the_whole_code_range():
    """avifhip{LimitedToFull,FullToLimited}{Y,UV} (host-only exports; src/reformat.c:1750-1840) against the reference compiled here and the
    oracle, for every depth the reference's switch knows (8 / 10 / 12), one it does not (9: the value comes back unchanged), and every code of
    the depth plus a margin either side (the reference clamps: out-of-range inputs included)."""
    import oracle_lib

    lib, o, r = native.load(), oracle_lib.oracle(), oracle_lib.ref()
    pairs = [("avifhipLimitedToFullY", "oracleLimitedToFullY", "avifLimitedToFullY"), ("avifhipLimitedToFullUV", "oracleLimitedToFullUV", "avifLimitedToFullUV"),
             ("avifhipFullToLimitedY", "oracleFullToLimitedY", "avifFullToLimitedY"), ("avifhipFullToLimitedUV", "oracleFullToLimitedUV", "avifFullToLimitedUV")]
    checked = 0
    for depth in (8, 10, 12, 9, 16):
        for v in range(-70, (1 << min(depth, 12)) + 70):
            for mine, orc, ref in pairs:
                got, want = getattr(lib, mine)(depth, v), getattr(o, orc)(depth, v)
                assert got == want, (mine, depth, v, got, want)
                if r is not None:
                    assert got == getattr(r, ref)(depth, v), (mine, depth, v)
                checked += 1
    assert checked > 4 * (256 + 1024 + 4096)


def test_no_silent_cpu_fallback_without_gpu():
    lib = native.load()
    if lib.avifhipDeviceCount() > 0:
        return  # on the GPU box the parity tests cover the real path
    img = abi.make_yuv(64, 8, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
    rgb = abi.make_rgb(64, 8, 8, abi.AVIF_RGB_FORMAT_RGBA, fill=0xA5)
    res = lib.avifhipImageYUVToRGB(img.struct, rgb.struct)
    assert res != abi.AVIF_RESULT_OK
    assert b"HIP device" in lib.avifhipLastError()
    assert (rgb.pixels == 0xA5).all()  # nothing was computed behind the caller's back
    # argument errors are still the reference's (checked before the device is touched)
    rgb.struct.depth = 9
    assert lib.avifhipImageYUVToRGB(img.struct, rgb.struct) == abi.AVIF_RESULT_REFORMAT_FAILED
