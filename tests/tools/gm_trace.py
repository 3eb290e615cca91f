"""Where the time of the host-resident gain-map calls goes: four calls each of avifhipRGBImageComputeGainMap and avifhipRGBImageApplyGainMap on a 4K
job, wall clock per call; with AVIFHIP_GAINMAP_TRACE=1 the library prints the phases of every call (uploads, passes, downloads) on stderr.
`AVIFHIP_GAINMAP_TRACE=1 python tests/tools/gm_trace.py`"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from libavif_amd import abi, native, synth
import gainmap_cases as G
lib = native.load()
lib.avifhipSetArithmetic(0)
c = G.ComputeCase(3840, 2160, alt_primaries=9, seed=3)
base, alt = G.make_compute_inputs(c)
gm, img = G.make_compute_gain_map(c)
diag = abi.avifDiagnostics()
for k in range(4):
    t0 = time.perf_counter()
    native.check(lib.avifhipRGBImageComputeGainMap(base.struct, 1, 13, alt.struct, 9, 16, C.byref(gm), C.byref(diag)))
    print("compute call ms", round((time.perf_counter() - t0) * 1e3, 3), flush=True)
W, H = 3840, 2160
b = abi.make_rgb(W, H, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False); synth.fill_rgb(b, 1)
gimg = abi.make_yuv(W, H, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 6); synth.fill_yuv(gimg, 2)
g2 = abi.avifGainMap()
for i in range(3):
    g2.gainMapMin[i].n, g2.gainMapMin[i].d = 0, 1; g2.gainMapMax[i].n, g2.gainMapMax[i].d = 3, 1; g2.gainMapGamma[i].n, g2.gainMapGamma[i].d = 1, 1
    g2.baseOffset[i].n, g2.baseOffset[i].d = 1, 64; g2.alternateOffset[i].n, g2.alternateOffset[i].d = 1, 64
g2.baseHdrHeadroom.n, g2.baseHdrHeadroom.d, g2.alternateHdrHeadroom.n, g2.alternateHdrHeadroom.d = 0, 1, 3, 1
g2.useBaseColorSpace = 1; g2.image = C.pointer(gimg.struct)
out = abi.make_rgb(W, H, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False, allocate=False)
clli = abi.avifContentLightLevelInformationBox()
for k in range(4):
    t0 = time.perf_counter()
    native.check(lib.avifhipRGBImageApplyGainMap(b.struct, 1, 13, C.byref(g2), 3.0, 9, 16, out.struct, C.byref(clli), C.byref(diag)))
    print("apply call ms", round((time.perf_counter() - t0) * 1e3, 3), flush=True)
