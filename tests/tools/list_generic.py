"""Which configurations of the parity sweeps still reach the universal (one lane per pixel) kernels, and why -- the input of
tests/test_gpu_parity.py::test_universal_kernels_serve_only_the_enumerated_rest.  One line per (reason, count, example)."""
import collections
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import harness as H  # noqa: E402
from libavif_amd import abi, native  # noqa: E402

TILED = [(512, 16), (300, 21), (256, 8), (777, 35), (1027, 18)]
lib = native.load()


def reason(c, arith):
    """The documented reasons, from the case's own properties."""
    r = []
    if c.rgb_format == abi.AVIF_RGB_FORMAT_RGB_565:
        r.append("565")
    if c.ignore_alpha and abi.rgb_format_has_alpha(c.rgb_format):
        r.append("ignoreAlpha")
    if c.matrix in (8, 16, 17):
        r.append("ycgco")
    if c.matrix == 0:
        r.append("identity")
    if c.row_pad:
        r.append("pad")
    if c.is_float:
        r.append("f16")
    if c.alpha and (c.image_premultiplied != c.rgb_premultiplied):
        r.append("alphamul")
    r.append(f"nch{abi.rgb_format_channel_count(c.rgb_format)}")
    r.append(f"rgb{c.rgb_depth}")
    return arith + ":" + ",".join(r)


for arith, setting, avoid in (("fp32", 1, True), ("auto", 0, False)):
    lib.avifhipSetArithmetic(setting)
    lib.avifhipSetTiledKernels(1)
    be = H.HipDeviceBackend()
    seen = collections.OrderedDict()
    total = 0
    import dataclasses
    for c in H.y2r_sweep(TILED, n_random=600, seed=5):
        c = dataclasses.replace(c, avoid_libyuv=avoid)
        res, _ = H.run_y2r(be, c)
        if res != 0:
            continue
        total += 1
        k = native.last_kernel()
        if "generic" in k:
            key = reason(c, arith)
            n, ex = seen.get(key, (0, c.ident()))
            seen[key] = (n + 1, ex)
    print(f"== {arith}: {sum(n for n, _ in seen.values())} of {total} conversions through the universal kernels")
    for key, (n, ex) in sorted(seen.items(), key=lambda kv: -kv[1][0]):
        print(f"{n:5d}  {key:60s} e.g. {ex}")
lib.avifhipSetArithmetic(0)
