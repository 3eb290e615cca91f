"""Seam A: the LD_PRELOAD interposer (integration/avif_preload_hip.c -> libavif_amd/csrc/libavifhip_preload.so) under a
stand-in application linked against a SHARED libavif (tests/tools/preload_probe.c, linked to the reference compiled from
its own sources).  Without a GPU every call must be forwarded to the real libavif; with one, the same bytes must come
out of the HIP kernels (the interposed libavif has no libyuv, so the interposer pins the fp32 arithmetic)."""
import os
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PROBE = ROOT / "oracle" / "_ref" / "preload_probe"
PRELOAD = ROOT / "libavif_amd" / "csrc" / "libavifhip_preload.so"

needs_probe = pytest.mark.skipif(not PROBE.exists() or not PRELOAD.exists(), reason="oracle/_ref/preload_probe not built (needs /root/reference at build time)")


def _run(tmp_path, name, preload, extra_env=None):
    out = tmp_path / name
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    if preload:
        env["LD_PRELOAD"] = os.fspath(PRELOAD)
    env.update(extra_env or {})
    proc = subprocess.run([os.fspath(PROBE), os.fspath(out)], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    m = re.search(r"results (\d+) (\d+) (\d+) launches (\d+)", proc.stdout)
    assert m, proc.stdout
    return [int(x) for x in m.groups()], out.read_bytes()


@needs_probe
def test_interposer_exports_the_public_symbols():
    import ctypes

    lib = ctypes.CDLL(os.fspath(PRELOAD))
    for sym in ("avifImageYUVToRGB", "avifImageRGBToYUV", "avifRGBImagePremultiplyAlpha", "avifRGBImageUnpremultiplyAlpha", "avifRGBImageApplyGainMap",
                "avifRGBImageComputeGainMap"):
        assert hasattr(lib, sym), sym


@needs_probe
def test_interposer_forwards_when_it_declines(tmp_path):
    """A size threshold nothing reaches: every call goes to the real libavif, bytes unchanged, no kernel launched."""
    plain, want = _run(tmp_path, "plain.bin", preload=False)
    assert plain == [0, 0, 0, 0]
    got_r, got = _run(tmp_path, "fwd.bin", preload=True, extra_env={"AVIFHIP_MIN_PIXELS": str(1 << 40)})
    assert got_r == [0, 0, 0, 0]
    assert got == want


@pytest.mark.gpu
def test_interposer_serves_the_calls_from_the_gpu(hip, tmp_path):
    # a GPU test: the prebuilt probe travels with the snapshot; its absence must fail, not skip
    assert PROBE.exists() and PRELOAD.exists(), "oracle/_ref/preload_probe or libavifhip_preload.so is missing: build them where /root/reference exists"
    _, want = _run(tmp_path, "plain.bin", preload=False)
    got_r, got = _run(tmp_path, "gpu.bin", preload=True, extra_env={"AVIFHIP_MIN_PIXELS": "0"})
    assert got_r[:3] == [0, 0, 0]
    assert got_r[3] >= 6, "the three conversions and the tone mapping (rescale, gain-map conversion, apply) must have run as HIP kernels"
    assert got == want
