"""The tiled kernels replace IEEE division by plan constants with fma(x, hi, x*lo) (libavif_amd/csrc/exactdiv.h).
That is only legitimate for divisors for which the form has been enumerated exhaustively: this test runs the
enumeration (tests/tools/verify_exact_division.cpp) over every divisor on the verified lists.  CPU only (needs an
x86-64 host with FMA, which both the build container and the GPU box have)."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _has_fma() -> bool:
    try:
        return " fma " in Path("/proc/cpuinfo").read_text()
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="host CPU lacks FMA")
def test_every_listed_divisor_is_exact(tmp_path):
    exe = tmp_path / "verify_exact_division"
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-I", os.fspath(ROOT / "libavif_amd" / "csrc"),
                    os.fspath(ROOT / "tests" / "tools" / "verify_exact_division.cpp"), "-o", os.fspath(exe)], check=True)
    proc = subprocess.run([os.fspath(exe)], capture_output=True, text=True)
    lines = proc.stdout.strip().splitlines()
    assert proc.returncode == 0, proc.stdout[-2000:]
    assert len(lines) == 15 + 12 + 28
    assert all(line.endswith("mismatches=0") for line in lines), proc.stdout


@pytest.mark.skipif(not _has_fma(), reason="host CPU lacks FMA")
def test_fp32_shortcuts_of_the_tiles_are_exact(tmp_path):
    """tile_impl.h / pixel_math.h / exactdiv.h: quantisation by one fma (every binary32 operand, four channel maxima); the un-premultiply's
    shared-reciprocal division in fp32 (every alpha code of 8/10/12-bit planes, one binade of colours -- the sequence is scale-invariant)
    and on integers (every code pair of every depth); ARGBUnattenuate's reciprocal from an estimate."""
    exe = tmp_path / "verify_fp32_shortcuts"
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-I", os.fspath(ROOT / "libavif_amd" / "csrc"),
                    os.fspath(ROOT / "tests" / "tools" / "verify_fp32_shortcuts.cpp"), "-o", os.fspath(exe), "-lpthread"], check=True)
    proc = subprocess.run([os.fspath(exe)], capture_output=True, text=True)
    lines = proc.stdout.strip().splitlines()
    assert proc.returncode == 0, proc.stdout[-2000:]
    assert len(lines) == 4 + 2 * 3 + 2 * 4 + 5
    assert all(line.endswith("mismatches=0") for line in lines), proc.stdout
