"""The AVIF_HIP_REFORMAT option for libavif's own build system: integration/libavif-AVIF_HIP_REFORMAT.patch applied to a copy of
the reference tree, configured with CMake + Ninja, built, and the resulting libavif checked for what seam B promises -- the six
*LibYUV hooks come from integration/reformat_libyuv_hip.c (avifLibYUVVersion() == 9500), src/reformat_libyuv.c is not compiled,
libavifhip.so is linked.  Needs /root/reference (build container only)."""
import ctypes
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")
PATCH = ROOT / "integration" / "libavif-AVIF_HIP_REFORMAT.patch"

pytestmark = pytest.mark.skipif(not (REFERENCE / "CMakeLists.txt").exists() or shutil.which("cmake") is None or shutil.which("ninja") is None,
                                reason="needs the reference tree, cmake and ninja")


def _run(cmd, cwd):
    proc = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, f"{' '.join(map(str, cmd))}\n{proc.stdout[-3000:]}\n{proc.stderr[-3000:]}"
    return proc.stdout


@pytest.fixture(scope="module")
def patched_tree(tmp_path_factory):
    tree = tmp_path_factory.mktemp("libavif")
    shutil.copytree(REFERENCE, tree, dirs_exist_ok=True, ignore=shutil.ignore_patterns(".git", "data", "android_jni", "ext"))
    _run(["patch", "-p1", "-i", os.fspath(PATCH)], tree)
    return tree


def test_option_is_off_by_default_and_checks_its_arguments(patched_tree):
    text = (patched_tree / "CMakeLists.txt").read_text()
    assert 'option(AVIF_HIP_REFORMAT "Serve YUV<->RGB' in text and "OFF)" in text.split("option(AVIF_HIP_REFORMAT", 1)[1].splitlines()[0]
    # without AVIFHIP_DIR the configuration must stop with a message, not build a libavif without a backend
    proc = subprocess.run(["cmake", "-S", ".", "-B", "build_bad", "-G", "Ninja", "-DAVIF_HIP_REFORMAT=ON", "-DAVIF_LIBYUV=OFF"], cwd=patched_tree, capture_output=True,
                          text=True, timeout=600)
    assert proc.returncode != 0 and "AVIFHIP_DIR" in proc.stderr


def test_configure_build_and_inspect(patched_tree):
    assert (ROOT / "libavif_amd" / "csrc" / "libavifhip.so").exists(), "build libavifhip.so first"
    _run(["cmake", "-S", ".", "-B", "build", "-G", "Ninja", "-DCMAKE_BUILD_TYPE=Release", "-DAVIF_HIP_REFORMAT=ON", f"-DAVIFHIP_DIR={ROOT}", "-DAVIF_LIBYUV=OFF",
          "-DBUILD_SHARED_LIBS=ON"], patched_tree)
    ninja = (patched_tree / "build" / "build.ninja").read_text()
    assert "integration/reformat_libyuv_hip.c" in ninja and "src/reformat_libyuv.c" not in ninja
    assert "third_party/libyuv/source/scale.c" in ninja  # the vendored scaler stays (src/scale.c needs it)
    _run(["ninja", "-C", "build", "avif"], patched_tree)
    so = next((patched_tree / "build").glob("libavif.so*"))
    needed = _run(["readelf", "-d", os.fspath(so)], patched_tree)
    assert "libavifhip.so" in needed
    env_path = os.environ.get("LD_LIBRARY_PATH", "")
    os.environ["LD_LIBRARY_PATH"] = os.fspath(ROOT / "libavif_amd" / "csrc") + (":" + env_path if env_path else "")
    try:
        ctypes.CDLL(os.fspath(ROOT / "libavif_amd" / "csrc" / "libavifhip.so"), mode=ctypes.RTLD_GLOBAL)
        lib = ctypes.CDLL(os.fspath(so), mode=os.RTLD_LOCAL)
    finally:
        os.environ["LD_LIBRARY_PATH"] = env_path
    lib.avifLibYUVVersion.restype = ctypes.c_uint
    assert lib.avifLibYUVVersion() == 9500  # integration/reformat_libyuv_hip.c answers for the backend
    lib.avifVersion.restype = ctypes.c_char_p
    assert lib.avifVersion().decode().startswith("1.4")
