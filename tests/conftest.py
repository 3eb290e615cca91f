import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library, with a GPU present (skips loudly otherwise; never falls back to CPU)."""
    from libavif_amd import native

    lib = native.load()
    if lib.avifhipDeviceCount() <= 0:
        pytest.skip("no HIP device visible")
    # Tests using this fixture compare with the fp32 oracle (a libavif built without libyuv): pin that arithmetic.
    # The integer path (the library's default, AVIFHIP_ARITHMETIC_AUTO) is tested through hip_auto_arithmetic.
    lib.avifhipSetArithmetic(1)
    return lib


@pytest.fixture()
def hip_auto_arithmetic(hip):
    """The library's default arithmetic: what a libavif built with libyuv computes (integer path where libyuv serves)."""
    hip.avifhipSetArithmetic(0)
    yield hip
    hip.avifhipSetArithmetic(1)
