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
    return lib
