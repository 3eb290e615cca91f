import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_addoption(parser):
    parser.addoption("--seed-rotation", action="store", default=None, metavar="N",
                     help="move the random part of every parity sweep to another seed stream (tests/harness.py: sweep_rng); 0 = the committed set")
    parser.addoption("--order-seed", action="store", default=None, metavar="N",
                     help="run the test FILES in a shuffled order (seeded; tests keep their order inside a file): the library keeps pooled contexts, "
                          "worker threads and cached pinned buffers between calls, and the reference's functions are stateless -- no result may "
                          "depend on what ran before (VERDICT round 5: the open fault was found by running two files in the other order)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--seed-rotation") is not None:
        os.environ["AVIFHIP_TEST_SEED_ROTATION"] = str(int(config.getoption("--seed-rotation")))
    # The GPU tier runs with poisoned device scratch (libavif_amd/csrc/api.cpp: reserve): whatever a kernel reads from scratch that nobody
    # wrote is 0xA5 bytes in every process -- not zeros in a young process and another context's pixels after the right history.  (Round 5's
    # open fault was such a read: the window scaling kernel past its column tables; it showed only after the device farm had recycled
    # its buffers, in two processes of three.)  AVIFHIP_POISON_SCRATCH=0 in the environment switches it off.
    os.environ.setdefault("AVIFHIP_POISON_SCRATCH", "1")


def pytest_collection_modifyitems(config, items):
    seed = config.getoption("--order-seed")
    if seed is None:
        return
    import random

    files = []
    for item in items:
        if item.fspath not in files:
            files.append(item.fspath)
    random.Random(int(seed)).shuffle(files)
    rank = {f: k for k, f in enumerate(files)}
    items.sort(key=lambda item: rank[item.fspath])  # (stable: the order inside a file stays)
    print(f"\n--order-seed {seed}: " + " ".join(Path(str(f)).name for f in files))


def _gpu_run(config) -> bool:
    expr = config.getoption("-m") or ""
    return "gpu" in expr and "not gpu" not in expr


def pytest_sessionstart(session):
    """`-m gpu` is the parity tier on the GPU box.  Its checkers are the prebuilt reference libraries under oracle/_ref (built here,
    where /root/reference exists, shipped with the snapshot) and a GPU: if either is missing, dozens of tests would skip and the
    run would still look green -- abort loudly instead."""
    if not _gpu_run(session.config):
        return
    needed = ["oracle/liboracle.so", "oracle/_ref/libavif_ref.so", "oracle/_ref/libavif_hipbackend.so", "oracle/_ref/libavifutil_ref.so", "oracle/_ref/preload_probe",
              "libavif_amd/csrc/libavifhip.so", "libavif_amd/csrc/libavifhip_preload.so"]
    missing = [n for n in needed if not (ROOT / n).exists()]
    if missing:
        raise pytest.UsageError("GPU parity run without its checkers / native libraries: " + ", ".join(missing) +
                                " -- run `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference exists")
    from libavif_amd import native

    if native.load().avifhipDeviceCount() <= 0:
        raise pytest.UsageError("`-m gpu` selected but no HIP device is visible: the HIP path is the only path, nothing would be tested")


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
