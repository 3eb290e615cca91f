"""The in-process device farm's planner and control surface (include/avifhip.h: avifhipSetDeviceSet, avifhipPlanFarmRows) -- host logic only,
no GPU: which row shares a device set gives an image, and that the set itself round-trips.  What the shares compute is checked on the GPU
(tests/test_gpu_device_farm.py) and by the C program tests/c/farm_check.c."""
import ctypes as C

import pytest

from libavif_amd import abi, native


@pytest.fixture(scope="module")
def lib():
    return native.load()


def plan(lib, w, h, workers):
    bands = (abi.avifCropRect * max(workers, 1))()
    n = C.c_uint32(0)
    assert lib.avifhipPlanFarmRows(w, h, workers, bands, max(workers, 1), C.byref(n)) == 0
    return [(b.x, b.y, b.width, b.height) for b in bands[: n.value]]


@pytest.mark.parametrize("w,h", [(7680, 4320), (15360, 8640), (3840, 2160), (4099, 3001), (1920, 1080), (64, 100000), (100000, 64)])
@pytest.mark.parametrize("workers", [1, 2, 3, 4, 8])
def test_shares_are_contiguous_aligned_and_cover_the_image(lib, w, h, workers):
    lib.avifhipSetFarmMinSharePixels(0)
    shares = plan(lib, w, h, workers)
    assert 1 <= len(shares) <= workers
    y = 0
    for k, (x, y0, sw, sh) in enumerate(shares):
        assert (x, sw) == (0, w) and y0 == y and sh > 0
        if k + 1 < len(shares):
            assert sh % 32 == 0  # whole tiles of the tiled kernels, even rows for subsampled chroma
        y += sh
    assert y == h
    if len(shares) > 1:
        # no share below ~2 megapixels except the last one (what rounding to 32 rows leaves)
        assert all(sw * sh >= (1 << 21) * 0.9 for (_, _, sw, sh) in shares[:-1])
        assert len({sh for (_, _, _, sh) in shares[:-1]}) == 1  # equal shares, the remainder last


def test_small_images_are_not_farmed(lib):
    lib.avifhipSetFarmMinSharePixels(0)
    assert plan(lib, 1920, 1080, 8) == [(0, 0, 1920, 1080)]          # 2 MP: one share
    assert len(plan(lib, 3840, 2160, 8)) == 3                        # 8.3 MP: three shares of 2.7 MP, not eight of 1
    assert [s[3] for s in plan(lib, 7680, 4320, 8)] == [544] * 7 + [512]  # the headline's frame on an 8-GPU node
    assert [s[3] for s in plan(lib, 15360, 8640, 8)] == [1088] * 7 + [1024]  # cfg5's canvas: a tile row of 1080 is not a multiple of 32


def test_min_share_knob(lib):
    lib.avifhipSetFarmMinSharePixels(64 * 32)
    try:
        assert [s[3] for s in plan(lib, 64, 100, 3)] == [64, 36]
        assert [s[3] for s in plan(lib, 300, 131, 3)] == [64, 64, 3]
    finally:
        lib.avifhipSetFarmMinSharePixels(0)
    assert plan(lib, 300, 131, 3) == [(0, 0, 300, 131)]


def test_device_set_round_trip_without_a_gpu(lib):
    have = (C.c_int * 8)()
    before = [have[k] for k in range(lib.avifhipGetDeviceSet(have, 8))]
    try:
        assert lib.avifhipSetDeviceSet((C.c_int * 3)(0, 0, 0), 3) == 0
        assert lib.avifhipGetDeviceSet(have, 8) == 3 and list(have[:3]) == [0, 0, 0]
        assert lib.avifhipGetDeviceSet(None, 0) == 3  # size only
        assert lib.avifhipSetDeviceSet((C.c_int * 1)(-1), 1) != 0  # negative index
        assert lib.avifhipSetDeviceSet(None, 2) != 0               # count without a list
        assert lib.avifhipSetDeviceSet(None, 0) == 0 and lib.avifhipGetDeviceSet(have, 8) == 0
        assert lib.avifhipLastFarmWorkers() == 0
        assert lib.avifhipLastFarmTransferBytes(0, None, None, None, None, None) != 0
    finally:
        lib.avifhipSetDeviceSet((C.c_int * max(len(before), 1))(*before), len(before))
