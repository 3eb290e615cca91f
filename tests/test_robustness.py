"""Round-2 plumbing around the kernels: the LD_PRELOAD interposer's version gate (CPU), per-thread device tables shared across streams
(ScratchScope), contexts leased from the pool by short-lived threads, banded host-resident calls from many threads at once (GPU)."""
import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, native

ROOT = Path(__file__).resolve().parent.parent
PRELOAD = ROOT / "libavif_amd" / "csrc" / "libavifhip_preload.so"

STUB = r"""
#include <stddef.h>
const char * avifVersion(void) { return STUB_VERSION; }
unsigned int avifLibYUVVersion(void) { return 0; }
int avifImageYUVToRGB(const void * image, void * rgb) { (void)image; (void)rgb; return 1234; }
int avifImageRGBToYUV(void * image, const void * rgb) { (void)image; (void)rgb; return 1235; }
"""
APP = r"""
#include <stdio.h>
#include <string.h>
int avifImageYUVToRGB(const void * image, void * rgb);
int main(void)
{
    /* structs large enough for the mirror (avifImage 224 bytes, avifRGBImage 64), 4096 x 4096 so that the size threshold passes */
    unsigned char image[512], rgb[512];
    memset(image, 0, sizeof(image)); memset(rgb, 0, sizeof(rgb));
    ((unsigned *)image)[0] = 4096; ((unsigned *)image)[1] = 4096; ((unsigned *)image)[2] = 8;
    ((unsigned *)rgb)[0] = 4096; ((unsigned *)rgb)[1] = 4096; ((unsigned *)rgb)[2] = 8;
    printf("result %d\n", avifImageYUVToRGB(image, rgb));
    return 0;
}
"""


@pytest.mark.skipif(not PRELOAD.exists(), reason="libavifhip_preload.so not built")
@pytest.mark.parametrize("version,expect_stub", [("0.11.1", True), ("1.3.0", True)])
def test_interposer_passes_through_other_libavif_versions(tmp_path, version, expect_stub):
    """Interposing a libavif whose avifVersion() is not the mirrored 1.4.x: every call goes straight to the real library (the structs
    would be read through the wrong layout otherwise) -- whatever the arguments are."""
    (tmp_path / "stub.c").write_text(STUB)
    (tmp_path / "app.c").write_text(APP)
    subprocess.run(["gcc", "-shared", "-fPIC", f'-DSTUB_VERSION="{version}"', "-o", "libavif.so", "stub.c"], cwd=tmp_path, check=True)
    subprocess.run(["gcc", "-o", "app", "app.c", "-L.", "-lavif", "-Wl,-rpath,$ORIGIN"], cwd=tmp_path, check=True)
    env = dict(os.environ, LD_PRELOAD=os.fspath(PRELOAD), AVIFHIP_MIN_PIXELS="0")
    env.pop("AVIFHIP_PRELOAD_FORCE", None)
    out = subprocess.run([os.fspath(tmp_path / "app")], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert ("result 1234" in out.stdout) == expect_stub, out.stdout


@pytest.mark.gpu
def test_tables_shared_across_streams_are_ordered(hip_auto_arithmetic):
    """Batch launches from ONE thread on TWO streams, alternating, with different descriptor tables each time: the second upload must not
    overwrite the table a kernel of the other stream is still reading (api_batch.cpp per-thread tables)."""
    from libavif_amd import device, farm

    lib = hip_auto_arithmetic
    s1, s2 = lib.avifhipStreamCreate(), lib.avifhipStreamCreate()
    cases = [H.Y2RCase(2048, 512, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, avoid_libyuv=False, seed=11 + k) for k in range(2)]
    wants, dimgs, drgbs, outs, crops = [], [], [], [], []
    for c in cases:
        res, px = H.run_y2r(H.oracle_libyuv_backend(), c)
        assert res == 0
        wants.append(px)
        img, out = H.make_y2r_inputs(c), H.make_y2r_output(c)
        dimgs.append(device.DeviceYUV(img)), drgbs.append(device.DeviceRGB(out, upload=True)), outs.append(out)
    # different rectangle lists per stream: 16 rectangles vs 4
    lists = [farm.grid_rects(2048, 512, 512, 128), farm.grid_rects(2048, 512, 1024, 256)]
    for rep in range(40):
        for k, stream in ((0, s1), (1, s2)):
            rects = lists[(k + rep) % 2]
            n = len(rects)
            imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(dimgs[k].struct)] * n)
            rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(drgbs[k].struct)] * n)
            cr = (abi.avifCropRect * n)(*[abi.avifCropRect(*r) for r in rects])
            native.check(lib.avifhipImageYUVToRGBBatchAsync(n, imgs, rgbs, cr, stream), "batch")
    native.check(lib.avifhipSynchronize(s1), "sync")
    native.check(lib.avifhipSynchronize(s2), "sync")
    for k in range(2):
        drgbs[k].download_into_host()
        wb = cases[k].w * 4
        assert np.array_equal(outs[k].pixels[:, :wb], wants[k][:, :wb]), H.describe_diff(wants[k][:, :wb], outs[k].pixels[:, :wb])
    lib.avifhipStreamDestroy(s1), lib.avifhipStreamDestroy(s2)


@pytest.mark.gpu
def test_short_lived_threads_share_pooled_contexts(hip_auto_arithmetic):
    """libavif creates its worker threads anew for every call: 6 rounds of 8 threads, each converting a host-resident image (banded,
    with the download helper) and checking it; every thread's first call leases a context another thread handed back."""
    lib = hip_auto_arithmetic
    c = H.Y2RCase(3840, 1088, yuv_format=3, yuv_range=0, matrix=1, upsampling=4, avoid_libyuv=False)
    res, want = H.run_y2r(H.oracle_libyuv_backend(), c)
    assert res == 0
    img = H.make_y2r_inputs(c)
    errors = []

    def worker(idx):
        try:
            out = H.make_y2r_output(c)
            for _ in range(3):
                r = lib.avifhipImageYUVToRGB(img.struct, out.struct)
                if r != 0:
                    raise RuntimeError(f"result {r}: {lib.avifhipLastError().decode()}")
            if not np.array_equal(out.pixels[:, : c.w * 4], want[:, : c.w * 4]):
                raise RuntimeError("bytes differ: " + H.describe_diff(want, out.pixels))
        except Exception as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    for _ in range(6):
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors[:4]
