"""Seam B on the GPU: libavif built FROM THE REFERENCE'S OWN SOURCES with src/reformat_libyuv.c replaced by
integration/reformat_libyuv_hip.c (oracle/_ref/libavif_hipbackend.so, oracle/Makefile), i.e. the reference's
avifImageYUVToRGB / avifImageRGBToYUV / premultiply entry points running unchanged on top of the HIP kernels.

Two arithmetic families, two expectations:
  * AVIFHIP_ARITHMETIC=float: the same entry points of the reference built without any backend
    (oracle/_ref/libavif_ref.so).  The only place where a backend changes what libavif computes is the one libyuv
    changes too: when the hook converted the colours, a pending alpha (un)multiply runs as the integer post-pass
    (src/reformat.c:1574-1585) instead of inside the built-in slow loop (:894-947).  Those cases are compared with the
    same two steps done by the backend-less reference.
  * default (auto): byte-identical to a stock libavif built WITH libyuv, i.e. to the integer-path oracle
    (oracle/libyuv_oracle.c, pinned against the libyuv-enabled binary), for every configuration.
"""
import ctypes as C
import os
from dataclasses import replace

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi, native

pytestmark = pytest.mark.gpu

BACKEND_SO = oracle_lib.ORACLE_DIR / "_ref" / "libavif_hipbackend.so"
SIZES = [(300, 21), (512, 16), (37, 21), (1027, 18)]


@pytest.fixture(scope="module")
def libs(hip):
    if not BACKEND_SO.exists() or oracle_lib.ref() is None:
        # these are GPU tests: the prebuilt reference libraries travel to the GPU box with the snapshot (oracle/_ref is not in
        # .gpurunignore); a run without them must not report green
        pytest.fail("oracle/_ref/libavif_hipbackend.so or libavif_ref.so is missing: build them where /root/reference exists (make -C oracle ref) and ship them")
    os.environ["AVIFHIP_MIN_PIXELS"] = "0"  # tiny test images must take the GPU route too
    be = oracle_lib._bind_libavif(C.CDLL(os.fspath(BACKEND_SO), mode=os.RTLD_LOCAL))
    assert be.avifLibYUVVersion() == 9500
    return H.libavif_backend(be, "reference+hip-hooks"), H.libavif_backend(oracle_lib.ref(), "reference")


def _mul_mode(c: H.Y2RCase) -> int:
    """src/reformat.c:1662-1677"""
    if not c.alpha:
        return 0
    has_alpha = abi.rgb_format_has_alpha(c.rgb_format)
    if not has_alpha or c.ignore_alpha:
        return 0 if c.image_premultiplied else 1
    if not c.image_premultiplied and c.rgb_premultiplied:
        return 1
    if c.image_premultiplied and not c.rgb_premultiplied:
        return 2
    return 0


def test_yuv_to_rgb_through_libavif_hooks(libs, hip):
    be, ref = libs
    cases = [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SIZES, n_random=400, seed=41)]
    bad, hooked = [], 0
    for c in cases:
        mul = _mul_mode(c)
        has_alpha = abi.rgb_format_has_alpha(c.rgb_format)
        two_step = mul != 0 and has_alpha
        if two_step and c.is_float:
            continue  # would need the reference's internal avifRGBImageToF16 as a third step
        before = hip.avifhipLaunchCount()
        rh, ph = H.run_y2r(be, c)
        hooked += hip.avifhipLaunchCount() > before
        if two_step:
            # colour + alpha without multiply, then the integer pass
            rr, pr = H.run_y2r(ref, replace(c, rgb_premultiplied=c.image_premultiplied, ignore_alpha=False))
            if rr == 0:
                out = H.make_y2r_output(c)
                out.pixels[...] = pr
                rr = (ref.premultiply if mul == 1 else ref.unpremultiply)(out.struct)
                pr = out.pixels
        else:
            rr, pr = H.run_y2r(ref, c)
        if rh != rr or not np.array_equal(ph, pr):
            bad.append(f"{c.ident()}: results {rh}/{rr}" + ("" if rh != rr else " " + H.describe_diff(pr, ph)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    assert hooked > len(cases) // 2, f"only {hooked} of {len(cases)} conversions reached the GPU through the hooks"


def test_rgb_to_yuv_through_libavif_hooks(libs, hip):
    be, ref = libs
    cases = [replace(c, avoid_libyuv=False) for c in H.r2y_sweep(SIZES[:3], n_random=300, seed=43)]
    bad, hooked = [], 0
    for c in cases:
        before = hip.avifhipLaunchCount()
        rh, ih = H.run_r2y(be, c)
        hooked += hip.avifhipLaunchCount() > before
        rr, ir = H.run_r2y(ref, c)
        d = None if rh != rr else H.planes_equal(ir, ih)
        if rh != rr or d:
            bad.append(f"{c.ident()}: results {rh}/{rr} {d or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    assert hooked > len(cases) // 4, f"only {hooked} of {len(cases)} conversions reached the GPU through the hooks"


@pytest.mark.parametrize("depth", [8, 16])
@pytest.mark.parametrize("fmt", [1, 2, 4, 5])
def test_premultiply_through_libavif_hooks(libs, hip, fmt, depth):
    from libavif_amd import synth

    be, ref = libs
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(261, 19, depth, fmt, row_pad=6, fill=0x5A)
        synth.fill_rgb(a, 0xBEEF + fmt + depth)
        b = abi.make_rgb(261, 19, depth, fmt, row_pad=6)
        b.pixels[...] = a.pixels
        before = hip.avifhipLaunchCount()
        assert getattr(ref, which)(a.struct) == getattr(be, which)(b.struct) == 0
        assert hip.avifhipLaunchCount() > before
        assert np.array_equal(a.pixels, b.pixels), (which, H.describe_diff(a.pixels, b.pixels))


def test_small_images_stay_on_the_cpu(libs, hip):
    """Below the size threshold the hooks decline (AVIF_RESULT_NOT_IMPLEMENTED) and libavif's own code runs."""
    be, ref = libs
    os.environ["AVIFHIP_MIN_PIXELS"] = "0"  # cached at first use inside the shim: this test only documents the knob
    c = H.Y2RCase(16, 16, avoid_libyuv=True)  # avoidLibYUV: the hook is not even asked (src/reformat.c:1453)
    before = hip.avifhipLaunchCount()
    rh, ph = H.run_y2r(be, c)
    rr, pr = H.run_y2r(ref, c)
    assert rh == rr == 0 and np.array_equal(ph, pr)
    assert hip.avifhipLaunchCount() == before


# ---- default arithmetic: the hip-backed libavif equals a libyuv-backed libavif -----------------------------------


def test_default_arithmetic_equals_libyuv_build_yuv_to_rgb(libs, hip_auto_arithmetic):
    be, _ = libs
    o = H.oracle_libyuv_backend()
    cases = H.libyuv_y2r_cases(SIZES, n_random=500, seed=47)[:900] + [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SIZES, n_random=400, seed=53)]
    bad, hooked = [], 0
    for c in cases:
        before = hip_auto_arithmetic.avifhipLaunchCount()
        rh, ph = H.run_y2r(be, c)
        hooked += hip_auto_arithmetic.avifhipLaunchCount() > before
        ro, po = H.run_y2r(o, c)
        if rh != ro or not np.array_equal(ph, po):
            bad.append(f"{c.ident()}: results {rh}/{ro}" + ("" if rh != ro else " " + H.describe_diff(po, ph)))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    assert hooked > len(cases) // 2


def test_default_arithmetic_equals_libyuv_build_rgb_to_yuv(libs, hip_auto_arithmetic):
    be, _ = libs
    o = H.oracle_libyuv_backend()
    cases = H.libyuv_r2y_cases(SIZES, n_random=300, seed=59) + [replace(c, avoid_libyuv=False) for c in H.r2y_sweep(SIZES[:3], n_random=200, seed=67)]
    bad = []
    for c in cases:
        rh, ih = H.run_r2y(be, c)
        ro, io = H.run_r2y(o, c)
        d = None if rh != ro else H.planes_equal(io, ih)
        if rh != ro or d:
            bad.append(f"{c.ident()}: results {rh}/{ro} {d or ''}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.parametrize("fmt", [1, 2, 4, 5])
def test_default_arithmetic_equals_libyuv_build_premultiply(libs, hip_auto_arithmetic, fmt):
    from libavif_amd import synth

    be, _ = libs
    o = H.oracle_libyuv_backend()
    for which in ("premultiply", "unpremultiply"):
        a = abi.make_rgb(261, 19, 8, fmt, row_pad=6, fill=0x5A)
        synth.fill_rgb(a, 0xBEEF + fmt)
        b = abi.make_rgb(261, 19, 8, fmt, row_pad=6)
        b.pixels[...] = a.pixels
        assert getattr(o, which)(a.struct) == getattr(be, which)(b.struct) == 0
        assert np.array_equal(a.pixels, b.pixels), (which, H.describe_diff(a.pixels, b.pixels))


# ---- threading: libavif calls the hooks from up to 8 threads on disjoint row bands --------------------------------


@pytest.mark.parametrize("arith", ["float", "auto"])
def test_hooks_from_libavif_worker_threads(libs, hip, arith):
    """rgb->maxThreads = 8 makes avifImageYUVToRGB split the image into row bands converted by 8 pthreads
    (src/reformat.c:1679-1745), each of which calls the backend hook concurrently: results must equal the single-threaded
    ones (the reference's own invariant, tests/gtest/avifrgbtoyuvthreadingtest.cc) in both arithmetics."""
    be, _ = libs
    hip.avifhipSetArithmetic(1 if arith == "float" else 0)
    try:
        cases = [
            H.Y2RCase(640, 360, yuv_format=1, matrix=1, yuv_range=0, avoid_libyuv=False),
            H.Y2RCase(640, 360, yuv_format=3, matrix=1, yuv_range=0, upsampling=3, avoid_libyuv=False, alpha=True),
            H.Y2RCase(1027, 201, yuv_depth=10, yuv_format=2, matrix=9, yuv_range=1, upsampling=4, rgb_depth=16, avoid_libyuv=False),
            H.Y2RCase(800, 600, yuv_depth=10, yuv_format=1, matrix=9, yuv_range=1, rgb_depth=8, alpha=True, rgb_premultiplied=True, avoid_libyuv=False),
        ]
        for c in cases:
            img = H.make_y2r_inputs(c)
            outs = []
            for threads in (1, 8, 8, 3):
                rgb = H.make_y2r_output(c)
                rgb.struct.maxThreads = threads
                before = hip.avifhipLaunchCount()
                assert be.yuv_to_rgb(img.struct, rgb.struct) == 0, c.ident()
                outs.append(rgb.pixels.copy())
            for o in outs[1:]:
                assert np.array_equal(outs[0], o), (c.ident(), H.describe_diff(outs[0], o))
            want_backend = H.oracle_backend() if arith == "float" else H.oracle_libyuv_backend()
            if arith == "auto" or not (c.alpha and c.rgb_premultiplied):
                ro, po = H.run_y2r(want_backend, c)
                assert ro == 0 and np.array_equal(po, outs[0]), (c.ident(), H.describe_diff(po, outs[0]))
    finally:
        hip.avifhipSetArithmetic(1)


# ---- the colour hook folds libavif's next steps (src/reformat.c:1574-1590) into its own pass ---------------------


def _fold_case():
    # cfg3's shape in small: 10-bit 4:4:4 + alpha -> premultiplied RGBA16 (libyuv declines 16-bit pixels: fp32 colour, integer post-pass)
    return H.Y2RCase(640, 66, yuv_depth=10, yuv_format=1, yuv_range=1, matrix=9, alpha=True, rgb_depth=16, rgb_premultiplied=True, avoid_libyuv=False)


def test_pending_premultiply_costs_no_second_pass(libs, hip_auto_arithmetic):
    """avifImageYUVToRGB with a pending premultiply used to be two hook calls, each moving the image across the bus both ways; the colour hook now
    returns the final pixels and answers libavif's follow-up call itself -- and only that one."""
    be, _ = libs
    lib, o = hip_auto_arithmetic, H.oracle_libyuv_backend()
    for c in (_fold_case(), replace(_fold_case(), is_float=True), replace(_fold_case(), rgb_depth=8, yuv_depth=8, image_premultiplied=True, rgb_premultiplied=False, yuv_format=1)):
        want_r, want = H.run_y2r(o, c)
        before = lib.avifhipLaunchCount()
        got_r, got = H.run_y2r(be, c)
        launches = lib.avifhipLaunchCount() - before
        assert got_r == want_r == 0 and np.array_equal(got, want), (c.ident(), H.describe_diff(want, got))
        # one conversion kernel (plus at most the two leftover launches of the tiled route): no separate (un)premultiply / half-float pass
        assert 1 <= launches <= 3, (c.ident(), launches)
    # ... and a later, separate premultiply of the application on the same buffer is a real one (no stale note)
    c = _fold_case()
    out = H.make_y2r_output(c)
    img = H.make_y2r_inputs(c)
    assert be.yuv_to_rgb(img.struct, out.struct) == 0
    out.pixels.view(np.uint16)[:, 3::4] //= 2  # the application edits alpha ...
    twin = H.make_y2r_output(c)
    twin.pixels[...] = out.pixels
    before = lib.avifhipLaunchCount()
    assert be.premultiply(out.struct) == 0 and o.premultiply(twin.struct) == 0  # ... and premultiplies again
    assert lib.avifhipLaunchCount() > before
    assert np.array_equal(out.pixels, twin.pixels), H.describe_diff(twin.pixels, out.pixels)


def test_changed_pixels_between_hook_calls_fall_back_to_the_staged_path(libs, hip_auto_arithmetic):
    """The hooks called directly (internal symbols), pixels changed in between: the follow-up hook must do real work on the changed pixels."""
    lib = hip_auto_arithmetic
    raw = C.CDLL(os.fspath(BACKEND_SO), mode=os.RTLD_LOCAL)
    y2r = raw.avifImageYUVToRGBLibYUV
    y2r.restype, y2r.argtypes = C.c_int, [C.POINTER(abi.avifImage), C.POINTER(abi.avifRGBImage), C.c_int, C.POINTER(C.c_int)]
    pre = raw.avifRGBImagePremultiplyAlphaLibYUV
    pre.restype, pre.argtypes = C.c_int, [C.POINTER(abi.avifRGBImage)]
    c = _fold_case()
    img = H.make_y2r_inputs(c)
    o = H.oracle_libyuv_backend()
    for mutate in (False, True):
        out = H.make_y2r_output(c)
        flag = C.c_int(0)
        assert y2r(img.struct, out.struct, 1, C.byref(flag)) == 0 and flag.value == 1
        folded = out.pixels.copy()  # final pixels: already premultiplied
        if mutate:
            out.pixels[...] ^= 0x01
        before = lib.avifhipLaunchCount()
        assert pre(out.struct) == 0
        if not mutate:
            assert lib.avifhipLaunchCount() == before and np.array_equal(out.pixels, folded)  # answered from the note
        else:
            assert lib.avifhipLaunchCount() > before  # a real premultiply of what is there now
            twin = H.make_y2r_output(c)
            twin.pixels[...] = folded ^ 0x01
            assert o.premultiply(twin.struct) == 0
            assert np.array_equal(out.pixels, twin.pixels), H.describe_diff(twin.pixels, out.pixels)


def test_fold_can_be_switched_off_and_a_late_follow_up_is_still_the_follow_up(libs, hip_auto_arithmetic, monkeypatch):
    """The fold rests on libavif's own call sequence: it is compiled in only for the libavif it was validated against (1.4.x) and
    AVIFHIP_FOLD=0 switches it off -- the colour hook then does only its own job and libavif's follow-up premultiply is a real pass with
    the same bytes at the end.  No clock takes part: a follow-up that arrives two seconds after the colour hook (a stopped process, a
    paused VM) is answered from the note like a prompt one -- a real premultiply would run over pixels that are final already
    (src/reformat.c:1574-1590: libavif issues the follow-up unconditionally, on the same thread)."""
    import time

    be, _ = libs
    lib, o = hip_auto_arithmetic, H.oracle_libyuv_backend()
    c = _fold_case()
    want_r, want = H.run_y2r(o, c)
    before = lib.avifhipLaunchCount()
    got_r, got = H.run_y2r(be, c)
    folded_launches = lib.avifhipLaunchCount() - before
    monkeypatch.setenv("AVIFHIP_FOLD", "0")
    before = lib.avifhipLaunchCount()
    off_r, off = H.run_y2r(be, c)
    unfolded_launches = lib.avifhipLaunchCount() - before
    monkeypatch.delenv("AVIFHIP_FOLD")
    assert got_r == off_r == want_r == 0 and np.array_equal(got, want) and np.array_equal(off, want), c.ident()
    assert unfolded_launches > folded_launches, (folded_launches, unfolded_launches)
    # the hooks called directly: the colour hook, a stall of two seconds, then libavif's follow-up premultiply of the same buffer
    raw = C.CDLL(os.fspath(BACKEND_SO), mode=os.RTLD_LOCAL)
    y2r = raw.avifImageYUVToRGBLibYUV
    y2r.restype, y2r.argtypes = C.c_int, [C.POINTER(abi.avifImage), C.POINTER(abi.avifRGBImage), C.c_int, C.POINTER(C.c_int)]
    pre = raw.avifRGBImagePremultiplyAlphaLibYUV
    pre.restype, pre.argtypes = C.c_int, [C.POINTER(abi.avifRGBImage)]
    monkeypatch.setenv("AVIFHIP_FOLD_EXPIRY_MS", "5")  # (the knob of rounds 4-5: must be dead)
    img, out, flag = H.make_y2r_inputs(c), H.make_y2r_output(c), C.c_int(0)
    assert y2r(img.struct, out.struct, 1, C.byref(flag)) == 0
    time.sleep(2.0)
    before = lib.avifhipLaunchCount()
    assert pre(out.struct) == 0
    assert lib.avifhipLaunchCount() == before, "the late follow-up ran a second pass over final pixels"
    assert np.array_equal(out.pixels, want), "a stalled follow-up: " + H.describe_diff(want, out.pixels)
    # ... and the note is gone: the application's own premultiply of that buffer afterwards is a real pass
    before = lib.avifhipLaunchCount()
    assert pre(out.struct) == 0
    assert lib.avifhipLaunchCount() > before
