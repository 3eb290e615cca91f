"""Gain-map application (avifRGBImageApplyGainMap, reference src/gainmap.c:73-315).

CPU: the oracle's restatement (oracle/gainmap_oracle.c: transfer functions, primaries matrices, float pixel accessors, the
tone-mapping loop) against avifRGBImageApplyGainMap of the reference compiled from its own sources -- same libm, same
flags: every output byte, the result code and the CLLI values are IDENTICAL."""
import ctypes as C

import numpy as np
import pytest

import gainmap_cases as G
import oracle_lib
from libavif_amd import abi

libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def run(fn, c, extra):
    base = G.make_base(c)
    gm, keep = G.make_gain_map(c)
    out = G.make_output(c)
    clli = abi.avifContentLightLevelInformationBox(0xFFFF, 0xFFFF)
    res = fn(base.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, out.struct, C.byref(clli), extra)
    pixels = G.output_bytes(out) if (res == 0 and out.struct.pixels) else None
    if out.struct.pixels:
        libc.free(C.cast(out.struct.pixels, C.c_void_p))
    return res, pixels, (clli.maxCLL, clli.maxPALL)


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_oracle_equals_reference():
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.cases(500, seed=1)
    for c in cases:
        ra, pa, ca = run(ref.avifRGBImageApplyGainMap, c, C.byref(diag))
        rb, pb, cb = run(o.oracleRGBImageApplyGainMap, c, 0)  # the from-source reference build has no libyuv
        if ra != rb or (ra == 0 and (not np.array_equal(pa, pb) or ca != cb)):
            bad.append(f"{c.ident()}: results {ra}/{rb} clli {ca}/{cb}" + ("" if ra or rb or pa is None or np.array_equal(pa, pb) else
                                                                       f" {int((pa != pb).sum())} bytes differ"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_primaries_matrices_equal_reference():
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    for a in G.PRIMARIES + [10]:
        for b in G.PRIMARIES + [10]:
            ma, mb = (C.c_double * 9)(), (C.c_double * 9)()
            ra, rb = ref.avifColorPrimariesComputeRGBToRGBMatrix(a, b, C.byref(ma)), o.oracleColorPrimariesComputeRGBToRGBMatrix(a, b, C.byref(mb))
            assert bool(ra) == bool(rb), (a, b)
            if ra:
                assert list(ma) == list(mb), (a, b)


@pytest.mark.skipif(oracle_lib.pillow() is None, reason="Pillow's bundled libavif (built with libyuv) not present")
def test_oracle_libyuv_build_equals_libyuv_enabled_binary():
    """The default-build flavour of the oracle (the gain map's own YUV -> RGB conversion and rescaling as a libavif built WITH
    libyuv computes them) against the only libyuv-enabled libavif binary available offline (Pillow's, libavif 1.4.1)."""
    pil, o = oracle_lib.pillow(), oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.cases(250, seed=5)
    for c in cases:
        if c.gm_depth > 8 and (c.gm_w or c.w, c.gm_h or c.h) != (c.w, c.h):
            continue  # that binary rescales with libyuv 1922's own ScalePlane_12, not with the scaler vendored in the reference tree
        ra, pa, ca = run(pil.avifRGBImageApplyGainMap, c, C.byref(diag))
        rb, pb, cb = run(o.oracleRGBImageApplyGainMap, c, 1)
        if rb == abi.AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE:
            continue  # the NaN check (src/gainmap.c:277-281) is newer than libavif 1.4.1
        if ra != rb or (ra == 0 and (not np.array_equal(pa, pb) or ca != cb)):
            bad.append(f"{c.ident()}: results {ra}/{rb} clli {ca}/{cb}")
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


def gain_map_state(gm, img_struct):
    """Everything avifRGBImageComputeGainMap writes: the metadata fractions and the gain map image."""
    meta = []
    for name in ("gainMapMin", "gainMapMax", "gainMapGamma", "baseOffset", "alternateOffset"):
        meta += [(f.n, f.d) for f in getattr(gm, name)]
    meta += [(gm.baseHdrHeadroom.n, gm.baseHdrHeadroom.d), (gm.alternateHdrHeadroom.n, gm.alternateHdrHeadroom.d), gm.useBaseColorSpace]
    import test_scale as TS

    planes = TS.planes_of(img_struct)
    return meta, (img_struct.width, img_struct.height), planes


def run_compute(fn, c, extra):
    import test_scale as TS

    base, alt = G.make_compute_inputs(c)
    gm, img = G.make_compute_gain_map(c)
    res = fn(base.struct, c.base_primaries, c.base_tc, alt.struct, c.alt_primaries, c.alt_tc, C.byref(gm), extra)
    state = gain_map_state(gm, img.struct) if res == 0 else None
    TS.free_owned(img.struct)
    return res, state


def states_equal(a, b):
    if a is None or b is None:
        return a is None and b is None
    if a[0] != b[0] or a[1] != b[1]:
        return False
    return all((x is None) == (y is None) and (x is None or np.array_equal(x, y)) for x, y in zip(a[2], b[2]))


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_compute_oracle_equals_reference():
    """avifRGBImageComputeGainMap (src/gainmap.c:535-843): metadata and gain-map planes identical."""
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.compute_cases(250, seed=11)
    for c in cases:
        ra, sa = run_compute(ref.avifRGBImageComputeGainMap, c, C.byref(diag))
        rb, sb = run_compute(o.oracleRGBImageComputeGainMap, c, 0)
        if ra != rb or not states_equal(sa, sb):
            bad.append(f"{c.ident()}: results {ra}/{rb}" + ("" if sa is None or sb is None else f" meta equal {sa[0] == sb[0]} size {sa[1]}/{sb[1]}"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.skipif(oracle_lib.pillow() is None, reason="Pillow's bundled libavif (built with libyuv) not present")
def test_compute_oracle_libyuv_build_equals_libyuv_enabled_binary():
    """avifRGBImageComputeGainMap of the libyuv-enabled binary (the gain map's RGB -> YUV conversion goes through libyuv there)
    against the default-build flavour of the oracle.  Rescaled gain maps are left out: that binary scales with libyuv 1922's own
    scaler, the reference tree (and the oracle) with the vendored one."""
    pil, o = oracle_lib.pillow(), oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = [c for c in G.compute_cases(150, seed=21) if (c.gm_w or c.w, c.gm_h or c.h) == (c.w, c.h)]
    for c in cases:
        ra, sa = run_compute(pil.avifRGBImageComputeGainMap, c, C.byref(diag))
        rb, sb = run_compute(o.oracleRGBImageComputeGainMap, c, 1)
        if ra != rb or not states_equal(sa, sb):
            bad.append(f"{c.ident()}: results {ra}/{rb}")
    assert len(cases) > 80 and not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


def test_argument_errors():
    o = oracle_lib.oracle()
    c = G.GainMapCase(8, 8)
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, headroom=-1.0), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, gm_gamma=((0, 1), (1, 1), (1, 1))), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, gm_min=((2, 1), (0, 1), (0, 1)), gm_max=((1, 1), (1, 1), (1, 1))), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, c, 0)[0] == 0


# ---------------------------------------------------------------------------------------------------
# GPU: the product against the oracle


def _compare_gpu(hip, cases, libyuv_build):
    from libavif_amd import native

    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad, pall_off = [], 0
    for c in cases:
        ra, pa, ca = run(o.oracleRGBImageApplyGainMap, c, int(libyuv_build))
        rb, pb, cb = run(hip.avifhipRGBImageApplyGainMap, c, C.byref(diag))
        ok = ra == rb
        if ok and ra == 0:
            # every byte identical; maxCLL identical; maxPALL within one nit (fp32 running sum vs fp64 partial sums, include/avifhip.h)
            ok = np.array_equal(pa, pb) and ca[0] == cb[0] and abs(ca[1] - cb[1]) <= 1
            pall_off += int(ca[1] != cb[1])
        if not ok:
            bad.append(f"{c.ident()} [{native.last_kernel()}]: results {ra}/{rb} clli {ca}/{cb}" +
                       ("" if ra or rb or pa is None or pb is None or np.array_equal(pa, pb) else
                        f" {int((pa != pb).sum())} bytes differ, max |delta| {int(np.abs(pa.astype(int) - pb.astype(int)).max())}"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    return pall_off


@pytest.mark.gpu
def test_gpu_equals_oracle_default_arithmetic(hip_auto_arithmetic):
    """The library's default: the gain map's own YUV -> RGB conversion as a libavif built with libyuv computes it."""
    off = _compare_gpu(hip_auto_arithmetic, G.cases(260, seed=2), libyuv_build=True)
    assert off <= 6, off  # maxPALL off by one nit: rare


@pytest.mark.gpu
def test_gpu_equals_oracle_fp32_arithmetic(hip):
    _compare_gpu(hip, G.cases(120, seed=3), libyuv_build=False)


def _fast_kernel_cases():
    """What the fast apply kernel serves (4-channel integer pixels up to 12 bits on both sides, a gain map, a curve with a locator): every
    channel order on both sides, every pixel-size combination, each of the four primaries-conversion variants, 8- / 10- / 12-bit gain maps (12 bits: with 8-bit images on both sides, whose tables leave the room), rows
    that end inside a lane's run of four pixels, rows shorter than a tile, one-row images."""
    four = [abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_BGRA, abi.AVIF_RGB_FORMAT_ABGR]
    out = []
    k = 0
    for bf in four:
        for of in four:
            bd, od = ((8, 8), (8, 10), (10, 8), (12, 12), (10, 10), (8, 12))[k % 6]
            w, h = ((37, 21), (64, 9), (261, 5), (4, 3), (5, 1), (7, 2), (515, 17), (258, 8))[k % 8]
            out.append(G.GainMapCase(w, h, base_format=bf, out_format=of, base_depth=bd, out_depth=od, out_tc=9 if od == 12 else (13, 16, 18, 1, 8)[k % 5],
                                     base_primaries=(1, 1, 9, 12)[k % 4], out_primaries=(1, 9, 9, 1)[k % 4], use_base_color_space=bool(k % 3),
                                     alt_primaries=9, gm_depth=(8, 8, 10, 12 if (bd, od) == (8, 8) else 10)[(k // 2) % 4], seed=900 + k))
            k += 1
    out.append(G.GainMapCase(1001, 67, base_depth=8, out_depth=10, out_tc=16, out_primaries=9, seed=77))  # many tiles per workgroup
    out.append(G.GainMapCase(640, 360, base_depth=10, out_depth=10, base_tc=16, out_tc=13, base_primaries=9, out_primaries=1, headroom=1.0,
                             base_headroom=(4, 1), alt_headroom=(0, 1), gm_min=((-4, 1),) * 3, gm_max=((0, 1),) * 3, gm_w=320, gm_h=180,
                             gm_format=abi.AVIF_PIXEL_FORMAT_YUV420))
    # NaN from degenerate metadata must still fail the call
    out.append(G.GainMapCase(37, 21, gm_max=((2000, 1),) * 3, gm_min=((-2000, 1),) * 3, alt_headroom=(1, 1), headroom=1.0, out_primaries=9, base_offset=((0, 1),) * 3))
    return out


@pytest.mark.gpu
def test_gpu_fast_kernel_and_general_kernel_on_the_same_cases(hip_auto_arithmetic, monkeypatch):
    """The fast apply kernel (one table read per output code, kernels_gainmap.hip) must be the one that runs on its cases, and the general kernel
    -- forced by AVIFHIP_GAINMAP_KERNEL=general -- must give the same bytes there: both against the oracle."""
    from libavif_amd import native

    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    cases = _fast_kernel_cases()
    want = [run(o.oracleRGBImageApplyGainMap, c, 1) for c in cases]
    for forced, name in ((None, "gainmap_apply_fast"), ("general", "gainmap_apply")):
        if forced:
            monkeypatch.setenv("AVIFHIP_GAINMAP_KERNEL", forced)
        bad = []
        for c, (ra, pa, ca) in zip(cases, want):
            rb, pb, cb = run(hip_auto_arithmetic.avifhipRGBImageApplyGainMap, c, C.byref(diag))
            # (the fast kernel has no NaN test: the library keeps calls whose tables could produce one on the general kernel)
            # (8-bit 4:4:4 / 4:0:0 gain maps: the fast kernel converts the map's planes itself, `gainmap_apply_fast<planes>`)
            assert native.last_kernel().split("<")[0] == (name if ra == 0 else "gainmap_apply"), (c.ident(), native.last_kernel())
            if ra != rb or (ra == 0 and not (np.array_equal(pa, pb) and ca[0] == cb[0] and abs(ca[1] - cb[1]) <= 1)):
                bad.append(f"{c.ident()} [{name}]: results {ra}/{rb} clli {ca}/{cb}" +
                           ("" if ra or rb or np.array_equal(pa, pb) else f" {int((pa != pb).sum())} bytes differ"))
        assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    assert sum(r == 0 for r, _, _ in want) >= len(cases) - 1 and any(r == abi.AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE for r, _, _ in want)


def _plane_cases():
    """8-bit 4:4:4 and 4:0:0 gain maps at the base image's size -- what the fast apply kernel converts itself (round 5): every matrix the conversion
    has a transform for (BT.709 / BT.601 / BT.2020 coefficients, identity, YCgCo), both ranges, every pixel-size combination and primaries
    variant of the kernel, rows that end inside a lane's run of four pixels (the last run starts at an odd byte of the planes)."""
    out = []
    k = 0
    for fmt in (abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_PIXEL_FORMAT_YUV400):
        for matrix in (1, 6, 9, 0, 8, 2):
            for rng in (abi.AVIF_RANGE_FULL, abi.AVIF_RANGE_LIMITED):
                if (matrix == 0 and fmt == abi.AVIF_PIXEL_FORMAT_YUV400) or (matrix == 8 and rng == abi.AVIF_RANGE_LIMITED):
                    continue  # (limited-range YCgCo: the conversion refuses it, below)
                bd, od = ((8, 8), (8, 10), (10, 8), (12, 12))[k % 4]
                w, h = ((37, 21), (261, 5), (64, 9), (5, 3), (515, 17), (258, 8), (7, 2))[k % 7]
                out.append(G.GainMapCase(w, h, base_depth=bd, out_depth=od, out_tc=9 if od == 12 else (13, 16, 18, 1, 8)[k % 5], base_primaries=(1, 1, 9, 12)[k % 4],
                                         out_primaries=(1, 9, 9, 1)[k % 4], use_base_color_space=bool(k % 3), alt_primaries=9, gm_format=fmt, gm_matrix=matrix,
                                         gm_range=rng, base_format=(abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGRA, abi.AVIF_RGB_FORMAT_ARGB)[k % 3], seed=1300 + k))
                k += 1
    out.append(G.GainMapCase(1001, 67, base_depth=8, out_depth=10, out_tc=16, out_primaries=9, gm_format=abi.AVIF_PIXEL_FORMAT_YUV400, seed=78))
    out.append(G.GainMapCase(1920, 1080, base_depth=8, out_depth=10, out_tc=16, out_primaries=9, gm_matrix=1, seed=79))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("arithmetic", ["float", "auto"])
def test_gpu_fast_kernel_converts_the_gain_maps_planes_itself(hip, monkeypatch, arithmetic):
    """Round 5: the gain map's own avifImageYUVToRGB inside the apply kernel.  On 8-bit 4:4:4 / 4:0:0 maps the fast kernel must
    read the planes (`gainmap_apply_fast<planes>`), and give the oracle's bytes in both arithmetics -- the reference's fp32 loops and what a
    libavif built with libyuv computes (the matrices libyuv has no constants for fall to the fp32 loops there too); with
    AVIFHIP_GAINMAP_PLANES=0 the same calls take a conversion launch and the RGBA copy (rounds 2-4), same bytes."""
    from libavif_amd import native

    hip.avifhipSetArithmetic(1 if arithmetic == "float" else 0)
    try:
        o = oracle_lib.oracle()
        diag = abi.avifDiagnostics()
        cases = _plane_cases()
        want = [run(o.oracleRGBImageApplyGainMap, c, int(arithmetic == "auto")) for c in cases]
        assert all(r == 0 for r, _, _ in want)
        for planes in (True, False):
            if not planes:
                monkeypatch.setenv("AVIFHIP_GAINMAP_PLANES", "0")
            bad = []
            for c, (ra, pa, ca) in zip(cases, want):
                rb, pb, cb = run(hip.avifhipRGBImageApplyGainMap, c, C.byref(diag))
                name = native.last_kernel()
                # (the full-range identity map is a copy of the samples; YCgCo and "unspecified" are transforms like the others)
                assert name == ("gainmap_apply_fast<planes>" if planes else "gainmap_apply_fast"), (c.ident(), name, arithmetic)
                if ra != rb or not (np.array_equal(pa, pb) and ca[0] == cb[0] and abs(ca[1] - cb[1]) <= 1):
                    bad.append(f"{c.ident()} m{c.gm_matrix} r{c.gm_range} [{name}]: results {ra}/{rb} clli {ca}/{cb}" +
                               ("" if rb or pb is None or np.array_equal(pa, pb) else f" {int((pa != pb).sum())} bytes differ"))
            assert not bad, f"{arithmetic}, planes={planes}: {len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
            # a map whose conversion the reference refuses (limited-range YCgCo, src/reformat.c:128-134) fails the call the same way
            refused = G.GainMapCase(37, 21, gm_matrix=8, gm_range=abi.AVIF_RANGE_LIMITED)
            ra = run(o.oracleRGBImageApplyGainMap, refused, int(arithmetic == "auto"))[0]
            assert ra != 0 and run(hip.avifhipRGBImageApplyGainMap, refused, C.byref(diag))[0] == ra
    finally:
        hip.avifhipSetArithmetic(1)


@pytest.mark.gpu
def test_gpu_buffers_of_a_previous_call_are_kept_when_they_fit(hip_auto_arithmetic):
    """Round 5: avifRGBImageApplyGainMap frees and allocates the tone-mapped pixels, avifRGBImageComputeGainMap the gain map's planes
    (src/gainmap.c:114, :792-793).  A buffer of the right size that a previous call left in the struct stays where it is (releasing memory the
    runtime had pinned costs milliseconds): same bytes as the oracle's, same address; another size gets a buffer of its own."""
    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    out = G.make_output(G.GainMapCase(258, 40, out_depth=10))
    seen = []
    for c in (G.GainMapCase(258, 40, out_depth=10, out_tc=16, seed=5), G.GainMapCase(258, 40, out_depth=10, out_tc=13, out_primaries=9, seed=6),
              G.GainMapCase(515, 33, out_depth=10, out_tc=16, seed=7), G.GainMapCase(64, 9, out_depth=10, out_tc=16, seed=8)):
        ra, pa, ca = run(o.oracleRGBImageApplyGainMap, c, 1)
        base = G.make_base(c)
        gm, keep = G.make_gain_map(c)
        clli = abi.avifContentLightLevelInformationBox(0xFFFF, 0xFFFF)
        out.struct.width, out.struct.height = 1, 1  # (the callee sets them, like the reference)
        rb = hip_auto_arithmetic.avifhipRGBImageApplyGainMap(base.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, out.struct,
                                                            C.byref(clli), C.byref(diag))
        assert ra == rb == 0 and (out.struct.width, out.struct.height) == (c.w, c.h)
        assert np.array_equal(G.output_bytes(out), pa), c.ident()
        seen.append(C.cast(out.struct.pixels, C.c_void_p).value)
    assert seen[0] == seen[1], seen  # kept for the same size (the larger and the smaller image get buffers of their own: wherever malloc puts them)
    libc.free(C.cast(out.struct.pixels, C.c_void_p))
    out.struct.pixels = None

    import test_scale as TS

    cases = [G.ComputeCase(261, 37, alt_primaries=9, seed=21), G.ComputeCase(261, 37, alt_primaries=1, seed=22), G.ComputeCase(64, 33, seed=23)]
    gm, img = G.make_compute_gain_map(cases[0])
    addresses = []
    for c in cases:
        ra, sa = run_compute(o.oracleRGBImageComputeGainMap, c, 1)
        base, alt = G.make_compute_inputs(c)
        want_gm, want_img = G.make_compute_gain_map(c)
        img.struct.width, img.struct.height, img.struct.depth, img.struct.yuvFormat = want_img.struct.width, want_img.struct.height, want_img.struct.depth, want_img.struct.yuvFormat
        rb = hip_auto_arithmetic.avifhipRGBImageComputeGainMap(base.struct, c.base_primaries, c.base_tc, alt.struct, c.alt_primaries, c.alt_tc, C.byref(gm), C.byref(diag))
        assert ra == rb == 0, (c.ident(), ra, rb)
        assert states_equal(sa, gain_map_state(gm, img.struct)), c.ident()
        addresses.append(C.cast(img.struct.yuvPlanes[0], C.c_void_p).value)
        TS.free_owned(want_img.struct)
    assert addresses[0] == addresses[1], addresses
    TS.free_owned(img.struct)


@pytest.mark.gpu
def test_gpu_larger_images_and_16_bit_tables(hip_auto_arithmetic):
    cases = [G.GainMapCase(1001, 333, base_depth=8, out_depth=8, gm_w=500, gm_h=167, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420, out_tc=16, out_primaries=9),
             G.GainMapCase(640, 360, base_depth=10, out_depth=10, base_tc=16, out_tc=13, base_primaries=9, out_primaries=1, headroom=1.0,
                           base_headroom=(4, 1), alt_headroom=(0, 1), gm_min=((-4, 1),) * 3, gm_max=((0, 1),) * 3),
             G.GainMapCase(515, 129, base_depth=16, out_depth=16, out_tc=18, gm_depth=10, gm_format=abi.AVIF_PIXEL_FORMAT_YUV444),
             G.GainMapCase(515, 129, base_depth=8, out_depth=16, out_float=True, out_format=abi.AVIF_RGB_FORMAT_RGBA, out_tc=8),
             G.GainMapCase(320, 200, base_depth=12, out_depth=12, headroom=0.0, out_tc=16, out_primaries=12)]
    _compare_gpu(hip_auto_arithmetic, cases, libyuv_build=True)


@pytest.mark.gpu
def test_gpu_device_resident(hip_auto_arithmetic):
    """avifhipRGBImageApplyGainMapAsync: base pixels, gain-map planes and tone-mapped pixels in HBM."""
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    for c in G.cases(0, seed=4)[:60:3] + [G.GainMapCase(777, 211, gm_w=389, gm_h=106, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420, out_tc=16, out_depth=10)]:
        ra, pa, ca = run(o.oracleRGBImageApplyGainMap, c, 1)
        base = G.make_base(c)
        gm, keep = G.make_gain_map(c)
        dbase = device.DeviceRGB(base, upload=True)
        dgm_img = device.DeviceYUV(keep)
        gm.image = C.pointer(dgm_img.struct)
        out = abi.make_rgb(c.w, c.h, c.out_depth, c.out_format, is_float=c.out_float, avoid_libyuv=False)
        dout = device.DeviceRGB(out, upload=True)
        clli = abi.avifContentLightLevelInformationBox(0xFFFF, 0xFFFF)
        diag = abi.avifDiagnostics()
        rb = hip_auto_arithmetic.avifhipRGBImageApplyGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries,
                                                                 c.out_tc, dout.struct, C.byref(clli), C.byref(diag), None)
        assert ra == rb, (c.ident(), ra, rb, diag.error)
        if ra == 0:
            # (round 6: the light levels of an asynchronous call are filled by the next avifhipSynchronize on its stream)
            assert hip_auto_arithmetic.avifhipSynchronize(None) == 0
            dout.download_into_host()
            wb = c.w * abi.rgb_pixel_size(c.out_format, c.out_depth)
            assert np.array_equal(out.pixels[:, :wb], pa[:, :wb]), (c.ident(), native.last_kernel())
            assert clli.maxCLL == ca[0] and abs(clli.maxPALL - ca[1]) <= 1, (c.ident(), ca, (clli.maxCLL, clli.maxPALL))


@pytest.mark.gpu
def test_gpu_device_resident_without_light_levels_and_in_place(hip_auto_arithmetic):
    """avifhipRGBImageApplyGainMapAsync with clli = NULL returns with its work enqueued (round 5: nothing of the answer depends on the pixels
    when the fast kernel's precondition rules NaNs out) -- the pixels after a synchronisation are the oracle's; and with the tone-mapped pixels
    ON TOP of the base pixels (same layout) the call keeps off the fast kernel, whose lanes re-read their neighbours' pixels at row ends."""
    from libavif_amd import device, native

    o = oracle_lib.oracle()
    for c in [G.GainMapCase(258, 40, base_depth=8, out_depth=8, out_tc=13, seed=11), G.GainMapCase(515, 33, base_depth=10, out_depth=10, out_tc=16, out_primaries=9, seed=12),
              G.GainMapCase(1001, 67, base_depth=8, out_depth=10, out_tc=16, out_primaries=9, seed=77)]:
        ra, pa, ca = run(o.oracleRGBImageApplyGainMap, c, 1)
        assert ra == 0
        base = G.make_base(c)
        gm, keep = G.make_gain_map(c)
        dbase = device.DeviceRGB(base, upload=True)
        dgm_img = device.DeviceYUV(keep)
        gm.image = C.pointer(dgm_img.struct)
        out = abi.make_rgb(c.w, c.h, c.out_depth, c.out_format, is_float=c.out_float, avoid_libyuv=False)
        dout = device.DeviceRGB(out, upload=True)
        diag = abi.avifDiagnostics()
        wb = c.w * abi.rgb_pixel_size(c.out_format, c.out_depth)
        rb = hip_auto_arithmetic.avifhipRGBImageApplyGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries,
                                                                 c.out_tc, dout.struct, None, C.byref(diag), None)
        assert rb == 0 and native.last_kernel() == "gainmap_apply_fast<planes>", (c.ident(), rb, native.last_kernel())
        native.check(hip_auto_arithmetic.avifhipSynchronize(None))
        dout.download_into_host()
        assert np.array_equal(out.pixels[:, :wb], pa[:, :wb]), c.ident()
        if c.base_depth == c.out_depth and c.base_format == c.out_format:
            # in place: the output struct points at the base pixels
            inplace = abi.avifRGBImage()
            C.memmove(C.byref(inplace), C.byref(dbase.struct), C.sizeof(abi.avifRGBImage))
            rb = hip_auto_arithmetic.avifhipRGBImageApplyGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries,
                                                                     c.out_tc, inplace, None, C.byref(diag), None)
            assert rb == 0 and native.last_kernel() == "gainmap_apply", (c.ident(), rb, native.last_kernel())
            native.check(hip_auto_arithmetic.avifhipSynchronize(None))
            dbase.download_into_host()
            assert np.array_equal(base.pixels[:, :wb], pa[:, :wb]), c.ident() + " in place"


@pytest.mark.gpu
def test_gpu_async_application_then_a_host_call_with_other_parameters(hip_auto_arithmetic):
    """avifhipRGBImageApplyGainMapAsync on a stream of the caller's may return with its kernel pending; that kernel reads the
    THREAD's scratch (the gain map scaled and converted to RGBA, the lookup tables).  A host-resident application or computation on the same
    thread right behind it -- other sizes, other curves: it rewrites all of that on the library's own stream -- must wait for the pending
    kernel.  The asynchronous call's pixels, read after its stream has drained, are the oracle's."""
    from libavif_amd import device, native

    lib, o = hip_auto_arithmetic, oracle_lib.oracle()
    stream = lib.avifhipStreamCreate()
    assert stream
    diag = abi.avifDiagnostics()
    # a scaled 4:2:0 gain map (scaled planes and an RGBA copy of them in scratch), large enough that its kernels are still running when
    # the call returns
    first = G.GainMapCase(2560, 1440, base_depth=8, out_depth=10, out_tc=16, out_primaries=9, gm_w=1280, gm_h=720, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420, seed=31)
    others = [G.GainMapCase(640, 360, base_depth=10, out_depth=8, out_tc=13, out_primaries=1, gm_w=320, gm_h=180, gm_format=abi.AVIF_PIXEL_FORMAT_YUV444,
                            gm_gamma=((2, 1), (2, 1), (2, 1)), headroom=1.5, seed=32),
              G.GainMapCase(1920, 1080, base_depth=8, out_depth=8, out_tc=1, gm_w=480, gm_h=270, gm_format=abi.AVIF_PIXEL_FORMAT_YUV400, headroom=2.0, seed=33)]
    ra, pa, _ = run(o.oracleRGBImageApplyGainMap, first, 1)
    assert ra == 0
    base = G.make_base(first)
    gm, keep = G.make_gain_map(first)
    dbase, dgm_img = device.DeviceRGB(base, upload=True), device.DeviceYUV(keep)
    gm.image = C.pointer(dgm_img.struct)
    wb = first.w * abi.rgb_pixel_size(first.out_format, first.out_depth)
    try:
        for rep in range(6):
            other = others[rep % len(others)]
            out = abi.make_rgb(first.w, first.h, first.out_depth, first.out_format, is_float=first.out_float, avoid_libyuv=False)
            dout = device.DeviceRGB(out, upload=True)
            rb = lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, first.base_primaries, first.base_tc, C.byref(gm), first.headroom, first.out_primaries,
                                                      first.out_tc, dout.struct, None, C.byref(diag), stream)
            assert rb == 0, (rb, diag.error)
            if rep % 2 == 0:
                rc, pc, _ = run(lib.avifhipRGBImageApplyGainMap, other, C.byref(diag))
                rw, pw, _ = run(o.oracleRGBImageApplyGainMap, other, 1)
                assert rc == rw == 0 and np.array_equal(pc, pw), other.ident()
            else:
                cc = G.compute_cases(0, seed=11)[rep]
                rc, sc = run_compute(lib.avifhipRGBImageComputeGainMap, cc, C.byref(diag))
                rw, sw = run_compute(o.oracleRGBImageComputeGainMap, cc, 1)
                assert rc == rw and states_equal(sc, sw), cc.ident()
            native.check(lib.avifhipSynchronize(stream))
            dout.download_into_host()
            assert np.array_equal(out.pixels[:, :wb], pa[:, :wb]), f"repetition {rep}: the pending application read scratch a later call rewrote"
    finally:
        lib.avifhipStreamDestroy(stream)


@pytest.mark.gpu
def test_gpu_argument_errors(hip):
    diag = abi.avifDiagnostics()
    fn = hip.avifhipRGBImageApplyGainMap
    assert run(fn, G.GainMapCase(8, 8, headroom=-1.0), C.byref(diag))[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert b"hdrHeadroom" in diag.error
    assert run(fn, G.GainMapCase(8, 8, gm_gamma=((0, 1), (1, 1), (1, 1))), C.byref(diag))[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(fn, G.GainMapCase(8, 8, gm_min=((2, 1), (0, 1), (0, 1)), gm_max=((1, 1), (1, 1), (1, 1))), C.byref(diag))[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    nan_case = G.GainMapCase(37, 21, gm_max=((2000, 1),) * 3, gm_min=((-2000, 1),) * 3, alt_headroom=(1, 1), headroom=1.0, out_primaries=9, base_offset=((0, 1),) * 3)
    assert run(fn, nan_case, C.byref(diag))[0] == abi.AVIF_RESULT_INVALID_TONE_MAPPED_IMAGE


@pytest.mark.gpu
def test_gpu_yuv_base_image(hip_auto_arithmetic):
    """avifhipImageApplyGainMap (reference avifImageApplyGainMap, src/gainmap.c:317-355): the base image arrives as YUV."""
    import harness as H

    o = oracle_lib.oracle()
    for c, yc in [(G.GainMapCase(320, 180, base_tc=13, base_primaries=1, out_tc=16, out_primaries=9, out_depth=10, gm_w=160, gm_h=90),
                   H.Y2RCase(320, 180, yuv_depth=8, yuv_format=3, yuv_range=1, matrix=6, avoid_libyuv=False)),
                  (G.GainMapCase(201, 77, base_depth=10, base_tc=14, base_primaries=9, out_tc=13, out_primaries=1, out_depth=8, headroom=1.0),
                   H.Y2RCase(201, 77, yuv_depth=10, yuv_format=1, yuv_range=0, matrix=9, avoid_libyuv=False))]:
        img = H.make_y2r_inputs(yc)
        img.struct.colorPrimaries, img.struct.transferCharacteristics = c.base_primaries, c.base_tc
        base = abi.make_rgb(c.w, c.h, yc.yuv_depth, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
        assert o.oracleLibyuvImageYUVToRGB(img.struct, base.struct) == 0
        gm, keep = G.make_gain_map(c)
        want, got = G.make_output(c), G.make_output(c)
        clli_a, clli_b = abi.avifContentLightLevelInformationBox(), abi.avifContentLightLevelInformationBox()
        diag = abi.avifDiagnostics()
        assert o.oracleRGBImageApplyGainMap(base.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, want.struct,
                                            C.byref(clli_a), 1) == 0
        assert hip_auto_arithmetic.avifhipImageApplyGainMap(img.struct, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, got.struct, C.byref(clli_b),
                                                            C.byref(diag)) == 0, diag.error
        assert np.array_equal(G.output_bytes(want), G.output_bytes(got)), c.ident()
        assert clli_a.maxCLL == clli_b.maxCLL and abs(clli_a.maxPALL - clli_b.maxPALL) <= 1
        libc.free(C.cast(want.struct.pixels, C.c_void_p))
        libc.free(C.cast(got.struct.pixels, C.c_void_p))


@pytest.mark.gpu
def test_gpu_compute_equals_oracle_default_arithmetic(hip_auto_arithmetic):
    """avifhipRGBImageComputeGainMap: metadata fractions and gain-map planes identical to the oracle's (the default-build flavour:
    the gain map's RGB -> YUV conversion as a libavif built with libyuv computes it)."""
    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.compute_cases(120, seed=12) + [G.ComputeCase(1001, 333, gm_w=500, gm_h=167, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420),
                                             G.ComputeCase(640, 360, alt_primaries=9, gm_format=abi.AVIF_PIXEL_FORMAT_YUV400, gm_depth=10)]
    for c in cases:
        ra, sa = run_compute(o.oracleRGBImageComputeGainMap, c, 1)
        rb, sb = run_compute(hip_auto_arithmetic.avifhipRGBImageComputeGainMap, c, C.byref(diag))
        if ra != rb or not states_equal(sa, sb):
            bad.append(f"{c.ident()}: results {ra}/{rb}" + ("" if sa is None or sb is None else f" meta equal {sa[0] == sb[0]} size {sa[1]}/{sb[1]}"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.gpu
def test_gpu_compute_equals_oracle_fp32_arithmetic(hip):
    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.compute_cases(60, seed=13)
    for c in cases:
        ra, sa = run_compute(o.oracleRGBImageComputeGainMap, c, 0)
        rb, sb = run_compute(hip.avifhipRGBImageComputeGainMap, c, C.byref(diag))
        if ra != rb or not states_equal(sa, sb):
            bad.append(f"{c.ident()}: results {ra}/{rb}" + ("" if sa is None or sb is None else f" meta equal {sa[0] == sb[0]} size {sa[1]}/{sb[1]}"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.gpu
def test_gpu_compute_from_yuv_images(hip_auto_arithmetic):
    """avifhipImageComputeGainMap (reference avifImageComputeGainMap, src/gainmap.c:843-912): both renditions as YUV images; also fills
    the alternate image's colorimetry into the gain map."""
    import harness as H
    import test_scale as TS

    o = oracle_lib.oracle()
    yb = H.Y2RCase(200, 120, yuv_depth=8, yuv_format=3, yuv_range=1, matrix=6, avoid_libyuv=False, seed=5)
    ya = H.Y2RCase(200, 120, yuv_depth=10, yuv_format=1, yuv_range=0, matrix=9, avoid_libyuv=False, seed=6)
    base_img, alt_img = H.make_y2r_inputs(yb), H.make_y2r_inputs(ya)
    base_img.struct.colorPrimaries, base_img.struct.transferCharacteristics = 1, 13
    alt_img.struct.colorPrimaries, alt_img.struct.transferCharacteristics, alt_img.struct.matrixCoefficients = 9, 16, 9
    base = abi.make_rgb(200, 120, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    alt = abi.make_rgb(200, 120, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    assert o.oracleLibyuvImageYUVToRGB(base_img.struct, base.struct) == 0 and o.oracleLibyuvImageYUVToRGB(alt_img.struct, alt.struct) == 0
    c = G.ComputeCase(200, 120, gm_w=100, gm_h=60, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420)
    gm_a, img_a = G.make_compute_gain_map(c)
    gm_b, img_b = G.make_compute_gain_map(c)
    assert o.oracleRGBImageComputeGainMap(base.struct, 1, 13, alt.struct, 9, 16, C.byref(gm_a), 1) == 0
    diag = abi.avifDiagnostics()
    assert hip_auto_arithmetic.avifhipImageComputeGainMap(base_img.struct, alt_img.struct, C.byref(gm_b), C.byref(diag)) == 0, diag.error
    assert states_equal(gain_map_state(gm_a, img_a.struct), gain_map_state(gm_b, img_b.struct))
    assert (gm_b.altColorPrimaries, gm_b.altTransferCharacteristics, gm_b.altMatrixCoefficients, gm_b.altDepth, gm_b.altPlaneCount) == (9, 16, 9, 10, 3)
    TS.free_owned(img_a.struct)
    TS.free_owned(img_b.struct)


@pytest.mark.gpu
def test_gpu_compute_device_resident(hip_auto_arithmetic):
    """avifhipRGBImageComputeGainMapAsync (round 6): both renditions and the gain map's planes in device memory, the caller's stream.  Metadata
    and planes equal the oracle's (src/gainmap.c:535-843) -- the same cases as the host entry point, larger images with odd widths included
    (the kernels' four-pixel lanes, their row ends, the histogram's vector tail)."""
    import test_scale as TS
    from libavif_amd import device

    lib = hip_auto_arithmetic
    o = oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    stream = lib.avifhipStreamCreate()
    assert stream
    bad = []
    cases = G.compute_cases(40, seed=21) + [G.ComputeCase(1001, 333, gm_w=500, gm_h=167, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420),
                                            G.ComputeCase(1027, 301, alt_primaries=9), G.ComputeCase(2050, 130, gm_format=abi.AVIF_PIXEL_FORMAT_YUV400, gm_depth=10),
                                            G.ComputeCase(1920, 1080, alt_primaries=9, seed=5), G.ComputeCase(1030, 517, base_format=abi.AVIF_RGB_FORMAT_RGB, alt_format=abi.AVIF_RGB_FORMAT_BGRA)]
    try:
        for c in cases:
            ra, sa = run_compute(o.oracleRGBImageComputeGainMap, c, 1)
            base, alt = G.make_compute_inputs(c)
            dbase, dalt = device.DeviceRGB(base, upload=True), device.DeviceRGB(alt, upload=True)
            gw, gh = c.gm_w or c.w, c.gm_h or c.h
            # (with an alpha plane: avifImageRGBToYUV gives the reference's gain-map image one -- opaque -- because the codes are RGBA, src/gainmap.c:792-800)
            host_gm = abi.make_yuv(gw, gh, c.gm_depth, c.gm_format, c.gm_range, c.gm_matrix, with_alpha=True)
            dgm = device.DeviceYUV(host_gm, upload=False)
            gm = abi.avifGainMap()
            gm.image = C.pointer(dgm.struct)
            rb = lib.avifhipRGBImageComputeGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, dalt.struct, c.alt_primaries, c.alt_tc, C.byref(gm), C.byref(diag), stream)
            sb = None
            if rb == 0:
                assert lib.avifhipSynchronize(stream) == 0
                dgm.download_into_host()
                sb = gain_map_state(gm, host_gm.struct)
            if ra != rb or not states_equal(sa, sb):
                bad.append(f"{c.ident()}: results {ra}/{rb}" + ("" if sa is None or sb is None else f" meta equal {sa[0] == sb[0]} size {sa[1]}/{sb[1]}"))
    finally:
        lib.avifhipStreamDestroy(stream)
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])
    # planes the caller did not allocate are the caller's error
    c = cases[0]
    base, alt = G.make_compute_inputs(c)
    dbase, dalt = device.DeviceRGB(base, upload=True), device.DeviceRGB(alt, upload=True)
    gm, img = G.make_compute_gain_map(c)
    assert lib.avifhipRGBImageComputeGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, dalt.struct, c.alt_primaries, c.alt_tc, C.byref(gm), C.byref(diag), None) == abi.AVIF_RESULT_INVALID_ARGUMENT


@pytest.mark.gpu
def test_gpu_exact_light_levels(hip_auto_arithmetic):
    """avifhipSetExactLightLevels(1): maxPALL is the reference's own -- one fp32 accumulator over the pixels in raster order (src/gainmap.c:223,
    293, 304) -- not the fp64 partial sums' rounding of it: no tolerance left, on the host entry point and on the device-resident one, small
    images and an 8-megapixel one (where the fp32 sum is past 2^23 and every addition rounds)."""
    from libavif_amd import device

    lib = hip_auto_arithmetic
    o = oracle_lib.oracle()
    lib.avifhipSetExactLightLevels(1)
    try:
        cases = G.cases(0, seed=31)[:40:2] + [G.GainMapCase(3840, 2160, seed=7), G.GainMapCase(1001, 333, gm_w=500, gm_h=167, gm_format=abi.AVIF_PIXEL_FORMAT_YUV420)]
        for c in cases:
            ra, pa, ca = run(o.oracleRGBImageApplyGainMap, c, 1)
            diag = abi.avifDiagnostics()
            rb, pb, cb = run(lib.avifhipRGBImageApplyGainMap, c, C.byref(diag))
            assert ra == rb and ca == cb, (c.ident(), ra, rb, ca, cb)
            if ra != 0:
                continue
            assert np.array_equal(pa, pb), c.ident()
            base = G.make_base(c)
            gm, keep = G.make_gain_map(c)
            dbase, dgm_img = device.DeviceRGB(base, upload=True), device.DeviceYUV(keep)
            gm.image = C.pointer(dgm_img.struct)
            dout = device.DeviceRGB(abi.make_rgb(c.w, c.h, c.out_depth, c.out_format, is_float=c.out_float, avoid_libyuv=False), upload=True)
            clli = abi.avifContentLightLevelInformationBox(0xFFFF, 0xFFFF)
            assert lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, dout.struct,
                                                        C.byref(clli), C.byref(diag), None) == 0
            assert (clli.maxCLL, clli.maxPALL) == ca, (c.ident(), ca, (clli.maxCLL, clli.maxPALL))
    finally:
        lib.avifhipSetExactLightLevels(0)
