"""The product's HOST logic on the CPU (no GPU needed): the pure-host planners behind the side-path kernels --
libavif_amd/csrc/scale_plan.cpp (plane-scaling schedules) and gainmap_plan.cpp (transfer functions, primaries matrices, fractions,
the fp32 steps of the quantised output transfer functions) -- compiled with g++ into a small test library and compared with the
oracle (itself pinned against the reference).  What the kernels do with the tables is covered by the -m gpu tests."""
import ctypes as C
import os
import random
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "libavif_amd" / "csrc"
SO = ROOT / "tests" / "tools" / "libhostlogic.so"


@pytest.fixture(scope="module")
def host():
    srcs = [ROOT / "tests" / "tools" / "hostlogic.cpp", CSRC / "scale_plan.cpp", CSRC / "gainmap_plan.cpp", CSRC / "plan.cpp"]
    deps = srcs + [CSRC / "scale_plan.h", CSRC / "gainmap_plan.h", CSRC / "gainmap_steps.h", CSRC / "plan.h"]
    if not SO.exists() or any(d.stat().st_mtime > SO.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", f"-I{CSRC}", f"-I{ROOT / 'include'}", "-o", os.fspath(SO)] + [os.fspath(s) for s in srcs],
                       check=True, capture_output=True)
    lib = C.CDLL(os.fspath(SO))
    ip = C.POINTER(C.c_int)
    lib.hostScaleSchedule.restype, lib.hostScaleSchedule.argtypes = C.c_int, [C.c_int] * 5 + [ip] * 5
    lib.hostCoverOfCrop.restype = None
    lib.hostCoverOfCrop.argtypes = [C.c_uint32] * 4 + [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32 * 4)]
    lib.hostScaleSpecialisation.restype, lib.hostScaleSpecialisation.argtypes = C.c_int, [C.c_int] * 5
    lib.hostTransferFunction.restype, lib.hostTransferFunction.argtypes = C.c_float, [C.c_int, C.c_int, C.c_float]
    lib.hostPrimariesMatrix.restype, lib.hostPrimariesMatrix.argtypes = C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double * 9)]
    lib.hostDoubleToSignedFraction.restype, lib.hostDoubleToSignedFraction.argtypes = C.c_int, [C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    lib.hostDoubleToUnsignedFraction.restype, lib.hostDoubleToUnsignedFraction.argtypes = C.c_int, [C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.hostOutputSteps.restype = C.c_uint32
    lib.hostOutputSteps.argtypes = [C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.hostChooseMathPrimaries.restype, lib.hostChooseMathPrimaries.argtypes = C.c_int, [C.c_int, C.c_int]
    lib.hostLocatorCodes.restype = C.c_uint32
    lib.hostLocatorCodes.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.hostCheckBucketSteps.restype = C.c_int
    lib.hostCheckBucketSteps.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(C.c_int)]
    lib.hostCheckStepSearch.restype, lib.hostCheckStepSearch.argtypes = C.c_int, [C.c_int, C.c_uint32]
    lib.hostCheckCodeSteps.restype = C.c_int
    lib.hostCheckCodeSteps.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int, C.c_uint32]
    return lib


def test_scale_schedules_equal_the_oracles(host):
    """Mode and every per-column / per-row schedule entry, over random plane geometries (both sample widths)."""
    o = oracle_lib.oracle()
    o.oracleScaleSchedule.restype, o.oracleScaleSchedule.argtypes = C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_int)] * 5
    rnd = random.Random(5)
    checked = 0
    for k in range(40000):
        if k % 3 == 0:
            sw, sh, dw, dh = rnd.randint(1, 300), rnd.randint(1, 300), rnd.randint(1, 300), rnd.randint(1, 300)
        elif k % 3 == 1:
            sw, sh = rnd.randint(1, 5000), rnd.randint(1, 40)
            dw, dh = max(1, int(sw * rnd.choice((0.1, 0.25, 1 / 3, 0.5, 0.6, 1, 1.5, 2, 3)))), max(1, int(sh * rnd.choice((0.25, 0.5, 1, 2, 3))))
        else:
            sw, sh = rnd.randint(1, 64), rnd.randint(1, 64)
            dw, dh = max(1, 2 * sw - rnd.choice((0, 1))), max(1, 2 * sh - rnd.choice((0, 1)))
        wide = k & 1
        a = [(C.c_int * n)() for n in (dw, dw, dh, dh, dh)]
        b = [(C.c_int * n)() for n in (dw, dw, dh, dh, dh)]
        ma = o.oracleScaleSchedule(sw, sh, dw, dh, wide, *a)
        mb = host.hostScaleSchedule(sw, sh, dw, dh, wide, *b)
        assert ma == mb, (sw, sh, dw, dh, wide, ma, mb)
        for x, y, name in zip(a, b, ("colA", "colB", "rowA", "rowB", "rowF")):
            if ma == 0 and name in ("colB", "rowB", "rowF"):
                continue  # point sampling reads only the first column / row entry
            if ma in (3, 4) and name == "rowF":
                continue  # boxes and the 2x upsamplers carry no row fraction
            assert list(x) == list(y), (sw, sh, dw, dh, wide, ma, name)
        checked += 1
    assert checked == 40000


def test_transfer_functions_equal_the_oracles(host):
    o = oracle_lib.oracle()
    rng = np.random.default_rng(1)
    values = np.concatenate([rng.uniform(-0.5, 1.5, 4000), rng.uniform(0, 60, 500), 10.0 ** rng.uniform(-9, 0, 2000), [0.0, -0.0, 1.0, 0.5, 1 / 12, 0.018053968510807,
                             4.5 * 0.018053968510807, float("inf"), -float("inf")]]).astype(np.float32)
    for tc in (1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 3, 99):
        for direction in (0, 1):
            for v in values:
                a, b = o.oracleTransferFunction(tc, direction, float(v)), host.hostTransferFunction(tc, direction, float(v))
                assert np.float32(a).tobytes() == np.float32(b).tobytes() or (a != a and b != b), (tc, direction, float(v), a, b)


def test_primaries_and_fractions_equal_the_oracles(host):
    o = oracle_lib.oracle()
    o.oracleDoubleToSignedFraction.restype = C.c_int
    o.oracleDoubleToUnsignedFraction.restype = C.c_int
    from libavif_amd import abi

    prim = (1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 22, 99)
    for a in prim:
        for b in prim:
            ma, mb = (C.c_double * 9)(), (C.c_double * 9)()
            ra, rb = o.oracleColorPrimariesComputeRGBToRGBMatrix(a, b, C.byref(ma)), host.hostPrimariesMatrix(a, b, C.byref(mb))
            assert bool(ra) == bool(rb) and (not ra or list(ma) == list(mb)), (a, b)
    rnd = random.Random(2)
    for _ in range(20000):
        v = rnd.choice((rnd.uniform(-8, 8), rnd.uniform(0, 1e-3), rnd.uniform(-1e6, 1e6), float(np.float32(rnd.uniform(-6, 6))), 0.0, 1.0, 1 / 64, 1e12, -3e9))
        fs, fu = abi.avifSignedFraction(), abi.avifUnsignedFraction()
        n, d, un, ud = C.c_int32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        o.oracleDoubleToSignedFraction.argtypes = [C.c_double, C.POINTER(abi.avifSignedFraction)]
        o.oracleDoubleToUnsignedFraction.argtypes = [C.c_double, C.POINTER(abi.avifUnsignedFraction)]
        ra, rb = o.oracleDoubleToSignedFraction(v, C.byref(fs)), host.hostDoubleToSignedFraction(v, C.byref(n), C.byref(d))
        assert ra == rb and (not ra or (fs.n, fs.d) == (n.value, d.value)), v
        ra, rb = o.oracleDoubleToUnsignedFraction(v, C.byref(fu)), host.hostDoubleToUnsignedFraction(v, C.byref(un), C.byref(ud))
        assert ra == rb and (not ra or (fu.n, fu.d) == (un.value, ud.value)), v


@pytest.mark.parametrize("tc", [1, 4, 8, 9, 11, 12, 13, 16, 18])
@pytest.mark.parametrize("depth,is_float", [(8, 0), (10, 0), (16, 1)])
def test_output_steps_reproduce_the_quantised_transfer_function(host, tc, depth, is_float):
    """For random linear values x, the code found by searching the host-built steps equals quantise(clamp(linearToGamma(x))) computed
    directly with the oracle's transfer function -- the property the gain-map kernel's exactness rests on."""
    o = oracle_lib.oracle()
    cap = 2 * 65536
    steps = (C.c_float * cap)()
    max_code = C.c_uint32()
    entries = host.hostOutputSteps(tc, depth, is_float, steps, cap, C.byref(max_code))
    T = np.frombuffer(steps, dtype=np.float32, count=2 * entries).copy()
    rng = np.random.default_rng(tc * 100 + depth)
    xs = np.concatenate([rng.uniform(-0.3, 1.3, 3000), 10.0 ** rng.uniform(-8, 2, 3000), -(10.0 ** rng.uniform(-8, 0, 500)), [0.0, -0.0, 1.0, 1e30, -1e30]]).astype(np.float32)
    max_f = np.float32((1 << depth) - 1)
    for x in xs:
        g = np.float32(o.oracleTransferFunction(tc, 1, float(x)))
        v = np.float32(min(np.float32(1.0), max(np.float32(0.0), g)))  # avifNanSafeClamp
        if is_float:
            want = int((np.float32(v * np.float32(1.9259299444e-34)).view(np.uint32) >> 13) & 0xFFFF)
        else:
            want = int(np.float32(np.float32(0.5) + v * max_f))
        piece = T[:entries] if x < 0 else T[entries:]
        with np.errstate(invalid="ignore"):
            got = int(np.nonzero(piece[: max_code.value + 1] <= x)[0].max())
        assert got == want, (tc, depth, is_float, float(x), got, want)


@pytest.mark.parametrize("tc", [1, 4, 5, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18])
@pytest.mark.parametrize("depth", [8, 10, 12])
def test_output_locator_reproduces_the_quantised_transfer_function(host, tc, depth):
    """The fast apply kernel finds the output code with ONE table read (GainMapSteps::locator).  For random linear values, for every
    step and for its predecessor, the locator must give quantise(clamp(linearToGamma(x))) as the oracle's transfer function computes
    it; a curve whose x < 0 piece reaches above code 0 (BT.1361) must not have a locator."""
    o = oracle_lib.oracle()
    cap = 2 * 65536
    steps = (C.c_float * cap)()
    max_code = C.c_uint32()
    entries = host.hostOutputSteps(tc, depth, 0, steps, cap, C.byref(max_code))
    T = np.frombuffer(steps, dtype=np.float32, count=2 * entries).copy()
    pos = T[entries: entries + max_code.value + 1]
    finite = pos[1:][np.isfinite(pos[1:])]
    rng = np.random.default_rng(tc * 1000 + depth)
    xs = np.concatenate([rng.uniform(-0.3, 1.3, 4000), 10.0 ** rng.uniform(-10, 2, 4000), -(10.0 ** rng.uniform(-8, 0, 500)), finite,
                         np.nextafter(finite, np.float32(-np.inf)), np.nextafter(finite, np.float32(np.inf)),
                         [0.0, -0.0, 1.0, 1e30, -1e30, np.inf, -np.inf, 1e-45, 1e-38]]).astype(np.float32)
    codes = (C.c_uint32 * len(xs))()
    shift = C.c_uint32()
    buckets = host.hostLocatorCodes(tc, depth, xs.ctypes.data_as(C.POINTER(C.c_float)), len(xs), codes, C.byref(shift))
    if tc == 12:
        assert buckets == 0, "BT.1361 reaches codes above 0 for negative x: it cannot use the locator"
        return
    if buckets == 0:
        assert depth > 8, (tc, depth)  # only the finer tables may outgrow the LDS budget (the general kernel serves those)
        return
    assert buckets <= 12288 and 6 <= shift.value <= 19
    max_f = np.float32((1 << depth) - 1)
    got = np.frombuffer(codes, dtype=np.uint32)
    for x, g in zip(xs[:9000], got[:9000]):  # against the transfer function itself
        v = np.float32(min(np.float32(1.0), max(np.float32(0.0), np.float32(o.oracleTransferFunction(tc, 1, float(x))))))
        assert int(g) == int(np.float32(np.float32(0.5) + v * max_f)), (tc, depth, float(x), int(g))
    with np.errstate(invalid="ignore"):  # against the step search the general kernel does, on everything
        want = np.array([0 if x < 0 else int(np.nonzero(pos <= x)[0].max()) for x in xs])
    assert np.array_equal(got, want), (tc, depth, xs[got != want][:5], got[got != want][:5], want[got != want][:5])


def test_gain_map_math_primaries_choice(host):
    """avifChooseColorSpaceForGainMapMath: the larger colour space; checked through the oracle's computation (metadata useBaseColorSpace)."""
    assert host.hostChooseMathPrimaries(1, 1) == 1
    assert host.hostChooseMathPrimaries(1, 9) == 9 and host.hostChooseMathPrimaries(9, 1) == 9  # BT.2020 contains BT.709
    assert host.hostChooseMathPrimaries(12, 1) == 12


def test_gain_map_computation_step_tables(host):
    """Gain-map computation rests on two host-built tables over the fp32 ratio: the outlier histogram's bucket steps and the final
    code steps.  Searching them must give what the reference's formulas give directly, also right at and next to the steps."""
    rnd = random.Random(9)
    with_buckets = 0
    for k in range(60):
        sign = rnd.choice((1.0, -1.0))
        min_r = 10.0 ** rnd.uniform(-3, 0.5)
        max_r = min_r * 10.0 ** rnd.uniform(0.01, 3)
        nb = C.c_int()
        assert host.hostCheckBucketSteps(sign, min_r, max_r, rnd.choice((10, 5000, 2_000_000, 33_177_600)), 20000, k, C.byref(nb)) == 0, (sign, min_r, max_r)
        with_buckets += nb.value > 0
        lo, hi = sorted((sign * np.log2(min_r), sign * np.log2(max_r)))
        cut = (hi - lo) * rnd.uniform(0, 0.2)
        assert host.hostCheckCodeSteps(sign, min_r, max_r, lo + cut, hi - cut * rnd.uniform(0, 1), rnd.choice((1.0, 1.0, 0.5, 2.2)), rnd.choice((8, 10, 12)), 20000, k) == 0
    assert with_buckets > 30


def test_scale_specialisations_follow_from_the_schedules(host):
    """The doubling kernel and the exact-box kernel ignore the schedule tables: they may only be chosen when the tables say exactly what the
    kernels compute -- near = k >> 1 with the far neighbour one step away, clamped, the LAST column's far neighbour itself (upsample2Axis);
    boxes of N x N on the N-grid -- and for 8-bit samples only (the 16-bit dispatch order differs)."""
    rnd = random.Random(11)
    seen = {"doubling": 0, 4: 0, 8: 0, "none": 0}
    for k in range(6000):
        kind = k % 4
        if kind == 0:
            sw, sh = rnd.randint(1, 700), rnd.randint(1, 90)
            dw, dh = max(1, 2 * sw - rnd.choice((0, 1))), max(1, 2 * sh - rnd.choice((0, 1)))
        elif kind == 1:
            n = rnd.choice((4, 8))
            dw, dh = rnd.randint(1, 300), rnd.randint(1, 60)
            sw, sh = n * dw, n * dh
        elif kind == 2:
            n = rnd.choice((3, 4, 5, 8))
            dw, dh = rnd.randint(1, 200), rnd.randint(1, 40)
            sw, sh = n * dw + rnd.choice((0, 0, 1, 3)), n * dh + rnd.choice((0, 0, 2))
        else:
            sw, sh, dw, dh = rnd.randint(1, 400), rnd.randint(1, 60), rnd.randint(1, 400), rnd.randint(1, 60)
        for wide in (0, 1):
            a = [(C.c_int * n_)() for n_ in (dw, dw, dh, dh, dh)]
            mode = host.hostScaleSchedule(sw, sh, dw, dh, wide, *a)
            spec = host.hostScaleSpecialisation(sw, sh, dw, dh, wide)
            doubling, box = spec & 1, spec >> 8
            colA, colB, rowA, rowB = (list(x) for x in a[:4])
            if doubling:
                assert mode == 4 and not box
                assert colA == [i >> 1 for i in range(dw)] and rowA == [j >> 1 for j in range(dh)]
                assert colB == [(i >> 1) if i == dw - 1 else min(max((i >> 1) + (1 if i & 1 else -1), 0), sw - 1) for i in range(dw)], (sw, dw)
                assert rowB == [min(max((j >> 1) + (1 if j & 1 else -1), 0), sh - 1) for j in range(dh)], (sh, dh)
                seen["doubling"] += 1
            elif box:
                assert mode == 3 and not wide and box in (4, 8) and sw == box * dw and sh == box * dh
                assert colA == [box * i for i in range(dw)] and colB == [box] * dw and rowA == [box * j for j in range(dh)] and rowB == [box] * dh
                seen[box] += 1
            else:
                seen["none"] += 1
                if not wide and mode == 3 and dw and sw in (4 * dw, 8 * dw) and sh * dw == sw * dh:
                    # an exact 4x / 8x reduction that reached the box filter must have been recognised
                    assert colB != [sw // dw] * dw or rowB != [sh // dh] * dh or colA != [sw // dw * i for i in range(dw)], (sw, sh, dw, dh)
    assert seen["doubling"] > 500 and seen[4] > 100 and seen[8] > 100 and seen["none"] > 2000, seen


@pytest.fixture(scope="module")
def geometry():
    """tests/tools/geometry_check.cpp over libavif_amd/csrc/tile_geom.h, host code only (hipcc --cuda-host-only: no GPU involved)."""
    src, so = ROOT / "tests" / "tools" / "geometry_check.cpp", ROOT / "tests" / "tools" / "libgeometrycheck.so"
    deps = [src, CSRC / "tile_geom.h", CSRC / "tile_shared.h", CSRC / "plan.h"]
    if not so.exists() or any(d.stat().st_mtime > so.stat().st_mtime for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{CSRC}", f"-I{ROOT / 'include'}",
                        "-o", os.fspath(so), os.fspath(src)], check=True, capture_output=True)
    lib = C.CDLL(os.fspath(so))
    lib.geomCheckPk.restype, lib.geomCheckPk.argtypes = C.c_int, [C.c_uint32] * 6
    lib.geomCheckPkShifted.restype, lib.geomCheckPkShifted.argtypes = C.c_int, [C.c_uint32] * 4
    lib.geomSweepPk.restype, lib.geomSweepPk.argtypes = C.c_uint64, [C.c_uint32, C.POINTER(C.c_uint32 * 7), C.POINTER(C.c_uint64)]
    lib.geomSweepRemap.restype, lib.geomSweepRemap.argtypes = C.c_uint64, [C.c_uint32]
    return lib


def test_every_wave_tile_of_a_launch_is_visited_exactly_once(geometry):
    """The wave-private kernels' launch geometry (pkGeometry) and index arithmetic (pkTileOf / pkPlaceOf: XCD chunks, multiply-high
    divisions), walked on the CPU: every band x strip run of a job has exactly one owner, for every tuning the launchers can form."""
    first, cases = (C.c_uint32 * 7)(), C.c_uint64()
    bad = geometry.geomSweepPk(400, C.byref(first), C.byref(cases))
    assert bad == 0 and cases.value > 1_000_000, (bad, list(first))
    for w, h in [(1920, 1080), (3840, 2160), (7680, 4320), (15360, 8640), (16384, 16384), (16380, 16382), (4, 2), (260, 16384), (16384, 2)]:
        for count in (1, 64):
            for strips in (0, 2, 4):
                for wxl in (0, 1, 2):
                    for chunk_rows in (0, 1, 2, 5, 15):
                        assert geometry.geomCheckPk(w & ~3, h & ~1, count, strips, wxl, chunk_rows) == 0, (w, h, count, strips, wxl, chunk_rows)


def test_a_tile_grid_that_starts_above_the_rectangle_still_visits_every_strip_once(geometry):
    """Quarter turns of single images start their tile grid up to three waves' worth of strips above the rectangle (launchSoloMapped: where
    the 128-byte runs of the transposing stores fall along a destination row); the waves up there own nothing, everything else exactly once."""
    for w, h in [(7664, 4312), (7680, 4320), (256, 8), (300, 34), (4032, 3024), (1100, 150), (16384, 16382)]:
        for strips in (2, 4):
            for waves in (0, 1, 2, 3):
                assert geometry.geomCheckPkShifted(w & ~3, h & ~1, strips, waves * strips) == 0, (w, h, strips, waves)


def test_a_grid_job_is_linked_to_the_right_neighbours(geometry):
    """tile_shared.h linkHalo (grids in one launch): whichever way round the job's planes were handed to the kernels, a neighbour's follow;
    a neighbour that does not exist is the job's own tile (its entry is never read: the coordinates stop at the window on that side)."""
    geometry.geomCheckHaloLink.restype, geometry.geomCheckHaloLink.argtypes = C.c_int, [C.c_int] * 5
    for code in range(32):
        bits = [(code >> k) & 1 for k in range(5)]
        assert geometry.geomCheckHaloLink(*bits) == 0, bits
    # (a tile whose U and V planes are one buffer -- the order is distillArgs' record, not a pointer comparison)
    geometry.geomCheckHaloLinkSharedPlanes.restype, geometry.geomCheckHaloLinkSharedPlanes.argtypes = C.c_int, [C.c_int]
    assert geometry.geomCheckHaloLinkSharedPlanes(0) == 0 and geometry.geomCheckHaloLinkSharedPlanes(1) == 0


def test_grid_batches_along_the_canvas_rows_visit_every_tile_once(geometry):
    """tile_geom.h pkBatchWhereOf / pkBatchGrid (round 5): the tiles of one canvas walked along the canvas rows -- grid x = (tile column of the
    canvas, tile of the job's row), y = tile row inside the job, z = tile row of the canvas -- visit every tile of every job exactly once, for
    any tile size, grid shape, strips per wave and wave arrangement; one column (or separate buffers) keeps the job-by-job order."""
    geometry.geomCheckCanvasOrder.restype, geometry.geomCheckCanvasOrder.argtypes = C.c_int, [C.c_uint32] * 6
    for (w, h) in ((1920, 1080), (512, 512), (256, 34), (1028, 66), (64, 2), (4100, 700)):
        for (cols, rows) in ((8, 8), (8, 6), (2, 3), (1, 4), (5, 1), (3, 3)):
            for strips in (0, 2, 4):
                for waves_x in (0, 1, 2):
                    assert geometry.geomCheckCanvasOrder(w & ~3, h & ~1, cols, rows, strips, waves_x) == 0, (w, h, cols, rows, strips, waves_x)


def test_the_cooperative_kernels_block_order_is_a_permutation(geometry):
    assert geometry.geomSweepRemap(20000) == 0


def test_cover_rectangle_of_a_fused_crop(host):
    """plan.h coverOfCrop: the rectangle a fused crop / rotate / mirror converts contains the crop, starts where the tile kernels can start
    (x a multiple of 8, y even), reaches at most one run of pixels above the crop -- and for quarter turns of 4- and 8-byte pixels its first
    row is the one that makes the 128-byte runs of the transposing stores whole cache lines if any candidate does, else the one that keeps
    them furthest from an even 64 + 64 split (the measured ranking)."""
    import random
    rnd = random.Random(5)
    for _ in range(4000):
        W, H = rnd.choice([(7680, 4320), (4032, 3024), (1100, 150), (701, 61)])
        cw, ch = rnd.randint(1, W), rnd.randint(1, H)
        cx, cy = rnd.randint(0, W - cw), rnd.randint(0, H - ch)
        turns, mirror, pb = rnd.choice([0, 1, 2, 3]), rnd.choice([-1, 0, 1]), rnd.choice([3, 4, 6, 8])
        address = rnd.choice([0, 64, 128 * rnd.randint(1, 1000), 16 * rnd.randint(1, 10000)])
        out = (C.c_uint32 * 4)()
        host.hostCoverOfCrop(cx, cy, cw, ch, turns, mirror, address, pb, C.byref(out))
        x, y, w, h = out
        assert x % 8 == 0 and y % 2 == 0 and x <= cx and y <= cy and x + w == cx + cw and y + h == cy + ch and cx - x < 8
        run = 128 // pb if pb in (4, 8) else 0
        if not (turns & 1) or not run:
            assert y == cy & ~1
            continue
        assert cy - y < run + 1
        # the destination x of the crop's rows: x = sx * jj + kx with the map's sx, kx (plan.h makePixelMap)
        dw = ch
        sx, kx = {1: (1, 0), 3: (-1, ch - 1)}[turns]
        if mirror == 1:
            sx, kx = -sx, dw - 1 - kx

        def offset(y0):
            d = y0 - cy
            start = kx + d if sx > 0 else kx - d - (run - 1)
            return (address + start * pb) % 128

        def score(y0):
            off = offset(y0)
            return 1000 if off == 0 else abs(off - 64)

        candidates = [y0 for y0 in range(cy & ~1, -1, -2) if (cy & ~1) - y0 + 2 <= run or y0 == cy & ~1]
        best = max(score(y0) for y0 in candidates)
        assert score(y) == best or (best == 1000 and offset(y) == 0), (cx, cy, cw, ch, turns, mirror, pb, address, y, [(c, offset(c)) for c in candidates[:10]])


def test_rebound_plans_equal_plans_made_from_scratch(host):
    """Tiles 1 .. N-1 of a batch get their plan by rebinding tile 0's (plan.cpp: rebindYuvToRgbPlan), which lists by hand what plan derivation
    reads.  Over random configurations, buffers, pitches and rectangles the rebound plan must equal the plan made from scratch BYTE FOR BYTE
    (padding included: the resident batch table is compared with memcmp), and every difference in a field a plan depends on must be refused."""
    host.hostCheckRebind.restype = C.c_int
    host.hostCheckRebind.argtypes = [C.c_uint32, C.c_int] + [C.POINTER(C.c_int)] * 3
    refused, mutated, rebound = C.c_int(), C.c_int(), C.c_int()
    bad = host.hostCheckRebind(20260922, 20000, C.byref(refused), C.byref(mutated), C.byref(rebound))
    assert bad == 0, bad
    assert rebound.value > 5000 and mutated.value > 1000 and refused.value == mutated.value, (rebound.value, mutated.value, refused.value)


def test_kernels_step_search_equals_the_walks(host):
    """gainmap_steps.h stepIndexFromGuess -- what the histogram and quantiser kernels of the gain-map computation run per sample -- decides a
    guess from four steps around it; the walks it replaced give the same index on 4 million (table, guess, sample) triples, NaN / infinite
    samples and wild guesses included.  (The device code is this very function: kernels_gainmap.hip includes the header.)"""
    assert host.hostCheckStepSearch(20000, 1) == 0 and host.hostCheckStepSearch(2000, 77) == 0
