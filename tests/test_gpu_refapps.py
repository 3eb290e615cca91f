"""The reference's own APPLICATION code over the HIP backend, on the reference's own fixtures.

oracle/Makefile (target `apps`) compiles libavif's apps/shared/*.c -- y4mRead / y4mWrite (apps/shared/y4m.c:256,481), avifPNGWrite /
avifPNGRead (apps/shared/avifpng.c:627,...), avifReadImage (apps/shared/avifutil.c:318), avifApplyTransforms -- and tests/avifyuv.c from
where they lie under /root/reference, twice: against the backend-less reference (oracle/_ref/libavif_ref.so) and against the reference
built over integration/reformat_libyuv_hip.c (oracle/_ref/libavif_hipbackend.so, seam B).  The four codec-free fixtures of the
reference's test data (tests/data/kodim03_yuv420_8bpc.y4m, kodim23..., cosmos1650_yuv444_10bpc_p3pq.y4m, webp_logo_animated.y4m)
travel in oracle/_ref/data.  What avifdec / avifenc do around the reformat path, without a codec:

  * y4m -> PNG at depth 8 and 16, every frame: the hip-backed build's files are byte-identical (same libpng, so: same pixel rows) to the
    backend-less build's with AVIFHIP_ARITHMETIC=float, and to a STOCK build's -- the same application code over Pillow's libavif 1.4.1 +
    libyuv binary, `refapp_yuvlib` -- with the library's default arithmetic;
  * PNG -> avifReadImage -> y4m: the Y4M files likewise;
  * the fixtures' frames through the C ABI against both oracles, every RGB layout avifdec can ask for;
  * tests/avifyuv.c -m limited / rgb / premultiply / (bounded) drift print the same text on the builds;
and in every case the hip-backed build really launched kernels (a preloaded shim prints avifhipLaunchCount() at exit).
"""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import harness as H
import oracle_lib
from libavif_amd import abi

pytestmark = pytest.mark.gpu

REF_DIR = oracle_lib.ORACLE_DIR / "_ref"
DATA = REF_DIR / "data"
FIXTURES = ["kodim03_yuv420_8bpc", "kodim23_yuv420_8bpc", "cosmos1650_yuv444_10bpc_p3pq", "webp_logo_animated"]
NEEDED = ["refapp_ref", "refapp_hip", "refapp_yuvlib", "avifyuv_ref", "avifyuv_hip", "avifyuv_yuvlib", "liblaunchcount.so", "libavif_hipbackend.so",
          "libavif_ref.so"]


@pytest.fixture(scope="module")
def apps(hip):
    missing = [n for n in NEEDED if not (REF_DIR / n).exists()] + [f for f in FIXTURES if not (DATA / f"{f}.y4m").exists()]
    if missing:
        # GPU tests: the prebuilt programs and the fixtures travel to the GPU box with the snapshot; a run without them must not report green
        pytest.fail(f"oracle/_ref lacks {missing}: build them where /root/reference exists (make -C oracle apps) and ship them")
    return REF_DIR


def _run(exe: Path, args, arithmetic="float", timeout=300, hip=False):
    """Runs one of the reference programs; returns (stdout, launches) -- launches is None for the backend-less build."""
    env = dict(os.environ)
    env["AVIFHIP_ARITHMETIC"] = arithmetic
    env["AVIFHIP_MIN_PIXELS"] = "0"  # the small fixtures must take the GPU route too
    if hip:
        env["LD_PRELOAD"] = os.fspath(REF_DIR / "liblaunchcount.so")
    proc = subprocess.run([os.fspath(exe)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=timeout)
    assert proc.returncode == 0, f"{exe.name} {args}: rc={proc.returncode}\n{proc.stdout[-2000:]}\n{proc.stderr[-2000:]}"
    launches = None
    if hip:
        lines = [ln for ln in proc.stderr.splitlines() if ln.startswith("avifhip launches=")]
        assert lines, proc.stderr[-2000:]
        launches = int(lines[-1].split("=")[1])
    return proc.stdout, launches


# ---------------------------------------------------------------------------------------------------
# Y4M, read back independently of the reference


def read_y4m(path: Path):
    """[(header dict, [planes])] per frame; planes as uint8 / uint16 arrays."""
    blob = path.read_bytes()
    eol = blob.index(b"\n")
    tags = blob[:eol].split(b" ")
    assert tags[0] == b"YUV4MPEG2"
    hdr = {"W": 0, "H": 0, "C": "420jpeg", "range": "LIMITED"}
    for t in tags[1:]:
        if t[:1] == b"W":
            hdr["W"] = int(t[1:])
        elif t[:1] == b"H":
            hdr["H"] = int(t[1:])
        elif t[:1] == b"C":
            hdr["C"] = t[1:].decode()
        elif t.startswith(b"XCOLORRANGE="):
            hdr["range"] = t.split(b"=")[1].decode()
    cs = hdr["C"]
    depth = 10 if "p10" in cs else 12 if "p12" in cs else 8
    fmt = (abi.AVIF_PIXEL_FORMAT_YUV444 if cs.startswith("444") else abi.AVIF_PIXEL_FORMAT_YUV422 if cs.startswith("422")
           else abi.AVIF_PIXEL_FORMAT_YUV400 if cs.startswith("mono") else abi.AVIF_PIXEL_FORMAT_YUV420)
    hdr["depth"], hdr["format"] = depth, fmt
    w, h = hdr["W"], hdr["H"]
    cw, ch = abi.chroma_dims(w, h, fmt)
    bps = 2 if depth > 8 else 1
    dims = [(w, h)] + ([] if fmt == abi.AVIF_PIXEL_FORMAT_YUV400 else [(cw, ch), (cw, ch)])
    frames, pos = [], eol + 1
    while pos < len(blob):
        assert blob[pos:pos + 5] == b"FRAME", blob[pos:pos + 16]
        pos = blob.index(b"\n", pos) + 1
        planes = []
        for pw, ph in dims:
            n = pw * ph * bps
            planes.append(np.frombuffer(blob, dtype=np.uint16 if bps == 2 else np.uint8, count=pw * ph, offset=pos).reshape(ph, pw).copy())
            pos += n
        frames.append(planes)
    return hdr, frames


# ---------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("png_depth", [8, 16])
@pytest.mark.parametrize("fixture", FIXTURES)
def test_y4m_to_png_over_the_backend(apps, tmp_path, fixture, png_depth):
    src = DATA / f"{fixture}.y4m"
    hdr, planes = read_y4m(src)
    for arithmetic, other in (("float", "refapp_ref"), ("auto", "refapp_yuvlib")):
        out_other, _ = _run(apps / other, ["y4m2png", src, tmp_path / f"{arithmetic}_want", png_depth])
        out_hip, launches = _run(apps / "refapp_hip", ["y4m2png", src, tmp_path / f"{arithmetic}_hip", png_depth], arithmetic=arithmetic, hip=True)
        assert out_other.splitlines()[-1] == out_hip.splitlines()[-1] == f"frames={len(planes)}"
        assert launches >= len(planes), (fixture, arithmetic, launches)
        for k in range(len(planes)):
            a, b = (tmp_path / f"{arithmetic}_want_{k}.png").read_bytes(), (tmp_path / f"{arithmetic}_hip_{k}.png").read_bytes()
            assert a == b, f"{fixture} frame {k}, {arithmetic} arithmetic: the PNG differs from the one {other} writes"


@pytest.mark.parametrize("seam", ["B", "A"])
def test_device_set_from_the_environment_under_the_reference_apps(apps, tmp_path, seam):
    """Round 5: AVIFHIP_DEVICES in the environment of an UNMODIFIED application makes the library share every host-resident conversion out over
    the device set (libavif_amd/csrc/api_farm.cpp).  The reference's own y4m -> PNG and PNG -> y4m programs over seam B (the hip-backed
    build) and seam A (a stock build under the LD_PRELOAD interposer), three workers on device 0, shares from 64 x 32 pixels up so that the
    768 x 512 fixtures are farmed: the files are byte-identical to the ones written without the variable, and the trace shows the shares."""
    src = DATA / "kodim03_yuv420_8bpc.y4m"
    exe = apps / ("refapp_hip" if seam == "B" else "refapp_ref")
    extra = {"AVIFHIP_DEVICES": "0,0,0", "AVIFHIP_FARM_MIN_PIXELS": "2048", "AVIFHIP_FARM_TRACE": "1"}

    def run(tag, farmed, args):
        env = dict(os.environ, AVIFHIP_MIN_PIXELS="0")
        if seam == "A":
            preload = oracle_lib.ORACLE_DIR.parent / "libavif_amd" / "csrc" / "libavifhip_preload.so"
            assert preload.exists(), "libavifhip_preload.so is missing"
            env["LD_PRELOAD"] = os.fspath(preload)
            env.pop("AVIFHIP_ARITHMETIC", None)
        else:
            env["AVIFHIP_ARITHMETIC"] = "auto"
        for name in extra:
            env.pop(name, None)
        if farmed:
            env.update(extra)
        proc = subprocess.run([os.fspath(exe)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
        assert proc.returncode == 0, (tag, proc.stdout[-1000:], proc.stderr[-1000:])
        return proc.stderr

    for depth in (8, 16):
        plain = run("plain", False, ["y4m2png", src, tmp_path / f"plain{depth}", depth])
        farmed = run("farmed", True, ["y4m2png", src, tmp_path / f"farmed{depth}", depth])
        assert "avifhip farm:" not in plain and "avifhip farm: 3 shares" in farmed, farmed[-500:]
        assert (tmp_path / f"plain{depth}_0.png").read_bytes() == (tmp_path / f"farmed{depth}_0.png").read_bytes()
    # ... and the encode direction: PNG -> avifReadImage (avifImageRGBToYUV) -> y4m
    png = tmp_path / "plain8_0.png"
    a = run("plain", False, ["png2y4m", png, tmp_path / "back_plain.y4m", "420", 8, 1, "limited"])
    b = run("farmed", True, ["png2y4m", png, tmp_path / "back_farmed.y4m", "420", 8, 1, "limited"])
    assert "avifhip farm:" not in a and "avifhip farm: 3 shares" in b, b[-500:]
    assert (tmp_path / "back_plain.y4m").read_bytes() == (tmp_path / "back_farmed.y4m").read_bytes()


@pytest.mark.parametrize("yuv", [("420", 8, 1, "limited"), ("444", 8, 6, "full"), ("422", 10, 9, "limited"), ("420", 8, 6, "limited")])
@pytest.mark.parametrize("fixture", ["kodim03_yuv420_8bpc", "cosmos1650_yuv444_10bpc_p3pq"])
def test_png_to_y4m_over_the_backend(apps, tmp_path, fixture, yuv):
    fmt, depth, mc, rng = yuv
    png_depth = 16 if "10bpc" in fixture else 8
    _run(apps / "refapp_ref", ["y4m2png", DATA / f"{fixture}.y4m", tmp_path / "src", png_depth])
    png = tmp_path / "src_0.png"
    for arithmetic, other in (("float", "refapp_ref"), ("auto", "refapp_yuvlib")):
        out_other, _ = _run(apps / other, ["png2y4m", png, tmp_path / f"{arithmetic}_want.y4m", fmt, depth, mc, rng])
        out_hip, launches = _run(apps / "refapp_hip", ["png2y4m", png, tmp_path / f"{arithmetic}_hip.y4m", fmt, depth, mc, rng], arithmetic=arithmetic, hip=True)
        assert out_other.splitlines()[-1] == out_hip.splitlines()[-1] and launches > 0  # (the line before names the output file)
        assert (tmp_path / f"{arithmetic}_want.y4m").read_bytes() == (tmp_path / f"{arithmetic}_hip.y4m").read_bytes(), \
            f"{arithmetic} arithmetic: the Y4M differs from the one {other} writes"
    # (the written file parses back to planes of the requested shape)
    hdr, frames = read_y4m(tmp_path / "auto_hip.y4m")
    assert hdr["depth"] == depth and len(frames) == 1 and frames[0][0].shape == (hdr["H"], hdr["W"])


@pytest.mark.parametrize("mode", ["limited", "rgb", "premultiply"])
def test_avifyuv_prints_the_same_on_the_builds(apps, mode):
    for arithmetic, other in (("float", "avifyuv_ref"), ("auto", "avifyuv_yuvlib")):
        out_other, _ = _run(apps / other, ["-m", mode])
        out_hip, launches = _run(apps / "avifyuv_hip", ["-m", mode], arithmetic=arithmetic, hip=True)
        # (first line: "avif version: ..." -- the stock binary is 1.4.1, the reference tree 1.4.2)
        assert out_other.splitlines()[1:] == out_hip.splitlines()[1:], f"-m {mode}, {arithmetic} arithmetic, against {other}"
        assert len(out_other.splitlines()) >= 5
        if mode != "limited":  # (-m limited only calls the scalar range helpers)
            assert launches > 0


# ---------------------------------------------------------------------------------------------------
# seam A: the UNMODIFIED reference programs (linked against the shared, backend-less libavif_ref.so) with the interposer preloaded


def _run_preloaded(exe: Path, args, timeout=300):
    """LD_PRELOAD = the interposer + the launch-count shim; returns (stdout, launches)."""
    preload = oracle_lib.ORACLE_DIR.parent / "libavif_amd" / "csrc" / "libavifhip_preload.so"
    assert preload.exists(), "libavifhip_preload.so is missing"
    env = dict(os.environ, AVIFHIP_MIN_PIXELS="0", LD_PRELOAD=f"{preload}:{REF_DIR / 'liblaunchcount.so'}")
    env.pop("AVIFHIP_ARITHMETIC", None)  # the interposer pins what the interposed libavif computes itself (no libyuv there: fp32)
    proc = subprocess.run([os.fspath(exe)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=timeout)
    assert proc.returncode == 0, f"{exe.name} {args} under the interposer: rc={proc.returncode}\n{proc.stdout[-2000:]}\n{proc.stderr[-2000:]}"
    lines = [ln for ln in proc.stderr.splitlines() if ln.startswith("avifhip launches=")]
    assert lines, proc.stderr[-2000:]
    return proc.stdout, int(lines[-1].split("=")[1])


@pytest.mark.parametrize("fixture", FIXTURES)
def test_seam_a_y4m_to_png_under_the_interposer(apps, tmp_path, fixture):
    """`LD_PRELOAD=libavifhip_preload.so refapp_ref y4m2png`: avifPNGWrite's avifImageYUVToRGB (apps/shared/avifpng.c:688) lands in the HIP
    kernels, the files equal the ones the same program writes on its own."""
    src = DATA / f"{fixture}.y4m"
    _, planes = read_y4m(src)
    for depth in (8, 16):
        out_plain, _ = _run(apps / "refapp_ref", ["y4m2png", src, tmp_path / f"plain{depth}", depth])
        out_hip, launches = _run_preloaded(apps / "refapp_ref", ["y4m2png", src, tmp_path / f"hip{depth}", depth])
        assert out_plain.splitlines()[-1] == out_hip.splitlines()[-1] == f"frames={len(planes)}"
        assert launches >= len(planes), (fixture, depth, launches)
        for k in range(len(planes)):
            assert (tmp_path / f"plain{depth}_{k}.png").read_bytes() == (tmp_path / f"hip{depth}_{k}.png").read_bytes(), (fixture, depth, k)


@pytest.mark.parametrize("yuv", [("420", 8, 1, "limited"), ("444", 8, 6, "full"), ("422", 10, 9, "limited")])
def test_seam_a_png_to_y4m_under_the_interposer(apps, tmp_path, yuv):
    """... and avifReadImage's avifImageRGBToYUV (apps/shared/avifpng.c:552) likewise."""
    fmt, depth, mc, rng = yuv
    _run(apps / "refapp_ref", ["y4m2png", DATA / "kodim03_yuv420_8bpc.y4m", tmp_path / "src", 8])
    png = tmp_path / "src_0.png"
    out_plain, _ = _run(apps / "refapp_ref", ["png2y4m", png, tmp_path / "plain.y4m", fmt, depth, mc, rng])
    out_hip, launches = _run_preloaded(apps / "refapp_ref", ["png2y4m", png, tmp_path / "hip.y4m", fmt, depth, mc, rng])
    assert out_plain.splitlines()[-1] == out_hip.splitlines()[-1] and launches > 0
    assert (tmp_path / "plain.y4m").read_bytes() == (tmp_path / "hip.y4m").read_bytes()


@pytest.mark.parametrize("mode", ["rgb", "premultiply"])
def test_seam_a_avifyuv_under_the_interposer(apps, mode):
    out_plain, _ = _run(apps / "avifyuv_ref", ["-m", mode])
    out_hip, launches = _run_preloaded(apps / "avifyuv_ref", ["-m", mode])
    assert out_plain.splitlines() == out_hip.splitlines() and len(out_plain.splitlines()) >= 5
    assert launches > 0


def test_avifyuv_drift_bounded(apps):
    """-m drift walks the whole RGB cube of every depth (hours on a CPU): both builds run for a bounded time with line-buffered output and
    must agree on every line both of them finished -- at least the 36 combinations of 8-bit RGB (tests/avifyuv.c:118-152)."""
    outs = []
    for exe in ("avifyuv_ref", "avifyuv_hip"):
        env = dict(os.environ, AVIFHIP_ARITHMETIC="float", AVIFHIP_MIN_PIXELS="0")
        proc = subprocess.run(["timeout", "-s", "INT", "30", "stdbuf", "-oL", os.fspath(apps / exe), "-m", "drift"], capture_output=True, text=True, env=env)
        lines = proc.stdout.splitlines()
        outs.append(lines[:-1] if lines else lines)  # the last line may be cut
    n = min(len(outs[0]), len(outs[1]))
    assert n >= 1 + 36, (len(outs[0]), len(outs[1]))
    assert outs[0][:n] == outs[1][:n]


def _image_from_frame(hdr, planes):
    """The avifImage y4mRead hands on: the defaults of avifImageCreateEmpty (src/avif.c:134-141) + the header's fields."""
    rng = abi.AVIF_RANGE_FULL if hdr["range"] == "FULL" else abi.AVIF_RANGE_LIMITED
    img = abi.make_yuv(hdr["W"], hdr["H"], hdr["depth"], hdr["format"], rng, abi.AVIF_MATRIX_COEFFICIENTS_UNSPECIFIED, color_primaries=2)
    bps = 2 if hdr["depth"] > 8 else 1
    for dst, src in zip(img.planes, planes):
        dst.view(np.uint16 if bps == 2 else np.uint8)[:, :src.shape[1]] = src
    return img


@pytest.mark.parametrize("fixture", FIXTURES)
def test_reference_fixtures_through_the_c_abi(apps, hip, fixture):
    """Real pictures instead of noise: every frame of the reference's fixtures through avifhipImageYUVToRGB (host buffers) in both
    arithmetics, against the pinned oracles, for the RGB layouts / depths / upsamplings avifdec's writers ask for."""
    hdr, frames = read_y4m(DATA / f"{fixture}.y4m")
    be = H.hip_host_backend()
    layouts = [(abi.AVIF_RGB_FORMAT_RGB, 8), (abi.AVIF_RGB_FORMAT_RGBA, 8), (abi.AVIF_RGB_FORMAT_BGRA, 8), (abi.AVIF_RGB_FORMAT_RGB, 16), (abi.AVIF_RGB_FORMAT_RGBA, hdr["depth"])]
    try:
        for k, planes in enumerate(frames[:4]):
            img = _image_from_frame(hdr, planes)
            for fmt, depth in layouts:
                for up in (abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC, abi.AVIF_CHROMA_UPSAMPLING_NEAREST):
                    for arithmetic, oracle, avoid in ((1, H.oracle_backend(), True), (0, H.oracle_libyuv_backend(), False)):
                        hip.avifhipSetArithmetic(arithmetic)
                        want = abi.make_rgb(hdr["W"], hdr["H"], depth, fmt, upsampling=up, avoid_libyuv=avoid, fill=H.FILL_BYTE)
                        got = abi.make_rgb(hdr["W"], hdr["H"], depth, fmt, upsampling=up, avoid_libyuv=avoid, fill=H.FILL_BYTE)
                        assert oracle.yuv_to_rgb(img.struct, want.struct) == 0
                        assert be.yuv_to_rgb(img.struct, got.struct) == 0
                        assert np.array_equal(want.pixels, got.pixels), f"{fixture} frame {k} fmt {fmt} depth {depth} up {up} {oracle.name}: " + H.describe_diff(want.pixels, got.pixels)
    finally:
        hip.avifhipSetArithmetic(1)  # what the `hip` fixture pins
