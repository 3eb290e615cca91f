"""The BASELINE configurations that stream (working set beyond the 256 MB Infinity Cache) under the result-preserving tuning knobs, with the
byte-movement ceiling of each shape beside them -- ONE process, one box, so that the rows compare (boxes differ by +-6 %).
    python tests/tools/stream_sweep.py [cfg3 cfg4 cfg5x64 cfg5grid cfg2cold cfg2cold_fp32 ...]
Each row: configuration, knob, microseconds per launch (median of 7 event-timed bursts of 30 after 30 ms of the same launches), fraction of
8 TB/s on the algorithmic bytes.  Knobs: plan.h TuningBits through avifhipSetTuning (bit 0 per-XCD bands, bits 8-11 strips
per wave, bits 16-17 1 + log2(waves side by side), bits 20-23 tile rows per XCD chunk), AVIFHIP_STREAM_LOADS, AVIFHIP_R2Y_SPW."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

# (tests/tools/lib_ab.py: the same measurement over another build of the library, interleaved on one box)
if os.environ.get("AVIFHIP_TOOLS_LIBRARY"):
    from pathlib import Path as _Path

    native.LIB_PATH = _Path(os.environ["AVIFHIP_TOOLS_LIBRARY"])

if os.environ.get("AVIFHIP_BENCH_LIB"):
    native.LIB_PATH = Path(os.environ["AVIFHIP_BENCH_LIB"]).resolve()
lib = native.load()
BIL = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR
PEAK = 8000.0
SEQ = 4  # frames per launch of the sequence rows (bench.py SEQUENCE_FRAMES)


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def burst(fn, *a, iters=30):
    spent = 0.0
    while spent < 30.0:
        ms = fn(*a, 0, 50, None)
        if ms < 0:
            raise SystemExit("timing call failed: " + lib.avifhipLastError().decode())
        spent += max(ms, 1e-3) * 50
    return median([fn(*a, 2, iters, None) for _ in range(7)])


def emit(cfg, knob, ms, alg_bytes, **extra):
    gb = alg_bytes / (ms * 1e-3) / 1e9
    print(json.dumps({"config": cfg, "knob": knob, "us": round(ms * 1e3, 2), "frac": round(gb / PEAK, 4), "kernel": native.last_kernel(), **extra}), flush=True)


def arr(pairs):
    n = len(pairs)
    return n, (C.POINTER(abi.avifImage) * n)(*[C.pointer(q[0].struct) for q in pairs]), (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(q[1].struct) for q in pairs])


def y2r(w, h, depth, fmt, rng, mc, rgb_depth, alpha=False, premult=False, avoid=False, seed=1):
    img = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
    synth.fill_yuv(img, 0x12345678 + seed)
    rgb = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, alpha_premultiplied=premult, avoid_libyuv=avoid, allocate=False)
    return device.DeviceYUV(img), device.DeviceRGB(rgb)


TUNINGS = [("default", 0x1), ("raster", 0x0), ("bands,2 strips", 0x201), ("bands,4 strips (rounds 2-4 for big frames)", 0x401), ("raster,2 strips", 0x200), ("raster,4 strips", 0x400),
           ("raster,4 waves wide,2 strips", 0x30200), ("raster,4 waves wide,4 strips", 0x30400), ("raster,2 waves wide,2 strips", 0x20200),
           ("bands,4 waves wide,2 strips", 0x30201), ("bands,2 rows/chunk", 0x200001), ("bands,4 rows/chunk", 0x400001),
           ("default + streaming loads (single 16-bit unfiltered)", 0x41), ("raster + streaming loads", 0x40), ("raster,4 waves wide,2 strips + streaming loads", 0x30240),
           ("plain stores off: nt bit", 0x3)]


def sweep_y2r(cfg, pairs, alg_bytes, tunings=TUNINGS, stream_env=True):
    n, imgs, rgbs = arr(pairs)
    ms = burst(lib.avifhipTimeStreamCeiling, n, imgs, rgbs)
    emit(cfg, "ceiling", ms, alg_bytes)
    for name, bits in tunings:
        lib.avifhipSetTuning(bits)
        emit(cfg, name, burst(lib.avifhipTimeYUVToRGBCycle, n, imgs, rgbs), alg_bytes, tuning=hex(bits))
    lib.avifhipSetTuning(1)
    if stream_env:
        for v in ("1", "0"):
            os.environ["AVIFHIP_STREAM_LOADS"] = v
            emit(cfg, "AVIFHIP_STREAM_LOADS=" + v, burst(lib.avifhipTimeYUVToRGBCycle, n, imgs, rgbs), alg_bytes)
        del os.environ["AVIFHIP_STREAM_LOADS"]


def run(name):
    lib.avifhipSetArithmetic(0)
    lib.avifhipSetTuning(1)
    if name == "cfg3":
        pairs = [y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, premult=True, seed=k) for k in range(2)]
        sweep_y2r("cfg3 (2 frames cycled)", pairs, 16.0 * 7680 * 4320)
    elif name in ("cfg2cold", "cfg2cold_fp32"):
        pairs = [y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=name.endswith("fp32"), seed=k) for k in range(12)]
        sweep_y2r(name + " (12 frames cycled)", pairs, 5.5 * 7680 * 4320)
    elif name in ("cfg2warm", "cfg2warm_fp32"):
        pairs = [y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=name.endswith("fp32"), seed=k) for k in range(4)]
        sweep_y2r(name + " (4 frames cycled)", pairs, 5.5 * 7680 * 4320, stream_env=False)
    elif name == "cfg3same":
        pairs = [y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, premult=True, seed=0)]
        sweep_y2r("cfg3 (same frame)", pairs, 16.0 * 7680 * 4320, tunings=[("default", 0x1), ("bands,4 strips (rounds 2-4)", 0x401), ("raster,4 waves wide,2 strips + streaming loads", 0x30240)], stream_env=False)
    elif name == "f16_444a":
        pairs = []
        for k in range(2):
            img = abi.make_yuv(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_LIMITED, 9, with_alpha=True)
            synth.fill_yuv(img, 0x4242 + k)
            rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=True, allocate=False)
            rgb.struct.isFloat = 1
            pairs.append((device.DeviceYUV(img), device.DeviceRGB(rgb)))
        sweep_y2r("f16_444a (2 frames cycled)", pairs, 16.0 * 7680 * 4320, tunings=[("default", 0x1), ("bands,4 strips (rounds 2-4)", 0x401), ("raster,4 waves wide,2 strips + streaming loads", 0x30240)], stream_env=False)
    elif name == "batch1080":
        # sequences of small frames: N 1080p frames (8-bit 4:2:0 -> RGBA8, API defaults) per launch through avifhipImageYUVToRGBBatchAsync, against
        # one launch per frame -- where the batch starts to pay (a 1080p frame alone is launch plus one load -> stage -> compute -> store chain)
        pairs = [y2r(1920, 1080, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, seed=k) for k in range(16)]
        n, imgs, rgbs = arr(pairs)
        alg = 5.5 * 1920 * 1080
        emit("1080p, one frame per launch (16 frames cycled)", "avifhipImageYUVToRGBAsync", burst(lib.avifhipTimeYUVToRGBCycle, n, imgs, rgbs), alg)
        emit("1080p, one frame per launch (16 frames cycled)", "ceiling", burst(lib.avifhipTimeStreamCeiling, n, imgs, rgbs), alg)
        for k in (2, 4, 8, 16):
            ms = burst(lib.avifhipTimeYUVToRGBBatch, k, imgs, rgbs, None)
            emit(f"1080p, {k} frames per launch", "avifhipImageYUVToRGBBatchAsync, per frame", ms / k, alg, us_per_launch=round(ms * 1e3, 2))
            ms = burst(lib.avifhipTimeStreamCeilingBatch, k, imgs, rgbs)
            emit(f"1080p, {k} frames per launch", "ceiling, per frame", ms / k, alg, us_per_launch=round(ms * 1e3, 2))
    elif name == "cfg2_4k":
        pairs = [y2r(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, seed=k) for k in range(4)]
        sweep_y2r("cfg2_4k (4 frames cycled)", pairs, 5.5 * 3840 * 2160)
    elif name == "cfg4":
        enc = []
        for k in range(8):
            rgb = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
            synth.fill_rgb(rgb, 0x12345678 + k % 2, opaque=True)
            img = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
            enc.append((device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)))
        n, imgs, rgbs = arr(enc)
        alg = 6.5 * 3840 * 2160
        emit("cfg4 (8 frames cycled)", "ceiling", burst(lib.avifhipTimeStreamCeilingRGBToYUV, n, imgs, rgbs), alg)
        emit("cfg4 (8 frames cycled)", "default", burst(lib.avifhipTimeRGBToYUVCycle, n, imgs, rgbs), alg)
        lib.avifhipSetTuning(0x81)
        emit("cfg4 (8 frames cycled)", "raster order (TUNE_R2Y_RASTER)", burst(lib.avifhipTimeRGBToYUVCycle, n, imgs, rgbs), alg)
        emit("cfg4 (same frame)", "raster order (TUNE_R2Y_RASTER)", burst(lib.avifhipTimeRGBToYUV, enc[0][0].struct, enc[0][1].struct), alg)
        lib.avifhipSetTuning(1)
        for spw in ("1", "2", "4"):
            os.environ["AVIFHIP_R2Y_SPW"] = spw
            emit("cfg4 (8 frames cycled)", "AVIFHIP_R2Y_SPW=" + spw, burst(lib.avifhipTimeRGBToYUVCycle, n, imgs, rgbs), alg)
        del os.environ["AVIFHIP_R2Y_SPW"]
        emit("cfg4 (same frame)", "ceiling", burst(lib.avifhipTimeStreamCeilingRGBToYUV, 1, imgs, rgbs), alg)
        emit("cfg4 (same frame)", "default", burst(lib.avifhipTimeRGBToYUV, enc[0][0].struct, enc[0][1].struct), alg)
    elif name in ("cfg5x64", "cfg5grid", "cfg5grid8"):
        depth = 8 if name.endswith("8") else 10
        tiles = []
        for t in range(64):
            img = abi.make_yuv(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            synth.fill_yuv(img, 0x12345678 + t)
            tiles.append(device.DeviceYUV(img))
        timgs = (C.POINTER(abi.avifImage) * 64)(*[C.pointer(t.struct) for t in tiles])
        px = 64 * 1920 * 1080
        alg = (3.0 + (8.0 if depth == 10 else 4.0)) * px
        if name == "cfg5x64":
            outs = [device.DeviceRGB(abi.make_rgb(1920, 1080, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False)) for _ in range(64)]
            rgbs = (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(o.struct) for o in outs])
            emit(name, "ceiling", burst(lib.avifhipTimeStreamCeilingBatch, 64, timgs, rgbs), alg)
            for tname, bits in TUNINGS:
                lib.avifhipSetTuning(bits)
                emit(name, tname, burst(lib.avifhipTimeYUVToRGBBatch, 64, timgs, rgbs, None), alg, tuning=hex(bits))
            lib.avifhipSetTuning(1)
        else:
            canvas = device.DeviceRGB(abi.make_rgb(15360, 8640, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False))
            pb = 8 if depth == 10 else 4
            views = []
            for t in range(64):
                v = abi.make_rgb(1920, 1080, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False)
                v.struct.pixels = canvas.buffer.ptr + (t // 8) * 1080 * canvas.struct.rowBytes + (t % 8) * 1920 * pb
                v.struct.rowBytes = canvas.struct.rowBytes
                views.append(v)
            rgbs = (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(v.struct) for v in views])
            emit(name, "ceiling", burst(lib.avifhipTimeStreamCeilingBatch, 64, timgs, rgbs), alg)
            grid = native.avifhipGrid(8, 8, 15360, 8640)
            for tname, bits in TUNINGS[:6] + [("job by job (TUNE_JOB_MAJOR: rounds 1-4)", 0x1000001), ("default again", 0x1), ("job by job again", 0x1000001)]:
                lib.avifhipSetTuning(bits)
                emit(name, tname, burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas.struct), alg, tuning=hex(bits))
            lib.avifhipSetTuning(1)
            for v in ("1", "0"):
                os.environ["AVIFHIP_GRID_SEAM_PASS"] = v
                emit(name, "AVIFHIP_GRID_SEAM_PASS=" + v, burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas.struct), alg)
            del os.environ["AVIFHIP_GRID_SEAM_PASS"]
    else:
        raise SystemExit("unknown configuration " + name)


def ab(name, tunings, rounds=5):
    """Interleaved A/B of tuning words on one configuration: `rounds` passes over the list, every entry preheated and timed each pass; the
    median per entry -- run-order effects (clock state after an ALU-heavy kernel: +-4 %) average out instead of favouring the last entry."""
    lib.avifhipSetArithmetic(0)
    if name in ("cfg5grid", "cfg5grid8", "photo_grid"):
        return ab_grid(name, tunings, rounds)
    fp32 = name.endswith("_fp32")
    base = name[:-5] if fp32 else name
    frames = {"cfg2cold": 12, "cfg2warm": 4, "cfg2_4k": 4, "cfg2_1080p": 4, "cfg2_4k_cold": 24}[base]
    w, h = {"cfg2cold": (7680, 4320), "cfg2warm": (7680, 4320), "cfg2_4k": (3840, 2160), "cfg2_4k_cold": (3840, 2160), "cfg2_1080p": (1920, 1080)}[base]
    pairs = [y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=fp32, seed=k) for k in range(frames)]
    n, imgs, rgbs = arr(pairs)
    got = {t: [] for t in tunings}
    for _ in range(rounds):
        for t in tunings:
            lib.avifhipSetTuning(t)
            got[t].append(burst(lib.avifhipTimeYUVToRGBCycle, n, imgs, rgbs))
    lib.avifhipSetTuning(1)
    for t in tunings:
        emit(f"{name} ({frames} frames cycled), interleaved x{rounds}", hex(t), median(got[t]), 5.5 * w * h, all_us=[round(x * 1e3, 2) for x in got[t]])


def ab_grid(name, tunings, rounds):
    """... for the grid entry point: cfg5's 64 tiles of 1080p 10-bit into one RGBA(10) / RGBA8 canvas, or the 12-megapixel photograph of 48 tiles"""
    if name == "photo_grid":
        tw, th, cols, rows_, cw, ch, ydepth, depth = 512, 512, 8, 6, 4032, 3024, 8, 8
    else:
        tw, th, cols, rows_, cw, ch, ydepth, depth = 1920, 1080, 8, 8, 15360, 8640, 10, (8 if name.endswith("8") else 10)
    tiles = []
    for t in range(cols * rows_):
        img = abi.make_yuv(tw, th, ydepth, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        synth.fill_yuv(img, 0x12345678 + t)
        tiles.append(device.DeviceYUV(img))
    timgs = (C.POINTER(abi.avifImage) * len(tiles))(*[C.pointer(t.struct) for t in tiles])
    canvas = device.DeviceRGB(abi.make_rgb(cw, ch, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False))
    grid = native.avifhipGrid(rows_, cols, cw, ch)
    alg = ((3.0 if ydepth > 8 else 1.5) + (8.0 if depth > 8 else 4.0)) * cw * ch
    # (tuning words at or above 2^32 carry AVIFHIP_GRID_SEAM_PASS in bits 32-33: 1 = "0" (link: tiles and seams in one launch), 2 = "1" (seam pass))
    got, kernels = {t: [] for t in tunings}, {}
    for _ in range(rounds):
        for t in tunings:
            lib.avifhipSetTuning(t & 0xffffffff)
            seam = {1: "0", 2: "1"}.get(t >> 32)
            if seam is not None:
                os.environ["AVIFHIP_GRID_SEAM_PASS"] = seam
            got[t].append(burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas.struct))
            kernels[t] = native.last_kernel()
            os.environ.pop("AVIFHIP_GRID_SEAM_PASS", None)
    lib.avifhipSetTuning(1)
    for t in tunings:
        knob = hex(t & 0xffffffff) + {0: "", 1: " one launch (AVIFHIP_GRID_SEAM_PASS=0)", 2: " seam pass (AVIFHIP_GRID_SEAM_PASS=1)"}[t >> 32]
        print(json.dumps({"config": f"{name}, interleaved x{rounds}", "knob": knob, "us": round(median(got[t]) * 1e3, 2), "frac": round(alg / (median(got[t]) * 1e-3) / 1e9 / PEAK, 4),
                          "kernel": kernels[t], "all_us": [round(x * 1e3, 2) for x in got[t]]}), flush=True)


def run_only(name, launches=400):
    """`run <cfg3|cfg4>`: nothing but `launches` launches of the configuration's default kernel over its cycled frames -- what the counter passes
    of tests/tools/traffic_cfgs.sh profile (FETCH_SIZE / WRITE_SIZE per dispatch against the algorithmic bytes)."""
    lib.avifhipSetArithmetic(0)
    lib.avifhipSetTuning(1)
    if name == "cfg3":
        pairs = [y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, premult=True, seed=k) for k in range(2)]
        n, imgs, rgbs = arr(pairs)
        ms = lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 20, launches // 4, None)
        emit("cfg3 (2 frames cycled)", "run", ms, 16.0 * 7680 * 4320, algorithmic_read_bytes=8 * 7680 * 4320, algorithmic_write_bytes=8 * 7680 * 4320)
    elif name in ("cfg4", "cfg4seq"):
        enc = []
        for k in range(8):
            rgb = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
            synth.fill_rgb(rgb, 0x12345678 + k % 2, opaque=True)
            img = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
            enc.append((device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)))
        n, imgs, rgbs = arr(enc)
        if name == "cfg4seq":  # the encode direction as a sequence: SEQ frames per launch (avifhipImageRGBToYUVBatchAsync)
            lib.avifhipTimeStreamCeilingRGBToYUV(n, imgs, rgbs, 0, 6000, None)  # ~60 ms of another kernel first (clocks)
            launches = 1200
            ms = lib.avifhipTimeRGBToYUVBatchCycle(n, imgs, rgbs, SEQ, 20, launches // SEQ, None)
            emit(f"cfg4 (8 frames cycled), {SEQ} frames per launch", "run", ms, 6.5 * 3840 * 2160 * SEQ, frames_per_launch=SEQ, us_per_frame=round(ms * 1e3 / SEQ, 3),
                 algorithmic_read_bytes=4 * 3840 * 2160 * SEQ, algorithmic_write_bytes=int(2.5 * 3840 * 2160) * SEQ)
        else:
            ms = lib.avifhipTimeRGBToYUVCycle(n, imgs, rgbs, 20, launches, None)
            emit("cfg4 (8 frames cycled)", "run", ms, 6.5 * 3840 * 2160, algorithmic_read_bytes=4 * 3840 * 2160, algorithmic_write_bytes=int(2.5 * 3840 * 2160))
    elif name in ("cfg2seq", "cfg2seq_fp32", "cfg2cold", "cfg2cold_fp32", "4kseq", "4kseq_fp32", "4kcold", "4kcold_fp32"):
        # the headline's frames where nothing is cache-resident (12 8K frames = 2.2 GB, 24 4K frames = 1.1 GB), one frame per launch (...cold) or
        # SEQ frames per launch (...seq: avifhipImageYUVToRGBBatchAsync over large frames)
        fp32 = name.endswith("_fp32")
        base = name[:-5] if fp32 else name
        w, h, frames = (3840, 2160, 24) if base.startswith("4k") else (7680, 4320, 12)
        pairs = [y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=fp32, seed=k % 4) for k in range(frames)]
        n, imgs, rgbs = arr(pairs)
        per = SEQ if base.endswith("seq") else 1
        lib.avifhipTimeStreamCeiling(n, imgs, rgbs, 0, 60000 // (33 if w > 4000 else 9), None)  # ~60 ms of ANOTHER kernel: the clocks are up before the profiled one starts
        launches = 1200
        if per > 1:
            ms = lib.avifhipTimeYUVToRGBBatchCycle(n, imgs, rgbs, per, 20, launches // per, None)
        else:
            ms = lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 20, launches, None)
        emit(f"{w}x{h} 8-bit 4:2:0 -> RGBA8 ({'fp32' if fp32 else 'integer'}), {frames} frames cycled, {per} per launch", "run", ms, 5.5 * w * h * per, frames_per_launch=per,
             us_per_frame=round(ms * 1e3 / per, 3), algorithmic_read_bytes=int(1.5 * w * h) * per, algorithmic_write_bytes=4 * w * h * per)
    else:
        raise SystemExit("run: cfg3, cfg4, cfg4seq, cfg2seq[_fp32], cfg2cold[_fp32], 4kseq[_fp32] or 4kcold[_fp32]")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "run":
        run_only(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[1] == "ab":
        ab(sys.argv[2], [int(x, 0) for x in sys.argv[3:]])
        sys.exit(0)
    for n in sys.argv[1:] or ["cfg3", "cfg4", "cfg5x64", "cfg5grid"]:
        run(n)
