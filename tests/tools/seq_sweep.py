"""Sequence launches (avifhipImageYUVToRGBBatchAsync over large frames: api_batch.cpp sequenceAsync) of cold 8K / 4K frames under the
result-preserving geometry knobs (plan.h TuningBits through avifhipSetTuning), one process, one box, so that the rows compare.
    python tests/tools/seq_sweep.py [8k] [4k] [per=4]
Each row: microseconds per FRAME and the fraction of 8 TB/s on the algorithmic bytes (5.5 B/px)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from stream_sweep import arr, burst, lib, y2r  # noqa: E402
from libavif_amd import abi, native  # noqa: E402

PEAK = 8000.0
KNOBS = [("default", 0x1), ("raster", 0x0), ("bands, 4 strips", 0x401), ("raster, 2 strips", 0x200), ("raster, 4 strips", 0x400),
         ("2 waves side by side", 0x20001), ("4 waves side by side", 0x30001), ("4 side by side, raster, 2 strips", 0x30200),
         ("4 side by side, raster, 4 strips", 0x30400), ("2 tile rows per chunk", 0x200001), ("4 tile rows per chunk", 0x400001),
         ("private halo", 0x11)]


def rows(label, w, h, count, per):
    alg = 5.5 * w * h
    for fam, avoid in (("integer", False), ("fp32", True)):
        pairs = [y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=avoid, seed=k % 4) for k in range(count)]
        n, imgs, rgbs = arr(pairs)
        for name, bits in KNOBS:
            lib.avifhipSetTuning(bits)
            ms = burst(lib.avifhipTimeYUVToRGBBatchCycle, n, imgs, rgbs, per) / per
            print(json.dumps({"config": label, "arithmetic": fam, "frames_per_launch": per, "knob": name, "tuning": hex(bits), "us_per_frame": round(ms * 1e3, 2),
                              "frac": round(alg / (ms * 1e-3) / 1e9 / PEAK, 4), "kernel": native.last_kernel()}), flush=True)
        lib.avifhipSetTuning(1)
        ceil = burst(lib.avifhipTimeStreamCeilingBatchCycle, n, imgs, rgbs, per) / per
        print(json.dumps({"config": label, "arithmetic": fam, "frames_per_launch": per, "knob": "byte-movement ceiling", "us_per_frame": round(ceil * 1e3, 2),
                          "frac": round(alg / (ceil * 1e-3) / 1e9 / PEAK, 4), "kernel": native.last_kernel()}), flush=True)
        del pairs


if __name__ == "__main__":
    args = sys.argv[1:]
    per = 4
    for a in args:
        if a.startswith("per="):
            per = int(a[4:])
    which = [a for a in args if not a.startswith("per=")] or ["8k"]
    lib.avifhipSetArithmetic(0)
    if "8k" in which:
        rows("cfg2 cold (7680x4320)", 7680, 4320, 12, per)
    if "4k" in which:
        rows("planes_4k cold (3840x2160)", 3840, 2160, 24, per)
