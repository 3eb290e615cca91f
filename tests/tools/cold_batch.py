"""Sequences walked N frames per launch (avifhipImageYUVToRGBBatchAsync) in the regime where every byte comes from and goes to HBM: 12 8K frames
(2.2 GB) or 24 4K frames (1.1 GB) cycled, both arithmetics, with the byte-movement ceiling of the same launch shape beside each row.
    python tests/tools/cold_batch.py [8k] [4k]
One process, one box, so that the rows compare.  Each row: microseconds per FRAME, fraction of 8 TB/s on the algorithmic bytes (5.5 B/px)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from stream_sweep import arr, burst, lib, y2r  # noqa: E402
from libavif_amd import abi, native  # noqa: E402

PEAK = 8000.0


def rows(label, w, h, count, per_launch):
    alg = 5.5 * w * h
    for fam, avoid in (("integer", False), ("fp32", True)):
        pairs = [y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=avoid, seed=k % 4) for k in range(count)]
        n, imgs, rgbs = arr(pairs)
        for per in per_launch:
            if per == 1:
                ms = burst(lib.avifhipTimeYUVToRGBCycle, n, imgs, rgbs)
                ceil = burst(lib.avifhipTimeStreamCeiling, n, imgs, rgbs)
            else:
                ms = burst(lib.avifhipTimeYUVToRGBBatchCycle, n, imgs, rgbs, per) / per
                kernel = native.last_kernel()
                ceil = burst(lib.avifhipTimeStreamCeilingBatchCycle, n, imgs, rgbs, per) / per
            kernel = native.last_kernel() if per == 1 else kernel
            print(json.dumps({"config": label, "arithmetic": fam, "frames_cycled": count, "frames_per_launch": per, "us_per_frame": round(ms * 1e3, 2),
                              "frac": round(alg / (ms * 1e-3) / 1e9 / PEAK, 4), "ceiling_us_per_frame": round(ceil * 1e3, 2),
                              "ceiling_frac": round(alg / (ceil * 1e-3) / 1e9 / PEAK, 4), "kernel": kernel}), flush=True)
        del pairs


if __name__ == "__main__":
    which = sys.argv[1:] or ["8k", "4k"]
    lib.avifhipSetArithmetic(0)
    if "8k" in which:
        rows("cfg2 cold (7680x4320)", 7680, 4320, 12, (1, 2, 3, 4, 6))
    if "4k" in which:
        rows("planes_4k cold (3840x2160)", 3840, 2160, 24, (1, 2, 4, 8))
    if "4kwarm" in which:
        rows("planes_4k, 4 frames cycled", 3840, 2160, 4, (1, 2, 4))
