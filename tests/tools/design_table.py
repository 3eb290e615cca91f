"""Prints DESIGN.md section 4.5's table from the round's committed evidence (profiles/<tag>_cfgs_bench.jsonl + <tag>_cfgs_kernel_stats.txt).
    python tests/tools/design_table.py [tag]"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
DESC = {
    "cfg2": "cfg2: 8K 8-bit 4:2:0 BT.709 limited → RGBA8 bilinear (4 frames cycled)",
    "cfg2_4k": "cfg2 at 4K (3840 × 2160; the north star's second plane size), 4 frames cycled (L3-resident planes)",
    "cfg2_cold": "cfg2, 12 frames cycled (HBM regime), one frame per launch",
    "cfg2_seq": "cfg2 as a sequence: 4 frames per launch, 4 cycled (L3-resident planes)",
    "cfg2_seq_cold": "cfg2 as a sequence: 4 frames per launch, 12 cycled (HBM regime)",
    "cfg2_4k_seq": "4K as a sequence: 4 frames per launch, 4 cycled (L3-resident planes)",
    "cfg2_4k_cold": "4K, 24 frames cycled (HBM regime), one frame per launch",
    "cfg2_4k_seq_cold": "4K as a sequence: 4 frames per launch, 24 cycled (HBM regime)",
    "cfg4_cycled": "cfg4, 8 frames cycled (HBM regime), one frame per launch",
    "cfg4_seq": "cfg4 as a sequence: 4 frames per launch (avifhipImageRGBToYUVBatchAsync), 8 cycled (HBM regime)",
    "gainmap4k_photo": "gain-map application on a photograph-like pair (neighbouring pixels hold neighbouring codes)",
    "gmcompute4k_dev": "gain-map computation, device-resident (avifhipRGBImageComputeGainMapAsync), per call",
    "gainmap4k_same": "gain-map application, output in the base image's primaries (no fp64 matrix per pixel)",
    "cfg2_565_odd": "… RGB565 on rows aligned to 2 bytes only (width 7679; round 5: universal kernel)",
    "cfg2_565_alpha": "… RGB565 from planes with an alpha plane, multiplied in inside the loop (round 5: universal kernel)",
    "cfg2_565_10": "… RGB565 from 10-bit planes (integer: Convert16To8Plane in the packed kernels' front end; round 5: universal kernel)",
    "cfg4_444_8k": "8K RGBA8 → BT.709 4:4:4 + A (avifenc -y 444)",
    "cfg4rgb_8k": "cfg4 from RGB8 at 8K",
    "cfg2n": "cfg2 with nearest upsampling",
    "cfg2_rgb": "cfg2 → RGB8 (3-byte pixels)",
    "cfg2_565": "cfg2 → RGB565, nearest (Android bitmaps)",
    "cfg2_alpha": "cfg2 + alpha plane → RGBA8",
    "cfg2_premul": "cfg2 + alpha plane → RGBA8 premultiplied",
    "cfg3": "cfg3: 8K 10-bit 4:4:4 + A → RGBA16 premultiplied",
    "cfg4": "cfg4: 4K RGBA8 → 8-bit 4:2:0 BT.709 + A plane",
    "cfg4rgb": "cfg4 from RGB8",
    "cfg4_601": "cfg4 with BT.601 (avifenc's default matrix)",
    "cfg4_8k": "cfg4 at 8K",
    "cfg4_premul_8k": "… with a pending alpha multiply (`avifenc --premultiply`)",
    "cfg4_unpremul_8k": "… with a pending alpha un-multiply",
    "cfg4_ycgco_8k": "8K RGBA8 → YCgCo 4:4:4 + A",
    "ident8_enc": "lossless encode: 8K RGBA8 → identity 8-bit 4:4:4 + A",
    "gray_enc_8k": "8K GRAY8 → luma plane",
    "graya_enc_8k": "8K GRAYA8 → luma + alpha planes",
    "cfg5": "cfg5 tile: 1080p 10-bit 4:2:0 → RGBA(10) bilinear, single launch (launch-bound)",
    "cfg5_8": "cfg5 tile → RGBA8",
    "cfg5x64": "cfg5 canvas: 64 tiles in one batch launch → RGBA(10)",
    "cfg5x64_8": "cfg5 canvas → RGBA8",
    "f16_420": "8K 10-bit 4:2:0 → RGBA F16",
    "f16_444a": "8K 10-bit 4:4:4 + A → RGBA F16",
    "ident8": "8K identity 8-bit 4:4:4 → RGBA8 (byte shuffle)",
    "ident8rgb": "… → RGB8",
    "gray8": "8K 8-bit 4:2:0 → GRAY8",
    "graya16": "8K 10-bit + alpha → GRAYA16",
    "premul8": "`avifRGBImagePremultiplyAlpha` in place, 8K RGBA8",
    "unpremul8": "`avifRGBImageUnpremultiplyAlpha` in place, 8K RGBA8",
    "unpremul16": "… 8K RGBA16",
    "tail0": "decode-side tail (§4.6): 8K cfg2 + crop (8,4,7664,4312), fused",
    "tail180": "… + half turn + mirror, fused",
    "tail90": "… + quarter turn + mirror, fused",
    "tail90_two_pass": "… + quarter turn, two passes (conversion, then `avifhipRGBImageTransformAsync`)",
    "tail0_10": "tail from 10-bit planes → RGBA8: crop only",
    "tail90_10": "… + quarter turn",
    "tail0_rgba10": "tail, 10-bit planes → RGBA(10) (the API default depth): crop only",
    "tail180_rgba10": "… + half turn + mirror",
    "tail90_rgba10": "… + quarter turn + mirror",
    "tail90_rgba10_two_pass": "… in two passes",
    "cfg5grid": "cfg5 through the grid entry point (tiles where the decoder left them) → RGBA(10); the library's choice: tile batch + seam kernel for the fp32 kernels above 32 megapixels",
    "cfg5grid_link": "… tiles and seams in ONE launch forced (`AVIFHIP_GRID_SEAM_PASS=0`: the seam-aware fp32 kernels)",
    "cfg5grid_8": "… → RGBA8; the library's choice: ONE launch in the packed 16-bit kernels, tile batch + seam kernel in the fp32 ones",
    "cfg5grid_8_pass": "… → RGBA8 with the seam pass forced (`AVIFHIP_GRID_SEAM_PASS=1`: two launches, rounds 1–3)",
    "photo_grid": "a phone photograph: 4032 × 3024 8-bit 4:2:0 as 8 × 6 tiles of 512 × 512 → RGBA8, grid entry point, same buffers call after call: ONE launch",
    "photo_grid_pass": "… with the seam pass forced (two launches, rounds 1–3)",
    "cfg2_keep": "cfg2 with `rgb->ignoreAlpha` (the destination's alpha bytes stay: read back per pixel, 4 B/pixel more from HBM than the fraction counts)",
    "cfg2_keep16": "… 8K 10-bit → RGBA(10) with `ignoreAlpha`",
    "xform90": "`avifhipRGBImageTransformAsync` alone: 8K RGBA8 crop + quarter turn",
    "xform180": "… crop + half turn",
    "scale_box4": "plane scaling 8K → 4K (§4.8)",
    "scale_up2": "… 4K → 8K",
    "scale_down_1_5": "… 8K → 5120 × 2880",
    "cfg2_unpremul": "cfg2 from PREMULTIPLIED planes + alpha → straight RGBA8 (un-multiply inside the conversion)",
    "cfg3_unpremul": "cfg3's planes, stored premultiplied → straight RGBA16",
    "cfg5x64_rot": "cfg5 canvas, the batch's OUTPUT buffers rotating between two sets (a fresh descriptor table per call)",
    "gainmap4k": "gain-map application (§4.10): 4K RGBA8 sRGB/BT.709 → RGBA10 PQ/BT.2020, 8-bit 4:4:4 gain map; whole call with light levels, one at a time under the profiler (the map's YUV→RGB inside the apply kernel since round 5; unprofiled and interleaved with the two-launch route: `r05_gainmap_call_ab.jsonl`, 44 µs, 29.7 µs without light levels)",
    "gainmap4k_half": "… with a half-size 4:2:0 gain map (rescaled on the device first)",
    "gmcompute4k": "gain-map computation from HOST images (4K RGBA8 + RGBA10 → 8-bit 4:4:4 gain map), transfers included",
}


def blocks():
    out = {}
    for block in (ROOT / "profiles" / f"{TAG}_cfgs_kernel_stats.txt").read_text().split("\n== ")[1:]:
        lines = block.splitlines()
        rows = []
        for l in lines[1:]:
            m = re.match(r"^\s{3}(\S.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", l)
            if m and not l.strip().startswith("kernel "):
                rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
        out[lines[0].strip()] = rows
    return out


DAGGER = []
rows = [json.loads(l) for l in (ROOT / "profiles" / f"{TAG}_cfgs_bench.jsonl").read_text().splitlines() if l.startswith("{")]
B = blocks()
print("| config | arithmetic | kernel | µs: HIP events or wall clock per call (rocprofv3 average of the same run) | fraction of 8 TB/s |")
print("|---|---|---|---|---|")
for r in rows:
    ks = B[r["config"]]
    per = r.get("frames_per_launch", 1)  # sequence rows: `us` is per FRAME, the profiler's average per launch of `per` frames
    closest = min((a for _, _, a in ks), key=lambda a: abs(a - r["us"] * per)) / per
    if r["clock"] == "events" or len(ks) == 1 or abs(closest - r["us"]) / closest < 0.06:
        prof = f"{closest:.1f}" + (f" = {closest * per:.1f} per launch of {per}" if per > 1 else "")
    else:
        most = max(c for _, c, _ in ks)
        prof = " + ".join(f"{a:.1f}" for _, c, a in ks if 2 * c >= most)
    clock = "" if r["clock"] == "events" else " per call"
    frac, mark = r["frac_of_8TBps"], ""
    if r["clock"] == "events" and closest < 12.0 and r["us"] - closest > 1.0:
        # a launch this short under the profiler (these rows' runs are profiled ones): rocprofv3 intercepts every launch, the stream runs dry between
        # two kernels and the events measure the launch rate (~10 us), not the kernel -- the fraction is the kernel's own duration's
        frac, mark = frac * r["us"] / closest, " †"
        DAGGER.append(r["config"])
    print(f"| {DESC.get(r['config'], r['config'])} | {r['arithmetic'].replace('float', 'fp32')} | `{r['kernel']}` | {r['us']:.1f}{clock} ({prof}) | {frac:.2f}{mark} |")
if DAGGER:
    print()
    print("† launches shorter than the ~10 µs at which a stream under rocprofv3 (every launch intercepted) issues kernels: the events of these profiled runs "
          "measure the launch rate, the fraction is taken from the kernel's own average duration in the same run (unprofiled, back to back: "
          "`tests/tools/cfg_bench.py` on its own, e.g. cfg4 7.7 µs, cfg4 from RGB8 6.6 µs on round 5's box).")

# the 4K rows without the profiler, from the same call (what the bench line's planes_4k block is compared with: tests/test_profiles_r06.py)
plain = ROOT / "profiles" / f"{TAG}_4k_rows.jsonl"
if plain.exists():
    got = [json.loads(l) for l in plain.read_text().splitlines() if l.startswith("{")]
    print()
    print("4K rows WITHOUT the profiler (same box, same call as the bench line; under rocprofv3 a launch this short is slowed by the interception): "
          + "; ".join(f"{DESC.get(r['config'], r['config']).split(',')[0].split(':')[0]} [{r['config']}] {r['arithmetic'].replace('float', 'fp32')} "
                      f"{r['us']:.2f} µs = {r['frac_of_8TBps']:.2f}" for r in got) + ".")
