"""The committed measurement evidence is self-consistent: the bench line printed under rocprofv3 and the rocprofv3 kernel statistics of
the same command agree on the dominant kernel's duration, the roofline fields follow from each other, and the HBM traffic the PMC
passes measured matches the algorithmic bytes the roofline is computed from (profiles/r01_bench_*)."""
import json
import re
from pathlib import Path

PROFILES = Path(__file__).resolve().parent.parent / "profiles"


def _line(name):
    return json.loads((PROFILES / name).read_text().strip().splitlines()[-1])


def test_bench_line_and_rocprof_stats_agree():
    line = _line("r01_bench_line_under_rocprof.json")
    stats = (PROFILES / "r01_bench_kernel_stats.txt").read_text().splitlines()
    assert "bench.py" in stats[0]
    dominant = stats[2]
    assert "yuvToRgbTileFxKernel<unsigned char" in dominant  # the integer path's 8-bit 4:2:0 bilinear kernel the bench line names
    assert line["config"]["kernel"].startswith("yuv2rgb_fixed_tile<u8,420,bilinear,rgba8")
    avg_us = float(re.split(r"\s{2,}", dominant.strip())[-4])
    event_us = line["roofline"]["kernel_ms_hbm_streaming"] * 1e3
    assert abs(avg_us - event_us) / avg_us < 0.10, (avg_us, event_us)


def test_roofline_fields_follow_from_each_other():
    for name in ("r01_bench_line.json", "r01_bench_line_under_rocprof.json", "r01_bench_line_default_run.json"):
        d = _line(name)
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        alg = 7680 * 4320 * 5.5
        assert r["algorithmic_bytes_per_launch"] == alg
        assert abs(r["achieved"] - alg / (r["kernel_ms_hbm_streaming"] * 1e-3) / 1e9) / r["achieved"] < 0.01
        assert abs(r["traffic"] - alg) / alg < 0.02  # measured HBM bytes per launch vs algorithmic bytes: no wasted re-reads
        assert abs(d["value"] - 7680 * 4320 / 1e6 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
        assert d["metric"].startswith("megapixels/sec") and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None


def test_default_run_carries_the_cpu_baseline():
    cb = _line("r01_bench_line_default_run.json")["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["unit"] == "megapixels/s" and 50 < cb["value"] < 500
