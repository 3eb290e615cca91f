"""bench.py's multi-rank plumbing without a GPU (`--dry-run`): process group over gloo, barriers around the timed regions, MAX over ranks,
rank 0 printing exactly ONE JSON line on stdout (communication libraries' banners kept off it), n_gpus = the ranks that ran, the weak /
strong scaling labels, cfg5's contiguous tile blocks.  The conversions themselves are replaced by a sleeping stand-in: the numbers mean
nothing and the line says so.  The 8-GPU run itself is the driver's; this keeps its first real run from being the code's first run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench(args, launcher_ranks=None, timeout=600, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    if launcher_ranks:  # the driver's way: torch.distributed.run starts the ranks
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={launcher_ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.fspath(ROOT / "bench.py")] + args
    else:  # started plainly: bench.py spawns its ranks itself
        cmd = [sys.executable, os.fspath(ROOT / "bench.py")] + args
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {proc.stdout[:2000]}"
    return json.loads(lines[0])


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("ranks,self_spawn", [(1, True), (2, False), (4, True), (8, False)])
def test_frames_workload_line(ranks, self_spawn):
    d = _bench(["--dry-run", "--gpus", str(ranks), "--steps", "30", "--warmup", "3", "--repeats", "2"], None if self_spawn else ranks)
    assert CONTRACT_KEYS <= set(d), sorted(CONTRACT_KEYS - set(d))
    assert d["n_gpus"] == ranks and d["steps"] == 30 and d["warmup"] == 3
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["data"].startswith("dry-run") and d["cpu_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # whole-job aggregate: every rank converts its own frames, the region's time is the MAX over ranks
    assert abs(d["value"] - ranks * 256 * 128 / 1e6 * 30 / (d["ms_per_step"] * 30 / 1e3)) / d["value"] < 1e-3
    for key in ("roofline", "fp32", "integer", "planes_4k"):
        assert key in d
    assert {"frac", "achieved", "peak", "bound", "traffic", "cold", "kernel_ms_inputs_cache_resident"} <= set(d["roofline"])
    assert {"kernel_ms", "frac", "value"} <= set(d["fp32"]) and {"integer", "fp32"} <= set(d["planes_4k"])
    # round 6: sequence rows beside the single-frame ones, and with more than one rank the other two multi-GPU shapes in the same line
    assert {"frames_per_launch", "frames_cycled", "us_per_frame", "frac", "ceiling", "traffic"} <= set(d["roofline"]["cold_batched"])
    assert {"inputs_cache_resident", "cold", "kernel"} <= set(d["integer"]["sequence"]) and "sequence" in d["planes_4k"]["integer"]
    if ranks > 1:
        strong = d["strong_scaling_cfg5"]
        assert strong["n_gpus"] == ranks and strong["scaling"] == "strong" and sum(strong["config"]["tiles_per_rank"]) == 64
        assert "in_process_host_to_host" in d  # (None in a dry run: it needs the devices)
    else:
        assert "strong_scaling_cfg5" not in d and "in_process_host_to_host" not in d


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_grid_workload_shards_the_64_tiles(ranks):
    d = _bench(["--dry-run", "--gpus", str(ranks), "--workload", "cfg5", "--steps", "5", "--warmup", "1", "--repeats", "2"], ranks)
    assert d["n_gpus"] == ranks and d["scaling"] == "strong"
    blocks = d["config"]["tiles_per_rank"]
    assert len(blocks) == ranks and sum(blocks) == 64 and max(blocks) - min(blocks) <= 1
    assert d["data"].startswith("dry-run") and "host_to_host" in d


def test_launcher_world_overrides_the_flag():
    """`--gpus 2` under a launcher that started 3 ranks: the line reports the ranks that ran."""
    d = _bench(["--dry-run", "--gpus", "2", "--steps", "10", "--warmup", "1", "--repeats", "1"], 3)
    assert d["n_gpus"] == 3


@pytest.mark.parametrize("self_spawn", [True, False])
def test_side_blocks_cannot_cost_the_headline(self_spawn):
    """A rank that hangs in the N > 1 side blocks (cfg5 strong scaling, the in-process farm: never yet run on more than one physical GPU):
    past the timeout rank 0 prints the line it already has, every rank leaves with status 0, stdout still carries exactly one line."""
    d = _bench(["--dry-run", "--gpus", "2", "--steps", "10", "--warmup", "1", "--repeats", "1"], None if self_spawn else 2, extra_env={"AVIFHIP_BENCH_SIDE_TIMEOUT": "0.001"})
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert "did not finish" in d["strong_scaling_cfg5"]["error"] and d["in_process_host_to_host"] is None
    d = _bench(["--dry-run", "--gpus", "2", "--steps", "10", "--warmup", "1", "--repeats", "1"], None if self_spawn else 2, extra_env={"AVIFHIP_BENCH_SIDE_BLOCKS": "0"})
    assert d["n_gpus"] == 2 and "strong_scaling_cfg5" not in d
