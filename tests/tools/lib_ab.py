"""Two BUILDS of the library over one measurement on one box, interleaved: `python tests/tools/lib_ab.py <previous .so> <rounds> <stream_sweep args...>`
runs `stream_sweep.py <args>` alternately over libavif_amd/csrc/libavifhip.so and the given build (AVIFHIP_TOOLS_LIBRARY), `rounds` times each, and prints
the medians per emitted line.  Boxes differ by +-6 % and a process's first kernels by more: only interleaved medians say what a code change did
(how to get the previous build: `git stash; make -C libavif_amd/csrc; cp libavif_amd/csrc/libavifhip.so tests/tools/_ab/libavifhip_prev.so; git stash pop; make ...`)."""
import json
import os
import subprocess
import sys
from pathlib import Path
from statistics import median

HERE = Path(__file__).resolve().parent


def main():
    prev, rounds, args = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    got = {}
    for _ in range(rounds):
        for tag, lib in (("now", None), ("previous", prev)):
            env = dict(os.environ)
            if lib:
                env["AVIFHIP_TOOLS_LIBRARY"] = os.fspath(Path(lib).resolve())
            out = subprocess.run([sys.executable, os.fspath(HERE / "stream_sweep.py")] + args, capture_output=True, text=True, env=env, timeout=600)
            if out.returncode != 0:
                raise SystemExit(f"{tag}: {out.stderr[-2000:]}")
            for line in out.stdout.splitlines():
                if line.startswith("{"):
                    row = json.loads(line)
                    got.setdefault((row.get("config"), row.get("knob")), {}).setdefault(tag, []).append(row["us"])
    for (config, knob), both in got.items():
        print(json.dumps({"config": config, "knob": knob, "rounds": rounds, **{tag: {"us": round(median(v), 2), "all_us": v} for tag, v in both.items()},
                          "now_over_previous": round(median(both["now"]) / median(both["previous"]), 4) if "previous" in both and "now" in both else None}), flush=True)


if __name__ == "__main__":
    main()
