"""profile_digest.py <gpurun_out/tag> <profiles/prefix> -- digest of the rocprofv3 databases written by
tests/tools/profile_round.sh into the summaries committed under profiles/:
  <prefix>_kernel_stats.txt   per-kernel calls / total / average / min / max duration (the --stats view)
  <prefix>_pmc.txt            per-kernel per-dispatch averages of every collected counter
  profiles/pmc_traffic.json   HBM bytes per launch of the dominant kernel (read by bench.py -> roofline.traffic)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 bytes... see MI355X_MICROARCH.md "HBM":
bytes = counter * 1024 on this stack is WRONG for gfx950 FETCH_SIZE, which tallies 128-byte requests at 64 bytes:
FETCH bytes = FETCH_SIZE * 1024 * 2 for wide coalesced streaming reads; WRITE_SIZE is uncalibrated and is reported
both raw and calibrated against the kernel's known store volume (every output byte is stored exactly once)."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def dbs(path):
    return sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))


def _geometry_name(kname, grid_z, workgroup_z):
    """Round 6: one kernel instantiation serves single images (grid z = 1) and sequence launches (grid z = frames): a profiler's per-kernel
    average would mix 30 us and 120 us launches and their bytes.  Launches with more than one layer of workgroups carry it in their name."""
    layers = (grid_z // workgroup_z) if workgroup_z else 1
    return kname if layers <= 1 else f"{kname} [grid z = {layers}]"


def kernel_stats(path, by_geometry=True):
    rows = defaultdict(list)
    for f in dbs(path):
        db = sqlite3.connect(f)
        names = {r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")}
        if "kernels" in names:
            cols = {r[1] for r in db.execute("pragma table_info(kernels)")}
            if by_geometry and {"grid_z", "workgroup_z"} <= cols:
                for kname, dur, gz, wz in db.execute("select name, duration, grid_z, workgroup_z from kernels"):
                    rows[_geometry_name(kname, gz, wz)].append(dur)
            else:
                for kname, dur in db.execute("select name, duration from kernels"):
                    rows[kname].append(dur)
    return rows


def counters(path, by_geometry=True):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for f in dbs(path):
        db = sqlite3.connect(f)
        names = {r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")}
        if "counters_collection" in names:
            layers = {}
            if by_geometry and "kernels" in names and {"grid_z", "workgroup_z", "dispatch_id"} <= {r[1] for r in db.execute("pragma table_info(kernels)")}:
                layers = {d: (gz, wz) for d, gz, wz in db.execute("select dispatch_id, grid_z, workgroup_z from kernels")}
            for kname, cname, value, disp in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
                gz, wz = layers.get(disp, (1, 1))
                acc[_geometry_name(kname, gz, wz)][cname][disp] += value
    return acc


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    stats = kernel_stats(os.path.join(src, "stats"))
    total = sum(sum(v) for v in stats.values()) or 1
    lines = ["rocprofv3 --kernel-trace --stats -- python bench.py --steps 1000 --warmup 100 --repeats 3 --streams 1 --no-second-stream-count --no-cpu-baseline --headline-only",
             f"{'kernel':110s} {'calls':>6s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}"]
    dominant = None
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        if dominant is None:
            dominant = k
        lines.append(f"{k[:110]:110s} {len(v):6d} {sum(v)/1e3:12.1f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} {100*sum(v)/total:6.1f}")
    open(prefix + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:6]))

    out = []
    per_kernel = defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        c = counters(os.path.join(src, sub))
        for kname, ctrs in c.items():
            for cname, per in ctrs.items():
                xs = list(per.values())
                per_kernel[kname][cname] = (len(xs), sum(xs) / len(xs))
    for kname, ctrs in per_kernel.items():
        out.append(kname[:150])
        for cname, (n, avg) in sorted(ctrs.items()):
            out.append(f"    {cname:28s} dispatches={n:5d} avg_per_dispatch={avg:18.1f}")
    open(prefix + "_pmc.txt", "w").write("\n".join(out) + "\n")

    if dominant and dominant in per_kernel:
        c = per_kernel[dominant]
        fetch = c.get("FETCH_SIZE", (0, 0.0))[1]
        write = c.get("WRITE_SIZE", (0, 0.0))[1]
        bench = {}
        try:
            bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
        except Exception:
            pass
        alg = bench.get("roofline", {}).get("algorithmic_bytes_per_launch", 182476800)
        traffic = {
            "kernel": dominant,
            "kernel_family": bench.get("config", {}).get("kernel", ""),  # the library's name for it: bench.py uses the figure only for this kernel
            "FETCH_SIZE_avg": fetch, "WRITE_SIZE_avg": write,
            "fetch_bytes_raw": fetch * 1024, "fetch_bytes_gfx950_corrected": fetch * 1024 * 2,
            "write_bytes_raw": write * 1024,
            "algorithmic_read_bytes": int(alg * 1.5 / 5.5), "algorithmic_write_bytes": int(alg * 4 / 5.5),
            "traffic_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated, raw*1024",
        }
        open(os.path.join(os.path.dirname(prefix) or ".", "pmc_traffic.json"), "w").write(json.dumps(traffic, indent=1) + "\n")
        print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
