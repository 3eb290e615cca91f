"""Summarise rocprofv3 --pmc / --kernel-trace sqlite outputs: python tests/tools/pmc_summary.py <dir-or-db>...
Prints, per kernel name, the dispatch count and the per-dispatch average of every counter (and the duration)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def summarise(path):
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for f in dbs:
        db = sqlite3.connect(f)
        names = {r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")}
        if "counters_collection" in names:
            acc = defaultdict(lambda: defaultdict(list))
            for kname, cname, value, disp in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
                acc[kname][cname].append((disp, value))
            for kname, ctrs in acc.items():
                print(f"{os.path.basename(f)}: {kname[:110]}")
                for cname, vals in sorted(ctrs.items()):
                    per = defaultdict(float)
                    for d, v in vals:
                        per[d] += v
                    xs = list(per.values())
                    print(f"    {cname:28s} dispatches={len(xs):4d} avg={sum(xs)/len(xs):16.1f}")
        if "kernels" in names:
            acc = defaultdict(list)
            for kname, dur in db.execute("select name, duration from kernels"):
                acc[kname].append(dur)
            for kname, xs in acc.items():
                xs.sort()
                print(f"{os.path.basename(f)}: {kname[:110]}  calls={len(xs)} avg={sum(xs)/len(xs)/1e3:.2f} us min={xs[0]/1e3:.2f} med={xs[len(xs)//2]/1e3:.2f}")


for a in sys.argv[1:]:
    summarise(a)
