#!/usr/bin/env python3
"""rocprofv3 --pmc passes over scripts/r06/exp_config5_affinity.py <S ...> -> per view and S the counters of the TIMED launches of
trace_kernel (launches 12 .. 17 of the 18 a case makes).  usage: affinity_counters.py <dir with the passes' csv> <S ...>"""
import collections
import csv
import glob
import re
import sys

base, sizes = sys.argv[1], [int(a) for a in sys.argv[2:]]
rows = collections.OrderedDict()
for f in glob.glob(f"{base}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not re.search(r"trace_kernel", r["Kernel_Name"]):
            continue
        key = (int(r["Start_Timestamp"]), r.get("Dispatch_Id", ""))
        d = rows.setdefault(key, {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
launches = [rows[k] for k in sorted(rows)]
per_case = 18
k = 0
for view in ("outside", "tunnel"):
    for S in sizes:
        timed = launches[k + 11:k + 17]
        k += per_case
        if not timed:
            continue
        names = sorted(n for n in timed[0] if n != "dur")
        mean = lambda n: sum(t.get(n, 0.0) for t in timed) / len(timed)  # noqa: E731
        out = [f"{view:8s} S = {S:2d}: {mean('dur') / 1e6:8.3f} ms under the profiler"]
        for n in names:
            out.append(f"{n} {mean(n):.4g}")
        if "TCC_HIT_sum" in names:
            out.append(f"L2 hit rate {mean('TCC_HIT_sum') / (mean('TCC_HIT_sum') + mean('TCC_MISS_sum')):.4f}")
        if "FETCH_SIZE" in names:
            out.append(f"fetched {mean('FETCH_SIZE') * 1024 / 1e9:.3f} GB per frame")
        print("; ".join(out))
