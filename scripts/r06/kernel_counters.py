#!/usr/bin/env python3
"""rocprofv3 --pmc csv -> one row per (kernel, grid size): dispatches, mean duration, mean of every counter, derived figures.
usage: kernel_counters.py <dir> <kernel regex>"""
import collections
import csv
import glob
import re
import sys

base, pat = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(f"{base}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(pat, r["Kernel_Name"])
        if not m:
            continue
        name = re.sub(r"^void |vxrt::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        key = (name, int(r["Grid_Size"]), int(r["Start_Timestamp"]), r.get("Dispatch_Id", ""))
        d = rows.setdefault(key, {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups = collections.OrderedDict()
for (name, grid, _, _), d in rows.items():
    groups.setdefault((name, grid), []).append(d)
for (name, grid), ds in groups.items():
    ds = ds[1::2] if len(ds) >= 2 else ds            # every case is launched twice: keep the second launches
    mean = lambda n: sum(d.get(n, 0.0) for d in ds) / len(ds)  # noqa: E731
    names = sorted(n for n in ds[0] if n != "dur")
    out = [f"{name:44s} grid {grid:9d}: {len(ds):2d} launches, {mean('dur') / 1e6:8.3f} ms"]
    if "FETCH_SIZE" in names:
        out.append(f"fetched {mean('FETCH_SIZE') * 1024 / 1e9:7.3f} GB")
    if "TCC_HIT_sum" in names:
        out.append(f"L2 hit rate {mean('TCC_HIT_sum') / max(mean('TCC_HIT_sum') + mean('TCC_MISS_sum'), 1):.3f}")
    if "SQ_INSTS_VALU" in names:
        out.append(f"VALU wave-instr {mean('SQ_INSTS_VALU') / 1e6:8.1f} M")
        if mean("SQ_ACTIVE_INST_VALU"):
            out.append(f"lanes {mean('SQ_THREAD_CYCLES_VALU') / (64 * mean('SQ_ACTIVE_INST_VALU')):.3f}")
        if mean("SQ_WAVE_CYCLES"):
            out.append(f"s_waitcnt share {mean('SQ_WAIT_ANY') / mean('SQ_WAVE_CYCLES'):.3f}")
        out.append(f"issue slots {mean('SQ_INSTS_VALU') * 2 / (1024 * 2.4e9 * mean('dur') * 1e-9):.3f}")
    print("; ".join(out))
