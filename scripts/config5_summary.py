#!/usr/bin/env python3
"""gpurun_out/c5_<tag>/ (scripts/profile_config5.sh) -> summary.json: trace_kernel on BASELINE config 5's scene, per view (outside /
inside a tunnel) and per scene format (8-byte records / wide records): average duration, FETCH_SIZE, L2 hit rate, SQ counters."""
import collections
import csv
import glob
import json
import re
import sys

tag = sys.argv[1]
base = f"gpurun_out/c5_{tag}"


def per_launch(sub, counters=None):
    """trace_kernel launches in submission order -> list of dicts (duration_ns + counters)"""
    rows = []
    pat = f"{base}/{sub}/**/*counter_collection.csv" if counters else f"{base}/{sub}/**/*kernel_trace.csv"
    by_id = collections.OrderedDict()
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if not re.search(r"trace_kernel", r["Kernel_Name"]):
                continue
            key = (int(r["Start_Timestamp"]), r.get("Dispatch_Id", ""))
            d = by_id.setdefault(key, {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            if counters:
                d[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(by_id):
        rows.append(by_id[k])
    return rows


def views(rows):
    """exp_config5.py: per view 3 warm-up + 10 timed launches"""
    return {"outside": rows[3:13], "tunnel": rows[16:26]}


def mean(xs):
    xs = list(xs)
    return sum(xs) / len(xs) if xs else None


out = {"workload": "scripts/exp_config5.py 2048: level-7 Menger clipped to 2048^3 (261 140 230 nodes, 5.58 GiB as 8-byte records + leaf words), "
                   "3840x2160, 8 bounces, all-in-one trace_kernel, one frame per launch", "formats": {}}
# the variants of a run: `prefix=label` arguments (scripts/profile_config5_order.sh), default the two scene formats of profile_config5.sh
variants = [a.split("=", 1) for a in sys.argv[2:]] or [("w0", "8-byte records"), ("w1", "wide records (two levels per 16-byte record)")]
for pre, label in variants:
    entry = {}
    try:
        entry["reported_by_the_run"] = [l.strip() for l in open(f"{base}/{pre}_stats.txt") if "ms/frame" in l or "built in" in l]
    except OSError:
        pass
    st = views(per_launch(f"{pre}_stats"))
    fe = views(per_launch(f"{pre}_fetch", True))
    tc = views(per_launch(f"{pre}_tcc", True))
    sq = views(per_launch(f"{pre}_sq", True))
    for v in ("outside", "tunnel"):
        e = {}
        if st[v]:
            e["avg_ms"] = round(mean(r["dur"] for r in st[v]) / 1e6, 4)
        if fe[v]:
            kb = mean(r.get("FETCH_SIZE", 0) for r in fe[v])
            ms = mean(r["dur"] for r in fe[v]) / 1e6
            e["fetch_size_kb"] = kb
            e["fetch_gbs_raw_at_profiled_duration"] = round(kb * 1024 / (ms * 1e-3) / 1e9, 1)
            if "avg_ms" in e:
                e["fetch_gbs_raw"] = round(kb * 1024 / (e["avg_ms"] * 1e-3) / 1e9, 1)
                e["fetch_gbs_read_doubled"] = round(2 * kb * 1024 / (e["avg_ms"] * 1e-3) / 1e9, 1)
        if tc[v]:
            h, m = mean(r.get("TCC_HIT_sum", 0) for r in tc[v]), mean(r.get("TCC_MISS_sum", 0) for r in tc[v])
            e["l2_hit_rate"] = round(h / (h + m), 4) if h + m else None
        if sq[v]:
            g = lambda k: mean(r.get(k, 0) for r in sq[v])  # noqa: E731
            e["valu_wave_instr"] = g("SQ_INSTS_VALU")
            e["salu_wave_instr"] = g("SQ_INSTS_SALU")
            e["vmem_rd_wave_instr"] = g("SQ_INSTS_VMEM_RD")
            e["lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 4) if g("SQ_ACTIVE_INST_VALU") else None
            e["waitcnt_share_of_wave_cycles"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None
            if "avg_ms" in e:
                e["valu_issue_slot_frac"] = round(g("SQ_INSTS_VALU") * 2 / (1024 * 2.4e9 * e["avg_ms"] * 1e-3), 4)
        entry[v] = e
    out["formats"][label] = entry
json.dump(out, open(f"{base}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
