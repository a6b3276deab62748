#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the summaries kept under profiles/: kernel stats of the default
bench run, HBM bytes per launch of the two trace-stage kernels (FETCH_SIZE / WRITE_SIZE passes), SQ instruction counters."""
import collections
import csv
import glob
import json
import re
import shutil
import sys

tag = sys.argv[1]
base = f"gpurun_out/prof_{tag}"
out = {}
KERNELS = r"(trace_kernel|bounce_kernel|path_kernel|tile_hist_kernel|tile_scan_kernel|tile_scatter_kernel|primary_kernel|shade_kernel|trace_rays_kernel)"


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{base}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(KERNELS, r["Kernel_Name"])
            if m:
                agg[m[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


stats = glob.glob(f"{base}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"{base}/kernel_stats.csv")
    rows = list(csv.DictReader(open(stats[0])))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:6]]
    for r in out["kernel_stats"]:
        m = re.search(KERNELS, r["Name"])
        r["Name"] = m[0] if m else r["Name"][:60]
try:
    out["bench_under_stats"] = json.loads(open(f"{base}/stats_bench.json").read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    out["bench_under_stats"] = f"unreadable: {e}"

fetch, write = counters("fetch"), counters("write")
traffic = {}
for k in ("trace_kernel", "bounce_kernel"):
    f = fetch.get(k, {}).get("FETCH_SIZE", [])
    w = write.get(k, {}).get("WRITE_SIZE", [])
    if f and w:
        # the warm-up launches are in the list too; the last ones are steady state (tile order trained)
        fk, wk = sum(f[-4:]) / len(f[-4:]), sum(w[-4:]) / len(w[-4:])
        traffic[k] = {"launches_seen": len(f), "fetch_size_kb": fk, "write_size_kb": wk,
                      "bytes_raw": (fk + wk) * 1024, "bytes_read_doubled": (2 * fk + wk) * 1024}
if traffic:
    try:
        b = json.loads(open(f"{base}/fetch_bench.json").read().strip().splitlines()[-1])
        fpl = b["roofline"]["launch"]["frames_per_launch"]
        alg = b["roofline"]["launch"]["algorithmic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        fpl, alg = None, None
    out["traffic"] = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- python3 bench.py --steps 96 --warmup 32 --blocks 2 --no-cpu-baseline",
        "frames_per_launch": fpl, "algorithmic_bytes_per_launch": alg, "per_kernel": traffic,
        "hbm_bytes_per_launch_raw": sum(t["bytes_raw"] for t in traffic.values()),
        "hbm_bytes_per_launch_corrected": sum(t["bytes_read_doubled"] for t in traffic.values()),
        "note": "one 'launch' = trace_kernel + bounce_kernel over frames_per_launch frames.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts a wide "
                "coalesced read at half its bytes, so 'corrected' doubles the read side; WRITE_SIZE is exact for 16-byte stores.  The reads here are 8-byte "
                "SVO records, 4-byte noise words and the 64-byte path records of the tail queue, so the true figure lies between raw and corrected.",
    }

sq = counters("sq")
out["sq"] = {}
for k, c in sq.items():
    n = len(c["SQ_WAVES"])
    if not n:
        continue
    valu, act, thr = sum(c["SQ_INSTS_VALU"]), sum(c["SQ_ACTIVE_INST_VALU"]), sum(c["SQ_THREAD_CYCLES_VALU"])
    out["sq"][k] = {"launches": n, "valu_wave_instr_per_launch": valu / n, "salu_per_launch": sum(c["SQ_INSTS_SALU"]) / n,
                    "lds_per_launch": sum(c["SQ_INSTS_LDS"]) / n, "waves_per_launch": sum(c["SQ_WAVES"]) / n,
                    "lane_utilisation": thr / (act * 64) if act else None,
                    "waitcnt_share_of_wave_cycles": sum(c["SQ_WAIT_ANY"]) / sum(c["SQ_WAVE_CYCLES"]) if sum(c["SQ_WAVE_CYCLES"]) else None}
json.dump(out, open(f"{base}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
