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

def traffic_of(suffix, args):
    fetch, write = counters("fetch" + suffix), counters("write" + suffix)
    traffic = {}
    for k in ("trace_kernel", "bounce_kernel"):
        f = fetch.get(k, {}).get("FETCH_SIZE", [])
        w = write.get(k, {}).get("WRITE_SIZE", [])
        if f and w:
            # the warm-up launches are in the list too; the last ones are steady state (tile order trained)
            fk, wk = sum(f[-4:]) / len(f[-4:]), sum(w[-4:]) / len(w[-4:])
            traffic[k] = {"launches_seen": len(f), "fetch_size_kb": fk, "write_size_kb": wk,
                          "bytes_raw": (fk + wk) * 1024, "bytes_read_doubled": (2 * fk + wk) * 1024}
    if not traffic:
        return None
    try:
        b = json.loads(open(f"{base}/fetch{suffix}_bench.json").read().strip().splitlines()[-1])
        fpl = b["roofline"]["launch"]["frames_per_launch"]
        alg = b["roofline"]["launch"]["algorithmic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        fpl, alg = None, None
    return {
        "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- python3 bench.py {args}",
        "command_args": args,
        "frames_per_launch": fpl, "algorithmic_bytes_per_launch": alg, "per_kernel": traffic,
        "hbm_bytes_per_launch_raw": sum(t["bytes_raw"] for t in traffic.values()),
        "hbm_bytes_per_launch_corrected": sum(t["bytes_read_doubled"] for t in traffic.values()),
        "note": "one 'launch' = trace_kernel + bounce_kernel over frames_per_launch frames.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts a wide "
                "coalesced read at half its bytes, so 'corrected' doubles the read side; WRITE_SIZE is exact for 16-byte stores.  The reads here are 8-byte "
                "SVO records, 4-byte noise words and the 64-byte path records of the tail queue, so the true figure lies between raw and corrected.",
    }


def sq_of(suffix):
    res = {}
    for k, c in counters("sq" + suffix).items():
        n = len(c["SQ_WAVES"])
        if not n:
            continue
        valu, act, thr = sum(c["SQ_INSTS_VALU"]), sum(c["SQ_ACTIVE_INST_VALU"]), sum(c["SQ_THREAD_CYCLES_VALU"])
        res[k] = {"launches": n, "valu_wave_instr_per_launch": valu / n, "salu_per_launch": sum(c["SQ_INSTS_SALU"]) / n,
                  "lds_per_launch": sum(c["SQ_INSTS_LDS"]) / n, "waves_per_launch": sum(c["SQ_WAVES"]) / n,
                  "lane_utilisation": thr / (act * 64) if act else None,
                  "waitcnt_share_of_wave_cycles": sum(c["SQ_WAIT_ANY"]) / sum(c["SQ_WAVE_CYCLES"]) if sum(c["SQ_WAVE_CYCLES"]) else None}
    return res


t = traffic_of("", "--steps 96 --warmup 32 --blocks 2 --no-cpu-baseline --no-extras")
if t:
    out["traffic"] = t
out["sq"] = sq_of("")
drv = {"traffic": traffic_of("_drv", "--steps 20 --warmup 5 --blocks 12 --no-cpu-baseline --no-extras"), "sq": sq_of("_drv")}
if drv["traffic"] or drv["sq"]:
    out["driver_schedule"] = drv

# One launch in flight: every kernel alone on the chip -> a per-kernel roofline from this file alone
stats1 = glob.glob(f"{base}/stats1/**/*kernel_stats.csv", recursive=True)
if stats1:
    shutil.copy(stats1[0], f"{base}/kernel_stats_inflight1.csv")
    try:
        b1 = json.loads(open(f"{base}/stats1_bench.json").read().strip().splitlines()[-1])
        alg = float(b1["roofline"]["launch"]["algorithmic_bytes_per_launch"])
        fpl = b1["roofline"]["launch"]["frames_per_launch"]
        rows = {re.search(KERNELS, r["Name"])[0]: r for r in csv.DictReader(open(stats1[0])) if re.search(KERNELS, r["Name"])}
        pair = sum(float(rows[k]["AverageNs"]) for k in ("trace_kernel", "bounce_kernel") if k in rows)
        out["per_kernel_roofline_inflight1"] = {
            "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --inflight 1 --blocks 20",
            "frames_per_launch": fpl, "algorithmic_bytes_per_launch": alg,
            "kernels": {k: {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"])} for k, r in rows.items()},
            "trace_stage_ns_per_launch": pair,
            "achieved_gbs": alg / pair if pair else None, "frac_of_8_tbs": alg / pair / 8000.0 if pair else None,
            "bench_line": {k: b1[k] for k in ("value", "ms_per_step", "steps")},
            "note": "one launch in flight: trace_kernel then bounce_kernel, nothing else on the chip; achieved = algorithmic bytes of a launch / "
                    "(trace_kernel + bounce_kernel average duration).  The default schedule overlaps two launches and is faster per frame."}
    except Exception as e:  # noqa: BLE001
        out["per_kernel_roofline_inflight1"] = f"unreadable: {e}"

json.dump(out, open(f"{base}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
