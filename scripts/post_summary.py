#!/usr/bin/env python3
"""gpurun_out/post_<tag>/ (scripts/profile_post.sh) -> summary.json: per post-stage kernel and (radius, mode) its average duration from
the kernel trace and its HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (read side doubled per MI355X_MICROARCH.md's
gfx950 note; both figures kept), with the algorithmic bytes of SURVEY.md 8d beside them."""
import collections
import csv
import glob
import json
import re
import sys

tag = sys.argv[1]
base = f"gpurun_out/post_{tag}"
W, H = 3840, 2160
px = W * H
N_STATS, N_PMC = 8, 4


def launches(sub, counter=None):
    """kernel launches of the post stages in submission order: (name, duration_ns or counter value)"""
    out = []
    pat = f"{base}/{sub}/**/*counter_collection.csv" if counter else f"{base}/{sub}/**/*kernel_trace.csv"
    for f in glob.glob(pat, recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            m = re.search(r"(temporal_kernel|denoise_pair_kernel|denoise_generic_kernel|denoise_passthrough_kernel)", r["Kernel_Name"])
            if not m:
                continue
            if counter:
                if r["Counter_Name"] == counter:
                    out.append((m[0], float(r["Counter_Value"])))
            else:
                out.append((m[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def split(seq, n):
    """scripts/post_stage_run.py's launch order -> {label: [values]} (warm-up launches dropped)"""
    groups = collections.defaultdict(list)
    temporal = [v for k, v in seq if k == "temporal_kernel"]
    den = [(k, v) for k, v in seq if k in ("denoise_pair_kernel", "denoise_generic_kernel")]
    # temporal: 3 radii x (2 warm-up + n)
    for i, r in enumerate((0, 2, 8)):
        groups[f"temporal_kernel (r={r}{', denoise fused' if r == 0 else ''})"] = temporal[i * (2 + n) + 2:(i + 1) * (2 + n)]
    # denoise: per radius in (2, 8): 2 warm-up + n in-frame (exact, fast kernel), then 1 + n each of: exact, tolerant (denoise_pair_kernel: two
    # outputs per lane), exact generic, tolerant generic (denoise_generic_kernel: round 2's kernel, one output per lane, full formula)
    per = 2 + n + 4 * (1 + n)
    for i, r in enumerate((2, 8)):
        blk = den[i * per:(i + 1) * per]
        for j, label in enumerate(("exact", "tolerant", "exact, generic kernel", "tolerant, generic kernel")):
            first = 2 + n + j * (1 + n) + 1
            part = blk[first:first + n]
            want = "denoise_generic_kernel" if "generic" in label else "denoise_pair_kernel"
            assert all(k == want for k, _ in part), (label, [k for k, _ in part])
            groups[f"denoise r={r} {label}"] = [v for _, v in part]
    return groups


out = {"workload": f"vox/monu10.vox {W}x{H}, 8 bounces, temporal + denoise (BASELINE configs[2]); scripts/post_stage_run.py", "kernels": {}}
dur = split(launches("stats"), N_STATS)
fetch = split(launches("fetch", "FETCH_SIZE"), N_PMC)
write = split(launches("write", "WRITE_SIZE"), N_PMC)
sq = {c: split(launches("sq", c), N_PMC) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES")}
for label, d in dur.items():
    if not d:
        continue
    ms = sum(d) / len(d) / 1e6
    alg = (80 + (64 if "fused" in label else 0) - (16 + 16 if "fused" in label else 0)) * px if label.startswith("temporal") else 64 * px
    k = {"avg_ms": round(ms, 4), "launches": len(d), "algorithmic_bytes": alg, "algorithmic_gbs": round(alg / (ms * 1e-3) / 1e9, 1)}
    f, w = fetch.get(label), write.get(label)
    if f and w:
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        k.update({"fetch_size_kb": fk, "write_size_kb": wk, "hbm_bytes_raw": (fk + wk) * 1024, "hbm_bytes_read_doubled": (2 * fk + wk) * 1024,
                  "hbm_gbs_raw": round((fk + wk) * 1024 / (ms * 1e-3) / 1e9, 1), "hbm_gbs_read_doubled": round((2 * fk + wk) * 1024 / (ms * 1e-3) / 1e9, 1),
                  "frac_of_8TBs_read_doubled": round((2 * fk + wk) * 1024 / (ms * 1e-3) / 8e12, 4)})
    if label.startswith("denoise r="):
        r = int(re.search(r"r=(\d+)", label)[1])
        taps = (2 * r + 1) ** 2 * px
        k["taps_per_s"] = round(taps / (ms * 1e-3), -9)
        for c, g in sq.items():
            if g.get(label):
                k[c.lower() + "_per_launch"] = sum(g[label]) / len(g[label])
        if "sq_insts_valu_per_launch" in k:
            k["valu_lane_instr_per_tap"] = round(k["sq_insts_valu_per_launch"] * 64 / taps, 2)
            k["lds_wave_instr_per_1000_taps"] = round(k.get("sq_insts_lds_per_launch", 0) * 1000 / taps, 3)
    out["kernels"][label] = k
out["note"] = ("temporal with radius 0 also does the denoise stage's work (mix with the albedo) in the same pass: 64 B read (sampled colour, new "
               "normal/depth, old colour, old normal/depth) + 16 B albedo read + 2 x 16 B written = 112 B/px algorithmic; sky pixels skip the "
               "history reads, so counter bytes sit below that.  FETCH_SIZE on gfx950 counts wide coalesced reads at half their bytes "
               "(MI355X_MICROARCH.md), hence the read-doubled figure; the truth lies between the two.")
json.dump(out, open(f"{base}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
