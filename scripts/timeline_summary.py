#!/usr/bin/env python3
"""Time line of the last blocks of scripts/exp_block_timeline.py from a rocprofv3 --kernel-trace CSV: for each block (kernels separated
by an idle gap > 60 us) the kernels in start order with start offset, duration and the gap to the previous kernel's end (us).
usage: timeline_summary.py <dir with *_kernel_trace.csv> [blocks to print]"""
import csv, glob, re, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []


def short(name):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", name)
    return m[0] if m else name[:40]


for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
blocks, cur, last_end = [], [], None
for s, e, n, q in rows:
    if last_end is not None and (s - last_end > 60_000 or (n == "trace_kernel" and cur and cur[-1][2] != "trace_kernel" and s - last_end > 15_000)):
        blocks.append(cur); cur = []
    cur.append((s, e, n, q))
    last_end = e if last_end is None else max(last_end, e)
blocks.append(cur)
spans = [(b[-1][1] if False else max(x[1] for x in b)) - b[0][0] for b in blocks]
print(f"{len(blocks)} blocks; GPU span of the last 20 (us): " + " ".join(f"{v / 1e3:.0f}" for v in spans[-20:]))
for b in blocks[-show:]:
    t0 = b[0][0]
    print(f"-- block: {len(b)} kernels, GPU span {(max(x[1] for x in b) - t0) / 1e3:.1f} us")
    prev_end = t0
    for s, e, n, q in b:
        print(f"   +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q}  {n}")
        prev_end = max(prev_end, e)
