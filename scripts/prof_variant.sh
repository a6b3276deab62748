#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage: scripts/prof_variant.sh <tag> [ENV=VAL ...] -- per-kernel average times (us) of bench.py --inflight 1 under rocprofv3
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --inflight 1 ${BENCH_ARGS} > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,collections,re
rows=list(csv.DictReader(open(glob.glob("gpurun_out/kt_$tag/*/*kernel_trace.csv")[0])))
rows=[r for r in rows if "vxrt" in r["Kernel_Name"] and "noise" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find period: sequence of kernel names between primary kernels
names=[(re.search(r"(primary_kernel|trace_kernel|tile_order_kernel|bounce_kernel|shade_kernel<\w+>|trace_rays_kernel)",r["Kernel_Name"]) or [r["Kernel_Name"]])[0] for r in rows]
starts=[i for i,n in enumerate(names) if n.startswith("primary") or n.startswith("trace_kernel")]
per=starts[1]-starts[0] if len(starts)>1 else len(rows)
acc=collections.defaultdict(list); gaps=collections.defaultdict(list)
for si in starts[5:-1]:
    for k in range(per):
        r=rows[si+k]; acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
        if k>0: gaps[k].append((int(r["Start_Timestamp"])-int(rows[si+k-1]["End_Timestamp"]))/1e3)
tot=0
for k in range(per):
    d=sum(acc[k])/len(acc[k]); g=sum(gaps[k])/len(gaps[k]) if k in gaps else 0; tot+=d+g
    print(f"  {k:2d} {names[starts[5]+k]:28s} {d:8.1f} us  (gap before {g:5.1f})")
print(f"  frame total {tot:.1f} us")
PY
