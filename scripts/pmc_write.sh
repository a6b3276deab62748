#!/bin/bash
# usage: scripts/pmc_write.sh <tag> <bench args...>   (env VXRT_* is inherited by the profiled program) -> WRITE_SIZE per kernel launch
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pw_$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,re,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/pw_$tag/**/*counter_collection.csv",recursive=True)[0])):
    m=re.search(r"(trace_kernel|bounce_kernel|primary_kernel)",r["Kernel_Name"])
    if m: agg[m[0]].append(float(r["Counter_Value"])/1024)
for k,v in agg.items(): print("$tag", k, "launches", len(v), "WRITE MB first %.1f last %.1f" % (v[0], v[-1]))
PY
