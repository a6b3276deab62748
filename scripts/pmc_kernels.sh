#!/bin/bash
# usage: scripts/pmc_kernels.sh <tag> [ENV=VAL ...]: per-kernel VALU instruction counts and lane utilisation (bench --inflight 1)
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pk_$tag -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 3 --no-cpu-baseline --inflight 1 ${BENCH_ARGS} > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob("gpurun_out/pk_$tag/*/*counter_collection.csv")[0])):
    m=re.search(r"(primary_kernel|trace_kernel|bounce_kernel|path_kernel|shade_kernel<\w+>|trace_rays_kernel|pool_rays_kernel|sun_kernel|resolve_kernel)",r["Kernel_Name"])
    if not m: continue
    agg[m[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    n=len(c["SQ_WAVES"]); per=lambda x: sum(c[x])/ (n/ (1 if k in("primary_kernel","trace_kernel","shade_kernel<true>") else 1))
    launches=n
    valu=sum(c["SQ_INSTS_VALU"]); act=sum(c["SQ_ACTIVE_INST_VALU"]); thr=sum(c["SQ_THREAD_CYCLES_VALU"])
    print(f"  {k:22s} launches {launches:3d}  VALU wave-instr/launch {valu/launches:.3e}  SALU {sum(c['SQ_INSTS_SALU'])/launches:.3e}  lane util {thr/(act*64)*100:5.1f} %  waitcnt share {sum(c['SQ_WAIT_ANY'])/sum(c['SQ_WAVE_CYCLES'])*100:4.0f} %")
PY
