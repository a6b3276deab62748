#!/bin/bash
# usage: scripts/pmc_trace.sh <tag> [env assignments...]   -- collects two SQ counter passes for the trace kernel
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
cd $R
python3 scripts/pmc_summary.py $tag
