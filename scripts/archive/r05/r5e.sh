#!/bin/bash
# round 5: finer band interleave for the trace-only bench at 8 ranks (2- and 4-row bands), both tracers, every rank
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5e; mkdir -p $O
for B in 2 4 8; do
 for V in 4 0; do
  echo "== band $B tracer $V" >> $O/deals.txt
  for R in 0 1 2 3 4 5 6 7; do
    BAND=$B VXRT_TRACE_VARIANT=$V python3 scripts/exp_block_timeline.py $R 8 1 20 20 150 >> $O/deals.txt || exit 1
  done
 done
done
cat $O/deals.txt
