#!/bin/bash
# round 5: 8 ranks' 20-frame block: all-in-one kernel, later hand-over, more tail waves
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5c; mkdir -p $O
run() { echo "== $*" >> $O/deals.txt; env "$@" python3 scripts/exp_block_timeline.py $R 8 $I $B 20 200 >> $O/deals.txt || exit 1; }
for R in 0 4; do
  I=1; B=20
  run VXRT_TRACE_VARIANT=0
  run VXRT_TRACE_VARIANT=4 VXRT_TAIL_FROM=2
  run VXRT_TRACE_VARIANT=4 VXRT_TAIL_FROM=0
  run VXRT_TRACE_VARIANT=4 VXRT_TRACE_BLOCKS=1024
  run VXRT_TRACE_VARIANT=4 VXRT_TRACE_BLOCKS=4096
  run VXRT_TRACE_VARIANT=4 VXRT_FRAME_LANES=0
  run VXRT_TRACE_VARIANT=4 VXRT_SPREAD=0
  run VXRT_TRACE_VARIANT=4 VXRT_TILE_ORDER=0
  I=2; B=10
  run VXRT_TRACE_VARIANT=0
  I=3; B=7
  run VXRT_TRACE_VARIANT=0
  I=4; B=5
  run VXRT_TRACE_VARIANT=0
done
cat $O/deals.txt
