#!/bin/bash
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5o; mkdir -p $O
run() { echo "== N=$N I=$I B=$B $*" >> $O/deals.txt; for R in $RANKS; do env "$@" python3 scripts/exp_block_timeline.py $R $N $I $B 20 150 >> $O/deals.txt || exit 1; done; }
N=8; RANKS="0 4"
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0
I=2; B=16; run BAND=4 VXRT_TRACE_VARIANT=0
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0 VXRT_FRAME_LANES=0
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0 VXRT_SPREAD=0
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0 VXRT_SPREAD=64
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0 VXRT_SKY_CULL=0
cat $O/deals.txt
