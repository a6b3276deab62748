#!/bin/bash
# round 5: the 20-frame block per rank for 2, 4 and 8 ranks: two kernels / all-in-one, 8- and 4-row bands (spin-wait sync in)
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5j; mkdir -p $O
run() { echo "== N=$N I=$I B=$B $*" >> $O/deals.txt; for R in $RANKS; do env "$@" python3 scripts/exp_block_timeline.py $R $N $I $B 20 150 >> $O/deals.txt || exit 1; done; }
N=1; RANKS="0"; I=3; B=8; run BAND=8
N=1; RANKS="0"; I=3; B=8; run BAND=8 VXRT_TRACE_VARIANT=0
N=2; RANKS="0 1"
I=3; B=8; run BAND=8
I=3; B=8; run BAND=4
I=2; B=10; run BAND=4 VXRT_TRACE_VARIANT=0
I=3; B=8; run BAND=4 VXRT_TRACE_VARIANT=0
N=4; RANKS="0 1 2 3"
I=2; B=16; run BAND=8
I=2; B=16; run BAND=4
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0
I=2; B=10; run BAND=4 VXRT_TRACE_VARIANT=0
N=8; RANKS="0 1 2 3 4 5 6 7"
I=1; B=20; run BAND=4 VXRT_TRACE_VARIANT=0
I=2; B=10; run BAND=4 VXRT_TRACE_VARIANT=0
I=1; B=20; run BAND=8
cat $O/deals.txt
