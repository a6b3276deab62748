#!/bin/bash
# round 5: steady state per rank (960-frame blocks' schedule), 8- against 4-row bands, emulated
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5s; mkdir -p $O
RANKS=1 INFL=2x16 BAND=8 python3 scripts/exp_rank_emulation.py >> $O/rank_emulation.txt || exit 1
for B in 8 4; do
  RANKS=2,4,8 INFL=3x16 BAND=$B python3 scripts/exp_rank_emulation.py >> $O/rank_emulation.txt || exit 1
done
cat $O/rank_emulation.txt
