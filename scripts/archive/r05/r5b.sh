#!/bin/bash
# round 5: the driver's 20-frame block at 8 ranks — per-rank time lines (kernel names), head stagger A/B, deals
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5b; mkdir -p $O
for r in 0 4; do
  for deal in "1 20" ; do
    set -- $deal
    python3 scripts/exp_block_timeline.py $r 8 $1 $2 20 200 >> $O/deals.txt || exit 1
  done
  for deal in "2 10" "3 7" "4 5" "2 12" "3 8"; do
    set -- $deal
    VXRT_HEAD_STAGGER=0 python3 scripts/exp_block_timeline.py $r 8 $1 $2 20 200 >> $O/deals.txt || exit 1
    echo "   ^ stagger 0" >> $O/deals.txt
    VXRT_HEAD_STAGGER=1 python3 scripts/exp_block_timeline.py $r 8 $1 $2 20 200 >> $O/deals.txt || exit 1
    echo "   ^ stagger 1" >> $O/deals.txt
  done
done
cat $O/deals.txt
export TMPDIR=/tmp
cd /tmp
for r in 0 4; do
rocprofv3 --kernel-trace --output-format csv -d $O/tl8_$r -- python3 $GRAFT_REPO_ROOT/scripts/exp_block_timeline.py $r 8 1 20 20 30 > $O/tl8_$r.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/timeline_summary.py $O/tl8_$r 3 > $O/tl8_${r}_summary.txt
rm -rf $O/tl8_$r
done
VXRT_HEAD_STAGGER=1 rocprofv3 --kernel-trace --output-format csv -d $O/tl8s -- python3 $GRAFT_REPO_ROOT/scripts/exp_block_timeline.py 4 8 2 10 20 30 > $O/tl8s.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/timeline_summary.py $O/tl8s 3 > $O/tl8s_summary.txt
rm -rf $O/tl8s
