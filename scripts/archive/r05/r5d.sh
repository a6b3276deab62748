#!/bin/bash
# round 5: the longest tiles as an all-in-one grid of their own (VXRT_OPT_LONG_TILES) — parity, then the 20-frame block at 8 ranks
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "every_trace_variant or frames_per_launch or in_flight" > $O/parity.log 2>&1; echo "pytest rc $?" >> $O/parity.log
tail -3 $O/parity.log
grep -q "pytest rc 0" $O/parity.log || exit 1
run() { echo "== $*" >> $O/deals.txt; env "$@" python3 scripts/exp_block_timeline.py $R 8 $I $B 20 200 >> $O/deals.txt || exit 1; }
for R in 0 4; do
  I=1; B=20
  run VXRT_LONG_TILES=0
  for pm in 3 6 12 25 50 100 200; do
    run VXRT_LONG_TILES=$pm VXRT_SPREAD=0
  done
  run VXRT_TRACE_VARIANT=0
done
cat $O/deals.txt
export TMPDIR=/tmp
cd /tmp
VXRT_LONG_TILES=25 VXRT_SPREAD=0 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $GRAFT_REPO_ROOT/scripts/exp_block_timeline.py 4 8 1 20 20 30 > $O/tl.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/timeline_summary.py $O/tl 3 > $O/tl_summary.txt
rm -rf $O/tl
