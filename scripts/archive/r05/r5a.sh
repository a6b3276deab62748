#!/bin/bash
# round 5, first GPU call: the suite (variants library loaded beside the product), the driver's 20-frame block for every rank of 8, time lines
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
scripts/short_block_all_ranks.sh $O/short_block.txt 8 "1 20" || exit 1
python3 scripts/exp_block_timeline.py 0 1 3 8 20 200 >> $O/short_block.txt || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl8 -- python3 $GRAFT_REPO_ROOT/scripts/exp_block_timeline.py 0 8 1 20 20 30 > $O/tl8.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/tl1 -- python3 $GRAFT_REPO_ROOT/scripts/exp_block_timeline.py 0 1 3 8 20 30 > $O/tl1.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 scripts/timeline_summary.py $O/tl8 2 > $O/tl8_summary.txt
python3 scripts/timeline_summary.py $O/tl1 2 > $O/tl1_summary.txt
cat $O/short_block.txt
rm -rf $O/tl8 $O/tl1
