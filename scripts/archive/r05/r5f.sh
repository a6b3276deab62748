#!/bin/bash
# round 5: fused head + tail (VXRT_OPT_FUSED_TAIL) — parity first (bounded), then the 20-frame block
cd $GRAFT_REPO_ROOT
export VXRT_ENV_KNOBS=1
O=$PWD/gpurun_out/r5f; mkdir -p $O
VXRT_FUSED_TAIL=1 timeout -k 10 420 python -m pytest tests/test_gpu_trace.py -x -q -m gpu > $O/parity.log 2>&1; echo "pytest rc $?" >> $O/parity.log
tail -5 $O/parity.log
grep -q "pytest rc 0" $O/parity.log || exit 1
run() { echo "== $*" >> $O/deals.txt; env "$@" timeout -k 10 120 python3 scripts/exp_block_timeline.py $R $N $I $B 20 200 >> $O/deals.txt || exit 1; }
N=8
for R in 0 4; do
  I=1; B=20
  run BAND=8 VXRT_FUSED_TAIL=1
  run BAND=4 VXRT_FUSED_TAIL=1
  run BAND=4 VXRT_TRACE_VARIANT=0
  run BAND=8
done
N=1; R=0
I=3; B=8; run BAND=8
I=3; B=8; run BAND=8 VXRT_FUSED_TAIL=1
I=1; B=20; run BAND=8 VXRT_FUSED_TAIL=1
I=2; B=10; run BAND=8 VXRT_FUSED_TAIL=1
N=4; R=0
I=2; B=16; run BAND=8
I=1; B=20; run BAND=8 VXRT_FUSED_TAIL=1
I=2; B=10; run BAND=8 VXRT_FUSED_TAIL=1
cat $O/deals.txt
