#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5h; mkdir -p $O; rm -f $O/summary.txt
export TMPDIR=/tmp
cd /tmp
for F in 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sq$F -- python3 $GRAFT_REPO_ROOT/scripts/exp_fused_one.py 0 8 1 20 $F 20 > $O/sq$F.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/pmc_fused_summary.py $O/sq$F >> $O/summary.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $O/mem$F -- python3 $GRAFT_REPO_ROOT/scripts/exp_fused_one.py 0 8 1 20 $F 20 > $O/mem$F.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/pmc_fused_summary.py $O/mem$F >> $O/summary.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl$F -- python3 $GRAFT_REPO_ROOT/scripts/exp_fused_one.py 0 8 1 20 $F 20 > $O/tl$F.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/timeline_summary.py $O/tl$F 2 >> $O/summary.txt
  rm -rf $O/sq$F $O/mem$F $O/tl$F
done
cat $O/summary.txt
