#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5i; mkdir -p $O; rm -f $O/prof.txt
timeout -k 10 200 python3 scripts/exp_fused_check.py > $O/fused_check.txt 2>&1; echo "rc $?" >> $O/fused_check.txt
head -6 $O/fused_check.txt; tail -1 $O/fused_check.txt
for cfg in "0 8 1 20" "4 8 1 20" "0 1 1 20" "0 1 3 8" "0 4 1 20"; do
  timeout -k 10 100 python3 scripts/exp_fused_one.py $cfg 1 100 >> $O/prof.txt 2>&1 || exit 1
done
cat $O/prof.txt
