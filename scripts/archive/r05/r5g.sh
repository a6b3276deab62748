#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5g; mkdir -p $O
timeout -k 10 240 python3 scripts/exp_fused_check.py > $O/fused_check.txt 2>&1; echo "rc $?" >> $O/fused_check.txt
cat $O/fused_check.txt
