#!/bin/bash
# round 5: the GPU suite twice more on one box (looking for cases that pass only most of the time)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5z; mkdir -p $O
for i in 1 2; do
  timeout -k 10 500 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests_$i.log 2>&1 || { tail -30 $O/gpu_tests_$i.log; exit 1; }
  tail -1 $O/gpu_tests_$i.log
done
