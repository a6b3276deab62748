#!/bin/bash
# round 5: the random-camera stress at twenty times its usual length (720 full 1080p frames from 240 random cameras against the oracle)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5w; mkdir -p $O
VXRT_STRESS_SCALE=20 VXRT_STRESS_SEED=20 timeout -k 10 1100 python3 -m pytest tests/test_gpu_stress.py -x -q -m gpu --durations=3 > $O/stress_x20.log 2>&1 || { tail -30 $O/stress_x20.log; exit 1; }
tail -8 $O/stress_x20.log
