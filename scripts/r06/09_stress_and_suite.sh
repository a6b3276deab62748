#!/bin/bash
# Round 6: the random-camera stress test at 20 x its default scale with a seed of its own (240 cameras x 3 whole 1080p frames against the
# oracle, every output value and ray count), then the whole GPU suite, on the round's final tree.
set -o pipefail
O=gpurun_out/r6s
mkdir -p $O
VXRT_STRESS_SCALE=20 VXRT_STRESS_SEED=26 timeout -k 10 900 python -m pytest tests/test_gpu_stress.py -x -q -m gpu > $O/gpu_stress_x20.log 2>&1 || { tail -30 $O/gpu_stress_x20.log; exit 1; }
tail -2 $O/gpu_stress_x20.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
