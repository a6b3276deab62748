#!/bin/bash
# Round 6: the root-exit shortcut (trace_common.h: VXRT_ROOT_EXIT — a pop back to a root with one occupied slot is a certain miss) A/B on
# one box: libvxrt.so (with it) against libvxrt_noexit.so (-DVXRT_ROOT_EXIT=0), the trace parity tests first.
set -o pipefail
O=gpurun_out/r6h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_degenerate.py tests/test_gpu_scenes.py tests/test_gpu_spirv_goldens.py tests/test_gpu_config5.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
BENCH_ARGS="--no-counters" bash scripts/ab_libs.sh $O/ab.txt noexit
BENCH_ARGS="--no-counters --view close" bash scripts/ab_libs.sh $O/ab_close.txt noexit
