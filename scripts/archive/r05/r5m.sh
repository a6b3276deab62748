#!/bin/bash
# round 5, closing run: the whole GPU suite, the bench lines, the RCCL fall-back's stderr
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -8 $O/gpu_tests.log
grep -q "pytest rc 0" $O/gpu_tests.log || exit 1
VXRT_BENCH_SPAWN_TIMEOUT=300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --blocks 3 > $O/bench_rccl_failure_fallback.json 2> $O/bench_rccl_failure_fallback.err; echo "fallback rc $?"
grep -c "destroy_process_group() was not called" $O/bench_rccl_failure_fallback.err
grep "bench.py rank" $O/bench_rccl_failure_fallback.err | head -6
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_schedule.json 2> $O/bench_driver.err || { tail -5 $O/bench_driver.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_driver_schedule.json')); print('driver schedule:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
