#!/bin/bash
# round 5: the whole GPU suite (variants library loaded beside the product), then the bench lines
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
grep -q "pytest rc 0" $O/gpu_tests.log || exit 1
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_schedule.json 2> $O/bench_driver.err || { tail -5 $O/bench_driver.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_driver_schedule.json')); print('driver schedule:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
