#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -6 $O/gpu_tests.log
grep -q "pytest rc 0" $O/gpu_tests.log || exit 1
VXRT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --blocks 3 --band-rows 4 --tracer 1 > $O/bench_2ranks_gloo_deal8.json 2> $O/bench_2ranks.err; echo "2-rank rehearsal of the 8-rank deal rc $?"
python3 -c "import json; d=json.load(open('$O/bench_2ranks_gloo_deal8.json')); print(d['value'], d['config']['parallelism'], d['config']['tracer'], d['roofline']['kernel'])"
python3 -c "import __graft_entry__ as g; g.smoke()"
