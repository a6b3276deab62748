#!/bin/bash
# round 5: the suite and both bench lines on the final tree
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5t; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
python3 - <<'PY'
import json
for f in ("bench_default", "bench_driver"):
    d = json.loads(open(f"gpurun_out/r5t/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
