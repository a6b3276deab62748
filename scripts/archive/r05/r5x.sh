#!/bin/bash
# round 5: the bench line with its own parity check, and the contract test
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5x; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5x/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["extra"]["parity_check"])
PY
timeout -k 10 900 python3 -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "bench_contract" > $O/contract.log 2>&1 || { tail -30 $O/contract.log; exit 1; }
tail -3 $O/contract.log
