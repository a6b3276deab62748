#!/bin/bash
# round 5: the launcher after its exit-order change: bench.py --gpus 2 on one GPU (RCCL refuses, the ranks agree on gloo), and the distributed GPU tests
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5y; mkdir -p $O
timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks.json 2> $O/bench_2ranks.err || { tail -30 $O/bench_2ranks.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5y/bench_2ranks.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["rccl"]["backend"], d["rccl"]["distinct_devices"])
PY
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu -rs > $O/gpu_distributed.log 2>&1 || { tail -30 $O/gpu_distributed.log; exit 1; }
tail -12 $O/gpu_distributed.log
