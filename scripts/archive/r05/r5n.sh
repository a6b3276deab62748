#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5n; mkdir -p $O
VXRT_BENCH_SPAWN_TIMEOUT=300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --blocks 3 > $O/bench_rccl_failure_fallback.json 2> $O/bench_rccl_failure_fallback.err; echo "fallback rc $?"
echo "warnings: $(grep -c 'destroy_process_group() was not called' $O/bench_rccl_failure_fallback.err)"
grep "bench.py rank" $O/bench_rccl_failure_fallback.err | cut -c1-160 | head -6
python3 -c "import json; d=json.load(open('$O/bench_rccl_failure_fallback.json')); print(d['value'], d['rccl']['backend'][:80])"
python3 scripts/nccl_sanity.py > $O/nccl_sanity.json 2> $O/nccl_sanity.err; echo "nccl sanity rc $?"; tail -c 400 $O/nccl_sanity.json
