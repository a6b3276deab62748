#!/bin/bash
# Round 6: what profiles/r06/ holds of the default command on the round's tree — the two bench lines (default, the driver's --steps 20
# --warmup 5), the rocprofv3 stats / PMC passes of scripts/profile_round.sh, the GPU suite and smoke().
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6m
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
echo "default line done"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_schedule.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
echo "driver line done"
timeout -k 10 1100 bash scripts/profile_round.sh r06 > $O/profile_round.log 2>&1 || { tail -20 $O/profile_round.log; exit 1; }
echo "profile passes done"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
cat $O/smoke.log
python - <<'PY'
import json
for name in ("bench_default", "bench_driver_schedule"):
    d = json.loads([l for l in open(f"gpurun_out/r6m/{name}.json") if l.startswith("{")][-1])
    print(name, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["timing"].get("latency_ms_one_frame_at_a_time"))
PY
