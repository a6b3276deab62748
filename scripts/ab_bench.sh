#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage: scripts/ab_bench.sh "<label>:<env assignments>" ...   -- bench.py (no CPU baseline) once per arm, alternating twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for arm in "$@"; do
    label=${arm%%:*}; envs=${arm#*:}
    out=$(env $envs python3 $R/bench.py --no-cpu-baseline --steps ${STEPS:-1000} ${BENCH_ARGS} 2>&1 | tail -1)
    echo "$label rep$rep $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s", d["ms_per_step"], "ms")' 2>/dev/null || echo "FAILED: $out")"
  done
done
