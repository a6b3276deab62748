#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage: scripts/ab_env.sh VAR "v1 v2 ..." [rounds] [bench args...] — bench.py alternating between values of an environment variable
var=$1; vals=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do
  for v in $vals; do
    env $var=$v python bench.py --no-cpu-baseline --no-extras --blocks 30 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'])" || exit 1
  done
done
