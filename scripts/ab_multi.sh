#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage: scripts/ab_multi.sh rounds "ENV1=a ENV2=b" "ENV1=c" ... [-- bench args] — bench.py alternating between environments
rounds=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $rounds); do
  for e in "${envs[@]}"; do
    env $e python bench.py --no-cpu-baseline --no-extras --blocks 20 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'])" || exit 1
  done
done
