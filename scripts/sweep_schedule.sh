#!/bin/bash
# usage: scripts/sweep_schedule.sh "batch:inflight ..." [rounds] [bench args] — bench.py over frames per launch x launches in flight, alternating
deals=$1; rounds=${2:-2}; shift 2
for i in $(seq $rounds); do
  for d in $deals; do
    b=${d%%:*}; f=${d##*:}
    python bench.py --no-cpu-baseline --no-extras --blocks 30 --batch $b --inflight $f "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['timing']['block_ms']; print('$b x $f', d['value'], d['ms_per_step'], t)" || exit 1
  done
done
