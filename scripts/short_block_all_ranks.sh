#!/bin/bash
# usage (GPU box): scripts/short_block_all_ranks.sh <outfile> [nranks] ["infl batch" deals...]
# The driver's 20-frame block for EVERY rank's band set of the trace-only bench, each alone on the GPU.
out=$1; nranks=${2:-8}; shift; shift
deals=("$@"); [ ${#deals[@]} -eq 0 ] && deals=("1 20")
for d in "${deals[@]}"; do
  set -- $d
  for r in $(seq 0 $((nranks - 1))); do
    python3 scripts/exp_block_timeline.py $r $nranks $1 $2 20 200 >> $out || exit 1
  done
done
