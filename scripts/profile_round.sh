#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# Collects what profiles/<round>/ holds for the default bench command:
#   1. rocprofv3 --kernel-trace --stats           -> per-kernel durations of the default run
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -> HBM bytes per launch
#   3. rocprofv3 --pmc SQ_* (one pass)            -> VALU instruction counts / lane utilisation per kernel
# Every rocprofv3 command has the program itself after "--" (no env / sh wrappers).
tag=${1:-run}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err
PMC_ARGS="--steps 96 --warmup 32 --blocks 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $PMC_ARGS > $O/fetch_bench.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $PMC_ARGS > $O/write_bench.json 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/bench.py $PMC_ARGS > $O/sq_bench.json 2> $O/sq.err
cd $R
python3 scripts/profile_summary.py $tag
