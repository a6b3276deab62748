#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# Collects what profiles/<round>/ holds for the default bench command:
#   1. rocprofv3 --kernel-trace --stats           -> per-kernel durations of the default run
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -> HBM bytes per launch
#   3. rocprofv3 --pmc SQ_* (one pass)            -> VALU instruction counts / lane utilisation per kernel
#   4. rocprofv3 --kernel-trace --stats with ONE launch in flight (--inflight 1) -> each kernel's duration alone on the chip: the
#      per-kernel roofline (algorithmic bytes of a launch / its average duration) can be recomputed from this CSV
#   5. the PMC passes again at the DRIVER's schedule (--steps 20 --warmup 5: 8 + 8 + 4 frames on three streams)
# Every rocprofv3 command has the program itself after "--" (no env / sh wrappers).
tag=${1:-run}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --blocks 20 > $O/stats1_bench.json 2> $O/stats1.err
PMC_ARGS="--steps 96 --warmup 32 --blocks 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $PMC_ARGS > $O/fetch_bench.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $PMC_ARGS > $O/write_bench.json 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/bench.py $PMC_ARGS > $O/sq_bench.json 2> $O/sq.err
DRV_ARGS="--steps 20 --warmup 5 --blocks 12 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_drv -- python3 $R/bench.py $DRV_ARGS > $O/fetch_drv_bench.json 2> $O/fetch_drv.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_drv -- python3 $R/bench.py $DRV_ARGS > $O/write_drv_bench.json 2> $O/write_drv.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $O/sq_drv -- python3 $R/bench.py $DRV_ARGS > $O/sq_drv_bench.json 2> $O/sq_drv.err
cd $R
python3 scripts/profile_summary.py $tag
