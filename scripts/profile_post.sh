#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_post.sh <tag>
# rocprofv3 evidence for the post stages (temporal_kernel, denoise_kernel / denoise_passthrough) of config 3 at 3840x2160:
#   1. --kernel-trace --stats            -> durations per kernel
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, kernel trace only) -> HBM bytes per launch
#   3. --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE ... -> instructions per tap of the two denoise kernels (fast and generic)
# The program itself stands after "--" (no env / sh wrappers).
tag=${1:-run}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/post_$tag
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/post_stage_run.py 8 > $O/stats_run.txt 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/post_stage_run.py 4 > $O/fetch_run.txt 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/post_stage_run.py 4 > $O/write_run.txt 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 $R/scripts/post_stage_run.py 4 > $O/sq_run.txt 2> $O/sq.err
cd $R
python3 scripts/post_summary.py $tag
