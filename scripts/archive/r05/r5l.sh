#!/bin/bash
# round 5: the profiler passes of the default bench command (scripts/profile_round.sh) and of the post stages (scripts/profile_post.sh)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 scripts/profile_round.sh r5 > gpurun_out/profile_round_r5.log 2>&1; echo "profile_round rc $?"
tail -3 gpurun_out/profile_round_r5.log
ls gpurun_out/prof_r5 | head -30
