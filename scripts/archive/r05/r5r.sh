#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5r; mkdir -p $O
BENCH_ARGS="--no-counters" scripts/ab_libs.sh $O/ab_shard.txt shard1 shard2
