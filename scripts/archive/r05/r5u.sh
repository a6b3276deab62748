#!/bin/bash
# round 5: the host's share of a rank of 8's short block
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5u; mkdir -p $O
timeout -k 10 300 python3 scripts/exp_block_host.py 4 300 > $O/block_host.txt 2>&1 || { tail -20 $O/block_host.txt; exit 1; }
cat $O/block_host.txt
