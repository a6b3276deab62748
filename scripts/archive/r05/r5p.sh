#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5p; mkdir -p $O
for s in menger castle room monu10; do
  timeout -k 10 300 python3 tests/diag_dda.py $s 2.0 > $O/dda_$s.txt 2>&1; echo "rc $?" >> $O/dda_$s.txt
  tail -4 $O/dda_$s.txt
done
timeout -k 10 300 python3 tests/diag_dda.py menger 1.0 > $O/dda_menger_margin1.txt 2>&1; tail -2 $O/dda_menger_margin1.txt
