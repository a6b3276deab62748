#!/bin/bash
# Round 6: the XCD-affine launch order on config 5's scene with counters: does it cut what comes in from beyond the L2s?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
SIZES="0 2 8 32"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/r06/exp_config5_affinity.py $SIZES > $O/fetch.txt 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 $R/scripts/r06/exp_config5_affinity.py $SIZES > $O/tcc.txt 2> $O/tcc.err || { tail -5 $O/tcc.err; exit 1; }
cd $R
cat $O/fetch.txt
python3 scripts/r06/affinity_counters.py $O/fetch $SIZES | tee $O/summary_fetch.txt
python3 scripts/r06/affinity_counters.py $O/tcc $SIZES | tee $O/summary_tcc.txt
