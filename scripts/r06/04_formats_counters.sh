#!/bin/bash
# Round 6, VERDICT r5 item 1b: the three scene formats on the SAME rays of config 5's scene (tests/diag_dda.py config5: the 8-byte
# records' walk, the wide records' walk, the bricked DDA) with FETCH_SIZE / TCC hit / SQ counters per kernel and ray set.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6d
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
K='cast_probe_kernel|dda_probe_kernel'
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tests/diag_dda.py config5 2.0 > $O/fetch.txt 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
echo fetch done
timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 $R/tests/diag_dda.py config5 2.0 > $O/tcc.txt 2> $O/tcc.err || { tail -5 $O/tcc.err; exit 1; }
echo tcc done
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $R/tests/diag_dda.py config5 2.0 > $O/sq.txt 2> $O/sq.err || { tail -5 $O/sq.err; exit 1; }
echo sq done
cd $R
for p in fetch tcc sq; do echo "## pass $p"; python3 scripts/r06/kernel_counters.py $O/$p "$K"; done | tee $O/summary.txt
