#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage (GPU box, repo root): scripts/profile_config5.sh <tag>
# BASELINE config 5's scene (2048^3 procedural Menger, 5.6 GB) at 3840x2160, 8 bounces, one frame at a time (scripts/exp_config5.py),
# with the 8-byte scene records (VXRT_WIDE=0) and the wide ones (VXRT_WIDE=1): kernel durations, HBM fetch bytes, L2 hit rate and
# the SQ instruction counters of trace_kernel.  Separate rocprofv3 passes per counter set; the program itself stands after "--".
tag=${1:-run}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c5_$tag
mkdir -p $O
cd /tmp
for wide in 0 1; do
  export VXRT_WIDE=$wide
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/w${wide}_stats -- python3 $R/scripts/exp_config5.py 2048 > $O/w${wide}_stats.txt 2> $O/w${wide}_stats.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/w${wide}_fetch -- python3 $R/scripts/exp_config5.py 2048 > $O/w${wide}_fetch.txt 2> $O/w${wide}_fetch.err
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/w${wide}_tcc -- python3 $R/scripts/exp_config5.py 2048 > $O/w${wide}_tcc.txt 2> $O/w${wide}_tcc.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/w${wide}_sq -- python3 $R/scripts/exp_config5.py 2048 > $O/w${wide}_sq.txt 2> $O/w${wide}_sq.err
done
unset VXRT_WIDE
cd $R
python3 scripts/config5_summary.py $tag
