#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage (GPU box, repo root): scripts/profile_config5_order.sh <tag>
# BASELINE config 5's scene (2048^3 procedural Menger, 5.6 GB) at 3840x2160, 8 bounces, one frame at a time (scripts/exp_config5.py),
# with the records breadth-first (VXRT_NODE_ORDER=0) and with the last three node levels as depth-first treelets (VXRT_NODE_ORDER=1):
# kernel durations, HBM fetch bytes, L2 hit rate and the SQ counters of trace_kernel.  Separate rocprofv3 passes per counter set.
tag=${1:-order}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c5_$tag
mkdir -p $O
cd /tmp
for order in 0 2 3; do
  export VXRT_NODE_ORDER=$order
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/o${order}_stats -- python3 $R/scripts/exp_config5.py 2048 > $O/o${order}_stats.txt 2> $O/o${order}_stats.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/o${order}_fetch -- python3 $R/scripts/exp_config5.py 2048 > $O/o${order}_fetch.txt 2> $O/o${order}_fetch.err
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/o${order}_tcc -- python3 $R/scripts/exp_config5.py 2048 > $O/o${order}_tcc.txt 2> $O/o${order}_tcc.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/o${order}_sq -- python3 $R/scripts/exp_config5.py 2048 > $O/o${order}_sq.txt 2> $O/o${order}_sq.err
  echo "order $order done: $(grep ms/frame $O/o${order}_stats.txt | tr '\n' ' ')"
done
unset VXRT_NODE_ORDER
cd $R
python3 scripts/config5_summary.py $tag "o0=breadth-first records" "o2=treelets of the last two node levels" "o3=treelets of the last three node levels" > $O/summary_stdout.txt
# keep the csv volume small: the merged-back directory is capped
find $O -name "*.csv" -size +2M -delete
