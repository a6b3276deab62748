#!/bin/bash
# Round 6, VERDICT r5 item 4a: DESIGN section 8.4's untried hardware-side candidate for the driver's 20-frame block on 8 ranks — the tiles
# that walk as a grid on a HIGH-priority stream, the tiles of sky on a low-priority one (VXRT_OPT_TRACE_PRIORITY) — for EVERY rank's band
# set, each alone on the GPU (emulated), against the same deal without it.  Stop rule: the slowest rank <= 0.39 ms, or leave the deal alone.
export VXRT_ENV_KNOBS=1 GPU_MAX_HW_QUEUES=8
# (the option lost and lives in the -DVXRT_VARIANTS=1 library since: both arms run in it; the recorded run had it in the product library)
export VXRT_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/gpu_voxel_raytracer_amd/libvxrt_variants.so
O=gpurun_out/r6f
mkdir -p $O
out=$O/short_block.txt
: > $out
for cfg in "BAND=4 VXRT_TRACE_VARIANT=0" "BAND=4" ; do
  for prio in 0 1; do
    echo "== N=8 I=1 B=20 $cfg VXRT_TRACE_PRIORITY=$prio" | tee -a $out
    for r in 0 1 2 3 4 5 6 7; do
      env $cfg VXRT_TRACE_PRIORITY=$prio timeout -k 10 120 python3 scripts/exp_block_timeline.py $r 8 1 20 20 200 | tee -a $out || exit 1
    done
  done
done
