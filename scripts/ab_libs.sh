#!/bin/bash
# usage (GPU box): scripts/ab_libs.sh <out file> <tag> [<tag> ...]   — bench.py (no extras) with libvxrt.so and each libvxrt_<tag>.so, alternating, 3 rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$1; shift
mkdir -p $(dirname $out)
for rep in 1 2 3; do
  for tag in default "$@"; do
    lib=$R/gpu_voxel_raytracer_amd/libvxrt.so
    [ "$tag" != default ] && lib=$R/gpu_voxel_raytracer_amd/libvxrt_$tag.so
    VXRT_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS} | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])" >> $out
  done
done
cat $out
