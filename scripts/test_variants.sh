#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# usage: scripts/test_variants.sh      (here: builds; on the GPU box: gpurun -- 'scripts/test_variants.sh run')
# The compile-time variants of the walk that are NOT in the default library stay bit-exact: this builds each
# of them as gpu_voxel_raytracer_amd/libvxrt_<tag>.so (hipcc cross-compiles without a GPU) and, with `run`, puts the trace, scene
# and degenerate-noise parity tests through it (VXRT_LIB selects the library).
set -e
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  # the schedules and the scene format that measured slower (tracers 2, 3, 5; the wide records): libvxrt_variants.so
  python -c "from gpu_voxel_raytracer_amd import _build; print(_build.build(variants=True, verbose=True))"
  scripts/ab_build.sh locate -DVXRT_LOCATE=1
  scripts/ab_build.sh defer1 -DVXRT_DEFER_SHADING=1
  scripts/ab_build.sh defer2 -DVXRT_DEFER_SHADING=2
  scripts/ab_build.sh w6 -DVXRT_TRACE_WAVES=6 -DVXRT_BOUNCE_WAVES=6 -DVXRT_SUN_SHORTCUT=0 -DVXRT_TAIL_REMAT=0
  exit 0
fi
echo "== variants (-DVXRT_VARIANTS=1: tracers 2, 3, 5 and the wide scene records)"
VXRT_LIB=$PWD/gpu_voxel_raytracer_amd/libvxrt_variants.so python -m pytest tests/test_gpu_trace.py tests/test_gpu_degenerate.py tests/test_gpu_config5.py -x -q -m gpu
[ "$2" = "variants-only" ] && exit 0
for tag in locate defer1 defer2 w6; do
  lib=$PWD/gpu_voxel_raytracer_amd/libvxrt_$tag.so
  [ -f "$lib" ] || { echo "build first: scripts/test_variants.sh"; exit 2; }
  echo "== $tag"
  VXRT_LIB=$lib python -m pytest tests/test_gpu_trace.py tests/test_gpu_scenes.py tests/test_gpu_degenerate.py -x -q -m gpu
done
