#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# rebuild libvxrt with different register budgets for the bounce kernel and bench each (GPU box)
for w in 4 5 6 8; do
  VXRT_HIPCC_FLAGS="-DVXRT_BOUNCE_WAVES=$w" python gpu_voxel_raytracer_amd/_build.py -f > /dev/null 2>&1
  for split in 0x1 0x3; do
    for view in bench close; do
      VXRT_TRACE_SPLIT=$split VXRT_TRACE_BLOCKS=4096 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --view $view | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w split=$split $view', d['value'], d['ms_per_step'])"
    done
  done
done
python gpu_voxel_raytracer_amd/_build.py -f > /dev/null 2>&1
