#!/bin/bash
export VXRT_ENV_KNOBS=1   # host.py translates the VXRT_* knobs below into vxrt_create_tuned options (the library reads no environment)
# A/B of trace variants on the GPU box: VXRT_TRACE_VARIANT (0 monolithic, 2 wavefront), launch split mask, grid size.
for view in bench close; do
  echo "== view $view"
  VXRT_TRACE_VARIANT=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --view $view | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('monolithic', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
  for split in 0x1 0x3 0x5 0x7 0xf; do
   for blocks in 2048 4096; do
    VXRT_TRACE_VARIANT=2 VXRT_TRACE_SPLIT=$split VXRT_TRACE_BLOCKS=$blocks python bench.py --steps 100 --warmup 10 --no-cpu-baseline --view $view | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wavefront split=$split blocks=$blocks', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
   done
  done
done
