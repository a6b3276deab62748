#!/bin/bash
# round 5: the HIP path against the outputs of the reference's compiled shaders (tests/golden/spirv_exec), smoke, and the oracle's side on this box
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5v; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_spirv_goldens.py -x -q -m gpu -rs > $O/gpu_spirv_goldens.log 2>&1 || { tail -40 $O/gpu_spirv_goldens.log; exit 1; }
tail -3 $O/gpu_spirv_goldens.log
timeout -k 10 600 python3 -m pytest tests/test_oracle_spirv_exec.py tests/test_oracle_spirv_pin.py -x -q -rs > $O/cpu_spirv_on_gpu_box.log 2>&1 || { tail -40 $O/cpu_spirv_on_gpu_box.log; exit 1; }
tail -12 $O/cpu_spirv_on_gpu_box.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
cat $O/smoke.log
