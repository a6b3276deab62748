#!/bin/bash
# Round 6, first GPU call: the new entry points under test (read-back, touch map, priority split, the device-built DDA grid), the whole
# GPU suite, then the two config-5 measurements VERDICT r5 item 1 asks for and one default bench line.
# usage (GPU box, repo root): scripts/r06/01_first_run.sh
set -o pipefail
O=gpurun_out/r6a
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_readback.py tests/test_gpu_touch_and_priority.py tests/test_gpu_dda_prototype.py "tests/test_gpu_pipeline.py::test_reference_loop_orbiting_camera_1080p_bit_exact" -x -q -m gpu > $O/new_tests.log 2>&1 || { tail -30 $O/new_tests.log; exit 1; }
tail -3 $O/new_tests.log
timeout -k 10 300 python -c "
import json, bench
print(json.dumps(bench.measure_config5_touch(0), indent=1))" > $O/config5_touch.json 2> $O/config5_touch.err || { tail -20 $O/config5_touch.err; exit 1; }
cat $O/config5_touch.json
timeout -k 10 600 python tests/diag_dda.py config5 2.0 > $O/dda_config5.txt 2> $O/dda_config5.err || { tail -20 $O/dda_config5.err; tail -20 $O/dda_config5.txt; exit 1; }
cat $O/dda_config5.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6a/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("config5_outside_view", "config5_tunnel_view"):
    print(k, json.dumps(d["extra"].get(k), indent=None)[:1500])
print("reference_loop", json.dumps(d["extra"].get("reference_loop"))[:3000])
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
