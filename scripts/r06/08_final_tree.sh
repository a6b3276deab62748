#!/bin/bash
# Round 6, last call: the two bench lines, the GPU suite and smoke() on the round's final tree.
set -o pipefail
O=gpurun_out/r6z
mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
echo "default line done"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_schedule.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
echo "driver line done"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
cat $O/smoke.log
python - <<'PY'
import json
for name in ("bench_default", "bench_driver_schedule"):
    d = json.loads([l for l in open(f"gpurun_out/r6z/{name}.json") if l.startswith("{")][-1])
    print(name, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["timing"].get("latency_ms_one_frame_at_a_time"))
    for k, r in d["extra"]["reference_loop"]["rows"].items():
        print("  ", k, r["ms_per_frame"], r["with_vxrt_read_ms_per_frame"], r["with_vxrt_read_async_ms_per_frame"], r["transfer_alone_ms"], r["read_async_over_max_of_render_and_transfer"], r["vxrt_render_path_ms_per_frame"])
PY
