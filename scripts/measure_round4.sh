#!/bin/bash
# usage (GPU box, repo root): scripts/measure_round4.sh   -> gpurun_out/r4g/*: the numbers profiles/r04/ keeps (each step appends a progress line)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a $O/progress.log; }
say "bench default"; python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
say "bench driver schedule"; python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_schedule.json 2> $O/bench_driver_schedule.err
say "bench --gpus 2 over gloo (self-spawned)"; VXRT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_selfspawn.json 2> $O/bench_2ranks.err
say "bench --pipeline --gpus 2 over gloo (self-spawned)"; VXRT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --pipeline --steps 12 --warmup 3 > $O/pipeline_2ranks_gloo.json 2> $O/pipeline_2ranks.err
say "bench --pipeline, 1 rank"; python3 bench.py --pipeline --steps 12 --warmup 3 > $O/pipeline_1rank.json 2> $O/pipeline_1rank.err
say "early fetch A/B (config 5 scene)"
export VXRT_ENV_KNOBS=1
for rep in 1 2; do
  python3 scripts/exp_config5.py 2048 >> $O/early_fetch_default.txt 2>&1
  VXRT_LIB=$R/gpu_voxel_raytracer_amd/libvxrt_early.so python3 scripts/exp_config5.py 2048 >> $O/early_fetch_early.txt 2>&1
done
say "early fetch A/B (bench)"
for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --no-extras | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("default", d["value"], d["ms_per_step"])' >> $O/early_fetch_bench.txt
  VXRT_LIB=$R/gpu_voxel_raytracer_amd/libvxrt_early.so python3 bench.py --no-cpu-baseline --no-extras | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("early", d["value"], d["ms_per_step"])' >> $O/early_fetch_bench.txt
done
say "vxrt_multi (copy transport) 2 and 4 ranks at 4K"
for n in 1 2 4; do
  ./gpu_voxel_raytracer_amd/vxrt_multi menger:4 3840 2160 8 8 8 $O/multi_$n.ppm --ranks $n --transport copy --spp 4 --check >> $O/vxrt_multi.txt 2>&1
done
rm -f $O/multi_*.ppm
say "baseline configs (band balance of config 4)"; python3 scripts/exp_baseline_configs.py > $O/baseline_configs.txt 2>&1
say "profile round"; bash scripts/profile_round.sh r4 > $O/profile_round.log 2>&1
find $R/gpurun_out/prof_r4 -name "*.csv" -size +3M -delete
say "done"
