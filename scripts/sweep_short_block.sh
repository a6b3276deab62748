cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
for rep in 1 2; do
for cfg in "8 3" "4 3" "4 4" "4 5" "8 2" "12 2" "20 1" "8 4" "4 6"; do
  set -- $cfg
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --batch $1 --inflight $2 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 inflight $2', d['value'], d['ms_per_step'], d['timing']['block_ms'])" >> gpurun_out/r4i/sweep20.txt
done
done
cat gpurun_out/r4i/sweep20.txt
python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "falls_back or starts_its_own" 2>&1 | tail -3
