#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pipeline_rehearsal.sh <tag> [ranks...]
# bench.py --pipeline (BASELINE configs[3]: castle 3840x2160, 4 spp, 8 bounces, temporal + denoise r = 8 with the halo exchange) with
# several ranks in separate processes that SHARE this box's one GPU, over gloo (the messages are staged through pinned host memory —
# on a node they travel GPU to GPU over RCCL).  A rehearsal of the frame loop, NOT a scaling measurement: the ranks take turns on one
# GPU.  What it does measure: halo bytes per rank, the pack / unpack kernels, the denoise stage split into interior and edge tiles, and
# the cost of the exchange done synchronously against the overlapped loop.  One JSON line per rank count -> gpurun_out/pipeline_<tag>/.
tag=${1:-run}; shift
ranks=${@:-"2 3"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pipeline_$tag
mkdir -p $O
python3 $R/bench.py --pipeline --steps 12 --warmup 3 > $O/pipeline_1rank.json 2> $O/pipeline_1rank.err
port=29610
for n in $ranks; do
  VXRT_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      $R/bench.py --gpus $n --pipeline --steps 12 --warmup 3 > $O/pipeline_${n}ranks_gloo.json 2> $O/pipeline_${n}ranks_gloo.err || exit 1
  port=$((port + 1))
done
# the same with the 16-row bands of round 2 (every row of a rank is a neighbour's halo), for comparison
VXRT_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    $R/bench.py --gpus 2 --pipeline --steps 12 --warmup 3 --band-rows 16 > $O/pipeline_2ranks_gloo_band16.json 2> $O/pipeline_2ranks_gloo_band16.err
grep -h '^{' $O/*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    h = d.get('halo', {})
    print(d['n_gpus'], 'ranks, band', h.get('band_rows'), ': ms/frame', d['ms_per_step'], 'sync', h.get('ms_per_step_synchronous'), 'halo MB/rank', round(h.get('bytes_per_rank_per_frame', 0) / 1e6, 2),
          'pack', h.get('pack_ms'), 'unpack', h.get('unpack_ms'), 'exchange(sync)', h.get('exchange_ms_synchronous'), 'stages', d['stage_ms_per_frame'])
"
