#!/bin/bash
# Round 6, second GPU call: the XCD-affine launch order on config 5's scene — its parity case, then kernel times per
# super-tile size and view.   usage (GPU box, repo root): scripts/r06/02_affinity.sh
set -o pipefail
O=gpurun_out/r6b
mkdir -p $O
timeout -k 10 300 python -m pytest "tests/test_gpu_touch_and_priority.py::test_xcd_affine_tile_order_gives_identical_frames" -x -q -m gpu > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -2 $O/test.log
timeout -k 10 600 python scripts/r06/exp_config5_affinity.py > $O/affinity.txt 2> $O/affinity.err || { tail -20 $O/affinity.err; cat $O/affinity.txt; exit 1; }
cat $O/affinity.txt
