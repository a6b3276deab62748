"""GPU: the multi-rank frame loop as it will run on a node — one PROCESS per rank, torch.distributed between them — rehearsed
with ranks that share this box's one GPU over gloo (the halo is staged through host memory; on a node it travels GPU to GPU
over RCCL, the only line that differs: distributed.HaloExchange(comm_device=...)).  No multi-GPU node was available to any
round so far: nothing here is a scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_gpu_pipeline import free_port

pytestmark = pytest.mark.gpu


def launch(nproc, script_args, env=None, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", free_port()] + script_args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc,radius", [(2, 3), (3, 8), (2, 0)])
def test_ranks_in_processes_stitch_to_the_single_context_frame(nproc, radius):
    """Camera at rest and a slow drift: the stitched frames equal the single-context frames bit for bit (sampled, accumulated and
    denoised) — also with radius 0, the reference's default, where the halo carries one history row per band edge for the temporal
    stage alone.  A fast pan: a reprojection that leaves a rank's rows + halo is treated as a disocclusion (the reference's rule
    for off-screen reprojection, temporal.comp:92) — the accumulated image differs from the single-context one there, and ONLY
    in that way; the sampled colours never differ.  The frame loop is distributed.render_frame: pack -> messages -> denoise of the
    interior tiles -> unpack -> denoise of the edge tiles, ordered by events."""
    d = launch(nproc, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")], env={"VXRT_TEST_RADIUS": str(radius)})
    for name in ("rest", "slow"):
        r = d[name]
        assert r["rays_equal"] and r["geometry_pixels"] > 10000
        assert r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)
    f = d["fast"]
    assert f["rays_equal"] and f["sampled_differing_pixels"] == 0
    assert f["accum_differs_only_where_treated_as_disocclusion"], f
    assert 0 < f["accum_differing_pixels"] < f["geometry_pixels"]      # a 17-row jump against 3 / 8 / 1 halo rows: most of the history is out of reach


def test_history_halo_sized_for_the_motion_keeps_the_fast_pan_exact():
    """VXRT_OPT_HALO_ROWS = the band height (32 rows, 2 ranks): every row of the neighbours' history travels, so even the fast pan
    (about 17 rows of image motion in one frame) reprojects into rows the rank can see — all three images equal the single context's."""
    d = launch(2, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")],
               env={"VXRT_TEST_RADIUS": "2", "VXRT_TEST_BAND": "32", "VXRT_TEST_HALO_ROWS": "32"})
    for name in ("rest", "slow", "fast"):
        r = d[name]
        assert r["rays_equal"] and r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)


def test_pipeline_bench_two_ranks_on_one_gpu():
    """bench.py --pipeline (BASELINE configs[3]: castle 3840x2160, 4 spp, temporal + denoise r = 8, halo exchange) as the driver
    would launch it on a node, rehearsed over gloo: one JSON line with the halo bytes and the pack / exchange / unpack times apart."""
    d = launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pipeline", "--steps", "4", "--warmup", "1"],
               env={"VXRT_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and "castle" in d["config"]["workload"] and "x2" in d["config"]["parallelism"]
    hl = d["halo"]
    # 64-row bands (>= 8 r), 34 of them over 2 ranks: two messages of 17 slots x 8 rows x 3840 px x 36 B per rank and frame
    assert hl["band_rows"] == 64 and hl["rows"] == 8 and hl["bytes_per_pixel"] == 36
    assert 2 * 17 * 8 * 3840 * 36 <= hl["bytes_per_rank_per_frame"] <= 2 * 17 * 8 * 3840 * 36 + 512
    assert hl["pack_ms"] > 0 and hl["unpack_ms"] > 0 and hl["exchange_ms_synchronous"] > 0 and d["value"] > 100.0
    s = d["stage_ms_per_frame"]
    assert s["trace"] > 0 and s["temporal"] > 0 and s["denoise"] > 0
