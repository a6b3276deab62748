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


@pytest.mark.parametrize("nproc,radius", [(2, 3), (3, 8)])
def test_ranks_in_processes_stitch_to_the_single_context_frame(nproc, radius):
    """Camera at rest and a slow drift: the stitched frames equal the single-context frames bit for bit (sampled, accumulated and
    denoised).  A fast pan: a reprojection that leaves a rank's rows + halo is treated as a disocclusion (the reference's rule for
    off-screen reprojection, temporal.comp:92) — the accumulated image differs from the single-context one there, and ONLY in
    that way; the sampled colours never differ."""
    d = launch(nproc, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")], env={"VXRT_TEST_RADIUS": str(radius)})
    for name in ("rest", "slow"):
        r = d[name]
        assert r["rays_equal"] and r["geometry_pixels"] > 10000
        assert r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)
    f = d["fast"]
    assert f["rays_equal"] and f["sampled_differing_pixels"] == 0
    assert f["accum_differs_only_where_treated_as_disocclusion"], f
    assert f["accum_differing_pixels"] < 0.5 * f["geometry_pixels"]


def test_pipeline_bench_two_ranks_on_one_gpu():
    """bench.py --pipeline (BASELINE configs[3]: castle 3840x2160, 4 spp, temporal + denoise r = 8, halo exchange) as the driver
    would launch it on a node, rehearsed over gloo: one JSON line with the halo bytes and the exchange time apart."""
    d = launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pipeline", "--steps", "4", "--warmup", "1"],
               env={"VXRT_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and "castle" in d["config"]["workload"] and "x2" in d["config"]["parallelism"]
    # per rank and frame: two messages of max_bands x r rows x 3 images x width float4
    assert d["halo"]["bytes_per_rank_per_frame"] == 2 * 68 * 8 * 3 * 3840 * 16
    assert d["halo"]["exchange_ms_per_frame"] > 0 and d["value"] > 100.0
    s = d["stage_ms_per_frame"]
    assert s["trace"] > 0 and s["temporal"] > 0 and s["denoise"] > 0
