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
    # 64-row bands (>= 8 r): 30 of them and a last round of 128 + 112 rows over 2 ranks: two messages of 16 slots x 8 rows x 3840 px x 36 B per rank and frame
    assert hl["band_rows"] == 64 and hl["rows"] == 8 and hl["bytes_per_pixel"] == 36
    assert 2 * 16 * 8 * 3840 * 36 <= hl["bytes_per_rank_per_frame"] <= 2 * 16 * 8 * 3840 * 36 + 512
    assert hl["pack_ms"] > 0 and hl["unpack_ms"] > 0 and hl["exchange_ms_synchronous"] > 0 and d["value"] > 100.0
    s = d["stage_ms_per_frame"]
    assert s["trace"] > 0 and s["temporal"] > 0 and s["denoise"] > 0
    pr = d["per_rank"]        # every rank's own view of the run, so that the first run on a node explains itself
    assert pr["blocks"] == 2 and [r["rank"] for r in pr["ranks"]] == [0, 1] and pr["launch_skew_after_the_barrier_ms"]["median"] >= 0
    for r in pr["ranks"]:     # block 0 = the overlapped loop, block 1 = the synchronous pass with the exchange on the host clock
        assert r["halo_exchange_ms_synchronous"] > 0 and r["halo_pack_ms"] > 0 and r["halo_unpack_ms"] > 0 and r["local_rows"] in (1088, 1072) and r["block_ms"]["median"] > 0


def run_bench(args, env=None, timeout=900):
    """bench.py started the way the driver starts it: `python bench.py --gpus N ...`, no launcher around it."""
    envd = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                         env={**envd, **(env or {})})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def gpu_count():
    import torch
    return torch.cuda.device_count()       # counts devices without initialising the GPU in this process


needs_two_gpus = pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` with no WORLD_SIZE in the environment — the driver's command — starts its two
    ranks as child processes (bench.spawn_ranks), relays ONE line with n_gpus 2 and the world the ranks saw.  Over gloo here, the two
    ranks sharing the one GPU; test_ranks_over_rccl_bench is the same command over RCCL on a box with two GPUs."""
    d = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--blocks", "5"], env={"VXRT_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong" and d["value"] > 1000.0
    w = d["rccl"]
    assert w["backend"] == "gloo" and w["world_size"] == 2 and [e["rank"] for e in w["devices"]] == [0, 1]
    assert len({e["pid"] for e in w["devices"]}) == 2 and "x2" in d["config"]["parallelism"]
    pr = d["per_rank"]        # per rank: its own block times, how far apart the ranks left the barrier and finished
    assert pr["blocks"] == 5 and [r["rank"] for r in pr["ranks"]] == [0, 1] and sum(r["blocks_it_finished_last"] for r in pr["ranks"]) == 5
    assert 0 <= pr["launch_skew_after_the_barrier_ms"]["median"] <= pr["launch_skew_after_the_barrier_ms"]["max"] < 50
    for r in pr["ranks"]:
        assert 0 < r["block_ms"]["min"] <= r["block_ms"]["median"] <= r["block_ms"]["max"] and r["local_rows"] in (544, 536) and r["rays_per_block"] > 0
    assert max(r["block_ms"]["median"] for r in pr["ranks"]) <= d["ms_per_step"] * 20 * 1.5


def test_halo_messages_over_rccl_on_one_gpu():
    """The RCCL branch of distributed.HaloExchange on this box's ONE GPU: a world of one rank whose context is rank 0 of 2, with both
    neighbours mapped onto the rank itself.  What travels is wrong by construction (a rank receives its own rows), so the test is about
    the transport and nothing else: batch_isend_irecv issued on the side stream, Work.wait() under torch.cuda.stream, and the tag-less
    in-order matching of the two messages that go to the same peer when nranks == 2 (NCCL ignores tags): the message addressed to the
    previous rank must land in the buffer for "from the next rank" and vice versa, bit for bit, several frames in a row, while the
    context's own stream keeps rendering."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_rccl_self_worker.py")], capture_output=True, text=True, timeout=600,
                         env={**os.environ, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": free_port(), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["frames"] >= 3 and d["message_bytes"] > 100000
    assert d["to_prev_arrived_as_from_next"] and d["to_next_arrived_as_from_prev"] and d["messages_differ"], d
    assert d["exchanges"] == d["frames"], d
    assert d["asynchronous_frames_equal_synchronous_frames"], d      # the event ordering holds: nothing ran ahead of its data


@needs_two_gpus
@pytest.mark.parametrize("nproc,radius", [(2, 3), (2, 0)])
def test_ranks_over_rccl_stitch_to_the_single_context_frame(nproc, radius):
    """test_ranks_in_processes_stitch_to_the_single_context_frame with one GPU per rank and the halo GPU to GPU over RCCL."""
    if gpu_count() < nproc:
        pytest.skip("fewer GPUs than ranks")
    d = launch(nproc, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")], env={"VXRT_TEST_RADIUS": str(radius), "VXRT_TEST_BACKEND": "nccl"})
    assert d["backend"] == "nccl" and d["world_size"] == nproc
    for name in ("rest", "slow"):
        r = d[name]
        assert r["rays_equal"] and r["geometry_pixels"] > 10000
        assert r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)
    f = d["fast"]
    assert f["rays_equal"] and f["sampled_differing_pixels"] == 0 and f["accum_differs_only_where_treated_as_disocclusion"], f


def check_config4(d, nproc, backend):
    assert d["config4"] and d["backend"] == backend and d["world_size"] == nproc and d["halo_rows"] == 8 and d["halo_exchanges"] == 2
    assert d["halo_bytes_per_rank_per_frame"] > 2 * 8 * 3840 * 36 and d["geometry_pixels"] > 4000000
    assert d["rays_equal"] and d["accum_differing_pixels"] == 0 and d["denoised_differing_pixels"] == 0, d


def test_config4_at_size_two_ranks_with_the_halo_between_processes():
    """BASELINE configs[3] at its size (castle 3840x2160, 4 spp, 8 bounces, temporal + denoise r = 8) from two PROCESSES with the halo
    through distributed.HaloExchange between them — over gloo here (the ranks share the one GPU, the messages are staged through the
    host); tests/test_gpu_configs.py has the same config from 8 contexts in one process with the halo buffers handed over by pointer.
    The stitched accumulated and denoised frames equal one context's, bit for bit."""
    d = launch(2, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")], env={"VXRT_TEST_CONFIG4": "1"})
    check_config4(d, 2, "gloo")


@needs_two_gpus
def test_config4_at_size_over_rccl_between_devices():
    """the same with one GPU per rank — as many ranks as the box has GPUs, up to 8 — and the halo device to device
    over RCCL send / recv: the exchange itself exercised between devices.  Arms on the first box with at least two GPUs."""
    n = min(gpu_count(), 8)
    d = launch(n, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")], env={"VXRT_TEST_CONFIG4": "1", "VXRT_TEST_BACKEND": "nccl"}, timeout=1500)
    check_config4(d, n, "nccl")


@needs_two_gpus
def test_ranks_over_rccl_history_halo_sized_for_the_motion():
    d = launch(2, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")],
               env={"VXRT_TEST_RADIUS": "2", "VXRT_TEST_BAND": "32", "VXRT_TEST_HALO_ROWS": "32", "VXRT_TEST_BACKEND": "nccl"})
    for name in ("rest", "slow", "fast"):
        r = d[name]
        assert r["rays_equal"] and r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)


@needs_two_gpus
def test_ranks_over_rccl_bench():
    """The driver's commands on two GPUs: the trace-stage bench and the frame loop with the halo over RCCL, each one line, each showing
    two distinct devices."""
    d = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert d["n_gpus"] == 2 and d["rccl"]["backend"].startswith("nccl") and d["rccl"]["distinct_devices"] == 2 and d["value"] > 10000.0
    p = run_bench(["--gpus", "2", "--pipeline", "--steps", "4", "--warmup", "1"])
    assert p["n_gpus"] == 2 and p["rccl"]["distinct_devices"] == 2 and "RCCL" in p["config"]["parallelism"]
    assert p["halo"]["bytes_per_rank_per_frame"] > 0 and p["value"] > 100.0


def test_halo_rows_sized_from_the_camera_path_keep_every_motion_exact():
    """The frame loop sizes each exchange for the NEXT frame's reprojection (distributed.halo_rows_for_motion from the two cameras,
    near plane 0.25): at rest the margin, a slow drift a few rows, the 17-row pan the whole 32-row band — and then two ranks equal the
    single context bit for bit on every path, the fast pan included.  The motion bound up to which N ranks and one rank are bit-equal
    is therefore band_rows - 2 rows of vertical image motion per frame (beyond it, rows of a band two bands away would be needed:
    those reprojections are treated as disocclusions, the reference's rule for off-screen ones)."""
    d = launch(2, [os.path.join(ROOT, "tests", "gpu_two_rank_worker.py")],
               env={"VXRT_TEST_RADIUS": "2", "VXRT_TEST_BAND": "32", "VXRT_TEST_HALO_ROWS": "auto"})
    for name in ("rest", "slow", "fast"):
        r = d[name]
        assert r["rays_equal"] and r["sampled_differing_pixels"] == 0 and r["accum_differing_pixels"] == 0 and r["denoised_differing_pixels"] == 0, (name, r)
    assert d["rest"]["halo_rows_chosen"] == [2, 2, 2]
    assert max(d["slow"]["halo_rows_chosen"]) <= 8
    assert d["fast"]["halo_rows_chosen"][0] >= 19 and d["fast"]["halo_rows_chosen"][1] == 2


def run_multi(args, timeout=600):
    from gpu_voxel_raytracer_amd import _build
    tool = _build.build_multi_tool()
    out = subprocess.run([tool] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("ranks,radius,spp,band", [(2, 8, 1, 64), (3, 3, 2, 48), (4, 0, 1, 16), (2, 2, 1, 16)])
def test_cpp_multi_rank_host_stitches_the_single_context_frame(tmp_path, ranks, radius, spp, band):
    """tools/vxrt_multi.cpp: the multi-rank frame loop in C++ over the C ABI, one host thread per rank — vxrt_render(TRACE | TEMPORAL),
    vxrt_halo_pack, vxrt_stream_wait_context, the four transfers, DENOISE_INTERIOR, vxrt_context_wait_stream, vxrt_halo_unpack,
    DENOISE_EDGE — here with its copy transport (hipMemcpyPeerAsync in place of ncclSend / ncclRecv: RCCL refuses two ranks on one
    device), several ranks on this box's one GPU.  --check renders the same frames in one context: bit-identical, rays equal."""
    out, d = run_multi(["menger:4", 512, 288, 4, 3, radius, tmp_path / "m.ppm", "--ranks", ranks, "--transport", "copy", "--band", band,
                        "--spp", spp, "--check"])
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    assert d["ranks"] == ranks and d["transport"] == "copy" and d["checked"] and d["differing_values"] == 0 and d["rays_equal"]
    assert d["halo_bytes_per_rank_per_frame"] > 0 and d["ms_per_frame"] > 0
    assert os.path.getsize(tmp_path / "m.ppm") > 512 * 288 * 3


def test_cpp_multi_rank_host_over_rccl_world_of_one(tmp_path):
    """The same program over its RCCL transport with the one rank this box's GPU allows: ncclCommInitAll, a frame loop that sends
    nothing, the frame equal to the single context's."""
    out, d = run_multi(["menger:3", 256, 160, 3, 3, 2, tmp_path / "m.ppm", "--ranks", 1, "--transport", "rccl", "--check"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert d["transport"] == "rccl" and d["rccl_version"] > 0 and d["differing_values"] == 0 and d["rays_equal"]
    bad, _ = run_multi(["menger:3", 64, 64, 1, 3, 0, tmp_path / "n.ppm", "--ranks", gpu_count() + 1, "--transport", "rccl"])
    assert bad.returncode == 1 and "one device per rank" in bad.stderr


@needs_two_gpus
@pytest.mark.parametrize("radius,spp", [(8, 1), (0, 1), (3, 4)])
def test_ranks_over_rccl_cpp_host(tmp_path, radius, spp):
    """tools/vxrt_multi.cpp with ncclSend / ncclRecv between as many GPUs as the box has (at most 8): stitched frame bit-identical to
    one context's."""
    n = min(gpu_count(), 8)
    out, d = run_multi(["menger:4", 1280, 720, 4, 4, radius, tmp_path / "m.ppm", "--ranks", n, "--transport", "rccl", "--spp", spp, "--check"])
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    assert d["ranks"] == n and d["transport"] == "rccl" and d["differing_values"] == 0 and d["rays_equal"]
    assert len(set(d["devices"])) == n


def test_bench_falls_back_to_gloo_when_rccl_does_not_come_up():
    """If RCCL cannot be brought up the ranks re-rendezvous over gloo and the line says so (VXRT_BENCH_FAIL_NCCL=1 stands in for the
    failure): the trace bench needs the process group for its barrier and two reductions only."""
    d = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--blocks", "3"], env={"VXRT_BENCH_FAIL_NCCL": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 1000.0
    assert d["rccl"]["backend"].startswith("gloo") and "nccl failed" in d["rccl"]["backend"]


@pytest.mark.skipif(gpu_count() != 1, reason="needs exactly one GPU: two ranks on it make RCCL refuse (duplicate GPU)")
def test_bench_survives_a_real_rccl_failure():
    """Two ranks on this box's one GPU over the default backend: ncclCommInitRank fails on both ("Duplicate GPU detected"), the ranks
    re-rendezvous over gloo on the next port and the line still comes, saying what happened."""
    d = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--blocks", "3"], env={"VXRT_BENCH_SPAWN_TIMEOUT": "300"})
    assert d["n_gpus"] == 2 and d["value"] > 1000.0
    assert d["rccl"]["backend"].startswith("gloo") and "nccl failed" in d["rccl"]["backend"]


def test_cpp_multi_rank_host_with_a_panning_camera(tmp_path):
    """tools/vxrt_multi.cpp --pan: the camera tilts by about ten rows per frame (its direction is ~30 units long: 2 units of y per
    frame); the host sizes the exchanges for the next frame's reprojection with vxrt_halo_rows_for_motion (the C ABI's form of
    distributed.halo_rows_for_motion) and three ranks still equal one context bit for bit — the accumulated history survives the band
    edges.  At rest the denoise radius' rows suffice."""
    args = ["menger:4", 384, 256, 5, 3, 2, tmp_path / "m.ppm", "--ranks", 3, "--transport", "copy", "--band", 32, "--check"]
    out, d = run_multi(args + ["--pan", "-2.0"])
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    assert d["differing_values"] == 0 and d["rays_equal"] and 8 <= d["halo_rows"] <= 32
    still, e = run_multi(args)
    assert still.returncode == 0 and e["halo_rows"] == 2 and e["differing_values"] == 0     # at rest: the denoise radius' rows suffice


def test_cpp_host_sends_and_receives_over_rccl_on_one_gpu(tmp_path):
    """tools/vxrt_multi.cpp --self-loop: ncclCommInitAll + the frame's ncclGroupStart / 2 x ncclSend / 2 x ncclRecv / ncclGroupEnd from
    C++ on this box's ONE GPU, both neighbours mapped onto the rank itself: after every frame the message for the previous rank has
    arrived in the buffer for "from the next rank" and vice versa, bit for bit (the tool exits non-zero otherwise) — the in-order
    matching of two sends and two receives to one peer that a 2-rank job relies on, this time without torch in the process."""
    out, d = run_multi(["menger:4", 512, 288, 4, 3, 3, tmp_path / "m.ppm", "--ranks", 1, "--transport", "rccl", "--band", 32, "--self-loop"])
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    assert d["transport"] == "rccl" and d["self_loop_frames_crossed"] == 4 and d["halo_bytes_per_rank_per_frame"] > 100000
