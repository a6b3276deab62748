#!/usr/bin/env python3
"""bench.py — headline benchmark of the vxrt hot path on MI355X.

Workload (BASELINE.json configs[1]): vox/menger.vox (81^3 Menger sponge, 160 000
voxels; voxel-list fixture tests/golden/scenes/menger.npz), 1920x1080, 1 sample per pixel per frame,
MAX_BOUNCES = 4, path-trace stage only (no temporal / denoise), fixed camera, Uniforms::default().
A "step" is one frame (frame_number advances every step, so every step draws different noise).

Metric: Mrays/s, where a ray is one cast_bounded_ray invocation (primary, bounce or sun shadow ray),
counted exactly on the device.  value = rays cast by all ranks / max-over-ranks wall time of K timed steps;
everything the timed region touches is resident in HBM before it starts.

Timing: a timed BLOCK is exactly K steps, bracketed by a barrier + device synchronisation on both sides.  The block is
repeated back to back (>= 50 times and for >= 1 s of GPU time) and the MEDIAN block is reported, so that a short block
(the driver's --steps 20 is 2.5 ms) is not at the mercy of one launch's jitter; `blocks` and the spread are in the line.

  python bench.py [--gpus N] [--steps K] [--warmup W]     N > 1 without a launcher: bench.py starts its N ranks itself (spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  ... bench.py --pipeline   the whole frame loop instead (BASELINE configs[3]: castle 3840x2160, 4 spp, temporal + denoise
                            r = 8 with the RCCL halo exchange between ranks), halo bytes and exchange time reported apart.

N > 1: one process per GPU; the frame's rows are dealt to the ranks in interleaved 8-row bands (BAND_ROWS)
(scene and noise table replicated, no data-path collective for the trace stage), so total work is fixed:
"scaling": "strong".  torch.distributed (RCCL) carries only the barrier and the time/ray reductions.
"""
import argparse
import json
import os

# Every trace launch in flight is a HIP stream, and a process's streams share GPU_MAX_HW_QUEUES hardware queues (4 by default).  With
# RCCL's and torch's streams in the process as well, two busy trace streams can land on one queue and serialise (measured on one of 8
# ranks' band sets, scripts/exp_first_context.py: 0.0266-0.0271 ms per frame with one or two foreign streams created first, 0.0178
# with 8 queues in every arrangement).  Read by the HIP runtime when it initialises, i.e. before the first HIP call below.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, BOUNCES, SCENE = 1920, 1080, 4, "menger"
# Rows are dealt to the ranks in interleaved bands of this height.  8 = the tracer's tile height: at 8 ranks the band sets' costs are
# within 0.0161-0.0179 ms per frame of each other, with 16-row bands 0.0153-0.0206 (1080 rows are 67.5 such bands, and the sponge's
# structure beats against the 128-row period).  (A denoise radius > 0 needs 16-row bands; the default benchmark is the trace stage.)
BAND_ROWS = 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the line's "data": the reference's own scene file (its voxel list as a fixture: /root/reference does not travel) — not a synthetic scene;
# the noise table is seeded because the reference does not ship its blue-noise archive (.MISSING_LARGE_BLOBS)
DATA_LABEL = "vox/menger.vox of the reference (voxel-list fixture tests/golden/scenes/menger.npz); noise table seeded (the reference's blue-noise zip is not shipped)"
PROFILE_DIRS = ("r05", "r04", "r03", "r02", "r01")   # newest first: where the rocprofv3 summaries of the default command are kept


def algorithmic_bytes(pixels, bounces, scene_bytes):
    """48 B/px written (3 x rgba32f) + the scene once + the noise layers one frame touches
    (8 rand() layers per bounce at most, 64 KiB each)."""
    return 48 * pixels + scene_bytes + min(8 * bounces, 512) * 128 * 128 * 4


def recorded_profile():
    """What rocprofv3 recorded for the default command (scripts/profile_round.sh -> profiles/rNN/trace_stage_summary.json):
    HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes and the SQ instruction counters.  These need separate profiler
    passes and cannot be collected from inside this process, so they are RECORDED values, not measurements of this run."""
    for d in PROFILE_DIRS:
        path = os.path.join(ROOT, "profiles", d, "trace_stage_summary.json")
        try:
            return json.load(open(path)), f"profiles/{d}/trace_stage_summary.json"
        except (OSError, ValueError):
            continue
    return None, None


SQ_COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]


def pmc_pass(counters, child_args, kernels, timeout=150):
    """One profiler pass as a CHILD process: `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py
    <child_args>` (cwd /tmp; the program itself after `--`; counters never combined with API tracing).  -> ({kernel: {counter: [value
    per dispatch, in order]}}, the child's JSON line).  Raises on any failure."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        raise RuntimeError("rocprofv3 not on PATH")
    tmp = tempfile.mkdtemp(prefix="vxrt_pmc_", dir="/tmp")
    try:
        out = subprocess.run([rocprof, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "--", sys.executable,
                              os.path.abspath(__file__)] + list(child_args), cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True,
                             text=True, timeout=timeout)
        js = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not js:
            raise RuntimeError(f"pass {counters[0]} failed (status {out.returncode}): {out.stderr[-200:]!r}")
        per_kernel = {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(kernels, r["Kernel_Name"])
                if m:
                    per_kernel.setdefault(m[0], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return per_kernel, json.loads(js[-1])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_counters(args):
    """HBM bytes and SQ instruction counters of THIS invocation's workload and schedule, from three profiler passes run as CHILD
    processes before this process touches the GPU (pmc_pass: FETCH_SIZE, WRITE_SIZE and the SQ set, each a pass of its own, over
    `bench.py <the same steps / schedule, no extras>`).  The HBM correction is MI355X_MICROARCH.md's: WRITE_SIZE as counted, the read
    side doubled (gfx950 counts a wide coalesced read at half its bytes; for the 4-16-byte gathers here the truth lies between raw and
    doubled: both are kept).  Returns (dict, note): dict None when rocprofv3 is missing or a pass fails — the line then falls back
    to the RECORDED values of profiles/ and says so."""
    steady = args.steps >= 96
    child = ["--steps", "96" if steady else str(args.steps), "--warmup", "32" if steady else str(max(args.warmup, 0)),
             "--blocks", "2" if steady else "12", "--no-cpu-baseline", "--no-extras", "--no-counters", "--bounces", str(args.bounces), "--view", args.view,
             "--frame", args.frame]
    if args.inflight:
        child += ["--inflight", str(args.inflight)]
    if args.batch:
        child += ["--batch", str(args.batch)]
    per_kernel, lines = {}, {}
    try:
        for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", SQ_COUNTERS)):
            got, lines[name] = pmc_pass(ctrs, child, r"(trace_kernel|bounce_kernel)")
            for k, c in got.items():
                per_kernel.setdefault(k, {}).update(c)
    except Exception as e:  # noqa: BLE001 — a profiler problem must not cost the line
        return None, f"profiler pass failed: {e!r}"
    child = [sys.executable, os.path.abspath(__file__)] + child
    fpl = float(lines["fetch"]["roofline"]["launch"]["frames_per_launch"])
    res = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_*> --output-format csv -- python3 bench.py " + " ".join(child[2:]),
           "frames_per_launch": fpl, "per_kernel": {}}
    raw = doubled = valu = 0.0
    for k in ("trace_kernel", "bounce_kernel"):
        c = per_kernel.get(k, {})
        f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
        if not f or not w or not c.get("SQ_INSTS_VALU"):
            return None, f"no counters for {k}"
        fk, wk = sum(f[-4:]) / len(f[-4:]), sum(w[-4:]) / len(w[-4:])        # the last launches: steady state (KB per launch)
        n = len(c["SQ_WAVES"])
        act, thr, cyc = sum(c["SQ_ACTIVE_INST_VALU"]), sum(c["SQ_THREAD_CYCLES_VALU"]), sum(c["SQ_WAVE_CYCLES"])
        res["per_kernel"][k] = {"launches_seen": len(f), "fetch_size_kb_per_launch": round(fk, 1), "write_size_kb_per_launch": round(wk, 1),
                                "valu_wave_instr_per_launch": round(sum(c["SQ_INSTS_VALU"]) / n), "salu_per_launch": round(sum(c["SQ_INSTS_SALU"]) / n),
                                "lane_utilisation": round(thr / (act * 64), 4) if act else None,
                                "waitcnt_share_of_wave_cycles": round(sum(c["SQ_WAIT_ANY"]) / cyc, 4) if cyc else None}
        raw += (fk + wk) * 1024
        doubled += (2 * fk + wk) * 1024
        valu += sum(c["SQ_INSTS_VALU"]) / n
    res["hbm_bytes_per_step_raw"] = raw / fpl
    res["hbm_bytes_per_step_read_doubled"] = doubled / fpl
    res["valu_wave_instr_per_step"] = valu / fpl
    res["ms_per_step_under_the_profiler"] = {k: v["ms_per_step"] for k, v in lines.items()}
    return res, "measured"


def live_probes():
    """The two secondary counter figures of the line, measured the same way (children, before this process touches the GPU):
    config 5's scene from outside -> FETCH_SIZE per frame of trace_kernel; config 3's frame loop at r = 8 -> SQ_INSTS_VALU per launch of
    denoise_pair_kernel.  -> dict (possibly partial); a probe that fails is simply absent and the line falls back to profiles/."""
    out = {}
    # run_probe("config5"): per view 4 warm-up + 8 timed launches of trace_kernel, outside view first, then the tunnel
    views = {"": slice(4, 12), "_tunnel": slice(16, 24)}
    try:
        got, line = pmc_pass(["FETCH_SIZE"], ["--probe", "config5"], r"trace_kernel", timeout=240)
        f = got["trace_kernel"]["FETCH_SIZE"]
        for tag, sl in views.items():
            out["config5_fetch_bytes_per_frame" + tag] = sum(f[sl]) / len(f[sl]) * 1024.0
        out["config5_probe_ms_per_frame"] = line.get("ms_per_frame")
    except Exception as e:  # noqa: BLE001
        out["config5_error"] = repr(e)[:200]
    try:    # the L2's hit rate of the same frames (a pass of its own: the TCC block has four counters, FETCH_SIZE takes three)
        got, _ = pmc_pass(["TCC_HIT_sum", "TCC_MISS_sum"], ["--probe", "config5"], r"trace_kernel", timeout=240)
        for tag, sl in views.items():
            h, m = got["trace_kernel"]["TCC_HIT_sum"][sl], got["trace_kernel"]["TCC_MISS_sum"][sl]
            out["config5_l2_hit_rate" + tag] = sum(h) / (sum(h) + sum(m))
    except Exception as e:  # noqa: BLE001
        out["config5_l2_error"] = repr(e)[:200]
    try:    # ... and their lane utilisation and VALU issue
        got, line = pmc_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES"], ["--probe", "config5"], r"trace_kernel", timeout=240)
        c = got["trace_kernel"]
        for tag, sl in views.items():
            act, thr = sum(c["SQ_ACTIVE_INST_VALU"][sl]), sum(c["SQ_THREAD_CYCLES_VALU"][sl])
            out["config5_lane_utilisation" + tag] = thr / (act * 64)
            out["config5_valu_wave_instr_per_frame" + tag] = sum(c["SQ_INSTS_VALU"][sl]) / len(c["SQ_INSTS_VALU"][sl])
    except Exception as e:  # noqa: BLE001
        out["config5_sq_error"] = repr(e)[:200]
    try:
        got, line = pmc_pass(["SQ_INSTS_VALU", "SQ_WAVES"], ["--probe", "config3"], r"denoise_pair_kernel")
        v = got["denoise_pair_kernel"]["SQ_INSTS_VALU"]
        out["denoise_r8_valu_wave_instr_per_launch"] = sum(v[-4:]) / len(v[-4:])
    except Exception as e:  # noqa: BLE001
        out["config3_error"] = repr(e)[:200]
    return out


def run_probe(which):
    """`bench.py --probe config5 | config3` (a child of live_probes, under rocprofv3): a few frames of the extra's workload, one line."""
    from gpu_voxel_raytracer_amd import ALL, TRACE, Camera, Context, scenes
    if which == "config5":
        with Context(3840, 2160, max_bounces=8, frames_in_flight=1, frames_per_launch=1) as ctx:
            ctx.set_menger(*scenes.CONFIG5)
            ms = {}
            for view in ("outside", "tunnel"):      # live_probes: per view 4 warm-up + 8 timed launches, in this order
                ctx.camera = Camera(*scenes.config5_cameras()[view])
                ctx.render_frames(TRACE, 4)
                ctx.sync()
                t0 = time.perf_counter()
                ctx.render_frames(TRACE, 8)
                ctx.sync()
                ms[view] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
            print(json.dumps({"probe": which, "ms_per_frame": ms["outside"], "ms_per_frame_tunnel": ms["tunnel"]}), flush=True)
    else:
        pos, mrgb, size = scenes.load_scene("monu10")
        with Context(3840, 2160, max_bounces=8, frames_in_flight=2, frames_per_launch=4) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*scenes.bench_camera(size))
            ctx.denoise_uniforms.radius = 8
            for _ in range(5):
                ctx.render_spp(ALL, 4)
            ctx.sync()
            print(json.dumps({"probe": which}), flush=True)


def recorded_json(name):
    """profiles/rNN/<name>, newest round first -> (parsed, path) or (None, None).  RECORDED values: counters need profiler passes of
    their own and are not measurements of this run (profiles/README.md says which script and schedule made each file)."""
    for d in PROFILE_DIRS:
        path = os.path.join(ROOT, "profiles", d, name)
        try:
            return json.load(open(path)), f"profiles/{d}/{name}"
        except (OSError, ValueError):
            continue
    return None, None


def measure_reference_loop(Context, Camera, flags, scenes, device, frames=120):
    """The loop the reference itself runs (src/main.rs:34-38 -> Context::update / render, src/context.rs:1959-2075, 2136-2162), timed as a
    drop-in host would call it: per frame `vxrt_set_camera` (a moving camera: frame_loop.orbit_camera, a quarter revolution over 960
    frames = 0.09 degrees per frame), `vxrt_render(ALL)` — the three dispatches — and `vxrt_sync`, with the reference's MAX_BOUNCES 3
    (shaders/voxels.comp:4), at 1920x1080 and at 1600x1600 (the reference's 800x800 logical window, src/context.rs:597, at scale factor 2),
    denoise radius 0 (the reference's default) and 2.  Four more rows per case: the same loop when the host pulls EVERY denoised frame
    with `vxrt_read` (synchronous, pageable: what round 5 had), with `vxrt_read_async` into two pinned buffers (frame n travels while
    frame n + 1 renders), the transfer alone, and the same frames through `vxrt_render_path` (the whole camera path handed over: 16
    frames per trace launch, two launches in flight) — the gap between "drop-in" and "pipelined" in one place — and the same per-frame calls
    from a host that does not wait for a frame before it submits the next (`frames_in_flight` 2, one sync at the end)."""
    from gpu_voxel_raytracer_amd import DENOISED
    from gpu_voxel_raytracer_amd.frame_loop import orbit_camera
    ALL, TIMED = flags
    pos, mrgb, size = scenes.load_scene(SCENE)
    path = [orbit_camera(size, 0.62 + 0.25 * f / 960.0) for f in range(frames + 8)]
    rows = {}
    for w, h in ((1920, 1080), (1600, 1600)):
        for radius in (0, 2):
            row = {}
            with Context(w, h, device=device, max_bounces=3) as ctx:          # one frame at a time on one stream: the reference's single queue
                ctx.recreate_octree(pos, mrgb)
                ctx.denoise_uniforms.radius = radius

                def loop(first, count, pull=None, timed=False):
                    for f in range(first, first + count):
                        ctx.camera = Camera(*path[f])
                        ctx.render(ALL | (TIMED if timed else 0))
                        if pull is None:
                            ctx.sync()
                        else:
                            pull(f)
                def median_ms(pull, count, blocks=3, after=None):
                    """median over `blocks` runs of `count` frames (ms per frame): a one-time cost — a queue made on first use, fresh pinned
                    pages — lands in one block and not in the figure"""
                    t = []
                    for _ in range(blocks):
                        t0 = time.perf_counter()
                        loop(8, count, pull)
                        if after is not None:
                            after()
                        t.append((time.perf_counter() - t0) / count * 1e3)
                    return sorted(t)[len(t) // 2]
                loop(0, 8)
                ctx.reset_stats()
                t0 = time.perf_counter()
                loop(8, frames, timed=True)
                dt = time.perf_counter() - t0
                st = ctx.stats()
                row["ms_per_frame"] = round(median_ms(None, frames // 2), 4)          # without the stage events of the loop above
                row["ms_per_frame_with_stage_events"] = round(dt / frames * 1e3, 4)
                row["mrays_per_s"] = round(st.rays / frames / row["ms_per_frame"] / 1e3, 1)
                row["rays_per_pixel"] = round(st.rays / frames / (w * h), 4)
                row["stage_ms"] = {"trace": round(st.trace_ms / frames, 4), "temporal": round(st.temporal_ms / frames, 4), "denoise": round(st.denoise_ms / frames, 4)}
                # ... the host takes every denoised frame: synchronously into pageable memory (vxrt_read)
                img = np.zeros((h, w, 4), np.float32)
                nbytes = img.nbytes

                def pull_sync(f):
                    ctx.read_into(DENOISED, img)
                loop(0, 4, pull_sync)
                row["with_vxrt_read_ms_per_frame"] = round(median_ms(pull_sync, frames // 3), 4)
                # ... without blocking: two pinned buffers, frame f's transfer is waited for after frame f + 1 has been submitted
                pinned = [ctx.pinned_image(), ctx.pinned_image()]

                def pull_async(f):
                    ctx.read_async(DENOISED, pinned[f & 1], f & 1)
                    ctx.read_wait((f + 1) & 1)             # the previous frame has arrived: the host may show it

                def drain():
                    ctx.read_wait(0)
                    ctx.read_wait(1)
                loop(0, 16, pull_async)
                drain()
                row["with_vxrt_read_async_ms_per_frame"] = round(median_ms(pull_async, frames // 2, after=drain), 4)
                ctx.sync()
                t0 = time.perf_counter()
                for k in range(16):
                    ctx.read_async(DENOISED, pinned[k & 1], k & 1)
                    ctx.read_wait(k & 1)
                row["transfer_alone_ms"] = round((time.perf_counter() - t0) / 16 * 1e3, 4)
                row["transfer_gb_per_s"] = round(nbytes / (row["transfer_alone_ms"] * 1e-3) / 1e9, 1)
                row["read_async_over_max_of_render_and_transfer"] = round(row["with_vxrt_read_async_ms_per_frame"] / max(row["ms_per_frame"], row["transfer_alone_ms"]), 3)
                for pb in pinned:
                    pb.close()
            with Context(w, h, device=device, max_bounces=3, frames_in_flight=2) as ctx:    # the same calls, but the host does not wait for a frame
                ctx.recreate_octree(pos, mrgb)                                              # before it submits the next (what wgpu's queue gives the reference
                ctx.denoise_uniforms.radius = radius                                        # too: submit returns at once): two frames' trace stages overlap
                for f in range(8):
                    ctx.camera = Camera(*path[f])
                    ctx.render(ALL)
                ctx.sync()
                t = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    for f in range(8, 8 + frames):
                        ctx.camera = Camera(*path[f])
                        ctx.render(ALL)
                    ctx.sync()
                    t.append((time.perf_counter() - t0) / frames * 1e3)
                row["without_waiting_for_each_frame_ms_per_frame"] = round(sorted(t)[1], 4)
            with Context(w, h, device=device, max_bounces=3, frames_in_flight=2, frames_per_launch=16) as ctx:   # the same frames, the path handed over
                ctx.recreate_octree(pos, mrgb)
                ctx.denoise_uniforms.radius = radius
                ctx.render_path(ALL, [p[0] for p in path[:8]], [p[1] for p in path[:8]], path[0][2])
                ctx.sync()
                n = frames // 16 * 16
                t0 = time.perf_counter()
                ctx.render_path(ALL, [p[0] for p in path[8:8 + n]], [p[1] for p in path[8:8 + n]], path[0][2])
                ctx.sync()
                row["vxrt_render_path_ms_per_frame"] = round((time.perf_counter() - t0) / n * 1e3, 4)
            rows[f"{w}x{h}_r{radius}"] = row
    return {"workload": f"vox/{SCENE}.vox, MAX_BOUNCES 3, orbiting camera (0.09 degrees per frame), trace + temporal + denoise; {frames} frames per figure",
            "loop": "per frame: vxrt_set_camera -> vxrt_render(VXRT_ALL) -> vxrt_sync, frames_in_flight 1, frames_per_launch 1 (src/context.rs:2004-2075)",
            "rows": rows}


def measure_config5_touch(device, views=("outside", "tunnel"), frames=2):
    """SURVEY 8d's "bricks actually touched" for BASELINE configs[4]'s scene: the UNIQUE 64-byte lines of node records and leaf words
    that one 3840x2160, 8-bounce frame reads, per view — the compulsory scene traffic of a frame, each byte once.  Measured with the
    touch map of the -DVXRT_VARIANTS=1 library (csrc/trace_common.h: VX_TOUCH; the product's kernels hold no such code), loaded
    beside the product for this probe only; the frames are the frames the timed view renders (same scene, camera, frame numbers from 3
    on — the noise differs per frame, so `frames` consecutive frames are measured apart).  -> {view: {...}}"""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context, host, scenes
    out = {}
    host.use_library(host.variants_library())
    try:
        with Context(3840, 2160, device=device, max_bounces=8, frames_in_flight=1, frames_per_launch=1) as ctx:
            ctx.set_menger(*scenes.CONFIG5)
            for view in views:
                ctx.camera = Camera(*scenes.config5_cameras()[view])
                ctx.render_frames(TRACE, 2)
                ctx.touch_map(True)
                per_frame = []
                for _ in range(frames):
                    ctx.render_frames(TRACE, 1)
                    per_frame.append(ctx.touch_count(reset=True))
                ctx.render_frames(TRACE, 8)                       # ... and what eight consecutive frames touch together
                eight = ctx.touch_count(reset=True)
                ctx.touch_map(False)
                t = per_frame[0]
                out[view] = {"unique_scene_bytes_per_frame": t["unique_bytes_64"], "at_128_byte_lines": t["unique_bytes_128"],
                             "node_record_bytes": 64 * t["node_lines_64"], "leaf_word_bytes": 64 * t["leaf_lines_64"],
                             "other_frames": [f["unique_bytes_64"] for f in per_frame[1:]], "eight_frames_together": eight["unique_bytes_64"],
                             "scene_bytes": t["scene_bytes"], "share_of_the_scene": round(t["unique_bytes_64"] / t["scene_bytes"], 5)}
    finally:
        host.use_library(None)
    return out


def measure_config5(Context, Camera, TRACE, scenes, device, probes):
    """BASELINE configs[4]'s scene (procedural Menger level 7 clipped to 2048^3: 5.6 GB, HBM-resident — north_star's "HBM-bound stress", the one
    config where achieved GB/s against 8 TB/s is the right axis) on this GPU: one 3840x2160, 8-bounce frame per launch (one GPU's share of
    the 7680x4320 frame is a band set of it), from outside and from inside a tunnel.  Per view: ms per frame (this run), and a roofline in
    SURVEY 8d's sense — ALGORITHMIC bytes = 48 B/px written + the UNIQUE scene bytes the frame touches (measure_config5_touch: every
    64-byte line of node records and leaf words it reads, counted once) + the noise layers, over the frame time — beside what the
    counters say the frame really moved (FETCH_SIZE, L2 hit rate, lanes: the run's own rocprofv3 child passes), and their ratio:
    refetch = bytes fetched / unique bytes = how often the same line comes from beyond the L2 again."""
    out = {}
    try:
        touch = measure_config5_touch(device)
    except Exception as e:  # noqa: BLE001 — an extra must never cost the headline line
        touch = {"error": repr(e)[:300]}
    for view, key, tag, frames, blocks in (("outside", "config5_outside_view", "", 24, 5), ("tunnel", "config5_tunnel_view", "_tunnel", 8, 3)):
        try:
            c5 = dict(
                measure_view(Context, Camera, TRACE, None, None, scenes.config5_cameras()[view], device, 8, 1, 1, width=3840, height=2160, frames=frames, blocks=blocks,
                             setup=lambda ctx: ctx.set_menger(*scenes.CONFIG5)),
                workload="BASELINE configs[4]'s scene (procedural Menger level 7 clipped to 2048^3, 5.6 GB: HBM-resident) at 3840x2160, "
                         f"8 bounces, one frame per launch, view '{view}'; one GPU's share is a band set of the 7680x4320 frame")
            sec = c5["ms_per_frame"] * 1e-3
            written = 48.0 * 3840 * 2160
            noise = min(8 * 8, 512) * 128 * 128 * 4
            roof = {"bound": "hbm", "kernel": "trace_kernel (all-in-one, 6 waves per SIMD)", "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_written_per_frame": int(written)}
            t = touch.get(view)
            if t:
                alg = written + t["unique_scene_bytes_per_frame"] + noise
                roof.update({"algorithmic_bytes_per_frame": int(alg), "unique_scene_bytes_per_frame": t["unique_scene_bytes_per_frame"],
                             "unique_scene_bytes": t, "noise_layer_bytes": noise,
                             "achieved": round(alg / sec / 1e9, 1), "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4),
                             "algorithmic_source": "48 B/px + the unique 64-byte lines of node records and leaf words one frame reads (touch map of the -DVXRT_VARIANTS=1 "
                                                   "library, MEASURED in this invocation on the same scene, view and frame size) + 64 noise layers"})
            elif "error" in touch:
                roof["unique_scene_bytes_error"] = touch["error"]
            fetch = probes.get("config5_fetch_bytes_per_frame" + tag)
            if fetch:
                roof.update({"bytes_fetched_per_frame": int(fetch), "traffic": int(written + 2 * fetch), "traffic_raw": int(written + fetch),
                             "achieved_traffic_raw": round((written + fetch) / sec / 1e9, 1), "frac_traffic_raw": round((written + fetch) / sec / 1e9 / HBM_PEAK_GBS, 4),
                             "achieved_traffic_read_doubled": round((written + 2 * fetch) / sec / 1e9, 1),
                             "frac_traffic_read_doubled": round((written + 2 * fetch) / sec / 1e9 / HBM_PEAK_GBS, 4)})
                if t:
                    roof["refetch"] = round(fetch / t["unique_scene_bytes_per_frame"], 2)                    # as counted
                    roof["refetch_read_doubled"] = round(2 * fetch / t["unique_scene_bytes_per_frame"], 2)   # the guide's gfx950 correction of the read side
                    roof["traffic_over_algorithmic"] = round((written + 2 * fetch) / alg, 2)
            if probes.get("config5_l2_hit_rate" + tag) is not None:
                roof["l2_hit_rate"] = round(probes["config5_l2_hit_rate" + tag], 4)
            if probes.get("config5_lane_utilisation" + tag) is not None:
                roof["lane_utilisation"] = round(probes["config5_lane_utilisation" + tag], 4)
                vi = probes["config5_valu_wave_instr_per_frame" + tag]
                roof["valu_wave_instr_per_frame"] = int(vi)
                roof["valu_issue_slot_frac"] = round(vi * 2 / (1024 * 2.4e9 * sec), 3)
            errs = [v for k, v in probes.items() if k.startswith("config5") and k.endswith("error")]
            roof["counters_source"] = ("FETCH_SIZE, TCC_HIT_sum / TCC_MISS_sum and the SQ counters MEASURED in this invocation: three rocprofv3 --pmc child passes over "
                                       "`bench.py --probe config5` (the same scene, views and frame size); traffic = written + 2 x fetched (the guide's gfx950 read-side "
                                       "correction), traffic_raw as counted" if not errs else f"a probe failed: {errs}")
            c5["roofline"] = roof
            out[key] = c5
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)[:300]}
    return out


def measure_config3(Context, Camera, ALL, TIMED, scenes, device, shown=12, live_valu=None):
    """BASELINE configs[2] on this GPU, whole frame loop: vox/monu10.vox 3840x2160, 4 samples per pixel per displayed frame
    (vxrt_render_spp), 8 bounces, temporal + denoise at radius 2 and radius 8 (exact mode, bit-identical to the oracle).  Per
    displayed frame: wall ms, per-stage ms from HIP events (the stages overlap the next frame's trace launch, so they do not add up),
    Gray/s, and the fraction of the HBM roofline the SURVEY 8d byte model amounts to: (48 + 16) spp + 16 + 80 + 64 B/px."""
    w, h, spp, bounces = 3840, 2160, 4, 8
    pos, mrgb, size = scenes.load_scene("monu10")
    out = {"workload": f"vox/monu10.vox {w}x{h}, {spp} spp, {bounces} bounces, temporal + denoise (BASELINE configs[2]); bench camera; "
                       f"{spp} frames per trace launch x 2 launches in flight; {shown} displayed frames after 3 of warm-up"}
    post, post_path = recorded_json("post_stages_summary.json")
    with Context(w, h, device=device, max_bounces=bounces, frames_in_flight=2, frames_per_launch=spp) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*scenes.bench_camera(size))
        alg = ((48 + 16) * spp + 16 + 80 + 64) * w * h
        for radius in (2, 8):
            ctx.denoise_uniforms.radius = radius
            for _ in range(3):
                ctx.render_spp(ALL, spp)
            ctx.sync()
            ctx.reset_stats()
            t0 = time.perf_counter()
            for _ in range(shown):
                ctx.render_spp(ALL | TIMED, spp)
            ctx.sync()
            dt = (time.perf_counter() - t0) / shown
            st = ctx.stats()
            r = {"ms_per_displayed_frame": round(dt * 1e3, 4), "gray_per_s": round(st.rays / shown / dt / 1e9, 2),
                 "rays_per_pixel_per_sample": round(st.rays / shown / spp / (w * h), 4),
                 "stage_ms": {"trace": round(st.trace_ms / shown, 4), "temporal": round(st.temporal_ms / shown, 4), "denoise": round(st.denoise_ms / shown, 4)},
                 "roofline": {"bound": "hbm", "algorithmic_bytes_per_displayed_frame": alg, "achieved": round(alg / dt / 1e9, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4)}}
            if radius == 8 and (post is not None or live_valu):
                post = post or {"kernels": {"denoise r=8 exact": {"sq_insts_valu_per_launch": live_valu}}}
                try:    # VALU issue of denoise_pair_kernel: wave-instructions (RECORDED) x 2 cycles on 1024 SIMDs at 2.4 GHz / this run's stage time
                    k = post["kernels"]["denoise r=8 exact"]
                    instr = float(live_valu) if live_valu else float(k["sq_insts_valu_per_launch"])
                    issue_ms = instr * 2 / (1024 * 2.4e9) * 1e3
                    r["denoise_valu"] = {"kernel": "denoise_pair_kernel<exact, 8>", "valu_wave_instr_per_launch": int(instr),
                                         "lane_instr_per_tap": round(instr * 64 / (w * h * 289), 2), "issue_ms": round(issue_ms, 3),
                                         "issue_slot_frac": round(issue_ms / (st.denoise_ms / shown), 3),
                                         "source": (("instruction count MEASURED in this invocation (a rocprofv3 --pmc SQ_INSTS_VALU child pass over `bench.py "
                                                     "--probe config3`)" if live_valu else
                                                     f"instruction count RECORDED in {post_path} (rocprofv3 --pmc SQ_INSTS_VALU pass of scripts/profile_post.sh)") +
                                                    ", divided by this run's HIP-event time of the stage")}
                except (KeyError, ZeroDivisionError, TypeError):
                    pass
            out[f"radius_{radius}"] = r
        # the same frame loop with the denoiser in TOLERANT mode (VXRT_OPT_DENOISE_MODE 1: reciprocal multiply, fused multiply-adds and the
        # hardware's v_exp_f32 instead of the contract's division and polynomial): not bit-exact, and far inside north_star's own tolerance
        # (per-pixel RMSE <= 1e-3) — stated here against the exact mode on the SAME accumulated frame: the denoise stage alone is run twice
        # on what the last displayed frame accumulated, once per mode, and the two denoised images are compared
        from gpu_voxel_raytracer_amd.host import DENOISE, DENOISED, OPT_DENOISE_MODE
        ctx.sync()
        ctx.render_stage(DENOISE)
        exact = ctx.read(DENOISED)[..., :3].astype(np.float64)
        ctx.set_option(OPT_DENOISE_MODE, 1)
        ctx.render_stage(DENOISE)
        tol = ctx.read(DENOISED)[..., :3].astype(np.float64)
        fin = np.isfinite(exact).all(-1) & np.isfinite(tol).all(-1)
        for _ in range(3):
            ctx.render_spp(ALL, spp)
        ctx.sync()
        ctx.reset_stats()
        t0 = time.perf_counter()
        for _ in range(shown):
            ctx.render_spp(ALL | TIMED, spp)
        ctx.sync()
        dt = (time.perf_counter() - t0) / shown
        st = ctx.stats()
        out["radius_8_tolerant"] = {
            "ms_per_displayed_frame": round(dt * 1e3, 4), "gray_per_s": round(st.rays / shown / dt / 1e9, 2),
            "stage_ms": {"trace": round(st.trace_ms / shown, 4), "temporal": round(st.temporal_ms / shown, 4), "denoise": round(st.denoise_ms / shown, 4)},
            "rmse_vs_exact_mode": float(np.sqrt(((exact - tol)[fin] ** 2).mean())), "max_abs_error_vs_exact_mode": float(np.abs((exact - tol)[fin]).max()),
            "north_star_tolerance_rmse": 1e-3,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_displayed_frame": alg, "achieved": round(alg / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4)},
            "note": "denoise.comp in tolerant mode (VXRT_OPT_DENOISE_MODE 1); the exact mode (radius_8) is the one the parity tests hold bit-exact"}
    return out


def measure_config4_rank(Context, Camera, flags, scenes, device, rank=0, nranks=8, shown=12):
    """BASELINE configs[3] seen from ONE of its 8 ranks, alone on this GPU: vox/castle.vox 3840x2160, 4 spp, 8 bounces, temporal +
    denoise r = 8, the rank's whole loop — trace + temporal, halo pack, denoise of the interior tiles, halo unpack, denoise of the edge
    tiles — with 64-row bands (distributed.band_rows_for).  No transfer: the two messages it unpacks were packed once, before the timed
    loop, by contexts of its two neighbour ranks (real rows of the right ranks, one frame old).  What a rank of the 8-GPU job has to do
    per displayed frame, not a measurement of the job."""
    import ctypes as C
    TRACE, TEMPORAL, TIMED, INTERIOR, EDGE = flags
    from gpu_voxel_raytracer_amd import distributed
    w, h, spp, bounces, radius = 3840, 2160, 4, 8, 8
    band = distributed.band_rows_for(radius, h, nranks)
    pos, mrgb, size = scenes.load_scene("castle")
    cam = Camera(*scenes.close_camera(size))
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipFree.argtypes = [C.c_void_p]
    bufs = {}

    def make(r, inflight):
        ctx = Context(w, h, device=device, max_bounces=bounces, rank=r, nranks=nranks, band_rows=band, frames_in_flight=inflight, frames_per_launch=spp)
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = cam
        ctx.denoise_uniforms.radius = radius
        n = ctx.halo_bytes()
        pair = (C.c_void_p(), C.c_void_p())
        for b in pair:
            if rt.hipMalloc(C.byref(b), n) != 0:
                raise RuntimeError("hipMalloc of a halo message failed")
        bufs[r] = pair
        return ctx

    try:
        for r in ((rank - 1) % nranks, (rank + 1) % nranks):        # the neighbours' messages, once
            with make(r, 1) as nb:
                nb.render_spp(TRACE | TEMPORAL, spp)
                nb.halo_export(bufs[r][0].value, bufs[r][1].value)
        with make(rank, 2) as ctx:
            from_prev, from_next = bufs[(rank - 1) % nranks][1].value, bufs[(rank + 1) % nranks][0].value

            def frame(extra=0):
                ctx.render_spp(TRACE | TEMPORAL | extra, spp)
                ctx.halo_pack(bufs[rank][0].value, bufs[rank][1].value)
                ctx.render_stage(INTERIOR | extra)
                ctx.halo_unpack(from_prev, from_next)
                ctx.render_stage(EDGE | extra)
            for _ in range(3):
                frame()
            ctx.sync()
            ctx.reset_stats()
            t0 = time.perf_counter()
            for _ in range(shown):
                frame(TIMED)
            ctx.sync()
            dt = (time.perf_counter() - t0) / shown
            st, info = ctx.stats(), ctx.halo_info()
            return {"workload": f"vox/castle.vox {w}x{h}, {spp} spp, {bounces} bounces, temporal + denoise r={radius} (BASELINE configs[3]; camera 'close' as in --pipeline): rank {rank} of "
                                f"{nranks} alone on this GPU, {band}-row bands (the last round lower: {st.local_rows} of {h} rows), its whole loop without "
                                f"transfer time; {shown} displayed frames after 3 of warm-up",
                    "ms_per_displayed_frame": round(dt * 1e3, 4), "gray_per_s_this_rank": round(st.rays / shown / dt / 1e9, 2),
                    "local_rows": int(st.local_rows), "halo_bytes_per_rank_per_frame": 2 * int(info.message_bytes), "halo_rows": int(info.rows),
                    "interior_tile_rows": int(info.interior_tile_rows), "edge_tile_rows": int(info.edge_tile_rows),
                    "stage_ms": {"trace": round(st.trace_ms / shown, 4), "temporal": round(st.temporal_ms / shown, 4), "denoise": round(st.denoise_ms / shown, 4),
                                 "halo_pack": round(st.halo_pack_ms / max(st.halo_exchanges, 1), 5), "halo_unpack": round(st.halo_unpack_ms / max(st.halo_exchanges, 1), 5)}}
    finally:
        for pair in bufs.values():
            for b in pair:
                if b.value:
                    rt.hipFree(b)


def cpu_baseline(pos, mrgb, cam, target_seconds=float(os.environ.get("VXRT_BENCH_CPU_SECONDS", "8"))):
    """The CPU oracle (restatement of shaders/voxels.comp, oracle/oshaders.cpp) timed on this host's cores on
    whole frames of the same workload: a reported baseline, not the target."""
    from oracle import oracle as O
    O.build()
    octree = O.create_octree(pos, mrgb)
    noise = O.noise_table()
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], WIDTH, HEIGHT))
    threads = os.cpu_count() or 1
    rays, frames, t0 = 0, 0, time.perf_counter()
    while True:
        frames += 1
        u.frame_number = frames
        rays += O.trace(octree, noise, u, WIDTH, HEIGHT, BOUNCES, crop=(0, 0, WIDTH, HEIGHT), nthreads=threads)[3]
        dt = time.perf_counter() - t0
        if dt >= target_seconds or frames >= 4096:
            break
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} full {WIDTH}x{HEIGHT} frames of the same workload ({rays} rays) in {dt:.1f} s, "
                      f"{threads} threads"}


def cpu_baseline_cpu_rs(target_seconds=float(os.environ.get("VXRT_BENCH_CPU_RS_SECONDS", "6"))):
    """BASELINE configs[0] as north_star asks for it: the reference's src/cpu.rs ray caster (restated, oracle/ocpu.cpp — the
    original is orphaned Rust that does not compile) on vox/3x3x3.vox at 256x256, threaded over the host's cores the way
    src/cpu.rs:43-46 uses rayon.  A ray here is one Octree::cast_ray (primary + one shadow ray per hit)."""
    from gpu_voxel_raytracer_amd import scenes
    from oracle import oracle as O
    pos, mrgb, size = scenes.load_scene("3x3x3")
    cam_pos, cam_dir, fov = scenes.bench_camera(size)
    basis = O.camera_axis_scaled(cam_pos, cam_dir, fov, 256, 256)
    backend = O.CpuRsBackend(pos.astype(np.uint16), mrgb[:, 1:])
    cores = os.cpu_count() or 1
    # rayon keeps a pool; the restatement starts its threads per frame, which costs more than a 256x256 frame's work once there are
    # hundreds of them: use the thread count that is fastest on this host (and report it as `cores`)
    best = (float("inf"), cores)
    for threads in sorted({cores, 64, 32, 16, 8} & set(range(1, cores + 1))):
        t0 = time.perf_counter()
        for _ in range(3):
            backend.render(cam_pos * 2, basis, 256, 256, 0.0, threads)
        best = min(best, ((time.perf_counter() - t0) / 3, threads))
    threads = best[1]
    img = backend.render(cam_pos * 2, basis, 256, 256, 0.0, threads)       # warm-up
    rays_per_frame = 256 * 256 + int((img.reshape(-1, 3).max(1) > 0).sum())  # lower bound: lit pixels cast a shadow ray
    times, t_end = [], time.perf_counter() + target_seconds
    while time.perf_counter() < t_end or len(times) < 20:
        t0 = time.perf_counter()
        backend.render(cam_pos * 2, basis, 256, 256, 0.0, threads)
        times.append(time.perf_counter() - t0)
    backend.close()
    ms = statistics.median(times) * 1e3
    return {"value": round(ms, 4), "unit": "ms/frame", "cores": threads, "kind": "port",
            "mrays_per_s": round(rays_per_frame / (ms * 1e-3) / 1e6, 2),
            "sample": f"vox/3x3x3.vox 256x256, src/cpu.rs shading at time 0, median of {len(times)} frames, {threads} threads "
                      f"(the fastest of 8..{cores} on this host)"}


def measure_latency(Context, Camera, TRACE, pos, mrgb, cam, device, bounces, frames=60):
    """One vxrt_render(TRACE) per frame, each waited for before the next is submitted: what a render loop that calls the library
    once per displayed frame sees (the library then takes its all-in-one kernel).  Median over `frames` frames, ms."""
    ctx = Context(WIDTH, HEIGHT, device=device, max_bounces=bounces, frames_in_flight=1, frames_per_launch=1)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)
    for _ in range(10):
        ctx.render(TRACE)
    ctx.sync()
    times = []
    for _ in range(frames):
        t0 = time.perf_counter()
        ctx.render(TRACE)
        ctx.sync()
        times.append(time.perf_counter() - t0)
    ctx.close()
    return statistics.median(times) * 1e3


def measure_view(Context, Camera, TRACE, pos, mrgb, cam, device, bounces, inflight, batch, width=None, height=None, frames=480, blocks=7,
                 setup=None):
    """Throughput of the trace stage for another view / scene with the headline's schedule: median of `blocks` blocks of `frames` frames."""
    width, height = width or WIDTH, height or HEIGHT
    ctx = Context(width, height, device=device, max_bounces=bounces, frames_in_flight=inflight, frames_per_launch=batch)
    if setup is not None:
        setup(ctx)
    else:
        ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)
    ctx.render_frames(TRACE, max(2 * batch * inflight, 32))
    ctx.sync()
    res = []
    for _ in range(blocks):
        ctx.reset_stats()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.render_frames(TRACE, frames)
        ctx.sync()
        dt = time.perf_counter() - t0
        res.append((dt, ctx.stats().rays))
    st = ctx.stats()
    ctx.close()
    res.sort()
    dt, rays = res[len(res) // 2]
    return {"value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dt / frames * 1e3, 4),
            "rays_per_pixel": round(rays / frames / (width * height), 4), "frames_per_block": frames, "blocks": blocks,
            "scene_bytes": int(st.scene_bytes)}


def pick_schedule(world, steps, inflight=0, batch=0):
    """Launches in flight and frames per launch for `world` ranks and a timed block of `steps` frames (0 = choose).
    A frame's longest tile is a serial chain of ~0.15-0.3 ms however few rows a rank owns, so a rank needs that much work in
    flight: 16 frames per launch, and the more launches overlapping the smaller its share (measured per rank with
    scripts/exp_rank_emulation.py, round 3: 2x16 for 1 rank, 3x16 for 2 / 4 / 8 ranks).  Never more than 3 trace streams: with the
    context's own stream that makes 4, and RCCL / torch bring streams of their own; a process gets GPU_MAX_HW_QUEUES = 8 hardware
    queues here (set above), and streams beyond the queues share one and serialise (scripts/exp_first_context.py).
    A short block (fewer frames than two full launches per stream) starts on an idle GPU and ends with a drain.  Its frames go out in
    launches of 8 on three streams for 1 and 2 ranks, of 16 on two streams for 4 ranks — whole groups of 8 frames, so that
    trace_kernel's waves hold 8 frames of a pixel row (VXRT_OPT_FRAME_LANES) — and as ONE launch from 8 ranks on: the smaller a rank's
    share, the more a launch is its longest tile's chain and nothing else, and one launch pays that chain once
    (scripts/exp_short_block.py with LANES=1, 20 steps, ms per block, rank 0 alone on the GPU: 1 rank 8+8+4 2.33, 7+7+6 2.36, 20 2.50;
    2 ranks 8+8+4 1.227, 16+4 1.227, 7+7+6 1.262; 4 ranks 16+4 0.700, 8+8+4 0.704, 20 0.714; 8 ranks 20 0.402, 16+4 0.411, 8+8+4 0.423)."""
    short = steps < 2 * 16 * (inflight if inflight > 0 else (2 if world == 1 else 3))
    if inflight <= 0 and batch <= 0 and short:
        if world <= 2:
            return 3, min(8, steps)
        if world < 8:
            return 2, min(16, steps)
        inflight = min(3, -(-steps // 32))
        return inflight, max(1, -(-steps // inflight))
    if inflight <= 0:
        inflight = 2 if world == 1 else 3
    if batch <= 0:
        batch = max(1, min(32, -(-steps // inflight))) if short else 16
    return inflight, batch


def pick_deal(world, steps, inflight=0, batch=0):
    """How a SHORT block is rendered on `world` ranks, beyond pick_schedule's launches: (band_rows, vxrt_config.tracer).
    A rank's share of a short block (the driver's --steps 20) on 8 GPUs is 2.5 frames' worth of work wrapped around two chains of
    lock-step waves: as head + compacted tail the launch is trace_kernel (172-195 us, the chip two thirds idle by its end) followed by
    bounce_kernel (220-235 us, drained again), 0.41-0.46 ms per block over the ranks; the all-in-one kernel needs 1.3 x the instructions
    but drains once: 0.37-0.44 ms.  And 8-row bands leave a rank only ~8 bands with geometry (1080 rows, 64-row period): the ranks'
    blocks differ by 16 %; 4-row bands — whose rows still pair up in the waves of a frame-lane launch (2 rows x 4 frames) — bring them
    within 4 %: all-in-one + 4-row bands 0.405-0.423 ms for every rank (scripts/archive/r05/r5e.sh, r5j.sh; profiles/r05/short_block.txt).  Two
    and four ranks, and every steady-state run, keep the default: their launches are long enough for the pair to win."""
    short = steps < 2 * 16 * (inflight if inflight > 0 else (2 if world == 1 else 3))
    if short and world >= 8 and inflight <= 0 and batch <= 0:
        return 4, 1
    # the steady state from 8 ranks on keeps head + tail and takes the finer interleave too: slowest rank 0.0148 against 0.0151 ms per
    # frame (emulated, profiles/r05/rank_emulation.txt; 2 and 4 ranks: no difference)
    return (4 if world >= 8 else BAND_ROWS), 0


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) started without a launcher (no WORLD_SIZE in the environment): start the N ranks as CHILD
    processes of this one, one per GPU, with the rendezvous variables torch.distributed.run would set, relay rank 0's one JSON line
    and exit with the children's status.  This parent makes no HIP / torch.cuda call (it does not even import torch) and never
    replaces itself (no os.exec*); a rank that fails ends the job with a non-zero status — nothing is restarted."""
    import signal
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()))
    env.setdefault("VXRT_BENCH_PORT2", str(free_port()))      # where the ranks agree on RCCL or gloo (init_dist): checked free here, once
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    deadline = time.time() + float(os.environ.get("VXRT_BENCH_SPAWN_TIMEOUT", "1500"))
    line, status = None, 0
    import threading
    out_lines = []
    reader = threading.Thread(target=lambda: out_lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    # one waiter per rank notes WHEN its rank ended: a rank that fails takes its peers down with it within milliseconds (their collective
    # breaks), and the job's status is that of the rank that went first, not of whichever a polling loop happens to look at first
    exits, lock = [], threading.Lock()

    def wait_for(r):
        rc = procs[r].wait()
        with lock:
            exits.append((time.monotonic(), r, rc))
    waiters = [threading.Thread(target=wait_for, args=(r,), daemon=True) for r in range(n)]
    for t in waiters:
        t.start()
    while True:
        with lock:
            done = list(exits)
        failed = [e for e in done if e[2] != 0]
        if failed:
            time.sleep(0.2)                      # exits that came in the same breath are all in by now
            with lock:
                _, r, rc = min(e for e in exits if e[2] != 0)
            status = rc if rc > 0 else 128 - rc
            print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
        elif len(done) == n:
            break
        elif time.time() > deadline:
            status = 124
            print("bench.py: ranks did not finish in time; stopping them", file=sys.stderr)
        if status != 0:
            pending = [r for r in range(n) if procs[r].poll() is None]
            for r in pending:                    # exactly the processes started above, by PID
                procs[r].send_signal(signal.SIGTERM)
            for r in pending:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    for l in out_lines:
        if l.startswith("{"):
            line = l.strip()
        else:
            sys.stderr.write(l)
    if status == 0 and line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        status = 1
    if line is not None and status == 0:
        print(line, flush=True)
    sys.exit(status)


def world_info(dist, torch, world, rank, device, backend):
    """The `rccl` object of an N > 1 line: the backend the ranks talked over and which device every rank drove (all-gathered: index,
    PCI bus id, name), so that N distinct GPUs can be seen in the line itself."""
    if dist is None:
        return None
    p = torch.cuda.get_device_properties(device)
    bus = "%04x:%02x:%02x" % tuple(int(getattr(p, k, -1)) & 0xffff for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    mine = {"rank": rank, "device": int(device), "pci": bus, "name": p.name, "host": os.uname().nodename, "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            ver = None
    note = os.environ.get("VXRT_BENCH_BACKEND_NOTE")
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else "") + (f" ({note})" if note else ""), "rccl_version": ver, "world_size": world,
            "distinct_devices": len({(e["host"], e["pci"], e["device"]) for e in everyone}), "devices": everyone}


def init_dist(need_gpu=True):
    """-> (world, rank, device, dist, torch, backend).  N > 1: one process per GPU, torch.distributed over RCCL (backend "nccl");
    VXRT_BENCH_BACKEND=gloo for rehearsals with several ranks on one GPU.  If RCCL cannot be brought up (its first collective is
    made here, so a failure shows now) the ranks fall back to gloo and say so in the line's `rccl.backend`: the trace bench uses the
    process group for its barrier and two reductions only, and a number measured with a gloo barrier is worth more than no number;
    `--pipeline` then stages its halo through the host, which its line says as well.
    The ranks AGREE on the fall-back before any of them switches: a TCPStore on a second port — chosen by the parent,
    VXRT_BENCH_PORT2; MASTER_PORT + 1 under a foreign launcher — is opened first; after its RCCL attempt (bounded by a 120 s
    collective timeout, so that a rank whose peers have failed gets out of its all-reduce) every rank publishes ok / failed there and
    reads every rank's word.  All ok: RCCL.  All failed: the failed group is destroyed (aborted if it cannot be destroyed) and gloo
    is opened OVER THAT STORE — no third port, no second rendezvous.  Mixed: every rank says so and exits with status 3."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist, torch, device = None, None, local_rank
    backend = os.environ.get("VXRT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on one GPU
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        device = local_rank % max(torch.cuda.device_count(), 1)
        if need_gpu:
            torch.cuda.set_device(device)
        if backend == "nccl":
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            port2 = int(os.environ.get("VXRT_BENCH_PORT2") or int(os.environ.get("MASTER_PORT", "29500")) + 1)
            store = dist.TCPStore(addr, port2, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=300))
            err = None
            try:
                if os.environ.get("VXRT_BENCH_FAIL_NCCL") == "1" or os.environ.get("VXRT_BENCH_FAIL_NCCL_RANK") == str(rank):      # test hooks of the fall-back
                    raise RuntimeError("VXRT_BENCH_FAIL_NCCL=1")
                if os.environ.get("VXRT_BENCH_FAKE_NCCL_OK_RANK") == str(rank):   # test hook: this rank claims its RCCL came up (a mixed outcome)
                    raise StopIteration
                # (no device_id: with it the communicator is made inside init_process_group, and one that fails there — two ranks on
                # one GPU — leaves a half-made group that nobody can destroy and that warns about it at exit; made by the first
                # collective instead, it fails inside a registered group, which destroy_process_group below shuts down.  The device is
                # set above, so the collectives know where they run.)
                try:
                    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=float(os.environ.get("VXRT_BENCH_NCCL_TIMEOUT", "120"))))
                    store.set(f"pre_{rank}", "ok")
                except Exception:
                    store.set(f"pre_{rank}", "failed")
                    raise
                # nobody enters the first collective unless EVERY rank got this far: a rank whose init failed has said so,
                # and its peers raise here instead of sitting in the all-reduce until the collective timeout.  (A failure INSIDE the first
                # collective is symmetric in every case seen — two ranks on one device — and raises on all ranks; an asymmetric one would
                # still end after the timeout, possibly by the NCCL watchdog's abort: a non-zero status, nothing restarted.)
                pre = [store.get(f"pre_{r}").decode() for r in range(world)]
                if any(w != "ok" for w in pre):
                    raise RuntimeError(f"a peer could not initialise its RCCL process group ({pre})")
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over RCCL returned {probe.item()} for a world of {world}")
            except StopIteration:
                store.set(f"pre_{rank}", "ok")
            except Exception as e:  # noqa: BLE001
                err = e
                store.set(f"pre_{rank}", "failed")          # (the test hooks raise before the group is made: the peers must not wait for this word)
                print(f"bench.py rank {rank}: RCCL did not come up ({e!r})", file=sys.stderr, flush=True)
            store.set(f"nccl_{rank}", "failed" if err is not None else "ok")
            words = [store.get(f"nccl_{r}").decode() for r in range(world)]       # blocks until every rank has spoken
            if any(w != "ok" for w in words):
                if dist.is_initialized():            # the failed (or, on a mixed outcome, the healthy) RCCL group goes before anything else opens
                    try:
                        dist.destroy_process_group()
                    except Exception:  # noqa: BLE001 — a communicator that never came up may refuse an orderly shutdown
                        try:
                            dist.distributed_c10d._abort_process_group()
                        except Exception:  # noqa: BLE001
                            pass
                if any(w == "ok" for w in words):
                    print(f"bench.py rank {rank}: the ranks disagree about RCCL ({words}); giving up", file=sys.stderr, flush=True)
                    sys.exit(3)
                print(f"bench.py rank {rank}: every rank failed to bring RCCL up; falling back to gloo", file=sys.stderr, flush=True)
                dist.init_process_group("gloo", store=dist.PrefixStore("gloo_fallback", store), rank=rank, world_size=world)
                backend = "gloo"
                os.environ["VXRT_BENCH_BACKEND_NOTE"] = f"nccl failed: {err!r}"[:200]
        else:
            dist.init_process_group(backend)
    return world, rank, device, dist, torch, backend


def dry_run(args):
    """VXRT_BENCH_DRY=1: the ranks rendezvous, report the world they see and leave — no context, no kernel.  What the CPU suite uses
    to test the launcher (tests/test_distributed_cpu.py); not a benchmark."""
    world, rank, device, dist, torch, backend = init_dist(need_gpu=False)
    info = None
    if dist is not None:
        everyone = [None] * world
        dist.all_gather_object(everyone, {"rank": rank, "local_rank": device, "pid": os.getpid()})
        info = {"backend": backend, "world_size": world, "devices": everyone}
        if os.environ.get("VXRT_BENCH_DRY_FAIL_RANK") == str(rank):
            sys.exit(3)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rccl": info}), flush=True)


def reduce_max(dist, torch, dev, values):
    if dist is None:
        return values
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def reduce_sum(dist, torch, dev, values):
    if dist is None:
        return values
    t = torch.tensor(values, dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def per_rank_report(dist, world, rank, block_s, start_ns, end_ns, extra=None):
    """N > 1: what every rank saw of the timed blocks, gathered AFTER the timed region (nothing is exchanged inside it), so that the first run on
    a real multi-GPU node explains itself: per rank the median / min / max of its own block time (barrier ->
    its own queues drained), and per block how far apart the ranks LEFT the barrier (launch skew; CLOCK_MONOTONIC is shared by the
    processes of one host) and how far apart they finished.  -> dict on rank 0, None elsewhere."""
    if dist is None:
        return None
    mine = {"rank": rank, "block_s": [float(v) for v in block_s], "start_ns": [int(v) for v in start_ns], "end_ns": [int(v) for v in end_ns]}
    if extra:
        mine.update(extra)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank != 0:
        return None
    everyone.sort(key=lambda e: e["rank"])
    nb = min(len(e["block_s"]) for e in everyone)
    med = lambda xs: float(sorted(xs)[len(xs) // 2]) if xs else None  # noqa: E731
    skew = [(max(e["start_ns"][b] for e in everyone) - min(e["start_ns"][b] for e in everyone)) * 1e-6 for b in range(nb)]
    spread = [(max(e["end_ns"][b] for e in everyone) - min(e["end_ns"][b] for e in everyone)) * 1e-6 for b in range(nb)]
    last = [max(range(len(everyone)), key=lambda r: everyone[r]["end_ns"][b]) for b in range(nb)]
    ranks = []
    for e in everyone:
        row = {"rank": e["rank"], "block_ms": {"median": round(med(e["block_s"]) * 1e3, 4), "min": round(min(e["block_s"]) * 1e3, 4), "max": round(max(e["block_s"]) * 1e3, 4)},
               "blocks_it_finished_last": sum(1 for r in last if r == e["rank"])}
        for k, v in e.items():
            if k not in ("rank", "block_s", "start_ns", "end_ns"):
                row[k] = v
        ranks.append(row)
    return {"blocks": nb, "launch_skew_after_the_barrier_ms": {"median": round(med(skew), 4), "max": round(max(skew), 4)},
            "finish_spread_ms": {"median": round(med(spread), 4), "max": round(max(spread), 4)}, "ranks": ranks,
            "note": "block_ms: a rank's own clock from leaving the barrier to its own queues being empty; the line's ms_per_step is the max over ranks, block by block"}


def block_count(est_block_s, asked):
    """>= 50 blocks and >= 1 s of GPU time (>= 0.25 s when asked for fewer blocks than that needs would take minutes)."""
    if asked > 0:
        return asked
    return int(min(4000, max(50, np.ceil(1.0 / max(est_block_s, 1e-5)))))


def trace_bench(args):
    world, rank, device, dist, torch, backend = init_dist()
    args.gpus = world          # a launcher's WORLD_SIZE decides (main() starts the ranks itself when there is none)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    band_rows, deal_tracer = pick_deal(world, args.steps, args.inflight, args.batch)
    args.inflight, args.batch = pick_schedule(world, args.steps, args.inflight, args.batch)
    if args.tracer == 0:
        args.tracer = deal_tracer
    if args.band_rows:
        band_rows = args.band_rows
    from gpu_voxel_raytracer_amd import Camera, Context, TIMED, TRACE, scenes

    pos, mrgb, size = scenes.load_scene(SCENE)
    cam = scenes.close_camera(size) if args.view == "close" else scenes.bench_camera(size)
    if args.view == "away":   # diagnostic: every primary ray misses (pure G-buffer write traffic)
        cam = (cam[0], -cam[1], cam[2])

    ctx = Context(WIDTH, HEIGHT, device=device, max_bounces=args.bounces, rank=rank, nranks=world, band_rows=band_rows,
                  frames_in_flight=args.inflight, frames_per_launch=args.batch, tracer=args.tracer)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ctx.render_frames(TRACE, args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.render_frames(TRACE, args.steps)        # one untimed block: sizes the number of repetitions
    ctx.sync()
    est = reduce_max(dist, torch, red_dev, [time.perf_counter() - t0])[0]
    blocks = block_count(est, args.blocks)

    times, rays_blk, kernel_ms, launches, frames_timed, local_px = [], [], 0.0, 0, 0, 0
    own_s, start_ns, end_ns = [], [], []
    for _ in range(blocks):
        ctx.reset_stats()
        barrier()
        start_ns.append(time.monotonic_ns())
        t0 = time.perf_counter()
        ctx.render_frames(TRACE | TIMED, args.steps)   # K steps = K frames, submitted back to back
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        end_ns.append(time.monotonic_ns())
        own_s.append(elapsed)
        st = ctx.stats()
        elapsed = reduce_max(dist, torch, red_dev, [elapsed])[0]
        rays = reduce_sum(dist, torch, red_dev, [st.rays])[0]
        times.append(elapsed)
        rays_blk.append(rays)
        kernel_ms += st.trace_ms
        launches += st.timed_launches
        frames_timed += st.timed_frames
        local_px = st.pixels // max(args.steps, 1)
    scene_bytes = st.scene_bytes
    # primary rays the sky cull answers without a walk (counted in `rays`: each is one cast_bounded_ray of the shader), all ranks
    culled = reduce_sum(dist, torch, red_dev, [ctx.culled_pixels()])[0]
    rccl = world_info(dist, torch, world, rank, device, backend)
    per_rank = per_rank_report(dist, world, rank, own_s, start_ns, end_ns, {"local_rows": int(ctx.stats().local_rows), "rays_per_block": int(st.rays)})

    if rank == 0:
        order = np.argsort(times)
        mid = int(order[len(order) // 2])
        elapsed, rays = times[mid], rays_blk[mid]            # the median block
        launch_ms = kernel_ms / max(launches, 1)
        frames_per_launch = frames_timed / max(launches, 1)
        alg = algorithmic_bytes(local_px, args.bounces, scene_bytes)       # per frame, this rank
        wall = alg * args.steps / elapsed / 1e9                          # GB/s the block's wall time amounts to
        conc = max(1, min(args.inflight, launches // max(blocks, 1)))
        prof, prof_path = recorded_profile()
        default_cfg = world == 1 and args.view == "bench" and args.bounces == 4 and args.frame == "1080p"
        roof = {
            # what the profile shows: VALU issue, not bandwidth, limits this stage (the `valu` object below).  The HBM
            # figures stay because the path is nominally HBM-bound work (no contraction, no MFMA): achieved = algorithmic bytes of
            # the K steps / the block's wall time; frac follows from wall time and nothing else.
            "bound": "hbm", "limited_by": "valu issue (see valu)",
            "kernel": "trace_kernel, all-in-one (one trace stage)" if args.tracer == 1 else "trace_kernel + bounce_kernel (one trace stage)",
            "achieved": round(wall, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(wall / HBM_PEAK_GBS, 5),
            "algorithmic_bytes_per_step": int(alg),
            "traffic": None, "traffic_source": None,
            # diagnostic, launch basis: HIP-event time around one launch (trace_kernel + bounce_kernel over frames_per_launch
            # frames); `concurrent_launches` of them overlap, so bytes / launch_ms understates and x concurrency overstates
            "launch": {"launch_ms": round(launch_ms, 4), "frames_per_launch": round(frames_per_launch, 2),
                       "concurrent_launches": conc, "algorithmic_bytes_per_launch": int(alg * frames_per_launch),
                       "gbs_one_launch_alone": round(alg * frames_per_launch / (launch_ms * 1e-3) / 1e9, 2) if launch_ms > 0 else None},
        }
        live = getattr(args, "live_counters", None)
        if live is not None:
            # counters of THIS invocation (child passes under rocprofv3, made before this process touched the GPU)
            roof["traffic"] = live["hbm_bytes_per_step_read_doubled"]
            roof["traffic_raw"] = live["hbm_bytes_per_step_raw"]
            roof["traffic_over_algorithmic"] = round(live["hbm_bytes_per_step_read_doubled"] / alg, 3)
            roof["traffic_source"] = ("MEASURED in this invocation: " + live["command"] + f" ({live['frames_per_launch']} frames per launch; separate "
                                      "FETCH_SIZE and WRITE_SIZE passes, read side doubled per MI355X_MICROARCH.md, `traffic_raw` as counted), per step")
            per_frame = live["valu_wave_instr_per_step"]
            roof["valu"] = {"source": "MEASURED in this invocation (the SQ_* pass of the same child command)",
                            "valu_wave_instr_per_step": round(per_frame),
                            "issue_slot_frac": round(per_frame * 2 / (1024 * 2.4e9 * (elapsed / args.steps)), 3),
                            "lane_utilisation": {k: v["lane_utilisation"] for k, v in live["per_kernel"].items()},
                            "waitcnt_share": {k: v["waitcnt_share_of_wave_cycles"] for k, v in live["per_kernel"].items()},
                            "per_kernel": live["per_kernel"], "ms_per_step_under_the_profiler": live["ms_per_step_under_the_profiler"]}
        elif prof is not None and default_cfg:
            roof["counters_note"] = getattr(args, "live_counters_note", None)
            drv = prof.get("driver_schedule") or {}
            t, same = prof.get("traffic"), False
            try:   # the recorded pass whose schedule is nearest to this run's (frames per launch)
                t = min([t for t in (prof.get("traffic"), drv.get("traffic")) if t], key=lambda t: abs(float(t["frames_per_launch"]) - frames_per_launch))
                same = t is drv.get("traffic")
                roof["traffic"] = float(t["hbm_bytes_per_launch_corrected"]) / float(t["frames_per_launch"])
                roof["traffic_source"] = (f"RECORDED in {prof_path} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, read side doubled per "
                                          f"MI355X_MICROARCH.md) with the schedule `{t.get('command_args', '--steps 96 --warmup 32 --blocks 2')}` "
                                          f"({t['frames_per_launch']} frames per launch), scaled per step; not measured by this run")
            except (KeyError, ValueError, TypeError, ZeroDivisionError):
                pass
            try:
                sq = drv["sq"] if (same and drv.get("sq")) else prof["sq"]
                per_frame = sum(k["valu_wave_instr_per_launch"] for k in sq.values()) / float(t["frames_per_launch"])
                # 1024 SIMDs; a wave64 VALU instruction occupies its SIMD's issue port for 2 cycles at ~2.4 GHz
                roof["valu"] = {"source": f"RECORDED in {prof_path} (rocprofv3 --pmc SQ_* pass), not measured by this run",
                                "valu_wave_instr_per_step": round(per_frame),
                                "issue_slot_frac": round(per_frame * 2 / (1024 * 2.4e9 * (elapsed / args.steps)), 3),
                                "lane_utilisation": {k: round(v["lane_utilisation"], 3) for k, v in sq.items() if v.get("lane_utilisation")},
                                "waitcnt_share": {k: round(v["waitcnt_share_of_wave_cycles"], 3) for k, v in sq.items()
                                                  if v.get("waitcnt_share_of_wave_cycles")}}
            except (KeyError, ValueError, TypeError, ZeroDivisionError):
                pass
        out = {
            "metric": "Mrays/s", "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": DATA_LABEL,
            "config": {"workload": f"vox/{SCENE}.vox {WIDTH}x{HEIGHT}, 1 spp, {args.bounces} bounces, trace stage only "
                                   f"(BASELINE configs[1]); camera '{args.view}' of SURVEY §8d; Uniforms::default()",
                       "parallelism": f"screen bands x{world} ({band_rows}-row interleave, scene replicated)",
                       "launches_in_flight": args.inflight, "frames_per_launch": args.batch,
                       "tracer": {0: "head + compacted tail (trace_kernel + bounce_kernel)", 1: "all-in-one trace_kernel"}.get(args.tracer, str(args.tracer)),
                       "rays_per_frame": rays // args.steps, "rays_per_pixel": round(rays / args.steps / (WIDTH * HEIGHT), 4),
                       "rays_walked_per_frame": rays // args.steps - culled, "primary_rays_answered_by_the_sky_cull_per_frame": culled,
                       "mpixels_per_s": round(WIDTH * HEIGHT * args.steps / elapsed / 1e6, 1)},
            "timing": {"blocks": blocks, "steps_per_block": args.steps, "reported": "median block",
                       "block_ms": {"min": round(min(times) * 1e3, 4), "median": round(elapsed * 1e3, 4), "max": round(max(times) * 1e3, 4)},
                       "timed_region_s_total": round(sum(times), 3),
                       "note": "ms_per_step is reciprocal THROUGHPUT (frames of a camera at rest, several per launch, launches "
                               "overlapping); latency_ms_one_frame_at_a_time is one vxrt_render(TRACE) per frame, each waited for"},
            "roofline": roof,
        }
        if rccl is not None:
            out["rccl"] = rccl
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_extras:
            # secondary figures, measured in this run after the headline block (none of them is `value`)
            out["timing"]["latency_ms_one_frame_at_a_time"] = round(measure_latency(Context, Camera, TRACE, pos, mrgb, cam, device, args.bounces), 4)
            extra = {}
            if args.view == "bench":
                # blocks of 480 frames: the steady-state schedule, whatever --steps made of the headline's
                view_inflight, view_batch = pick_schedule(world, 480)
                extra["close_view"] = dict(measure_view(Context, Camera, TRACE, pos, mrgb, scenes.close_camera(size), device, args.bounces,
                                                        view_inflight, view_batch),
                                           workload=f"vox/{SCENE}.vox {WIDTH}x{HEIGHT}, {args.bounces} bounces, camera 'close' (geometry fills the frame); "
                                                    f"{view_batch} frames per launch x {view_inflight} launches in flight")
            if default_cfg and args.frame == "1080p":
                # north_star asks for 1080p AND 4K frames: the same scene, camera and schedule at 3840x2160 (`--frame 4k` makes it the headline)
                m4 = measure_view(Context, Camera, TRACE, pos, mrgb, cam, device, args.bounces, view_inflight, view_batch, width=3840, height=2160, frames=240, blocks=5)
                alg4 = algorithmic_bytes(3840 * 2160, args.bounces, m4["scene_bytes"])
                m4["roofline"] = {"bound": "hbm", "limited_by": "valu issue", "algorithmic_bytes_per_frame": int(alg4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "achieved": round(alg4 / (m4["ms_per_frame"] * 1e-3) / 1e9, 1), "frac": round(alg4 / (m4["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                m4["workload"] = f"vox/{SCENE}.vox 3840x2160, 1 spp, {args.bounces} bounces, trace stage, camera 'bench'; {view_batch} frames per launch x {view_inflight} launches in flight"
                extra["menger_4k"] = m4
            if default_cfg and not args.no_config5:
                extra.update(measure_config5(Context, Camera, TRACE, scenes, device, getattr(args, "live_probes", {})))
            if default_cfg and not args.no_config3:
                try:
                    from gpu_voxel_raytracer_amd import ALL
                    extra["config3_pipeline"] = measure_config3(Context, Camera, ALL, TIMED, scenes, device,
                                                                live_valu=getattr(args, "live_probes", {}).get("denoise_r8_valu_wave_instr_per_launch"))
                except Exception as e:  # noqa: BLE001
                    extra["config3_pipeline"] = {"error": repr(e)}
                try:
                    from gpu_voxel_raytracer_amd import DENOISE_EDGE, DENOISE_INTERIOR, TEMPORAL
                    extra["config4_one_rank_of_8"] = measure_config4_rank(Context, Camera, (TRACE, TEMPORAL, TIMED, DENOISE_INTERIOR, DENOISE_EDGE), scenes, device)
                except Exception as e:  # noqa: BLE001
                    extra["config4_one_rank_of_8"] = {"error": repr(e)}
            if default_cfg:
                try:
                    from gpu_voxel_raytracer_amd import ALL as _ALL
                    extra["reference_loop"] = measure_reference_loop(Context, Camera, (_ALL, TIMED), scenes, device)
                except Exception as e:  # noqa: BLE001
                    extra["reference_loop"] = {"error": repr(e)[:300]}
                try:
                    extra["parity_check"] = parity_check(Context, Camera, TRACE, scenes, pos, mrgb, cam, device)
                except Exception as e:  # noqa: BLE001
                    extra["parity_check"] = {"error": repr(e)}
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, mrgb, cam)
            out["cpu_baseline_cpu_rs"] = cpu_baseline_cpu_rs()
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def parity_check(Context, Camera, TRACE, scenes, pos, mrgb, cam, device):
    """The frame this benchmark times (menger 1920x1080, 4 bounces, frame 1) rendered once more and hashed slab by slab against what the
    REFERENCE'S COMPILED voxels.comp gives for it (tests/golden/spirv_exec/full_size.json: the module executed instruction by instruction
    where the reference is mounted; data here).  Says in the line itself that the timed kernels compute the reference's frame."""
    import hashlib
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "spirv_exec", "full_size.json")
    with open(path) as f:
        case = next(c for c in json.load(f)["cases"] if c["scene"] == SCENE and (c["w"], c["h"]) == (1920, 1080))

    def digest(img, node_channel):
        a = np.ascontiguousarray(img, np.float32)
        v = a.view(np.uint32).copy()
        keep = v[..., 3].copy()
        v[np.isnan(a)] = 0x7fc00000          # what IEEE leaves open, made canonical (tests/spirv_pipeline.py: canonical_sha256)
        v[v == 0x80000000] = 0
        if node_channel:
            v[..., 3] = keep                 # the albedo image's .w is the leaf word, an integer
        return hashlib.sha256(v.tobytes()).hexdigest()
    with Context(1920, 1080, device=device, max_bounces=case["bounces"]) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.render(TRACE)
        rows, slabs, bad = case["slab_rows"], 0, []
        for key, img in (("color", 0), ("nd", 1), ("albedo", 2)):
            a = ctx.read(img)
            for s, want in enumerate(case["sha256"][key]):
                slabs += 1
                if digest(a[s * rows:(s + 1) * rows], key == "albedo") != want:
                    bad.append(f"{key} rows {s * rows}-{(s + 1) * rows}")
    return {"against": "the reference's compiled shaders/voxels.comp.spv executed by oracle/ospirv.cpp (fixture tests/golden/spirv_exec/full_size.json)",
            "frame": f"vox/{SCENE}.vox 1920x1080, {case['bounces']} bounces, frame 1: colour, normal/depth, albedo/node images", "slabs_hashed": slabs,
            "bit_exact": not bad, "differing": bad}


def pipeline_bench(args):
    """BASELINE configs[3]: vox/castle.vox at 3840x2160, 4 spp, 8 bounces, screen bands over the ranks, temporal + denoise (r = 8) with
    the halo exchange, as gpu_voxel_raytracer_amd/distributed.py runs it: per displayed frame

        render_spp(TRACE | TEMPORAL, 4) -> halo pack -> [two sends + two receives] -> DENOISE_INTERIOR -> halo unpack -> DENOISE_EDGE

    with band_rows chosen by distributed.band_rows_for (48 or 64 rows at r = 8: the halo is a third or a quarter of a rank's rows, whichever
    leaves the busiest rank fewer rows) and everything ordered by events.  One step = one
    displayed frame.  Reported apart: halo bytes per rank and frame, the pack / unpack kernel times (HIP events), and — from a second,
    synchronised pass in which nothing overlaps — the time of the exchange alone and the frame time without overlap."""
    world, rank, device, dist, torch, backend = init_dist()
    args.gpus = world
    red_dev = "cuda" if backend == "nccl" else "cpu"
    from gpu_voxel_raytracer_amd import DENOISE, TEMPORAL, TIMED, TRACE, Camera, Context, distributed, scenes
    w, h, bounces, spp, radius = 3840, 2160, 8, 4, args.radius
    band = args.band_rows or distributed.band_rows_for(radius, h, world)
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    # two trace streams: the next displayed frame's trace launch runs beside this frame's exchange and denoise (nothing in the loop
    # waits on the host when the messages travel over RCCL)
    ctx = Context(w, h, device=device, max_bounces=bounces, rank=rank, nranks=world, band_rows=band, frames_in_flight=max(args.inflight, 1) if args.inflight else 2,
                  frames_per_launch=spp)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)
    ctx.denoise_uniforms.radius = radius
    if world > 1 and torch is None:
        import torch
    halo = None
    if world > 1:   # halo buffers on this rank's GPU; messages over RCCL, or staged through the host for a gloo rehearsal
        halo = distributed.HaloExchange(ctx, dist, rank, world, torch.device("cuda", device), torch,
                                        comm_device=None if backend == "nccl" else "cpu")
    xchg = [0.0]

    def frame(overlap=True):
        ctx.render_spp(TRACE | TEMPORAL | TIMED, spp)
        if world > 1 and not overlap:      # diagnostic: every part waits for the one before, the exchange on the host clock
            ctx.sync()
            t0 = time.perf_counter()
            halo.exchange()
            ctx.sync()
            xchg[0] += time.perf_counter() - t0
            ctx.render_stage(DENOISE | TIMED)
        else:
            distributed.finish_frame(ctx, world, radius, halo, overlap=True, extra_flags=TIMED)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    def timed_block(overlap):
        for _ in range(args.warmup):
            frame(overlap)
        barrier()
        ctx.reset_stats()
        xchg[0] = 0.0
        barrier()
        own["start_ns"].append(time.monotonic_ns())
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame(overlap)
        ctx.sync()
        if dist is not None and backend == "nccl":
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        own["end_ns"].append(time.monotonic_ns())
        own["block_s"].append(elapsed)
        own["exchange_s"].append(xchg[0])
        st = ctx.stats()
        elapsed, x = reduce_max(dist, torch, red_dev, [elapsed, xchg[0]])
        rays = reduce_sum(dist, torch, red_dev, [st.rays])[0]
        return elapsed, x, rays, st

    own = {"start_ns": [], "end_ns": [], "block_s": [], "exchange_s": []}
    elapsed, _, rays, st = timed_block(True)
    sync_elapsed, x, _, sync_st = timed_block(False) if world > 1 else (elapsed, 0.0, rays, st)
    info = ctx.halo_info()
    rccl = world_info(dist, torch, world, rank, device, backend)
    # per rank: block 0 = the overlapped loop, block 1 = the synchronous pass, whose exchange time on this rank's host
    # clock (pack + both messages on the wire + unpack, nothing overlapping) is the halo's wire time as this rank sees it
    per_rank = per_rank_report(dist, world, rank, own["block_s"], own["start_ns"], own["end_ns"],
                               {"local_rows": int(st.local_rows), "halo_exchange_ms_synchronous": round(own["exchange_s"][-1] / args.steps * 1e3, 4),
                                "halo_pack_ms": round(st.halo_pack_ms / max(st.halo_exchanges, 1), 5), "halo_unpack_ms": round(st.halo_unpack_ms / max(st.halo_exchanges, 1), 5),
                                "trace_ms_per_frame": round(st.trace_ms / args.steps, 4), "denoise_ms_per_frame": round(st.denoise_ms / args.steps, 4)})
    if rank == 0:
        px = w * h
        alg = (48 * spp + 16 * spp + 16 + 80 + 64) * px     # spp trace frames + their average + temporal + denoise
        out = {"metric": "Mrays/s", "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": DATA_LABEL.replace("menger", "castle"),
               "config": {"workload": f"vox/castle.vox {w}x{h}, {spp} spp, {bounces} bounces, temporal + denoise r={radius} "
                                      f"(BASELINE configs[3]); one step = one displayed frame",
                          "parallelism": f"screen bands x{world} ({band}-row interleave, scene replicated), halo over "
                                         f"{'RCCL send/recv' if backend == 'nccl' else backend + ' (staged through pinned host memory: a rehearsal)'}"
                                         f", overlapped with the denoise of the interior tiles"},
               "halo": {"band_rows": band, "rows": int(info.rows), "bytes_per_pixel": int(info.bytes_per_pixel),
                        "bytes_per_rank_per_frame": 2 * int(info.message_bytes) if world > 1 else 0,
                        "interior_tile_rows": int(info.interior_tile_rows), "edge_tile_rows": int(info.edge_tile_rows),
                        "pack_ms": round(st.halo_pack_ms / max(st.halo_exchanges, 1), 5),
                        "unpack_ms": round(st.halo_unpack_ms / max(st.halo_exchanges, 1), 5),
                        "exchange_ms_synchronous": round(x / args.steps * 1e3, 4),
                        "ms_per_step_synchronous": round(sync_elapsed / args.steps * 1e3, 4),
                        "note": "pack_ms / unpack_ms: HIP events around the one pack / unpack kernel of an exchange, this rank.  "
                                "exchange_ms_synchronous: host clock around pack + messages + unpack in a second pass where every part "
                                "waits for the one before (ms_per_step_synchronous); ms_per_step is the overlapped loop."},
               "stage_ms_per_frame": {"trace": round(st.trace_ms / args.steps, 4), "temporal": round(st.temporal_ms / args.steps, 4),
                                      "denoise": round(st.denoise_ms / args.steps, 4),
                                      "denoise_synchronous_pass": round(sync_st.denoise_ms / args.steps, 4),
                                      "note": "HIP-event durations of each stage's launches; the next frame's trace launch runs on a stream of "
                                              "its own beside this frame's denoise, so they are longer than alone and do not add up to ms_per_step"},
               "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": "whole frame loop", "achieved": round(alg * args.steps / elapsed / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5), "traffic": None}}
        if rccl is not None:
            out["rccl"] = rccl
        if per_rank is not None:
            out["per_rank"] = per_rank
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="frames per timed block (default 960: whole launches for every schedule, 2x16 .. 3x32; "
                                                          "--pipeline: 24 displayed frames)")
    ap.add_argument("--warmup", type=int, default=-1)
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks of --steps frames (default: >= 50 and >= 1 s of GPU time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (latency, close view, config 5's scene)")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra that builds the 5.6 GB procedural scene")
    ap.add_argument("--no-counters", action="store_true", help="skip the three rocprofv3 child passes that measure this invocation's HBM bytes and SQ counters "
                                                                "(the line then quotes the RECORDED values of profiles/)")
    ap.add_argument("--probe", default="", choices=["", "config5", "config3"], help=argparse.SUPPRESS)     # a child of live_probes
    ap.add_argument("--no-config3", action="store_true", help="skip the extra that times BASELINE configs[2]'s frame loop at 4K")
    ap.add_argument("--pipeline", action="store_true", help="time the whole frame loop with the denoise halo exchange (BASELINE configs[3])")
    ap.add_argument("--radius", type=int, default=8, help="--pipeline: denoise radius")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band (trace bench: 8, 4 for a short block from 8 ranks on; --pipeline: >= 8 radius, a multiple of 16)")
    ap.add_argument("--view", default="bench", choices=["bench", "close", "away"])
    ap.add_argument("--bounces", type=int, default=BOUNCES, help="diagnostic only; the benchmark is 4")
    ap.add_argument("--tracer", type=int, default=0, help="diagnostic only (vxrt_config.tracer; 2 / 3 / 5 need the variants build); the benchmark is 0")
    ap.add_argument("--inflight", type=int, default=0,
                    help="trace launches that may be on the GPU together, one HIP stream each (default: 2 on one GPU, 3 per rank otherwise)")
    ap.add_argument("--batch", type=int, default=0,
                    help="consecutive frames per trace launch (vxrt_config.frames_per_launch; default: 16, 32 from 4 ranks on; a short block "
                         "is dealt to the launches in equal parts)")
    ap.add_argument("--frame", default="1080p", choices=["1080p", "4k"],
                    help="frame size of the trace bench: 1920x1080 (BASELINE configs[1], the headline) or 3840x2160 (north_star: 'Mrays/s on 1080p / 4K frames')")
    args = ap.parse_args()
    if args.frame == "4k":
        global WIDTH, HEIGHT
        WIDTH, HEIGHT = 3840, 2160
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])      # does not return
    if os.environ.get("VXRT_BENCH_DRY") == "1":
        return dry_run(args)
    if args.probe:
        return run_probe(args.probe)
    args.live_counters, args.live_counters_note, args.live_probes = None, "not asked for", {}
    if (not args.pipeline and "WORLD_SIZE" not in os.environ and args.gpus == 1 and not args.no_counters and not args.no_extras and args.tracer == 0
            and not any(k.startswith("ROCPROF") for k in os.environ) and "rocprofiler" not in os.environ.get("LD_PRELOAD", "")):   # not under a profiler already
        # BEFORE this process makes its first HIP call: the profiler passes run as children (no exec from a process that has the GPU)
        a = argparse.Namespace(**vars(args))
        a.steps = a.steps or 960
        a.warmup = 96 if a.warmup < 0 else a.warmup
        args.live_counters, args.live_counters_note = live_counters(a)
        if args.view == "bench" and args.bounces == 4 and args.frame == "1080p" and not args.no_config5 and not args.no_config3:
            args.live_probes = live_probes()       # whatever --steps is (round 4 skipped them for the driver's short block and quoted recorded counters)
    if args.pipeline:
        args.steps = args.steps or 24
        args.warmup = 4 if args.warmup < 0 else args.warmup
        pipeline_bench(args)
    else:
        args.steps = args.steps or 960
        args.warmup = 96 if args.warmup < 0 else args.warmup
        trace_bench(args)


if __name__ == "__main__":
    main()
