#!/usr/bin/env python3
"""bench.py — headline benchmark of the vxrt hot path on MI355X.

Workload (BASELINE.json configs[1]): vox/menger.vox (81^3 Menger sponge, 160 000
voxels; voxel-list fixture tests/golden/scenes/menger.npz), 1920x1080, 1 sample per pixel per frame,
MAX_BOUNCES = 4, path-trace stage only (no temporal / denoise), fixed camera, Uniforms::default().
A "step" is one frame (frame_number advances every step, so every step draws different noise).

Metric: Mrays/s, where a ray is one cast_bounded_ray invocation (primary, bounce or sun shadow ray),
counted exactly on the device.  value = rays cast by all ranks / max-over-ranks wall time of the K
timed steps; everything the timed region touches is resident in HBM before it starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the frame's rows are dealt to the ranks in interleaved 8-row bands (BAND_ROWS)
(scene and noise table replicated, no data-path collective for this stage), so total work is fixed:
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
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Frames in flight run on separate HIP streams; the ROCm runtime maps streams onto 4 hardware queues unless told
# otherwise, which would cap the overlap at 4 kernels.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

WIDTH, HEIGHT, BOUNCES, SCENE = 1920, 1080, 4, "menger"
# Rows are dealt to the ranks in interleaved bands of this height.  8 = the tracer's tile height: at 8 ranks the band sets' costs are
# within 0.0161-0.0179 ms per frame of each other, with 16-row bands 0.0153-0.0206 (1080 rows are 67.5 such bands, and the sponge's
# structure beats against the 128-row period).  (A denoise radius > 0 needs 16-row bands; this benchmark is the trace stage.)
BAND_ROWS = 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(pixels, bounces, scene_bytes):
    """48 B/px written (3 x rgba32f) + the scene once + the noise layers one frame touches
    (8 rand() layers per bounce at most, 64 KiB each)."""
    return 48 * pixels + scene_bytes + min(8 * bounces, 512) * 128 * 128 * 4


def measured_traffic(frames_per_launch):
    """HBM bytes per trace-stage launch (trace_kernel + bounce_kernel over frames_per_launch frames) from the rocprofv3 PMC
    passes recorded in profiles/ (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be collected from inside this
    process), scaled to this run's frames per launch; None if absent."""
    path = os.path.join(ROOT, "profiles", "r01", "trace_stage_summary.json")
    try:
        t = json.load(open(path))["traffic"]
        return float(t["hbm_bytes_per_launch_corrected"]) / float(t["frames_per_launch"]) * frames_per_launch
    except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def cpu_baseline(pos, mrgb, cam, target_seconds=float(os.environ.get("VXRT_BENCH_CPU_SECONDS", "12"))):
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


def pick_schedule(world, steps, inflight=0, batch=0):
    """Launches in flight and frames per launch for `world` ranks and a run of `steps` frames (0 = choose).
    A frame's longest tile is a serial chain of ~0.15-0.3 ms however few rows a rank owns, so a rank needs that much work in
    flight: 16-32 frames per launch, and the more launches overlapping the smaller its share of the frame (measured per rank with
    scripts/exp_rank_emulation.py: 2x16 / 3x16 / 3x32 / 3x32 for 1 / 2 / 4 / 8 ranks).  Never more than 3: with the context's own
    stream that makes 4, the number of hardware queues a process gets by default (GPU_MAX_HW_QUEUES) — a 5th stream shares a queue
    with another one and its launches serialise behind that one's (scripts/exp_first_context.py: 0.0236 vs 0.0181 ms per frame for
    one of 8 ranks).  Short runs get smaller launches so that the pipeline still holds a few of them."""
    if inflight <= 0:
        inflight = 2 if world == 1 else 3
    if batch <= 0:
        batch = 16 if world <= 2 else 32
        while batch > 1 and batch * inflight * 2 > max(steps, 1):
            batch //= 2
    return inflight, batch


def main():
    global BOUNCES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960, help="frames timed (default 960: whole launches for every schedule, 2x16 .. 3x32)")
    ap.add_argument("--warmup", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--view", default="bench", choices=["bench", "close", "away"])
    ap.add_argument("--bounces", type=int, default=BOUNCES, help="diagnostic only; the benchmark is 4")
    ap.add_argument("--inflight", type=int, default=0,
                    help="trace launches that may be on the GPU together, one HIP stream each (default: 2 on one GPU, 3 per rank otherwise)")
    ap.add_argument("--batch", type=int, default=0,
                    help="consecutive frames per trace launch (vxrt_config.frames_per_launch; default: 16, 32 from 4 ranks on, fewer for short runs)")
    args = ap.parse_args()
    BOUNCES = args.bounces

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    device = local_rank
    backend = os.environ.get("VXRT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on one GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        device = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    args.inflight, args.batch = pick_schedule(world, args.steps, args.inflight, args.batch)
    from gpu_voxel_raytracer_amd import Camera, Context, TIMED, TRACE, scenes

    pos, mrgb, size = scenes.load_scene(SCENE)
    cam = scenes.close_camera(size) if args.view == "close" else scenes.bench_camera(size)
    if args.view == "away":   # diagnostic: every primary ray misses (pure G-buffer write traffic)
        cam = (cam[0], -cam[1], cam[2])

    ctx = Context(WIDTH, HEIGHT, device=device, max_bounces=BOUNCES, rank=rank, nranks=world, band_rows=BAND_ROWS,
                  frames_in_flight=args.inflight, frames_per_launch=args.batch)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ctx.render_frames(TRACE, args.warmup)
    barrier()
    ctx.reset_stats()
    barrier()
    t0 = time.perf_counter()
    ctx.render_frames(TRACE | TIMED, args.steps)   # K steps = K frames, submitted back to back
    ctx.sync()
    if dist is not None:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    rays, kernel_ms, local_px = st.rays, st.trace_ms, st.pixels // max(args.steps, 1)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=red_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = int(r.item())

    if rank == 0:
        # Roofline of the trace stage = trace_kernel (every pixel up to its second hit) + bounce_kernel (the paths still
        # alive there, compacted), back to back on one stream, covering `frames_per_launch` consecutive frames;
        # launch_ms is the HIP-event time around the pair (rocprofv3's two average durations add up to it).  With S
        # launches in flight S such pairs overlap on the GPU, so a pair's own duration is ~S x the time the chip
        # spends on it: achieved = S x algorithmic bytes per launch / average pair duration.
        launches = max(st.timed_launches, 1)
        launch_ms = kernel_ms / launches
        frames_per_launch = st.timed_frames / launches
        alg = algorithmic_bytes(local_px, BOUNCES, st.scene_bytes)       # per frame
        conc = max(1, min(args.inflight, launches))
        achieved = conc * frames_per_launch * alg / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s", "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"vox/{SCENE}.vox {WIDTH}x{HEIGHT}, 1 spp, {BOUNCES} bounces, trace stage only "
                                   f"(BASELINE configs[1]); camera '{args.view}' of SURVEY §8d; Uniforms::default()",
                       "parallelism": f"screen bands x{world} ({BAND_ROWS}-row interleave, scene replicated)",
                       "launches_in_flight": args.inflight, "frames_per_launch": args.batch,
                       "rays_per_frame": rays // args.steps, "rays_per_pixel": round(rays / args.steps / (WIDTH * HEIGHT), 4),
                       "mpixels_per_s": round(WIDTH * HEIGHT * args.steps / elapsed / 1e6, 1)},
            "roofline": {"bound": "hbm", "kernel": "trace_kernel + bounce_kernel (one trace stage)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(frames_per_launch) if (world == 1 and args.view == "bench" and BOUNCES == 4) else None,
                         "launch_ms": round(launch_ms, 4), "concurrent_launches": conc,
                         "frames_per_launch": round(frames_per_launch, 2),
                         "algorithmic_bytes_per_launch": int(alg * frames_per_launch),
                         "achieved_wall_basis": round(alg * args.steps / elapsed / 1e9, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, mrgb, cam)
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
