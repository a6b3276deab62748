"""Round 5: fused head + tail (VXRT_OPT_FUSED_TAIL) against the two-kernel launch — same frames bit for bit, and the time of a block.
usage: python scripts/exp_fused_check.py   (prints as it goes; every wait inside the kernel is bounded)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
from gpu_voxel_raytracer_amd import host
from gpu_voxel_raytracer_amd.host import OPT_FUSED_TAIL, OPT_TRACER_OVERRIDE, VxrtError
host.use_library(host.variants_library())          # fused_kernel lives in the -DVXRT_VARIANTS=1 build

def frames(scene, w, h, cam, bounces, infl, batch, n, fused, rank=0, nranks=1, band=8):
    pos, mrgb, size = scenes.load_scene(scene)
    c = getattr(scenes, cam + "_camera")(size)
    with Context(w, h, max_bounces=bounces, rank=rank, nranks=nranks, band_rows=band, frames_in_flight=infl, frames_per_launch=batch,
                 tuning=[(OPT_TRACER_OVERRIDE, 4), (OPT_FUSED_TAIL, 1 if fused else 0)]) as ctx:
        ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*c)
        t0 = time.perf_counter()
        ctx.render_frames(TRACE, n); ctx.sync()
        dt = time.perf_counter() - t0
        return [ctx.read(i) for i in range(3)], ctx.stats().rays, dt

for (scene, w, h, cam, b, infl, batch, n, nranks) in (("castle", 200, 120, "close", 4, 1, 1, 3, 1), ("castle", 200, 120, "close", 5, 2, 1, 5, 1),
                                                     ("menger", 256, 144, "bench", 4, 2, 4, 12, 1), ("menger", 1920, 1080, "bench", 4, 1, 20, 40, 8),
                                                     ("menger", 1920, 1080, "bench", 4, 3, 8, 40, 1)):
    print(f"{scene} {w}x{h} {cam} bounces {b} inflight {infl} batch {batch} frames {n} ranks {nranks}: ", end="", flush=True)
    ref = frames(scene, w, h, cam, b, infl, batch, n, False, 0, nranks)
    print(f"two kernels {ref[2] * 1e3:.2f} ms; ", end="", flush=True)
    try:
        got = frames(scene, w, h, cam, b, infl, batch, n, True, 0, nranks)
    except VxrtError as e:
        print("FUSED ERROR", e); continue
    same = all(np.array_equal(a.view(np.uint32), g.view(np.uint32)) for a, g in zip(ref[0], got[0]))
    print(f"fused {got[2] * 1e3:.2f} ms; rays {ref[1]} / {got[1]}; last frame identical: {same}", flush=True)

pos, mrgb, size = scenes.load_scene("menger")
for nranks, rank, infl, batch in ((8, 0, 1, 20), (8, 4, 1, 20), (1, 0, 3, 8), (1, 0, 1, 20), (4, 0, 2, 16), (4, 0, 1, 20)):
    for fused in (0, 1):
        with Context(1920, 1080, max_bounces=4, rank=rank, nranks=nranks, band_rows=8, frames_in_flight=infl, frames_per_launch=batch,
                     tuning=[(OPT_FUSED_TAIL, fused)]) as ctx:
            ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*scenes.bench_camera(size))
            for _ in range(20):
                ctx.render_frames(TRACE, 20); ctx.sync()
            ts = []
            for _ in range(150):
                t0 = time.perf_counter(); ctx.render_frames(TRACE, 20); ctx.sync(); ts.append(time.perf_counter() - t0)
            print(f"rank {rank}/{nranks} {batch}x{infl} fused {fused}: block of 20 frames {statistics.median(ts) * 1e3:.4f} ms (min {min(ts) * 1e3:.4f})", flush=True)
