"""Trace stage of the 8-bounce configs (BASELINE configs[2] / [3]: monu10 and castle at 3840x2160, 4 spp) under different re-compaction
masks of the tail (VXRT_OPT_TAIL_SPLIT: bit k = the tail compacts again, in a new launch, at path segment k).  ms per displayed frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Camera, Context, TRACE, scenes
from gpu_voxel_raytracer_amd.host import OPT_TAIL_SPLIT
W, H, B, SPP = 3840, 2160, 8, 4
masks = [int(m, 0) for m in (sys.argv[1:] or ["0x8", "0x0", "0x28", "0x18", "0x48", "0x14", "0x54", "0xa8", "0xfc"])]
for scene, view in (("monu10", "bench"), ("castle", "close")):
    pos, mrgb, size = scenes.load_scene(scene)
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    for rep in range(2):
        for m in masks:
            with Context(W, H, max_bounces=B, frames_in_flight=2, frames_per_launch=SPP, tuning=[(OPT_TAIL_SPLIT, m)]) as ctx:
                ctx.recreate_octree(pos, mrgb)
                ctx.camera = Camera(*cam)
                for _ in range(4):
                    ctx.render_spp(TRACE, SPP)
                ctx.sync(); ctx.reset_stats()
                n = 16
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.render_spp(TRACE, SPP)
                ctx.sync()
                dt = (time.perf_counter() - t0) / n
                print(f"{scene} {view} split {m:#04x}: {dt * 1e3:.4f} ms per displayed frame, {ctx.stats().rays / n / dt / 1e9:.2f} Gray/s", flush=True)
