"""The 8-bounce 4K configs (monu10 from outside, castle close up; 4 spp, whole frame loop with temporal + denoise r = 2) under scheduling
knobs: tail waves (VXRT_OPT_TRACE_BLOCKS), trace launches in flight, tail hand-over point.  ms per displayed frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import ALL, Camera, Context, TRACE, scenes
from gpu_voxel_raytracer_amd.host import OPT_TAIL_FROM, OPT_TAIL_SPLIT, OPT_TRACE_BLOCKS
W, H, B, SPP = 3840, 2160, 8, 4
cases = [("default", 2, []), ("blocks 1024", 2, [(OPT_TRACE_BLOCKS, 1024)]), ("blocks 4096", 2, [(OPT_TRACE_BLOCKS, 4096)]), ("blocks 8192", 2, [(OPT_TRACE_BLOCKS, 8192)]),
         ("inflight 3", 3, []), ("inflight 1", 1, []), ("tail from 2", 2, [(OPT_TAIL_FROM, 2), (OPT_TAIL_SPLIT, 0x58)]), ("tail from 0", 2, [(OPT_TAIL_FROM, 0), (OPT_TAIL_SPLIT, 0x2e)])]
for scene, view in (("monu10", "bench"), ("castle", "close")):
    pos, mrgb, size = scenes.load_scene(scene)
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    for rep in range(2):
        for label, inflight, tuning in cases:
            with Context(W, H, max_bounces=B, frames_in_flight=inflight, frames_per_launch=SPP, tuning=tuning) as ctx:
                ctx.recreate_octree(pos, mrgb)
                ctx.camera = Camera(*cam)
                ctx.denoise_uniforms.radius = 2
                for flags, what in ((TRACE, "trace"), (ALL, "loop r=2")):
                    for _ in range(4):
                        ctx.render_spp(flags, SPP)
                    ctx.sync()
                    n = 16
                    t0 = time.perf_counter()
                    for _ in range(n):
                        ctx.render_spp(flags, SPP)
                    ctx.sync()
                    dt = (time.perf_counter() - t0) / n
                    print(f"{scene} {view} | {label:12s} | {what:8s}: {dt * 1e3:.4f} ms", flush=True)
