import sys, time, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, TIMED, scenes
W, H = 1920, 1080
pos, mrgb, size = scenes.load_scene("menger")
p, d, fov = scenes.bench_camera(size)
for name, cam in (("bench", (p, d, fov)), ("away(all sky, miss root)", (p, -d, fov)), ("close", scenes.close_camera(size)),
                  ("far(all sky but through root)", (p * 8, d, fov * 0.05))):
    for bounces in (1, 4):
        with Context(W, H, max_bounces=bounces) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            for _ in range(5):
                ctx.render(TRACE)
            ctx.sync(); ctx.reset_stats()
            for _ in range(30):
                ctx.render(TRACE | TIMED)
            st = ctx.stats()
            print(f"{name:32s} B={bounces} kernel {st.trace_ms / st.timed_frames * 1e3:8.1f} us  rays/px {st.rays / st.frames / (W * H):.3f}")
