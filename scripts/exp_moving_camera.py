"""Trace stage along a camera path (vxrt_render_path: one camera per frame, 16 frames per launch, 2 launches in flight): ms per frame.
usage: python scripts/exp_moving_camera.py [degrees per frame]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
from gpu_voxel_raytracer_amd.frame_loop import orbit_camera
step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
pos, mrgb, size = scenes.load_scene("menger")
with Context(1920, 1080, max_bounces=4, frames_in_flight=2, frames_per_launch=16) as ctx:
    ctx.recreate_octree(pos, mrgb)
    n = 480
    path = [orbit_camera(size, 0.62 + f * step / 360.0) for f in range(4 * n)]
    P = np.array([p[0] for p in path], np.float32); D = np.array([p[1] for p in path], np.float32)
    ctx.render_path(TRACE, P[:n], D[:n], path[0][2]); ctx.sync()
    ts = []
    for b in range(1, 4):
        ctx.reset_stats()
        t0 = time.perf_counter(); ctx.render_path(TRACE, P[b * n:(b + 1) * n], D[b * n:(b + 1) * n], path[0][2]); ctx.sync(); ts.append(time.perf_counter() - t0)
        rays = ctx.stats().rays
    t = sorted(ts)[1]
    print(f"{step} deg/frame: {t / n * 1e3:.4f} ms/frame, {rays / t / 1e9:.2f} Gray/s, frame-lane launches {ctx.stats().frame_lane_launches}")
