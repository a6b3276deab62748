"""How much do independent frames gain from being in flight together?  N contexts (own streams) on one GPU."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
W, H = 1920, 1080
pos, mrgb, size = scenes.load_scene("menger")
for view in ("bench", "close"):
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    for n in (1, 2, 3, 4, 6):
        ctxs = [Context(W, H, max_bounces=4) for _ in range(n)]
        for c in ctxs:
            c.recreate_octree(pos, mrgb); c.camera = Camera(*cam)
            for _ in range(5): c.render(TRACE)
        for c in ctxs: c.sync(); c.reset_stats()
        frames = 240
        t0 = time.perf_counter()
        for f in range(frames):
            ctxs[f % n].render(TRACE)
        for c in ctxs: c.sync()
        dt = time.perf_counter() - t0
        rays = sum(c.stats().rays for c in ctxs)
        print(f"{view}: {n} in flight: {dt / frames * 1e3:.4f} ms/frame, {rays / dt / 1e9:.2f} Gray/s")
        for c in ctxs: c.close()
