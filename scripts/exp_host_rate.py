import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
pos, mrgb, size = scenes.load_scene("castle")
for infl in (1, 3):
    with Context(64, 64, max_bounces=1, frames_in_flight=infl) as ctx:
        ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*scenes.bench_camera(size))
        for _ in range(100): ctx.render(TRACE)
        ctx.sync()
        t0 = time.perf_counter(); n = 5000
        for _ in range(n): ctx.render(TRACE)
        t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        print(f"inflight={infl}: host issue {((t1 - t0) / n) * 1e6:.1f} us/frame, total {((t2 - t0) / n) * 1e6:.1f} us/frame")
        t0 = time.perf_counter()
        for _ in range(n): ctx.render_stage(TRACE)
        t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        print(f"   render_stage only: host issue {((t1 - t0) / n) * 1e6:.1f} us/frame, total {((t2 - t0) / n) * 1e6:.1f} us/frame")
        t0 = time.perf_counter()
        ctx.render_frames(TRACE, n)
        t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        print(f"   render_frames (C loop): host issue {((t1 - t0) / n) * 1e6:.1f} us/frame, total {((t2 - t0) / n) * 1e6:.1f} us/frame")
