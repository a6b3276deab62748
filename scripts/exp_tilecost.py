import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, TIMED, scenes, host
W, H = 1920, 1080
pos, mrgb, size = scenes.load_scene("menger")
for view in ("bench", "close"):
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    with Context(W, H, max_bounces=4) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        for _ in range(5):
            ctx.render(TRACE)
        ctx.sync(); ctx.reset_stats()
        for _ in range(10):
            ctx.render(TRACE | TIMED)
        st = ctx.stats()
        n = ((W + 15) // 16) * ((H + 15) // 16)
        cost = np.zeros(n, np.uint32)
        host._check(host.lib().vxrt_debug_tile_costs(ctx._h, cost.ctypes.data_as(C.c_void_p), C.c_size_t(n)), "tilecost")
        us = cost / 100.0   # s_memtime ticks at 100 MHz
        k = st.trace_ms / st.timed_frames * 1e3
        print(f"{view}: kernel {k:.1f} us; tile duration us: max {us.max():.1f} p99 {np.percentile(us,99):.1f} p90 {np.percentile(us,90):.1f} p50 {np.percentile(us,50):.1f} mean {us.mean():.2f}; "
              f"sum {us.sum():.0f} us -> /1024 block slots = {us.sum()/1024:.1f} us; tiles>50us: {(us>50).sum()}")
