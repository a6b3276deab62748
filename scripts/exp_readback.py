"""The PCIe-inclusive rate of the boundary: full pipeline frames (trace + temporal + denoise r = 0) at 1080p, each followed by
vxrt_read of the denoised image into host memory (33 MB per frame) — what a host that displays every frame from host memory pays.
bench.py's `value` never includes this: its timed region touches HBM-resident buffers only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import ALL, DENOISED, Camera, Context, scenes
pos, mrgb, size = scenes.load_scene("menger")
with Context(1920, 1080, max_bounces=4) as ctx:
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*scenes.bench_camera(size))
    for _ in range(5):
        ctx.render(ALL); ctx.read(DENOISED)
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render(ALL)
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(n):
        ctx.render(ALL); ctx.read(DENOISED)
    t2 = time.perf_counter()
    rays = ctx.stats().rays / (2 * n + 5)
    print(f"frames one at a time, resident: {(t1 - t0) / n * 1e3:.3f} ms/frame ({rays / ((t1 - t0) / n) / 1e9:.2f} Gray/s); "
          f"with a 33 MB read-back per frame: {(t2 - t1) / n * 1e3:.3f} ms/frame ({rays / ((t2 - t1) / n) / 1e9:.2f} Gray/s, "
          f"{33.1776 / ((t2 - t1) / n) / 1e3:.1f} GB/s over PCIe incl. the frame)")
