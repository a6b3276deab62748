"""A few full-pipeline frames (trace + temporal + denoise) of the bench scene at 1080p — a target for rocprofv3 passes over the post kernels.
RADIUS env: denoise radius (0 = fused into the temporal kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import ALL, Camera, Context, scenes
pos, mrgb, size = scenes.load_scene("menger")
with Context(1920, 1080, max_bounces=4) as ctx:
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*scenes.bench_camera(size))
    ctx.denoise_uniforms.radius = int(os.environ.get("RADIUS", "0"))
    for _ in range(6):
        ctx.render(ALL)
    ctx.sync()
