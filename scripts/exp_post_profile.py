"""Target for rocprofv3 --stats over the post kernels: monu10 at 3840x2160, 8 bounces, 8 frames each with denoise radius 0 (fused
into temporal_kernel), 2 and 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import ALL, Camera, Context, scenes
pos, mrgb, size = scenes.load_scene("monu10")
for radius in (0, 2, 8):
    with Context(3840, 2160, max_bounces=8) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*scenes.bench_camera(size))
        ctx.denoise_uniforms.radius = radius
        for _ in range(8):
            ctx.render(ALL)
        ctx.sync()
