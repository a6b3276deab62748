"""BASELINE config 3: vox/monu10.vox at 3840x2160, 8 bounces, temporal + denoise; per-stage kernel times."""
import sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, ALL, TIMED, scenes
W, H, B = 3840, 2160, 8
name = sys.argv[1] if len(sys.argv) > 1 else "monu10"
pos, mrgb, size = scenes.load_scene(name)
for view in ("bench", "close"):
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    for radius in (0, 2, 8):
        with Context(W, H, max_bounces=B) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            ctx.denoise_uniforms.radius = radius
            for _ in range(4):
                ctx.render(ALL)
            ctx.sync(); ctx.reset_stats()
            n = 12
            for _ in range(n):
                ctx.render(ALL | TIMED)
            st = ctx.stats()
            px = W * H
            t, tm, dn = st.trace_ms / n, st.temporal_ms / n, st.denoise_ms / n
            print(f"{name} {view} r={radius}: trace {t*1e3:8.1f} us ({st.rays / n / (t * 1e-3) / 1e9:.2f} Gray/s, {st.rays / n / px:.2f} rays/px) | "
                  f"temporal {tm*1e3:7.1f} us ({80 * px / (tm * 1e-3) / 1e9:7.0f} GB/s of 80 B/px) | denoise {dn*1e3:8.1f} us" + (f" ({64 * px / (dn * 1e-3) / 1e9:7.0f} GB/s of 64 B/px) | " if dn > 0 else " (fused into temporal) | ")
                  + f"frame {(t + tm + dn):.3f} ms")
