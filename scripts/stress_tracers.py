"""Random-camera stress on the GPU: the default tracer (4: head + compacted tail) with 16 frames per launch and 2 launches in flight
against the all-in-one kernel (tracer 1) one frame at a time — every image of the last frame and the ray totals must be
bit-identical.  At 1080p a run of 20 cameras x 16 frames traces ~1e9 rays, enough to meet the ~1e-7 'exotic' rays (a direction
component exactly 0) a few hundred times.  usage: stress_tracers.py [scene] [cameras]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gpu_voxel_raytracer_amd import TRACE, Camera, Context, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "menger"
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H, B, F = 1920, 1080, 4, 16
pos, mrgb, size = scenes.load_scene(name)
ext = scenes.world_extent(size)
rng = np.random.default_rng(7)
bad = 0
t0 = time.time()
with Context(W, H, max_bounces=B, tracer=1) as ref, Context(W, H, max_bounces=B, tracer=0, frames_per_launch=F, frames_in_flight=2) as new:
    for c in (ref, new):
        c.recreate_octree(pos, mrgb)
    total = 0
    for i in range(cams):
        centre = ext * np.float32(0.5)
        p = (centre + ext.max() * rng.uniform(-1.1, 1.1, 3)).astype(np.float32)
        if i % 4 == 3:
            p = (centre + ext * rng.uniform(-0.45, 0.45, 3)).astype(np.float32)      # inside the model's box
        d = (centre + ext * rng.uniform(-0.3, 0.3, 3) - p).astype(np.float32)
        if i % 5 == 4:
            d = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.001]][(i // 5) % 3], np.float32)   # axis-aligned views: zero components in primary rays
        spec = 0.3 if i % 3 == 0 else 0.0
        for c in (ref, new):
            c.camera = Camera(p, d, scenes.FOV_70)
            c.uniforms.specularity = spec
            c.reset_stats()
            c.render_frames(TRACE, F)
        for img in range(3):
            a, b = ref.read(img), new.read(img)
            diff = (a != b) & ~(np.isnan(a) & np.isnan(b))
            if diff.any():
                bad += 1
                idx = np.argwhere(diff.any(-1))
                print(f"camera {i} image {img}: {len(idx)} pixels differ, first {idx[:3].tolist()}: {a[tuple(idx[0])]} vs {b[tuple(idx[0])]}")
        ra, rb = ref.stats().rays, new.stats().rays
        total += ra
        if ra != rb:
            bad += 1
            print(f"camera {i}: rays {ra} vs {rb}")
print(f"{name}: {cams} cameras x {F} frames, {total / 1e6:.0f} M rays compared, mismatches: {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
