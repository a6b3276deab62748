"""ORACLE-BASED DIAGNOSTIC (not collected by pytest): walks the camera sequence of tests/test_gpu_stress.py for a scene and compares
EVERY frame, one at a time, image by image and ray count by ray count, with the oracle; prints cameras and pixels that differ.
usage: debug_stress.py <scene> <bounces> <cameras> [tracer] [seed offset]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from gpu_voxel_raytracer_amd import TRACE, Camera, Context, scenes
from oracle import oracle as O

import zlib
name, bounces, cams = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tracer = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w, h, frames = 1920, 1080, 3
pos, mrgb, size = scenes.load_scene(name)
octree = O.create_octree(pos, mrgb)
noise = O.noise_table()
ext = scenes.world_extent(size)
centre = ext * np.float32(0.5)
rng = np.random.default_rng(zlib.crc32(name.encode()) + (int(sys.argv[5]) if len(sys.argv) > 5 else 0))
bad = 0
with Context(w, h, max_bounces=bounces, noise=noise, tracer=tracer) as ctx:
    ctx.recreate_octree(pos, mrgb)
    for i in range(cams):
        p = (centre + ext.max() * rng.uniform(-1.1, 1.1, 3)).astype(np.float32)
        if i % 4 == 1:
            p = (centre + ext * rng.uniform(-0.45, 0.45, 3)).astype(np.float32)
        d = (centre + ext * rng.uniform(-0.3, 0.3, 3) - p).astype(np.float32)
        if i % 4 == 2:
            d = np.array([[1, 0, 0], [0, 0, 1]][(i // 4) % 2], np.float32)
        spec = 0.3 if i % 3 == 0 else 0.0
        u = O.Uniforms.default()
        u.specularity = spec
        u.set_camera(p, O.camera_axis_scaled(p, d, scenes.FOV_70, w, h))
        ctx.camera = Camera(p, d, scenes.FOV_70)
        ctx.uniforms.specularity = spec
        only = os.environ.get("DEBUG_ONLY")
        if only and str(i) not in only.split(","):
            continue
        for f in range(frames):
            fn = i * frames + f + 1
            ctx.set_frame_number(fn - 1)
            ctx.reset_stats()
            ctx.render(TRACE)
            got = [ctx.read(img) for img in range(3)]
            rays = ctx.stats().rays
            u.frame_number = fn
            ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
            lines = []
            if rays != ref[3]:
                lines.append(f"   rays gpu {rays} oracle {ref[3]}")
            for img, label in zip(range(3), ("colour", "nd", "albedo")):
                a, b = got[img], ref[img]
                diff = ((a != b) & ~(np.isnan(a) & np.isnan(b))).any(-1)
                if diff.any():
                    idx = np.argwhere(diff)
                    lines.append(f"   {label}: {len(idx)} pixels differ; first {idx[:4].tolist()}")
                    for y, x in idx[:3]:
                        lines.append(f"      ({y},{x}) gpu {a[y, x]} oracle {b[y, x]} | nd gpu {got[1][y, x]} oracle {ref[1][y, x]}")
            if lines:
                bad += 1
                print(f"camera {i} frame {fn}: position {p.tolist()} direction {d.tolist()} spec {spec}")
                print("\n".join(lines), flush=True)
print(f"{name}: {cams} cameras x {frames} frames checked, {bad} frames differ")
