"""ORACLE-BASED DIAGNOSTIC (not collected by pytest): traces with noise tables full of special values (exact 0, 0.5, 0.25 ...) so that
the degenerate shading paths (plane_radius = 0, phi = 0, rand_dir = 0 ...) occur at every pixel; compares GPU and oracle per pixel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from gpu_voxel_raytracer_amd import TRACE, Camera, Context, scenes
from oracle import oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "menger"
w, h, bounces = 256, 144, 4
pos, mrgb, size = scenes.load_scene(name)
octree = O.create_octree(pos, mrgb)
cam = scenes.close_camera(size)
u = O.Uniforms.default()
u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
rng = np.random.default_rng(1)
base = O.noise_table()
tables = {
    "all zero": np.zeros_like(base),
    "all 0.5": np.full_like(base, 0.5),
    "30% zeros": np.where(rng.random(base.size) < 0.3, np.float32(0), base).astype(np.float32),
    "specials": rng.choice(np.array([0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0 ** -24], np.float32), base.size).astype(np.float32),
}
for label, table in tables.items():
    with Context(w, h, max_bounces=bounces, noise=table) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        for frame in (1, 2):
            ctx.set_frame_number(frame - 1)
            ctx.reset_stats()
            ctx.render(TRACE)
            got = [ctx.read(i) for i in range(3)]
            rays = ctx.stats().rays
            u.frame_number = frame
            ref = O.trace(octree, table, u, w, h, bounces, crop=(0, 0, w, h))
            a, b = got[0], ref[0]
            diff = ((a != b) & ~(np.isnan(a) & np.isnan(b))).any(-1)
            print(f"{label} frame {frame}: rays gpu {rays} oracle {ref[3]}, colour pixels differing {int(diff.sum())} of {w * h}", flush=True)
            for y, x in np.argwhere(diff)[:3]:
                print(f"    ({y},{x}) gpu {a[y, x]} oracle {b[y, x]}")
                ctx.set_frame_number(frame - 1)
                gl, ol = ctx.path_log(int(x), int(y)), O.path_log(octree, table, u, bounces, int(x), int(y))
                print(f"      casts gpu {len(gl)} oracle {len(ol)}")
                for k in range(max(len(gl), len(ol))):
                    gs = gl[k].view(np.uint32) if k < len(gl) else None
                    os_ = ol[k].view(np.uint32) if k < len(ol) else None
                    same = gs is not None and os_ is not None and (gs == os_).all()
                    print(f"      cast {k} {'same' if same else 'DIFF'}")
                    if not same:
                        print(f"        gpu    {None if gs is None else gl[k].tolist()}")
                        print(f"        oracle {None if os_ is None else ol[k].tolist()}")
                        print(f"        bits gpu    {None if gs is None else [hex(v) for v in gs]}")
                        print(f"        bits oracle {None if os_ is None else [hex(v) for v in os_]}")
                        break
