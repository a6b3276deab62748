import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VXRT_ENV_KNOBS", "1")   # host.py: translate the VXRT_* knobs into create-time options
os.environ.update(VXRT_TRACE_VARIANT="2", VXRT_TRACE_SPLIT="0xff", VXRT_TRACE_BLOCKS="64")
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
pos, mrgb, size = scenes.load_scene("castle")
cam = scenes.close_camera(size)
ref = None
bad = 0
for it in range(60):
    with Context(200, 120, max_bounces=5) as ctx:
        ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam); ctx.uniforms.specularity = 0.1
        imgs = []
        for f in range(3):
            ctx.render(TRACE); imgs.append(ctx.read(0).copy())
        rays = ctx.stats().rays
    if ref is None: ref = (imgs, rays)
    for f in range(3):
        d = (imgs[f] != ref[0][f]) & ~(np.isnan(imgs[f]) & np.isnan(ref[0][f]))
        if d.any():
            bad += 1
            idx = np.argwhere(d.any(-1))
            print(f"iter {it} frame {f}: {len(idx)} pixels differ, first {idx[:5].tolist()}, values {imgs[f][tuple(idx[0])]} vs {ref[0][f][tuple(idx[0])]}, rays {rays} vs {ref[1]}")
print("bad", bad)
