"""ORACLE-BASED DIAGNOSTIC (not collected by pytest): the kernels' cast_bounded_ray (vxrt_debug_cast_rays) against the oracle's for
rays with zero direction components — origins near / exactly on voxel planes, +0 and -0 — and for the logged ray of the stress failure."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, scenes
from oracle import oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "menger"
pos, mrgb, size = scenes.load_scene(name)
octree = O.create_octree(pos, mrgb)
ext = scenes.world_extent(size)
rng = np.random.default_rng(3)
n = 400000
o = (rng.uniform(-0.1, 1.1, (n, 3)) * ext).astype(np.float32)
snap = rng.random((n, 3)) < 0.5
o = np.where(snap, np.round(o * 2) / 2, o).astype(np.float32)                       # many coordinates exactly on voxel planes
o = (o + np.where(rng.random((n, 3)) < 0.3, np.float32(1e-5) * rng.choice([-1, 1], (n, 3)), 0)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
zero = rng.random((n, 3)) < 0.45
zero[zero.all(1), 0] = False
d = np.where(zero, np.where(rng.random((n, 3)) < 0.5, np.float32(0.0), np.float32(-0.0)), d).astype(np.float32)
axis = rng.random(n) < 0.3
d[axis] = np.sign(d[axis]) * (np.abs(d[axis]) > 0)                                    # exact axis directions with signed zeros kept
d = np.where((d == 0) & axis[:, None], np.where(rng.random((n, 3)) < 0.5, np.float32(0.0), np.float32(-0.0)), d).astype(np.float32)
# the ray of the stress failure (menger, camera 14, frame 45, pixel (1002, 265), 5th cast)
if name == "menger":
    p = np.array([4.765862464904785, 30.47458839416504, 5.789773941040039], np.float32); dd = np.array([0, 0, 1], np.float32)
    u = O.Uniforms.default(); u.set_camera(p, O.camera_axis_scaled(p, dd, scenes.FOV_70, 1920, 1080)); u.frame_number = 45
    log = np.zeros((32, 12), np.float32)
    k = O.lib().orc_trace_pixel_log(O._p(octree), O._p(O.noise_table()), C.byref(u), C.c_int(4), C.c_int(1002), C.c_int(265), O._p(log))
    o[:k] = log[:k, 0:3]; d[:k] = log[:k, 3:6]
with Context(64, 64) as ctx:
    ctx.recreate_octree(pos, mrgb)
    gh, gt, gn, gnorm = ctx.cast_rays(o, d)
oh, ot, on, onorm, _ = O.cast_rays(octree, o, d)
def same(a, b): return (a == b) | (np.isnan(a) & np.isnan(b))
bad = (gh != oh) | (gh & ~(same(gt, ot) & (gn == on) & same(gnorm, onorm).all(1))) | (~gh & ~same(gt, ot))
print(f"{name}: {n} rays, {int(zero.any(1).sum())} with a zero component, hits {int(oh.sum())}; disagreements: {int(bad.sum())}")
for i in np.nonzero(bad)[0][:12]:
    print(f"  ray {i}: o {o[i].tolist()} d {d[i].tolist()} (bits {d[i].view(np.uint32).tolist()})\\n     gpu hit {gh[i]} t {gt[i]} node {hex(int(np.uint32(gn[i])))} n {gnorm[i]} | oracle hit {oh[i]} t {ot[i]} node {hex(int(np.uint32(on[i])))} n {onorm[i]}")
