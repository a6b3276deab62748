"""GPU: north_star's design — a DDA through a bricked dense grid, 8^3 bricks — built so that it is BIT-EXACT (csrc/trace_dda.hip, a prototype in
the -DVXRT_VARIANTS=1 library), against the octree walk the product ships.

Every time the walk of voxels.comp:134-247 computes is the crossing time fl(fl(p - o) * inv) of a dyadic grid plane, so a DDA that steps
by the same computed times makes the same comparisons and visits the same cells; the one decision the walk makes from a POSITION
(current_octant on a descend) is guarded by a certificate — the point keeps a margin of a few ulps from every unit plane — and a ray that
fails it is flagged for the exact walk.  The test: on the bench scene and two others, primary, sun and hemisphere rays (what a frame
casts) — every UNFLAGGED ray's hit flag, time, leaf word and normal equal cast_ray's bit for bit, and few rays are flagged.

The measurement that goes with it (tests/diag_dda.py, profiles/r05/dda_prototype.txt): lock-step waves of DDA rays take 1.4-6 x as long as
waves of octree walks — the octree IS the hierarchical DDA, it crosses the empty three quarters of the root cube in two trips where 8^3
bricks take thirty steps — which is why the product keeps the octree in a denser encoding.  Parity status: the
comparison is product against product; the reference pins neither (parity unpinned)."""
import numpy as np
import pytest

from conftest import require_variants

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene,view", [("menger", "bench"), ("castle", "close"), ("monu10", "close")])
def test_unflagged_dda_rays_equal_the_octree_walk_bit_for_bit(H, scenes, scene, view):
    import diag_dda as D
    from gpu_voxel_raytracer_amd import Camera, Context
    require_variants(H, tracer=5)                      # the prototype lives in the variants library
    pos, mrgb, size = scenes.load_scene(scene)
    w, h = 640, 360
    rng = np.random.default_rng(3)
    with Context(w, h, max_bounces=2) as ctx:
        ctx.recreate_octree(pos, mrgb)
        cam = Camera(*getattr(scenes, view + "_camera")(size))
        r, u, f = cam.axis_scaled(w, h)
        ys, xs = np.mgrid[0:h, 0:w]
        d = (xs.reshape(-1, 1).astype(np.float32) * r - ys.reshape(-1, 1).astype(np.float32) * u).astype(np.float32) + f
        d = (d / np.sqrt((d * d).sum(1, keepdims=True, dtype=np.float32))).astype(np.float32)
        o = np.broadcast_to(cam.position, d.shape).astype(np.float32)
        ow, od = D.run(ctx, o, d, 1, 2.0)[:2]
        sets = [("primary", ow, od)]
        hit = ow[:, 0] != 0
        assert hit.sum() > 5000
        so = ((o[hit] + d[hit] * ow[hit, 1:2]).astype(np.float32) + np.float32(1e-5) * ow[hit, 3:6]).astype(np.float32)
        v = rng.normal(size=so.shape).astype(np.float32)
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        flip = (v * ow[hit, 3:6]).sum(1) < 0
        v[flip] = -v[flip]
        sets.append(("bounce",) + D.run(ctx, so, v, 1, 2.0)[:2])
        sets.append(("bounce, super-brick bits in LDS",) + D.run(ctx, so, v, 1, 2.0, lds_top=1)[:2])
        # axis-parallel and zero-component directions: never decided by the DDA
        z = v.copy(); z[:, 1] = 0.0
        sets.append(("zero component",) + D.run(ctx, so[:4096], z[:4096], 1, 2.0)[:2])
    for name, a, b in sets:
        flagged = b[:, 6] != 0
        same = (a[:, :6].view(np.uint32) == b[:, :6].view(np.uint32)).all(1) | ((a[:, 0] == 0) & (b[:, 0] == 0))
        assert (same | flagged).all(), (scene, name, int((~same & ~flagged).sum()))
        if name == "zero component":
            assert flagged.all()
        else:
            assert flagged.mean() < 0.02, (scene, name, flagged.mean())
            assert (b[~flagged, 0] != 0).sum() > 1000            # and the DDA does decide hits by itself
