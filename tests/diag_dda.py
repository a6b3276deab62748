#!/usr/bin/env python3
"""PROTOTYPE DIAGNOSTIC (round 5; not collected by pytest; needs a GPU and the -DVXRT_VARIANTS=1 library): csrc/trace_dda.hip against the exact
walk on the same rays.  The library builds the bit grid on the device from the scene in place (round 6; round 5 built it here from
the voxel list with numpy); this makes realistic ray sets — the bench camera's
primary rays, sun rays and hemisphere rays from the primary hits — and reports, per set: rays, hits, FLAGGED rays (the certificate says
"let the walk decide"), unflagged rays whose result differs from the walk's in any bit (must be 0), mean / max DDA steps, and the
kernels' times (HIP events, lock-step waves of 64 rays in input order).
usage: python tests/diag_dda.py [scene | config5] [margin_scale]      config5 = BASELINE configs[4]'s 2048^3 scene at 3840x2160 (HBM-resident)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpu_voxel_raytracer_amd import Camera, Context, host, scenes  # noqa: E402


def run(ctx, o, d, certify=1, margin=2.0, max_steps=600, lds_top=0, skip_walk=False, skip_dda=False):
    """vxrt_debug_dda_rays (vxrt_debug.h is silent about it: a prototype of the -DVXRT_VARIANTS=1 build): the rays through the walk of the
    context's scene format and through the DDA over the grid the library builds on the device from the scene in place.
    -> (walk results [n, 8], DDA results [n, 8], walk ms, DDA ms, grid build ms, grid bytes)"""
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    n = len(o)
    ow, od = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32)
    ms = (C.c_double * 4)()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    flags = (1 if certify else 0) | (2 if lds_top else 0) | (4 if skip_walk else 0) | (8 if skip_dda else 0)
    rc = ctx._L.vxrt_debug_dda_rays(ctx._h, p(o), p(d), C.c_size_t(n), C.c_uint32(flags), C.c_float(margin), C.c_uint32(max_steps), p(ow), p(od), ms)
    assert rc == 0, (rc, ctx._L.vxrt_last_error())
    return ow, od, ms[0], ms[1], ms[2], ms[3]


def report(name, ow, od, t_walk, t_dda):
    flagged = od[:, 6] != 0
    same = (ow[:, :6].view(np.uint32) == od[:, :6].view(np.uint32)).all(1) | ((ow[:, 0] == 0) & (od[:, 0] == 0))
    bad = ~same & ~flagged
    print(f"{name:34s}: {len(ow):8d} rays, {int((ow[:, 0] != 0).sum()):8d} hits, flagged {int(flagged.sum()):6d} ({flagged.mean() * 100:6.3f} %), "
          f"unflagged and different {int(bad.sum()):5d}, flagged and different {int((~same & flagged).sum()):5d}; steps mean {od[:, 7].mean():5.1f} max {int(od[:, 7].max())}; "
          f"walk {t_walk:7.3f} ms, DDA {t_dda:7.3f} ms ({t_walk / t_dda:4.2f} x)", flush=True)
    if bad.any():
        k = np.nonzero(bad)[0][:3]
        for i in k:
            print("    walk", ow[i], "dda", od[i])
    return int(bad.sum())


def ray_sets(ctx, cam, w, h, rng, trim=0):
    """The camera's primary rays in 8 x 8 tiles (as the tracer's waves see them), and from their hits (the walk's results, asked for here)
    sun rays — the reference's sun direction, jittered as voxels.comp:339-356 jitters it — and hemisphere rays about the normal."""
    r, u, f = cam.axis_scaled(w, h)
    ys, xs = np.mgrid[0:h, 0:w]
    xs = xs.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1); ys = ys.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
    d = (xs[:, None].astype(np.float32) * r - ys[:, None].astype(np.float32) * u).astype(np.float32) + f
    d = (d / np.sqrt((d * d).sum(1, keepdims=True, dtype=np.float32))).astype(np.float32)
    o = np.broadcast_to(cam.position, d.shape).astype(np.float32)
    if trim:                   # a view's primary set a few waves shorter than the other view's: a profiler's rows can be told apart by grid size
        o, d = o[:-trim], d[:-trim]
    yield "primary rays", o, d
    ow = run(ctx, o, d, skip_dda=True)[0]
    hit = ow[:, 0] != 0
    hp = (o[hit] + d[hit] * ow[hit, 1:2]).astype(np.float32)
    nrm = ow[hit, 3:6]
    so = (hp + np.float32(1e-5) * nrm).astype(np.float32)
    yaw, pitch = np.float32(1.32), np.float32(1.0)
    sun = np.array([np.cos(yaw) * np.cos(pitch), -np.sin(pitch), np.sin(yaw) * np.cos(pitch)], np.float32)
    sd = (-sun / np.linalg.norm(sun)).astype(np.float32) + rng.normal(0, 0.03, so.shape).astype(np.float32)
    sd = (sd / np.linalg.norm(sd, axis=1, keepdims=True)).astype(np.float32)
    yield "sun rays from the hits", so, sd
    v = rng.normal(size=so.shape).astype(np.float32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    flip = (v * nrm).sum(1) < 0
    v[flip] = -v[flip]
    yield "bounce rays from the hits", so[:-64], v[:-64]      # (64 rays fewer than the sun rays: a launch of its own size, so that a profiler's rows can be told apart)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "menger"
    margin = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    host.use_library(host.variants_library())
    rng = np.random.default_rng(7)
    bad = 0
    config5 = scene == "config5"
    w, h = (3840, 2160) if config5 else (1920, 1080)
    steps = 2048 if config5 else 600
    sets = {}
    with Context(w, h, max_bounces=4) as ctx:
        if config5:
            ctx.set_menger(*scenes.CONFIG5)
            views = list(scenes.config5_cameras().items())
        else:
            pos, mrgb, size = scenes.load_scene(scene)
            ctx.recreate_octree(pos, mrgb)
            views = [(v, getattr(scenes, v + "_camera")(size)) for v in ("bench", "close")]
        for vi, (view, cam) in enumerate(views):
            for name, o, d in ray_sets(ctx, Camera(*cam), w, h, rng, trim=256 * vi):
                sets[(view, name)] = (o, d)
                ow, od, tw, td, tb, gb = run(ctx, o, d, 1, margin, steps)
                if tb:
                    print(f"# the grid of the scene in place: {gb / 2**30:.2f} GiB of arrays, built on the device in {tb:.1f} ms", flush=True)
                bad += report(f"{scene} {view}: {name}", ow, od, tw, td)
                _, _, tw0, td0, _, _ = run(ctx, o, d, 0, margin, steps)
                _, _, _, tl, _, _ = run(ctx, o, d, 0, margin, steps, lds_top=1, skip_walk=True)
                print(f"{'':34s}  without the certificate: DDA {td0:7.3f} ms ({tw0 / td0:4.2f} x); with the super-brick bits in LDS: {tl:7.3f} ms", flush=True)
    if config5:
        # the wide records (two tree levels per 16-byte record) on the SAME rays: a context of its own (built on the host: ~10 s + 5.6 GB up)
        with Context(w, h, max_bounces=4, tuning=[(host.OPT_SCENE_FORMAT, 1)]) as ctx:
            ctx.set_menger(*scenes.CONFIG5)
            assert ctx.stats().scene_format == 1
            for (view, name), (o, d) in sets.items():
                _, _, tw, _, _, _ = run(ctx, o, d, skip_dda=True)
                print(f"{scene} {view}: {name:28s}: wide records, the walk: {tw:7.3f} ms", flush=True)
    print("unflagged differences in all:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
