#!/usr/bin/env python3
"""PROTOTYPE DIAGNOSTIC (round 5; not collected by pytest; needs a GPU and the -DVXRT_VARIANTS=1 library): csrc/trace_dda.hip against the exact
walk on the same rays.  Builds the dense bit grid of a scene from its voxel list (numpy), makes realistic ray sets — the bench camera's
primary rays, sun rays and hemisphere rays from the primary hits — and reports, per set: rays, hits, FLAGGED rays (the certificate says
"let the walk decide"), unflagged rays whose result differs from the walk's in any bit (must be 0), mean / max DDA steps, and the
kernels' times (HIP events, lock-step waves of 64 rays in input order).
usage: python tests/diag_dda.py [scene] [margin_scale]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpu_voxel_raytracer_amd import Camera, Context, host, scenes  # noqa: E402


def grids(pos, mrgb, words, depth):
    """-> (bricks uint64[nb^3 * 8], brick_bits uint32, leaf int32[n^3], levels) for the tree of `pos` (root cube from the octree's header)."""
    hdr = words[:5].view(np.float32)
    center, size = hdr[:3].astype(np.float64), float(hdr[3])
    levels = depth + 1                      # node levels: the leaf octants are half the finest node
    n = 1 << levels
    cell = size / n
    rmin = center - size / 2
    j = np.rint((pos.astype(np.float64) * 0.5 - rmin) / cell).astype(np.int64)          # voxel (integer position p) = world cube [p/2, p/2 + 1/2)
    assert (j >= 0).all() and (j < n).all() and cell == 0.5
    leaf = np.zeros(n ** 3, np.int32)
    m = mrgb.astype(np.uint32)
    word = (np.uint32(0x80000000) | (m[:, 0] << 24) | (m[:, 1] << 16) | (m[:, 2] << 8) | m[:, 3]).astype(np.uint32)
    lin = (j[:, 0] << (2 * levels)) | (j[:, 1] << levels) | j[:, 2]
    leaf[lin] = word.view(np.int32)
    x, y, z = j[:, 0], j[:, 1], j[:, 2]
    bit = ((((x >> 2) & 1) << 2 | ((y >> 2) & 1) << 1 | ((z >> 2) & 1)) << 6) | ((((x >> 1) & 1) << 2 | ((y >> 1) & 1) << 1 | ((z >> 1) & 1)) << 3) | \
          ((x & 1) << 2 | (y & 1) << 1 | (z & 1))
    nb = n >> 3
    blin = ((x >> 3) * nb + (y >> 3)) * nb + (z >> 3)
    bricks = np.zeros(nb ** 3 * 8, np.uint64)
    np.bitwise_or.at(bricks, blin * 8 + (bit >> 6), np.uint64(1) << (bit & 63).astype(np.uint64))
    bb = np.zeros((nb ** 3 + 31) // 32, np.uint32)
    np.bitwise_or.at(bb, blin >> 5, np.uint32(1) << (blin & 31).astype(np.uint32))
    return bricks, bb, leaf, levels


def run(ctx, g, o, d, certify=1, margin=2.0):
    bricks, bb, leaf, levels = g
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    n = len(o)
    ow, od = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32)
    ms = (C.c_double * 2)()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = ctx._L.vxrt_debug_dda_rays(ctx._h, p(bricks), p(bb), p(leaf), C.c_int32(levels), p(o), p(d), C.c_size_t(n), C.c_int32(certify), C.c_float(margin),
                                    p(ow), p(od), ms)
    assert rc == 0, (rc, ctx._L.vxrt_last_error())
    return ow, od, ms[0], ms[1]


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


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "menger"
    margin = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    host.use_library(host.variants_library())
    pos, mrgb, size = scenes.load_scene(scene)
    words, depth = host.build_octree(pos, mrgb)
    g = grids(pos, mrgb, words, depth)
    w, h = 1920, 1080
    rng = np.random.default_rng(7)
    bad = 0
    with Context(w, h, max_bounces=4) as ctx:
        ctx.recreate_octree(pos, mrgb)
        for view in ("bench", "close"):
            cam = Camera(*getattr(scenes, view + "_camera")(size))
            r, u, f = cam.axis_scaled(w, h)
            ys, xs = np.mgrid[0:h, 0:w]
            # 8 x 8 tiles as the tracer's waves see them
            xs = xs.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1); ys = ys.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
            d = (xs[:, None].astype(np.float32) * r - ys[:, None].astype(np.float32) * u).astype(np.float32) + f
            d = (d / np.sqrt((d * d).sum(1, keepdims=True, dtype=np.float32))).astype(np.float32)
            o = np.broadcast_to(cam.position, d.shape).astype(np.float32)
            ow, od, tw, td = run(ctx, g, o, d, 1, margin)
            bad += report(f"{scene} {view}: primary rays", ow, od, tw, td)
            _, od0, tw0, td0 = run(ctx, g, o, d, 0, margin)
            print(f"{'':34s}  without the certificate: DDA {td0:7.3f} ms ({tw0 / td0:4.2f} x)")
            hit = ow[:, 0] != 0
            hp = (o[hit] + d[hit] * ow[hit, 1:2]).astype(np.float32)
            nrm = ow[hit, 3:6]
            so = (hp + np.float32(1e-5) * nrm).astype(np.float32)
            # sun rays: the reference's sun direction, jittered as voxels.comp:339-356 jitters it
            yaw, pitch = np.float32(1.32), np.float32(1.0)
            sun = np.array([np.cos(yaw) * np.cos(pitch), -np.sin(pitch), np.sin(yaw) * np.cos(pitch)], np.float32)
            sd = (-sun / np.linalg.norm(sun)).astype(np.float32) + rng.normal(0, 0.03, so.shape).astype(np.float32)
            sd = (sd / np.linalg.norm(sd, axis=1, keepdims=True)).astype(np.float32)
            ow2, od2, tw2, td2 = run(ctx, g, so, sd, 1, margin)
            bad += report(f"{scene} {view}: sun rays from the hits", ow2, od2, tw2, td2)
            # hemisphere rays about the normal
            v = rng.normal(size=so.shape).astype(np.float32)
            v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
            flip = (v * nrm).sum(1) < 0
            v[flip] = -v[flip]
            ow3, od3, tw3, td3 = run(ctx, g, so, v, 1, margin)
            bad += report(f"{scene} {view}: bounce rays from the hits", ow3, od3, tw3, td3)
            _, _, tw4, td4 = run(ctx, g, so, v, 0, margin)
            print(f"{'':34s}  without the certificate: DDA {td4:7.3f} ms ({tw4 / td4:4.2f} x)")
    print("unflagged differences in all:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
