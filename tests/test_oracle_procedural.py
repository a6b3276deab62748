"""CPU: the oracle's procedural scene (oracle/oprocedural.cpp) — BASELINE config 5's level-7 Menger sponge clipped to 2048^3,
which is too large to store as the reference's octree buffer (261 M nodes = 8.4 GB) and is therefore materialised from the
voxel predicate as the walk touches it.  Pins, in this order:
 1. on sizes that CAN be stored, the lazily materialised octree renders the same frames, bit for bit, as create_octree's
    buffer built from the explicit voxel list (so the full-size frames are the restated shader's frames);
 2. at full size, the walk agrees with an independent binary64 DDA over the predicate (no octree, no grid).
Parity status of the scene itself: BASELINE.json's config, not a reference fixture — parity unpinned by the reference."""
import numpy as np
import pytest

from gpu_voxel_raytracer_amd.scenes import CONFIG5 as FULL       # level, clip, colour, emissive period
from gpu_voxel_raytracer_amd.scenes import config5_cameras

MRGB = FULL[2]


def voxel_list(O, level, clip, mrgb, period):
    side = min(3 ** level, clip or 3 ** level)
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    solid, word = O.menger_cells(level, clip, mrgb, period, g)
    w = word[solid].view(np.uint32)
    return g[solid].astype(np.int16), np.stack([(w >> 24) & 0x7f, (w >> 16) & 0xff, (w >> 8) & 0xff, w & 0xff], 1).astype(np.uint8), side


def full_size_cameras():
    return config5_cameras()


def primary_rays(O, cam, w, h, n, rng):
    """n primary rays of a w x h frame seen through `cam`, as voxels.comp:299-303 makes them (float32 throughout)."""
    b = O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h)
    x = rng.integers(0, w, n).astype(np.float32)[:, None]
    y = rng.integers(0, h, n).astype(np.float32)[:, None]
    d = (x * b[0:3] - y * b[3:6]).astype(np.float32) + b[6:9]
    d = (d / np.sqrt((d * d).sum(1, keepdims=True, dtype=np.float32))).astype(np.float32)
    return np.broadcast_to(np.asarray(cam[0], np.float32), d.shape).copy(), d


def compare_with_dda(o, d, hit, t, node, normal, dda, words_of):
    """The walk's result against the predicate DDA's: same hit flag, same voxel (through its leaf word), same face, same t —
    except on grazing ties.  Returns (fraction fully agreeing, fraction of hard disagreements, number of compared hits)."""
    dhit, dt, daxis, dcell = dda
    inside = dhit & (daxis < 0)
    both = hit & dhit & ~inside
    single = np.abs(normal).sum(1) == 1
    axis = np.argmax(np.abs(normal), 1)
    sign_ok = normal[np.arange(len(o)), axis] == -np.sign(d[np.arange(len(o)), axis])
    want_word = words_of(dcell)
    ok = (hit == dhit) & (~both | ((node == want_word) & (axis == daxis) & single & sign_ok))
    tie = np.abs(t.astype(np.float64) - dt) < 1e-3 * np.maximum(dt, 1.0)
    hard = ~ok & ~(hit & dhit & (tie | inside)) & ~inside
    rel = np.abs(t[both].astype(np.float64) - dt[both]) / np.maximum(dt[both], 1e-3)
    return ok.mean(), hard.mean(), int(both.sum()), rel


@pytest.mark.parametrize("level,clip,period", [(4, 64, 37), (3, 20, 5), (4, 0, 0), (5, 200, 301), (1, 2, 2), (3, 1, 0)])
def test_lazy_octree_renders_like_the_stored_octree(O, noise, level, clip, period):
    pos, m, side = voxel_list(O, level, clip, (0, 40, 200, 90), period)
    octree = O.create_octree(pos, m)
    assert O.voxel_depth(pos) == O.menger_depth(level, clip)
    ext = np.float32(side / 2)
    cam = (np.array([-0.4, 0.8, -0.7], np.float32) * ext + ext / 2, np.array([0.9, -0.55, 1.2], np.float32), 1.1)
    w, h, bounces = 160, 112, 4
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    for frame in (1, 2):
        u.frame_number = frame
        a = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
        b = O.trace_menger(level, clip, (0, 40, 200, 90), period, noise, u, bounces, (0, 0, w, h))
        for x, y in zip(a[:3], b[:3]):
            assert ((x == y) | (np.isnan(x) & np.isnan(y))).all()
        assert a[3] == b[3] and (a[1][..., 3] >= 0).any()
    rng = np.random.default_rng(5)
    o = (rng.uniform(-0.3, 1.3, (20000, 3)) * side / 2).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r0 = O.cast_rays(octree, o, d)
    r1 = O.cast_rays_menger(level, clip, (0, 40, 200, 90), period, o, d)
    for x, y in zip(r0, r1):
        assert np.array_equal(x, y, equal_nan=True)


def test_full_size_walk_matches_predicate_dda(O):
    level, clip, mrgb, period = FULL
    rng = np.random.default_rng(11)
    n = 60000
    for name, cam in full_size_cameras().items():
        o, d = primary_rays(O, cam, 7680, 4320, n, rng)
        hit, t, node, normal, iters = O.cast_rays_menger(level, clip, mrgb, period, o, d)
        dda = O.dda_menger(level, clip, o, d)
        ok, hard, compared, rel = compare_with_dda(o, d, hit, t, node, normal, dda,
                                                   lambda cells: O.menger_cells(level, clip, mrgb, period, cells)[1])
        assert compared > n * 0.15, (name, compared)
        assert ok > 0.9995 and hard < 2e-4, (name, ok, hard)
        assert np.quantile(rel, 0.999) < 1e-4
        assert iters.max() < 2048
        assert ((node[hit] >> 30) & 1).sum() > 0 or name == "tunnel"     # emissive seeds are seen
    assert O.menger_depth(level, clip) == 11                            # 12 node levels <= MAX_DEPTH 16 (voxels.comp:3)
