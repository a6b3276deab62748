"""CPU: the oracle's octree walk (restated shaders/voxels.comp:134-247) cross-checked against an
INDEPENDENT first-hit finder — a binary64 Amanatides-Woo DDA over a dense occupancy grid (oracle/odda.cpp)
that shares no code or data structure with it.  Contract: first non-empty unit voxel
along the ray, its entry t, its axis-aligned entry face."""
import numpy as np
import pytest


def dense_grid(pos):
    lo = pos.min(0).astype(np.int32)
    hi = pos.max(0).astype(np.int32)
    g = np.zeros(tuple(hi - lo + 1), np.uint8)
    p = pos.astype(np.int32) - lo
    g[p[:, 0], p[:, 1], p[:, 2]] = 1
    return g, lo


def random_rays(rng, n, extent, inside_frac=0.3):
    c = extent / 2
    e = float(extent.max())
    o = np.where(rng.random((n, 1)) < inside_frac, rng.uniform(-0.2, 1.2, (n, 3)) * extent,
                 c + rng.normal(size=(n, 3)) * e * 1.5).astype(np.float32)
    target = (rng.uniform(0, 1, (n, 3)) * extent).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def compare(O, pos, mrgb, o, d):
    octree = O.create_octree(pos, mrgb)
    grid, base = dense_grid(pos)
    hit, t, node, normal, iters = O.cast_rays(octree, o, d)
    dhit, dt, daxis, dcell = O.dda_cast(grid, base, o, d)
    started_inside = dhit & (daxis < 0)                      # origin inside a solid voxel: no entry face to compare
    both = hit & dhit & ~started_inside
    # entry cell of the octree hit: the point just behind the entry face
    p = o.astype(np.float64) + t[:, None].astype(np.float64) * d.astype(np.float64)
    inward = p - 0.25 * normal.astype(np.float64) / np.maximum(np.abs(normal).sum(1, keepdims=True), 1)
    cell = np.floor(2 * inward).astype(np.int32)
    single = np.abs(normal).sum(1) == 1
    axis = np.argmax(np.abs(normal), 1)
    agree_hit = hit == dhit
    agree_cell = (cell == dcell).all(1)
    agree_axis = axis == daxis
    ok = agree_hit & (~both | (agree_cell & agree_axis & single))
    # disagreements must be numerical ties: grazing rays where two plane crossings are closer than fp32 resolves
    tie = np.abs(t.astype(np.float64) - dt) < 1e-3
    hard = ~ok & ~(hit & dhit & (tie | started_inside)) & ~started_inside
    return ok, hard, both, t, dt, iters


@pytest.mark.parametrize("name,n", [("castle", 1000000), ("8x8x8", 1000000), ("menger", 1000000), ("room", 1000000), ("monu10", 1000000)])   # SURVEY 8c pin 2: >= 10^6 rays per scene
def test_octree_walk_matches_dense_dda(O, scenes, name, n):
    pos, mrgb, size = scenes.load_scene(name)
    rng = np.random.default_rng(1234)
    o, d = random_rays(rng, n, scenes.world_extent(size))
    ok, hard, both, t, dt, iters = compare(O, pos, mrgb, o, d)
    assert both.sum() > n * 0.2                               # the test really exercises hits
    assert ok.mean() > 0.9995, ok.mean()
    # where octree and DDA disagree on hit/miss the ray grazes a voxel edge/corner: rare, never systematic
    assert hard.mean() < 2e-4, int(hard.sum())
    rel = np.abs(t[both].astype(np.float64) - dt[both]) / np.maximum(dt[both], 1e-3)
    assert np.quantile(rel, 0.999) < 1e-4
    assert iters.max() < 2048           # none of these rays is cut short; the cap itself: test_iteration_cap_is_reached


def test_axis_parallel_rays(O, scenes):
    """1/0 = inf directions (voxels.comp:140): rays along +-x,+-y,+-z through voxel centres."""
    pos, mrgb, size = scenes.load_scene("castle")
    ext = scenes.world_extent(size)
    octree = O.create_octree(pos, mrgb)
    grid, base = dense_grid(pos)
    rng = np.random.default_rng(7)
    n = 6000
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    o = ((rng.integers(0, int(2 * ext.max()), (n, 3)) + 0.5) * 0.5).astype(np.float32)   # voxel centres + offset grid
    o[np.arange(n), axis] = np.where(sign > 0, -3.0, ext.max() + 3.0)
    d = np.zeros((n, 3), np.float32)
    d[np.arange(n), axis] = sign
    hit, t, node, normal, _ = O.cast_rays(octree, o, d)
    dhit, dt, daxis, dcell = O.dda_cast(grid, base, o, d)
    assert np.array_equal(hit, dhit) and hit.sum() > 100
    assert np.allclose(t[hit], dt[hit], rtol=1e-6, atol=1e-6)
    assert np.array_equal(np.argmax(np.abs(normal[hit]), 1), axis[hit])
    assert np.array_equal(normal[hit][np.arange(hit.sum()), axis[hit]], -sign[hit].astype(np.float32))


def test_miss_and_bounds(O, scenes):
    pos, mrgb, size = scenes.load_scene("8x8x8")
    octree = O.create_octree(pos, mrgb)
    vx, vy = (pos[0, 0] + 0.5) * 0.5 + 0.01, (pos[0, 1] + 0.5) * 0.5 - 0.02      # through the first voxel, off every plane
    o = np.array([[100, 100, 100], [-5, 1.1, 1.2], [vx, vy, -5]], np.float32)
    d = np.array([[1, 0, 0], [-1, 0, 0], [0, 0, 1]], np.float32)
    hit, t, node, normal, _ = O.cast_rays(octree, o, d)
    assert hit.tolist() == [False, False, True]
    # max_distance cuts the walk (voxels.comp:171-173)
    hit2, *_ = O.cast_rays(octree, o[2:], d[2:], max_distance=float(t[2]) - 0.5)
    assert hit2.tolist() == [False]
    assert node[2] < 0 and normal[2].tolist() == [0, 0, -1]


def test_origin_inside_solid_voxel_hits_at_time_zero(O):
    pos = np.array([[0, 0, 0]], np.int16)
    mrgb = np.array([[0, 10, 20, 30]], np.uint8)
    octree = O.create_octree(pos, mrgb)
    hit, t, node, normal, _ = O.cast_rays(octree, np.array([[0.25, 0.2, 0.3]], np.float32), np.array([[0.6, 0.0, 0.8]], np.float32))
    assert hit[0] and t[0] == 0.0 and (int(node[0]) & 0xffffff) == (10 << 16 | 20 << 8 | 30)


def test_zero_times_infinity_quirk(O, scenes):
    """with a zero direction component 1/dir = inf, and an origin exactly on a node mid-plane
    makes (center - origin) * inv = 0 * inf = NaN (voxels.comp:140,191).  The restatement keeps that: the
    walk still terminates and reports a hit, but the reported time is NaN."""
    pos, mrgb, size = scenes.load_scene("8x8x8")
    octree = O.create_octree(pos, mrgb)
    hit, t, node, normal, iters = O.cast_rays(octree, np.array([[1, 1, -5]], np.float32), np.array([[0, 0, 1]], np.float32))
    assert hit[0] and np.isnan(t[0]) and node[0] < 0 and iters[0] < 2048


def test_iteration_cap_is_reached(O):
    """voxels.comp:163-169: `if (iterations >= 2048) { out_node = LEAF_BIT; return true; }` — a ray along a 4 096-voxel row,
    through the empty cells beside it, is still walking when trip 2 048 begins: hit with node 0x80000000, the time reached,
    normal unwritten (0 here, U1).  A shorter row of the same kind stays below the cap and misses."""
    n = 4096
    pos = np.zeros((n, 3), np.int16)
    pos[:, 0] = np.arange(n)
    mrgb = np.tile(np.array([[0, 200, 100, 50]], np.uint8), (n, 1))
    o = np.array([[-1, 0.75, 0.25]] * 4, np.float32)
    d = np.array([[1, 0, 0], [1, 1e-5, 1e-5], [1, 1e-4, -2e-5], [1, 0, 1e-6]], np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, t, node, normal, iters = O.cast_rays(O.create_octree(pos, mrgb), o, d)
    assert hit.all() and (iters == 2048).all() and (node == -2 ** 31).all() and (normal == 0).all()
    assert (t > 100).all() and (t < 2048).all()            # far along the row, still inside the root cube
    hit, t, node, normal, iters = O.cast_rays(O.create_octree(pos[:512], mrgb[:512]), o, d)
    assert (~hit).all() and (iters < 2048).all() and (iters > 700).all()
