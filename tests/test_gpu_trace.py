"""GPU parity, trace stage: libvxrt's HIP path-trace kernel (through the C ABI) against the CPU oracle's
restatement of shaders/voxels.comp on the same seeded inputs.  Bar: BIT-EXACT — node words and normals
are integers/flags, and every float follows include/vxrt_detmath.h on both sides."""
import numpy as np
import pytest

from conftest import assert_bits_equal, require_variants

pytestmark = pytest.mark.gpu


def render_both(O, H, scenes, noise, name, width, height, bounces, frames=(1,), specularity=0.0, camera="bench",
                sun_strength=None, emit=None, crop=None):
    from gpu_voxel_raytracer_amd import Context, Camera, TRACE, SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE
    pos, mrgb, size = scenes.load_scene(name)
    cam_pos, cam_dir, fov = {"bench": scenes.bench_camera, "close": scenes.close_camera}[camera](size) \
        if camera != "start" else scenes.reference_start_camera()
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.specularity = specularity
    if sun_strength is not None:
        u.sun_strength = sun_strength
    if emit is not None:
        u.emit_strength = emit
    u.set_camera(cam_pos, O.camera_axis_scaled(cam_pos, cam_dir, fov, width, height))
    out = []
    with Context(width, height, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(cam_pos, cam_dir, fov)
        ctx.uniforms.specularity = specularity
        if sun_strength is not None:
            ctx.uniforms.sun_strength = sun_strength
        if emit is not None:
            ctx.uniforms.emit_strength = emit
        for f in frames:
            ctx.set_frame_number(f - 1)
            ctx.reset_stats()
            ctx.render(TRACE)
            g = [ctx.read(SAMPLED_COLOR), ctx.read(NORMAL_DEPTH), ctx.read(ALBEDO_NODE)]
            rays = ctx.stats().rays
            u.frame_number = f
            c = crop or (0, 0, width, height)
            ref = O.trace(octree, noise, u, width, height, bounces, crop=c)
            g = [a[c[1]:c[3], c[0]:c[2]] for a in g]
            out.append((g, rays, ref))
    return out


@pytest.mark.parametrize("name,w,h,bounces,camera", [
    ("8x8x8", 64, 48, 3, "close"),
    ("castle", 160, 96, 3, "bench"),
    ("castle", 96, 64, 8, "close"),
    ("menger", 256, 144, 4, "bench"),
    ("menger", 128, 80, 4, "close"),
    ("monu10", 160, 96, 3, "close"),
    ("room", 128, 96, 3, "close"),     # emissive voxels
    ("3x3x3", 128, 128, 1, "bench"),   # BASELINE config 1 scene, 1 bounce
    ("3x3x3", 256, 256, 1, "bench"),   # ... at config 1's own size (the reference's CPU caster for it: tests/test_cpu_rs.py)
])
def test_trace_bit_exact(O, H, scenes, noise, name, w, h, bounces, camera):
    for (g, rays, ref) in render_both(O, H, scenes, noise, name, w, h, bounces, frames=(1, 2, 513), camera=camera):
        color, nd, alb, ref_rays = ref
        # node words (bit pattern in albedo.w) and normals: exact integers / flags
        assert np.array_equal(g[2][..., 3].view(np.uint32), alb[..., 3].view(np.uint32)), "leaf words differ"
        assert_bits_equal(g[1], nd, "normal/depth")
        assert_bits_equal(g[2][..., :3], alb[..., :3], "albedo")
        assert_bits_equal(g[0], color, "sampled colour")
        assert rays == ref_rays, "ray count"


@pytest.mark.parametrize("name", ["custom", "teapot", "nature", "chr_knight", "doom", "shelf", "monu9", "chr_sword", "monu1"])
def test_trace_bit_exact_remaining_reference_scenes(O, H, scenes, noise, name):
    """The other nine files of the reference's vox/ (with the eight above: all fifteen), two views each."""
    for camera in ("bench", "close"):
        for (g, rays, ref) in render_both(O, H, scenes, noise, name, 192, 112, 4, frames=(1, 2), camera=camera, specularity=0.05):
            assert_bits_equal(g[0], ref[0], f"colour {name} {camera}")
            assert_bits_equal(g[1], ref[1], f"nd {name} {camera}")
            assert_bits_equal(g[2], ref[2], f"albedo {name} {camera}")
            assert rays == ref[3]
        assert (g[1][..., 3] >= 0).any(), name     # the view sees the model


def test_trace_specular_and_no_sun(O, H, scenes, noise):
    for kw in (dict(specularity=0.5), dict(specularity=1.0), dict(sun_strength=0.0), dict(emit=0.0, specularity=0.25)):
        for (g, rays, ref) in render_both(O, H, scenes, noise, "castle", 128, 80, 4, camera="close", **kw):
            assert_bits_equal(g[0], ref[0], f"colour {kw}")
            assert_bits_equal(g[1], ref[1], f"nd {kw}")
            assert rays == ref[3]


def test_trace_camera_inside_root_and_odd_size(O, H, scenes, noise):
    # the reference's start camera sits inside the root cube; 150x70 is not a multiple of the 16x16 tile
    for (g, rays, ref) in render_both(O, H, scenes, noise, "castle", 150, 70, 3, camera="start"):
        assert_bits_equal(g[0], ref[0], "colour")
        assert_bits_equal(g[1], ref[1], "nd")
        assert rays == ref[3]


def test_trace_golden_fixture(O, H, scenes, noise):
    """The committed oracle frames (tests/golden/frames_menger.npz) are reproduced by the GPU."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "frames_menger.npz"))
    w, h, b = int(z["width"]), int(z["height"]), int(z["bounces"])
    res = render_both(O, H, scenes, noise, "menger", w, h, b, frames=(1, 2, 8))
    for f, (g, rays, ref) in zip((1, 2, 8), res):
        assert_bits_equal(g[0], z[f"f{f}_color"], f"golden colour f{f}")
        assert_bits_equal(g[1], z[f"f{f}_nd"], f"golden nd f{f}")
        assert rays == int(z["rays"][f - 1])


def test_full_size_bench_config_crops(O, H, scenes, noise):
    """BASELINE config 2 at full size (menger, 1920x1080, 4 bounces): the oracle checks three 1920x24
    strips (it would need minutes for the whole frame); the whole frame is checked through properties."""
    from gpu_voxel_raytracer_amd import Context, Camera, TRACE, SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE
    W, Hh, B = 1920, 1080, 4
    pos, mrgb, size = scenes.load_scene("menger")
    cam_pos, cam_dir, fov = scenes.bench_camera(size)
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam_pos, O.camera_axis_scaled(cam_pos, cam_dir, fov, W, Hh))
    u.frame_number = 1
    with Context(W, Hh, max_bounces=B, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(cam_pos, cam_dir, fov)
        ctx.render(TRACE)
        color, nd, alb = ctx.read(SAMPLED_COLOR), ctx.read(NORMAL_DEPTH), ctx.read(ALBEDO_NODE)
        st = ctx.stats()
    sq_err, n = 0.0, 0
    for y0 in (300, 528, 760):
        c, d, a, _ = O.trace(octree, noise, u, W, Hh, B, crop=(0, y0, W, y0 + 24))
        assert_bits_equal(color[y0:y0 + 24], c, f"colour strip {y0}")
        assert_bits_equal(nd[y0:y0 + 24], d, f"nd strip {y0}")
        sq_err += float(((color[y0:y0 + 24, :, :3] - c[..., :3]) ** 2).sum())
        n += c[..., :3].size
    assert (sq_err / n) ** 0.5 <= 1e-3  # BASELINE.json tolerance (bit-exact => 0)
    # whole-frame properties: finite radiance, alpha 1, hit <=> depth >= 0 <=> leaf bit, ray accounting
    assert np.isfinite(color).all() and (color[..., 3] == 1).all() and (color[..., :3] >= 0).all()
    node = alb[..., 3].view(np.int32)
    hit = nd[..., 3] >= 0
    assert ((node < 0) == hit).all() and (node[~hit] == 0xffffff).all()
    assert (nd[~hit][:, :3] == 2.0 ** 30).all() and (nd[~hit][:, 3] == -1).all()
    nrm = nd[hit][:, :3]
    assert np.isin(nrm, (-1.0, 0.0, 1.0)).all() and (np.abs(nrm).sum(1) >= 1).all()
    assert W * Hh <= st.rays <= 2 * B * W * Hh and st.pixels == W * Hh


@pytest.mark.parametrize("env", [
    {"VXRT_TRACE_VARIANT": "0", "VXRT_TILE_ORDER": "0"},      # monolithic kernel, tiles in raster order
    {"VXRT_TRACE_VARIANT": "0", "VXRT_TILE_ORDER": "1"},      # ... longest-tile-first (default)
    {"VXRT_SPREAD": "256"},                                   # ... with the tiles that walk spread over the whole launch, sky tiles in between
    {"VXRT_SPREAD": "128", "VXRT_INFLIGHT": "2"},             # ... over half of it
    {"VXRT_SPREAD": "0"},                                     # ... not at all
    {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0x1"},   # wavefront: primary launch + one launch for all segments
    {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0x5"},   # wavefront: queues compacted before segments 0 and 2
    {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0xff", "VXRT_TRACE_BLOCKS": "64"},  # every segment its own launch; few waves loop over many chunks
    {"VXRT_TRACE_VARIANT": "3"},                                                        # ray queues: shade / trace launches, per-lane refill
    {"VXRT_TRACE_VARIANT": "3", "VXRT_TRACE_BLOCKS": "3", "VXRT_SHADE_BLOCKS": "5"},    # ... few waves: many refills per lane, several trips
    {"VXRT_TRACE_VARIANT": "3", "VXRT_INFLIGHT": "3"},                                  # ... with frames in flight
    {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0x3", "VXRT_INFLIGHT": "4"},       # wavefront with frames in flight
    {"VXRT_TRACE_VARIANT": "3", "VXRT_RAYS_PER_WAVE": "1000", "VXRT_INFLIGHT": "2"},    # fat trace waves: ~16 refills per lane
    {"VXRT_TRACE_VARIANT": "4"},                                                        # monolithic head + compacted tail (from hit 1)
    {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_FROM": "2", "VXRT_INFLIGHT": "4"},           # ... tail from hit 2, frames in flight
    {"VXRT_TRACE_VARIANT": "4", "VXRT_TRACE_BLOCKS": "16", "VXRT_TILE_ORDER": "0"},     # ... few tail waves, raster tile order
    {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_SPLIT": "0xc", "VXRT_INFLIGHT": "3"},        # ... tail compacted again at segments 2 and 3
    {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_FROM": "0"},                                 # ... tail from the first hit: the head casts primary rays only
    {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_FROM": "0", "VXRT_INFLIGHT": "3", "VXRT_TAIL_SPLIT": "0x6"},
    {"VXRT_TRACE_VARIANT": "4", "VXRT_LONG_TILES": "100", "VXRT_INFLIGHT": "2"},        # ... the longest tenth of the tiles as an all-in-one grid on a second stream
    {"VXRT_TRACE_VARIANT": "4", "VXRT_LONG_TILES": "500", "VXRT_SPREAD": "0", "VXRT_TAIL_CAPACITY": "64"},
    {"VXRT_TRACE_VARIANT": "4", "VXRT_FUSED_TAIL": "1"},                                # head and tail as one grid of persistent waves (fused_kernel)
    {"VXRT_TRACE_VARIANT": "4", "VXRT_FUSED_TAIL": "1", "VXRT_INFLIGHT": "3", "VXRT_TAIL_CAPACITY": "64"},   # ... launches in flight, one chunk per shard: most paths stay in the head
    {"VXRT_TRACE_VARIANT": "5"},                                                        # monolithic head + path_kernel (lanes refilled path by path)
    {"VXRT_TRACE_VARIANT": "5", "VXRT_PATH_BLOCKS": "1", "VXRT_INFLIGHT": "2"},         # ... four waves take the whole queue: many refills per lane
    {"VXRT_TRACE_VARIANT": "5", "VXRT_TAIL_FROM": "2", "VXRT_BATCH": "4"},              # ... tail from hit 2 (the API calls here are single frames)
    {"VXRT_WIDE": "1"},                                                                 # the wide scene records (two levels per record), default tracer
    {"VXRT_WIDE": "1", "VXRT_TRACE_VARIANT": "0"},                                      # ... all-in-one kernel
    {"VXRT_WIDE": "1", "VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_SPLIT": "0xc", "VXRT_INFLIGHT": "3", "VXRT_TAIL_CAPACITY": "128"},
])
def test_every_trace_variant_is_bit_exact(O, H, scenes, noise, monkeypatch, env):
    """The scheduling variants of the tracer (read from the environment when a context is created) change
    which lane / launch executes a path segment, never a result."""
    require_variants(H, env)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, w, h, b, cam in (("castle", 200, 120, 5, "close"), ("room", 136, 104, 3, "close"), ("menger", 256, 144, 4, "bench")):
        for (g, rays, ref) in render_both(O, H, scenes, noise, name, w, h, b, frames=(1, 2, 3), camera=cam, specularity=0.1):
            assert_bits_equal(g[0], ref[0], f"colour {name} {env}")
            assert_bits_equal(g[1], ref[1], f"nd {name} {env}")
            assert_bits_equal(g[2], ref[2], f"albedo {name} {env}")
            assert rays == ref[3]


def test_tile_cost_feedback(H, scenes, noise):
    """The longest-tile-first scheduler's input: per-tile durations of the last frame (vxrt_debug_tile_costs)."""
    import ctypes as C
    from gpu_voxel_raytracer_amd import Context, Camera, TRACE
    pos, mrgb, size = scenes.load_scene("castle")
    w, h = 320, 192
    with Context(w, h, max_bounces=3, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*scenes.bench_camera(size))
        ctx.render(TRACE)
        ctx.render(TRACE)
        n = (w // 16) * (h // 16)
        cost = np.zeros(n, np.uint32)
        H._check(H.lib().vxrt_debug_tile_costs(ctx._h, cost.ctypes.data_as(C.c_void_p), C.c_size_t(n)), "vxrt_debug_tile_costs")
        hit = (ctx.read(1)[..., 3] >= 0).reshape(h // 16, 16, w // 16, 16).any(axis=(1, 3)).ravel()
        assert (cost > 0).all() and hit.any() and (~hit).any()
        assert np.median(cost[hit]) > 2 * np.median(cost[~hit])     # tiles that see geometry take longer than sky tiles


def _render_voxels(O, noise, pos, mrgb, cam, w, h, bounces, frames=(1,)):
    from gpu_voxel_raytracer_amd import Context, Camera, TRACE
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    out = []
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        for f in frames:
            ctx.set_frame_number(f - 1)
            ctx.reset_stats()
            ctx.render(TRACE)
            got = [ctx.read(i) for i in range(3)]
            u.frame_number = f
            ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
            out.append((got, ctx.stats().rays, ref))
    return out


@pytest.mark.parametrize("wide", ["0", "1"])     # the 8-byte scene records / the wide ones (odd and even numbers of tree levels)
def test_empty_scene_and_single_voxel(O, H, noise, monkeypatch, wide):
    require_variants(H, wide=wide)
    monkeypatch.setenv("VXRT_WIDE", wide)
    """Edge cases of the scene format: no voxels at all (root node of eight empty slots, depth 0) and one voxel."""
    f32 = np.float32
    cam = (np.array([3, 2, -4], f32), np.array([-3, -2, 4], f32), 1.0)
    for pos, mrgb in ((np.zeros((0, 3), np.int16), np.zeros((0, 4), np.uint8)),
                      (np.array([[0, 0, 0]], np.int16), np.array([[0x40, 200, 100, 50]], np.uint8)),
                      (np.array([[-1, -1, -1]], np.int16), np.array([[0, 1, 2, 3]], np.uint8))):
        for (g, rays, ref) in _render_voxels(O, noise, pos, mrgb, cam, 96, 64, 3):
            for a, b, label in zip(g, ref[:3], ("colour", "nd", "albedo")):
                assert_bits_equal(a, b, f"{label} n={len(pos)}")
            assert rays == ref[3]
            if len(pos) == 0:
                assert (g[1][..., 3] == -1).all()


@pytest.mark.parametrize("wide", ["0", "1"])     # the 8-byte scene records / the wide ones (odd and even numbers of tree levels)
def test_deep_octree_negative_coordinates_and_depth_limit(O, H, noise, monkeypatch, wide):
    require_variants(H, wide=wide)
    monkeypatch.setenv("VXRT_WIDE", wide)
    """Random voxels on both sides of the origin (all eight root octants used), a depth-10 tree, and the deepest
    tree i16 coordinates allow (depth 15)."""
    rng = np.random.default_rng(42)
    f32 = np.float32
    pos = rng.integers(-40, 40, (6000, 3)).astype(np.int16)
    mrgb = rng.integers(0, 256, (6000, 4)).astype(np.uint8)
    cam = (np.array([45, 30, -50], f32), np.array([-45, -30, 50], f32), 1.1)
    for (g, rays, ref) in _render_voxels(O, noise, pos, mrgb, cam, 160, 96, 4, frames=(1, 7)):
        for a, b, label in zip(g, ref[:3], ("colour", "nd", "albedo")):
            assert_bits_equal(a, b, label)
        assert rays == ref[3]
    # depth 10: a bumpy slab of voxels far from the origin
    xs, ys = np.meshgrid(np.arange(900, 960), np.arange(900, 960))
    far = np.stack([xs.ravel(), ys.ravel(), 900 + (xs.ravel() * 7 + ys.ravel() * 3) % 5], 1).astype(np.int16)
    far_m = np.tile(np.array([[0, 255, 128, 0]], np.uint8), (len(far), 1))
    assert O.voxel_depth(far) == 10
    cam = (np.array([462, 470, 438], f32), np.array([0.1, -0.25, 1.0], f32), 1.2)
    for (g, rays, ref) in _render_voxels(O, noise, far, far_m, cam, 128, 96, 2):
        assert_bits_equal(g[0], ref[0], "colour depth 10")
        assert_bits_equal(g[1], ref[1], "nd depth 10")
        assert (g[1][..., 3] >= 0).any()
    # the extremes of i16 both give depth 15 = MAX_DEPTH - 1 pushes (voxels.comp:3): the deepest tree there is
    for extreme in (32767, -32768):
        p15 = np.array([[extreme, 3, -2], [0, 0, 0]], np.int16)
        m15 = np.array([[0, 10, 200, 30], [0x40, 255, 255, 255]], np.uint8)
        assert O.voxel_depth(p15) == 15
        cam = (np.array([extreme / 2 + (3 if extreme > 0 else -3), 4, -6], f32), np.array([-0.5 if extreme > 0 else 0.5, -0.4, 1.0], f32), 0.9)
        for (g, rays, ref) in _render_voxels(O, noise, p15, m15, cam, 64, 48, 3):
            assert_bits_equal(g[0], ref[0], "colour depth 15")
            assert_bits_equal(g[1], ref[1], "nd depth 15")
            assert rays == ref[3]


@pytest.mark.parametrize("wide", ["0", "1"])
def test_zero_times_infinity_rays_through_the_gpu(O, H, scenes, noise, monkeypatch, wide):
    """An axis-aligned camera on integer coordinates sends its centre rays exactly along +z through node mid-planes:
    (center - origin) * (1/0) = 0 * inf = NaN in the shader (voxels.comp:140,191).  The kernel reproduces the
    oracle's NaNs and everything around them."""
    require_variants(H, wide=wide)
    monkeypatch.setenv("VXRT_WIDE", wide)
    pos, mrgb, size = scenes.load_scene("8x8x8")
    f32 = np.float32
    cam = (np.array([1, 1, -5], f32), np.array([0, 0, 1], f32), 1.0)
    for (g, rays, ref) in _render_voxels(O, noise, pos, mrgb, cam, 64, 64, 3):
        assert np.isnan(ref[1][..., 3]).any() or np.isnan(ref[0]).any()      # the quirk really is exercised
        for a, b, label in zip(g, ref[:3], ("colour", "nd", "albedo")):
            assert_bits_equal(a, b, label)
        assert rays == ref[3]


def cap_scene():
    """A row of 4 096 voxels along x (depth 12).  A ray that runs along the row through the EMPTY cells beside it enters
    every leaf-parent node of the row (descend, step to the second empty octant, pop: >= 3 trips per two cells), never hits,
    and is still inside the root cube when its 2 048th trip begins (tests/test_oracle_traversal.py asserts this in the oracle)."""
    n = 4096
    pos = np.zeros((n, 3), np.int16)
    pos[:, 0] = np.arange(n)
    return pos, np.tile(np.array([[0, 200, 100, 50]], np.uint8), (n, 1))


CAP_RAYS_O = np.array([[-1, 0.75, 0.25]] * 4, np.float32)
CAP_RAYS_D = np.array([[1, 0, 0], [1, 1e-5, 1e-5], [1, 1e-4, -2e-5], [1, 0, 1e-6]], np.float32)   # rows 0 and 3: a zero component -> walk_step


@pytest.mark.parametrize("env", [{}, {"VXRT_TRACE_VARIANT": "0"}, {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0x3"},
                                 {"VXRT_TRACE_VARIANT": "3"}, {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_FROM": "0"},
                                 {"VXRT_TRACE_VARIANT": "5"}, {"VXRT_WIDE": "1"}])
def test_iteration_cap(O, H, noise, monkeypatch, env):
    """voxels.comp:163-169: the 2 048th trip of the loop returns TRUE with out_node = LEAF_BIT, the time the walk is at, and
    the normal unwritten (defined as 0 here and in the oracle, U1 — the shader leaves it undefined; the compiled module executed over
    zeroed memory gives the same 0: tests/test_oracle_spirv_exec.py::test_undefined_reads_matter_only_at_the_trip_cap).
    Probe rays (both walks: regular rays -> walkf_step, a zero direction component -> walk_step) and a rendered frame in
    which hundreds of primary rays are capped and their paths go on from the capped "hit" — for every tracer variant."""
    from gpu_voxel_raytracer_amd import Camera, Context
    require_variants(H, env)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pos, mrgb = cap_scene()
    octree = O.create_octree(pos, mrgb)
    d = (CAP_RAYS_D / np.linalg.norm(CAP_RAYS_D, axis=1, keepdims=True)).astype(np.float32)
    hit, t, node, normal, iters = O.cast_rays(octree, CAP_RAYS_O, d)
    assert hit.all() and (iters == 2048).all() and (node == -2 ** 31).all() and (normal == 0).all()     # the oracle is capped
    with Context(64, 64, max_bounces=2, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ghit, gt, gnode, gnormal = ctx.cast_rays(CAP_RAYS_O, d)
    assert ghit.all() and (gnode == -2 ** 31).all() and (gnormal == 0).all()
    assert_bits_equal(gt, t, "time at the cap")
    # a frame: camera looking along the row, 0.01 rad field of view
    cam = (np.array([-1, 0.6, 0.25], np.float32), np.array([1, 0, 0], np.float32), 0.01)
    for (g, rays, ref) in _render_voxels(O, noise, pos, mrgb, cam, 96, 64, 3, frames=(1, 2)):
        capped = ref[2][..., 3].view(np.uint32) == 0x80000000
        assert capped.sum() > 500 and (ref[1][..., 3] >= 0).sum() > capped.sum() + 500          # capped and ordinary hits
        assert (ref[1][capped][:, :3] == 0).all() and (ref[2][capped][:, :3] == 0).all()       # normal 0 (U1), albedo of rgb 0
        gcap = g[2][..., 3].view(np.uint32) == 0x80000000
        assert np.array_equal(gcap, capped)
        for a, b, label in zip(g, ref[:3], ("colour", "nd", "albedo")):
            assert_bits_equal(a, b, f"{label} with capped rays {env}")
        assert rays == ref[3]


def test_tracer_field_of_the_config(O, H, scenes, noise):
    """vxrt_config.tracer picks the scheduling variant (0 = auto: monolithic head + compacted tail); same image either way."""
    from gpu_voxel_raytracer_amd import Context, Camera, TRACE, VxrtError
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    imgs = []
    for tracer, bounces in ((0, 8), (1, 8), (2, 8), (3, 8), (4, 8), (5, 8)):
        if tracer in (2, 3, 5) and not H.has_variants():     # not in the product library: refused, loudly
            with pytest.raises(VxrtError, match="VXRT_VARIANTS"):
                Context(128, 80, max_bounces=bounces, noise=noise, tracer=tracer)
            require_variants(H, tracer=tracer)               # ... and run in the -DVXRT_VARIANTS=1 build, loaded beside it
        with Context(128, 80, max_bounces=bounces, noise=noise, tracer=tracer) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)
            imgs.append(ctx.read(0))
    for im in imgs[1:]:
        assert_bits_equal(im, imgs[0], "tracer variants")
    with pytest.raises(VxrtError):
        Context(64, 64, tracer=7)


def test_tail_queue_full_and_growth(O, H, scenes, noise):
    """The compacted tail's path queue is sized by need (an eighth of the worst case to start with), not for the worst case:
    a path that finds its shard full stays in the head kernel — same image — and the queues grow before the stream's next launch.
    (a) a capacity of 64 records per shard forces the queue-full path on most paths of every launch: bit-exact against the oracle;
    (b) a view filled with geometry overflows the initial sizing, the queues grow, and later launches fit."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    w, h, bounces = 256, 160, 6                 # 6 bounces: the tail compacts a second time (second queue)
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    for batch, inflight in ((1, 1), (4, 2)):
        with Context(w, h, max_bounces=bounces, noise=noise, tracer=4, frames_per_launch=batch, frames_in_flight=inflight) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            ctx.set_option(H.OPT_TAIL_CAPACITY, 64)
            ctx.render_frames(TRACE, 2 * batch)
            st = ctx.stats()
            assert st.queue_overflow_paths > 1000 and st.queue_bytes <= inflight * 2 * (64 * 64 + 1) * 64
            u.frame_number = 2 * batch
            ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
            for i, label in enumerate(("colour", "nd", "albedo")):
                assert_bits_equal(ctx.read(i), ref[i], f"{label} with full queues")
            assert st.rays == sum(O.trace(octree, noise, (setattr(u, "frame_number", f) or u), w, h, bounces, crop=(0, 0, w, h))[3]
                                  for f in range(1, 2 * batch + 1))
    # (b) automatic sizing: menger from close by at 1920x1080 — more than an eighth of the pixels are alive at their second hit
    pos, mrgb, size = scenes.load_scene("menger")
    cam = scenes.close_camera(size)
    W, Hh, B = 1920, 1080, 4
    with Context(W, Hh, max_bounces=B, noise=noise, frames_per_launch=16, frames_in_flight=2) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        first = ctx.stats().queue_bytes
        assert first <= 1 << 30                    # <= 1 GB at 1080p, 16 frames per launch x 2 in flight (worst case: 8.5 GB)
        ctx.render_frames(TRACE, 32)
        a = ctx.stats()
        ctx.reset_stats()
        ctx.render_frames(TRACE, 64)
        b = ctx.stats()
        assert a.queue_overflow_paths > 0 and a.queue_bytes > first     # the first launches did not fit, the queues grew ...
        assert b.queue_overflow_paths == 0 and b.queue_bytes == a.queue_bytes    # ... and now everything fits
        ctx.set_frame_number(0)
        ctx.render(TRACE)
        u = O.Uniforms.default()
        u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], W, Hh))
        u.frame_number = 1
        c, d, al, _ = O.trace(octree := O.create_octree(pos, mrgb), noise, u, W, Hh, B, crop=(0, 500, W, 516))
        assert_bits_equal(ctx.read(0)[500:516], c, "colour after the queues grew")
    # an 8K single context with 32 frames per launch x 3 in flight can be created (worst-case queues would need 204 GB)
    with Context(7680, 4320, max_bounces=4, frames_per_launch=32, frames_in_flight=3) as ctx:
        assert ctx.stats().queue_bytes < 40 << 30


@pytest.mark.parametrize("name,view", [("menger", "bench"), ("menger", "close"), ("castle", "bench"), ("monu10", "start"), ("room", "close"), ("3x3x3", "bench")])
def test_sky_cull_changes_no_value(O, H, scenes, noise, name, view):
    """VXRT_OPT_SKY_CULL: a pixel whose primary ray provably misses takes voxels.comp's miss outputs without walking (csrc/trace.hip:
    primary_miss_is_certain).  Frames with the cull on and off are equal value for value, ray counts included, from outside, from
    close up and from inside the root cube; the box it tests against holds every voxel and is tight to one leaf-parent cell."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene(name)
    cam = {"bench": scenes.bench_camera, "close": scenes.close_camera}[view](size) if view != "start" else scenes.reference_start_camera()
    w, h, bounces = 320, 200, 3
    imgs = {}
    for cull in (1, 0):
        with Context(w, h, max_bounces=bounces, noise=noise, frames_per_launch=3) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.set_option(H.OPT_SKY_CULL, cull)
            ctx.camera = Camera(*cam)
            ctx.render_frames(TRACE, 3)
            st = ctx.stats()
            imgs[cull] = ([ctx.read(i) for i in range(3)], st.rays)
            if cull:
                lo, hi = np.array(st.cull_box_min), np.array(st.cull_box_max)
                vmin, vmax = pos.min(0) * 0.5, (pos.max(0) + 1) * 0.5          # a voxel p occupies [p / 2, p / 2 + 0.5)
                assert st.cull_box_valid and (lo <= vmin).all() and (hi >= vmax).all() and (vmin - lo < 1.0).all() and (hi - vmax < 1.0).all()
    for a, b, label in zip(imgs[1][0], imgs[0][0], ("colour", "nd", "albedo")):
        assert_bits_equal(a, b, f"{label}: sky cull on vs off, {name} {view}")
    assert imgs[1][1] == imgs[0][1]
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 3
    ref = O.trace(O.create_octree(pos, mrgb), noise, u, w, h, bounces, crop=(0, 0, w, h))
    for a, b, label in zip(imgs[1][0], ref[:3], ("colour", "nd", "albedo")):
        assert_bits_equal(a, b, f"{label}: sky cull vs oracle, {name} {view}")


def test_sky_cull_beside_the_iteration_cap_scene(O, H, noise):
    """The 4 096-voxel row (a depth-12 tree whose rays CAN reach the 2 048-trip cap): the cull's box is made of level-7 cells there,
    rays inside it walk (and are capped as in the oracle), rays outside it are culled — among them rays that run the whole length of
    the row a few cells away from it, the worst case for the trip bound in the proof.  Every pixel equals the oracle's."""
    pos, mrgb = cap_scene()
    f32 = np.float32
    for cam in ((np.array([-40, 33, 33], f32), np.array([1, 0.001, 0.001], f32), 0.9),      # along the row, skimming past the box's corner
                (np.array([-40, 30, 20], f32), np.array([1, -0.012, -0.008], f32), 0.9),
                (np.array([-1, 20.0, 0.25], f32), np.array([1, -0.009, 0], f32), 0.4),
                (np.array([-1, 0.6, 0.25], f32), np.array([1, 0, 0], f32), 0.05)):
        for (g, rays, ref) in _render_voxels(O, noise, pos, mrgb, cam, 160, 96, 2, frames=(1, 2)):
            for a, b, label in zip(g, ref[:3], ("colour", "nd", "albedo")):
                assert_bits_equal(a, b, f"{label} beside the row")
            assert rays == ref[3]


@pytest.mark.parametrize("sun_size", [0.02, 0.05, 0.2, 0.8, 3.0])
def test_sun_power_shortcut_with_other_sun_sizes(O, H, scenes, noise, sun_size):
    """sun_power_of (csrc/trace_common.h): pow(x, 1 / sun_size^2) of voxels.comp:378-381 is skipped where a host-made bound proves
    it exactly +0.  The default sun_size (0.05: exponent 400) is what every other test runs; here the exponent is 2500, 25, 1.56 and
    0.11 (no shortcut: the bound exists only for exponents > 1), with the camera turned towards the sun so that the disc, its rim and
    the exactly-zero region are all in the frame — sky pixels (culled and walked ones) equal the oracle's bit for bit."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("castle")
    w, h = 256, 160
    u = O.Uniforms.default()
    u.sun_size = sun_size
    sun = np.array([np.cos(u.sun_yaw) * np.cos(u.sun_pitch), -np.sin(u.sun_pitch), np.sin(u.sun_yaw) * np.cos(u.sun_pitch)], np.float32)
    cam_pos = scenes.bench_camera(size)[0]
    cam = (cam_pos, (-sun + np.array([0.15, -0.1, 0.05], np.float32)).astype(np.float32), 1.3)     # looking at the sun, a little off-centre
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 1
    ref = O.trace(O.create_octree(pos, mrgb), noise, u, w, h, 2, crop=(0, 0, w, h))
    sky = ref[1][..., 3] < 0
    lum = ref[0][..., 0][sky]
    assert sky.mean() > 0.5 and lum.max() > lum.min()                      # the sun's disc shows in the sky
    if sun_size <= 0.05:
        assert (lum == lum.min()).mean() > 0.1                             # ... and so does the region where its power is exactly 0
    for cull in (1, 0):
        with Context(w, h, max_bounces=2, noise=noise) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.set_option(H.OPT_SKY_CULL, cull)
            ctx.camera = Camera(*cam)
            ctx.uniforms.sun_size = sun_size
            ctx.render(TRACE)
            for i, label in enumerate(("colour", "nd", "albedo")):
                assert_bits_equal(ctx.read(i), ref[i], f"{label}, sun_size {sun_size}, cull {cull}")
            assert ctx.stats().rays == ref[3]


@pytest.mark.parametrize("w,h,nranks,rank,batch,tracer", [(150, 70, 1, 0, 8, 0), (150, 70, 3, 1, 16, 0), (97, 41, 1, 0, 24, 1), (64, 48, 2, 1, 32, 4),
                                                         (150, 70, 1, 0, 4, 0), (97, 41, 2, 0, 12, 0), (64, 48, 1, 0, 20, 1)])
def test_frame_lanes_change_nothing(O, H, scenes, noise, w, h, nranks, rank, batch, tracer):
    """VXRT_OPT_FRAME_LANES: a launch of 8 / 16 / 24 / 32 frames of one camera gives each wave of trace_kernel a row of 8 pixels in 8
    frames (of 4 / 12 / 20 / 28 frames: two rows in 4 frames) instead of an 8 x 8 tile of one frame (csrc/trace.hip).  Frame sizes that are no multiples of 8, row bands, the all-in-one
    kernel and the head + tail pair: every image of the launch's last frame equals the oracle's, the ray total of ALL its frames
    equals the oracle's sum, and a launch that is no multiple of 4 frames (or whose camera sweeps across the scene) falls back to one
    frame per wave; a camera path that barely moves keeps the frame lanes (each lane reads its frame's camera)."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    bounces = 3
    got = {}
    for lanes in (1, 0):
        with Context(w, h, max_bounces=bounces, noise=noise, tracer=tracer, frames_per_launch=batch, frames_in_flight=2, rank=rank, nranks=nranks,
                     band_rows=8) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            ctx.set_option(H.OPT_FRAME_LANES, lanes)
            ctx.render_frames(TRACE, batch)
            st = ctx.stats()
            assert st.frame_lane_launches == (1 if lanes else 0)
            got[lanes] = [ctx.read(i) for i in range(3)] + [st.rays]
            rows = ctx.local_rows()
            if lanes:
                ctx.render_frames(TRACE, batch + 3)           # a second launch of `batch` frames, then 3 frames: no frame lanes for those
                assert ctx.stats().frame_lane_launches == 2
                pose = np.stack([np.asarray(cam[0], np.float32)] * batch), np.stack([np.asarray(cam[1], np.float32)] * batch)
                ctx.render_path(TRACE, pose[0], pose[1], cam[2])   # a camera path that barely moves: frame lanes, a camera per lane
                assert ctx.stats().frame_lane_launches == 3
                side = np.cross(np.asarray(cam[1], np.float32), np.array([0.0, 1.0, 0.0], np.float32)).astype(np.float32)
                swing = np.stack([pose[1][k] if k % 2 == 0 else side for k in range(batch)])   # every other frame looks 90 degrees away
                ctx.render_path(TRACE, pose[0], swing, cam[2])     # ... that sweeps across the scene: one frame per wave
                assert ctx.stats().frame_lane_launches == 3
    u.frame_number = batch
    ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
    for i, label in enumerate(("colour", "nd", "albedo")):
        assert_bits_equal(got[1][i], ref[i][rows], f"{label} with frame lanes")
        assert_bits_equal(got[0][i], got[1][i], f"{label}: one frame per wave against frame lanes")
    assert got[0][3] == got[1][3]
    if nranks == 1:
        assert got[1][3] == sum(O.trace(octree, noise, (setattr(u, "frame_number", f) or u), w, h, bounces, crop=(0, 0, w, h))[3]
                                for f in range(1, batch + 1))


def _tile_key(c):
    """csrc/trace.hip: tile_key — the sort's bins: 4 * log2 of the cost with two mantissa bits."""
    c = int(c)
    if c < 4:
        return c
    e = c.bit_length() - 1
    return ((e << 2) | ((c >> (e - 2)) & 3)) - 4


@pytest.mark.parametrize("w,h,batch,inflight,frames", [(1920, 1080, 16, 2, 96), (256, 144, 1, 1, 3), (640, 360, 8, 3, 72)])
def test_tile_order_is_a_permutation_that_spreads_the_walking_tiles(H, scenes, noise, w, h, batch, inflight, frames):
    """The launch order of trace_kernel's tiles (csrc/trace.hip: tile_hist / tile_scan / tile_scatter_kernel, read back through
    vxrt_debug_tile_order): every tile exactly once; the tiles that walked the octree in descending order of their cost bins; each
    followed by k tiles of sky, k even, as far as the sort decided to spread them (a long launch against its chains) — not at all for
    a small frame one launch at a time, whose order is plain longest-first."""
    import ctypes as C
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("menger")
    with Context(w, h, max_bounces=4, noise=noise, tracer=4, frames_per_launch=batch, frames_in_flight=inflight) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*scenes.bench_camera(size))
        ctx.render_frames(TRACE, frames)
        n = ((w + 7) // 8) * ((h + 7) // 8)
        order, cost = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        walking, spread = C.c_uint32(0), C.c_uint32(0)
        H._check(H.lib().vxrt_debug_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                               C.byref(walking), C.byref(spread)), "vxrt_debug_tile_order")
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))            # a permutation
    heavy = cost[order] >= 4                                                        # in launch order: did the tile walk?
    nh, nl = int(heavy.sum()), int(n - heavy.sum())
    assert walking.value == nh and 0 < nh < n and ((cost == 1) | (cost >= 4)).all()  # a tile of sky records 1 (no tile is left out: cost 0)
    keys = np.array([_tile_key(c) for c in cost[order][heavy]])
    assert (np.diff(keys) <= 0).all()                                               # the walking tiles: longest first
    k = (nl * spread.value // 256 // nh) & ~1
    at = np.flatnonzero(heavy)
    if k == 0:
        assert np.array_equal(at, np.arange(nh))
    else:
        assert np.array_equal(at, np.arange(nh) * (k + 1))
    big = w * h * batch >= 1920 * 1080 * 16
    assert (spread.value > 0) == big and (k > 0) == big, (spread.value, k)
