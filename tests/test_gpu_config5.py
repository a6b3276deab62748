"""GPU parity for BASELINE config 5 AT ITS SIZE: the 2048^3 procedural Menger grid (level 7 clipped to [0, 2048)^3, 261 M octree
nodes, 5.6 GB of SVO records + leaf words built by vxrt_set_menger), 7680x4320, 16 spp, 8 bounces, one of 8 ranks' band sets
(what one GPU of the 8 x MI355X configuration renders).  The scene stresses the format limits of shaders/voxels.comp:3
(MAX_DEPTH 16: this tree has 12 node levels) and :175 (`8*node + octant` as int: 261 M x 8 = 2.09e9 < 2^31).

The oracle reaches this size by materialising the reference's octree buffer from the voxel predicate as the walk touches it
(oracle/oprocedural.py: pinned against stored octrees in tests/test_oracle_procedural.py), so the comparison is bit for bit
with the restated shader, not a sub-volume stand-in; an independent binary64 DDA over the predicate checks the walk's
result contract (first voxel, face, time) for > 10^5 primary hits from outside the sponge and from inside one of its tunnels.
Parity status: unpinned by the reference (it has no procedural scene; the config is BASELINE.json's)."""
import numpy as np
import pytest

from conftest import assert_bits_equal, require_variants
from test_oracle_procedural import FULL, compare_with_dda, full_size_cameras, primary_rays

pytestmark = pytest.mark.gpu

W, HH, BOUNCES, SPP, NRANKS, RANK, BAND = 7680, 4320, 8, 16, 8, 5, 16


def make_context(noise, tracer, rank=RANK, wide=False):
    from gpu_voxel_raytracer_amd import Context
    from gpu_voxel_raytracer_amd.host import OPT_SCENE_FORMAT
    ctx = Context(W, HH, max_bounces=BOUNCES, noise=noise, rank=rank, nranks=NRANKS, band_rows=BAND, frames_per_launch=SPP,
                  frames_in_flight=1, tracer=tracer)
    if wide:
        ctx.set_option(OPT_SCENE_FORMAT, 1)     # also build and walk the two-levels-per-record form of the scene
    ctx.set_menger(*FULL)
    return ctx


def oracle_uniforms(O, cam):
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], W, HH))
    return u


def test_config5_one_rank_of_the_8k_frame_vs_oracle(O, H, noise):
    """Frame 1 and the 16-sample displayed frame of rank 5's bands against the oracle on strips; whole-band-set properties;
    the compacted-tail tracer and the all-in-one kernel against each other on every pixel of the band set."""
    from gpu_voxel_raytracer_amd import ALBEDO_NODE, NORMAL_DEPTH, SAMPLED_COLOR, TEMPORAL, TRACE, ACCUM_COLOR, Camera
    level, clip, mrgb, period = FULL
    cam = full_size_cameras()["outside"]
    u = oracle_uniforms(O, cam)
    results = {}
    # 0: what the library picks for a scene of this size; 1: all-in-one kernel; 4: head + compacted tail; "wide": the wide scene records
    for tracer in (0, 1, 4, "wide"):
        require_variants(H, wide=tracer)       # the wide records live in the -DVXRT_VARIANTS=1 build, loaded beside the product
        with make_context(noise, 4 if tracer == "wide" else tracer, wide=tracer == "wide") as ctx:
            st = ctx.stats()
            assert (st.scene_format, st.wide_nodes > 30_000_000) == ((1, True) if tracer == "wide" else (0, False))
            assert st.octree_depth == 11 == O.menger_depth(level, clip) and st.octree_nodes > 250_000_000
            assert st.octree_nodes * 8 < 2 ** 31                       # voxels.comp:175 would still index it with an int
            assert st.scene_bytes > 5 * 2 ** 30                        # HBM-resident, far beyond the 256 MB Infinity Cache
            rows = ctx.local_rows()
            assert len(rows) == st.local_rows and abs(len(rows) - HH // NRANKS) <= BAND
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)                                          # frame_number 1, one sample
            one = [ctx.read(i) for i in (SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE)]
            rays_one = ctx.stats().rays
            ctx.set_frame_number(0)
            ctx.reset_history()
            ctx.reset_stats()
            ctx.render_spp(TRACE | TEMPORAL, SPP)                      # the config's displayed frame: 16 samples, then temporal
            results[tracer] = (one, rays_one, [ctx.read(i) for i in (SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE, ACCUM_COLOR)], ctx.stats().rays, rows)

    one, rays_one, shown, rays_shown, rows = results[0]
    for tracer in [t for t in (1, 4, "wide") if t in results]:
        for a, b, label in zip(results[0][0] + results[0][2], results[tracer][0] + results[tracer][2],
                               ("colour", "nd", "albedo", "mean colour", "nd16", "albedo16", "accum")):
            assert_bits_equal(a, b, f"config 5: tracer {tracer} vs default, {label}")
        assert results[tracer][1] == rays_one and results[tracer][3] == rays_shown

    # (a) frame 1 against the oracle: three full-width 8-row strips of this rank's bands that see geometry
    hit_rows = np.nonzero((one[1][..., 3] >= 0).mean(1) > 0.2)[0]
    assert len(hit_rows) > 100
    u.frame_number = 1
    sq, n = 0.0, 0
    for lr in (hit_rows[len(hit_rows) // 8], hit_rows[len(hit_rows) // 2], hit_rows[-len(hit_rows) // 8]):
        lr = int(lr) // 8 * 8
        y0 = int(rows[lr])
        assert np.array_equal(rows[lr:lr + 8], np.arange(y0, y0 + 8))
        c, d, a, _ = O.trace_menger(level, clip, mrgb, period, noise, u, BOUNCES, (0, y0, W, y0 + 8))
        assert_bits_equal(one[0][lr:lr + 8], c, f"config 5 colour, rows {y0}..")
        assert_bits_equal(one[1][lr:lr + 8], d, f"config 5 normal/depth, rows {y0}..")
        assert_bits_equal(one[2][lr:lr + 8], a, f"config 5 albedo/node, rows {y0}..")
        sq += float(((one[0][lr:lr + 8, :, :3] - c[..., :3]) ** 2).sum())
        n += c[..., :3].size
    assert (sq / n) ** 0.5 <= 1e-3                                     # BASELINE.json's stated tolerance (bit-exact => 0)

    # (b) the 16-sample displayed frame: samples summed in frame order in binary32, divided once (vxrt_render_spp)
    lr = int(hit_rows[len(hit_rows) // 2]) // 8 * 8
    y0, x0, x1 = int(rows[lr]), 2560, 5120
    total = None
    for f in range(1, SPP + 1):
        u.frame_number = f
        c, d, a, _ = O.trace_menger(level, clip, mrgb, period, noise, u, BOUNCES, (x0, y0, x1, y0 + 8))
        total = c.copy() if total is None else (total + c).astype(np.float32)
    mean = (total / np.float32(SPP)).astype(np.float32)
    assert_bits_equal(shown[0][lr:lr + 8, x0:x1], mean, "config 5, 16 spp mean colour")
    assert_bits_equal(shown[1][lr:lr + 8, x0:x1], d, "config 5, 16 spp normal/depth")
    assert_bits_equal(shown[3][lr:lr + 8, x0:x1, :3], mean[..., :3], "config 5, temporal of a first frame")

    # (c) whole band set: finite radiance, hit <=> depth >= 0 <=> leaf bit, normals are face flags, ray accounting,
    #     emissive seeds present, leaf words = the scene's two words
    for color, nd, alb in (one, shown[:3]):
        assert np.isfinite(color).all() and (color[..., 3] == 1).all() and (color[..., :3] >= 0).all()
        node = alb[..., 3].view(np.int32)
        hit = nd[..., 3] >= 0
        assert ((node < 0) == hit).all() and (node[~hit] == 0xffffff).all()
        assert (nd[~hit][:, :3] == 2.0 ** 30).all() and (nd[~hit][:, 3] == -1).all()
        nrm = nd[hit][:, :3]
        assert np.isin(nrm, (-1.0, 0.0, 1.0)).all() and (np.abs(nrm).sum(1) >= 1).all()
        base = np.int32(-2 ** 31 | (mrgb[0] & 0x7f) << 24 | mrgb[1] << 16 | mrgb[2] << 8 | mrgb[3])
        assert np.isin(node[hit], (base, base | np.int32(0x40 << 24))).all()
        assert (node[hit] == (base | np.int32(0x40 << 24))).sum() > 0
    px = len(rows) * W
    assert px <= rays_one <= 2 * BOUNCES * px and SPP * px <= rays_shown <= SPP * 2 * BOUNCES * px


def test_config5_cast_rays_vs_oracle_walk_and_predicate_dda(O, H, noise):
    """>= 10^5 primary hits of the full scene through vxrt_debug_cast_rays: bit-equal to the oracle's walk, and the same voxel
    (leaf word incl. the emissive bit), face and time as the independent predicate DDA — from outside and from inside a tunnel."""
    level, clip, mrgb, period = FULL
    rng = np.random.default_rng(2026)
    n, hits = 150_000, 0
    with make_context(noise, 0, rank=0) as ctx:
        for name, cam in full_size_cameras().items():
            o, d = primary_rays(O, cam, W, HH, n, rng)
            hit, t, node, normal = ctx.cast_rays(o, d)
            ohit, ot, onode, onormal, iters = O.cast_rays_menger(level, clip, mrgb, period, o, d)
            assert np.array_equal(hit, ohit) and np.array_equal(node[hit], onode[hit])
            assert_bits_equal(t[hit], ot[hit], f"config 5 {name}: hit times")
            assert_bits_equal(normal[hit], onormal[hit], f"config 5 {name}: normals")
            assert iters.max() < 2048
            ok, hard, compared, rel = compare_with_dda(o, d, hit, t, node, normal, O.dda_menger(level, clip, o, d),
                                                       lambda cells: O.menger_cells(level, clip, mrgb, period, cells)[1])
            assert ok > 0.9995 and hard < 2e-4, (name, ok, hard)
            assert np.quantile(rel, 0.999) < 1e-4
            hits += compared
        # secondary-like rays: origins just off voxel faces inside the sponge, random directions
        o, d = primary_rays(O, full_size_cameras()["tunnel"], W, HH, 40_000, rng)
        hit, t, node, normal = ctx.cast_rays(o, d)
        o2 = (o + d * t[:, None] + np.float32(1e-5) * normal)[hit].astype(np.float32)
        d2 = rng.normal(size=o2.shape).astype(np.float32)
        d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
        got = ctx.cast_rays(o2, d2)
        want = O.cast_rays_menger(level, clip, mrgb, period, o2, d2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2][got[0]], want[2][got[0]])
        assert_bits_equal(got[1][got[0]], want[1][got[0]], "config 5 secondary rays: hit times")
        assert_bits_equal(got[3][got[0]], want[3][got[0]], "config 5 secondary rays: normals")
    assert hits >= 100_000, hits
