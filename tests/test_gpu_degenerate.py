"""Degenerate random numbers: noise tables full of exact zeros and other special values drive the shading through its corner
cases at every pixel (plane_radius = 0, phi = 0, rand_dir = 0 or parallel to the sun, bounce directions with zero components and
both signs of zero, rays that start on voxel faces).  Whatever voxels.comp's text does with them — NaN colours included — the GPU
and the oracle must do the same.  (With the seeded table exactly one of its 8 M entries is 0; tests/test_gpu_stress.py at scale 12
walked into it.)"""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def tables(base):
    rng = np.random.default_rng(1)
    return {
        "zeros30": np.where(rng.random(base.size) < 0.3, np.float32(0), base).astype(np.float32),
        "specials": rng.choice(np.array([0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0 ** -24], np.float32), base.size).astype(np.float32),
        "all_zero": np.zeros_like(base),
        "halves": np.full_like(base, 0.5),
    }


@pytest.mark.parametrize("label", ["zeros30", "specials", "all_zero", "halves"])
@pytest.mark.parametrize("name,tracer", [("menger", "0"), ("menger", "1"), ("castle", "0")])
def test_degenerate_noise_tables(O, H, scenes, noise, monkeypatch, label, name, tracer):
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    monkeypatch.setenv("VXRT_TRACE_VARIANT", tracer)
    table = tables(noise)[label]
    w, h, bounces = 256, 144, 4
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
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
            for i, what in enumerate(("colour", "normal/depth", "albedo/node")):
                assert_bits_equal(got[i], ref[i], f"{what} {name} {label} frame {frame}")
            assert rays == ref[3]


@pytest.mark.parametrize("label", ["zeros30", "specials"])
@pytest.mark.parametrize("specularity,sun,emit", [(0.3, None, None), (1.0, None, 2.0), (0.0, 0.0, 3.0), (0.26, 0.0, None)])
@pytest.mark.parametrize("name", ["menger", "room"])
def test_degenerate_noise_other_shading_branches(O, H, scenes, noise, label, specularity, sun, emit, name):
    """The same tables through the specular branch (reflect + normalize of degenerate vectors; 0.25 < 0.26 but not < 0.25: the
    branch compare sits on a table value), with the sun off, and with emitters."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    table = tables(noise)[label]
    w, h, bounces = 192, 112, 5
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.specularity = specularity
    if sun is not None:
        u.sun_strength = sun
    if emit is not None:
        u.emit_strength = emit
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    with Context(w, h, max_bounces=bounces, noise=table) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.uniforms.specularity = specularity
        if sun is not None:
            ctx.uniforms.sun_strength = sun
        if emit is not None:
            ctx.uniforms.emit_strength = emit
        for frame in (1, 2):
            ctx.set_frame_number(frame - 1)
            ctx.reset_stats()
            ctx.render(TRACE)
            got = [ctx.read(i) for i in range(3)]
            rays = ctx.stats().rays
            u.frame_number = frame
            ref = O.trace(octree, table, u, w, h, bounces, crop=(0, 0, w, h))
            for i, what in enumerate(("colour", "normal/depth", "albedo/node")):
                assert_bits_equal(got[i], ref[i], f"{what} {name} {label} spec {specularity} sun {sun} emit {emit} frame {frame}")
            assert rays == ref[3]


def test_path_log_matches_the_oracle(O, H, scenes, noise):
    """vxrt_debug_path_log: the casts of single pixels, bit for bit (signs of zero included) those of the oracle's path."""
    from gpu_voxel_raytracer_amd import Camera, Context
    table = tables(noise)["specials"]
    w, h, bounces = 256, 144, 4
    pos, mrgb, size = scenes.load_scene("menger")
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 1
    with Context(w, h, max_bounces=bounces, noise=table) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        for (x, y) in [(221, 10), (196, 13), (224, 7), (183, 8), (219, 10), (0, 0), (128, 72), (255, 143)]:
            g, o = ctx.path_log(x, y), O.path_log(octree, table, u, bounces, x, y)
            assert g.shape == o.shape, (x, y, g.shape, o.shape)
            same = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
            # the `time` a cast that MISSES leaves behind is no output of cast_bounded_ray (voxels.comp:134-247 returns false and its
            # callers read nothing of the ray then): the kernels' walk may end such a ray a few trips early (trace_common.h: VXRT_ROOT_EXIT)
            same[:, 7] |= g[:, 6] == 0
            assert same.all(), (x, y, g[~same.all(1)][:1], o[~same.all(1)][:1])


@pytest.mark.parametrize("label,radius,moving", [("specials", 2, False), ("zeros30", 1, True), ("all_zero", 0, False), ("specials", 8, True)])
def test_degenerate_noise_through_temporal_and_denoise(O, H, scenes, noise, label, radius, moving):
    """The traced frames of such tables hold NaN and inf colours (0/0 in normalize, 0 * inf weights); temporal.comp's mix / clamp and
    denoise.comp's exp / log / division must carry them exactly as the oracle does, frame after frame (NaN == NaN here)."""
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    from test_gpu_pipeline import OraclePipeline
    table = tables(noise)[label]
    w, h, bounces = 160, 96, 4
    ref = OraclePipeline(O, scenes, table, "menger", w, h, bounces, radius)
    cam = list(scenes.close_camera(ref.size))
    pos, mrgb, _ = scenes.load_scene("menger")
    with Context(w, h, max_bounces=bounces, noise=table) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = radius
        for f in range(4):
            if moving:
                cam[0] = (np.asarray(cam[0], np.float32) + np.float32(0.37) * np.array([1, 0.25, -0.5], np.float32)).astype(np.float32)
            ctx.camera = Camera(*cam)
            ctx.render(ALL)
            want = ref.render(cam)
            for img, wimg, what in zip(range(5), want, ("colour", "nd", "albedo", "accum", "denoised")):
                assert_bits_equal(ctx.read(img), wimg, f"{what} frame {f + 1} {label} r={radius} moving={moving}")
