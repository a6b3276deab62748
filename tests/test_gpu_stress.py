"""GPU parity at full size against the ORACLE: whole 1920x1080 frames from random cameras (outside, inside the model's box,
axis-aligned views whose primary rays have zero direction components), default tracer, several frames per launch.  ~40 M rays per
run — two orders of magnitude more than the small-frame tests, enough to meet rays that take the shader-text walk (a direction
component exactly 0) and near-tie traversals.  Bar: bit-exact for all three outputs, exact ray counts."""
import os
import zlib

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,bounces,cams", [("menger", 4, 8), ("room", 3, 4)])
def test_full_frames_from_random_cameras(O, H, scenes, noise, name, bounces, cams):
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    cams *= int(os.environ.get("VXRT_STRESS_SCALE", "1"))      # a longer one-off run: VXRT_STRESS_SCALE=10
    w, h, frames = 1920, 1080, 3
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    ext = scenes.world_extent(size)
    centre = ext * np.float32(0.5)
    rng = np.random.default_rng(zlib.crc32(name.encode()) + int(os.environ.get("VXRT_STRESS_SEED", "0")))   # (str hashes are salted per process)
    with Context(w, h, max_bounces=bounces, noise=noise, frames_per_launch=frames, frames_in_flight=2) as ctx:
        ctx.recreate_octree(pos, mrgb)
        for i in range(cams):
            p = (centre + ext.max() * rng.uniform(-1.1, 1.1, 3)).astype(np.float32)
            if i % 4 == 1:
                p = (centre + ext * rng.uniform(-0.45, 0.45, 3)).astype(np.float32)
            d = (centre + ext * rng.uniform(-0.3, 0.3, 3) - p).astype(np.float32)
            if i % 4 == 2:
                d = np.array([[1, 0, 0], [0, 0, 1]][(i // 4) % 2], np.float32)
            spec = 0.3 if i % 3 == 0 else 0.0
            ctx.camera = Camera(p, d, scenes.FOV_70)
            ctx.uniforms.specularity = spec
            first = ctx.stats().frames
            ctx.reset_stats()
            ctx.render_frames(TRACE, frames)
            got = [ctx.read(img) for img in range(3)]
            rays = ctx.stats().rays
            u = O.Uniforms.default()
            u.specularity = spec
            u.set_camera(p, O.camera_axis_scaled(p, d, scenes.FOV_70, w, h))
            want_rays = 0
            for f in range(frames):
                u.frame_number = i * frames + f + 1
                ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
                want_rays += ref[3]
            where = f"camera {i} of {name}: position {p.tolist()} direction {d.tolist()} specularity {spec}"
            for img, label in zip(range(3), ("colour", "normal/depth", "albedo/node")):
                assert_bits_equal(got[img], ref[img], f"{label}, {where}")
            assert rays == want_rays, (where, rays, want_rays)
            del first
