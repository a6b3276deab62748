"""GPU parity on the wider scene inputs of SURVEY.md §8f n3: the reference's start-up scene (Context::create_voxels,
src/context.rs:838-910: 554 614 voxels, negative coordinates, depth 9, 1 % emissive) and a multi-model .vox scene placed
through its scene graph — traced by the HIP kernel and by the oracle from the same voxel list.  Bar: bit-exact."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from test_scene_extensions import ngrp, nshp, ntrn, scene_file

pytestmark = pytest.mark.gpu


def trace_both(O, noise, pos, mrgb, cam, w, h, bounces, frame=1):
    from gpu_voxel_raytracer_amd import ALBEDO_NODE, NORMAL_DEPTH, SAMPLED_COLOR, TRACE, Camera, Context
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = frame
    ref = O.trace(O.create_octree(pos, mrgb), noise, u, w, h, bounces, crop=(0, 0, w, h))
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.set_frame_number(frame - 1)
        ctx.render(TRACE)
        got = [ctx.read(i) for i in (SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE)]
        st = ctx.stats()
    for g, r, what in zip(got, ref[:3], ("colour", "normal/depth", "albedo/node")):
        assert_bits_equal(g, r, what)
    assert st.rays == ref[3]
    return got, st


def test_start_up_scene_with_the_reference_camera(O, H, scenes, noise):
    pos, mrgb = H.default_scene_voxels(1)
    # the reference's start camera (0, 0, -2) looking +z (src/context.rs:618-622) sits inside the bowl, below the rim
    got, st = trace_both(O, noise, pos, mrgb, scenes.reference_start_camera(), 192, 112, 3)
    assert st.octree_depth == 9
    hit = got[1][..., 3] >= 0
    assert 0.3 < hit.mean() <= 1.0
    # and from above, looking down into the bowl
    cam = (np.array([0.0, 60.0, -150.0], np.float32), np.array([0.0, -0.5, 1.0], np.float32), scenes.FOV_70)
    trace_both(O, noise, pos, mrgb, cam, 160, 96, 4, frame=7)


def test_multi_model_scene_through_the_scene_graph(O, H, scenes, noise):
    rng = np.random.default_rng(11)

    def blob(size, n, colours):
        cells = {(int(rng.integers(size[0])), int(rng.integers(size[1])), int(rng.integers(size[2]))) for _ in range(n)}
        return size, [(x, y, z, colours[(x + y + z) % len(colours)]) for x, y, z in sorted(cells)]
    models = [blob((20, 20, 20), 3000, (1, 3)), blob((12, 30, 8), 1500, (2, 3)), blob((40, 40, 2), 2500, (1,))]
    graph = (ntrn(0, 1) + ngrp(1, [2, 4, 6, 8]) + ntrn(2, 3, t=(0, 0, 10)) + nshp(3, [0])
             + ntrn(4, 5, t=(25, 5, 15), r=4 | (1 << 4)) + nshp(5, [1]) + ntrn(6, 7, t=(0, 0, -1)) + nshp(7, [2])
             + ntrn(8, 9, t=(-22, -8, 9), r=9 | (1 << 5)) + nshp(9, [0]))
    pos, mrgb, (lo, hi) = H.vox_scene_to_voxels(scene_file(models, graph), H.VOX_ALL_MODELS)
    assert len(pos) == 2 * len(models[0][1]) + len(models[1][1]) + len(models[2][1]) and (mrgb[:, 0] == 0x40).any() and pos.min() < 0
    ext = (np.array(hi) - np.array(lo) + 1).astype(np.float32) * np.float32(0.5)
    centre = (np.array(lo) + np.array(hi) + 1).astype(np.float32) * np.float32(0.25)
    position = (centre + ext.max() * np.array([-0.7, 0.5, -0.9], np.float32)).astype(np.float32)
    cam = (position, (centre - position).astype(np.float32), scenes.FOV_70)
    got, _ = trace_both(O, noise, pos, mrgb, cam, 192, 112, 4)
    assert (got[1][..., 3] >= 0).mean() > 0.05
