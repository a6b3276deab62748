"""GPU: the procedural scene path of BASELINE config 5 (vxrt_set_menger builds SVO records directly) against
the generic path (voxel list -> reference-layout octree -> SVO) and the oracle, on sizes both can handle.
SURVEY.md §8d: "the oracle checks this config on a 256^3 sub-volume only"."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level,clip,period", [(2, 0, 0), (3, 20, 5), (4, 64, 37), (5, 200, 301), (6, 256, 4096), (3, 1, 0), (1, 2, 2)])
def test_procedural_menger_equals_voxel_list_scene(O, H, noise, level, clip, period):
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    mrgb = (0, 40, 200, 90)
    pos, m = H.menger_voxels(level, mrgb, clip=clip, emissive_period=period)
    side = min(3 ** level, clip or 3 ** level)
    assert len(pos) > 0 and pos.max() == side - 1
    if period:
        assert (m[:, 0] == 0x40).any() or len(pos) < period
    ext = np.float32(side / 2)
    cam = (np.array([-0.4, 0.8, -0.7], np.float32) * ext + ext / 2, np.array([0.9, -0.55, 1.2], np.float32), 1.1)
    w, h, bounces = 160, 112, 4
    images = []
    for procedural in (True, False):
        with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
            if procedural:
                ctx.set_menger(level, clip, mrgb, period)
            else:
                ctx.recreate_octree(pos, m)
            st = ctx.stats()
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)
            images.append([ctx.read(i) for i in range(3)] + [ctx.stats().rays, st.octree_depth])
    for a, b, label in zip(images[0][:3], images[1][:3], ("colour", "nd", "albedo")):
        assert_bits_equal(a, b, f"{label} procedural vs voxel list")
    assert images[0][3] == images[1][3] and images[0][4] == images[1][4] == O.voxel_depth(pos)
    assert (images[0][1][..., 3] >= 0).any()
    if len(pos) <= 300000:   # and the oracle on the same voxel list
        octree = O.create_octree(pos, m)
        u = O.Uniforms.default()
        u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
        u.frame_number = 1
        ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
        assert_bits_equal(images[0][0], ref[0], "colour vs oracle")
        assert_bits_equal(images[0][2], ref[2], "albedo vs oracle")
        assert images[0][3] == ref[3]


def test_procedural_menger_node_counts(H):
    """The SVO built procedurally has the node count of the octree built from the voxel list (menger.vox: 44 877)."""
    from gpu_voxel_raytracer_amd import Context
    with Context(32, 32) as ctx:
        ctx.set_menger(4)
        st = ctx.stats()
        assert st.octree_nodes == 44877 and st.octree_depth == 7
        assert st.scene_bytes == 44877 * 8 + 160000 * 4
