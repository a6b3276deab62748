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


@pytest.mark.parametrize("level,clip,period", [(2, 0, 0), (3, 20, 5), (4, 64, 37), (5, 200, 301), (6, 256, 4096), (7, 512, 8192), (2, 4, 3)])
def test_device_built_scene_is_byte_identical_to_the_host_builders(H, monkeypatch, level, clip, period):
    """vxrt_set_menger builds the scene on the device (csrc/scene_device.hip: dense bottom-up masks, then a top-down enumeration with
    scans); the records and leaf words equal the host builder's (csrc/scene_procedural.cpp, VXRT_HOST_BUILD=1) byte for byte."""
    from gpu_voxel_raytracer_amd import Context
    mrgb = (0, 150, 170, 120)
    with Context(32, 32) as ctx:
        ctx.set_menger(level, clip, mrgb, period)
        dev = ctx.read_scene()
        dev_stats = ctx.stats()
    monkeypatch.setenv("VXRT_HOST_BUILD", "1")
    with Context(32, 32) as ctx:
        ctx.set_menger(level, clip, mrgb, period)
        host = ctx.read_scene()
        host_stats = ctx.stats()
    assert dev[0].shape == host[0].shape and dev[1].shape == host[1].shape
    assert np.array_equal(dev[0], host[0]) and np.array_equal(dev[1], host[1])
    assert (dev_stats.octree_depth, dev_stats.octree_nodes, dev_stats.scene_bytes) == (host_stats.octree_depth, host_stats.octree_nodes, host_stats.scene_bytes)
    assert len(dev[0]) > 1


def test_full_size_scene_builds_on_the_device_in_a_fraction_of_a_second(H):
    """BASELINE config 5's scene (level 7 clipped to 2048^3: 261 M nodes, 1.05e9 leaf words, 5.6 GB): <= 0.5 s on the device
    (the host builder needs ~9 s of threads plus the upload); the frames it renders are checked in test_gpu_config5.py."""
    import time
    from gpu_voxel_raytracer_amd import Context
    with Context(64, 64) as ctx:
        ctx.set_menger(3, 20, (0, 1, 2, 3), 0)            # first use: module load, allocator warm-up
        t0 = time.perf_counter()
        ctx.set_menger(7, 2048, (0, 150, 170, 120), 8192)
        dt = time.perf_counter() - t0
        st = ctx.stats()
        print(f"vxrt_set_menger(7, 2048) on the device: {dt:.3f} s, {st.octree_nodes} nodes, {st.scene_bytes / 2**30:.2f} GiB")
        assert st.octree_nodes == 261140230 and st.octree_depth == 11
        assert dt <= 0.5
        svo, leaves = ctx.read_scene()
        # structural checks on the whole 5.6 GB: the root, breadth-first child bases, leaf parents at the end, two leaf words
        assert svo[0].tolist() == [0x80, 1] and svo[1, 1] == 2
        child = svo[:, 0] & 0xff
        inner = np.nonzero(child)[0]
        counts = np.zeros(len(inner), np.int64)
        for b in range(8):
            counts += (child[inner] >> b) & 1
        assert np.array_equal(svo[inner[1:], 1].astype(np.int64), svo[inner[0], 1] + np.cumsum(counts)[:-1])   # bases = running child count
        lp = np.nonzero(svo[:, 0] >> 8)[0]
        assert lp[0] == inner[-1] + 1 and lp[-1] == len(svo) - 1
        assert set(np.unique(leaves).tolist()) == {-2 ** 31 | 150 << 16 | 170 << 8 | 120, -2 ** 31 | 0x40 << 24 | 150 << 16 | 170 << 8 | 120}
