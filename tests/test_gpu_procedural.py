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


def _levels(svo, depth):
    """Per node level 1 .. depth: (record indices in breadth-first order, masks) by following the bases from the root — whatever the order in memory."""
    masks, base = svo[:, 0], svo[:, 1]
    idx = np.array([0], np.int64)
    out = []
    for level in range(depth + 1):
        m = masks[idx] & 0xff
        out.append((idx, masks[idx].copy()))
        if level == depth:
            break
        cnt = np.array([bin(int(v)).count("1") for v in range(256)], np.int64)[m]
        first = np.repeat(base[idx].astype(np.int64), cnt)
        within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        idx = first + within
    return out


@pytest.mark.parametrize("order", [2, 3])
def test_treelet_node_order_is_the_same_tree_and_the_same_image(H, O, noise, order):
    """VXRT_OPT_NODE_ORDER 2 / 3 (csrc/scene_device.hip: reorder_bottom_treelets): the records of the last two / three node levels as
    depth-first treelets.  Same tree — followed from the root, every level holds the same masks in the same order and the leaf parents
    the same leaf-word bases — all descendants of a node of level depth - order lie in one run of memory right where its base points,
    the device-built and the host-built scene are still byte-identical, and the frames (colour, normal / depth, leaf words, ray count)
    are bit-equal to the breadth-first order's.  A voxel-list scene of depth 10 goes the same way (vxrt_set_voxels -> the host builder
    -> the same device pass)."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    from gpu_voxel_raytracer_amd.host import OPT_HOST_SCENE_BUILD, OPT_NODE_ORDER
    level, clip, period, mrgb = 6, 0, 997, (0, 150, 170, 120)     # 729^3: depth 10
    f32 = np.float32
    cam = (np.array([-300, 500, -420], f32), np.array([0.9, -0.6, 1.2], f32), 1.0)
    got = {}
    for o, host_build in ((0, 0), (order, 0), (order, 1)):
        with Context(320, 192, max_bounces=4, noise=noise, tuning=[(OPT_NODE_ORDER, o), (OPT_HOST_SCENE_BUILD, host_build)]) as ctx:
            ctx.set_menger(level, clip, mrgb, period)
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)
            st = ctx.stats()
            assert st.node_order == o and st.octree_depth == 10
            got[(o, host_build)] = (ctx.read_scene(), [ctx.read(i) for i in range(3)], st.rays)
    (svo0, lw0), img0, rays0 = got[(0, 0)]
    (svo1, lw1), img1, rays1 = got[(order, 0)]
    (svo1h, lw1h), _, _ = got[(order, 1)]
    assert np.array_equal(svo1, svo1h) and np.array_equal(lw1, lw1h)                    # device builder == host builder, reordered alike
    assert np.array_equal(lw0, lw1) and svo0.shape == svo1.shape and not np.array_equal(svo0, svo1)
    a, b = _levels(svo0, 10), _levels(svo1, 10)
    for level_no, ((ia, ma), (ib, mb)) in enumerate(zip(a, b)):
        assert np.array_equal(ma, mb), level_no
    assert np.array_equal(svo0[a[10][0], 1], svo1[b[10][0], 1])                          # leaf parents: the same first leaf word
    top = 10 - order                                                                     # the treelets' parents: they and everything above stay put
    assert np.array_equal(a[top][0], b[top][0])
    # every record below `top` belongs to exactly one treelet, and a parent's treelet is the run [base, next parent's base)
    parents = b[top][0]
    starts = svo1[parents, 1].astype(np.int64)
    assert (np.diff(starts) > 0).all() and starts[0] == a[top + 1][0][0]
    sizes = np.diff(np.append(starts, len(svo1)))
    assert sizes.max() <= (8 + 64 + 512 if order == 3 else 8 + 64)
    below = np.sort(np.concatenate([b[l][0] for l in range(top + 1, 11)]))
    assert np.array_equal(below, np.arange(starts[0], len(svo1)))
    for l in range(top + 1, 11):                                                         # each level's nodes fall into their own ancestor's run
        owner = np.searchsorted(starts, b[l][0], side="right") - 1
        assert (np.diff(owner) >= 0).all()
    for x, y, name in zip(img0, img1, ("colour", "normal / depth", "albedo / leaf word")):
        assert_bits_equal(x, y, name)
    assert rays0 == rays1 and (img0[1][..., 3] >= 0).mean() > 0.1
    # a voxel list of depth 10 through vxrt_set_voxels
    rng = np.random.default_rng(3)
    pos = np.concatenate([rng.integers(0, 1000, (20000, 3)), rng.integers(400, 440, (30000, 3))]).astype(np.int16)
    m = rng.integers(0, 256, (len(pos), 4)).astype(np.uint8)
    cam = (np.array([150, 260, 120], f32), np.array([0.4, -0.3, 0.6], f32), 1.1)
    frames = []
    for o in (0, order):
        with Context(256, 160, max_bounces=3, noise=noise, tuning=[(OPT_NODE_ORDER, o)]) as ctx:
            ctx.recreate_octree(pos, m)
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)
            assert ctx.stats().node_order == o
            frames.append(([ctx.read(i) for i in range(3)], ctx.stats().rays))
    for x, y in zip(frames[0][0], frames[1][0]):
        assert_bits_equal(x, y, "voxel-list scene")
    assert frames[0][1] == frames[1][1] and (frames[0][0][1][..., 3] >= 0).any()
