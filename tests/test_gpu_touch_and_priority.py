"""GPU: two instruments of round 6 that must not change a frame.
* The touch map (vxrt_debug.h: vxrt_debug_touch_map / _count; -DVXRT_VARIANTS=1 library only): which 64-byte lines of the scene a frame
  reads — SURVEY 8d's "bricks actually touched", the unique scene bytes behind config 5's roofline in bench.py.
* VXRT_OPT_TRACE_PRIORITY: a trace launch as two grids on streams of different priority (the tiles that walk / the tiles that only
  store sky) — an experiment for the short block of a rank of 8; same image."""
import numpy as np
import pytest

from conftest import assert_bits_equal, require_variants

pytestmark = pytest.mark.gpu


def test_touch_map_counts_the_lines_a_frame_reads_and_changes_nothing(H, scenes, noise):
    from gpu_voxel_raytracer_amd import SAMPLED_COLOR, TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("menger")
    cam = Camera(*scenes.bench_camera(size))
    with Context(320, 200, max_bounces=4, noise=noise) as ctx:          # the product: no such code in its kernels
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = cam
        with pytest.raises(H.VxrtError, match="VXRT_VARIANTS"):
            ctx.touch_map(True)
        ctx.render(TRACE)
        want = ctx.read(SAMPLED_COLOR)
        scene_bytes = ctx.stats().scene_bytes
    require_variants(H, tracer=5)
    for wide in (0, 1):
        with Context(320, 200, max_bounces=4, noise=noise, tuning=[(H.OPT_SCENE_FORMAT, wide)]) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = cam
            with pytest.raises(H.VxrtError):
                ctx.touch_count()                                      # no map yet
            ctx.touch_map(True)
            ctx.render(TRACE)
            assert_bits_equal(ctx.read(SAMPLED_COLOR), want, "frame 1 with the touch map on")
            t = ctx.touch_count(reset=True)
            assert 0 < t["node_lines_64"] <= t["node_lines_total"] and 0 < t["leaf_lines_64"] <= t["leaf_lines_total"]
            assert t["node_lines_128"] <= t["node_lines_64"] <= 2 * t["node_lines_128"]
            assert t["unique_bytes_64"] < t["scene_bytes"]              # the view from outside sees a part of the sponge only
            if not wide:
                assert abs(t["scene_bytes"] - scene_bytes) <= 128      # whole lines of the two arrays
            again = ctx.touch_count(reset=False)
            assert again["unique_bytes_64"] == 0                        # reset cleared it
            ctx.set_frame_number(0)
            ctx.render(TRACE)                                           # the same frame once more: the same lines
            t2 = ctx.touch_count()
            assert t2["unique_bytes_64"] == t["unique_bytes_64"]
            ctx.touch_map(False)
            ctx.render(TRACE)                                           # and off again: frame 2, unmarked
            ctx.recreate_octree(pos, mrgb)                              # a new scene drops a map that is on
            ctx.touch_map(True)
            ctx.recreate_octree(pos, mrgb)
            with pytest.raises(H.VxrtError):
                ctx.touch_count()


@pytest.mark.parametrize("inflight,batch,tracer", [(1, 1, 1), (1, 4, 1), (2, 8, 0), (1, 20, 1)])
def test_priority_split_launches_give_identical_frames(H, scenes, noise, inflight, batch, tracer):
    from gpu_voxel_raytracer_amd import NORMAL_DEPTH, SAMPLED_COLOR, TRACE, Camera, Context
    pos, mrgb, size = scenes.load_scene("menger")
    cam = Camera(*scenes.bench_camera(size))
    n = 3 * inflight * batch + 1
    with pytest.raises(H.VxrtError, match="VXRT_VARIANTS"):          # measured slower: like every schedule that lost, the product refuses it
        Context(64, 64, tuning=[(H.OPT_TRACE_PRIORITY, 1)])
    require_variants(H, tracer=5)

    def frames(tuning, **kw):
        with Context(480, 272, max_bounces=4, noise=noise, frames_in_flight=inflight, frames_per_launch=batch, tracer=tracer, tuning=tuning, **kw) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = cam
            ctx.render_frames(TRACE, n)
            ctx.sync()                      # the walking-tile count of the first sort has reached the host: the next launches may split
            ctx.render_frames(TRACE, n)
            st = ctx.stats()
            return ctx.read(SAMPLED_COLOR), ctx.read(NORMAL_DEPTH), st
    c0, n0, st0 = frames([])
    c1, n1, st1 = frames([(H.OPT_TRACE_PRIORITY, 1)])
    assert_bits_equal(c1, c0, "colour")
    assert_bits_equal(n1, n0, "normal / depth")
    assert st1.rays == st0.rays and st0.split_launches == 0
    assert st1.split_launches >= 1, "no launch went out as two grids"
    # a rank's band set as well (the case the experiment is for)
    c2, _, st2 = frames([(H.OPT_TRACE_PRIORITY, 1)], rank=1, nranks=4, band_rows=8)
    c3, _, _ = frames([], rank=1, nranks=4, band_rows=8)
    assert_bits_equal(c2, c3, "colour, rank 1 of 4")


def test_xcd_affine_tile_order_gives_identical_frames(H, scenes, noise):
    """VXRT_OPT_XCD_AFFINITY (vxrt_debug.h): for a scene beyond the Infinity Cache the launch order deals the walking tiles to the XCDs by
    screen region.  A launch order never changes a pixel: same frames as without it, and the order is a permutation of the tiles."""
    import ctypes as C
    from gpu_voxel_raytracer_amd import NORMAL_DEPTH, SAMPLED_COLOR, TRACE, Camera, Context
    require_variants(H, tracer=5)                                    # an experiment's option: the variants library holds it
    w, h = 640, 360
    ext = np.float32(729 * 0.5)                                        # world extent of 729 voxels
    pos = (np.array([-0.45, 0.30, -0.55], np.float32) * ext + ext / 2).astype(np.float32)
    cam = Camera(pos, (np.full(3, ext / 2, np.float32) - pos).astype(np.float32), 1.2217305)

    def frames(S):
        with Context(w, h, max_bounces=4, noise=noise) as ctx:
            ctx.set_menger(6, 0, (0, 150, 170, 120), 4096)          # 729^3: 64 M voxels, > 256 MB of records and leaf words
            assert ctx.stats().scene_bytes > (256 << 20)
            ctx.camera = cam
            ctx.set_option(H.OPT_XCD_AFFINITY, S)
            ctx.render_frames(TRACE, 11)
            order = np.zeros((w // 8) * (h // 8), np.uint32)
            walking, spread = C.c_uint32(0), C.c_uint32(0)
            ctx._chk(ctx._L.vxrt_debug_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), None, C.c_size_t(order.size), C.byref(walking), C.byref(spread)), "tile order")
            return ctx.read(SAMPLED_COLOR), ctx.read(NORMAL_DEPTH), ctx.stats().rays, order
    c0, n0, r0, o0 = frames(0)
    assert (n0[..., 3] >= 0).mean() > 0.2                            # the sponge is in view
    for S in (1, 4, 16):
        c1, n1, r1, o1 = frames(S)
        assert_bits_equal(c1, c0, f"colour, S = {S}")
        assert_bits_equal(n1, n0, f"normal / depth, S = {S}")
        assert r1 == r0
        assert sorted(o1.tolist()) == list(range(o1.size)), "the order is not a permutation of the tiles"
        assert not np.array_equal(o1, o0)
