"""GPU: the multi-GPU decomposition exercised inside ONE process on ONE device — N contexts
with rank = 0..N-1 render their interleaved 16-row bands; stitched together they must equal the
single-context frame bit for bit, including the denoise stage fed through the halo export/import path
(the buffers RCCL would carry between GPUs are handed over directly here)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipFree.argtypes = [C.c_void_p]
    lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return lib


@pytest.mark.parametrize("nranks,w,h,radius,band,split", [(2, 128, 80, 3, 16, False), (3, 96, 100, 8, 16, True), (4, 160, 72, 1, 16, False),
                                                          (8, 64, 200, 8, 16, False), (5, 80, 40, 2, 16, True), (2, 100, 300, 8, 64, True),
                                                          (3, 64, 250, 4, 32, True), (2, 72, 130, 8, 48, False),
                                                          # frames that are not a whole number of rounds (kernels.h: BandMap): the last round takes
                                                          # the remainder in taller bands (240 rows / 3 ranks: 80-row bands; 224 / 4: 64 + 64 + 64 + 32)
                                                          (3, 64, 240, 8, 48, True), (4, 96, 224, 4, 48, False), (8, 64, 432, 8, 48, True)])
def test_bands_and_halo_equal_single_context(H, scenes, noise, nranks, w, h, radius, band, split):
    """split: the denoise stage in two launches around the exchange (DENOISE_INTERIOR before the unpack, DENOISE_EDGE after)."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, DENOISE_EDGE, DENOISE_INTERIOR, TEMPORAL, TRACE, Camera, Context
    from gpu_voxel_raytracer_amd.distributed import BandLayout
    pos, mrgb, size = scenes.load_scene("castle")
    cam = Camera(*scenes.close_camera(size))
    rt = hip()

    def setup(ctx):
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = cam
        ctx.denoise_uniforms.radius = radius

    with Context(w, h, max_bounces=3, noise=noise) as single:
        setup(single)
        ctxs = [Context(w, h, max_bounces=3, noise=noise, rank=r, nranks=nranks, band_rows=band) for r in range(nranks)]
        try:
            for c in ctxs:
                setup(c)
            rows = [c.local_rows() for c in ctxs]
            assert sorted(np.concatenate(rows).tolist()) == list(range(h))          # a partition of the frame
            layout = BandLayout(w, h, nranks, band)
            for r in range(nranks):                                                # ... the one distributed.BandLayout states,
                assert np.array_equal(rows[r], layout.rows(r))
            assert max(len(x) for x in rows) - h / nranks < 16                      # every rank within a tile row of an even share
            for r, c in enumerate(ctxs):      # the message layout and the tile split as distributed.py states them
                info = c.halo_info()
                interior, edge = layout.tile_rows(r)
                assert (info.rows, info.slots, info.bytes_per_pixel) == (layout.halo_rows(radius), layout.max_bands(), 36)
                assert info.message_bytes == layout.message_floats(info.rows) * 4 == c.halo_bytes()
                assert (info.interior_tile_rows, info.edge_tile_rows) == (len(interior), len(edge))
            for frame in range(3):
                single.render(ALL)
                for c in ctxs:
                    c.render(TRACE | TEMPORAL)
                # halo exchange: rank r sends its band edges to r-1 and r+1 (mod N)
                nbytes = ctxs[0].halo_bytes()
                bufs = {}
                for r, c in enumerate(ctxs):
                    p, n = C.c_void_p(), C.c_void_p()
                    assert rt.hipMalloc(C.byref(p), nbytes) == 0 and rt.hipMalloc(C.byref(n), nbytes) == 0
                    rt.hipMemset(p, 0xff, nbytes); rt.hipMemset(n, 0xff, nbytes)
                    rt.hipDeviceSynchronize()   # hipMemset is not ordered against the context's non-blocking stream
                    c.halo_export(p.value, n.value)
                    bufs[r] = (p, n)
                    if split:
                        c.render_stage(DENOISE_INTERIOR)           # needs nothing from the neighbours
                for r, c in enumerate(ctxs):
                    from_prev = bufs[(r - 1) % nranks][1]      # what the previous rank addressed to its next
                    from_next = bufs[(r + 1) % nranks][0]      # what the next rank addressed to its prev
                    c.halo_import(from_prev.value, from_next.value)
                    c.render_stage(DENOISE_EDGE if split else DENOISE)
                for c in ctxs:
                    c.sync()
                for p, n in bufs.values():
                    rt.hipFree(p); rt.hipFree(n)
                for img in range(5):
                    want = single.read(img)
                    got = np.zeros_like(want)
                    for c, rr in zip(ctxs, rows):
                        if len(rr):
                            got[rr] = c.read(img)
                    assert_bits_equal(got, want, f"image {img} frame {frame + 1} nranks {nranks}")
            total = sum(c.stats().rays for c in ctxs)
            assert total == single.stats().rays
        finally:
            for c in ctxs:
                c.close()


def test_multirank_denoise_without_halo_is_refused(H, scenes, noise):
    from gpu_voxel_raytracer_amd import ALL, DENOISE, DENOISE_EDGE, DENOISE_INTERIOR, TEMPORAL, TRACE, Camera, Context, VxrtError
    pos, mrgb, size = scenes.load_scene("8x8x8")
    with Context(64, 64, rank=1, nranks=2, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = 2
        with pytest.raises(VxrtError):
            ctx.render(ALL)
        ctx.render(TRACE | TEMPORAL)
        ctx.render_stage(DENOISE_INTERIOR)                # the tiles that need no neighbour may run before the exchange
        with pytest.raises(VxrtError, match="no halo"):
            ctx.render_stage(DENOISE_EDGE)
        with pytest.raises(VxrtError, match="no halo"):
            ctx.render_stage(DENOISE)
        with pytest.raises(VxrtError, match="does not combine"):
            ctx.render_stage(DENOISE | DENOISE_EDGE)
        ctx.denoise_uniforms.radius = 0
        ctx.render(ALL)                                   # radius 0 needs no neighbours
    with pytest.raises(VxrtError):
        Context(64, 64, rank=2, nranks=2)
    with pytest.raises(VxrtError):
        Context(64, 64, rank=0, nranks=2, band_rows=12)          # a multiple of 8 (the tracer's tiles); 16 for a denoise window
    with pytest.raises(VxrtError):
        Context(64, 64, rank=0, nranks=2, band_rows=6)           # 2, 4 or a multiple of 8: 6 is none of them (BandLayout refuses it too)


@pytest.mark.parametrize("nranks,w,h,band", [(8, 96, 136, 8), (3, 64, 100, 8), (8, 96, 136, 4), (8, 72, 150, 2), (3, 64, 100, 4)])
def test_eight_row_bands_without_denoise_window(H, scenes, noise, nranks, w, h, band):
    """band_rows = 8 (bench.py's deal at N > 1: the tracer's tile height), and 4 or 2 (round 5: the finer interleave bench.py takes for a
    short block from 8 ranks on — a tile of 8 local rows then spans two or four bands, each lane finds its own frame row), through
    trace, temporal and the radius-0 denoise equals the single-context frame; a denoise radius > 0 is refused for such bands (its
    16-row tiles would straddle two of them)."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, TEMPORAL, TRACE, Camera, Context
    from gpu_voxel_raytracer_amd.host import VxrtError
    pos, mrgb, size = scenes.load_scene("castle")
    cam = Camera(*scenes.close_camera(size))
    with Context(w, h, max_bounces=3, noise=noise) as single:
        single.recreate_octree(pos, mrgb)
        single.camera = cam
        ctxs = [Context(w, h, max_bounces=3, noise=noise, rank=r, nranks=nranks, band_rows=band, frames_per_launch=2) for r in range(nranks)]
        try:
            rows = []
            for c in ctxs:
                c.recreate_octree(pos, mrgb)
                c.camera = cam
                rows.append(c.local_rows())
            assert sorted(np.concatenate(rows).tolist()) == list(range(h))
            assert rows[1][:band].tolist() == list(range(band, 2 * band))
            from gpu_voxel_raytracer_amd.distributed import BandLayout
            for r in range(nranks):
                assert np.array_equal(rows[r], BandLayout(w, h, nranks, band, radius=0).rows(r))
            for frame in range(3):
                single.render(ALL)
                for c in ctxs:
                    c.render(ALL)
                for img in range(5):
                    want = single.read(img)
                    got = np.zeros_like(want)
                    for c, rr in zip(ctxs, rows):
                        got[rr] = c.read(img)
                    assert_bits_equal(got, want, f"image {img} frame {frame + 1}")
            ctxs[0].denoise_uniforms.radius = 2
            with pytest.raises(VxrtError, match="multiple of 16"):
                ctxs[0].render(DENOISE)
            del TEMPORAL, TRACE
        finally:
            for c in ctxs:
                c.close()


@pytest.mark.parametrize("radius,halo_rows,band", [(0, 1, 16), (0, 4, 8), (2, 1, 16), (1, 16, 32)])
def test_history_rows_cross_band_edges_without_a_denoise_window(H, scenes, noise, radius, halo_rows, band):
    """The halo carries max(radius, VXRT_OPT_HALO_ROWS) rows, at least 1 by default, also with radius 0 (the reference's default): the
    temporal stage of the next frame finds the neighbours' history there.  A camera at rest (whose reprojection still touches the next
    row for the ~1 % of pixels where the divide rounds off the texel centre) and a slow drift (image motion below halo_rows rows per
    frame) give the single-context frames bit for bit, accumulated colour included; without any history rows they would not."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, TEMPORAL, TRACE, Camera, Context
    from gpu_voxel_raytracer_amd.host import OPT_HALO_ROWS
    w, h, nranks = 192, 208, 2
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    rt = hip()
    step = np.float32(0.0004 * halo_rows) * np.array([0.3, 1.0, 0.0], np.float32)
    for name, path in (("rest", [(p0, d0)] * 3), ("drift", [(p0 + np.float32(k) * step, d0) for k in range(4)])):
        with Context(w, h, max_bounces=2, noise=noise) as single:
            ctxs = [Context(w, h, max_bounces=2, noise=noise, rank=r, nranks=nranks, band_rows=band) for r in range(nranks)]
            try:
                for c in [single] + ctxs:
                    c.recreate_octree(pos, mrgb)
                    c.denoise_uniforms.radius = radius
                for c in ctxs:
                    c.set_option(OPT_HALO_ROWS, halo_rows)
                rows = [c.local_rows() for c in ctxs]
                for cp, cd in path:
                    for c in [single] + ctxs:
                        c.camera = Camera(cp, cd, fov)
                    single.render(ALL)
                    nbytes = None
                    bufs = {}
                    for r, c in enumerate(ctxs):
                        c.render(TRACE | TEMPORAL | (DENOISE if radius == 0 else 0))
                        nbytes = c.halo_bytes()
                        assert c.halo_info().rows == max(radius, halo_rows)
                        p, n = C.c_void_p(), C.c_void_p()
                        assert rt.hipMalloc(C.byref(p), nbytes) == 0 and rt.hipMalloc(C.byref(n), nbytes) == 0
                        c.halo_export(p.value, n.value)
                        bufs[r] = (p, n)
                    for r, c in enumerate(ctxs):
                        c.halo_import(bufs[(r - 1) % nranks][1].value, bufs[(r + 1) % nranks][0].value)
                        if radius > 0:
                            c.render_stage(DENOISE)
                    for c in ctxs:
                        c.sync()
                    for p, n in bufs.values():
                        rt.hipFree(p); rt.hipFree(n)
                    for img in (0, 3, 4):
                        want = single.read(img)
                        got = np.zeros_like(want)
                        for c, rr in zip(ctxs, rows):
                            got[rr] = c.read(img)
                        assert_bits_equal(got, want, f"{name}: image {img}, radius {radius}, halo rows {halo_rows}")
            finally:
                for c in ctxs:
                    c.close()


def test_fast_pan_keeps_its_history_on_every_band_edge_when_the_frame_is_not_whole_rounds(H, scenes, noise):
    """round 4 dealt a frame's remainder as an extra round of LOWER bands and silently capped the halo of the whole frame at
    them (304 rows on 4 ranks with 32-row bands: 16 rows), so a pan faster than that lost its history at every band edge although the
    documentation promised band_rows - 2.  Now the last round is taller instead (bands of 48 there), vxrt_halo_info.max_rows says what the
    layout carries (32), and a ~20-row tilt sized by halo_rows_for_motion against that cap gives the single-context frames bit for bit."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, TEMPORAL, TRACE, Camera, Context, distributed
    from gpu_voxel_raytracer_amd.host import OPT_HALO_ROWS
    w, h, nranks, band, radius = 160, 304, 4, 32, 2
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    tilt = np.float32(0.06 * float(np.linalg.norm(d0))) * np.array([0, -1, 0], np.float32)
    path = [(p0, d0), (p0, d0), (p0, d0 + tilt), (p0, d0 + tilt)]
    layout = distributed.BandLayout(w, h, nranks, band, radius=radius)
    assert (layout.full_bands, layout.tail_rows, layout.halo_rows_max()) == (4, 48, 32)
    rt = hip()
    with Context(w, h, max_bounces=2, noise=noise) as single:
        ctxs = [Context(w, h, max_bounces=2, noise=noise, rank=r, nranks=nranks, band_rows=band) for r in range(nranks)]
        try:
            for c in [single] + ctxs:
                c.recreate_octree(pos, mrgb)
                c.denoise_uniforms.radius = radius
            rows = [c.local_rows() for c in ctxs]
            for r in range(nranks):
                assert np.array_equal(rows[r], layout.rows(r))
            assert ctxs[0].halo_info().max_rows == 32
            chosen = []
            for k, (cp, cd) in enumerate(path):
                nxt = path[min(k + 1, len(path) - 1)]
                axes = [(c[0],) + Camera(c[0], c[1], fov).axis_scaled(w, h) for c in ((cp, cd), nxt)]
                want_rows = distributed.halo_rows_for_motion(axes[0], axes[1], w, h, near=0.25, band_rows=layout.halo_rows_max())
                chosen.append(want_rows)
                for c in [single] + ctxs:
                    c.camera = Camera(cp, cd, fov)
                single.render(ALL)
                bufs = {}
                for r, c in enumerate(ctxs):
                    c.set_option(OPT_HALO_ROWS, want_rows)
                    c.render(TRACE | TEMPORAL)
                    nbytes = c.halo_bytes()
                    assert c.halo_info().rows == max(radius, want_rows)
                    p, n = C.c_void_p(), C.c_void_p()
                    assert rt.hipMalloc(C.byref(p), nbytes) == 0 and rt.hipMalloc(C.byref(n), nbytes) == 0
                    c.halo_export(p.value, n.value)
                    bufs[r] = (p, n)
                for r, c in enumerate(ctxs):
                    c.halo_import(bufs[(r - 1) % nranks][1].value, bufs[(r + 1) % nranks][0].value)
                    c.render_stage(DENOISE)
                for c in ctxs:
                    c.sync()
                for p, n in bufs.values():
                    rt.hipFree(p); rt.hipFree(n)
                for img in (0, 3, 4):
                    want = single.read(img)
                    got = np.zeros_like(want)
                    for c, rr in zip(ctxs, rows):
                        got[rr] = c.read(img)
                    assert_bits_equal(got, want, f"frame {k}: image {img}, halo rows {want_rows}")
            assert chosen[0] == 2 and 17 < chosen[1] < 32 and chosen[2] == 2, chosen      # deeper than round 4's cap of 16, below this layout's 32
            assert (single.read(1)[..., 3] >= 0).mean() > 0.3                             # the pan looks at geometry, not at sky
        finally:
            for c in ctxs:
                c.close()


def test_short_block_deal_of_eight_ranks_equals_one_context(H, scenes, noise):
    """bench.py's deal of the driver's 20-frame block on 8 ranks (pick_deal: 4-row bands, the all-in-one kernel, all 20 frames in ONE
    launch — frame lanes of 2 rows x 4 frames) against one context that renders the same 20 frames the default way (8 + 8 + 4 frames
    on three streams, head + compacted tail): the last frame of the block, stitched from the 8 band sets, bit for bit — colour,
    normal / depth, albedo / node — and the ray totals.  (the new deal's frames against the single-launch frames.)"""
    import importlib.util, os
    from conftest import ROOT
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    band, tracer = bench.pick_deal(8, 20)
    inflight, batch = bench.pick_schedule(8, 20)
    assert (band, tracer, inflight, batch) == (4, 1, 1, 20)
    w, h, nranks, frames = 1920, 1080, 8, 20
    pos, mrgb, size = scenes.load_scene("menger")
    cam = Camera(*scenes.bench_camera(size))
    i1, b1 = bench.pick_schedule(1, 20)
    with Context(w, h, max_bounces=4, noise=noise, frames_in_flight=i1, frames_per_launch=b1) as single:
        single.recreate_octree(pos, mrgb)
        single.camera = cam
        single.render_frames(TRACE, frames)
        want = [single.read(i) for i in range(3)]
        want_rays = single.stats().rays
    got = [np.zeros_like(x) for x in want]
    rays = 0
    for r in range(nranks):
        with Context(w, h, max_bounces=4, noise=noise, rank=r, nranks=nranks, band_rows=band, frames_in_flight=inflight, frames_per_launch=batch, tracer=tracer) as c:
            c.recreate_octree(pos, mrgb)
            c.camera = cam
            c.render_frames(TRACE, frames)
            rows = c.local_rows()
            assert len(rows) - h / nranks < 4                   # the busiest rank within a tile row of an even share (one rank takes what is left: 128 rows)
            for i in range(3):
                got[i][rows] = c.read(i)
            rays += c.stats().rays
            assert c.stats().frame_lane_launches == 1          # the one launch of 20 frames held 4 frames of 2 rows per wave
    for i, label in enumerate(("colour", "normal/depth", "albedo/node")):
        assert_bits_equal(got[i], want[i], f"frame 20 of the block, {label}: 8 ranks' deal against one context")
    assert rays == want_rays
