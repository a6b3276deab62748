"""GPU: the non-blocking read-back of the C ABI (vxrt_read_async / vxrt_read_wait / vxrt_host_alloc, include/vxrt.h) — what a host
that shows or encodes every frame uses in place of the reference's present (src/context.rs:2046-2070).  A transfer carries the image
as the stages enqueued before it left it, whatever is rendered afterwards; two transfers can be in flight; every image is readable."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def _ctx(scenes, noise, w=200, h=120, **kw):
    from gpu_voxel_raytracer_amd import Camera, Context
    pos, mrgb, size = scenes.load_scene("castle")
    ctx = Context(w, h, max_bounces=3, noise=noise, **kw)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*scenes.close_camera(size))
    ctx.denoise_uniforms.radius = 2
    return ctx


def test_async_read_equals_sync_read_for_every_image(H, scenes, noise):
    from gpu_voxel_raytracer_amd import ALL
    with _ctx(scenes, noise) as ctx:
        ctx.render(ALL)
        ctx.render(ALL)
        buf = ctx.pinned_image()
        for which in range(5):
            want = ctx.read(which)
            ctx.read_async(which, buf, which & 1)
            ctx.read_wait(which & 1)
            assert_bits_equal(buf.array, want, f"image {which}")
        buf.close()


@pytest.mark.parametrize("inflight,batch", [(1, 1), (2, 4)])
def test_a_transfer_carries_the_frame_it_was_asked_for(H, scenes, noise, inflight, batch):
    """Frames keep being rendered while earlier frames travel: every transfer holds ITS frame (the snapshot is taken on the context's
    stream before later stages may overwrite the image), slots alternate, nothing is waited for until the end."""
    from gpu_voxel_raytracer_amd import ALL, DENOISED, SAMPLED_COLOR
    with _ctx(scenes, noise, frames_in_flight=inflight, frames_per_launch=batch) as ref:
        want = []
        for f in range(4):
            ref.render(ALL)
            want.append((ref.read(DENOISED), ref.read(SAMPLED_COLOR)))
    with _ctx(scenes, noise, frames_in_flight=inflight, frames_per_launch=batch) as ctx:
        bufs = [[ctx.pinned_image(), ctx.pinned_image()] for _ in range(2)]      # [slot][kept copy]
        got = []
        for f in range(4):
            ctx.render(ALL)
            if f >= 2:          # the slot's previous transfer: wait for it, keep what arrived
                ctx.read_wait(f & 1)
                got.append(bufs[f & 1][0].array.copy())
            ctx.read_async(DENOISED, bufs[f & 1][0], f & 1)
        for f in (2, 3):
            ctx.read_wait(f & 1)
            got.append(bufs[f & 1][0].array.copy())
        for f in range(4):
            assert_bits_equal(got[f], want[f][0], f"denoised frame {f + 1}")
        # a trace image, read while the next frame is already being traced into the ring
        ctx.read_async(SAMPLED_COLOR, bufs[0][1], 0)
        ctx.render(ALL)
        ctx.read_wait(0)
        assert_bits_equal(bufs[0][1].array, want[3][1], "sampled colour of frame 4, read across frame 5")
        for pair in bufs:
            for b in pair:
                b.close()


def test_pageable_destination_resize_and_bad_arguments(H, scenes, noise):
    from gpu_voxel_raytracer_amd import ALL, DENOISED
    with _ctx(scenes, noise) as ctx:
        ctx.render(ALL)
        plain = np.zeros((120, 200, 4), np.float32)             # pageable memory works too (slowly)
        ctx.read_async(DENOISED, plain, 1)
        ctx.read_wait(1)
        assert_bits_equal(plain, ctx.read(DENOISED), "pageable destination")
        L, h = ctx._L, ctx._h
        assert L.vxrt_read_async(h, C.c_int(DENOISED), plain.ctypes.data_as(C.c_void_p), C.c_size_t(plain.nbytes), C.c_uint32(2)) == H.E_INVALID      # slot
        assert L.vxrt_read_async(h, C.c_int(DENOISED), plain.ctypes.data_as(C.c_void_p), C.c_size_t(plain.nbytes - 16), C.c_uint32(0)) == H.E_INVALID  # size
        assert L.vxrt_read_async(h, C.c_int(DENOISED), None, C.c_size_t(plain.nbytes), C.c_uint32(0)) == H.E_INVALID
        assert L.vxrt_read_async(h, C.c_int(9), plain.ctypes.data_as(C.c_void_p), C.c_size_t(plain.nbytes), C.c_uint32(0)) == H.E_INVALID
        assert L.vxrt_read_wait(h, C.c_uint32(2)) == H.E_INVALID
        assert L.vxrt_read_wait(h, C.c_uint32(0)) == 0           # nothing in flight: returns at once
        ctx.resize(320, 176)                                       # larger images: the slots' stages grow
        ctx.render(ALL)
        big = ctx.pinned_image()
        assert big.array.shape == (176, 320, 4)
        ctx.read_async(DENOISED, big, 1)
        ctx.read_wait(1)
        assert_bits_equal(big.array, ctx.read(DENOISED), "after resize")
        big.close()


def test_async_read_of_a_rank_s_band_set(H, scenes, noise):
    """A multi-GPU context owns interleaved bands of the frame: the non-blocking read-back carries its local rows (vxrt_local_rows), as
    vxrt_read does — what a rank hands to whoever assembles or shows the frame."""
    from gpu_voxel_raytracer_amd import ACCUM_COLOR, SAMPLED_COLOR, TEMPORAL, TRACE
    for rank in (0, 2):
        with _ctx(scenes, noise, w=200, h=136, rank=rank, nranks=3, band_rows=8) as ctx:
            ctx.denoise_uniforms.radius = 0
            ctx.render(TRACE | TEMPORAL)
            rows = ctx.local_rows()
            buf = ctx.pinned_image()
            assert buf.array.shape == (len(rows), 200, 4) and 0 < len(rows) < 136
            for which in (SAMPLED_COLOR, ACCUM_COLOR):
                ctx.read_async(which, buf, 1)
                ctx.read_wait(1)
                assert_bits_equal(buf.array, ctx.read(which), f"rank {rank} image {which}")
            buf.close()
