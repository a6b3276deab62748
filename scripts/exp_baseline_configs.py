"""BASELINE.json's configs 2-5 as literally defined, measured on ONE MI355X.  Configs 4 and 5 name 8 GPUs: a rank's
band set is rendered alone on this GPU (what that rank would do on its own GPU, no halo transfer time) for two of the eight ranks.
One "displayed frame" = `spp` trace frames averaged (vxrt_render_spp) + temporal + denoise where the config says so."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from gpu_voxel_raytracer_amd import ALL, DENOISE_EDGE, DENOISE_INTERIOR, TRACE, TEMPORAL, Camera, Context, distributed, scenes


def run(label, w, h, bounces, spp, flags, radius, scene=None, menger=None, rank=0, nranks=1, cam=None, shown=12, batch=None, inflight=2, band=16,
        halo_loop=False):
    """halo_loop: a rank's whole frame of the multi-rank loop — trace + temporal, halo pack, denoise of the interior tiles, halo unpack,
    denoise of the edge tiles — with the rank's own messages handed back to it (the work of a rank without the transfer time)."""
    batch = batch or min(spp, 16)
    with Context(w, h, max_bounces=bounces, rank=rank, nranks=nranks, frames_per_launch=batch, frames_in_flight=inflight, band_rows=band) as ctx:
        if menger:
            ctx.set_menger(*menger)
        else:
            pos, mrgb, size = scenes.load_scene(scene)
            ctx.recreate_octree(pos, mrgb)
            cam = cam or scenes.bench_camera(size)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = radius
        render = (lambda: ctx.render_spp(flags, spp)) if spp > 1 else (lambda: ctx.render_frames(flags, 16))
        if halo_loop:
            rt = C.CDLL("libamdhip64.so")
            rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            bufs = [C.c_void_p() for _ in range(2)]
            nbytes = ctx.halo_bytes()
            for b in bufs:
                assert rt.hipMalloc(C.byref(b), nbytes) == 0
            # The rank's own messages stand in for its neighbours'.  A message is addressed by the RECEIVER's band slots, so rank 0's
            # message to its previous rank leaves the last slot unwritten and the last rank's message to its next rank the first one;
            # read back as "received", those slots would be whatever hipMalloc returned — non-finite colours, which the denoiser
            # treats by its slow literal path (round 3's table had rank 0 at 0.81-0.84 ms for that reason: an artefact of this
            # stand-in, not of the rank).  Fill them with another slot's rows after the first pack.
            info = ctx.halo_info()
            layout = distributed.BandLayout(w, h, nranks, band)
            plane = layout.plane(info.rows) * 16          # bytes of plane A (= B); plane C: a quarter
            slot = info.rows * w * 16
            rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

            def patch(buf, dst_slot, src_slot):
                for off, unit in ((0, slot), (plane, slot), (2 * plane, slot // 4)):
                    assert rt.hipMemcpy(buf.value + off + dst_slot * unit, buf.value + off + src_slot * unit, unit, 3) == 0

            ctx.render_spp(TRACE | TEMPORAL, spp)
            ctx.halo_pack(bufs[0].value, bufs[1].value)
            ctx.sync()
            if info.slots > 1:
                patch(bufs[0], info.slots - 1, 0)
                patch(bufs[1], 0, info.slots - 1)

            def render():
                ctx.render_spp(TRACE | TEMPORAL, spp)
                ctx.halo_pack(bufs[0].value, bufs[1].value)
                ctx.render_stage(DENOISE_INTERIOR)
                ctx.halo_unpack(bufs[0].value, bufs[1].value)
                ctx.render_stage(DENOISE_EDGE)
        per_call = spp if spp > 1 else 16
        for _ in range(3):
            render()
        ctx.sync(); ctx.reset_stats()
        t0 = time.perf_counter()
        for _ in range(shown):
            render()
        ctx.sync()
        dt = (time.perf_counter() - t0) / shown
        st = ctx.stats()
        frames = shown * per_call
        print(f"{label}: {dt * 1e3 / (1 if spp > 1 else 16):8.3f} ms per displayed frame ({spp} spp), {st.rays / frames / (w * h * (st.local_rows / h)):.2f} rays/px/sample, "
              f"{st.rays / (dt * shown) / 1e9:6.2f} Gray/s on this GPU", flush=True)


if os.environ.get("VXRT_EXP_ONLY_CONFIG4") != "1":
    run("config 2  menger 1920x1080, 1 spp, 4 bounces, trace only", 1920, 1080, 4, 1, TRACE, 0, scene="menger", batch=16)
for r in (2, 8) if os.environ.get("VXRT_EXP_ONLY_CONFIG4") != "1" else ():
    run(f"config 3  monu10 3840x2160, 4 spp, 8 bounces, temporal + denoise r={r}", 3840, 2160, 8, 4, ALL, r, scene="monu10")
only4 = os.environ.get("VXRT_EXP_ONLY_CONFIG4") == "1"
for rank in range(8):     # every rank: the slowest one sets the frame
    if rank in (0, 1, 5, 7) and not only4:
        run(f"config 4  castle 3840x2160, 4 spp, 8 bounces, rank {rank} of 8, 16-row bands (trace + temporal)", 3840, 2160, 8, 4,
            TRACE | TEMPORAL, 2, scene="castle", rank=rank, nranks=8)
    for band in (48, 64):
        run(f"config 4  castle 3840x2160, 4 spp, 8 bounces, rank {rank} of 8, {band}-row bands (trace + temporal)", 3840, 2160, 8, 4,
            TRACE | TEMPORAL, 2, scene="castle", rank=rank, nranks=8, band=band)
        run(f"config 4  castle 3840x2160, 4 spp, 8 bounces, rank {rank} of 8, {band}-row bands, the rank's whole loop with denoise r=8 (no transfer time)",
            3840, 2160, 8, 4, ALL, 8, scene="castle", rank=rank, nranks=8, band=band, halo_loop=True)
ext = np.float32(1024)
outside = (np.array([-0.9, 0.6, -1.2], np.float32) * ext + ext / 2, np.array([0.9, -0.6, 1.2], np.float32), 1.2217305)
for rank in (0, 5) if not only4 else ():
    run(f"config 5  2048^3 procedural Menger 7680x4320, 16 spp, 8 bounces, rank {rank} of 8 (trace + temporal)", 7680, 4320, 8, 16,
        TRACE | TEMPORAL, 0, menger=(7, 2048, (0, 150, 170, 120), 8192), rank=rank, nranks=8, cam=outside, shown=4)
