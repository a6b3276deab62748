"""The halo's two kernels alone on the chip (one process, one rank's band set): castle at 3840x2160, denoise r = 8, 64-row bands —
BASELINE configs[3]'s layout — for 2, 4 and 8 ranks: message bytes, pack and unpack time per exchange (HIP events inside the library).
usage (GPU box): python scripts/exp_halo_kernels.py [N]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import TEMPORAL, TRACE, Camera, Context, distributed, scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H, R = 3840, 2160, 8
rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
pos, mrgb, size = scenes.load_scene("castle")
for nranks in (2, 4, 8):
    band = distributed.band_rows_for(R)
    with Context(W, H, max_bounces=8, rank=1, nranks=nranks, band_rows=band) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*scenes.close_camera(size))
        ctx.denoise_uniforms.radius = R
        ctx.render(TRACE | TEMPORAL)
        info = ctx.halo_info()
        bufs = [C.c_void_p() for _ in range(4)]
        for b in bufs:
            assert rt.hipMalloc(C.byref(b), info.message_bytes) == 0
        for _ in range(5):
            ctx.halo_pack(bufs[0].value, bufs[1].value)
            ctx.halo_unpack(bufs[0].value, bufs[1].value)
        ctx.sync(); ctx.reset_stats()
        for _ in range(N):
            ctx.halo_pack(bufs[0].value, bufs[1].value)
            ctx.halo_unpack(bufs[0].value, bufs[1].value)
        st = ctx.stats()
        layout = distributed.BandLayout(W, H, nranks, band)
        px = layout.halo_pixels_per_rank(1, info.rows)
        pack, unpack = st.halo_pack_ms / N * 1e3, st.halo_unpack_ms / N * 1e3
        print(f"{nranks} ranks, {band}-row bands, {info.rows} halo rows: 2 x {info.message_bytes / 1e6:.2f} MB per rank and frame ({px} halo pixels sent); "
              f"pack {pack:.1f} us ({px * 84 / pack / 1e6:.2f} TB/s of 84 B/px), unpack {unpack:.1f} us ({4 * info.message_bytes / unpack / 1e6:.2f} TB/s), "
              f"pack + unpack {pack + unpack:.1f} us", flush=True)
