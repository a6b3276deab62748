"""usage: python scripts/exp_fused_one.py <rank> <nranks> <inflight> <batch> <fused 0|1> [blocks]: the 20-frame block, for a profiler pass"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
from gpu_voxel_raytracer_amd import host
from gpu_voxel_raytracer_amd.host import OPT_FUSED_TAIL
if int(sys.argv[5]):
    host.use_library(host.variants_library())      # fused_kernel lives in the -DVXRT_VARIANTS=1 build
rank, nranks, infl, batch, fused = (int(v) for v in sys.argv[1:6])
blocks = int(sys.argv[6]) if len(sys.argv) > 6 else 30
pos, mrgb, size = scenes.load_scene("menger")
with Context(1920, 1080, max_bounces=4, rank=rank, nranks=nranks, band_rows=8, frames_in_flight=infl, frames_per_launch=batch, tuning=[(OPT_FUSED_TAIL, fused)]) as ctx:
    ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*scenes.bench_camera(size))
    for _ in range(10):
        ctx.render_frames(TRACE, 20); ctx.sync()
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter(); ctx.render_frames(TRACE, 20); ctx.sync(); ts.append(time.perf_counter() - t0)
    print(f"rank {rank}/{nranks} {batch}x{infl} fused {fused}: {statistics.median(ts) * 1e3:.4f} ms", flush=True)
    if fused:
        import ctypes as C
        out = (C.c_uint64 * 8)()
        ctx._L.vxrt_debug_fused_profile(ctx._h, out)
        print("   fused profile (every 64th wave): %.1f us to heads done, %.1f us to end; chunks before / after %d / %d; idle sleeps %d; stamp polls %d; head claims %d" % ((out[0] / 100.0, out[1] / 100.0) + tuple(out[2:7])), flush=True)
