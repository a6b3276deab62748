"""Host cost of submitting trace launches against the GPU time they take, for one rank's band set of the trace-only bench
(menger 1080p): is a rank of 8 host-bound?  usage: python scripts/exp_host_submit.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)
for nranks in (1, 8):
    for infl, batch in ((2, 16), (3, 16), (3, 8), (3, 32)):
        with Context(1920, 1080, max_bounces=4, rank=0, nranks=nranks, frames_in_flight=infl, frames_per_launch=batch, band_rows=8) as ctx:
            ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
            ctx.render_frames(TRACE, 20 * batch * infl); ctx.sync()
            n = 60 * batch
            t0 = time.perf_counter(); ctx.render_frames(TRACE, n); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
            launches = n // batch
            print(f"nranks={nranks} {batch}x{infl}: submit {(t1 - t0) / launches * 1e6:.1f} us per launch (host), wall {(t2 - t0) / launches * 1e6:.1f} us per launch "
                  f"= {(t2 - t0) / n * 1e3:.4f} ms/frame; host share {(t1 - t0) / (t2 - t0):.2f}", flush=True)
