"""The driver's block (--steps 20: 20 frames between two synchronisations) for one rank's band set of the trace-only bench: which deal of
the 20 frames to launches and streams is fastest, for 1 / 2 / 4 / 8 ranks (rank 0 of each, alone on the GPU).
usage: python scripts/exp_short_block.py [steps]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)
deals = ((3, 7), (1, 20), (2, 10), (4, 5), (2, 5), (3, 4), (1, 10))
if os.environ.get("UNEVEN"):   # unequal parts: a launch of `batch` frames, then the rest
    deals = ((1, 20), (2, 10), (2, 12), (2, 14), (2, 16), (3, 8), (3, 10))
if os.environ.get("LANES"):    # round 3, frame lanes: launches of 8 or 16 frames hold 8 frames per wave
    deals = ((3, 7), (3, 8), (2, 8), (2, 16), (1, 16), (1, 20), (1, 8), (2, 10))
for nranks in (1, 2, 4, 8):
    for infl, batch in deals:
        with Context(1920, 1080, max_bounces=4, rank=0, nranks=nranks, frames_in_flight=infl, frames_per_launch=batch, band_rows=8) as ctx:
            ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
            for _ in range(20):
                ctx.render_frames(TRACE, steps); ctx.sync()
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); ctx.render_frames(TRACE, steps); ctx.sync(); ts.append(time.perf_counter() - t0)
            print(f"nranks={nranks} {batch}x{infl}: block of {steps} frames {statistics.median(ts) * 1e3:.4f} ms = {statistics.median(ts) / steps * 1e3:.4f} ms/frame", flush=True)
