"""Predict the N-GPU trace-stage frame time on one GPU: run each rank's band set alone and take the slowest."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
W, H = 1920, 1080
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)
import os
for nranks in [int(v) for v in os.environ.get('RANKS', '1,2,4,8').split(',')]:
    for infl, batch in [tuple(int(x) for x in v.split('x')) if 'x' in v else (int(v), 1) for v in os.environ.get('INFL', '3,6,8').split(',')]:
        worst, total_rays, per_rank = 0.0, 0, []
        band = int(os.environ.get('BAND', '16'))
        for rank in (reversed(range(nranks)) if os.environ.get('ORDER') == 'rev' else range(nranks)):
            with Context(W, H, max_bounces=4, rank=rank, nranks=nranks, frames_in_flight=infl, frames_per_launch=batch, band_rows=band) as ctx:
                ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
                ctx.render_frames(TRACE, 20 * batch * infl); ctx.sync(); ctx.reset_stats()
                n = 40 * batch * infl if batch > 1 else 400
                t0 = time.perf_counter(); ctx.render_frames(TRACE, n); ctx.sync(); dt = (time.perf_counter() - t0) / n
                worst = max(worst, dt); total_rays += ctx.stats().rays / n; per_rank.append((round(dt * 1e3, 4), int(ctx.stats().rays / n)))
        print(f"nranks={nranks} inflight={infl} batch={batch}: slowest rank {worst * 1e3:.4f} ms/frame -> {total_rays / worst / 1e9:.1f} Gray/s aggregate; band rows {band}; per rank {per_rank}")
