"""One rank's band set of the trace-only bench, the driver's short block (--steps 20 between two synchronisations), a few blocks:
run under `rocprofv3 --kernel-trace --output-format csv` to see the block's kernels on a time line (scripts/timeline_summary.py).
usage: python scripts/exp_block_timeline.py <rank> <nranks> <inflight> <batch> [steps] [blocks]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
rank, nranks, infl, batch = (int(v) for v in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
blocks = int(sys.argv[6]) if len(sys.argv) > 6 else 40
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)
with Context(1920, 1080, max_bounces=4, rank=rank, nranks=nranks, frames_in_flight=infl, frames_per_launch=batch, band_rows=int(os.environ.get('BAND', '8'))) as ctx:
    ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
    for _ in range(20):
        ctx.render_frames(TRACE, steps); ctx.sync()
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter(); ctx.render_frames(TRACE, steps); ctx.sync(); ts.append(time.perf_counter() - t0)
    print(f"rank={rank}/{nranks} band {os.environ.get('BAND', '8')} {batch}x{infl}: block of {steps} frames {statistics.median(ts) * 1e3:.4f} ms (min {min(ts) * 1e3:.4f})", flush=True)
