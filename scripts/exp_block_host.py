"""Where the host's share of a short block goes: one rank of 8 (4-row bands, the all-in-one kernel, 20 frames in one launch), with and
without the TIMED flag (two HIP events around the launch), the time until render_frames returns, and the floor of one launch + one
synchronisation on this stack (a 64 x 64 frame).  usage: python scripts/exp_block_host.py [rank] [blocks]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, TIMED, scenes
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)


def run(ctx, flags, steps, label):
    for _ in range(20):
        ctx.render_frames(flags, steps); ctx.sync()
    sub, tot = [], []
    ctx.reset_stats()
    for _ in range(blocks):
        t0 = time.perf_counter(); ctx.render_frames(flags, steps); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        sub.append(t1 - t0); tot.append(t2 - t0)
    st = ctx.stats()
    k = st.trace_ms / st.timed_launches * 1e3 if st.timed_launches else float("nan")
    print(f"{label}: block {statistics.median(tot) * 1e6:.1f} us (min {min(tot) * 1e6:.1f}), render_frames returns after {statistics.median(sub) * 1e6:.1f} us, "
          f"kernel by events {k:.1f} us", flush=True)


with Context(1920, 1080, max_bounces=4, rank=rank, nranks=8, frames_in_flight=1, frames_per_launch=20, band_rows=4, tracer=1) as ctx:
    ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
    run(ctx, TRACE, 20, f"rank {rank}/8 20x1 all-in-one, untimed")
    run(ctx, TRACE | TIMED, 20, f"rank {rank}/8 20x1 all-in-one, TIMED  ")
    run(ctx, TRACE, 20, f"rank {rank}/8 20x1 all-in-one, untimed")
with Context(64, 64, max_bounces=4, frames_in_flight=1, frames_per_launch=1, tracer=1) as ctx:
    ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
    run(ctx, TRACE, 1, "64x64 one frame, untimed")
    run(ctx, TRACE | TIMED, 1, "64x64 one frame, TIMED  ")
