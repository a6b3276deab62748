"""Config-3 trace stage (monu10 4K, 8 bounces), throughput with 4 frames in flight, per tracer variant."""
import sys, os, time
os.environ.setdefault("VXRT_ENV_KNOBS", "1")   # host.py: translate the VXRT_* knobs into create-time options
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
W, H, B = 3840, 2160, 8
pos, mrgb, size = scenes.load_scene("monu10")
for view in ("bench", "close"):
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    with Context(W, H, max_bounces=B, frames_in_flight=4) as ctx:
        ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
        ctx.render_frames(TRACE, 10); ctx.sync(); ctx.reset_stats()
        n = 60
        t0 = time.perf_counter(); ctx.render_frames(TRACE, n); ctx.sync(); dt = (time.perf_counter() - t0) / n
        print(f"variant {os.environ.get('VXRT_TRACE_VARIANT', '0')} tail {os.environ.get('VXRT_TAIL_FROM', '-')}/{os.environ.get('VXRT_TAIL_SPLIT', '-')} {view}: {dt * 1e3:.3f} ms/frame, {ctx.stats().rays / n / dt / 1e9:.2f} Gray/s")
