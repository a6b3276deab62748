"""The post stages of BASELINE config 3 (vox/monu10.vox at 3840x2160) for profiling: temporal_kernel, denoise_kernel with radius 0
(fused into temporal), 2 and 8, exact and tolerant mode.  The denoise stage is re-run on the same inputs (vxrt_render(DENOISE)),
so the kernels' durations and counters can be read per (radius, mode) from the launch order:
   per radius r in (0, 2, 8): 2 warm-up frames, then N x [ALL frame] ; for r > 0 additionally N x DENOISE exact, N x DENOISE tolerant.
usage: post_stage_run.py [N]   -> prints per-stage ms from the library's own HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import ALL, DENOISE, TIMED, Camera, Context, scenes
from gpu_voxel_raytracer_amd.host import OPT_DENOISE_MODE
W, H, B = 3840, 2160, 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pos, mrgb, size = scenes.load_scene("monu10")
cam = scenes.bench_camera(size)
px = W * H
with Context(W, H, max_bounces=B) as ctx:
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = Camera(*cam)
    for radius in (0, 2, 8):
        ctx.denoise_uniforms.radius = radius
        for _ in range(2):
            ctx.render(ALL)
        ctx.sync(); ctx.reset_stats()
        for _ in range(N):
            ctx.render(ALL | TIMED)
        st = ctx.stats()
        print(f"r={radius}: trace {st.trace_ms / N:.4f} ms, temporal {st.temporal_ms / N:.4f} ms ({80 * px / (st.temporal_ms / N * 1e-3) / 1e9:.0f} GB/s of 80 B/px), "
              f"denoise (exact, in the frame) {st.denoise_ms / N:.4f} ms", flush=True)
        if radius == 0:
            continue
        for mode, label in ((0, "exact"), (1, "tolerant"), (2, "exact, generic kernel"), (3, "tolerant, generic kernel")):
            ctx.set_option(OPT_DENOISE_MODE, mode)
            ctx.update_bindings()
            ctx.render_stage(DENOISE)
            ctx.sync(); ctx.reset_stats()
            for _ in range(N):
                ctx.render_stage(DENOISE | TIMED)
            st = ctx.stats()
            t = st.denoise_ms / N
            taps = (2 * radius + 1) ** 2
            print(f"   denoise r={radius} {label}: {t:.4f} ms = {64 * px / (t * 1e-3) / 1e9:.0f} GB/s of 64 B/px, {taps * px / (t * 1e-3) / 1e12:.2f} Ttaps/s", flush=True)
        ctx.set_option(OPT_DENOISE_MODE, 0)
