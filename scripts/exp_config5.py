"""BASELINE config 5 (scaled to one GPU): procedural level-7 Menger clipped to CLIP^3, 8 bounces; build + trace time."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, TIMED
clip = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
B = 8
with Context(W, H, max_bounces=B) as ctx:
    t0 = time.perf_counter()
    ctx.set_menger(7, clip, (0, 150, 170, 120), 8192)
    st = ctx.stats()
    print(f"clip {clip}: built in {time.perf_counter() - t0:.1f} s: {st.octree_nodes} nodes, depth {st.octree_depth}, scene {st.scene_bytes / 2**30:.2f} GiB", flush=True)
    ext = np.float32(clip / 2)
    for name, cam in (("outside", (np.array([-0.9, 0.6, -1.2], np.float32) * ext + ext / 2, np.array([0.9, -0.6, 1.2], np.float32), 1.2217305)),
                      ("inside a tunnel", (np.array([0.5, 0.5, 0.02], np.float32) * ext, np.array([0.05, 0.03, 1.0], np.float32), 1.2217305))):
        ctx.camera = Camera(*cam)
        for _ in range(3): ctx.render(TRACE)
        ctx.sync(); ctx.reset_stats()
        n = 10
        for _ in range(n): ctx.render(TRACE | TIMED)
        st = ctx.stats()
        t = st.trace_ms / n
        print(f"   {name}: {W}x{H} B={B}: trace {t:.3f} ms/frame, {st.rays / n / (t * 1e-3) / 1e9:.2f} Gray/s, {st.rays / n / (W * H):.2f} rays/px", flush=True)
