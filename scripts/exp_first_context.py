"""Is the first context of a process slower than later ones (seen in exp_rank_emulation.py at 8 ranks)?  One rank's band set, timed in
several passes inside one context, then again in further contexts.  It is, when its trace streams share a hardware queue:
INFL=4 (5 streams) in the first context, or INFL=3 with PRE_STREAMS=1..2 foreign streams created before it, at the default
GPU_MAX_HW_QUEUES=4 (0.0236-0.0271 vs 0.0173-0.0181 ms per frame); INFL=3 with GPU_MAX_HW_QUEUES=8 is fast in every arrangement.
env: INFL, BATCH, BAND, NRANKS, RANK, PRE_STREAMS, CONTEXTS, GPU_MAX_HW_QUEUES (host.py defaults it to 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_voxel_raytracer_amd import Context, Camera, TRACE, scenes
W, H = 1920, 1080
pos, mrgb, size = scenes.load_scene("menger")
cam = scenes.bench_camera(size)
infl, batch, band = int(os.environ.get("INFL", "4")), int(os.environ.get("BATCH", "32")), int(os.environ.get("BAND", "8"))
nranks, rank = int(os.environ.get("NRANKS", "8")), int(os.environ.get("RANK", "3"))
# other streams of the process that exist before the context does (RCCL's, torch's): they take hardware-queue slots too
pre = int(os.environ.get("PRE_STREAMS", "0"))
if pre:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    keep = []
    for _ in range(pre):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), 1 << 20) == 0
        assert hip.hipMemsetAsync(buf, 0, 1 << 20, st) == 0      # something runs on it, so the queue exists
        assert hip.hipStreamSynchronize(st) == 0
        keep.append((st, buf))
for rep in range(int(os.environ.get("CONTEXTS", "3"))):
    with Context(W, H, max_bounces=4, rank=rank, nranks=nranks, frames_in_flight=infl, frames_per_launch=batch, band_rows=band) as ctx:
        ctx.recreate_octree(pos, mrgb); ctx.camera = Camera(*cam)
        for p in range(4):
            n = max(20 * batch * infl // (8 // nranks), batch * infl)
            t0 = time.perf_counter(); ctx.render_frames(TRACE, n); ctx.sync(); dt = (time.perf_counter() - t0) / n
            print(f"context {rep} pass {p}: {dt * 1e3:.4f} ms/frame", flush=True)
