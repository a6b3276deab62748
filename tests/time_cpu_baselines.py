"""ORACLE-BASED MEASUREMENT (lives under tests/ because it calls the oracle; not collected by pytest): the CPU baselines of
BASELINE.md's plan on this host's cores — config 1 (restated src/cpu.rs on vox/3x3x3.vox, 256x256, time 0) and configs 2 / 3
(restated shaders on whole frames), median of 20 timed frames after 3 warm-up frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_voxel_raytracer_amd import scenes
from oracle import oracle as O

O.build()
threads = os.cpu_count()


def median_ms(fn, warm=3, n=20):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


pos, mrgb, size = scenes.load_scene("3x3x3")
cam_pos, cam_dir, fov = scenes.bench_camera(size)
basis = O.camera_axis_scaled(cam_pos, cam_dir, fov, 256, 256)
coords, rgb = pos.astype(np.uint16), mrgb[:, 1:]
ms = median_ms(lambda: O.cpu_rs_render(coords, rgb, cam_pos * 2, basis, 256, 256, time=0.0))
print(f"config 1 (cpu.rs restated, 3x3x3 256x256, 2 rays/px max): {ms:.2f} ms/frame, 1 thread (the restatement of cpu.rs is a serial loop; the reference would spread it with rayon)")

noise = O.noise_table()
for name, w, h, b in (("menger", 1920, 1080, 4), ("monu10", 3840, 2160, 8)):
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam = scenes.bench_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    state = {"f": 0, "rays": 0}

    def frame():
        state["f"] += 1
        u.frame_number = state["f"]
        state["rays"] = O.trace(octree, noise, u, w, h, b, crop=(0, 0, w, h), nthreads=threads)[3]
    ms = median_ms(frame)
    print(f"{name} {w}x{h} {b} bounces (voxels.comp restated): {ms:.1f} ms/frame, {state['rays'] / ms / 1e3:.1f} Mrays/s, {threads} threads")
