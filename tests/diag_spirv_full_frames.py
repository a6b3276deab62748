#!/usr/bin/env python3
"""ORACLE-BASED DIAGNOSTIC (test infrastructure; not collected by pytest; needs /root/reference): the reference's compiled voxels.comp
(oracle/ospirv.cpp) against the oracle at BASELINE's FULL frame sizes — the whole 1920 x 1080 frame of configs[1] (4 bounces), and
strips of config 3's (monu10, 3840 x 2160, 8 bounces) and config 4's (castle close up, 3840 x 2160, 8 bounces) frames.  Minutes of CPU
time; the summary goes to profiles/r05/spirv_exec_full_size.json.
usage: python tests/diag_spirv_full_frames.py [threads]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_voxel_raytracer_amd import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402
import spirv_pipeline as SP  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
noise = O.noise_table()
out = []
for label, scene, view, w, h, bounces, crop, frame in (
        ("configs[1]: menger 1920x1080, bench view, 4 bounces, WHOLE FRAME", "menger", "bench", 1920, 1080, 4, (0, 0, 1920, 1080), 1),
        ("config 3: monu10 3840x2160, close view, 8 bounces, rows 1000-1100", "monu10", "close", 3840, 2160, 8, (0, 1000, 3840, 1100), 2),
        ("config 4: castle 3840x2160, close view, 8 bounces, rows 1040-1120", "castle", "close", 3840, 2160, 8, (0, 1040, 3840, 1120), 3)):
    pos, mrgb, size = scenes.load_scene(scene)
    octree = O.create_octree(pos, mrgb)
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = frame
    t0 = time.time()
    ref = O.trace(octree, noise, u, w, h, bounces, crop=crop)
    t1 = time.time()
    x0, y0, x1, y1 = crop
    differing, instructions = 0, 0
    for ys in range(y0, y1, 40):                                   # in slabs, so that progress is visible
        ye = min(ys + 40, y1)
        got = SP.spirv_trace(O, octree, noise, u, w, h, crop=(x0, ys, x1, ye), nthreads=threads, bounces=bounces)
        instructions += got[3]
        for a, b in zip(got[:3], ref[:3]):
            b = b[ys - y0:ye - y0]
            bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
            differing += int(bad.sum())
        print(f"{label}: rows {ys}-{ye} done, {differing} differing values so far, {time.time() - t1:.0f} s", flush=True)
    row = {"case": label, "pixels": (x1 - x0) * (y1 - y0), "rays": int(ref[3]), "hit_pixels": int((ref[1][..., 3] >= 0).sum()),
           "values_compared": 3 * 4 * (x1 - x0) * (y1 - y0), "differing_values": differing, "spirv_instructions_interpreted": int(instructions),
           "oracle_seconds": round(t1 - t0, 2), "interpreter_seconds": round(time.time() - t1, 1), "threads": threads}
    print(row, flush=True)
    out.append(row)
with open(os.path.join(ROOT, "profiles", "r05", "spirv_exec_full_size.json"), "w") as f:
    json.dump({"what": "the reference's compiled voxels.comp.spv, interpreted (oracle/ospirv.cpp; loop bound re-specialised to the config's bounce count), "
                       "against oracle/oshaders.cpp at BASELINE's frame sizes; every output value of the three images compared bit for bit (NaN = NaN)",
               "script": "tests/diag_spirv_full_frames.py", "cases": out}, f, indent=1)
