#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 1c: BASELINE config 5's scene (5.6 GB, HBM-resident) at 3840x2160, 8 bounces, one frame per launch, with the
tiles that walk dealt to the XCDs by screen region (VXRT_OPT_XCD_AFFINITY S, super-tiles of S x S tiles) instead of round robin.
Per view and S: kernel ms per frame (HIP events around the launches: the host-made order's own stall is not in it), the balance of the
eight lists, and whether the frame equals the frame of S = 0 bit for bit.  usage: exp_config5_affinity.py [S ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpu_voxel_raytracer_amd import SAMPLED_COLOR, TIMED, TRACE, Camera, Context, host, scenes  # noqa: E402

host.use_library(host.variants_library())      # the option is an experiment: the product refuses it
sizes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 32]
with Context(3840, 2160, max_bounces=8, frames_in_flight=1, frames_per_launch=1) as ctx:
    ctx.set_menger(*scenes.CONFIG5)
    for view, cam in scenes.config5_cameras().items():
        ctx.camera = Camera(*cam)
        ref = None
        for S in sizes:
            ctx.set_option(host.OPT_XCD_AFFINITY, S)
            ctx.render_frames(TRACE, 11)            # the order is re-made after the next launch, and once more 8 launches later
            ctx.sync()
            ctx.reset_stats()
            ctx.render_frames(TRACE | TIMED, 6)     # launches 12 .. 17: no sort in between (every 8th)
            st = ctx.stats()
            ms = st.trace_ms / max(st.timed_launches, 1)
            ctx.set_frame_number(1000)
            ctx.render_frames(TRACE, 1)
            img = ctx.read(SAMPLED_COLOR)
            same = "-" if ref is None else ("identical" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT")
            if ref is None:
                ref = img
            print(f"{view:8s} S = {S:2d}: {ms:8.3f} ms per frame (kernel), {st.rays / max(st.frames, 1) / 1e6:6.2f} Mrays per frame; frame vs S = {sizes[0]}: {same}", flush=True)
