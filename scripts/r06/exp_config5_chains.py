#!/usr/bin/env python3
"""Round 6: is a frame of config 5's scene bound by its LONGEST CHAIN?  A pixel's path is one lane's serial chain of casts (up to 17 at 8
bounces), a wave lasts as long as its slowest lane, and a frame cannot end before its longest wave.  Per view: the per-tile durations
the kernel records for the longest-first order (vxrt_debug_tile_costs: the longest wave of every 16x16 pixels, in ticks of s_memtime:
the shader clock, ~2.4 GHz on this part — the first run of this script assumed 100 MHz and printed 24 x too many ms) against the frame's kernel time: the longest chain, the 99.9th percentile, and the time a perfectly packed chip would
need for the sum (tiles' wave-durations / the wave slots of trace_kernel at 6 waves per SIMD).  usage: exp_config5_chains.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpu_voxel_raytracer_amd import TIMED, TRACE, Camera, Context, scenes  # noqa: E402

W, H = 3840, 2160
CLOCK_HZ = 2.4e9      # s_memtime ticks per second (checked by the sum: tiles' wave-durations / wave slots comes to the frame's kernel time)
with Context(W, H, max_bounces=8, frames_in_flight=1, frames_per_launch=1) as ctx:
    ctx.set_menger(*scenes.CONFIG5)
    for view, cam in scenes.config5_cameras().items():
        ctx.camera = Camera(*cam)
        ctx.render_frames(TRACE, 10)      # sorts after launches 1 and 9: the costs read below are those of launch 9's sort
        ctx.sync()
        ctx.reset_stats()
        ctx.render_frames(TRACE | TIMED, 4)
        st = ctx.stats()
        ms = st.trace_ms / st.timed_launches
        n = ((W + 15) // 16) * ((H + 15) // 16)
        cost = np.zeros(n, np.uint32)
        ctx._chk(ctx._L.vxrt_debug_tile_costs(ctx._h, cost.ctypes.data_as(C.c_void_p), C.c_size_t(n)), "tile costs")
        walk = cost[cost >= 4].astype(np.float64) / CLOCK_HZ * 1e3      # ms; costs are a running maximum over the 8 launches before the sort
        order = np.zeros((W // 8) * (H // 8), np.uint32)
        c8 = np.zeros_like(order)
        wk, sp = C.c_uint32(0), C.c_uint32(0)
        ctx._chk(ctx._L.vxrt_debug_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), c8.ctypes.data_as(C.c_void_p), C.c_size_t(order.size), C.byref(wk), C.byref(sp)), "order")
        w8 = c8[c8 >= 4].astype(np.float64) / CLOCK_HZ * 1e3           # one wave per 8x8 tile: its duration
        slots = 256 * 4 * 6
        print(f"{view:8s}: frame {ms:7.3f} ms (kernel); waves that walk {w8.size} of {c8.size}; longest wave {w8.max():7.3f} ms, 99.9th percentile {np.percentile(w8, 99.9):7.3f}, "
              f"99th {np.percentile(w8, 99):7.3f}, median {np.median(w8):7.3f}; sum of wave durations / {slots} slots = {w8.sum() / slots:7.3f} ms; "
              f"waves longer than half the frame: {(w8 > ms / 2).sum()}", flush=True)
        hist, edges = np.histogram(w8, bins=[0, 0.05, 0.1, 0.2, 0.4, 0.8, 1.2, 1.6, 2.0, 3.0, 5.0, 10.0, 100.0])
        print("          wave durations (ms) " + ", ".join(f"<{e:g}: {h}" for h, e in zip(hist, edges[1:])), flush=True)
