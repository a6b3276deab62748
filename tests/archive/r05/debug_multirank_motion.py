"""Diagnostic (not collected): 2 contexts in one process, slow moving camera; where does the stitched accumulated image differ?"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_voxel_raytracer_amd import ALL, DENOISE, TEMPORAL, TRACE, Camera, Context, scenes
from test_gpu_bands import hip
w, h, bounces, radius, band, nranks = 320, 200, 3, 3, 16, 2
pos, mrgb, size = scenes.load_scene("castle")
p0, d0, fov = scenes.close_camera(size)
path = [(p0 + np.float32(0.01 * k) * np.array([1, 0.5, 0], np.float32), d0) for k in range(4)]
rt = hip()
single = Context(w, h, max_bounces=bounces)
ctxs = [Context(w, h, max_bounces=bounces, rank=r, nranks=nranks, band_rows=band) for r in range(nranks)]
for c in [single] + ctxs:
    c.recreate_octree(pos, mrgb); c.denoise_uniforms.radius = radius
rows = [c.local_rows() for c in ctxs]
for k, (cp, cd) in enumerate(path):
    for c in [single] + ctxs:
        c.camera = Camera(cp, cd, fov)
    single.render(ALL)
    for c in ctxs:
        c.render(TRACE | TEMPORAL)
    nbytes = ctxs[0].halo_bytes()
    bufs = {}
    for r, c in enumerate(ctxs):
        p, n = C.c_void_p(), C.c_void_p()
        rt.hipMalloc(C.byref(p), nbytes); rt.hipMalloc(C.byref(n), nbytes)
        c.halo_export(p.value, n.value); bufs[r] = (p, n)
    for r, c in enumerate(ctxs):
        c.halo_import(bufs[(r - 1) % nranks][1].value, bufs[(r + 1) % nranks][0].value)
        c.render_stage(DENOISE)
    for c in ctxs: c.sync()
    want = single.read(3); got = np.zeros_like(want)
    for c, rr in zip(ctxs, rows): got[rr] = c.read(3)
    diff = ~((got == want) | (np.isnan(got) & np.isnan(want))).all(-1)
    ys, xs = np.nonzero(diff)
    print(f"frame {k+1}: {diff.sum()} accum pixels differ; rows mod 16: {sorted(set((ys % 16).tolist()))}; first: {list(zip(ys[:6].tolist(), xs[:6].tolist()))}")
    for y, x in list(zip(ys[:4].tolist(), xs[:4].tolist())):
        print("   ", y, x, "got", got[y, x], "want", want[y, x], "sampled", single.read(0)[y, x], "nd", single.read(1)[y, x])

# ---- second experiment: are the exported halo buffers what the single context holds? -------------------------------------------
print("halo buffer check")
single2 = Context(w, h, max_bounces=bounces)
ctxs2 = [Context(w, h, max_bounces=bounces, rank=r, nranks=nranks, band_rows=band) for r in range(nranks)]
for c in [single2] + ctxs2:
    c.recreate_octree(pos, mrgb); c.denoise_uniforms.radius = radius; c.camera = Camera(p0, d0, fov)
single2.render(TRACE | TEMPORAL)
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for r, c in enumerate(ctxs2):
    c.render(TRACE | TEMPORAL)
    nbytes = c.halo_bytes()
    p, n = C.c_void_p(), C.c_void_p()
    rt.hipMalloc(C.byref(p), nbytes); rt.hipMalloc(C.byref(n), nbytes)
    rt.hipMemset(p, 0xff, nbytes); rt.hipMemset(n, 0xff, nbytes); rt.hipDeviceSynchronize()
    c.halo_export(p.value, n.value)
    hp = np.zeros(nbytes // 4, np.float32); hn = np.zeros(nbytes // 4, np.float32)
    rt.hipMemcpy(hp.ctypes.data_as(C.c_void_p), p, nbytes, 2); rt.hipMemcpy(hn.ctypes.data_as(C.c_void_p), n, nbytes, 2)
    mb = (((h + band - 1) // band) + nranks - 1) // nranks
    hp = hp.reshape(mb, radius, 3, w, 4); hn = hn.reshape(mb, radius, 3, w, 4)
    imgs = [single2.read(3), single2.read(1), single2.read(2)]
    bad = 0
    for gb in range(r, (h + band - 1) // band, nranks):
        y0 = gb * band
        if gb >= 1:
            j = (gb - 1) // nranks
            for k in range(radius):
                for im in range(3):
                    if y0 + k < h and not np.array_equal(hp[j, k, im], imgs[im][y0 + k], equal_nan=True):
                        bad += 1; print("  to_prev mismatch rank", r, "band", gb, "k", k, "im", im, np.nonzero((hp[j,k,im] != imgs[im][y0+k]).any(-1))[0][:8])
        if y0 + band < h:
            j = (gb + 1) // nranks
            for k in range(radius):
                for im in range(3):
                    if not np.array_equal(hn[j, k, im], imgs[im][y0 + band - radius + k], equal_nan=True):
                        bad += 1; print("  to_next mismatch rank", r, "band", gb, "k", k, "im", im, np.nonzero((hn[j,k,im] != imgs[im][y0+band-radius+k]).any(-1))[0][:8])
    print("rank", r, "mismatching rows:", bad)
# geometry mask at row 160 in frame 2 of the first experiment
nd = single.read(1)
print("row 160 geometry x:", np.nonzero(nd[160, :, 3] >= 0)[0][:20], "count", (nd[160, :, 3] >= 0).sum())
