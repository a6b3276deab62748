"""CPU, world_size 2 over gloo: rehearses the N > 1 path of gpu_voxel_raytracer_amd/distributed.py — band
ownership, the halo message layout and its routing (who sends which rows to whom, which received buffer
feeds which side) — with the ORACLE standing in for the GPU kernels (the product itself has no CPU path).
Two spawned processes each render their bands, swap halos through torch.distributed, denoise, and rank 0
checks the stitched frame against the single-process oracle frame, bit for bit."""
import ctypes
import os
import socket

import numpy as np
import pytest

from conftest import ROOT


class OracleBandContext:
    """Stand-in with host.Context's halo / render interface; mirrors csrc/api_halo.hip + csrc/halo.hip (message layout:
    distributed.BandLayout.message_views) with the oracle as the denoiser."""

    def __init__(self, O, layout, rank, radius, full_accum, full_nd, full_alb, cam16, min_rows=1):
        self.O, self.layout, self.rank, self.r = O, layout, rank, radius
        self.rows = layout.rows(rank)
        self.imgs = [a[self.rows].copy() for a in (full_accum, full_nd, full_alb)]   # only OUR rows are known
        self.cam16 = cam16
        self.hrows = layout.halo_rows(radius, min_rows)
        self.halo = None
        self.denoised = np.zeros_like(self.imgs[0])
        self.log = []

    def halo_bytes(self):
        return self.layout.message_floats(self.hrows) * 4

    def _view(self, ptr):
        n = self.layout.message_floats(self.hrows)
        return self.layout.message_views(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), (n,)), self.hrows)

    def stream_wait_context(self, stream):
        self.log.append("stream_wait_context")

    def context_wait_stream(self, stream):
        self.log.append("context_wait_stream")

    def halo_pack(self, to_prev, to_next):
        self.log.append("pack")
        L, H = self.layout, self.hrows
        accum, nd, alb = self.imgs
        mat = ((alb[..., 3].view(np.int32) >> 24) & 0xff).view(np.float32)        # denoise.comp:67

        def put(views, slot, k, lrow):
            A, B, C = views
            A[slot, k, :, :3] = accum[lrow, :, :3]; A[slot, k, :, 3] = nd[lrow, :, 3]
            B[slot, k, :, :3] = nd[lrow, :, :3]; B[slot, k, :, 3] = mat[lrow]
            C[slot, k] = accum[lrow, :, 3]
        vp, vn = self._view(to_prev), self._view(to_next)
        for lb, gb in enumerate(L.local_bands(self.rank)):
            y0, nominal = L.band_first_row(gb), L.band_nominal_rows(gb)
            nrows = L.band_rows_here(gb)
            l0 = L.local_band_first_row(lb)
            if gb >= 1:
                for k in range(min(H, nrows)):
                    put(vp, (gb - 1) // L.nranks, k, l0 + k)
            if nrows == nominal and y0 + nominal < L.height:
                for k in range(H):
                    put(vn, (gb + 1) // L.nranks, k, l0 + nominal - H + k)

    def halo_unpack(self, from_prev, from_next):
        self.log.append("unpack")
        self.halo = tuple(tuple(v.copy() for v in self._view(p)) for p in (from_prev, from_next))

    def render_stage(self, flags):
        # rebuild, for each of our bands, the band + its halo rows, denoise that strip with the oracle; keep the rows of the
        # tile rows the flags select (DENOISE: all; DENOISE_INTERIOR / DENOISE_EDGE: distributed.BandLayout.tile_rows)
        from gpu_voxel_raytracer_amd import distributed as D
        flags &= D.DENOISE | D.DENOISE_INTERIOR | D.DENOISE_EDGE
        self.log.append({D.DENOISE: "denoise", D.DENOISE_INTERIOR: "interior", D.DENOISE_EDGE: "edge"}[flags])
        L, r, H, O = self.layout, self.r, self.hrows, self.O
        interior, edge = L.tile_rows(self.rank)
        chosen = set(interior + edge) if flags == D.DENOISE else set(interior if flags == D.DENOISE_INTERIOR else edge)
        du = O.Denoise.default()
        du.radius = r
        for lb, gb in enumerate(L.local_bands(self.rank)):
            y0, nominal = L.band_first_row(gb), L.band_nominal_rows(gb)
            nrows = L.band_rows_here(gb)
            l0 = L.local_band_first_row(lb)
            top, bot = max(y0 - r, 0), min(y0 + nrows + r, L.height)
            strip = [np.zeros((L.height, L.width, 4), np.float32) for _ in range(3)]   # full-frame canvas
            for im in range(3):
                strip[im][y0:y0 + nrows] = self.imgs[im][l0: l0 + nrows]
            if flags != D.DENOISE_INTERIOR:     # the interior launch runs BEFORE the unpack: it must not need the halo
                for side, ys in ((0, range(top, y0)), (1, range(y0 + nrows, bot))):
                    A, B, C = self.halo[side]
                    for y in ys:
                        k = y - (y0 - H) if side == 0 else y - (y0 + nominal)
                        strip[0][y, :, :3] = A[lb, k, :, :3]; strip[0][y, :, 3] = C[lb, k]
                        strip[1][y, :, :3] = B[lb, k, :, :3]; strip[1][y, :, 3] = A[lb, k, :, 3]
                        strip[2][y, :, 3] = (B[lb, k, :, 3].view(np.int32) << 24).view(np.float32)   # only the material id is read
            den = O.denoise(strip[0], strip[1], strip[2], self.cam16, du, nthreads=2)
            for t in range(nrows // 16 + (1 if nrows % 16 else 0)):
                if l0 // 16 + t in chosen:
                    lo, hi = t * 16, min(t * 16 + 16, nrows)
                    self.denoised[l0 + lo: l0 + hi] = den[y0 + lo:y0 + hi]


def _worker(rank, world, port, w, h, radius, band, overlap, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from gpu_voxel_raytracer_amd import distributed as D
    from gpu_voxel_raytracer_amd import scenes
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos, mrgb, size = scenes.load_scene("castle")
        cam = scenes.close_camera(size)
        octree = O.create_octree(pos, mrgb)
        noise = O.noise_table()
        u = O.Uniforms.default()
        u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
        u.frame_number = 1
        cam16 = u.camera16()
        layout = D.BandLayout(w, h, world, band)
        # each rank traces ONLY its rows (pixel coordinates are frame-absolute)
        mine = layout.rows(rank)
        color = np.zeros((h, w, 4), np.float32); nd = np.zeros_like(color); alb = np.zeros_like(color)
        for gb in layout.local_bands(rank):
            y0 = layout.band_first_row(gb)
            y1 = y0 + layout.band_rows_here(gb)
            c, n_, a, _ = O.trace(octree, noise, u, w, h, 3, crop=(0, y0, w, y1), nthreads=2)
            color[y0:y1], nd[y0:y1], alb[y0:y1] = c, n_, a
        accum = O.temporal(color, nd, np.zeros_like(color), np.zeros_like(nd), cam16, cam16, O.Temporal.default(), False, nthreads=2)
        ctx = OracleBandContext(O, layout, rank, radius, accum, nd, alb, cam16)
        halo = D.HaloExchange(ctx, dist, rank, world, "cpu", torch)
        D.finish_frame(ctx, world, radius, halo, overlap=overlap)
        want_log = (["pack", "stream_wait_context", "interior", "context_wait_stream", "unpack", "edge"] if overlap else
                    ["pack", "stream_wait_context", "context_wait_stream", "unpack", "denoise"])
        assert ctx.log == want_log, ctx.log
        assert ctx.halo_bytes() == layout.message_floats(layout.halo_rows(radius)) * 4 >= layout.max_bands() * layout.halo_rows(radius) * w * 36
        full = D.gather_image(ctx.denoised, layout, rank, dist, torch, "cpu")
        if rank == 0:
            # single-process reference
            c, n_, a, _ = O.trace(octree, noise, u, w, h, 3, crop=(0, 0, w, h), nthreads=2)
            acc = O.temporal(c, n_, np.zeros_like(c), np.zeros_like(n_), cam16, cam16, O.Temporal.default(), False, nthreads=2)
            du = O.Denoise.default(); du.radius = radius
            want = O.denoise(acc, n_, a, cam16, du, nthreads=2)
            same = (full == want) | (np.isnan(full) & np.isnan(want))
            q.put(("ok", bool(same.all()), int((~same).sum())))
    except Exception as e:  # pragma: no cover
        q.put(("error", repr(e), rank))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,radius,band,overlap", [(96, 72, 3, 16, True), (64, 50, 8, 16, False), (48, 150, 4, 32, True)])
def test_two_rank_gloo_halo_exchange(O, w, h, radius, band, overlap):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, w, h, radius, band, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == "ok" and res[1], res


def test_band_layout_matches_library(H):
    from gpu_voxel_raytracer_amd import distributed as D
    for (w, h, n) in ((64, 200, 8), (1920, 1080, 8), (1920, 1080, 3), (100, 16, 4), (80, 40, 5)):
        L = D.BandLayout(w, h, n, 16)
        allrows = np.concatenate([L.rows(r) for r in range(n)])
        assert sorted(allrows.tolist()) == list(range(h))
        assert all(L.owner(y) == r for r in range(n) for y in L.rows(r))
        assert max(len(L.local_bands(r)) for r in range(n)) == L.max_bands()
    L = D.BandLayout(1920, 1080, 8)
    sizes = [len(L.rows(r)) for r in range(8)]
    assert sum(sizes) == 1080 and max(sizes) - 1080 / 8 < 16               # the busiest rank (what a frame takes) within a tile row of an even share


def test_band_layout_row_rule_is_the_librarys():
    """vxrt_create takes bands that are multiples of 8 rows (the tracer's tiles); only a denoise window (radius > 0) needs 16."""
    from gpu_voxel_raytracer_amd import distributed as D
    L = D.BandLayout(1920, 1080, 8, 8, radius=0)
    assert sum(len(L.rows(r)) for r in range(8)) == 1080 and max(len(L.rows(r)) for r in range(8)) - min(len(L.rows(r)) for r in range(8)) <= 8
    for bad in ((8, None), (8, 2), (24, 8), (12, 0), (0, 0)):
        with pytest.raises(ValueError):
            D.BandLayout(1920, 1080, 8, bad[0], radius=bad[1])
    D.BandLayout(1920, 1080, 8, 32, radius=8)
    # band height for the frame loop: >= 8 r without a frame size; with one, the candidate in [48, 8 r] that leaves the busiest rank fewest rows
    assert D.band_rows_for(0) == 16 and D.band_rows_for(2) == 16 and D.band_rows_for(8) == 64 and D.band_rows_for(3) == 32
    assert D.band_rows_for(8, 2160, 8) == 64 and D.band_rows_for(8, 2160, 2) == 64 and D.band_rows_for(8, 2160, 1) == 64
    for radius, h, n in ((8, 2160, 8), (8, 2160, 4), (8, 1080, 8), (4, 2160, 3), (1, 720, 2)):
        band = D.band_rows_for(radius, h, n)
        assert band % 16 == 0 and 48 <= band <= max(64, 8 * radius)
        most = max(len(D.BandLayout(w := 64, h, n, band).rows(r)) for r in range(n))
        assert all(most <= max(len(D.BandLayout(w, h, n, other).rows(r)) for r in range(n)) for other in range(48, max(64, 8 * radius) + 1, 16))
        assert most - h / n < 16 + 1e-9                      # within one tile row of an even deal (round 3 dealt whole bands: up to a band more)


def test_band_layout_folds_the_remainder_into_a_taller_last_round():
    """Whole rounds of nranks bands at band_rows rows; the LAST round takes the remainder too, in bands just high enough (a multiple of
    the tile height): every rank within one tile row of height / nranks, every row owned once, local rows in frame order, bands
    tile-aligned, and — round 5, ADVICE r4 — no band that has a band below it is lower than band_rows, so the halo may be band_rows
    deep at EVERY band edge (round 4 dealt the remainder as an extra round of lower bands and capped the whole frame's halo at them)."""
    from gpu_voxel_raytracer_amd import distributed as D
    for h, n, band in ((2160, 8, 48), (2160, 8, 64), (2160, 4, 64), (2160, 3, 48), (200, 2, 16), (1080, 8, 8), (1080, 5, 32), (50, 2, 16), (4320, 8, 64), (37, 3, 16),
                       (304, 4, 32), (2160, 8, 16), (1080, 2, 8), (1080, 8, 4), (1080, 8, 2), (150, 8, 2)):
        L = D.BandLayout(7, h, n, band, radius=0 if band % 16 else None)
        tile = 16 if band % 16 == 0 else (8 if band % 8 == 0 else band)
        rows = [L.rows(r) for r in range(n)]
        assert sorted(np.concatenate(rows).tolist()) == list(range(h))
        counts = [len(r) for r in rows]
        assert max(counts) - h / n < tile and L.tail_rows % tile == 0 and L.full_bands % n == 0
        one_round_only = h < n * band
        assert (L.tail_rows <= band) if one_round_only else (band <= L.tail_rows < 2 * band + tile)
        for gb in range(L.bands - 1):                                    # every band but the frame's last is whole ...
            assert L.band_rows_here(gb) == L.band_nominal_rows(gb) >= (L.tail_rows if one_round_only else band)
        if L.bands > 1:                                                  # ... so the lowest of them is what a halo can carry
            assert L.halo_rows_max() == min(L.band_nominal_rows(gb) for gb in range(L.bands - 1)) >= (L.tail_rows if one_round_only else band)
        assert L.halo_rows(8, 40) == min(40, L.halo_rows_max()) and L.halo_rows(8) == min(8, L.halo_rows_max())
        for r in range(n):
            assert (np.diff(rows[r]) > 0).all()
            for lb, gb in enumerate(L.local_bands(r)):
                y0 = L.band_first_row(gb)
                assert L.owner(y0) == r and L.band_of_row(y0) == gb and L.band_of_row(y0 + L.band_rows_here(gb) - 1) == gb
                assert L.local_band_first_row(lb) % tile == 0 and rows[r][L.local_band_first_row(lb)] == y0
    L = D.BandLayout(7, 2160, 8, 48)          # 45 bands' worth: 4 whole rounds (1536 rows), then 624 rows in bands of 80
    assert (L.full_bands, L.tail_y0, L.tail_rows, L.bands) == (32, 1536, 80, 40)
    assert [len(L.rows(r)) for r in range(8)] == [272] * 7 + [256]
    L = D.BandLayout(7, 2160, 8, 64)          # config 4's layout: 3 whole rounds, then 624 rows in bands of 80 (seven of 80, one of 64)
    assert (L.full_bands, L.tail_y0, L.tail_rows, L.bands, L.halo_rows_max()) == (24, 1536, 80, 32, 64)
    assert [len(L.rows(r)) for r in range(8)] == [272] * 7 + [256]
    assert L.halo_rows(8, 40) == 40            # a 38-row pan keeps its history on every edge (round 4: capped at 16 rows)


def test_bench_ranks_agree_on_the_fall_back_before_anyone_switches():
    """if RCCL fails on some ranks only, the ones that failed must not wait in a gloo rendezvous while the others sit in an
    RCCL collective.  Every rank publishes ok / failed on a second TCP store (port chosen by the parent) and reads every rank's word.
    Here, without a GPU: the default backend cannot come up on any rank -> all agree -> gloo over that same store, and the line says
    so; with one rank claiming success (test hook) -> every rank sees the mixed outcome -> the job ends with status 3, nothing hangs."""
    import json
    out = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5"], {"VXRT_BENCH_DRY": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl"]["backend"] == "gloo" and d["rccl"]["world_size"] == 2
    assert out.stderr.count("every rank failed to bring RCCL up; falling back to gloo") == 2
    out = _bench(["--gpus", "3", "--steps", "20", "--warmup", "5"], {"VXRT_BENCH_DRY": "1", "VXRT_BENCH_FAKE_NCCL_OK_RANK": "1", "VXRT_BENCH_NCCL_TIMEOUT": "8"})
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert out.stderr.count("the ranks disagree about RCCL") == 3 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _bench(args, env):
    import subprocess
    import sys
    envd = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env={**envd, **env})


def test_bench_launcher_starts_n_ranks_and_relays_one_line():
    """`python bench.py --gpus N` without a launcher (the driver's command) starts N child ranks itself (bench.spawn_ranks) and relays
    rank 0's line — rehearsed without a GPU: VXRT_BENCH_DRY=1 makes the ranks rendezvous over gloo, all-gather who they are and leave."""
    import json
    out = _bench(["--gpus", "3", "--steps", "20", "--warmup", "5"], {"VXRT_BENCH_DRY": "1", "VXRT_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["steps"] == 20 and d["warmup"] == 5
    assert d["rccl"]["world_size"] == 3 and [e["rank"] for e in d["rccl"]["devices"]] == [0, 1, 2]
    assert len({e["pid"] for e in d["rccl"]["devices"]}) == 3      # three processes, none of them the parent


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that exits non-zero ends the job with that status: no line, nothing restarted, the other ranks stopped."""
    out = _bench(["--gpus", "2"], {"VXRT_BENCH_DRY": "1", "VXRT_BENCH_BACKEND": "gloo", "VXRT_BENCH_DRY_FAIL_RANK": "1"})
    assert out.returncode == 3 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with status 3" in out.stderr


def test_bench_parent_never_touches_the_gpu_or_execs():
    """The launcher half of bench.py imports neither torch nor the library and never replaces the process."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "spawn_ranks")
    names = {n.names[0].name for n in ast.walk(fn) if isinstance(n, ast.Import)} | {n.module for n in ast.walk(fn) if isinstance(n, ast.ImportFrom)}
    assert names <= {"signal", "subprocess", "threading"}, names
    assert not [n for n in ast.walk(tree) if isinstance(n, ast.Attribute) and (n.attr.startswith(("execv", "execl", "spawn", "posix_spawn")))]
    top = {n.names[0].name for n in tree.body if isinstance(n, ast.Import)} | {n.module for n in tree.body if isinstance(n, ast.ImportFrom)}
    assert "torch" not in top and not any(m and m.startswith("gpu_voxel") for m in top)


def test_halo_rows_for_motion_bounds_the_reprojection(H):
    """distributed.halo_rows_for_motion against brute force: for cameras a -> b, every point at distance >= near on every pixel's ray
    of frame b reprojects into frame a (temporal.comp:75-85) within the returned number of rows, less the margin, of its own row;
    a camera at rest needs the margin only; a jump larger than a band is capped at the band height."""
    from gpu_voxel_raytracer_amd import distributed as D
    w, h, fov = 320, 200, H.Camera().fov
    rng = np.random.default_rng(5)

    def axes(p, d):
        return (np.asarray(p, np.float32),) + H.Camera(p, d, fov).axis_scaled(w, h)
    p0, d0 = np.array([3.0, 2.0, -7.0], np.float32), np.array([0.1, -0.2, 1.0], np.float32)
    assert D.halo_rows_for_motion(axes(p0, d0), axes(p0, d0), w, h, 0.25, 64) == 2
    cases = [(p0 + np.float32(0.01) * np.array([1, 0.5, 0], np.float32), d0), (p0, d0 + np.array([0, -0.12, 0], np.float32)),
             (p0 + np.array([0, 0.3, 0.2], np.float32), d0 + np.array([0.05, 0.03, 0], np.float32))]
    for p1, d1 in cases:
        a, b = axes(p0, d0), axes(p1, d1)
        rows = D.halo_rows_for_motion(a, b, w, h, 0.25, 10 ** 6)
        oa, ra, ua, fa = (np.asarray(v, np.float64) for v in a)
        ob, rb, ub, fb = (np.asarray(v, np.float64) for v in b)
        inv = np.linalg.inv(np.stack([ra, ua, fa], 1))
        xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        dirs = xs[..., None] * rb - ys[..., None] * ub + fb
        dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
        worst = 0.0
        for dist in np.concatenate([[0.25, 1e8], 0.25 + rng.exponential(5.0, 6)]):
            s = (ob + dist * dirs - oa) @ inv.T
            worst = max(worst, float(np.abs(-(s[..., 1] / s[..., 2]) - ys).max()))
        assert worst <= rows - 2 + 0.51, (worst, rows)      # the 33 x 33 grid holds the borders, where the motion peaks
        assert rows <= worst + 3
        assert D.halo_rows_for_motion(a, b, w, h, 0.25, 16) == min(16, rows)
        # the C ABI's version of the same arithmetic (a C++ / Rust host has no distributed.py)
        assert H.halo_rows_for_motion(H.Camera(p0, d0, fov), H.Camera(p1, d1, fov), w, h, 0.25, 10 ** 6) == rows
        assert H.halo_rows_for_motion(H.Camera(p0, d0, fov), H.Camera(p1, d1, fov), w, h, 0.25, 16) == min(16, rows)
    assert H.halo_rows_for_motion(H.Camera(p0, d0, fov), H.Camera(p0, d0, fov), w, h, 0.25, 64) == 2


def _per_rank_worker(rank, world, port, q):
    import importlib.util
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # rank r leaves the barrier r * 0.2 ms after rank 0 and needs (1 + r) ms per block
    starts = [1_000_000_000 + b * 10_000_000 + rank * 200_000 for b in range(3)]
    ends = [s + (1 + rank) * 1_000_000 for s in starts]
    out = bench.per_rank_report(dist, world, rank, [(1 + rank) * 1e-3] * 3, starts, ends, {"local_rows": 540 - rank})
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_per_rank_report_over_two_gloo_ranks():
    """bench.py: per_rank_report — what an N > 1 line says per rank — over a real world-size-2 gloo group: rank 0 gets
    every rank's block times, the skew with which the ranks left the barrier, the finish spread and who finished last; the others None."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_per_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None
    r = got[0]
    assert r["blocks"] == 3 and [e["rank"] for e in r["ranks"]] == [0, 1]
    assert r["launch_skew_after_the_barrier_ms"] == {"median": 0.2, "max": 0.2}
    assert r["finish_spread_ms"]["median"] == pytest.approx(1.2)
    assert r["ranks"][0]["block_ms"]["median"] == 1.0 and r["ranks"][1]["block_ms"]["median"] == 2.0
    assert r["ranks"][1]["blocks_it_finished_last"] == 3 and r["ranks"][0]["blocks_it_finished_last"] == 0
    assert r["ranks"][0]["local_rows"] == 540 and r["ranks"][1]["local_rows"] == 539
