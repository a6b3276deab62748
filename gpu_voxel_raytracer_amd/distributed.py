"""Multi-GPU orchestration of the vxrt frame loop: one process per GPU, torch.distributed (backend "nccl"
= RCCL over xGMI on ROCm, "gloo" for CPU rehearsals) for the only exchange the path has — the halo.

Decomposition: the frame's rows are dealt to the ranks in interleaved bands of `band_rows`
rows (band b -> rank b % nranks); scene and noise table are replicated; the trace stage needs no
communication.  denoise.comp's (2r+1)^2 window (shaders/denoise.comp:51-57) reaches r rows into the
neighbouring bands, and temporal.comp's reprojection (shaders/temporal.comp:85-113) reads the previous
frame's history at a texel that may lie in them; both live on rank-1 and rank+1 (mod nranks).  After a
frame's temporal stage each rank therefore sends two messages and receives two, point-to-point, each on its
own xGMI link — `rows` = max(r, VXRT_OPT_HALO_ROWS) rows per band edge, 36 bytes per pixel (include/vxrt.h
"halo").  No all-reduce anywhere.  The exchange is overlapped with the denoise of the tiles that need no
halo:

    render(TRACE | TEMPORAL) -> halo.start() -> render_stage(DENOISE_INTERIOR) -> halo.finish() -> render_stage(DENOISE_EDGE)

`ctx` is anything with halo_bytes() / halo_pack(ptr, ptr) / halo_unpack(ptr, ptr) / stream_wait_context(stream) /
context_wait_stream(stream) / render_stage(flags) — host.Context on a GPU; the CPU tests substitute an
oracle-backed stand-in to rehearse the routing and the message layout.
"""
import numpy as np

TRACE, TEMPORAL, DENOISE, DENOISE_INTERIOR, DENOISE_EDGE = 1, 2, 4, 16, 32
HALO_BYTES_PER_PIXEL = 36


def band_rows_for(radius, height=None, nranks=None, minimum=16):
    """Band height for the frame loop with a denoise window of `radius` (a multiple of 16: the denoise tiles; 16 without a window).
    Without a frame size: >= 8 radius, so that the halo is at most a quarter of the rows a rank owns.  With
    `height` and `nranks`: the candidate between 48 rows (three rows of tiles: one of them needs no neighbour, so the exchange has
    work to hide behind) and 8 radius that leaves the busiest rank the fewest rows; ties go to the taller band (less halo).  Since
    the last round of bands takes the remainder (BandLayout: the busiest rank within one tile row of height / nranks) the candidates
    differ by a tile row at most, and the tall band usually wins: 64 rows at radius 8 for 2160 rows on 8 ranks (272 rows on the
    busiest rank, 270 would be even; round 3's whole-band deal: 288 with 48-row bands, 320 with 64)."""
    if radius <= 0:
        return minimum
    tall = max(minimum, (8 * radius + 15) // 16 * 16)
    if height is None or nranks is None or nranks < 2:
        return tall
    best = None
    for band in range(max(48, (4 * radius + 15) // 16 * 16), max(tall, 48) + 1, 16):
        layout = BandLayout(1, height, nranks, band)
        most = max(len(layout.rows(rank)) for rank in range(nranks))
        if best is None or most <= best[0]:
            best = (most, band)
    return best[1]


class BandLayout:
    """Row ownership and halo message layout; mirrors BandMap (csrc/kernels.h) and csrc/api_halo.hip / csrc/halo_view.h.
    Band gb -> rank gb % nranks.  Whole rounds of nranks bands are band_rows rows high (`full_bands` of them, rows [0, tail_y0)); when
    the frame is not a whole number of rounds the LAST round takes the remainder too, in bands `tail_rows` high — the smallest multiple
    of the tile height (16, or 8 for 8-row bands) that covers it in nranks bands, band_rows <= tail_rows < 2 band_rows + tile — so the
    BUSIEST rank owns within one tile row of height / nranks rows (the last ranks of that taller round, clipped by the frame's edge, may
    own up to tail_rows fewer: 1080 rows, 8 ranks, 64-row bands give 144 x 7 + 72) and no band but the frame's last is lower than band_rows (a frame lower than
    one round has only that round, in bands lower than band_rows)."""

    def __init__(self, width, height, nranks, band_rows=16, radius=None):
        # the library's rule (vxrt_create / check_render): bands are multiples of the tracer's 8-row tiles;
        # the denoise stage with a window (radius > 0) works on 16x16 tiles that must not straddle bands
        if band_rows <= 0 or (band_rows % 8 and band_rows not in (2, 4)):
            raise ValueError("band_rows must be 2, 4 or a multiple of 8")
        if band_rows % 16 and (radius is None or radius > 0):
            raise ValueError("band_rows must be a multiple of 16 for a denoise radius > 0 (pass radius=0 for 8-row bands)")
        self.width, self.height, self.nranks, self.band_rows = width, height, nranks, band_rows
        tile = 16 if band_rows % 16 == 0 else (8 if band_rows % 8 == 0 else band_rows)
        rounds = height // (nranks * band_rows)
        if height % (nranks * band_rows) and rounds > 0:
            rounds -= 1                     # the last whole round takes the remainder as well (taller bands)
        self.full_bands = rounds * nranks
        self.tail_y0 = self.full_bands * band_rows
        rest = height - self.tail_y0
        self.tail_rows = band_rows if rest == 0 else ((rest + nranks - 1) // nranks + tile - 1) // tile * tile
        self.bands = self.full_bands + (rest + self.tail_rows - 1) // self.tail_rows

    def band_of_row(self, y):
        return y // self.band_rows if y < self.tail_y0 else self.full_bands + (y - self.tail_y0) // self.tail_rows

    def band_first_row(self, gb):
        return gb * self.band_rows if gb < self.full_bands else self.tail_y0 + (gb - self.full_bands) * self.tail_rows

    def band_nominal_rows(self, gb):
        return self.band_rows if gb < self.full_bands else self.tail_rows

    def band_rows_here(self, gb):
        """Rows of band gb inside the frame."""
        return min(self.band_nominal_rows(gb), self.height - self.band_first_row(gb))

    def local_band_first_row(self, lb):
        """First LOCAL row of a rank's lb-th band."""
        r = self.full_bands // self.nranks
        return lb * self.band_rows if lb < r else r * self.band_rows

    def owner(self, y):
        return self.band_of_row(y) % self.nranks

    def rows(self, rank):
        y = np.arange(self.height)
        band = np.where(y < self.tail_y0, y // self.band_rows, self.full_bands + (y - self.tail_y0) // self.tail_rows)
        return y[band % self.nranks == rank]

    def local_bands(self, rank):
        return list(range(rank, self.bands, self.nranks))

    def max_bands(self):
        return (self.bands + self.nranks - 1) // self.nranks

    def halo_rows_max(self):
        """The most rows per band edge the layout can carry (vxrt_halo_info.max_rows): the lowest band that has a band below it —
        band_rows, unless the frame is lower than one round of bands.  The cap to pass to halo_rows_for_motion."""
        if self.nranks < 2:
            return 0
        if self.tail_y0 < self.height:
            return min(self.band_rows, self.tail_rows) if self.full_bands > 0 else self.tail_rows
        return self.band_rows

    def halo_rows(self, radius, min_rows=1):
        """Rows per band edge an exchange carries (vxrt_halo_info.rows): at most halo_rows_max()."""
        if self.nranks < 2:
            return 0
        return min(self.halo_rows_max(), max(radius, min_rows))

    def plane(self, rows):
        """float4 per A / B plane of a message."""
        return self.max_bands() * rows * self.width

    def message_floats(self, rows):
        """float32 count of one halo message: planes A and B (a float4 per pixel) and C (a float per pixel), rounded up
        to whole 256-byte lines (vxrt_halo_info.message_bytes / 4)."""
        plane = self.plane(rows)
        f4 = 2 * plane + (plane + 3) // 4
        return (f4 + 15) // 16 * 16 * 4

    def message_views(self, buf, rows):
        """numpy views into a message (a float32 array of message_floats(rows)): A[slot, row, x, 4] = (r, g, b, depth),
        B[slot, row, x, 4] = (normal, bits(material id)), C[slot, row, x] = blending factor."""
        plane, shape = self.plane(rows), (self.max_bands(), rows, self.width)
        a = buf[:plane * 4].reshape(shape + (4,))
        b = buf[plane * 4: plane * 8].reshape(shape + (4,))
        c = buf[plane * 8: plane * 9].reshape(shape)
        return a, b, c

    def halo_pixels_per_rank(self, rank, rows):
        """Pixels this rank SENDS per exchange (both neighbours)."""
        n = 0
        for gb in self.local_bands(rank):
            y0, here, nominal = self.band_first_row(gb), self.band_rows_here(gb), self.band_nominal_rows(gb)
            if gb >= 1:
                n += min(rows, here)
            if here == nominal and y0 + nominal < self.height:
                n += rows
        return n * self.width

    def tile_rows(self, rank):
        """(interior, edge): the rank's rows of 16x16 denoise tiles (index = local row // 16) whose window stays inside the rank's own
        rows, and those that read rows of a neighbour — a band's first tile row when a band lies above it, its last when one lies
        below (csrc/api_halo.hip: build_tile_rows)."""
        interior, edge = [], []
        if self.band_rows % 16:
            return interior, edge
        for lb, gb in enumerate(self.local_bands(rank)):
            y0 = self.band_first_row(gb)
            end = y0 + self.band_rows_here(gb)
            t = self.local_band_first_row(lb) // 16
            for ty in range(y0, end, 16):
                above = self.nranks > 1 and ty == y0 and y0 > 0
                below = self.nranks > 1 and min(ty + 16, end) == end and end < self.height
                (edge if above or below else interior).append(t + (ty - y0) // 16)
        return interior, edge

    def neighbours(self, rank):
        return (rank - 1) % self.nranks, (rank + 1) % self.nranks


class HaloExchange:
    """The halo exchange of one rank, in two halves so that the caller can put work between them:

        start()   one pack kernel fills to_prev / to_next (on the context's stream); the communication stream waits for it with
                  an event; over RCCL the two sends and two receives are posted (asynchronous); nothing waits on the host.
        finish()  the communication stream waits for the four transfers, the context's stream waits for it (event), one unpack
                  kernel moves both received messages into the context's halo store.

    Messages: to_prev (tag 0) and to_next (tag 1); with nranks == 2 both go to the same peer.  The four message buffers live
    as long as this object; a frame's receives are ordered after the previous frame's unpack through the same events.
    comm_device "cpu" (a gloo rehearsal of several ranks on one GPU) stages the messages through pinned host memory; its
    finish() is where the host blocks (copy out, gloo send/recv, copy in)."""

    def __init__(self, ctx, dist, rank, nranks, device, torch, comm_device=None):
        """device: where the context's halo buffers live (the rank's GPU; "cpu" for the oracle-backed stand-in of the CPU tests).
        comm_device: where the messages travel — the same device over RCCL (default); "cpu" stages them through host memory."""
        self.ctx, self.dist, self.rank, self.nranks, self.device, self.torch = ctx, dist, rank, nranks, device, torch
        self.comm_device = device if comm_device is None else comm_device
        self.on_gpu = str(device) != "cpu"
        self.nfloats, self.bufs, self.staged, self.works, self.copied = 0, None, None, None, None
        self.stream = torch.cuda.Stream(device) if self.on_gpu else None
        self.exchanges = 0

    def _handle(self):
        return self.stream.cuda_stream if self.stream is not None else 0

    def _buffers(self):
        n = self.ctx.halo_bytes() // 4
        if self.bufs is None or n != self.nfloats:
            torch = self.torch
            if self.on_gpu and self.bufs is not None:
                torch.cuda.synchronize()       # a launch may still be reading the old buffers (the layout changes rarely)
            self.nfloats = n
            self.bufs = [torch.zeros(max(n, 1), dtype=torch.float32, device=self.device) for _ in range(4)]
            self.staged = None
            if str(self.comm_device) != str(self.device):
                self.staged = [torch.zeros(max(n, 1), dtype=torch.float32, device=self.comm_device, pin_memory=self.on_gpu) for _ in range(4)]
        return self.bufs

    def _ops(self, send_prev, send_next, recv_prev, recv_next):
        dist, rank, nranks = self.dist, self.rank, self.nranks
        prev, nxt = (rank - 1) % nranks, (rank + 1) % nranks
        return [dist.P2POp(dist.isend, send_prev, prev, tag=0), dist.P2POp(dist.isend, send_next, nxt, tag=1),
                dist.P2POp(dist.irecv, recv_next, nxt, tag=0),    # what the next rank addressed to ITS prev (me)
                dist.P2POp(dist.irecv, recv_prev, prev, tag=1)]   # what the previous rank addressed to ITS next (me)

    def start(self):
        to_prev, to_next, from_prev, from_next = self._buffers()
        self.works, self.copied = None, None
        if self.nranks < 2 or self.nfloats == 0:
            return
        torch = self.torch
        self.ctx.halo_pack(to_prev.data_ptr(), to_next.data_ptr())
        self.ctx.stream_wait_context(self._handle())     # also orders this frame's receives after the last frame's unpack
        if self.staged is None:
            if self.on_gpu:
                with torch.cuda.stream(self.stream):
                    self.works = self.dist.batch_isend_irecv(self._ops(to_prev, to_next, from_prev, from_next))
            else:
                self.works = self.dist.batch_isend_irecv(self._ops(to_prev, to_next, from_prev, from_next))
        else:   # copies to pinned host memory, asynchronous; finish() waits for them
            with torch.cuda.stream(self.stream):
                self.staged[0].copy_(to_prev, non_blocking=True)
                self.staged[1].copy_(to_next, non_blocking=True)
                self.copied = torch.cuda.Event()
                self.copied.record(self.stream)

    def finish(self):
        to_prev, to_next, from_prev, from_next = self.bufs
        if self.nranks < 2 or self.nfloats == 0:
            return
        torch = self.torch
        if self.staged is None:
            if self.on_gpu:
                with torch.cuda.stream(self.stream):
                    for w in self.works:
                        w.wait()                 # RCCL: the communication stream waits, not the host
            else:
                for w in self.works:
                    w.wait()
        else:
            s_prev, s_next, r_prev, r_next = self.staged
            self.copied.synchronize()
            for w in self.dist.batch_isend_irecv(self._ops(s_prev, s_next, r_prev, r_next)):
                w.wait()
            with torch.cuda.stream(self.stream):
                from_prev.copy_(r_prev, non_blocking=True)
                from_next.copy_(r_next, non_blocking=True)
        self.ctx.context_wait_stream(self._handle())
        self.ctx.halo_unpack(from_prev.data_ptr(), from_next.data_ptr())
        self.exchanges += 1

    def exchange(self):
        self.start()
        self.finish()


def exchange_halo(ctx, dist, rank, nranks, device, torch):
    """One exchange with buffers of its own (see HaloExchange, which keeps them across frames)."""
    HaloExchange(ctx, dist, rank, nranks, device, torch).exchange()


OPT_HALO_ROWS = 4      # vxrt_option VXRT_OPT_HALO_ROWS (include/vxrt.h)


def halo_rows_for_motion(cam_a, cam_b, width, height, near, band_rows, margin=2):
    """Rows of the neighbours' history a rank must see so that temporal.comp's reprojection (shaders/temporal.comp:75-113) from the
    frame of camera `cam_b` into the frame of camera `cam_a` stays inside what it has: the largest vertical image motion, in rows,
    of a point at distance >= `near` along any pixel's ray, + `margin` (the bilinear footprint's second row and rounding), capped
    at band_rows — pass BandLayout.halo_rows_max() / vxrt_halo_info.max_rows, what the layout can carry; a result equal to the cap means
    every row of the neighbours travels and a faster motion reaches rows of a third rank (a disocclusion there).  A camera is (origin, right, up, forward) as Camera.axis_scaled gives
    them (pixel ray = x right - y up + forward, shaders/voxels.comp:299-303).  Along a pixel's ray the reprojected row is a
    linear-fractional function of 1 / distance, so its extremes over [near, inf) are at the two ends; over the screen the motion is
    evaluated on a 33 x 33 grid of pixels including the borders.  Every rank computes the same number from the same cameras — the
    message sizes of an exchange must agree.  This is what the exchange AFTER frame a must carry for frame b, so a frame loop sets
    it one frame ahead (a camera path), or from the last motion with a margin of its own (interactive)."""
    oa, ra, ua, fa = (np.asarray(v, np.float64) for v in cam_a)
    ob, rb, ub, fb = (np.asarray(v, np.float64) for v in cam_b)
    try:
        inv = np.linalg.inv(np.stack([ra, ua, fa], axis=1))
    except np.linalg.LinAlgError:
        return int(band_rows)
    xs, ys = np.meshgrid(np.linspace(0, width - 1, 33), np.linspace(0, height - 1, 33))
    d = xs[..., None] * rb - ys[..., None] * ub + fb
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    worst = 0.0
    for dist in (float(near), 1e9):
        s = (ob + dist * d - oa) @ inv.T
        ok = s[..., 2] > 1e-12
        if not ok.all():
            return int(band_rows)              # a point behind the old camera: no bound
        rows = np.abs(-(s[..., 1] / s[..., 2]) - ys)
        worst = max(worst, float(rows.max()))
    if not np.isfinite(worst):
        return int(band_rows)
    return int(min(band_rows, np.ceil(worst - 1e-6) + margin))


def finish_frame(ctx, nranks, radius, halo, overlap=True, extra_flags=0):
    """What follows a frame's TRACE | TEMPORAL on a rank: the halo exchange and the denoise stage, overlapped.
    radius 0: the denoise stage has no window (render_frame fuses it into the temporal pass; a caller that comes here with the
    stage still to do gets the pass-through between the two halves of the exchange); the exchange still runs — the next frame's
    temporal stage reads the neighbours' history rows.  extra_flags: e.g. TIMED."""
    if nranks < 2:
        ctx.render_stage(DENOISE | extra_flags)
        return
    if radius == 0:
        halo.start()
        ctx.render_stage(DENOISE | extra_flags)
        halo.finish()
    elif overlap:
        halo.start()
        ctx.render_stage(DENOISE_INTERIOR | extra_flags)
        halo.finish()
        ctx.render_stage(DENOISE_EDGE | extra_flags)
    else:
        halo.exchange()
        ctx.render_stage(DENOISE | extra_flags)


def render_frame(ctx, dist, rank, nranks, device, torch, radius, halo=None, overlap=True, halo_rows=None):
    """One frame of Context::render (src/context.rs:2004-2075) on a rank: trace -> temporal -> [halo] -> denoise.
    halo: a HaloExchange to re-use across frames (one is made for the call otherwise).
    halo_rows: VXRT_OPT_HALO_ROWS for THIS frame's exchange, i.e. what the NEXT frame's reprojection may reach
    (halo_rows_for_motion(this frame's camera, the next frame's, ...)); None leaves the option as it is (default 1 row: a camera at
    rest or drifting less than a row per frame)."""
    if nranks > 1 and halo is None:
        halo = HaloExchange(ctx, dist, rank, nranks, device, torch)
    if nranks > 1 and halo_rows is not None:
        ctx.set_option(OPT_HALO_ROWS, int(halo_rows))
    if radius == 0:
        # no window: the denoise stage is a per-pixel pass that the library fuses into the temporal kernel when both are asked for
        # together (32 B/px read and 16 B/px written less than a pass of its own), and the exchange follows
        ctx.render(TRACE | TEMPORAL | DENOISE)
        if nranks > 1:
            halo.exchange()
        return
    ctx.render(TRACE | TEMPORAL)
    finish_frame(ctx, nranks, radius, halo, overlap)


def gather_image(local_rows_img, layout, rank, dist, torch, device):
    """Collect every rank's rows on rank 0 (rows x width x 4 float32); returns the full frame on rank 0."""
    t = torch.from_numpy(np.ascontiguousarray(local_rows_img)).to(device)
    if dist is None or layout.nranks == 1:
        return local_rows_img
    if rank == 0:
        full = np.zeros((layout.height, layout.width, 4), np.float32)
        full[layout.rows(0)] = local_rows_img
        for r in range(1, layout.nranks):
            rows = layout.rows(r)
            buf = torch.empty((len(rows), layout.width, 4), dtype=torch.float32, device=device)
            if len(rows):
                dist.recv(buf, src=r)
                full[rows] = buf.cpu().numpy()
        return full
    if t.numel():
        dist.send(t, dst=0)
    return None
