"""Multi-GPU orchestration of the vxrt frame loop: one process per GPU, torch.distributed (backend "nccl"
= RCCL over xGMI on ROCm, "gloo" for CPU rehearsals) for the only exchange the path has — the denoise halo.

Decomposition: the frame's rows are dealt to the ranks in interleaved bands of `band_rows`
rows (band b -> rank b % nranks); scene and noise table are replicated; trace and temporal need no
communication (temporal history is same-pixel for a static camera; a reprojection that leaves the rank's
rows is treated as a disocclusion).  denoise.comp's (2r+1)^2 window reaches r rows into the neighbouring
bands, which live on rank-1 and rank+1 (mod nranks): each rank sends two messages and receives two,
point-to-point, each on its own xGMI link.  No all-reduce anywhere.

`ctx` is anything with halo_bytes() / halo_export(ptr, ptr) / halo_import(ptr, ptr) / render_stage(flags)
— host.Context on a GPU; the CPU tests substitute an oracle-backed stand-in to rehearse the routing.
"""
import numpy as np

TRACE, TEMPORAL, DENOISE = 1, 2, 4


class BandLayout:
    """Row ownership and halo message layout; mirrors BandMap / vxrt_halo_* in csrc/vxrt_api.hip."""

    def __init__(self, width, height, nranks, band_rows=16, radius=None):
        # the library's rule (vxrt_create / check_render in csrc/vxrt_api.hip): bands are multiples of the tracer's 8-row tiles;
        # the denoise stage with a window (radius > 0) works on 16x16 tiles that must not straddle bands
        if band_rows <= 0 or band_rows % 8:
            raise ValueError("band_rows must be a multiple of 8")
        if band_rows % 16 and (radius is None or radius > 0):
            raise ValueError("band_rows must be a multiple of 16 for a denoise radius > 0 (pass radius=0 for 8-row bands)")
        self.width, self.height, self.nranks, self.band_rows = width, height, nranks, band_rows
        self.bands = (height + band_rows - 1) // band_rows

    def owner(self, y):
        return (y // self.band_rows) % self.nranks

    def rows(self, rank):
        y = np.arange(self.height)
        return y[(y // self.band_rows) % self.nranks == rank]

    def local_bands(self, rank):
        return list(range(rank, self.bands, self.nranks))

    def max_bands(self):
        return (self.bands + self.nranks - 1) // self.nranks

    def halo_floats(self, radius):
        """float32 count of one halo message: max_bands x radius rows x 3 images x width x rgba."""
        return self.max_bands() * radius * 3 * self.width * 4

    def neighbours(self, rank):
        return (rank - 1) % self.nranks, (rank + 1) % self.nranks


class HaloExchange:
    """The denoise-halo exchange of one rank: export this rank's band-edge rows, swap them with rank-1 / rank+1, import what
    arrived.  Messages: to_prev (tag 0) and to_next (tag 1); with nranks == 2 both go to the same peer.  The four message
    buffers live as long as this object (vxrt_halo_import copies out of them on the context's stream and returns after that
    copy has finished, so they may be re-used by the next frame's exchange)."""

    def __init__(self, ctx, dist, rank, nranks, device, torch, comm_device=None):
        """device: where the context's halo buffers live (the rank's GPU; "cpu" for the oracle-backed stand-in of the CPU tests).
        comm_device: where the messages travel — the same device over RCCL (default); "cpu" stages them through host memory
        for a gloo rehearsal of several ranks on one GPU."""
        self.ctx, self.dist, self.rank, self.nranks, self.device, self.torch = ctx, dist, rank, nranks, device, torch
        self.comm_device = device if comm_device is None else comm_device
        self.nfloats, self.bufs, self.staged = 0, None, None

    def _buffers(self):
        n = self.ctx.halo_bytes() // 4
        if self.bufs is None or n != self.nfloats:
            self.nfloats = n
            self.bufs = [self.torch.empty(max(n, 1), dtype=self.torch.float32, device=self.device) for _ in range(4)]
            self.staged = None
            if str(self.comm_device) != str(self.device):
                self.staged = [self.torch.empty(max(n, 1), dtype=self.torch.float32, device=self.comm_device) for _ in range(4)]
        return self.bufs

    def exchange(self):
        to_prev, to_next, from_prev, from_next = self._buffers()
        if self.nranks < 2 or self.nfloats == 0:
            return
        dist, rank, nranks = self.dist, self.rank, self.nranks
        self.ctx.halo_export(to_prev.data_ptr(), to_next.data_ptr())      # synchronous: the rows are in the buffers on return
        prev, nxt = (rank - 1) % nranks, (rank + 1) % nranks
        s_prev, s_next, r_prev, r_next = (to_prev, to_next, from_prev, from_next) if self.staged is None else self.staged
        if self.staged is not None:
            s_prev.copy_(to_prev)
            s_next.copy_(to_next)
        ops = [dist.P2POp(dist.isend, s_prev, prev, tag=0), dist.P2POp(dist.isend, s_next, nxt, tag=1),
               dist.P2POp(dist.irecv, r_next, nxt, tag=0),    # what the next rank addressed to ITS prev (me)
               dist.P2POp(dist.irecv, r_prev, prev, tag=1)]   # what the previous rank addressed to ITS next (me)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.staged is not None:
            from_prev.copy_(r_prev)
            from_next.copy_(r_next)
        if str(self.device) != "cpu":
            self.torch.cuda.synchronize()
        self.ctx.halo_import(from_prev.data_ptr(), from_next.data_ptr())


def exchange_halo(ctx, dist, rank, nranks, device, torch):
    """One exchange with buffers of its own (see HaloExchange, which keeps them across frames)."""
    HaloExchange(ctx, dist, rank, nranks, device, torch).exchange()


def render_frame(ctx, dist, rank, nranks, device, torch, radius, halo=None):
    """One frame of Context::render (src/context.rs:2004-2075) on a rank: trace -> temporal -> [halo] -> denoise.
    halo: a HaloExchange to re-use across frames (one is made for the call otherwise)."""
    ctx.render(TRACE | TEMPORAL)
    if nranks > 1 and radius > 0:
        (halo or HaloExchange(ctx, dist, rank, nranks, device, torch)).exchange()
    ctx.render_stage(DENOISE)


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
