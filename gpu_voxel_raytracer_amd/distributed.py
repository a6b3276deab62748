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

    def __init__(self, width, height, nranks, band_rows=16):
        if band_rows % 16:
            raise ValueError("band_rows must be a multiple of 16")
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


def exchange_halo(ctx, dist, rank, nranks, device, torch):
    """Export this rank's band-edge rows, swap them with rank-1 / rank+1, import what arrived.
    Messages: to_prev (tag 0) and to_next (tag 1); with nranks == 2 both go to the same peer."""
    nfloats = ctx.halo_bytes() // 4
    if nranks < 2 or nfloats == 0:
        return
    to_prev = torch.empty(nfloats, dtype=torch.float32, device=device)
    to_next = torch.empty(nfloats, dtype=torch.float32, device=device)
    from_prev = torch.empty(nfloats, dtype=torch.float32, device=device)
    from_next = torch.empty(nfloats, dtype=torch.float32, device=device)
    ctx.halo_export(to_prev.data_ptr(), to_next.data_ptr())
    prev, nxt = (rank - 1) % nranks, (rank + 1) % nranks
    ops = [dist.P2POp(dist.isend, to_prev, prev, tag=0), dist.P2POp(dist.isend, to_next, nxt, tag=1),
           dist.P2POp(dist.irecv, from_next, nxt, tag=0),    # what the next rank addressed to ITS prev (me)
           dist.P2POp(dist.irecv, from_prev, prev, tag=1)]   # what the previous rank addressed to ITS next (me)
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if device != "cpu" and str(device) != "cpu":
        torch.cuda.synchronize()
    ctx.halo_import(from_prev.data_ptr(), from_next.data_ptr())


def render_frame(ctx, dist, rank, nranks, device, torch, radius):
    """One frame of Context::render (src/context.rs:2004-2075) on a rank: trace -> temporal -> [halo] -> denoise."""
    ctx.render(TRACE | TEMPORAL)
    if nranks > 1 and radius > 0:
        exchange_halo(ctx, dist, rank, nranks, device, torch)
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
