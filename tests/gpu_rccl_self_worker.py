"""Worker of tests/test_gpu_distributed.py::test_halo_messages_over_rccl_on_one_gpu (not collected by pytest) and of
scripts/nccl_sanity.py: the RCCL branch of distributed.HaloExchange exercised on ONE GPU.  RCCL refuses two ranks on one device, so
this is a world of one rank whose context believes it is rank 0 of 2; both neighbours are mapped onto the rank itself
(`SelfLoop.nranks = 1` for the peer arithmetic only).  The messages a rank sends then come back to it — NOT what a frame needs, the
image is not looked at — and the test is the transport: batch_isend_irecv on the side stream, Work.wait() under torch.cuda.stream, the
in-order matching of two messages to one peer, the events between the context's stream and the communication stream."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from gpu_voxel_raytracer_amd import TEMPORAL, TRACE, DENOISE_INTERIOR, DENOISE_EDGE, Camera, Context, distributed, scenes
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    w, h, radius, band = 640, 384, 3, 32
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)

    class SelfLoop(distributed.HaloExchange):
        """prev and next are this rank: `nranks` 1 for the peer arithmetic of _ops, 2 for everything else."""
        def _ops(self, *bufs):
            self.nranks = 1
            try:
                return super()._ops(*bufs)
            finally:
                self.nranks = 2

    res = {"backend": dist.get_backend(), "frames": 0, "to_prev_arrived_as_from_next": True, "to_next_arrived_as_from_prev": True,
           "messages_differ": True}

    def run(synchronous):
        """Four frames of the overlapped loop.  synchronous: the host waits after every step, so nothing depends on the events between
        the context's stream, the communication stream and RCCL's — the reference for the run in which everything is asynchronous."""
        from gpu_voxel_raytracer_amd import ACCUM_COLOR, DENOISED
        ctx = Context(w, h, device=0, max_bounces=3, rank=0, nranks=2, band_rows=band)
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = radius
        halo = SelfLoop(ctx, dist, 0, 2, torch.device("cuda", 0), torch)

        def settle():
            if synchronous:
                ctx.sync()
                torch.cuda.synchronize()
        images = []
        for frame in range(4):
            ctx.render(TRACE | TEMPORAL); settle()
            halo.start(); settle()
            ctx.render_stage(DENOISE_INTERIOR); settle()
            halo.finish(); settle()
            ctx.render_stage(DENOISE_EDGE); settle()
            if not synchronous:      # look at the messages only after the frame: the loop itself never waits on the host
                ctx.sync()
                torch.cuda.synchronize()
                to_prev, to_next, from_prev, from_next = (b.cpu().numpy().view(np.uint32) for b in halo.bufs)
                res["to_prev_arrived_as_from_next"] &= bool((to_prev == from_next).all())
                res["to_next_arrived_as_from_prev"] &= bool((to_next == from_prev).all())
                res["messages_differ"] &= bool((to_prev != to_next).any() and to_prev.any() and to_next.any())
                res["frames"] += 1
            images.append((ctx.read(ACCUM_COLOR).view(np.uint32).copy(), ctx.read(DENOISED).view(np.uint32).copy()))
        info = (int(ctx.halo_info().message_bytes), halo.exchanges)
        ctx.close()
        return images, info

    asynchronous, (message_bytes, exchanges) = run(False)
    reference, _ = run(True)
    # the unpack must have waited for the receives and the edge tiles for the unpack: the asynchronous loop's frames equal the frames of
    # the loop in which the host waited after every step (both see the same — deliberately wrong — neighbours)
    res["asynchronous_frames_equal_synchronous_frames"] = all(bool((a[0] == b[0]).all() and (a[1] == b[1]).all()) for a, b in zip(asynchronous, reference))
    res["message_bytes"], res["exchanges"] = message_bytes, exchanges
    res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    res["device"] = torch.cuda.get_device_name(0)
    dist.barrier()
    torch.cuda.synchronize()
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
