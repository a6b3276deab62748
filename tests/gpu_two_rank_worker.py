"""Worker of tests/test_gpu_distributed.py (not collected by pytest): one of N ranks launched with torch.distributed.run.
VXRT_TEST_BACKEND = "gloo" (default): the ranks SHARE this box's GPU 0 and the halo is staged through host memory, because gloo
carries CPU tensors.  "nccl": one GPU per rank (LOCAL_RANK), the halo travels GPU to GPU over RCCL — the path a node runs; needs
as many GPUs as ranks.  Every rank renders its bands of each frame through gpu_voxel_raytracer_amd.distributed (trace + temporal
-> halo exchange -> denoise); rank 0 stitches the ranks' rows, renders the same frames in a single context and prints one JSON
line with the comparison."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from gpu_voxel_raytracer_amd import ACCUM_COLOR, ALL, DENOISED, SAMPLED_COLOR, Camera, Context, distributed, scenes
    from gpu_voxel_raytracer_amd.host import OPT_HALO_ROWS
    backend = os.environ.get("VXRT_TEST_BACKEND", "gloo")
    if backend == "nccl":
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        assert dev < torch.cuda.device_count(), "the nccl mode needs one GPU per rank"
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        comm, red = None, torch.device("cuda", dev)      # messages GPU to GPU; reductions on the device
    else:
        dev = 0
        dist.init_process_group(backend)
        comm, red = "cpu", "cpu"
    gpu = torch.device("cuda", dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    if os.environ.get("VXRT_TEST_CONFIG4") == "1":
        return config4(dist, torch, rank, world, dev, gpu, comm, red, backend)
    w, h, bounces, radius = 320, 200, 3, int(os.environ.get("VXRT_TEST_RADIUS", "3"))
    band, halo_rows = int(os.environ.get("VXRT_TEST_BAND", "16")), os.environ.get("VXRT_TEST_HALO_ROWS", "1")
    auto_rows = halo_rows == "auto"      # sized frame by frame from the camera path (distributed.halo_rows_for_motion)
    halo_rows = 1 if auto_rows else int(halo_rows)
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    layout = distributed.BandLayout(w, h, world, band, radius=radius)
    # camera paths: at rest; a slow drift (every reprojection stays within the halo rows: the nearest geometry, 0.6 units from the
    # camera, moves by 0.3 rows per frame); one fast pan (reprojections leave the rank's rows + halo: the frame after the jump)
    paths = {"rest": [(p0, d0)] * 3,
             "slow": [(p0 + np.float32(0.001 * k) * np.array([1, 0.5, 0], np.float32), d0) for k in range(4)],
             "fast": [(p0, d0), (p0, d0 + np.float32(0.12 * float(np.linalg.norm(d0))) * np.array([0, -1, 0], np.float32))]}   # ~0.12 rad: 17 rows
    out = {}
    for name, path in paths.items():
        ctx = Context(w, h, device=dev, max_bounces=bounces, rank=rank, nranks=world, band_rows=band)
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = radius
        ctx.set_option(OPT_HALO_ROWS, halo_rows)
        halo = distributed.HaloExchange(ctx, dist, rank, world, gpu, torch, comm_device=comm)
        chosen = []
        for k, (cp, cd) in enumerate(path):
            ctx.camera = Camera(cp, cd, fov)
            rows = None
            if auto_rows:       # this frame's exchange carries what the NEXT frame's reprojection can reach
                nxt = path[min(k + 1, len(path) - 1)]
                axes = [(c[0],) + Camera(c[0], c[1], fov).axis_scaled(w, h) for c in ((cp, cd), nxt)]
                rows = distributed.halo_rows_for_motion(axes[0], axes[1], w, h, near=0.25, band_rows=band)
                chosen.append(rows)
            distributed.render_frame(ctx, dist, rank, world, gpu, torch, radius, halo=halo, halo_rows=rows)
        imgs = {k: distributed.gather_image(ctx.read(i), layout, rank, dist, torch, red)
                for k, i in (("sampled", SAMPLED_COLOR), ("accum", ACCUM_COLOR), ("denoised", DENOISED))}
        rays = torch.tensor([ctx.stats().rays], device=red)
        dist.all_reduce(rays)
        ctx.close()
        if rank == 0:
            single = Context(w, h, device=dev, max_bounces=bounces)
            single.recreate_octree(pos, mrgb)
            single.denoise_uniforms.radius = radius
            for cp, cd in path:
                single.camera = Camera(cp, cd, fov)
                single.render(ALL)
            want = {"sampled": single.read(SAMPLED_COLOR), "accum": single.read(ACCUM_COLOR), "denoised": single.read(DENOISED)}
            res = {"rays_equal": int(rays.item()) == single.stats().rays}
            for k in want:
                a, b = imgs[k], want[k]
                same = (a == b) | (np.isnan(a) & np.isnan(b))
                res[k + "_differing_pixels"] = int((~same.all(-1)).sum())
            # a reprojection that leaves the rank's rows (+ halo) is a disocclusion: blending 1, i.e. accumulated = sampled colour,
            # and the next frame's blending factor is clamp(0.5 * 1) = 0.5
            diff = ~((imgs["accum"] == want["accum"]) | (np.isnan(imgs["accum"]) & np.isnan(want["accum"]))).all(-1)
            res["accum_differs_only_where_treated_as_disocclusion"] = bool(
                (imgs["accum"][diff][:, :3] == imgs["sampled"][diff][:, :3]).all() and (imgs["accum"][diff][:, 3] == 0.5).all())
            res["geometry_pixels"] = int((single.read(1)[..., 3] >= 0).sum())
            res["halo_rows_chosen"] = chosen
            single.close()
            out[name] = res
    if rank == 0:
        out["backend"], out["world_size"] = backend, world
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def config4(dist, torch, rank, world, dev, gpu, comm, red, backend):
    """BASELINE configs[3] at its size through the path a node runs: vox/castle.vox 3840x2160, 4 samples per pixel
    per displayed frame, 8 bounces, temporal + denoise with the 17 x 17 window, one context per rank over interleaved bands
    (distributed.band_rows_for), the halo — 8 rows per band edge, 36 B/px — through distributed.HaloExchange BETWEEN THE RANKS
    (RCCL send / recv between devices with VXRT_TEST_BACKEND=nccl; staged through the host over gloo), overlapped with the denoise of
    the interior tiles.  Rank 0 stitches the ranks' rows of two displayed frames and compares them with one context's, bit for bit."""
    from gpu_voxel_raytracer_amd import ACCUM_COLOR, ALL, DENOISED, TEMPORAL, TRACE, Camera, Context, distributed, scenes
    w, h, bounces, spp, radius, frames = 3840, 2160, 8, 4, 8, 2
    band = distributed.band_rows_for(radius, h, world)
    pos, mrgb, size = scenes.load_scene("castle")
    cam = Camera(*scenes.close_camera(size))
    layout = distributed.BandLayout(w, h, world, band, radius=radius)
    ctx = Context(w, h, device=dev, max_bounces=bounces, rank=rank, nranks=world, band_rows=band, frames_in_flight=2, frames_per_launch=spp)
    ctx.recreate_octree(pos, mrgb)
    ctx.camera = cam
    ctx.denoise_uniforms.radius = radius
    halo = distributed.HaloExchange(ctx, dist, rank, world, gpu, torch, comm_device=comm)
    for _ in range(frames):
        ctx.render_spp(TRACE | TEMPORAL, spp)
        distributed.finish_frame(ctx, world, radius, halo, overlap=True)
    info = ctx.halo_info()
    imgs = {k: distributed.gather_image(ctx.read(i), layout, rank, dist, torch, red) for k, i in (("accum", ACCUM_COLOR), ("denoised", DENOISED))}
    st = ctx.stats()
    rays = torch.tensor([st.rays], device=red)
    dist.all_reduce(rays)
    ctx.close()
    if rank == 0:
        single = Context(w, h, device=dev, max_bounces=bounces, frames_in_flight=2, frames_per_launch=spp)
        single.recreate_octree(pos, mrgb)
        single.camera = cam
        single.denoise_uniforms.radius = radius
        for _ in range(frames):
            single.render_spp(ALL, spp)
        want = {"accum": single.read(ACCUM_COLOR), "denoised": single.read(DENOISED)}
        res = {"backend": backend, "world_size": world, "config4": True, "band_rows": band, "halo_rows": int(info.rows),
               "halo_bytes_per_rank_per_frame": 2 * int(info.message_bytes), "halo_exchanges": halo.exchanges,
               "rays_equal": int(rays.item()) == single.stats().rays, "geometry_pixels": int((single.read(1)[..., 3] >= 0).sum())}
        for k in want:
            a, b = imgs[k], want[k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            res[k + "_differing_pixels"] = int((~same.all(-1)).sum())
        single.close()
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
