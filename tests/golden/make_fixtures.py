#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ .  Run in the authoring container, where
/root/reference is mounted:   python tests/golden/make_fixtures.py

What it writes and where each comes from:

  scenes/<name>.npz    INPUT fixtures: the voxel list (pos int16[n,3], mrgb uint8[n,4], size) that the
                       reference's vox::parse + Context::voxels_from_vox produce for vox/<name>.vox,
                       computed with the oracle's restatement of those functions (oracle/ovox.cpp).
                       These are data derived from the reference's scene files (MagicaVoxel models),
                       in the exact form `Context::create_octree` consumes; no reference source text.
  octree_kat.json      known answers for the octree builder: depth, nodes per level, total nodes, byte
                       size, emissive voxel count, sha256 of the int32 buffer.  The per-level counts
                       reproduce SURVEY.md Appendix C (computed there by an unrelated throw-away parser).
  config1_3x3x3_256.npz  the restated src/cpu.rs image (u8) + primary-hit colours for BASELINE config 1.
  frames_<scene>.npz   small full-pipeline frames rendered BY THE ORACLE (trace -> temporal -> denoise
                       for frames 1..N): self-goldens that pin the oracle against silent change.  (The
                       fixtures that come from the reference itself — its compiled shaders, executed —
                       are tests/golden/spirv_exec/, written by make_spirv_exec_fixture.py.)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpu_voxel_raytracer_amd import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF_VOX = "/root/reference/vox"
OUT = os.path.dirname(os.path.abspath(__file__))
SCENES = ["menger", "monu10", "castle", "3x3x3", "8x8x8", "room",
          "custom", "teapot", "nature", "chr_knight", "doom", "shelf", "monu9", "chr_sword", "monu1"]   # all 15 files of vox/


def levels(octree):
    arr = octree[5:].reshape(-1, 8)
    out, cur = [], np.array([0])
    while len(cur):
        out.append(int(len(cur)))
        v = arr[cur].ravel()
        cur = v[v > 0]
    return out


def golden_frames(name, width, height, bounces, nframes, radii, specularity=0.0, camera="bench"):
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    noise = O.noise_table()
    cam_pos, cam_dir, fov = scenes.bench_camera(size) if camera == "bench" else scenes.close_camera(size)
    basis = O.camera_axis_scaled(cam_pos, cam_dir, fov, width, height)
    u = O.Uniforms.default()
    u.specularity = specularity
    u.set_camera(cam_pos, basis)
    tu = O.Temporal.default()
    out = {"width": width, "height": height, "bounces": bounces, "specularity": specularity,
           "cam_pos": cam_pos, "cam_dir": cam_dir, "fov": np.float32(fov)}
    old_color = np.zeros((height, width, 4), np.float32)
    old_nd = np.zeros((height, width, 4), np.float32)
    cam16 = u.camera16()
    total_rays = []
    for frame in range(1, nframes + 1):
        u.frame_number = frame
        color, nd, alb, rays = O.trace(octree, noise, u, width, height, bounces, crop=(0, 0, width, height))
        accum = O.temporal(color, nd, old_color, old_nd, cam16, cam16, tu, has_history=frame > 1)
        total_rays.append(rays)
        if frame in (1, 2, nframes):
            out[f"f{frame}_color"] = color
            out[f"f{frame}_nd"] = nd
            out[f"f{frame}_albedo"] = alb
            out[f"f{frame}_accum"] = accum
            for r in radii:
                du = O.Denoise.default()
                du.radius = r
                out[f"f{frame}_denoised_r{r}"] = O.denoise(accum, nd, alb, cam16, du)
        old_color, old_nd = accum, nd
    out["rays"] = np.array(total_rays, np.int64)
    return out


def main():
    os.makedirs(os.path.join(OUT, "scenes"), exist_ok=True)
    kat = {}
    for name in SCENES:
        data = open(os.path.join(REF_VOX, name + ".vox"), "rb").read()
        pos, mrgb, size = O.voxels_from_vox(data)
        np.savez_compressed(os.path.join(OUT, "scenes", name + ".npz"), pos=pos, mrgb=mrgb, size=np.array(size, np.uint32))
        octree = O.create_octree(pos, mrgb)
        kat[name] = {
            "size": list(size), "voxels": int(len(pos)), "depth": int(O.voxel_depth(pos)),
            "nodes_per_level": levels(octree), "nodes": int((len(octree) - 5) // 8), "bytes": int(octree.nbytes),
            "emissive": int((mrgb[:, 0] == 0x40).sum()), "sha256": hashlib.sha256(octree.tobytes()).hexdigest(),
            "vox_sha256": hashlib.sha256(data).hexdigest(),
        }
        print(name, kat[name]["voxels"], kat[name]["nodes_per_level"])
    json.dump(kat, open(os.path.join(OUT, "octree_kat.json"), "w"), indent=1)

    jobs = [
        ("menger", dict(width=128, height=72, bounces=4, nframes=8, radii=(0, 1, 8))),
        ("castle", dict(width=96, height=64, bounces=3, nframes=4, radii=(0, 2), specularity=0.5, camera="close")),
        ("room", dict(width=96, height=64, bounces=3, nframes=3, radii=(0, 2), camera="close")),
    ]
    for name, kw in jobs:
        g = golden_frames(name, **kw)
        np.savez_compressed(os.path.join(OUT, f"frames_{name}.npz"), **g)
        print("frames", name, "rays/frame", g["rays"].tolist())

    # BASELINE config 1: vox/3x3x3.vox, 256x256, restated src/cpu.rs at time 0 (voxel units = 2 x world units)
    pos, mrgb, size = scenes.load_scene("3x3x3")
    cam_pos, cam_dir, fov = scenes.bench_camera(size)
    basis = O.camera_axis_scaled(cam_pos, cam_dir, fov, 256, 256)
    pixels, _, _, hv = O.cpu_rs_render(pos.astype(np.uint16), mrgb[:, 1:], cam_pos * 2, basis, 256, 256, time=0.0)
    np.savez_compressed(os.path.join(OUT, "config1_3x3x3_256.npz"), pixels=pixels, hit_value=hv)


if __name__ == "__main__":
    main()
