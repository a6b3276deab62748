"""CPU: BASELINE.json config 1 — vox/3x3x3.vox at 256x256 through the restatement of the reference's
orphaned CPU ray caster (src/cpu.rs + src/cpu/octree.rs, oracle/ocpu.cpp).  That code is not compiled
into the reference (src/main.rs:7-12) and misses Backend/Coord/Ray/Camera::cast_rays, so it pins only
"plumbing": its pointer-octree walk (sorted mid-plane crossings) is a second, structurally different
traversal, and must find the same first voxel and face as the shader walk (voxels.comp:134-247)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

W = H_ = 256


def config1(O, scenes):
    pos, mrgb, size = scenes.load_scene("3x3x3")
    cam_pos, cam_dir, fov = scenes.bench_camera(size)
    basis = O.camera_axis_scaled(cam_pos, cam_dir, fov, W, H_)
    return pos, mrgb, cam_pos, basis


def test_config1_image_and_golden(O, scenes):
    pos, mrgb, cam_pos, basis = config1(O, scenes)
    # cpu.rs works in voxel units (unit voxels on [0, 2^depth]^3): world * 2; light at time 0 = (53.5, 15, 63.5)
    pixels, ht, hn, hv = O.cpu_rs_render(pos.astype(np.uint16), mrgb[:, 1:], cam_pos * 2, basis, W, H_, time=0.0)
    assert pixels.shape == (H_, W, 3) and pixels.dtype == np.uint8
    hit = hv >= 0
    assert 0.1 < hit.mean() < 0.9 and (pixels[~hit] == 0).all() and pixels[hit].max() > 0
    z = np.load(os.path.join(GOLDEN, "config1_3x3x3_256.npz"))
    assert np.array_equal(pixels, z["pixels"])                     # self-golden of the restatement
    assert np.array_equal(hv, z["hit_value"])


def test_config1_primary_hits_agree_with_shader_walk(O, scenes, noise):
    pos, mrgb, cam_pos, basis = config1(O, scenes)
    pixels, ht, hn, hv = O.cpu_rs_render(pos.astype(np.uint16), mrgb[:, 1:], cam_pos * 2, basis, W, H_, time=0.0)
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam_pos, basis)
    u.frame_number = 1
    color, nd, alb, rays = O.trace(octree, noise, u, W, H_, 1, crop=(0, 0, W, H_))   # 1 bounce = config 1
    node = alb[..., 3].view(np.int32)
    s_hit = nd[..., 3] >= 0
    c_hit = hv >= 0
    assert (s_hit == c_hit).mean() > 0.9995
    both = s_hit & c_hit
    assert ((node[both] & 0xffffff) == hv[both]).mean() > 0.999          # same voxel colour
    assert np.allclose(ht[both], 2 * nd[..., 3][both], rtol=2e-5, atol=1e-3)   # voxel units = 2 x world units
    single = np.abs(nd[..., :3]).sum(-1) == 1
    m = both & single
    assert (hn[m] == nd[..., :3][m]).all(-1).mean() > 0.999               # same entry face


def test_config1_threaded_render_is_the_same_image(O, scenes):
    """src/cpu.rs:43-46 renders with rayon's par_iter_mut over the pixels; the restatement deals rows to std::threads
    (bench.py times it that way, next to the GPU number).  Pixels are independent: any thread count gives the golden image."""
    pos, mrgb, cam_pos, basis = config1(O, scenes)
    z = np.load(os.path.join(GOLDEN, "config1_3x3x3_256.npz"))
    backend = O.CpuRsBackend(pos.astype(np.uint16), mrgb[:, 1:])
    for threads in (1, 3, 8):
        assert np.array_equal(backend.render(cam_pos * 2, basis, W, H_, time=0.0, nthreads=threads), z["pixels"])
    backend.close()
