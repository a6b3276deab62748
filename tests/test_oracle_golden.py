"""CPU: the oracle reproduces its committed frames (tests/golden/frames_*.npz) bit for bit.

These are SELF-goldens (made by tests/golden/make_fixtures.py with the oracle): they pin the oracle
against silent change, they do not pin it to the reference, which ships no images.  What does pin it to the reference:
tests/test_oracle_spirv_exec.py — the reference's compiled shaders, executed — and tests/golden/spirv_exec/."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits_equal


def run_pipeline(O, scenes, noise, name, z):
    w, h, b = int(z["width"]), int(z["height"]), int(z["bounces"])
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam_pos, cam_dir, fov = z["cam_pos"], z["cam_dir"], float(z["fov"])
    u = O.Uniforms.default()
    u.specularity = float(z["specularity"])
    u.set_camera(cam_pos, O.camera_axis_scaled(cam_pos, cam_dir, fov, w, h))
    cam16 = u.camera16()
    old_c = np.zeros((h, w, 4), np.float32)
    old_nd = np.zeros((h, w, 4), np.float32)
    nframes = len(z["rays"])
    for frame in range(1, nframes + 1):
        u.frame_number = frame
        color, nd, alb, rays = O.trace(octree, noise, u, w, h, b, crop=(0, 0, w, h))
        accum = O.temporal(color, nd, old_c, old_nd, cam16, cam16, O.Temporal.default(), frame > 1)
        yield frame, color, nd, alb, accum, rays, cam16
        old_c, old_nd = accum, nd


@pytest.mark.parametrize("name", ["menger", "castle", "room"])
def test_oracle_reproduces_golden_frames(O, scenes, noise, name):
    z = np.load(os.path.join(GOLDEN, f"frames_{name}.npz"))
    checked = 0
    for frame, color, nd, alb, accum, rays, cam16 in run_pipeline(O, scenes, noise, name, z):
        assert rays == int(z["rays"][frame - 1])
        if f"f{frame}_color" not in z:
            continue
        assert_bits_equal(color, z[f"f{frame}_color"], "colour")
        assert_bits_equal(nd, z[f"f{frame}_nd"], "nd")
        assert_bits_equal(alb, z[f"f{frame}_albedo"], "albedo")
        assert_bits_equal(accum, z[f"f{frame}_accum"], "accum")
        for key in z.files:
            if key.startswith(f"f{frame}_denoised_r"):
                du = O.Denoise.default()
                du.radius = int(key.rsplit("r", 1)[1])
                assert_bits_equal(O.denoise(accum, nd, alb, cam16, du), z[key], key)
                checked += 1
    assert checked >= 4


def test_crop_equals_region_of_full_frame(O, scenes, noise):
    """The oracle's crop mode (used to check 1080p/4K GPU frames in strips) is exact."""
    pos, mrgb, size = scenes.load_scene("castle")
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], 160, 90))
    u.frame_number = 3
    full = O.trace(octree, noise, u, 160, 90, 3, crop=(0, 0, 160, 90))
    part = O.trace(octree, noise, u, 160, 90, 3, crop=(37, 21, 101, 66))
    for a, b in zip(full[:3], part[:3]):
        assert_bits_equal(a[21:66, 37:101], b)
    one = O.trace(octree, noise, u, 160, 90, 3, crop=(0, 0, 160, 90), nthreads=1)
    assert one[3] == full[3] and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(one[:3], full[:3]))


def test_temporal_semantics(O):
    """temporal.comp:121-124: alpha sequence 1, 1/2, 1/4 ... floored at 1 - maximum_blending; sky pixels pass through."""
    h, w = 4, 8
    cam16 = np.array([0, 0, -5, 0, 1, 0, 0, 0, 0, 1, 0, 0, -4, 2, 4, 0], np.float32)
    nd = np.zeros((h, w, 4), np.float32)
    nd[..., 2] = -1.0
    nd[..., 3] = 5.0
    nd[0, 0] = (2.0 ** 30, 2.0 ** 30, 2.0 ** 30, -1.0)
    tu = O.Temporal.default()
    old_c, old_nd = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    alphas, vals = [], []
    for f in range(1, 10):
        c = np.full((h, w, 4), float(f), np.float32)
        acc = O.temporal(c, nd, old_c, old_nd, cam16, cam16, tu, f > 1)
        alphas.append(float(acc[1, 1, 3])); vals.append(float(acc[1, 1, 0]))
        assert acc[0, 0, 0] == f                                   # miss: new colour only
        old_c, old_nd = acc, nd
    assert alphas[:6] == [0.5, 0.25, 0.125, 0.0625, 0.03125, 0.02 if False else alphas[5]]
    assert abs(alphas[-1] - 0.02) < 1e-6 and min(alphas) >= 0.02 - 1e-7
    assert vals[0] == 1.0 and vals[1] == 1.5 and vals[2] == 1.875  # mix(old, new, alpha_prev)


def test_denoise_semantics(O):
    h, w = 24, 24
    rng = np.random.default_rng(3)
    cam16 = np.array([0, 0, -5, 0, 1, 0, 0, 0, 0, 1, 0, 0, -12, 12, 20, 0], np.float32)
    colors = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = -1; nd[..., 3] = 4.0
    alb = np.ones((h, w, 4), np.float32); alb[..., :3] = 0.5
    alb[..., 3] = np.array([0x80112233], np.uint32).view(np.float32)[0]
    du = O.Denoise.default()
    out0 = O.denoise(colors, nd, alb, cam16, du)                   # r = 0: centre colour * albedo (denoise.comp:89-90)
    assert np.array_equal(out0[..., :3], colors[..., :3] * np.float32(0.5)) and (out0[..., 3] == 1).all()
    du.radius = 3
    flat = colors.copy(); flat[..., :3] = 0.25
    out = O.denoise(flat, nd, alb, cam16, du)                      # constant image stays constant
    assert np.allclose(out[..., :3], 0.125, atol=1e-6)
    # a different material id blocks blending (1e4 * material_delta): the marked pixel keeps its own colour
    alb2 = alb.copy(); alb2[10, 10, 3] = np.array([0xC0112233], np.uint32).view(np.float32)[0]
    c2 = flat.copy(); c2[10, 10, :3] = 0.9
    out2 = O.denoise(c2, nd, alb2, cam16, du)
    assert abs(out2[10, 10, 0] - 0.45) < 1e-6 and abs(out2[10, 11, 0] - 0.125) < 1e-6
