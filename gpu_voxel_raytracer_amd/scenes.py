"""Benchmark / test scene set-up shared by bench.py and the tests: fixed cameras and
the voxel-list fixtures.  Pure numpy; no GPU, no oracle."""
import os

import numpy as np

FIXTURE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")

# 70.0f32.to_radians() (src/context.rs:621): 70 * (PI_f32 / 180)
FOV_70 = float(np.float32(70.0) * (np.float32(np.pi) / np.float32(180.0)))


def load_scene(name):
    """-> (pos int16[n,3], mrgb uint8[n,4], size_xyz) of tests/golden/scenes/<name>.npz: the voxel list the
    reference's voxels_from_vox (src/context.rs:913-933) produces for vox/<name>.vox."""
    z = np.load(os.path.join(FIXTURE_DIR, name + ".npz"))
    return z["pos"], z["mrgb"], tuple(int(v) for v in z["size"])


def world_extent(size_xyz):
    """World-space extent of a model: voxel (x,y,z) -> integer (x,z,y) -> world cube of side 0.5."""
    sx, sy, sz = size_xyz
    return np.array([sx, sz, sy], np.float32) * np.float32(0.5)


def bench_camera(size_xyz):
    """The fixed outside view of SURVEY.md §8d: position = c + e*(-0.9, 0.6, -1.2), looking at c."""
    ext = world_extent(size_xyz)
    c = ext * np.float32(0.5)
    e = np.float32(ext.max())
    position = (c + e * np.array([-0.9, 0.6, -1.2], np.float32)).astype(np.float32)
    direction = (c - position).astype(np.float32)
    return position, direction, FOV_70


def close_camera(size_xyz):
    """A second view with the camera close to the model so that geometry fills the frame."""
    ext = world_extent(size_xyz)
    c = ext * np.float32(0.5)
    e = np.float32(ext.max())
    position = (c + e * np.array([-0.45, 0.30, -0.55], np.float32)).astype(np.float32)
    direction = (c - position).astype(np.float32)
    return position, direction, FOV_70


def reference_start_camera():
    """The reference's start-up camera (src/context.rs:618-622)."""
    return np.array([0.0, 0.0, -2.0], np.float32), np.array([0.0, 0.0, 1.0], np.float32), FOV_70


# BASELINE configs[4]: the procedural level-7 Menger sponge clipped to 2048^3 — level, clip, colour
# (material, r, g, b), emissive period — and its two views (Context.set_menger(*CONFIG5)).
CONFIG5 = (7, 2048, (0, 150, 170, 120), 8192)


def config5_cameras():
    ext = np.float32(1024)   # world extent of 2048 voxels
    return {"outside": (np.array([-0.9, 0.6, -1.2], np.float32) * ext + ext / 2, np.array([0.9, -0.6, 1.2], np.float32), 1.2217305),
            "tunnel": (np.array([0.5, 0.5, 0.02], np.float32) * ext, np.array([0.05, 0.03, 1.0], np.float32), 1.2217305)}
