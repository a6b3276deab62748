"""GPU parity, blue-noise table: the HIP void-and-cluster kernel (csrc/noise.hip, through
vxrt_blue_noise) against the CPU restatement of include/vxrt_bluenoise.h (oracle/onoise.cpp).  Bar: BIT-EXACT (the
algorithm is integer choices over binary32 sums accumulated in a fixed order).  The reference ships neither the table nor a
generator (resources/blue-noise-128.zip is missing), so parity with the reference is unpinned; what is checked is spec ==
kernel, plus the properties a blue-noise rank table must have at full size."""
import time

import numpy as np
import pytest

from conftest import assert_bits_equal
from test_scene_extensions import radial_power

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001


@pytest.mark.parametrize("size,layers", [(16, (0, 1)), (32, (0, 7, 300)), (64, (2,)), (128, (0, 511))])
def test_layers_bit_exact_vs_oracle(O, H, size, layers):
    for layer in layers:
        got = H.blue_noise(SEED, size=size, first_layer=layer, layers=1)[0]
        assert_bits_equal(got, O.blue_noise_layer(SEED, layer, size), f"size {size} layer {layer}")
    other = H.blue_noise(SEED + 1, size=size, first_layer=layers[0], layers=1)[0]
    assert_bits_equal(other, O.blue_noise_layer(SEED + 1, layers[0], size), "other seed")


def test_full_table_properties_and_batch_consistency(O, H):
    t0 = time.perf_counter()
    table = H.blue_noise(SEED)                       # 512 x 128 x 128, the shape shaders/voxels.comp:65-71 indexes
    dt = time.perf_counter() - t0
    print(f"\n512 layers of 128x128 void-and-cluster on the GPU: {dt * 1e3:.0f} ms")
    assert table.shape == (512, 128, 128)
    ranks = np.sort((table.reshape(512, -1).astype(np.float64) * 16384 - 0.5).round().astype(np.int64), axis=1)
    assert (ranks == np.arange(16384)).all()         # every layer uses every value (r + 0.5) / 16384 once
    assert table.min() > 0.0 and table.max() < 1.0
    # a layer does not depend on which launch made it
    assert_bits_equal(table[37], H.blue_noise(SEED, first_layer=37, layers=1)[0], "layer 37 alone")
    assert_bits_equal(table[0], O.blue_noise_layer(SEED, 0, 128), "layer 0 of the batch")
    # blue spectrum: almost no energy at low spatial frequency (white noise has ratio ~1); layers are distinct
    for layer in (1, 100, 510):
        f, r = radial_power(table[layer])
        assert f[(r > 0) & (r <= 0.1)].mean() < 0.01 * f[r > 0].mean()
    assert len({table[i].tobytes() for i in range(0, 512, 37)}) == len(range(0, 512, 37))
    # thresholding at any level leaves well-separated points: the nearest-neighbour distance of the 5 % darkest cells
    pts = np.argwhere(table[5] < 0.05).astype(np.float64)
    d = np.abs(pts[:, None, :] - pts[None, :, :])
    d = np.minimum(d, 128 - d)
    dist = np.sqrt((d ** 2).sum(-1)) + np.eye(len(pts)) * 1e9
    assert dist.min(axis=1).min() >= 2.0 and dist.min(axis=1).mean() > 3.5   # random points: mean ~2.2, min 1


def test_trace_with_a_blue_noise_table_is_bit_exact(O, H, scenes, tmp_path):
    """The generated table through the reference's archive format into the tracer: same pixels as the oracle fed with it."""
    from gpu_voxel_raytracer_amd import ALBEDO_NODE, NORMAL_DEPTH, SAMPLED_COLOR, TRACE, Camera, Context
    few = H.blue_noise(SEED, layers=16)
    table = np.tile(few, (32, 1, 1))                 # 512 layers (a full table is tested above; the tracer needs the shape)
    path = str(tmp_path / "blue-noise-128.zip")
    H.save_blue_noise(path, table)
    size, loaded = H.load_blue_noise(path)
    assert size == 128 and loaded.size == 512 * 128 * 128
    w, h, bounces = 160, 96, 4
    pos, mrgb, scene_size = scenes.load_scene("castle")
    cam = scenes.close_camera(scene_size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 1
    ref = O.trace(O.create_octree(pos, mrgb), loaded, u, w, h, bounces, crop=(0, 0, w, h))
    with Context(w, h, max_bounces=bounces) as ctx:   # created with the white table, then switched
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.set_noise(loaded)
        ctx.render(TRACE)
        for img, want, what in ((SAMPLED_COLOR, ref[0], "colour"), (NORMAL_DEPTH, ref[1], "normal/depth"), (ALBEDO_NODE, ref[2], "albedo")):
            assert_bits_equal(ctx.read(img), want, what)
        assert ctx.stats().rays == ref[3]
