"""GPU parity against OUTPUTS OF THE REFERENCE'S COMPILED SHADERS: tests/golden/spirv_exec/*.npz hold what
shaders/{voxels,temporal,denoise}.comp.spv — the modules the reference hands to its GPU — produce for six short frame sequences
when executed instruction by instruction (oracle/ospirv.cpp; made by tests/golden/make_spirv_exec_fixture.py where the reference is
mounted; driver-defined operations bound to the documented choices U1-U8).  The HIP path, through the C ABI, must give the same
images: moving camera with a denoise window, the 17 x 17 window, specular and sun-off shading, 0 * inf rays, the 2 048-trip cap."""
import os

import numpy as np
import pytest

import spirv_pipeline as SP
from conftest import GOLDEN, assert_bits_equal

pytestmark = pytest.mark.gpu

FIXTURES = os.path.join(GOLDEN, "spirv_exec")
CASES = sorted(f[:-4] for f in os.listdir(FIXTURES) if f.endswith(".npz") and f != "scene_sweep.npz")


@pytest.mark.parametrize("name", CASES)
def test_hip_frames_equal_the_compiled_shaders_frames(H, scenes, noise, name):
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    z = np.load(os.path.join(FIXTURES, name + ".npz"))
    w, h = int(z["w"]), int(z["h"])
    pos, mrgb = SP.cap_scene() if str(z["scene"]) == "cap" else scenes.load_scene(str(z["scene"]))[:2]
    with Context(w, h, max_bounces=int(z["max_bounces"]), noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.uniforms.specularity = float(z["specularity"])
        ctx.uniforms.sun_strength = float(z["sun_strength"])
        ctx.uniforms.emit_strength = float(z["emit_strength"])
        ctx.denoise_uniforms.radius = int(z["radius"])
        for f in range(1, len(z["fov"]) + 1):
            ctx.camera = Camera(z["cam_pos"][f - 1], z["cam_dir"][f - 1], float(z["fov"][f - 1]))
            ctx.render(ALL)
            if f == 1:
                for img, key in ((0, "f1_color"), (1, "f1_nd"), (2, "f1_albedo")):
                    assert_bits_equal(ctx.read(img), z[key], f"{name} {key}")
            assert_bits_equal(ctx.read(3), z[f"f{f}_accum"], f"{name} accumulated colour of frame {f}")
            assert_bits_equal(ctx.read(4), z[f"f{f}_denoised"], f"{name} denoised colour of frame {f}")
        assert ctx.stats().frames == len(z["fov"])


def test_hip_full_size_frames_hash_to_the_compiled_shaders_known_answers(H, scenes, noise):
    """BASELINE's frames at their full sizes — configs[1] (menger 1920 x 1080, 4 bounces: the frame bench.py times) and the frames of
    configs 3 and 4 (monu10 / castle, 3840 x 2160, 8 bounces): the HIP trace stage's three images, hashed slab by slab, equal what the
    reference's compiled voxels.comp gave (tests/golden/spirv_exec/full_size.json)."""
    import json
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    with open(os.path.join(FIXTURES, "full_size.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 3
    for c in cases:
        pos, mrgb, size = scenes.load_scene(c["scene"])
        cam = scenes.bench_camera(size) if c["view"] == "bench" else scenes.close_camera(size)
        with Context(c["w"], c["h"], max_bounces=c["bounces"], noise=noise) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*cam)
            ctx.render(TRACE)
            images = {"color": ctx.read(0), "nd": ctx.read(1), "albedo": ctx.read(2)}
        rows = c["slab_rows"]
        for k, img in images.items():
            for s, want in enumerate(c["sha256"][k]):
                got = SP.canonical_sha256(img[s * rows:(s + 1) * rows], node_channel=(k == "albedo"))
                assert got == want, f"{c['name']}: rows {s * rows}-{(s + 1) * rows} of the {k} image"


def test_hip_frame_loop_at_config3_size_hashes_to_the_compiled_shaders_known_answers(H, scenes, noise):
    """Config 3's frame loop at its size (monu10, 3840 x 2160, 8 bounces, camera at rest): two frames through all three stages — the
    accumulated colour of both frames and frame 2 denoised with the 5 x 5 window hash, slab by slab, to what the reference's three compiled
    modules gave; so does the strip of frame 2 denoised again with the 17 x 17 window."""
    import json
    from gpu_voxel_raytracer_amd import ALL, DENOISE, Camera, Context
    with open(os.path.join(FIXTURES, "full_size.json")) as f:
        z = json.load(f)["frame_loop"]
    pos, mrgb, size = scenes.load_scene(z["scene"])
    cam = scenes.close_camera(size)
    with Context(z["w"], z["h"], max_bounces=z["bounces"], noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = z["radius"]
        for f in (1, 2):
            ctx.render(ALL)
            assert SP.slab_hashes(ctx.read(3), z["slab_rows"]) == z["sha256"][f"f{f}_accum"], f"accumulated colour of frame {f}"
        assert SP.slab_hashes(ctx.read(4), z["slab_rows"]) == z["sha256"]["f2_denoised_r2"]
        ctx.denoise_uniforms.radius = 8
        ctx.update_bindings()
        ctx.render_stage(DENOISE)
        y0, y1 = z["r8_rows"]
        assert SP.canonical_sha256(ctx.read(4)[y0:y1]) == z["sha256"]["f2_denoised_r8_rows"]


@pytest.mark.parametrize("env", [{"VXRT_TRACE_VARIANT": "0"}, {"VXRT_TRACE_VARIANT": "2", "VXRT_TRACE_SPLIT": "0x3"}, {"VXRT_TRACE_VARIANT": "3"},
                                 {"VXRT_TRACE_VARIANT": "4", "VXRT_TAIL_FROM": "0"}, {"VXRT_TRACE_VARIANT": "5"}, {"VXRT_WIDE": "1"},
                                 {"VXRT_FUSED_TAIL": "1"}])
@pytest.mark.parametrize("name", ["castle_moving_r2", "cap_row_r0", "monu10_8_bounces_r2"])
def test_every_schedule_and_scene_format_gives_the_compiled_shaders_frames(H, scenes, noise, monkeypatch, env, name):
    """The all-in-one kernel, the wavefront, ray-queue and path-refill tracers, the hand-over at the first hit, the fused head + tail and the
    two-level scene records (the variants library, loaded beside the product): the same frames as the reference's compiled shaders."""
    from conftest import require_variants
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    require_variants(H, env)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = np.load(os.path.join(FIXTURES, name + ".npz"))
    pos, mrgb = SP.cap_scene() if str(z["scene"]) == "cap" else scenes.load_scene(str(z["scene"]))[:2]
    with Context(int(z["w"]), int(z["h"]), max_bounces=int(z["max_bounces"]), noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.uniforms.specularity = float(z["specularity"])
        ctx.denoise_uniforms.radius = int(z["radius"])
        for f in range(1, len(z["fov"]) + 1):
            ctx.camera = Camera(z["cam_pos"][f - 1], z["cam_dir"][f - 1], float(z["fov"][f - 1]))
            ctx.render(ALL)
            if f == 1:
                for img, key in ((0, "f1_color"), (1, "f1_nd"), (2, "f1_albedo")):
                    assert_bits_equal(ctx.read(img), z[key], f"{name} {key} {env}")
            assert_bits_equal(ctx.read(4), z[f"f{f}_denoised"], f"{name} denoised colour of frame {f} {env}")


@pytest.mark.parametrize("nranks", [2, 4])
def test_banded_contexts_with_a_halo_equal_the_compiled_shaders_frames(H, scenes, noise, nranks):
    """Row (e) against the same reference-made frames: the moving-camera sequence rendered by `nranks` contexts in interleaved 16-row bands —
    history and denoise window across band edges through the halo export / import path (what RCCL carries between GPUs, handed over
    directly) — stitched, equals the accumulated and denoised frames of the reference's compiled shaders."""
    import ctypes as C
    from gpu_voxel_raytracer_amd import DENOISE, TEMPORAL, TRACE, Camera, Context
    from gpu_voxel_raytracer_amd.host import OPT_HALO_ROWS
    z = np.load(os.path.join(FIXTURES, "castle_moving_r2.npz"))
    w, h = int(z["w"]), int(z["h"])
    pos, mrgb = scenes.load_scene("castle")[:2]
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipFree.argtypes = [C.c_void_p]
    ctxs = [Context(w, h, max_bounces=int(z["max_bounces"]), noise=noise, rank=r, nranks=nranks, band_rows=16) for r in range(nranks)]
    try:
        for c in ctxs:
            c.recreate_octree(pos, mrgb)
            c.denoise_uniforms.radius = int(z["radius"])
            c.set_option(OPT_HALO_ROWS, 16)
        rows = [c.local_rows() for c in ctxs]
        for f in range(1, len(z["fov"]) + 1):
            bufs = {}
            for r, c in enumerate(ctxs):
                c.camera = Camera(z["cam_pos"][f - 1], z["cam_dir"][f - 1], float(z["fov"][f - 1]))
                c.render(TRACE | TEMPORAL)
                p, n = C.c_void_p(), C.c_void_p()
                assert rt.hipMalloc(C.byref(p), c.halo_bytes()) == 0 and rt.hipMalloc(C.byref(n), c.halo_bytes()) == 0
                c.halo_export(p.value, n.value)
                bufs[r] = (p, n)
            for r, c in enumerate(ctxs):
                c.halo_import(bufs[(r - 1) % nranks][1].value, bufs[(r + 1) % nranks][0].value)
                c.render_stage(DENOISE)
            for c in ctxs:
                c.sync()
            for p, n in bufs.values():
                rt.hipFree(p); rt.hipFree(n)
            for img, key in ((3, f"f{f}_accum"), (4, f"f{f}_denoised")):
                got = np.zeros((h, w, 4), np.float32)
                for c, rr in zip(ctxs, rows):
                    got[rr] = c.read(img)
                assert_bits_equal(got, z[key], f"{key} from {nranks} banded contexts")
    finally:
        for c in ctxs:
            c.close()


def test_hip_trace_of_every_scene_file_equals_the_compiled_shaders(H, scenes, noise):
    """All 15 scene files of the reference, close view at 64 x 40, 3 bounces: the HIP trace stage's three images equal the compiled
    voxels.comp's (tests/golden/spirv_exec/scene_sweep.npz)."""
    from gpu_voxel_raytracer_amd import TRACE, Camera, Context
    z = np.load(os.path.join(FIXTURES, "scene_sweep.npz"))
    names = SP.sweep_scenes()
    assert len(names) == 15
    with Context(int(z["w"]), int(z["h"]), max_bounces=int(z["max_bounces"]), noise=noise) as ctx:
        for name in names:
            pos, mrgb, size = scenes.load_scene(name)
            ctx.recreate_octree(pos, mrgb)
            ctx.camera = Camera(*scenes.close_camera(size))
            ctx.set_frame_number(int(z["frame_number"]) - 1)
            ctx.render(TRACE)
            for img, key in ((0, "color"), (1, "nd"), (2, "albedo")):
                assert_bits_equal(ctx.read(img), z[f"{name}_{key}"], f"{name} {key}")


def test_the_product_needs_no_interpreter(H):
    """The fixtures are data: nothing of oracle/ospirv.cpp is linked into or loaded by the product library."""
    import subprocess
    out = subprocess.run(["nm", "-D", H._build.LIB], capture_output=True, text=True).stdout
    assert "orc_spirv" not in out and "orc_" not in out
