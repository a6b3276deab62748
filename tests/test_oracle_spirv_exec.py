"""CPU: the oracle against the reference's COMPILED shaders, executed.

The reference ships shaders/{voxels,temporal,denoise}.comp.spv — the modules src/context/shader.rs:6-45 hands to the GPU.
oracle/ospirv.cpp interprets them instruction by instruction (control flow, memory layouts, integer and IEEE binary32 arithmetic in the
module's order); what SPIR-V leaves to the driver — exp / log / pow / sin / cos / sqrt / normalize, the sampler, inverse(mat4), float ->
int conversion out of range, the association of dot products — is bound to the documented choices U2-U8 (that file's header).

  * tests/golden/spirv_exec/ holds what the compiled shaders produce — eight short frame sequences as images, BASELINE's whole frames as hashes (made by
    tests/golden/make_spirv_exec_fixture.py in the container where /root/reference is mounted): the ORACLE must reproduce every image
    bit for bit — here and on the GPU box, where tests/test_gpu_spirv_goldens.py holds the HIP path to the same files;
  * where the reference is mounted, the fixtures are regenerated from its modules and must equal the committed files, and the oracle
    is compared LIVE against the modules on every scene file, both views, specular / sun-off / emissive shading, a moving camera with
    every denoise radius, synthetic G-buffers with NaN and inf, the 0 * inf rays and the 2 048-trip cap;
  * undefined reads: with every Function variable poisoned at each function entry the outputs do not change — except the `normal` of a
    ray that ends at the trip cap (U1), for exactly those pixels.
"""
import os

import numpy as np
import pytest

import spirv_pipeline as SP
from conftest import GOLDEN, assert_bits_equal, needs_reference

FIXTURES = os.path.join(GOLDEN, "spirv_exec")
CASES = sorted(f[:-4] for f in os.listdir(FIXTURES) if f.endswith(".npz") and f != "scene_sweep.npz")
IMAGES = ("f1_color", "f1_nd", "f1_albedo")


def raw_equal(a, b, what):
    """Every bit, signs of zero included; a NaN equals a NaN (IEEE 754 leaves a NaN's sign and payload open, and x86 takes them from
    the FIRST operand of a commutative operation — which one that is, is the compiler's choice on either side)."""
    fa, fb = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    a, b = fa.view(np.uint32), fb.view(np.uint32)
    assert a.shape == b.shape, what
    bad = (a != b) & ~(np.isnan(fa) & np.isnan(fb))
    assert not bad.any(), f"{what}: {int(bad.any(axis=-1).sum())} texels differ, first at {tuple(np.argwhere(bad)[0])}"


def spec_of(z):
    """The sequence a fixture file describes (it is self-contained: scene name, size, radius, cameras, the uniforms that differ)."""
    frames = [(z["cam_pos"][k], z["cam_dir"][k], float(z["fov"][k])) for k in range(len(z["fov"]))]
    return dict(scene=str(z["scene"]), w=int(z["w"]), h=int(z["h"]), radius=int(z["radius"]), frames=frames, specularity=float(z["specularity"]),
                sun_strength=float(z["sun_strength"]), emit_strength=float(z["emit_strength"]), bounces=int(z["max_bounces"]))


def test_the_fixture_set_is_complete():
    assert len(CASES) == 8 and {"castle_moving_r2", "menger_static_r8", "zero_times_inf_r0", "cap_row_r0"} <= set(CASES)
    assert sorted(int(np.load(os.path.join(FIXTURES, name + ".npz"))["max_bounces"]) for name in CASES) == [3] * 6 + [4, 8]
    for name in CASES:
        z = np.load(os.path.join(FIXTURES, name + ".npz"))
        assert int(z["noise_seed"]) == 0x5EED0001
        assert (z["f1_nd"][..., 3] >= 0).sum() > 500, name                       # the frames see geometry
    z = np.load(os.path.join(FIXTURES, "zero_times_inf_r0.npz"))
    assert np.isnan(z["f1_nd"][..., 3]).sum() > 0                                 # the 0 * inf quirk is in the compiled shader's output
    z = np.load(os.path.join(FIXTURES, "cap_row_r0.npz"))
    assert (z["f1_albedo"][..., 3].view(np.uint32) == 0x80000000).sum() > 500     # capped primary rays


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_compiled_shaders_outputs(O, scenes, noise, name):
    """trace -> temporal -> denoise of every frame of the sequence through oracle/oshaders.cpp == the committed outputs of the
    reference's compiled modules, every bit of every image."""
    z = np.load(os.path.join(FIXTURES, name + ".npz"))
    got = SP.run_case(O, scenes, noise, spec_of(z), compiled=False)
    keys = [k for k in z.files if k.startswith("f") and k[1].isdigit()]
    assert sorted(keys) == sorted(got)
    for k in keys:
        raw_equal(got[k], z[k], f"{name} {k}")


@needs_reference
@pytest.mark.parametrize("name", CASES)
def test_fixtures_are_what_the_reference_modules_give(O, scenes, noise, name):
    """Provenance: the committed files equal a fresh run of /root/reference/shaders/*.comp.spv through the interpreter, and the case
    list of tests/spirv_pipeline.py is what they were made from."""
    z = np.load(os.path.join(FIXTURES, name + ".npz"))
    spec = SP.cases(scenes)[name]
    fresh = SP.run_case(O, scenes, noise, spec, compiled=True)
    for k, img in fresh.items():
        raw_equal(img, z[k], f"{name} {k}")
    again = spec_of(z)
    assert again["scene"] == spec["scene"] and (again["w"], again["h"], again["radius"]) == (spec["w"], spec["h"], spec["radius"])
    assert again["bounces"] == spec.get("bounces", SP.MAX_BOUNCES)
    for (p, d, f), (p2, d2, f2) in zip(again["frames"], spec["frames"]):
        assert np.array_equal(p, np.asarray(p2, np.float32)) and np.array_equal(d, np.asarray(d2, np.float32)) and f == float(np.float32(f2))


@needs_reference
def test_compiled_voxels_shader_equals_the_oracle_on_every_scene(O, scenes, noise):
    """All 15 scene files x 2 views at 240 x 135 (+ specular, sun-off and strongly emissive shading on two of them): ~2 M rays through the
    compiled voxels.comp and through the oracle, every output bit equal."""
    w, h = 240, 135
    names = sorted(f[:-4] for f in os.listdir(os.path.join(GOLDEN, "scenes")) if f.endswith(".npz"))
    assert len(names) == 15
    rays = 0
    for name in names:
        pos, mrgb, size = scenes.load_scene(name)
        octree = O.create_octree(pos, mrgb)
        for view, cam in (("bench", scenes.bench_camera(size)), ("close", scenes.close_camera(size))):
            for variant in ("default", "specular", "sun off", "emissive") if name in ("castle", "room") else ("default",):
                u = O.Uniforms.default()
                u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
                u.frame_number = 3 + len(name)
                if variant == "specular":
                    u.specularity = 0.4
                if variant == "sun off":
                    u.sun_strength = 0.0
                if variant == "emissive":
                    u.emit_strength = 9.0
                ref = O.trace(octree, noise, u, w, h, SP.MAX_BOUNCES, crop=(0, 0, w, h))
                got = SP.spirv_trace(O, octree, noise, u, w, h)
                for a, b, label in zip(got[:3], ref[:3], IMAGES):
                    raw_equal(a, b, f"{name} {view} {variant} {label}")
                rays += ref[3]
    assert rays > 2_000_000


@needs_reference
def test_compiled_shaders_equal_the_oracle_on_a_crop_of_the_bench_frame(O, scenes, noise):
    """BASELINE configs[1]'s frame (menger, 1920 x 1080, the bench camera) at full size: a 256 x 96 crop across the model's silhouette."""
    pos, mrgb, size = scenes.load_scene("menger")
    octree = O.create_octree(pos, mrgb)
    cam = scenes.bench_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], 1920, 1080))
    u.frame_number = 11
    crop = (832, 492, 1088, 588)
    ref = O.trace(octree, noise, u, 1920, 1080, SP.MAX_BOUNCES, crop=crop)
    got = SP.spirv_trace(O, octree, noise, u, 1920, 1080, crop=crop)
    assert (ref[1][..., 3] >= 0).mean() > 0.3
    for a, b, label in zip(got[:3], ref[:3], IMAGES):
        raw_equal(a, b, f"bench frame crop {label}")


@needs_reference
def test_compiled_temporal_and_denoise_equal_the_oracle(O, scenes, noise):
    """A panning camera over five frames (history accepted for most pixels, rejected at the silhouettes) with every denoise radius on
    the accumulated frames; then synthetic G-buffers with NaN, inf, negative depth and odd normals through both stages."""
    w, h = 112, 72
    pos, mrgb, size = scenes.load_scene("castle")
    octree = O.create_octree(pos, mrgb)
    p0, d0, fov = scenes.close_camera(size)
    a = SP.Pipeline(O, octree, noise, w, h, 1, compiled=True)
    b = SP.Pipeline(O, octree, noise, w, h, 1, compiled=False)
    for f in range(5):
        cam = (p0 + np.float32(0.03 * f) * np.array([1, 0.2, 0.1], np.float32), d0 + np.float32(0.01 * f) * np.array([0, 1, 0], np.float32), fov)
        ga, gb = a.render(cam), b.render(cam)
        for x, y, label in zip(ga, gb, ("colour", "nd", "albedo", "accum", "denoised")):
            raw_equal(x, y, f"moving camera frame {f + 1} {label}")
    hit = gb[1][..., 3] >= 0
    blend = gb[3][..., 3][hit]
    assert (blend < 0.2).mean() > 0.5 and (blend == 0.5).sum() > 20              # history reused for most pixels, refused for some
    color, nd, alb, accum, _ = gb
    cam16 = b.u.camera16()
    for r in range(0, 9):
        du = O.Denoise.default()
        du.radius = r
        raw_equal(SP.spirv_denoise(O, accum, nd, alb, cam16, du), O.denoise(accum, nd, alb, cam16, du), f"denoise radius {r}")
    # synthetic images: special values in every channel
    rng = np.random.default_rng(7)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-40, -1.0, 3e38, 0.5], np.float32)

    def salted(img, p):
        out = np.array(img)
        mask = rng.random(out.shape) < p
        out[mask] = rng.choice(specials, size=int(mask.sum()))
        return out
    for trial in range(3):
        c2, nd2, alb2, old_c, old_nd = salted(color, 0.02), salted(nd, 0.02), salted(alb, 0.01), salted(accum, 0.02), salted(nd, 0.02)
        tu = O.Temporal.default()
        acc_o = O.temporal(c2, nd2, old_c, old_nd, cam16, a.old_cam16, tu, True)
        acc_s = SP.spirv_temporal(O, c2, nd2, old_c, old_nd, cam16, a.old_cam16, tu)
        raw_equal(acc_s, acc_o, f"temporal on salted images, trial {trial}")
        du = O.Denoise.default()
        du.radius = 2 + trial
        raw_equal(SP.spirv_denoise(O, acc_o, nd2, alb2, cam16, du), O.denoise(acc_o, nd2, alb2, cam16, du), f"denoise on salted images, trial {trial}")


@needs_reference
def test_undefined_reads_matter_only_at_the_trip_cap(O, scenes, noise):
    """ORC_SPV_POISON fills every Function variable with a NaN pattern at each function entry.  Ordinary frames do not change by a bit
    (no output depends on a variable that was not written); on the cap scene exactly the pixels whose primary ray ends at the 2 048-trip
    cap change — `normal` is returned unwritten (voxels.comp:166-169, U1) — and only in the images that carry the normal."""
    for name in ("castle_moving_r2", "room_sun_off_r0", "zero_times_inf_r0"):
        spec = SP.cases(scenes)[name]
        pipe, _ = SP.build_case(O, scenes, noise, spec, compiled=True)
        cam = spec["frames"][0]
        pipe.u.frame_number = 1
        pipe.u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], spec["w"], spec["h"]))
        plain = SP.spirv_trace(O, pipe.octree, noise, pipe.u, spec["w"], spec["h"])
        poisoned = SP.spirv_trace(O, pipe.octree, noise, pipe.u, spec["w"], spec["h"], flags=O.SPV_POISON)
        for x, y, label in zip(plain[:3], poisoned[:3], IMAGES):
            raw_equal(x, y, f"{name} {label} with poisoned variables")
        nd, alb = plain[1], plain[2]
        acc = SP.spirv_temporal(O, plain[0], nd, np.zeros_like(nd), np.zeros_like(nd), pipe.u.camera16(), np.zeros(16, np.float32), pipe.tu)
        raw_equal(SP.spirv_temporal(O, plain[0], nd, np.zeros_like(nd), np.zeros_like(nd), pipe.u.camera16(), np.zeros(16, np.float32), pipe.tu,
                                    flags=O.SPV_POISON), acc, f"{name} temporal with poisoned variables")
        du = O.Denoise.default()
        du.radius = 2
        raw_equal(SP.spirv_denoise(O, acc, nd, alb, pipe.u.camera16(), du, flags=O.SPV_POISON), SP.spirv_denoise(O, acc, nd, alb, pipe.u.camera16(), du),
                  f"{name} denoise with poisoned variables")
    spec = SP.cases(scenes)["cap_row_r0"]
    pipe, _ = SP.build_case(O, scenes, noise, spec, compiled=True)
    cam = spec["frames"][0]
    pipe.u.frame_number = 1
    pipe.u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], spec["w"], spec["h"]))
    plain = SP.spirv_trace(O, pipe.octree, noise, pipe.u, spec["w"], spec["h"])
    poisoned = SP.spirv_trace(O, pipe.octree, noise, pipe.u, spec["w"], spec["h"], flags=O.SPV_POISON)
    capped = plain[2][..., 3].view(np.uint32) == 0x80000000
    assert capped.sum() > 500
    changed_nd = (plain[1].view(np.uint32) != poisoned[1].view(np.uint32)).any(axis=2)
    changed_color = (plain[0].view(np.uint32) != poisoned[0].view(np.uint32)).any(axis=2)
    assert np.array_equal(changed_nd, capped) and not (changed_color & ~capped).any()
    raw_equal(plain[2], poisoned[2], "albedo of the cap frame")
    assert (plain[1][capped][:, :3] == 0).all()                                   # zeroed memory = the oracle's definition of U1


@needs_reference
def test_interpreter_refuses_what_it_cannot_run_safely(O, scenes, noise):
    """The module is untrusted data: a wrong binding, a buffer too small for the index the shader computes, an image window that does
    not cover a pixel, a truncated or foreign word stream — all end in SpirvError, none in a crash or a guess."""
    pos, mrgb, size = scenes.load_scene("3x3x3")
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], 32, 32))
    mod = SP.module("voxels")
    out = [np.zeros((8, 8, 4), np.float32) for _ in range(3)]

    def bindings(octree_words=octree, noise_table=noise, window=(0, 0), skip=None):
        b = [O.spirv_image(k, out[k], (32, 32), window) for k in range(3)]
        b += [O.spirv_buffer(3, SP._block(u, 160)), O.spirv_buffer(4, np.zeros(64, np.uint8)), O.spirv_buffer(5, octree_words), O.spirv_buffer(6, noise_table)]
        return [x for x in b if x[0] != skip]
    assert O.spirv_dispatch(mod, bindings(), 0, 0, 8, 8) > 0
    with pytest.raises(O.SpirvError, match="nothing bound"):
        O.spirv_dispatch(mod, bindings(skip=5), 0, 0, 8, 8)
    with pytest.raises(O.SpirvError, match="beyond the bound buffer|outside the bound memory"):
        O.spirv_dispatch(mod, bindings(octree_words=octree[:9]), 12, 12, 20, 20)           # the root's children are gone
    with pytest.raises(O.SpirvError, match="beyond the bound buffer|outside the bound memory"):
        O.spirv_dispatch(mod, bindings(noise_table=noise[:4096], window=(12, 12)), 12, 12, 20, 20)
    with pytest.raises(O.SpirvError, match="window"):
        O.spirv_dispatch(mod, bindings(), 4, 4, 12, 12)                                    # pixels beyond the 8 x 8 window that was bound
    with pytest.raises(O.SpirvError):
        O.spirv_dispatch(mod[: len(mod) // 2 - 2], bindings(), 0, 0, 8, 8)                  # cut inside a word
    with pytest.raises(O.SpirvError):
        O.spirv_dispatch(mod[: len(mod) // 8 * 4], bindings(), 0, 0, 8, 8)                  # cut between two words: half a module
    with pytest.raises(O.SpirvError, match="not a SPIR-V module"):
        O.spirv_dispatch(b"\0" * 64, bindings(), 0, 0, 8, 8)
    words = np.frombuffer(mod, "<u4").copy()
    rng = np.random.default_rng(3)
    refused = 0
    for trial in range(40):                                                                 # flipped words: refused or run to an end, never a crash
        w2 = words.copy()
        at = rng.integers(5, len(w2), size=3)
        w2[at] ^= rng.integers(1, 1 << 16, size=3).astype(np.uint32)
        try:
            O.spirv_dispatch(w2.tobytes(), bindings(), 0, 0, 4, 4)
        except O.SpirvError:
            refused += 1
    assert refused > 0


def test_interpreter_rejects_garbage_without_the_reference(O):
    for bad in (b"", b"\0" * 20, b"\x03\x02\x23\x07" + b"\0" * 16, b"\x03\x02\x23\x07" + b"\0\0\1\0" + b"\0" * 4 + b"\x10\0\0\0" + b"\0" * 4 + b"\x05\0\x09\0"):
        with pytest.raises(O.SpirvError):
            O.spirv_dispatch(bad + b"\0" * (-len(bad) % 4), [], 0, 0, 1, 1)


@needs_reference
def test_interpreter_under_sanitizers_on_mutated_modules(tmp_path):
    """oracle/ospirv.cpp built with -fsanitize=address,undefined and fed the three modules with 1-4 words flipped per trial (another id,
    another opcode, a random word, a truncation) over small synthetic bindings: every trial ends in a status, no sanitizer report."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "asan_spirv_driver")
    src = [os.path.join(ROOT, "tests", "asan_spirv_driver.cpp")] + [os.path.join(ROOT, "oracle", f) for f in ("ospirv.cpp", "oshaders.cpp", "oprocedural.cpp", "ovox.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-pthread",
                            "-I" + os.path.join(ROOT, "oracle")] + src + ["-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no sanitizer runtime on this host")
    assert build.returncode == 0, build.stderr[-2000:]
    for name, kinds in (("voxels", "1110000"), ("temporal", "321121000"), ("denoise", "111100")):
        run = subprocess.run([exe, os.path.join(SP.SHADERS, name + ".comp.spv"), "700", str(17 + len(name)), kinds], capture_output=True, text=True,
                             env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"}, timeout=600)
        assert run.returncode == 0, (name, run.stdout[-500:], run.stderr[-3000:])
        ran = int(run.stdout.split("ran to the end")[1].split(",")[0])
        assert ran >= 20, run.stdout                                  # the harness does run modules, not only refuse them


@needs_reference
def test_respecialised_loop_bound(O, scenes, noise):
    """with_max_bounces changes ONE operand of the module: asked for the module's own 3 it reproduces the module's outputs; with 1 a path ends
    after its first hit's sun ray; and the module grows by exactly one 4-word OpConstant."""
    pos, mrgb, size = scenes.load_scene("castle")
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], 64, 40))
    u.frame_number = 2
    own = SP.spirv_trace(O, octree, noise, u, 64, 40)
    mod = SP.module("voxels")
    assert len(SP.with_max_bounces(mod, 5)) == len(mod) + 16
    out = [np.zeros((40, 64, 4), np.float32) for _ in range(3)]
    b = [O.spirv_image(k, out[k]) for k in range(3)] + [O.spirv_buffer(3, SP._block(u, 160)), O.spirv_buffer(4, np.zeros(64, np.uint8)),
                                                        O.spirv_buffer(5, octree), O.spirv_buffer(6, noise)]
    O.spirv_dispatch(SP.with_max_bounces(mod, 3), b, 0, 0, 64, 40)
    for x, y, label in zip(out, own[:3], IMAGES):
        raw_equal(x, y, f"bound 3 by re-specialisation: {label}")
    for bounces in (1, 2, 4, 8):
        ref = O.trace(octree, noise, u, 64, 40, bounces, crop=(0, 0, 64, 40))
        got = SP.spirv_trace(O, octree, noise, u, 64, 40, bounces=bounces)
        for x, y, label in zip(got[:3], ref[:3], IMAGES):
            raw_equal(x, y, f"{bounces} bounces: {label}")


def _full_size():
    import json
    with open(os.path.join(FIXTURES, "full_size.json")) as f:
        return json.load(f)["cases"]


def test_oracle_reproduces_the_full_size_known_answers(O, scenes, noise):
    """BASELINE's whole frames — configs[1] (menger, 1920 x 1080, 4 bounces: the bench frame) and the frames of configs 3 and 4 (monu10 and
    castle at 3840 x 2160, 8 bounces) — through the oracle: every slab of rows of all three images hashes to what the reference's
    compiled shader gave (tests/golden/spirv_exec/full_size.json; 18.7 M pixels, 226 M output values in all)."""
    cases = _full_size()
    assert [c["name"] for c in cases] == [c["name"] for c in SP.FULL_SIZE]
    for c in cases:
        n = (c["h"] + c["slab_rows"] - 1) // c["slab_rows"]
        assert c["slab_rows"] == SP.SLAB_ROWS and all(len(c["sha256"][k]) == n for k in ("color", "nd", "albedo"))
        got = SP.full_size_slab_hashes(O, scenes, noise, c, range(n), compiled=False)
        for s in range(n):
            for k in ("color", "nd", "albedo"):
                assert got[s][k] == c["sha256"][k][s], f"{c['name']}: rows {s * SP.SLAB_ROWS}-{(s + 1) * SP.SLAB_ROWS} of the {k} image"


@needs_reference
def test_full_size_known_answers_are_what_the_module_gives(O, scenes, noise):
    """Provenance of full_size.json: slabs of every frame regenerated from the reference's module (the three middle slabs of configs[1]'s
    frame — the model — and one of each 4K frame; tests/diag_spirv_full_frames.py and the fixture script run the whole frames)."""
    for c in _full_size():
        n = (c["h"] + c["slab_rows"] - 1) // c["slab_rows"]
        slabs = (3, 4, 5) if c["h"] == 1080 else (n // 2 - 1,)
        fresh = SP.full_size_slab_hashes(O, scenes, noise, c, slabs, compiled=True)
        for s in slabs:
            for k in ("color", "nd", "albedo"):
                assert fresh[s][k] == c["sha256"][k][s], (c["name"], s, k)


def _frame_loop():
    import json
    with open(os.path.join(FIXTURES, "full_size.json")) as f:
        return json.load(f)["frame_loop"]


def test_oracle_reproduces_the_full_size_frame_loop(O, scenes, noise):
    """Config 3's frame loop at its size (monu10, 3840 x 2160, 8 bounces): two frames through trace, temporal and denoise — frame 2 blends
    frame 1's history — hash to what the reference's three compiled modules gave: accumulated colour of both frames, frame 2 denoised with
    the 5 x 5 window (whole frame) and the 17 x 17 window (a strip of rows)."""
    z = _frame_loop()
    assert z["name"] == SP.PIPELINE_CASE["name"] and tuple(z["r8_rows"]) == SP.R8_ROWS and z["slab_rows"] == SP.SLAB_ROWS
    got = SP.pipeline_hashes(O, scenes, noise, compiled=False)
    assert sorted(got) == sorted(z["sha256"])
    for k, want in z["sha256"].items():
        assert got[k] == want, k


@needs_reference
def test_full_size_frame_loop_hashes_are_what_the_modules_give(O, scenes, noise):
    """Provenance of the frame-loop hashes.  The trace stage's whole frames are covered by the per-slab hashes above; here the temporal
    and denoise MODULES run on frame 2's inputs (the oracle's images, which those hashes say are the modules' own): temporal over the
    whole frame, denoise with radius 2 on one slab and with radius 8 on its strip."""
    z = _frame_loop()
    pos, mrgb, cam, _ = SP.full_size_uniforms(O, scenes, SP.PIPELINE_CASE)
    pipe = SP.Pipeline(O, O.create_octree(pos, mrgb), noise, z["w"], z["h"], 2, compiled=False, bounces=z["bounces"])
    _, nd1, _, acc1, _ = pipe.render(cam)
    old_cam16 = pipe.old_cam16.copy()
    color, nd, alb, acc2, _ = pipe.render(cam)
    cam16 = pipe.u.camera16()
    assert SP.slab_hashes(acc2) == z["sha256"]["f2_accum"]
    fresh = SP.spirv_temporal(O, color, nd, acc1, nd1, cam16, old_cam16, pipe.tu)
    assert SP.slab_hashes(fresh) == z["sha256"]["f2_accum"]
    du = O.Denoise.default()
    du.radius = 2
    for s in (8,):
        rows = SP.spirv_denoise_rows(O, acc2, nd, alb, cam16, du, s * SP.SLAB_ROWS, (s + 1) * SP.SLAB_ROWS)
        assert SP.canonical_sha256(rows) == z["sha256"]["f2_denoised_r2"][s], s
    du.radius = 8
    assert SP.canonical_sha256(SP.spirv_denoise_rows(O, acc2, nd, alb, cam16, du, *SP.R8_ROWS)) == z["sha256"]["f2_denoised_r8_rows"]


@needs_reference
def test_compiled_voxels_shader_equals_the_oracle_from_random_cameras(O, scenes, noise):
    """Cameras drawn at random — outside the model, INSIDE its box (rays that start inside the tree, some inside voxels), axis-aligned
    views whose rays have zero direction components — with random shading parameters and bounce counts, on five scenes: the compiled
    module and the oracle agree on every bit."""
    import zlib
    w, h = 96, 60
    for name in ("menger", "room", "castle", "nature", "3x3x3"):
        pos, mrgb, size = scenes.load_scene(name)
        octree = O.create_octree(pos, mrgb)
        ext = scenes.world_extent(size)
        centre = ext * np.float32(0.5)
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        inside = 0
        for i in range(10):
            p = (centre + ext.max() * rng.uniform(-1.1, 1.1, 3)).astype(np.float32)
            if i % 3 == 1:
                p = (centre + ext * rng.uniform(-0.45, 0.45, 3)).astype(np.float32)
                inside += 1
            d = (centre + ext * rng.uniform(-0.3, 0.3, 3) - p).astype(np.float32)
            if i % 5 == 2:
                d = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]][(i // 5) % 3], np.float32)
                p = np.round(p)                                   # integer coordinates: rays along node mid-planes
            u = O.Uniforms.default()
            u.specularity = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
            u.sun_strength = float(rng.choice([4.0, 4.0, 0.0]))
            u.sun_size = float(rng.choice([0.05, 0.3]))
            u.sun_yaw, u.sun_pitch = float(rng.uniform(0, 6.28)), float(rng.uniform(0.1, 1.5))
            u.emit_strength = float(rng.choice([4.0, 0.0, 20.0]))
            u.frame_number = int(rng.integers(1, 5000))
            bounces = int(rng.choice([1, 3, 3, 5]))
            u.set_camera(p, O.camera_axis_scaled(p, d, scenes.FOV_70, w, h))
            ref = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
            got = SP.spirv_trace(O, octree, noise, u, w, h, bounces=bounces)
            for a, b, label in zip(got[:3], ref[:3], IMAGES):
                raw_equal(a, b, f"{name} camera {i} at {p.tolist()} looking {d.tolist()}, {bounces} bounces: {label}")
        assert inside >= 3


@needs_reference
@pytest.mark.parametrize("name", ["castle", "menger", "nature"])
def test_compiled_traversal_over_the_built_octree_finds_the_voxel_lists_first_hits(O, scenes, noise, name):
    """The octree BUILDER (src/context.rs:710-834, restated; Rust: no compiled form to run) is held by known answers only — but its output
    is what the reference's compiled traversal reads.  Primary rays of a frame through the compiled voxels.comp over the built buffer
    against an independent binary64 DDA over the plain voxel list (oracle/odda.cpp): the same hit or miss, the same voxel, the same
    entry face, and the leaf word the shader returns is that voxel's material and colour — the layout `nodes[8 * node + octant]`, the
    leaf and emittance bits and the header are the ones the compiled shader expects."""
    from test_oracle_traversal import dense_grid
    w, h = 240, 135
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    basis = O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h)
    u.set_camera(cam[0], basis)
    u.frame_number = 1
    _, nd, alb, _ = SP.spirv_trace(O, octree, noise, u, w, h)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d = xs[..., None] * basis[0:3] - ys[..., None] * basis[3:6] + basis[6:9]
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
    o = np.tile(np.asarray(cam[0], np.float32), (w * h, 1))
    grid, base = dense_grid(pos)
    dhit, dt, daxis, dcell = O.dda_cast(grid, base, o, d)
    hit = (nd[..., 3] >= 0).ravel()
    assert hit.sum() > 5000 and (hit == dhit).mean() > 0.9995
    both = hit & dhit
    t = nd[..., 3].ravel()
    assert np.quantile(np.abs(t[both] - dt[both]) / np.maximum(dt[both], 1e-3), 0.999) < 1e-4
    normal = nd[..., :3].reshape(-1, 3)
    assert (np.argmax(np.abs(normal[both]), 1) == daxis[both]).mean() > 0.9995
    # the voxel the DDA found, looked up in the voxel list, against the leaf word the shader wrote (intBitsToFloat, voxels.comp:396)
    lut = {tuple(p): k for k, p in enumerate(pos.astype(np.int32).tolist())}
    word = alb[..., 3].ravel().view(np.uint32)
    idx = np.nonzero(both)[0][::7]
    want = np.array([lut[tuple(dcell[i])] for i in idx])
    m, r, g, b = (mrgb[want, k].astype(np.uint32) for k in range(4))
    expect = 0x80000000 | ((m & 0x7f) << 24) | (r << 16) | (g << 8) | b
    assert (word[idx] == expect).mean() > 0.9995


def test_oracle_reproduces_the_scene_sweep(O, scenes, noise):
    """Every scene file of the reference (15), its close view at 64 x 40: the three trace images of the compiled shader, from the oracle."""
    z = np.load(os.path.join(FIXTURES, "scene_sweep.npz"))
    names = SP.sweep_scenes()
    assert len(names) == 15 and (int(z["w"]), int(z["h"]), int(z["max_bounces"])) == (SP.SWEEP["w"], SP.SWEEP["h"], SP.MAX_BOUNCES)
    hits = 0
    for name in names:
        for key, img in zip(("color", "nd", "albedo"), SP.sweep_frame(O, scenes, noise, name, compiled=False)):
            raw_equal(img, z[f"{name}_{key}"], f"{name} {key}")
        hits += int((z[f"{name}_nd"][..., 3] >= 0).sum())
    assert hits > 15 * 200


@needs_reference
def test_scene_sweep_is_what_the_module_gives(O, scenes, noise):
    z = np.load(os.path.join(FIXTURES, "scene_sweep.npz"))
    for name in SP.sweep_scenes():
        for key, img in zip(("color", "nd", "albedo"), SP.sweep_frame(O, scenes, noise, name, compiled=True)):
            raw_equal(img, z[f"{name}_{key}"], f"{name} {key}")


@needs_reference
def test_compiled_voxels_shader_equals_the_oracle_on_edge_scenes(O, noise):
    """The scene format's corners through the compiled module: no voxels at all (a root of eight empty slots), one voxel, one voxel in the
    negative octant, 6 000 random voxels on both sides of the origin, a depth-10 slab far from it, and the deepest tree int16 coordinates
    allow (depth 15 = MAX_DEPTH - 1 frames of the shader's stack, voxels.comp:3,127-130) at both extremes."""
    f32 = np.float32
    rng = np.random.default_rng(42)
    xs, ys = np.meshgrid(np.arange(900, 960), np.arange(900, 960))
    far = np.stack([xs.ravel(), ys.ravel(), 900 + (xs.ravel() * 7 + ys.ravel() * 3) % 5], 1).astype(np.int16)
    cases = [
        (np.zeros((0, 3), np.int16), np.zeros((0, 4), np.uint8), (np.array([3, 2, -4], f32), np.array([-3, -2, 4], f32), 1.0), 96, 64),
        (np.array([[0, 0, 0]], np.int16), np.array([[0x40, 200, 100, 50]], np.uint8), (np.array([3, 2, -4], f32), np.array([-3, -2, 4], f32), 1.0), 96, 64),
        (np.array([[-1, -1, -1]], np.int16), np.array([[0, 1, 2, 3]], np.uint8), (np.array([3, 2, -4], f32), np.array([-3, -2, 4], f32), 1.0), 96, 64),
        (rng.integers(-40, 40, (6000, 3)).astype(np.int16), rng.integers(0, 256, (6000, 4)).astype(np.uint8),
         (np.array([45, 30, -50], f32), np.array([-45, -30, 50], f32), 1.1), 160, 96),
        (far, np.tile(np.array([[0, 255, 128, 0]], np.uint8), (len(far), 1)), (np.array([462, 470, 438], f32), np.array([0.1, -0.25, 1.0], f32), 1.2), 128, 96),
    ]
    for extreme in (32767, -32768):
        cases.append((np.array([[extreme, 3, -2], [0, 0, 0]], np.int16), np.array([[0, 10, 200, 30], [0x40, 255, 255, 255]], np.uint8),
                      (np.array([extreme / 2 + (3 if extreme > 0 else -3), 4, -6], f32), np.array([-0.5 if extreme > 0 else 0.5, -0.4, 1.0], f32), 0.9), 64, 48))
    depths = []
    for pos, mrgb, cam, w, h in cases:
        octree = O.create_octree(pos, mrgb)
        depths.append(O.voxel_depth(pos) if len(pos) else 0)
        u = O.Uniforms.default()
        u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
        for frame in (1, 7):
            u.frame_number = frame
            ref = O.trace(octree, noise, u, w, h, SP.MAX_BOUNCES, crop=(0, 0, w, h))
            got = SP.spirv_trace(O, octree, noise, u, w, h)
            for a, b, label in zip(got[:3], ref[:3], IMAGES):
                raw_equal(a, b, f"{len(pos)} voxels, depth {depths[-1]}, frame {frame}: {label}")
    assert 10 in depths and depths.count(15) == 2
