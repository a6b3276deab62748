"""CPU: how far can an image move between two CONFORMING implementations of the reference's shaders?  

Parity of this build is "bit-exact against an oracle the reference cannot pin" (DESIGN.md section 2): the reference holds no expected
output, and three families of values are left to the driver by GLSL / Vulkan — U6 the precision of sin cos tan exp log pow and the
method of normalize(), U4 the sampler's sub-texel precision, U5 how inverse(mat4) is evaluated.  The oracle and the kernels make ONE
choice for each (include/vxrt_detmath.h, oracle/oshaders.cpp) and agree with each other by construction.  This test builds the oracle a
SECOND way — `make -C oracle alt`: the same restatement with each choice made the other way, switched at run time (oracle/ovec.h) —

    ALT_LIBM       sin cos tan exp log pow from binary64 libm, rounded once          (instead of the header's binary32 polynomials)
    ALT_NORMALIZE  normalize(v) = v * inversesqrt(dot(v, v))                         (instead of the true division v / length(v))
    ALT_SAMPLER    bilinear weights at full binary32 precision                       (instead of 8 fractional bits)
    ALT_INVERSE    the camera inverse by adjugate / determinant in binary32          (instead of binary64 rounded once)

and compares it with the shipped oracle on every BASELINE config's frame (crops at the configs' own sizes, spp and bounce counts) and on
a moving-camera pipeline.  What it reports per case — and what `python tests/test_oracle_builtin_sensitivity.py` writes to
profiles/r05/builtin_sensitivity.json for DESIGN.md — is the image-level RMSE and maximum error of the radiance, the pixels whose PATH
changed (a hit became a miss somewhere along it, or another voxel was hit: |error| > 1e-3), the pixels whose primary hit flipped, and
the RMSE over the remaining pixels.

The finding (thresholds below are set from what was measured, with head room):
 * the built-ins' own rounding is harmless where nothing flips: over pixels whose path did not change the two oracles agree to ~1e-7;
 * but a path tracer that offsets its bounce origins by 1e-5 (voxels.comp:323,339) turns ONE differently rounded operation into a
   different path for a few pixels in 10^5 (configs 2-4: coordinates <= 64) and for 4 % of the pixels of config 5 (coordinates up to
   1024, where the offset is below half an ulp), each of them wrong by O(1).  With the `normalize` lowering alone the whole-image RMSE is
   2.6e-3 on config 2's frame, 5.7e-3 on config 3's crop and 0.2 on config 5's (config 4's 768 000-pixel crop happens to hold no such
   pixel; three 1.5 M-pixel runs of it held 3-4): north_star's "per-pixel RMSE <= 1e-3 against a fixed-seed CPU reference" cannot be
   met between ANY two implementations that differ in one rounding, the reference's own GPU driver and its own CPU included.  Bit-exact
   against one fully specified restatement is the only checkable contract, which is what tests/ hold the kernels to; what it cannot
   say — which of the conforming images the reference's driver would have produced — is bounded by these numbers.
What this measures is the part of the contract that NO output of the reference could pin: tests/test_oracle_spirv_exec.py holds the oracle to
the reference's compiled shaders for everything SPIR-V defines; the built-ins varied here are what SPIR-V leaves to a driver."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def compare(a, b):
    """a, b = (colour, normal/depth, albedo/node, rays) of the shipped oracle and of the alternative."""
    ca, cb = a[0][..., :3].astype(np.float64), b[0][..., :3].astype(np.float64)
    fin = np.isfinite(ca).all(-1) & np.isfinite(cb).all(-1)
    err = np.abs(ca - cb).max(-1)
    changed = fin & (err > 1e-3)            # a decision along the path changed (a rounding-only difference is ~1e-7)
    d, kept = (ca - cb)[fin], (ca - cb)[fin & ~changed]
    return {"pixels": int(fin.sum()), "rmse": float(np.sqrt((d ** 2).mean())), "max_abs": float(np.abs(d).max()),
            "pixels_whose_path_changed": int(changed.sum()), "fraction_changed": float(changed.sum() / max(fin.sum(), 1)),
            "rmse_over_unchanged_pixels": float(np.sqrt((kept ** 2).mean())) if kept.size else 0.0,
            "primary_hit_flips": int(((a[1][..., 3] >= 0) != (b[1][..., 3] >= 0)).sum()),
            "first_voxel_differs": int((a[2][..., 3].view(np.uint32) != b[2][..., 3].view(np.uint32)).sum()),
            "rays": [int(a[3]), int(b[3])]}


def mean_of_samples(trace_one, spp):
    total, rays, nd, alb = None, 0, None, None
    for f in range(1, spp + 1):
        col, nd, alb, r = trace_one(f)
        total = col.copy() if total is None else (total + col).astype(np.float32)
        rays += r
    return ((total / np.float32(spp)).astype(np.float32) if spp > 1 else total), nd, alb, rays


def config_frame(O, scenes, noise, name, w, h, bounces, spp, camera, crop):
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    cam = camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))     # the camera basis goes through vx_tan / normalize too

    def one(f):
        u.frame_number = f
        return O.trace(octree, noise, u, w, h, bounces, crop=crop)
    return mean_of_samples(one, spp)


def config5_frame(O, scenes, noise, crop, spp):
    level, clip, mrgb, period = scenes.CONFIG5
    cam = scenes.config5_cameras()["outside"]
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], 7680, 4320))

    def one(f):
        u.frame_number = f
        return O.trace_menger(level, clip, mrgb, period, noise, u, 8, crop)
    return mean_of_samples(one, spp)


def moving_camera_pipeline(O, scenes, noise, frames=4, w=480, h=270, bounces=3, radius=2):
    """trace -> temporal -> denoise with the camera panning and drifting: the sampler's weights (U4) and the camera inverse (U5) only
    matter when the reprojection leaves the texel centres.  -> (accumulated colour, nd, albedo, rays) and the denoised frame."""
    pos, mrgb, size = scenes.load_scene("castle")
    p0, d0, fov = scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    du = O.Denoise.default()
    du.radius = radius
    old_c, old_nd, old16, rays = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32), None, 0
    for k in range(frames):
        p = (p0 + np.float32(0.013 * k) * np.array([1.0, 0.4, -0.3], np.float32)).astype(np.float32)
        d = (d0 + np.float32(0.004 * k * float(np.linalg.norm(d0))) * np.array([0.3, -1.0, 0.2], np.float32)).astype(np.float32)
        u = O.Uniforms.default()
        u.set_camera(p, O.camera_axis_scaled(p, d, fov, w, h))
        u.frame_number = k + 1
        cam16 = u.camera16()
        color, nd, alb, r = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
        rays += r
        accum = O.temporal(color, nd, old_c, old_nd, cam16, old16 if old16 is not None else cam16, O.Temporal.default(), k > 0)
        old_c, old_nd, old16 = accum, nd, cam16
    den = O.denoise(accum, nd, alb, cam16, du)
    return (accum, nd, alb, rays), (den, nd, alb, rays)


CASES = {   # name -> (runner, the whole-image RMSE an alternative may reach, the fraction of pixels whose path may change)
    "config 2: menger 1920x1080, 1 spp, 4 bounces, whole frame":
        (lambda O, s, n: config_frame(O, s, n, "menger", 1920, 1080, 4, 1, s.bench_camera, (0, 0, 1920, 1080)), 2e-2, 1e-4),
    "config 3: monu10 3840x2160, 4 spp, 8 bounces, rows 900-1200":
        (lambda O, s, n: config_frame(O, s, n, "monu10", 3840, 2160, 8, 4, s.bench_camera, (0, 900, 3840, 1200)), 2e-2, 5e-4),
    "config 4: castle 3840x2160 close up, 4 spp, 8 bounces, rows 1000-1200":
        (lambda O, s, n: config_frame(O, s, n, "castle", 3840, 2160, 8, 4, s.close_camera, (0, 1000, 3840, 1200)), 2e-2, 1e-4),
    "config 5: 2048^3 procedural Menger 7680x4320, 2 of 16 spp, 8 bounces, 4000x32 crop":
        (lambda O, s, n: config5_frame(O, s, n, (2000, 2000, 6000, 2032), 2), 0.5, 0.12),
}
MASKS = ("ALT_LIBM", "ALT_NORMALIZE", "ALT_ALL")


def measure(O, scenes, noise, names=None):
    out = {}
    for name, (run, _, _) in CASES.items():
        if names and name not in names:
            continue
        base = run(O, scenes, noise)
        with O.alt_builtins(0):
            same = run(O, scenes, noise)
        out[name] = {"mask 0 (the alternative build with nothing switched)": compare(base, same)}
        for m in MASKS:
            with O.alt_builtins(getattr(O, m)):
                out[name][m] = compare(base, run(O, scenes, noise))
        with O.contracted():      # U9: a * b + c fused wherever the compiler can (the modules carry no NoContraction decoration)
            out[name]["CONTRACTED"] = compare(base, run(O, scenes, noise))
    if not names:
        base = moving_camera_pipeline(O, scenes, noise)
        out["moving camera: castle 480x270, 4 frames, trace + temporal + denoise r = 2"] = pipe = {}
        for m in ("ALT_SAMPLER", "ALT_INVERSE", "ALT_SAMPLER | ALT_INVERSE", "ALT_ALL"):
            with O.alt_builtins(sum(getattr(O, k.strip()) for k in m.split("|"))):
                alt = moving_camera_pipeline(O, scenes, noise)
            pipe[m] = {"accumulated": compare(base[0], alt[0]), "denoised": compare(base[1], alt[1])}
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_image_moves_by_path_flips_not_by_rounding(O, scenes, noise, name):
    r = measure(O, scenes, noise, names=[name])[name]
    _, rmse_cap, flip_cap = CASES[name]
    zero = r["mask 0 (the alternative build with nothing switched)"]
    assert zero["rmse"] == 0.0 and zero["rays"][0] == zero["rays"][1]          # the second build IS the oracle until a choice is switched
    for m in MASKS:
        c = r[m]
        assert c["primary_hit_flips"] == 0 and c["first_voxel_differs"] == 0    # primary rays: never (their hit is decided far from any edge case)
        # where no decision changed the built-ins' rounding is ~1e-7 (config 5: 1e-5 — coordinates of 10^3 and, among 6 % changed paths,
        # some whose change stays below the 1e-3 that defines "changed")
        assert c["rmse_over_unchanged_pixels"] < (5e-5 if "config 5" in name else 2e-6), (m, c)
        assert c["fraction_changed"] <= flip_cap, (m, c)
        assert c["rmse"] <= rmse_cap, (m, c)
    # the finding: one differently rounded built-in breaks north_star's RMSE <= 1e-3 through a handful of changed paths
    worst = max(r[m]["rmse"] for m in MASKS)
    if "config 5" in name:
        assert worst > 1e-2 and r["ALT_NORMALIZE"]["fraction_changed"] > 0.01    # the 1e-5 offset is below half an ulp at |x| ~ 1000
    assert r["ALT_LIBM"]["rmse"] <= r["ALT_NORMALIZE"]["rmse"] + 1e-9 or r["ALT_LIBM"]["pixels_whose_path_changed"] > 0


@pytest.mark.parametrize("name", [n for n in CASES if "config 5" not in n])
def test_contraction_is_one_more_freedom_of_the_same_kind(O, scenes, noise, name):
    """U9.  The reference's SPIR-V carries no NoContraction decoration, so its driver may fuse a * b + c; the oracle and the kernels never do
    (-ffp-contract=off, include/vxrt_detmath.h).  The oracle compiled with -ffp-contract=fast moves the image exactly as a differently
    rounded built-in does: ~1e-7 where no decision changes, a handful of pixels whose path changes, no primary ray."""
    run, rmse_cap, flip_cap = CASES[name]
    base = run(O, scenes, noise)
    with O.contracted():
        c = compare(base, run(O, scenes, noise))
    assert c["primary_hit_flips"] == 0 and c["first_voxel_differs"] == 0
    assert c["rmse_over_unchanged_pixels"] < 2e-6 and c["fraction_changed"] <= flip_cap and c["rmse"] <= rmse_cap, c
    assert c["rmse_over_unchanged_pixels"] > 0.0                                   # the fused build really computes differently


def test_sampler_and_inverse_choices_under_a_moving_camera(O, scenes, noise):
    base = moving_camera_pipeline(O, scenes, noise)
    with O.alt_builtins(0):
        same = moving_camera_pipeline(O, scenes, noise)
    assert compare(base[0], same[0])["rmse"] == 0.0 and compare(base[1], same[1])["rmse"] == 0.0
    for mask in (O.ALT_SAMPLER, O.ALT_INVERSE, O.ALT_SAMPLER | O.ALT_INVERSE):
        with O.alt_builtins(mask):
            alt = moving_camera_pipeline(O, scenes, noise)
        acc, den = compare(base[0], alt[0]), compare(base[1], alt[1])
        assert acc["rays"][0] == acc["rays"][1]                                  # neither choice touches the trace stage
        # U4 / U5 move the reprojected texel coordinate by ~1e-5 of a texel and the weights by < 2^-9: a smooth change of the history
        # colour, except where it tips the same-position test of temporal.comp:111 for a pixel (history dropped or kept)
        assert acc["rmse"] < 5e-3 and den["rmse"] < 5e-3, (mask, acc, den)
        assert acc["fraction_changed"] < 0.02, (mask, acc)


if __name__ == "__main__":
    from gpu_voxel_raytracer_amd import scenes as S
    from oracle import oracle as OR
    OR.build()
    res = measure(OR, S, OR.noise_table())
    path = os.path.join(ROOT, "profiles", "r05", "builtin_sensitivity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"what": __doc__.split("\n\n")[0], "made_by": "python tests/test_oracle_builtin_sensitivity.py", "cases": res}, f, indent=1)
    for name, r in res.items():
        print(name)
        for m, c in r.items():
            print("   ", m, json.dumps(c))
