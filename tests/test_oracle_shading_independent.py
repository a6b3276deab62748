"""CPU: a SECOND, independent restatement of the shading of shaders/voxels.comp:253-397 — written here from the shader's text in
numpy binary64, sharing no code with oracle/oshaders.cpp — run on the casts the oracle logs for a pixel (orc_trace_pixel_log).
What it pins: the blue-noise indexing and the ORDER of rand() draws (voxels.comp:268-275, 326, 342, 346-347, 278-280), the sun
jitter frame, the hemisphere flip, which rays are cast from where, the radiance bookkeeping (demodulated first albedo, emission,
`/ ambient_rays`) and the three outputs.  What it cannot pin: rounding (binary64 here, the contract's binary32 there: compared to
1e-4) and the octree walk itself (taken from the log; the walk has its own independent check, tests/test_oracle_traversal.py).
The reference ships nothing to compare against: parity stays unpinned by the reference; this narrows what a
transcription error in the oracle could hide in."""
import numpy as np
import pytest

N_LAYER, N_TOTAL = 128 * 128, 128 * 128 * 512
EMIT_BIT = 1 << 30


def norm(v):
    return v / np.sqrt((v * v).sum())


def rgb_of(node):
    return np.array([(node >> 16) & 0xff, (node >> 8) & 0xff, node & 0xff], np.float64)


def shade_pixel(noise, u, x, y, bounces, casts):
    """voxels.comp main() for pixel (x, y), the casts' results taken from `casts` (rows: origin, dir, hit, time, node bits, normal).
    Returns (colour, first normal, first time, first node, casts used); raises AssertionError when the oracle cast a different ray."""
    state = {"i": x % 128 + (y % 128) * 128 + (int(u.frame_number) % 512) * N_LAYER}

    def rand():
        state["i"] = (state["i"] + N_LAYER) % N_TOTAL
        return float(noise[state["i"]])

    sun_dir = np.array([np.cos(u.sun_yaw) * np.cos(u.sun_pitch), -np.sin(u.sun_pitch), np.sin(u.sun_yaw) * np.cos(u.sun_pitch)])
    sun_color = u.sun_strength * np.array(u.sun_color[:3], np.float64)          # SUN_COLOR, voxels.comp:6
    sky = np.array(u.sky_color[:3], np.float64)
    ray_o = np.array(u.camera_origin[:3], np.float64)
    ray_d = norm(x * np.array(u.camera_right[:3], np.float64) - y * np.array(u.camera_up[:3], np.float64) + np.array(u.camera_forward[:3], np.float64))
    first = (np.full(3, 2.0 ** 30), -1.0, 0xffffff)
    sample, blend, ambient, k, fragile = np.zeros(3), np.ones(3), 1, 0, False

    def next_cast(o, d):
        nonlocal k
        row = casts[k]
        k += 1
        assert np.allclose(row[0:3], o, rtol=2e-5, atol=2e-5), ("origin", k, row[0:3], o)
        assert np.allclose(row[3:6], d, rtol=0, atol=3e-5), ("direction", k, row[3:6], d)
        return bool(row[6]), float(row[7]), int(row[8:9].view(np.int32)[0]), row[9:12].astype(np.float64)

    for bounce in range(bounces):
        hit, t, node, n = next_cast(ray_o, ray_d)
        if not hit:
            if bounce == 0:
                blend = np.ones(3)
                sun_power = max(0.0, float(ray_d @ norm(-sun_dir))) ** (1.0 / u.sun_size ** 2)
                sample = sample + (sky + sun_color * sun_power) * blend
            else:
                sample = sample + sky * blend
            break
        hit_pos = ray_o + ray_d * t
        color = np.ones(3) if bounce == 0 else rgb_of(node) / 255.0
        emit = (1.0 if node & EMIT_BIT else 0.0) * u.emit_strength * rgb_of(node) / 255.0
        if bounce == 0:
            first = (n, t, node)
        if rand() < u.specularity:
            refl = norm(ray_d - 2.0 * (n @ ray_d) * n)
            sample = sample + emit * blend
            blend = blend * (2 * color * (refl @ n))
        else:
            if u.sun_strength > 0:
                rd = np.array([rand(), rand(), rand()])
                up = norm(np.cross(rd, sun_dir))
                right = norm(np.cross(sun_dir, up))
                dx, dy = 2 * rand() - 1, 2 * rand() - 1
                light = norm(sun_dir) + (dx * right + dy * up) * u.sun_size
                blocked, _, _, _ = next_cast(hit_pos + 1e-5 * n, norm(-light))
                ambient += 1
                if not blocked:
                    sample = sample + sun_color * color * blend * max(0.0, float(n @ norm(-light)))
            phi = 2 * np.pi * rand()
            rx = 2 * rand() - 1
            pr = np.sqrt(max(0.0, 1 - rx * rx))
            refl = np.array([rx, pr * np.cos(phi), pr * np.sin(phi)])
            fragile |= abs(float(n @ refl)) < 1e-5          # the flip below is decided by a sign that binary32 may see differently
            refl = refl - n * min(0.0, 2 * float(n @ refl))
            sample = sample + emit * blend
            blend = blend * (color * (n @ refl))
        ray_o, ray_d = hit_pos + 1e-5 * n, refl
    return sample / ambient, first, k, fragile


@pytest.mark.parametrize("name,bounces,spec,sun", [("castle", 4, 0.0, 4.0), ("room", 3, 0.3, 4.0), ("menger", 5, 0.0, 0.0), ("monu10", 8, 0.15, 4.0)])
def test_oracle_shading_against_an_independent_restatement(O, scenes, noise, name, bounces, spec, sun):
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    w, h = 96, 64
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.specularity, u.sun_strength = spec, sun
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    checked, hits = 0, 0
    for frame in (1, 700):          # 700 % 512 = 188: the frame index wraps
        u.frame_number = frame
        color, nd, alb, rays = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
        rng = np.random.default_rng(frame)
        total_casts = 0
        for _ in range(250):
            x, y = int(rng.integers(w)), int(rng.integers(h))
            log = O.path_log(octree, noise, u, bounces, x, y)
            got, first, used, fragile = shade_pixel(noise, u, x, y, bounces, log)
            assert used == len(log)                                        # the same number of rays, in the same order
            total_casts += used
            if fragile:
                continue
            assert np.allclose(color[y, x, :3], got, rtol=2e-4, atol=2e-5), (name, frame, x, y, color[y, x], got)
            assert color[y, x, 3] == 1.0
            assert np.array_equal(nd[y, x, :3], np.asarray(first[0], np.float32)) and nd[y, x, 3] == np.float32(first[1])
            node = int(alb[y, x, 3:4].view(np.int32)[0])
            assert node == first[2]
            want_alb = np.ones(3) if node & EMIT_BIT else rgb_of(node) / 255.0   # voxels.comp:392 (miss: node 0xffffff -> white)
            assert np.allclose(alb[y, x, :3], want_alb, rtol=1e-6)
            checked += 1
            hits += first[1] >= 0
        assert total_casts > 250
    assert checked > 400 and hits > 100


# ---- denoise.comp and temporal.comp, again from the shaders' text, vectorised numpy binary64 --------------------------------
def pixel_dirs(cam16, w, h):
    o, r, u, f = cam16[0:3].astype(np.float64), cam16[4:7].astype(np.float64), cam16[8:11].astype(np.float64), cam16[12:15].astype(np.float64)
    x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d = x[..., None] * r - y[..., None] * u + f
    return o, d / np.sqrt((d * d).sum(-1, keepdims=True))


def denoise_independent(colors, nd, alb, cam16, radius, sigma_distance, sigma_range, albedo_factor):
    """denoise.comp:24-93 (pow(v, 2) read as v*v)."""
    h, w = colors.shape[:2]
    _, dirs = pixel_dirs(cam16, w, h)
    c, n, dep = colors[..., :3].astype(np.float64), nd[..., :3].astype(np.float64), nd[..., 3].astype(np.float64)
    mat = alb[..., 3].view(np.int32) >> 24
    bias = np.maximum(0.0, (n * -dirs).sum(-1))
    sd2, sr2 = 2 * sigma_distance ** 2, 2 * sigma_range ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        logd = np.log(np.abs(dep))
    norm_, total = np.zeros((h, w)), np.zeros((h, w, 3))
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            ys, xs = np.arange(h)[:, None] + dy, np.arange(w)[None, :] + dx
            ok = (ys >= 0) & (ys < h) & (xs >= 0) & (xs < w)
            yc, xc = np.clip(ys, 0, h - 1), np.clip(xs, 0, w - 1)
            wc, wn, wl, wm = c[yc, xc], n[yc, xc], logd[yc, xc], mat[yc, xc]
            with np.errstate(invalid="ignore", over="ignore"):
                fr = (((c - wc) ** 2).sum(-1) + 1e4 * ((n - wn) ** 2).sum(-1) + 1e4 * (bias * (logd - wl)) ** 2 + 1e4 * (mat != wm)) / sr2
                fac = np.where(ok, np.exp(-fr - (dx * dx + dy * dy) / sd2), 0.0)
            norm_ += fac
            total += wc * fac[..., None]
    with np.errstate(invalid="ignore", divide="ignore"):
        out = c if radius == 0 else total / norm_[..., None]
    a = alb[..., :3].astype(np.float64)
    return out * (1 - albedo_factor) + a * out * albedo_factor


def temporal_independent_static(sampled, nd, old_color, old_nd, cam16, tu):
    """temporal.comp:48-125 for a camera AT REST (old camera = camera): the reprojected texel is the pixel itself, so the sampler's
    filtering (the one thing the shader's text does not define) drops out."""
    h, w = sampled.shape[:2]
    o, dirs = pixel_dirs(cam16, w, h)
    depth = nd[..., 3].astype(np.float64)
    world = o + depth[..., None] * dirs
    old_pos = o + old_nd[..., 3].astype(np.float64)[..., None] * dirs
    to_cam = o - world
    with np.errstate(invalid="ignore", divide="ignore"):
        cam_dir = to_cam / np.sqrt((to_cam * to_cam).sum(-1, keepdims=True))
        bias = np.maximum(0.0, (cam_dir * nd[..., :3].astype(np.float64)).sum(-1))
    dist = np.sqrt(((old_pos - world) ** 2).sum(-1))
    accept = (depth >= 0) & (dist < bias * tu.blending_distance_cutoff * depth)
    blending = np.where(accept, old_color[..., 3].astype(np.float64), 1.0)
    old_rgb = np.where(accept[..., None], old_color[..., :3].astype(np.float64), 0.0)
    new = sampled[..., :3].astype(np.float64)
    blended = np.where((depth >= 0)[..., None], old_rgb * (1 - blending[..., None]) + new * blending[..., None], new)
    nxt = np.clip((1 - tu.sample_blending) * blending, 1 - tu.maximum_blending, 1)
    return blended, nxt, accept


@pytest.mark.parametrize("name,radius", [("castle", 2), ("room", 5), ("menger", 8)])
def test_oracle_post_stages_against_independent_restatements(O, scenes, noise, name, radius):
    pos, mrgb, size = scenes.load_scene(name)
    octree = O.create_octree(pos, mrgb)
    w, h, bounces = 80, 56, 3
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    cam16 = u.camera16()
    tu = O.Temporal.default()
    accum, old_nd = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    for frame in (1, 2, 3):
        u.frame_number = frame
        color, nd, alb, _ = O.trace(octree, noise, u, w, h, bounces, crop=(0, 0, w, h))
        new_accum = O.temporal(color, nd, accum, old_nd, cam16, cam16, tu, frame > 1)
        if frame > 1:
            blended, nxt, accept = temporal_independent_static(color, nd, accum, old_nd, cam16, tu)
            assert accept[nd[..., 3] >= 0].mean() > 0.9                                    # a static camera re-finds its surfaces
            assert np.allclose(new_accum[..., :3], blended, rtol=1e-5, atol=1e-6)
            assert np.allclose(new_accum[..., 3], nxt, rtol=1e-6)
            assert np.isclose(new_accum[..., 3][accept].max(), 0.5 ** frame)                # the blending factor halves per frame (floor 0.02)
        else:
            assert np.array_equal(new_accum[..., :3], color[..., :3]) and (new_accum[..., 3] == 0.5).all()
        accum, old_nd = new_accum, nd
    du = O.Denoise.default()
    du.radius = radius
    got = O.denoise(accum, nd, alb, cam16, du)
    want = denoise_independent(accum, nd, alb, cam16, radius, du.sigma_distance, du.sigma_range, du.albedo_factor)
    ok = np.isfinite(want).all(-1)
    assert ok.mean() > 0.99 and np.array_equal(np.isfinite(got[..., :3]).all(-1), ok)
    assert np.allclose(got[..., :3][ok], want[ok], rtol=2e-4, atol=2e-6)
    assert (got[..., 3] == 1).all()
