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
