"""CPU: the product's host-side scene preparation (libvxrt.so, no GPU needed) against the oracle and the
fixtures: .vox decoding + adapter (src/vox.rs, src/context.rs:913-933), octree builder
(src/context.rs:710-834), camera basis (src/camera.rs), noise table, procedural Menger, error codes."""
import os
import struct

import numpy as np
import pytest

from conftest import REFERENCE, needs_reference
from test_oracle_scene import KAT, chunk, make_vox, matl


@pytest.mark.parametrize("name", sorted(KAT))
def test_octree_builder_equals_oracle(O, H, scenes, name):
    pos, mrgb, _ = scenes.load_scene(name)
    words, depth = H.build_octree(pos, mrgb)
    assert depth == KAT[name]["depth"]
    assert np.array_equal(words, O.create_octree(pos, mrgb))       # word for word, header included


def test_octree_builder_random_negative_coordinates(O, H):
    rng = np.random.default_rng(5)
    for n, lo, hi in ((1, 0, 1), (50, -8, 8), (2000, -100, 60), (500, -512, 511), (10, 0, 30000)):
        pos = rng.integers(lo, hi, (n, 3)).astype(np.int16)
        mrgb = rng.integers(0, 256, (n, 4)).astype(np.uint8)
        words, depth = H.build_octree(pos, mrgb)
        assert depth == O.voxel_depth(pos) and np.array_equal(words, O.create_octree(pos, mrgb))
    words, depth = H.build_octree(np.zeros((0, 3), np.int16), np.zeros((0, 4), np.uint8))
    assert depth == 0 and len(words) == 13


def test_vox_decoder_equals_oracle_on_synthetic_files(O, H):
    files = [make_vox(), make_vox(voxels=((1, 2, 3, 5), (7, 8, 9, 6)), matls=(5, 6), emit=(6,)),
             make_vox(voxels=((0, 0, 0, 1), (0, 0, 1, 2), (0, 0, 2, 255)), rgba=False, matls=(1, 2, 255)),
             make_vox(pack=2, extra=chunk(b"nTRN", b"\x01\x02\x03") + chunk(b"LAYR", b"", b"zz")),
             make_vox(voxels=())]
    for data in files:
        p1, m1, s1 = H.vox_to_voxels(data)
        p2, m2, s2 = O.voxels_from_vox(data)
        assert s1 == s2 and np.array_equal(p1, p2) and np.array_equal(m1, m2)


def test_vox_decoder_error_codes(H):
    good = make_vox()

    def code(data):
        with pytest.raises(H.VxrtError) as e:
            H.vox_to_voxels(data)
        return e.value.status

    hdr = b"VOX " + struct.pack("<i", 150)
    empty_model = chunk(b"SIZE", struct.pack("<III", 1, 1, 1)) + chunk(b"XYZI", struct.pack("<I", 0))
    assert code(b"VOXX" + good[4:]) == H.E_VOX_MAGIC
    assert code(make_vox(version=200)) == H.E_VOX_VERSION
    assert code(hdr + chunk(b"PACK")) == H.E_VOX_NOMAIN
    assert code(good[:30]) == H.E_VOX_EOF and code(good[:-3]) == H.E_VOX_EOF
    assert code(hdr + chunk(b"MAIN", b"", chunk(b"XYZI", struct.pack("<I", 0)))) == H.E_VOX_CHUNK
    assert code(hdr + chunk(b"MAIN", b"", empty_model + matl(1, _type="_glass"))) == H.E_VOX_MATERIAL
    assert code(hdr + chunk(b"MAIN", b"", empty_model + matl(1, _flux="abc"))) == H.E_VOX_MATERIAL
    assert code(make_vox(matls=())) == H.E_VOX_NOMATL
    assert b"MATL" in H.lib().vxrt_last_error() or b"material" in H.lib().vxrt_last_error().lower()
    # a chunk whose two u32 sizes overflow when added is rejected, not wrapped
    evil = hdr + b"MAIN" + struct.pack("<II", 0xfffffff0, 0x20) + b"\0" * 64
    assert code(evil) == H.E_VOX_EOF


@needs_reference
def test_vox_decoder_equals_oracle_on_every_reference_scene(O, H):
    for f in sorted(os.listdir(os.path.join(REFERENCE, "vox"))):
        data = open(os.path.join(REFERENCE, "vox", f), "rb").read()
        p1, m1, s1 = H.vox_to_voxels(data)
        p2, m2, s2 = O.voxels_from_vox(data)
        assert s1 == s2 and np.array_equal(p1, p2) and np.array_equal(m1, m2), f


def test_camera_basis_equals_oracle(O, H):
    from gpu_voxel_raytracer_amd import Camera
    rng = np.random.default_rng(11)
    for _ in range(200):
        pos = rng.normal(size=3).astype(np.float32) * 30
        d = rng.normal(size=3).astype(np.float32)
        fov = float(np.float32(rng.uniform(0.3, 2.5)))
        w, h = int(rng.integers(16, 4000)), int(rng.integers(16, 2200))
        r, u, f = Camera(pos, d, fov).axis_scaled(w, h)
        assert np.array_equal(np.concatenate([r, u, f]), O.camera_axis_scaled(pos, d, fov, w, h))


def test_noise_table_spec(O, H):
    a = H.noise_table(0x5EED0001, 1 << 16)
    assert np.array_equal(a, O.noise_table(0x5EED0001, 1 << 16))
    # the documented generator (include/vxrt.h), third implementation in numpy
    i = np.arange(1 << 16, dtype=np.uint32)
    z = i * np.uint32(0x9E3779B9) + np.uint32(0x5EED0001)
    z ^= z >> np.uint32(16); z *= np.uint32(0x85EBCA6B); z ^= z >> np.uint32(13); z *= np.uint32(0xC2B2AE35); z ^= z >> np.uint32(16)
    assert np.array_equal(a, (z >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))
    full = H.noise_table()
    assert full.size == 512 * 128 * 128 and full.min() >= 0 and full.max() < 1 and abs(full.mean() - 0.5) < 1e-3
    assert not np.array_equal(H.noise_table(1, 64), H.noise_table(2, 64))


def test_menger_generator(H, scenes):
    pos, mrgb, _ = scenes.load_scene("menger")
    gp, gm = H.menger_voxels(4, mrgb[0])
    assert len(gp) == 20 ** 4 == len(pos)
    assert set(map(tuple, gp.tolist())) == set(map(tuple, pos.tolist()))   # same voxel set as vox/menger.vox
    assert (gm == mrgb[0]).all()
    for level, n in ((0, 1), (1, 20), (2, 400), (3, 8000)):
        assert len(H.menger_voxels(level)[0]) == n


def test_defaults_match_reference_values(H):
    u, t, d = H.Uniforms.default(), H.TemporalUniforms.default(), H.DenoiseUniforms.default()
    f = np.float32
    assert (u.emit_strength, u.sun_strength, f(u.sun_size), f(u.sun_yaw), u.sun_pitch) == (4.0, 4.0, f(0.05), f(1.32), 1.0)
    assert list(u.sun_color)[:3] == [1, 1, 1] and [f(v) for v in list(u.sky_color)[:3]] == [f(0.45), f(0.6), f(0.65)]
    assert u.specularity == 0 and u.frame_number == 0
    assert (f(t.sample_blending), f(t.maximum_blending), f(t.blending_distance_cutoff)) == (f(0.5), f(0.98), f(1e-2))
    assert (d.radius, d.sigma_distance, d.sigma_range, d.albedo_factor) == (0, 2.0, 1.5, 1.0)
    import ctypes
    assert ctypes.sizeof(H.Uniforms) == 148 and H.Uniforms.specularity.offset == 144 and H.Uniforms.sun_color.offset == 112


def test_product_fails_loudly_without_gpu(H):
    """No CPU fallback: without a HIP device the context cannot be created."""
    import ctypes
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is present")
    from gpu_voxel_raytracer_amd import Context, VxrtError
    with pytest.raises(VxrtError) as e:
        Context(64, 64)
    assert e.value.status == H.E_DEVICE


def test_bench_schedule_choice():
    """bench.py's frames per launch / launches in flight for the driver's N and K (pure host logic)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.pick_schedule(1, 1000) == (2, 16) and bench.pick_schedule(2, 1000) == (3, 16)
    assert bench.pick_schedule(4, 1000) == (3, 16) and bench.pick_schedule(8, 1000) == (3, 16)   # never more than 3 trace streams
    assert bench.pick_schedule(1, 20) == (3, 8)                     # the driver's --steps 20: launches of 8, 8 and 4 frames (frame lanes)
    assert bench.pick_schedule(2, 20) == (3, 8) and bench.pick_schedule(4, 20) == (2, 16) and bench.pick_schedule(8, 20) == (1, 20)
    assert bench.pick_schedule(8, 50) == (2, 25) and bench.pick_schedule(8, 95) == (3, 32)
    for world in (1, 2, 4, 8):
        for steps in (1, 2, 5, 20, 63, 64, 100):
            inflight, batch = bench.pick_schedule(world, steps)
            assert 1 <= batch <= min(32, steps) and 1 <= inflight <= 3
    assert bench.block_count(0.0025, 0) == 400 and bench.block_count(0.12, 0) == 50 and bench.block_count(5.0, 0) == 50
    assert bench.block_count(0.12, 7) == 7
    assert bench.pick_schedule(8, 1000, inflight=2, batch=4) == (2, 4)   # explicit values are kept
    # a short block from 8 ranks on: 4-row bands and the all-in-one kernel (round 5); everything else the default deal
    assert bench.pick_deal(8, 20) == (4, 1) and bench.pick_deal(4, 20) == (8, 0) and bench.pick_deal(1, 20) == (8, 0)
    assert bench.pick_deal(8, 1000) == (4, 0) and bench.pick_deal(8, 20, inflight=2, batch=10) == (4, 0) and bench.pick_deal(4, 1000) == (8, 0)
    assert bench.algorithmic_bytes(1920 * 1080, 4, 359016 + 640000) == 48 * 1920 * 1080 + 999016 + 32 * 65536


def decode_wide(wide, leaves, depth):
    """Every (x, y, z, leaf word) a walk over the wide records (csrc/kernels.h: WideRec) can reach — written from the format's
    description, independently of the kernels: levels paired from the bottom, byte s of the 64-bit mask = occupancy of sub s,
    grandchild (s, o) = base + popcount(mask below bit 8s+o); with an odd number of node levels the root is sub 0 of a virtual top."""
    L = depth + 1
    out = []

    def rank(mask, pos):
        return bin(mask & ((1 << pos) - 1)).count("1")

    def visit_sub(rec, s, cx, cy, cz, extent, level):
        # node at `level` (a sub of rec) centred (cx, cy, cz) with half-size `extent` (src/context.rs:749-753 geometry)
        mask = int(rec[0]) | int(rec[1]) << 32
        byte = (mask >> (8 * s)) & 0xff
        for o in range(8):
            if not byte >> o & 1:
                continue
            dx, dy, dz = o >> 2 & 1, o >> 1 & 1, o & 1
            idx = int(rec[2]) + rank(mask, 8 * s + o)
            if level == L - 1:      # leaf parent: unit voxels at c - 1 + d
                out.append((cx - 1 + dx, cy - 1 + dy, cz - 1 + dz, int(leaves[idx])))
            else:
                h = extent // 2
                visit_top(wide[idx], cx - h + dx * extent, cy - h + dy * extent, cz - h + dz * extent, h, level + 1)

    def visit_top(rec, cx, cy, cz, extent, level):
        mask = int(rec[0]) | int(rec[1]) << 32
        for s in range(8):
            occupied = (mask >> (8 * s)) & 0xff != 0
            assert occupied == bool(int(rec[3]) >> s & 1)          # .top mirrors "byte s != 0"
            if occupied:
                h = extent // 2
                visit_sub(rec, s, cx - h + (s >> 2 & 1) * extent, cy - h + (s >> 1 & 1) * extent, cz - h + (s & 1) * extent, h, level + 1)

    if L % 2:
        visit_sub(wide[0], 0, 0, 0, 0, 1 << depth, 0)
    else:
        visit_top(wide[0], 0, 0, 0, 1 << depth, 0)
    return sorted(out)


@pytest.mark.parametrize("case", ["castle", "8x8x8", "single", "empty", "negative", "depth4", "depth10", "depth15"])
def test_wide_records_hold_the_voxel_set(H, scenes, case):
    """The two-levels-per-record scene format (16-byte WideRec) decodes to exactly the voxel list it was built from — for even and odd
    numbers of tree levels (odd: the root hangs under a virtual top), an empty scene, one voxel, all eight root octants."""
    rng = np.random.default_rng(9)
    if case in ("castle", "8x8x8"):
        pos, mrgb, _ = scenes.load_scene(case)
    elif case == "single":
        pos, mrgb = np.array([[0, 0, 0]], np.int16), np.array([[0x40, 1, 2, 3]], np.uint8)
    elif case == "empty":
        pos, mrgb = np.zeros((0, 3), np.int16), np.zeros((0, 4), np.uint8)
    elif case == "negative":
        pos, mrgb = rng.integers(-40, 40, (3000, 3)).astype(np.int16), rng.integers(0, 256, (3000, 4)).astype(np.uint8)
    else:
        hi = {"depth4": 15, "depth10": 1000, "depth15": 32767}[case]
        pos = np.concatenate([rng.integers(0, hi + 1, (500, 3)), [[hi, 0, 3]]]).astype(np.int16)
        mrgb = rng.integers(0, 256, (len(pos), 4)).astype(np.uint8)
    svo, wide, leaves, depth = H.build_records(pos, mrgb)
    if case.startswith("depth"):
        assert depth == int(case[5:])
    # the reference's overwrite rule: the last voxel at a position wins (src/context.rs:732-735)
    want = {}
    for p, m in zip(pos.tolist(), mrgb.tolist()):
        want[tuple(p)] = -2 ** 31 | (m[0] & 0x7f) << 24 | m[1] << 16 | m[2] << 8 | m[3]
    got = decode_wide(wide, leaves, depth)
    assert [(x, y, z) for x, y, z, _ in got] == sorted(want)
    assert all(want[(x, y, z)] == w for x, y, z, w in got)
    assert len(wide) <= len(svo) and (len(svo) < 16 or 16 * len(wide) < 8 * len(svo))     # fewer bytes than the 8-byte records


def read_exr_uncompressed(path):
    """An independent reader of the subset of OpenEXR that frame_loop.save_exr writes (version 2, scan lines, NO_COMPRESSION, FLOAT
    channels): walks the attribute list, the offset table and the scan-line blocks -> ({channel: float32 [h, w]}, attributes)."""
    import struct
    b = open(path, "rb").read()
    assert struct.unpack_from("<i", b, 0)[0] == 20000630 and b[:4] == bytes([0x76, 0x2f, 0x31, 0x01])
    version = struct.unpack_from("<i", b, 4)[0]
    assert version & 0xff == 2 and version >> 8 == 0                  # no tiles, no long names, no deep data, single part
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); kind = b[pos:e].decode(); pos = e + 1
        size = struct.unpack_from("<i", b, pos)[0]; pos += 4
        attrs[name] = (kind, b[pos:pos + size]); pos += size
    pos += 1
    for required in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"):
        assert required in attrs, required
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, p, ch = [], 0, attrs["channels"][1]
    while ch[p] != 0:
        e = ch.index(b"\0", p); name = ch[p:e].decode(); p = e + 1
        ptype, plinear, xs, ys = struct.unpack_from("<iB3xii", ch, p); p += 16
        assert ptype == 2 and xs == 1 and ys == 1
        chans.append(name)
    assert chans == sorted(chans)                                      # the file format wants them in alphabetical order
    offsets = np.frombuffer(b, "<u8", h, pos)
    out = {c: np.zeros((h, w), np.float32) for c in chans}
    for row in range(h):
        o = int(offsets[row])
        y, size = struct.unpack_from("<ii", b, o)
        assert y == y0 + row and size == 4 * w * len(chans)
        data = np.frombuffer(b, "<f4", w * len(chans), o + 8).reshape(len(chans), w)
        for i, c in enumerate(chans):
            out[c][row] = data[i]
    assert int(offsets[-1]) + 8 + 4 * w * len(chans) == len(b)        # nothing after the last block
    return out, attrs


def test_exr_writer_round_trip(tmp_path):
    """frame_loop.save_exr (PNG / EXR dumps): the file parses as OpenEXR 2 scan lines and gives every float back bit for
    bit, NaN, infinities and denormals included."""
    from gpu_voxel_raytracer_amd import frame_loop
    rng = np.random.default_rng(11)
    img = rng.standard_normal((37, 53, 4)).astype(np.float32)
    img[0, 0] = (np.nan, np.inf, -np.inf, 1e-42)
    img[5, 7] = (0.0, -0.0, 3.4e38, 1.0)
    path = str(tmp_path / "f.exr")
    frame_loop.save_exr(img, path)
    planes, attrs = read_exr_uncompressed(path)
    assert sorted(planes) == ["A", "B", "G", "R"] and attrs["channels"][0] == "chlist" and attrs["dataWindow"][0] == "box2i"
    for c, k in (("R", 0), ("G", 1), ("B", 2), ("A", 3)):
        assert np.array_equal(planes[c].view(np.uint32), img[..., k].view(np.uint32)), c
    assert os.path.getsize(path) == 8 + sum(len(n) + 1 + len(k) + 1 + 4 + len(v) for n, (k, v) in attrs.items()) + 1 + 37 * 8 + 37 * (8 + 53 * 16)
