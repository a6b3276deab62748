"""SURVEY.md §8f rows n2 / n3 on the CPU: the reference's blue-noise archive format, whole MagicaVoxel scenes (scene graph,
lenient materials) and the start-up scene.  The reference holds no fixtures for any of these (its blue-noise zip is missing
and its vox files have one model each), so the checks are: the oracle's restatement, an independent placement written
here with float rotation matrices, python's zipfile, and the reference's single-model files."""
import os
import struct
import zipfile

import numpy as np
import pytest
from conftest import REFERENCE, needs_reference
from test_oracle_scene import chunk, make_vox, matl, vstr


# ---- start-up scene (Context::create_voxels, src/context.rs:838-910) ---------------------------------------------
def test_default_scene_equals_oracle_and_has_the_reference_shape(O, H):
    pos, mrgb = H.default_scene_voxels(7)
    opos, omrgb = O.default_scene(7)
    assert (pos == opos).all() and (mrgb == omrgb).all()
    assert (pos.min(0) == [-256, -256, -256]).all() and (pos.max(0) == [256, 0, 256]).all()
    # the strip of light closes the list (src/context.rs:907-910)
    assert (pos[-513:, 0] == np.arange(-256, 257)).all() and (pos[-513:, 1] == -10).all() and (pos[-513:, 2] == 0).all()
    assert (mrgb[-513:] == [0x40, 255, 255, 255]).all()
    body = mrgb[:-513]
    assert body[:, 1:].min() >= 50 and 0.005 < (body[:, 0] == 0x40).mean() < 0.02 and set(np.unique(body[:, 0])) <= {0, 0x40}
    # the bowl: centre column reaches -256, outside the radius the ground is flat at 0
    centre = pos[(pos[:, 0] == 0) & (pos[:, 2] == 0)][:, 1]
    assert centre.min() == -256
    corner = pos[(pos[:, 0] == 250) & (pos[:, 2] == 250)][:, 1]
    assert (corner == 0).all() and len(corner) == 1
    # a different seed changes colours only
    pos2, mrgb2 = H.default_scene_voxels(8)
    assert (pos2 == pos).all() and not (mrgb2 == mrgb).all()
    # depth rule: |min| = 256 -> 8, max + 1 = 257 -> 9 (src/context.rs:813-834)
    _, depth = H.build_octree(pos[:2000], mrgb[:2000])
    assert depth == 9


# ---- whole .vox scenes -------------------------------------------------------------------------------------------
def vdict(**kv):
    body = struct.pack("<I", len(kv))
    for k, v in kv.items():
        body += vstr(k.encode()) + vstr(str(v).encode())
    return body


def ntrn(nid, child, t=None, r=None):
    frame = {}
    if t is not None:
        frame["_t"] = "%d %d %d" % tuple(t)
    if r is not None:
        frame["_r"] = r
    return chunk(b"nTRN", struct.pack("<i", nid) + vdict() + struct.pack("<iiiI", child, -1, 0, 1) + vdict(**frame))


def ngrp(nid, children):
    return chunk(b"nGRP", struct.pack("<i", nid) + vdict() + struct.pack("<I", len(children)) + b"".join(struct.pack("<i", c) for c in children))


def nshp(nid, models):
    return chunk(b"nSHP", struct.pack("<i", nid) + vdict() + struct.pack("<I", len(models)) + b"".join(struct.pack("<i", m) + vdict() for m in models))


def rot_matrix(byte):
    c0, c1 = byte & 3, (byte >> 2) & 3
    c2 = 3 - c0 - c1
    m = np.zeros((3, 3))
    m[0, c0] = -1 if byte & 16 else 1
    m[1, c1] = -1 if byte & 32 else 1
    m[2, c2] = -1 if byte & 64 else 1
    return m


def scene_file(models, graph, matls=(1, 2, 3), emit=(2,), types=None):
    body = b""
    for size, vox in models:
        body += chunk(b"SIZE", struct.pack("<III", *size)) + chunk(b"XYZI", struct.pack("<I", len(vox)) + b"".join(bytes(v) for v in vox))
    body += graph
    for m in matls:
        body += matl(m, _type=(types or {}).get(m, "_emit" if m in emit else "_diffuse"))
    return b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", body)


def test_scene_graph_placement_equals_independent_float_placement(H):
    rng = np.random.default_rng(5)
    models = []
    for size in ((3, 4, 5), (6, 2, 3), (1, 1, 7)):
        cells = {(int(rng.integers(size[0])), int(rng.integers(size[1])), int(rng.integers(size[2]))) for _ in range(20)}
        models.append((size, [(x, y, z, 1 + (x + y + z) % 3) for x, y, z in sorted(cells)]))
    # root -> group -> three transforms (one nested under a rotated group) -> shapes; model 1 is instanced twice
    r_a, r_b, r_c = 4 | (1 << 4), 9 | (1 << 5), 2 | (1 << 2) | (1 << 6)   # valid signed permutations
    for r in (r_a, r_b, r_c):
        assert abs(round(np.linalg.det(rot_matrix(r)))) == 1
    graph = (ntrn(0, 1) + ngrp(1, [2, 4, 6]) + ntrn(2, 3, t=(10, -3, 4), r=r_a) + nshp(3, [0])
             + ntrn(4, 5, t=(-7, 20, 1)) + nshp(5, [1])
             + ntrn(6, 7, t=(2, 2, 30), r=r_b) + ngrp(7, [8, 10]) + ntrn(8, 9, t=(5, 0, -6), r=r_c) + nshp(9, [1])
             + ntrn(10, 11, t=(0, 9, 0)) + nshp(11, [2]))
    data = scene_file(models, graph)
    pos, mrgb, (lo, hi) = H.vox_scene_to_voxels(data, H.VOX_ALL_MODELS)

    def place(model, R, t):
        size, vox = model
        pivot = np.array([s // 2 for s in size], float)
        out = []
        for x, y, z, c in vox:
            w = np.floor(R @ (np.array([x, y, z], float) + 0.5 - pivot) + t).astype(int)
            out.append((w[0], w[2], w[1], c))   # renderer axes (x, z, y)
        return out
    Ra, Rb, Rc = rot_matrix(r_a), rot_matrix(r_b), rot_matrix(r_c)
    want = (place(models[0], Ra, np.array([10, -3, 4.0])) + place(models[1], np.eye(3), np.array([-7, 20, 1.0]))
            + place(models[1], Rb @ Rc, Rb @ np.array([5, 0, -6.0]) + np.array([2, 2, 30.0]))
            + place(models[2], Rb, Rb @ np.array([0, 9, 0.0]) + np.array([2, 2, 30.0])))
    assert len(pos) == len(want)
    assert [tuple(int(v) for v in p) for p in pos] == [w[:3] for w in want]
    # palette of make_vox-less file: default palette; colour 2 is emissive
    assert [(int(m[0]) == 0x40) for m in mrgb] == [w[3] == 2 for w in want]
    assert lo == tuple(int(v) for v in pos.min(0)) and hi == tuple(int(v) for v in pos.max(0))
    # rebase: same cloud, minimum corner at the origin
    rpos, _, (rlo, rhi) = H.vox_scene_to_voxels(data, H.VOX_ALL_MODELS | H.VOX_REBASE)
    assert (rpos == pos - pos.min(0)).all() and rlo == (0, 0, 0) and rhi == tuple(int(v) for v in (pos.max(0) - pos.min(0)))
    # without the flag only models[0] is taken, at its raw coordinates (src/context.rs:916-927)
    p0, _, _ = H.vox_scene_to_voxels(data, H.VOX_LENIENT_MATERIALS)
    assert [tuple(int(v) for v in p) for p in p0] == [(x, z, y) for x, y, z, _ in models[0][1]]


def test_files_without_a_scene_graph_place_every_model_raw(H):
    a = ((2, 2, 2), [(0, 0, 0, 1), (1, 1, 1, 1)])
    b = ((4, 4, 4), [(3, 0, 2, 3)])
    pos, _, _ = H.vox_scene_to_voxels(scene_file([a, b], b""), H.VOX_ALL_MODELS)
    assert [tuple(int(v) for v in p) for p in pos] == [(0, 0, 0), (1, 1, 1), (3, 2, 0)]


def test_lenient_materials_and_errors(H):
    model = ((2, 2, 2), [(0, 0, 0, 1), (1, 0, 0, 2), (0, 1, 0, 9)])
    glass = scene_file([model], b"", matls=(1, 2), emit=(2,), types={1: "_glass"})
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(glass, H.VOX_ALL_MODELS)
    assert e.value.status == H.E_VOX_MATERIAL               # the reference's rule, src/vox.rs:82-89
    ok_mats = scene_file([model], b"", matls=(1, 2), emit=(2,))
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(ok_mats, H.VOX_ALL_MODELS)
    assert e.value.status == H.E_VOX_NOMATL                 # colour 9 has no MATL (src/context.rs:919)
    pos, mrgb, _ = H.vox_scene_to_voxels(glass, H.VOX_ALL_MODELS | H.VOX_LENIENT_MATERIALS)
    assert len(pos) == 3 and [int(m) for m in mrgb[:, 0]] == [0, 0x40, 0]
    # graph errors: missing node, bad rotation, cycle, coordinates beyond i16
    for graph in (ntrn(0, 5), ntrn(0, 1, r=0) + nshp(1, [0]), ntrn(0, 1) + ngrp(1, [0]), ntrn(0, 1, t=(40000, 0, 0)) + nshp(1, [0]),
                  ntrn(0, 1) + nshp(1, [3])):
        with pytest.raises(H.VxrtError) as e:
            H.vox_scene_to_voxels(scene_file([model], graph, matls=(1, 2, 9)), H.VOX_ALL_MODELS)
        assert e.value.status == H.E_SCENE
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(ok_mats[:60], H.VOX_ALL_MODELS)
    assert e.value.status == H.E_VOX_EOF
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(ok_mats, 64)
    assert e.value.status == H.E_INVALID


def test_flags_zero_is_the_reference_adapter(H):
    data = make_vox(voxels=((1, 2, 3, 5), (0, 0, 1, 5)), pack=2)
    a = H.vox_to_voxels(data)
    b = H.vox_scene_to_voxels(data, 0)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and b[2] == ((0, 0, 0), (3, 5, 4))


@needs_reference
def test_reference_files_through_their_scene_graph(H):
    """Every file in the reference's vox/ is one model under root -> group -> transform(_t, no rotation) -> shape (mostly
    _t = (0, 0, size_z / 2): centred in x / y, standing on z = 0); re-based it is the reference's own list."""
    import re
    for f in sorted(os.listdir(os.path.join(REFERENCE, "vox"))):
        data = open(os.path.join(REFERENCE, "vox", f), "rb").read()
        pos, mrgb, size = H.vox_to_voxels(data)
        gpos, gmrgb, (lo, hi) = H.vox_scene_to_voxels(data, H.VOX_ALL_MODELS)
        assert (gmrgb == mrgb).all(), f
        ts = re.findall(rb"_t....(-?\d+ -?\d+ -?\d+)", data, re.S)
        assert len(ts) == 1 and b"_r" not in data[data.find(b"nTRN"):data.find(b"nSHP")], f
        t = [int(v) for v in ts[0].split()]
        d = [t[a] - size[a] // 2 for a in range(3)]               # world = cell - pivot + t, file axes
        assert (gpos == pos + np.array([d[0], d[2], d[1]])).all(), f   # renderer axes (x, z_file, y_file)
        rpos, _, _ = H.vox_scene_to_voxels(data, H.VOX_ALL_MODELS | H.VOX_REBASE)
        assert (rpos == pos - pos.min(0)).all(), f


# ---- blue-noise archive (Context::load_blue_noise / parse_raw_f32img, src/context.rs:1042-1116) --------------------
def raw_image(px, w=None, h=None):
    h0, w0 = px.shape
    return struct.pack(">II", w0 if w is None else w, h0 if h is None else h) + px.astype(">f4").tobytes()


def test_noise_zip_round_trip_and_python_zipfile_interop(O, H, tmp_path):
    rng = np.random.default_rng(3)
    table = rng.random((5, 16, 16), dtype=np.float32)
    table[0, 0, 0] = np.float32(1e-30)
    path = str(tmp_path / "blue-noise-16.zip")
    H.save_blue_noise(path, table, size=16)
    size, px = H.load_blue_noise(path)
    assert size == 16 and (px.view(np.uint32) == table.reshape(-1).view(np.uint32)).all()
    with zipfile.ZipFile(path) as z:           # a standard reader accepts the archive and sees the reference's layout
        assert z.testzip() is None and len(z.namelist()) == 5
        for i, name in enumerate(z.namelist()):
            img = O.parse_raw_f32img(z.read(name))
            assert img.shape == (16, 16) and (img.view(np.uint32) == table[i].view(np.uint32)).all()
    # an archive made by a standard writer (deflated, with a directory entry, which is_file() skips)
    other = str(tmp_path / "made-by-zipfile.zip")
    with zipfile.ZipFile(other, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("noise/", b"")
        for i in range(3):
            z.writestr("noise/%d.bin" % i, raw_image(table[i]) + b"trailing bytes are not read")
    size, px = H.load_blue_noise(other)
    assert size == 16 and (px.view(np.uint32) == table[:3].reshape(-1).view(np.uint32)).all()


def test_noise_zip_errors(H, tmp_path):
    img = np.zeros((4, 4), np.float32)

    def code(entries, name="t.zip"):
        path = str(tmp_path / name)
        with zipfile.ZipFile(path, "w") as z:
            for n, b in entries:
                z.writestr(n, b)
        with pytest.raises(H.VxrtError) as e:
            H.load_blue_noise(path)
        return e.value.status, e.value.detail

    assert "did not contain any files" in code([])[1]
    assert "did not contain any images" in code([("d/", b"")])[1]
    assert "non-square" in code([("a", raw_image(np.zeros((2, 4), np.float32)))])[1]
    assert "same size" in code([("a", raw_image(img)), ("b", raw_image(np.zeros((2, 2), np.float32)))])[1]
    assert "failed to read image" in code([("a", raw_image(img)[:-1])])[1]
    assert code([("a", b"\0\0")])[0] == H.E_NOISE
    bad = tmp_path / "notzip.zip"
    bad.write_bytes(b"hello")
    with pytest.raises(H.VxrtError) as e:
        H.load_blue_noise(str(bad))
    assert e.value.status == H.E_NOISE
    with pytest.raises(H.VxrtError) as e:
        H.load_blue_noise(str(tmp_path / "missing.zip"))
    assert e.value.status == H.E_NOISE and "while opening" in e.value.detail


# ---- void-and-cluster spec (oracle side; the HIP kernel is checked against it in test_gpu_noise.py) -----------------
def radial_power(layer):
    n = layer.shape[0]
    f = np.abs(np.fft.fft2(layer - layer.mean())) ** 2
    fy, fx = np.meshgrid(np.fft.fftfreq(n), np.fft.fftfreq(n), indexing="ij")
    r = np.hypot(fx, fy)
    return f, r


def test_oracle_blue_noise_is_a_rank_permutation_with_a_blue_spectrum(O):
    for size, layer in ((32, 0), (64, 3)):
        a = O.blue_noise_layer(0x5EED0001, layer, size)
        ranks = np.sort((a.reshape(-1) * size * size - 0.5).astype(np.int64))
        assert (ranks == np.arange(size * size)).all()
        f, r = radial_power(a)
        assert f[(r > 0) & (r <= 0.1)].mean() < 0.01 * f[r > 0].mean()      # white noise: ~1
    assert not (O.blue_noise_layer(1, 0, 32) == O.blue_noise_layer(1, 1, 32)).all()
    assert not (O.blue_noise_layer(1, 0, 32) == O.blue_noise_layer(2, 0, 32)).all()


def test_oracle_blue_noise_known_answer(O):
    import hashlib
    a = O.blue_noise_layer(0x5EED0001, 0, 128)
    assert hashlib.sha256(a.tobytes()).hexdigest() == BLUE_NOISE_LAYER0_SHA256


BLUE_NOISE_LAYER0_SHA256 = "9a30192a9871f35e68cbcbbc57ad9a94a9369f07426a9fa652de7b9b8fb08c7e"


def test_untrusted_scene_limits(H):
    """Damaged or hostile scene graphs end in a status code, never in wrapped coordinates, undefined arithmetic or an
    allocation the file's size does not justify."""
    model = ((2, 2, 2), [(0, 0, 0, 1), (1, 1, 1, 1)])
    # a cloud wider than 16-bit coordinates can hold once re-based: two instances 40 000 apart (each fits on its own)
    wide = ntrn(0, 1) + ngrp(1, [2, 4]) + ntrn(2, 3, t=(-20000, 0, 0)) + nshp(3, [0]) + ntrn(4, 5, t=(20000, 0, 0)) + nshp(5, [0])
    pos, _, (lo, hi) = H.vox_scene_to_voxels(scene_file([model], wide), H.VOX_ALL_MODELS)
    assert hi[0] - lo[0] > 32767
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(scene_file([model], wide), H.VOX_ALL_MODELS | H.VOX_REBASE)
    assert e.value.status == H.E_SCENE
    # a translation no 16-bit scene can use (it would overflow the 64-bit composition a few levels down)
    for t in ((2 ** 40, 0, 0), (0, -(2 ** 62), 0)):
        with pytest.raises(H.VxrtError) as e:
            H.vox_scene_to_voxels(scene_file([model], ntrn(0, 1, t=t) + nshp(1, [0])), H.VOX_ALL_MODELS)
        assert e.value.status == H.E_SCENE
    # one 64^3-cell model instanced 1 024 times = 2^28 voxels: refused before the vector grows that far
    cells = [(x, y, z, 1) for x in range(64) for y in range(64) for z in range(64)]
    big = ((64, 64, 64), cells)
    graph = ntrn(0, 1) + ngrp(1, [2]) + ntrn(2, 3) + nshp(3, [0] * 1024)
    with pytest.raises(H.VxrtError) as e:
        H.vox_scene_to_voxels(scene_file([big], graph), H.VOX_ALL_MODELS)
    assert e.value.status == H.E_SCENE


@pytest.mark.parametrize("text,ok", [("1.5", True), ("-0.25e-3", True), ("+7", True), (".5", True), ("5.", True), ("1E10", True), ("inf", True),
                                     ("-Infinity", True), ("NaN", True), ("", False), (" 1.0", False), ("1.0 ", False), ("0x1p3", False),
                                     ("nan(0x1)", False), (".", False), ("1e", False), ("1e+", False), ("--1", False), ("1.5f", False), ("e5", False)])
def test_flux_follows_rusts_f32_grammar(O, H, text, ok):
    """MATL `_flux` is validated with value.parse::<f32>() in the reference (src/vox.rs:93-96): Rust's grammar, not C's strtof
    (which also takes blanks, hex floats and nan(...)).  Library, whole-scene decoder and oracle agree on every case."""
    body = struct.pack("<II", 5, 2) + vstr(b"_type") + vstr(b"_diffuse") + vstr(b"_flux") + vstr(text.encode())
    data = make_vox(matls=(), extra=chunk(b"MATL", body))
    for decode in (H.vox_to_voxels, lambda d: H.vox_scene_to_voxels(d, H.VOX_ALL_MODELS)):
        if ok:
            assert len(decode(data)[0]) == 1
        else:
            with pytest.raises(H.VxrtError) as e:
                decode(data)
            assert e.value.status == H.E_VOX_MATERIAL
    if ok:
        assert len(O.voxels_from_vox(data)[0]) == 1
    else:
        with pytest.raises(O.OracleError) as e:
            O.voxels_from_vox(data)
        assert e.value.code == -6
