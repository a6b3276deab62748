"""CPU: the oracle's scene preparation (oracle/ovox.cpp) against the known answers the repo holds.

The reference ships NO tests or golden vectors: these pins are the node-count table of
SURVEY.md Appendix C (tests/golden/octree_kat.json), the reference's own scene files when mounted, and
hand-made .vox byte strings for the parser's error paths (src/vox.rs)."""
import hashlib
import json
import os
import re
import struct

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, needs_reference

KAT = json.load(open(os.path.join(GOLDEN, "octree_kat.json")))


def levels(octree):
    arr = octree[5:].reshape(-1, 8)
    out, cur = [], np.array([0])
    while len(cur):
        out.append(int(len(cur)))
        v = arr[cur].ravel()
        cur = v[v > 0]
    return out


@pytest.mark.parametrize("name", sorted(KAT))
def test_octree_known_answers(O, scenes, name):
    pos, mrgb, size = scenes.load_scene(name)
    k = KAT[name]
    assert list(size) == k["size"] and len(pos) == k["voxels"]
    assert O.voxel_depth(pos) == k["depth"]
    octree = O.create_octree(pos, mrgb)
    hdr = octree[:5].view(np.float32)
    assert hdr.tolist() == [0.0, 0.0, 0.0, float(1 << k["depth"]), 1.0]          # src/context.rs:782-791
    assert levels(octree) == k["nodes_per_level"] and (len(octree) - 5) // 8 == k["nodes"]
    assert octree.nbytes == k["bytes"] and int((mrgb[:, 0] == 0x40).sum()) == k["emissive"]
    assert hashlib.sha256(octree.tobytes()).hexdigest() == k["sha256"]


# SURVEY.md Appendix C, computed there with an unrelated throw-away parser
APPENDIX_C = {
    "3x3x3": (420891, 7, [1, 1, 4, 32, 174, 1008, 7054, 54824], 63098, 2019156, 199574),
    "menger": (160000, 7, [1, 1, 8, 27, 200, 1184, 6480, 36976], 44877, 1436084, 0),
    "monu10": (150764, 7, [1, 1, 7, 22, 106, 514, 3131, 20341], 24123, 771956, 0),
    "castle": (2628, 5, [1, 1, 8, 20, 119, 595], 744, 23828, 0),
    "8x8x8": (80, 3, [1, 1, 8, 32], 42, 1364, 0),
}


@pytest.mark.parametrize("name", sorted(APPENDIX_C))
def test_kat_matches_survey_appendix_c(name):
    v, d, lv, n, b, e = APPENDIX_C[name]
    k = KAT[name]
    assert (k["voxels"], k["depth"], k["nodes_per_level"], k["nodes"], k["bytes"], k["emissive"]) == (v, d, lv, n, b, e)


def test_octree_leaf_words_and_overwrite(O):
    # leaf word = 1<<31 | (m & 0x7f)<<24 | r<<16 | g<<8 | b ; a later voxel at the same cell wins (src/context.rs:732-735)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0], [-1, -1, -1]], np.int16)
    mrgb = np.array([[0, 1, 2, 3], [0x40, 255, 254, 253], [0xff, 9, 8, 7], [0, 10, 20, 30]], np.uint8)
    octree = O.create_octree(pos, mrgb)
    assert O.voxel_depth(pos) == 1                       # max=1 -> npot(2)=2 -> 1 ; min=-1 -> npot(1)=1 -> 0
    nodes = octree[5:].reshape(-1, 8)
    leaves = sorted(int(v) & 0xffffffff for v in nodes.ravel() if v < 0)
    assert leaves == sorted([0x80000000 | 0x7f << 24 | 9 << 16 | 8 << 8 | 7, 0x80000000 | 0x40 << 24 | 255 << 16 | 254 << 8 | 253,
                             0x80000000 | 10 << 16 | 20 << 8 | 30])
    # root slot = 4*(x>=0) + 2*(y>=0) + (z>=0): three positive voxels share slot 7, the negative one is in slot 0
    assert nodes[0][7] > 0 and nodes[0][0] > 0 and (nodes[0][1:7] == 0).all()


def test_voxel_depth_rule(O):
    # depth = max(tz(npot(|min|)), tz(npot(|max|+1)))  (src/context.rs:813-834)
    cases = {(0, 0): 0, (0, 1): 1, (0, 3): 2, (0, 4): 3, (-1, 0): 0, (-2, 0): 1, (-3, 0): 2, (-4, 3): 2, (-5, 3): 3, (0, 127): 7, (0, 128): 8}
    for (lo, hi), want in cases.items():
        pos = np.array([[lo, hi, 0]], np.int16)
        assert O.voxel_depth(pos) == want, (lo, hi)
    assert O.voxel_depth(np.zeros((0, 3), np.int16)) == 0


def test_empty_scene_octree(O):
    octree = O.create_octree(np.zeros((0, 3), np.int16), np.zeros((0, 4), np.uint8))
    assert len(octree) == 13 and (octree[5:] == 0).all() and octree[:5].view(np.float32)[3] == 1.0


# ---- .vox parser ------------------------------------------------------------------------------------
def chunk(cid, content=b"", children=b""):
    return cid + struct.pack("<II", len(content), len(children)) + content + children


def vstr(s):
    return struct.pack("<I", len(s)) + s


def matl(mid, **kv):
    body = struct.pack("<II", mid, len(kv))
    for k, v in kv.items():
        body += vstr(k.encode()) + vstr(v.encode())
    return chunk(b"MATL", body)


def make_vox(voxels=((1, 2, 3, 5),), size=(4, 5, 6), rgba=True, matls=(5,), emit=(), extra=b"", version=150, pack=None):
    body = b""
    if pack is not None:
        body += chunk(b"PACK", struct.pack("<I", pack))
    for _ in range(pack or 1):
        body += chunk(b"SIZE", struct.pack("<III", *size))
        body += chunk(b"XYZI", struct.pack("<I", len(voxels)) + b"".join(bytes(v) for v in voxels))
    body += extra
    if rgba:
        body += chunk(b"RGBA", b"".join(struct.pack("<I", 0xff000000 | (i * 3 % 256) << 16 | (i * 5 % 256) << 8 | i) for i in range(256)))
    for m in matls:
        body += matl(m, _type="_emit" if m in emit else "_diffuse")
    return b"VOX " + struct.pack("<i", version) + chunk(b"MAIN", b"", body)


def test_vox_adapter_axes_palette_material(O):
    data = make_vox(voxels=((1, 2, 3, 5), (7, 8, 9, 6)), matls=(5, 6), emit=(6,))
    pos, mrgb, size = O.voxels_from_vox(data)
    assert size == (4, 5, 6)
    assert pos.tolist() == [[1, 3, 2], [7, 9, 8]]                     # (x, z, y)  src/context.rs:927
    # RGBA entry i-1 lands in palette[i] (src/vox.rs:50-54); rgb = low three bytes (src/vox.rs:184-190)
    assert mrgb.tolist() == [[0, 4, 20, 12], [0x40, 5, 25, 15]]


def test_vox_default_palette_when_no_rgba(O):
    data = make_vox(voxels=((0, 0, 0, 1), (0, 0, 1, 2), (0, 0, 2, 255)), rgba=False, matls=(1, 2, 255))
    _, mrgb, _ = O.voxels_from_vox(data)
    assert mrgb[:, 1:].tolist() == [[0xff, 0xff, 0xff], [0xff, 0xff, 0xcc], [0x11, 0x11, 0x11]]


@needs_reference
def test_default_palette_equals_reference_table(O):
    text = open(os.path.join(REFERENCE, "src/vox.rs")).read()
    table = text[text.index("DEFAULT_PALETTE: [u32; 256]"):]
    table = table[:table.index("];")]
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]{8})", table)]
    assert len(words) == 256 and O.default_palette().tolist() == words


def test_vox_pack_uses_model_zero_and_skips_unknown_chunks(O):
    data = make_vox(pack=2, extra=chunk(b"nTRN", b"\x01\x02\x03") + chunk(b"LAYR", b"", b"zz"))
    pos, _, _ = O.voxels_from_vox(data)
    assert pos.tolist() == [[1, 3, 2]]


def test_vox_errors(O):
    good = make_vox()

    def code(data):
        with pytest.raises(O.OracleError) as e:
            O.voxels_from_vox(data)
        return e.value.code

    assert code(b"VOXX" + good[4:]) == -1                                  # invalid magic number
    assert code(make_vox(version=200)) == -2                               # unsupported version
    assert code(b"VOX " + struct.pack("<i", 150) + chunk(b"PACK")) == -3   # missing MAIN
    assert code(good[:30]) == -4 and code(good[:-3]) == -4                 # unexpected end of file
    assert code(b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", chunk(b"XYZI", struct.pack("<I", 0)))) == -5
    bad_type = b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", chunk(b"SIZE", struct.pack("<III", 1, 1, 1)) +
                                                          chunk(b"XYZI", struct.pack("<I", 0)) + matl(1, _type="_glass"))
    assert code(bad_type) == -6                                            # unsupported material type
    bad_flux = b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", chunk(b"SIZE", struct.pack("<III", 1, 1, 1)) +
                                                          chunk(b"XYZI", struct.pack("<I", 0)) + matl(1, _flux="abc"))
    assert code(bad_flux) == -6
    assert code(make_vox(matls=())) == -7                                  # colour without MATL: reference panics


@needs_reference
@pytest.mark.parametrize("name", sorted(KAT))
def test_reference_vox_files_reproduce_fixtures(O, scenes, name):
    data = open(os.path.join(REFERENCE, "vox", name + ".vox"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == KAT[name]["vox_sha256"]
    pos, mrgb, size = O.voxels_from_vox(data)
    fpos, fmrgb, fsize = scenes.load_scene(name)
    assert size == fsize and np.array_equal(pos, fpos) and np.array_equal(mrgb, fmrgb)


@needs_reference
def test_every_reference_scene_parses(O):
    # SURVEY.md Appendix C "other files": voxel counts
    want = {"chr_knight": 398, "chr_sword": 334, "custom": 13382, "doom": 3894, "monu1": 156942, "monu9": 32832,
            "nature": 75835, "room": 82536, "shelf": 1602, "teapot": 28411}
    for name, n in want.items():
        pos, mrgb, _ = O.voxels_from_vox(open(os.path.join(REFERENCE, "vox", name + ".vox"), "rb").read())
        assert len(pos) == n, name


def test_camera_axis_scaled(O):
    # src/camera.rs:12-28: right = norm((0,1,0) x fwd), up = fwd x right, forward_ray = -w/2 R + h/2 U + (h/2)/tan(fov/2) F
    fov = float(np.float32(70.0) * (np.float32(np.pi) / np.float32(180.0)))
    b = O.camera_axis_scaled((0, 0, -2), (0, 0, 1), fov, 800, 600)
    r, u, f = b[0:3], b[3:6], b[6:9]
    assert r.tolist() == [1, 0, 0] and u.tolist() == [0, 1, 0]
    assert f[0] == -400 and f[1] == 300 and abs(f[2] - 300 / np.tan(np.deg2rad(35.0))) < 1e-3
    b = O.camera_axis_scaled((1, 2, 3), (0.3, -0.2, 0.9), fov, 1920, 1080)
    r, u = b[0:3].astype(np.float64), b[3:6].astype(np.float64)
    d = np.array([0.3, -0.2, 0.9]) / np.linalg.norm([0.3, -0.2, 0.9])
    assert abs(np.dot(r, u)) < 1e-6 and abs(np.dot(r, d)) < 1e-6 and abs(np.linalg.norm(r) - 1) < 1e-6 and r[1] == 0
    # pixel (w/2, h/2) looks along the camera direction
    centre = 960 * r - 540 * u + b[6:9]
    assert np.allclose(centre / np.linalg.norm(centre), d, atol=1e-6)
