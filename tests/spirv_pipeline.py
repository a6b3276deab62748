"""Test infrastructure: the reference's frame loop (Context::render, src/context.rs:2014-2043) over its COMPILED shaders — the three
SPIR-V modules it hands to the GPU (shaders/{voxels,temporal,denoise}.comp.spv), executed by oracle/ospirv.cpp — and the cases that
tests/golden/spirv_exec/ holds their outputs for.  Used by tests/golden/make_spirv_exec_fixture.py (writes the fixtures; needs
/root/reference), tests/test_oracle_spirv_exec.py (oracle against fixtures; compiled shaders against oracle where the reference is
mounted) and tests/test_gpu_spirv_goldens.py (the HIP path against the fixtures).  Nothing here is imported by the product."""
import os

import numpy as np

SHADERS = "/root/reference/shaders"
MAX_BOUNCES = 3      # `#define MAX_BOUNCES 3` (voxels.comp:4) is compiled into the module


def have_shaders():
    return all(os.path.isfile(os.path.join(SHADERS, f"{n}.comp.spv")) for n in ("voxels", "temporal", "denoise"))


def module(name):
    with open(os.path.join(SHADERS, f"{name}.comp.spv"), "rb") as f:
        return f.read()


def with_max_bounces(module_bytes, bounces):
    """voxels.comp.spv with its loop bound re-specialised.  `#define MAX_BOUNCES 3` (voxels.comp:4) reaches the module as ONE operand: the
    bound of `for (bounce = 0; bounce < MAX_BOUNCES; bounce++)` (voxels.comp:309) is the OpSLessThan of main() whose right-hand side is
    the integer constant 3.  BASELINE's configs ask for 4 and 8 bounces; this inserts an integer constant of that value and points
    that one operand at it — nothing else of the module changes (bounces = 3 gives the module's own outputs: tested)."""
    import struct
    w = list(struct.unpack("<%dI" % (len(module_bytes) // 4), module_bytes))
    insts, i = [], 5
    while i < len(w):
        insts.append((i, w[i] & 0xffff, w[i] >> 16))
        i += w[i] >> 16
    int_types = {w[at + 1] for at, op, n in insts if op == 21 and w[at + 2] == 32 and w[at + 3] == 1}           # OpTypeInt 32 signed
    threes = [(at, w[at + 2]) for at, op, n in insts if op == 43 and w[at + 1] in int_types and w[at + 3] == 3]   # OpConstant %int 3
    assert len(threes) == 1
    const_at, three = threes[0]
    main_id = next(w[at + 2] for at, op, n in insts if op == 15)                                                # OpEntryPoint
    inside, uses = False, []
    for at, op, n in insts:
        if op == 54:
            inside = w[at + 2] == main_id
        if inside and op == 177 and w[at + 4] == three:                                                         # OpSLessThan x, 3
            uses.append(at)
    assert len(uses) == 1, uses
    new_id = w[3]
    w[3] += 1
    w[uses[0] + 4] = new_id
    at = const_at + 4
    w[at:at] = [(4 << 16) | 43, w[const_at + 1], new_id, int(bounces)]
    return struct.pack("<%dI" % len(w), *w)


def _block(struct, size):
    """A uniform block's bytes, padded to the size the module's layout may read (std140 rounds a block up to 16)."""
    raw = np.frombuffer(bytes(struct), np.uint8)
    return np.concatenate([raw, np.zeros(size - len(raw), np.uint8)])


def spirv_trace(O, octree, noise, u, w, h, crop=None, flags=0, nthreads=None, bounces=MAX_BOUNCES):
    """voxels.comp.spv over crop = (x0, y0, x1, y1) of a w x h frame -> colour, normal/depth, albedo (float32[ch, cw, 4]), instructions.
    bounces != 3: the module with its loop bound re-specialised (with_max_bounces)."""
    x0, y0, x1, y1 = crop or (0, 0, w, h)
    out = [np.zeros((y1 - y0, x1 - x0, 4), np.float32) for _ in range(3)]
    b = [O.spirv_image(k, out[k], (w, h), (x0, y0)) for k in range(3)]                     # bindings 0-2: voxels.comp:17-25
    b += [O.spirv_buffer(3, _block(u, 160)), O.spirv_buffer(4, np.zeros(64, np.uint8)),      # uniforms, old_uniforms (unused by main)
          O.spirv_buffer(5, octree), O.spirv_buffer(6, noise)]                              # octree_data, randomness
    mod = module("voxels") if bounces == MAX_BOUNCES else with_max_bounces(module("voxels"), bounces)
    n = O.spirv_dispatch(mod, b, x0, y0, x1, y1, flags=flags, nthreads=nthreads)
    return out[0], out[1], out[2], n


def spirv_temporal(O, color, nd, old_color, old_nd, cam16, old_cam16, tu, flags=0):
    """temporal.comp.spv over a whole frame -> the accumulated colour (rgb + blending)."""
    h, w = color.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    b = [O.spirv_sampler(0), O.spirv_image(1, old_color, sampled=True), O.spirv_image(2, np.array(color)), O.spirv_image(3, out),
         O.spirv_image(4, old_nd, sampled=True), O.spirv_image(5, np.array(nd)), O.spirv_buffer(6, _block(tu, 16)),
         O.spirv_buffer(7, np.array(cam16, np.float32)), O.spirv_buffer(8, np.array(old_cam16, np.float32))]
    O.spirv_dispatch(module("temporal"), b, 0, 0, w, h, flags=flags)
    return out


def spirv_denoise(O, colors, nd, albedo, cam16, du, flags=0):
    """denoise.comp.spv over a whole frame."""
    h, w = colors.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    b = [O.spirv_image(0, out), O.spirv_image(1, np.array(colors)), O.spirv_image(2, np.array(nd)), O.spirv_image(3, np.array(albedo)),
         O.spirv_buffer(4, np.array(cam16, np.float32)), O.spirv_buffer(5, _block(du, 16))]
    O.spirv_dispatch(module("denoise"), b, 0, 0, w, h, flags=flags)
    return out


class Pipeline:
    """Frame sequencing of Context::render / update_bindings (frame_number first; the old camera and the history are the previous frame's)
    over either set of kernels: `compiled=True` the reference's SPIR-V modules, False the oracle's restatement."""

    def __init__(self, O, octree, noise, w, h, radius, compiled, specularity=0.0, sun_strength=None, emit_strength=None, bounces=MAX_BOUNCES):
        self.O, self.w, self.h, self.compiled, self.bounces = O, w, h, compiled, bounces
        self.octree, self.noise = octree, noise
        self.u = O.Uniforms.default()
        self.u.specularity = specularity
        if sun_strength is not None:
            self.u.sun_strength = sun_strength
        if emit_strength is not None:
            self.u.emit_strength = emit_strength
        self.du = O.Denoise.default()
        self.du.radius = radius
        self.tu = O.Temporal.default()
        self.old_c = np.zeros((h, w, 4), np.float32)
        self.old_nd = np.zeros((h, w, 4), np.float32)
        self.old_cam16 = np.zeros(16, np.float32)
        self.frame = 0

    def render(self, cam):
        O, w, h = self.O, self.w, self.h
        self.frame += 1
        self.u.frame_number = self.frame
        self.u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
        cam16 = self.u.camera16()
        if self.compiled:
            color, nd, alb, _ = spirv_trace(O, self.octree, self.noise, self.u, w, h, bounces=self.bounces)
            accum = spirv_temporal(O, color, nd, self.old_c, self.old_nd, cam16, self.old_cam16, self.tu)   # frame 1: all-zero history and old camera (U3)
            den = spirv_denoise(O, accum, nd, alb, cam16, self.du)
        else:
            color, nd, alb, _ = O.trace(self.octree, self.noise, self.u, w, h, self.bounces, crop=(0, 0, w, h))
            accum = O.temporal(color, nd, self.old_c, self.old_nd, cam16, self.old_cam16, self.tu, self.frame > 1)
            den = O.denoise(accum, nd, alb, cam16, self.du)
        self.old_c, self.old_nd, self.old_cam16 = accum, nd, cam16
        return color, nd, alb, accum, den


def cap_scene():
    """A row of 4 096 voxels along x: rays beside it reach the 2 048-trip cap (tests/test_gpu_trace.py: cap_scene)."""
    n = 4096
    pos = np.zeros((n, 3), np.int16)
    pos[:, 0] = np.arange(n)
    return pos, np.tile(np.array([[0, 200, 100, 50]], np.uint8), (n, 1))


def _f32(*v):
    return np.array(v, np.float32)


def cases(scenes):
    """name -> dict(scene, w, h, radius, frames: [(position, direction, fov)], uniforms overrides): the sequences the fixtures hold."""
    out = {}

    def close(name):
        _, _, size = scenes.load_scene(name)
        return scenes.close_camera(size)

    p, d, fov = close("castle")
    out["castle_moving_r2"] = dict(scene="castle", w=96, h=64, radius=2, frames=[
        (p + np.float32(0.05 * f) * _f32(1, 0.2, 0.1), d + np.float32(0.015 * f) * _f32(0, 1, 0), fov) for f in range(3)])
    p, d, fov = close("menger")
    out["menger_static_r8"] = dict(scene="menger", w=80, h=48, radius=8, frames=[(p, d, fov)] * 2)
    p, d, fov = close("monu10")
    out["monu10_specular_r1"] = dict(scene="monu10", w=80, h=48, radius=1, frames=[(p, d, fov)] * 2, specularity=0.4)
    p, d, fov = close("room")
    out["room_sun_off_r0"] = dict(scene="room", w=80, h=48, radius=0, frames=[(p, d, fov)] * 2, sun_strength=0.0, emit_strength=3.0)
    # BASELINE's bounce counts (the module's loop bound re-specialised, with_max_bounces): configs[1]'s scene and view at 4 bounces,
    # config 3's scene at 8 bounces with its 5 x 5 window
    _, _, size = scenes.load_scene("menger")
    out["menger_bench_view_4_bounces_r0"] = dict(scene="menger", w=96, h=54, radius=0, frames=[scenes.bench_camera(size)] * 2, bounces=4)
    p, d, fov = close("monu10")
    out["monu10_8_bounces_r2"] = dict(scene="monu10", w=80, h=48, radius=2, frames=[(p, d, fov)] * 2, bounces=8)
    # an axis-aligned camera on integer coordinates sends its centre rays exactly along +z through node mid-planes: (center - origin) *
    # (1 / 0) = 0 * inf = NaN (voxels.comp:140,191) goes through the compiled code as it does through the oracle, NaN outputs included
    out["zero_times_inf_r0"] = dict(scene="8x8x8", w=64, h=64, radius=0, frames=[(_f32(1, 1, -5), _f32(0, 0, 1), 1.0)])
    # the 2 048-trip cap (voxels.comp:166-169) returns true with `normal` unwritten (U1): the module then reads memory this interpreter
    # zeroed at the invocation's start, which is the oracle's definition (normal = 0)
    out["cap_row_r0"] = dict(scene="cap", w=96, h=64, radius=0, frames=[(_f32(-1, 0.6, 0.25), _f32(1, 0, 0), 0.01)])
    return out


def build_case(O, scenes, noise, spec, compiled):
    if spec["scene"] == "cap":
        pos, mrgb = cap_scene()
    else:
        pos, mrgb, _ = scenes.load_scene(spec["scene"])
    pipe = Pipeline(O, O.create_octree(pos, mrgb), noise, spec["w"], spec["h"], spec["radius"], compiled,
                    specularity=spec.get("specularity", 0.0), sun_strength=spec.get("sun_strength"), emit_strength=spec.get("emit_strength"),
                    bounces=spec.get("bounces", MAX_BOUNCES))
    return pipe, (pos, mrgb)


def run_case(O, scenes, noise, spec, compiled):
    """{key: image}: frame 1's three trace outputs, every frame's accumulated and denoised colour."""
    pipe, _ = build_case(O, scenes, noise, spec, compiled)
    out = {}
    for f, cam in enumerate(spec["frames"], 1):
        color, nd, alb, accum, den = pipe.render(cam)
        if f == 1:
            out["f1_color"], out["f1_nd"], out["f1_albedo"] = color, nd, alb
        out[f"f{f}_accum"], out[f"f{f}_denoised"] = accum, den
    return out


# ---- BASELINE's frames at their full size: too large to keep as images, kept as hashes --------------------------------------------
# The compiled voxels.comp over the WHOLE frame of configs[1] and of configs 3 and 4 (their scenes, views, sizes and bounce counts; one
# sample, frame 1), hashed in slabs of rows so that a slab can be checked on its own (tests/golden/spirv_exec/full_size.json).
SLAB_ROWS = 120
FULL_SIZE = [
    dict(name="configs1_menger_1080p_4_bounces", scene="menger", view="bench", w=1920, h=1080, bounces=4),
    dict(name="config3_monu10_4k_8_bounces", scene="monu10", view="close", w=3840, h=2160, bounces=8),
    dict(name="config4_castle_4k_8_bounces", scene="castle", view="close", w=3840, h=2160, bounces=8),
]


def canonical_sha256(img, node_channel=False):
    """sha256 of an rgba32f image's bits with what IEEE leaves open made canonical: every NaN one pattern, -0 as +0.  node_channel: the
    albedo image's .w holds the leaf WORD (intBitsToFloat, voxels.comp:396) — an integer, hashed as it is."""
    import hashlib
    a = np.ascontiguousarray(img, np.float32)
    v = a.view(np.uint32).copy()
    keep = v[..., 3].copy()
    v[np.isnan(a)] = 0x7fc00000
    v[v == 0x80000000] = 0
    if node_channel:
        v[..., 3] = keep
    return hashlib.sha256(v.tobytes()).hexdigest()


def full_size_uniforms(O, scenes, case):
    pos, mrgb, size = scenes.load_scene(case["scene"])
    cam = scenes.bench_camera(size) if case["view"] == "bench" else scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], case["w"], case["h"]))
    u.frame_number = 1
    return pos, mrgb, cam, u


def full_size_slab_hashes(O, scenes, noise, case, slabs, compiled):
    """{slab index: {"color" | "nd" | "albedo": sha256}} of the listed slabs, through the compiled module or through the oracle."""
    pos, mrgb, _, u = full_size_uniforms(O, scenes, case)
    octree = O.create_octree(pos, mrgb)
    out = {}
    for s in slabs:
        crop = (0, s * SLAB_ROWS, case["w"], min((s + 1) * SLAB_ROWS, case["h"]))
        if compiled:
            color, nd, alb, _ = spirv_trace(O, octree, noise, u, case["w"], case["h"], crop=crop, bounces=case["bounces"])
        else:
            color, nd, alb, _ = O.trace(octree, noise, u, case["w"], case["h"], case["bounces"], crop=crop)
        out[s] = {"color": canonical_sha256(color), "nd": canonical_sha256(nd), "albedo": canonical_sha256(alb, node_channel=True)}
    return out


# Config 3's frame loop at its full size (monu10, 3840 x 2160, 8 bounces; camera at rest, one sample per frame): two frames through all
# three compiled modules — frame 2's temporal stage blends frame 1's history — hashed like the trace images above: the accumulated
# colour of both frames, frame 2 denoised with the 5 x 5 window (whole frame) and with the 17 x 17 window (rows R8_ROWS only: the
# interpreter needs ~45 000 instructions per pixel there).
PIPELINE_CASE = dict(name="config3_monu10_4k_8_bounces_frame_loop", scene="monu10", view="close", w=3840, h=2160, bounces=8)
R8_ROWS = (1040, 1080)


def slab_hashes(img, rows=SLAB_ROWS):
    return [canonical_sha256(img[s:s + rows]) for s in range(0, img.shape[0], rows)]


def spirv_denoise_rows(O, colors, nd, albedo, cam16, du, y0, y1):
    """denoise.comp.spv for the rows [y0, y1) of a frame whose inputs are bound whole."""
    h, w = colors.shape[:2]
    out = np.zeros((y1 - y0, w, 4), np.float32)
    b = [O.spirv_image(0, out, (w, h), (0, y0)), O.spirv_image(1, np.array(colors)), O.spirv_image(2, np.array(nd)), O.spirv_image(3, np.array(albedo)),
         O.spirv_buffer(4, np.array(cam16, np.float32)), O.spirv_buffer(5, _block(du, 16))]
    O.spirv_dispatch(module("denoise"), b, 0, y0, w, y1)
    return out


def pipeline_hashes(O, scenes, noise, compiled, case=PIPELINE_CASE, log=None):
    """{"f1_accum", "f2_accum", "f2_denoised_r2": [sha256 per slab], "f2_denoised_r8_rows": sha256 of rows R8_ROWS}"""
    pos, mrgb, cam, u = full_size_uniforms(O, scenes, case)
    pipe = Pipeline(O, O.create_octree(pos, mrgb), noise, case["w"], case["h"], 2, compiled, bounces=case["bounces"])
    out = {}
    for f in (1, 2):
        color, nd, alb, accum, den = pipe.render(cam)
        out[f"f{f}_accum"] = slab_hashes(accum)
        if log:
            log(f"frame {f} done")
    out["f2_denoised_r2"] = slab_hashes(den)
    du = O.Denoise.default()
    du.radius = 8
    cam16 = pipe.u.camera16()
    if compiled:
        rows = spirv_denoise_rows(O, accum, nd, alb, cam16, du, *R8_ROWS)
    else:
        rows = O.denoise(accum, nd, alb, cam16, du)[R8_ROWS[0]:R8_ROWS[1]]
    out["f2_denoised_r8_rows"] = canonical_sha256(rows)
    return out


# ---- every scene file once: the trace stage's three images of a small frame (tests/golden/spirv_exec/scene_sweep.npz) --------------
SWEEP = dict(w=64, h=40, frame_number=1)


def sweep_scenes():
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")
    return sorted(f[:-4] for f in os.listdir(golden) if f.endswith(".npz"))


def sweep_frame(O, scenes, noise, name, compiled):
    """(colour, normal/depth, albedo) of the scene's close view at SWEEP's size, 3 bounces, through the compiled module or the oracle."""
    pos, mrgb, size = scenes.load_scene(name)
    cam = scenes.close_camera(size)
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], SWEEP["w"], SWEEP["h"]))
    u.frame_number = SWEEP["frame_number"]
    octree = O.create_octree(pos, mrgb)
    if compiled:
        return spirv_trace(O, octree, noise, u, SWEEP["w"], SWEEP["h"])[:3]
    return O.trace(octree, noise, u, SWEEP["w"], SWEEP["h"], MAX_BOUNCES, crop=(0, 0, SWEEP["w"], SWEEP["h"]))[:3]
