"""Host-side mirror of the reference's render-loop interface, over the C ABI of libvxrt.so.

The reference is a Rust binary whose host code (src/main.rs, src/context.rs, src/camera.rs, src/vox.rs)
drives the GPU through wgpu.  Rust is not available in this image, so the host side above the C ABI is
this thin ctypes layer; it keeps the reference's names and argument meaning:

    Camera{position, direction, fov}                      src/camera.rs:5-9
    Uniforms / TemporalUniforms / DenoiseUniforms         src/context.rs:425-525, 304-325
    Context.recreate_octree(voxels) / load_vox(path)      src/context.rs:799-810, 1817-1821
    Context.render()                                      src/context.rs:2004-2075
    Context.resize(w, h)                                  src/context.rs:1430-1461

There is NO CPU fallback: if libvxrt.so is missing or no HIP device is present, construction raises.
Nothing here imports the test oracle.
"""
import ctypes as C
import os

# One HIP stream per trace launch in flight; a process's streams share GPU_MAX_HW_QUEUES hardware queues (default 4) with everything
# else in it (RCCL, torch).  8 keeps busy streams apart.  Only effective before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

from . import _build

_LIB = None


class VxrtError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        self.detail = detail
        super().__init__(f"{where}: status {status} ({detail})")


# vxrt_status (include/vxrt.h)
OK, E_INVALID, E_DEVICE = 0, -1, -2
E_VOX_MAGIC, E_VOX_VERSION, E_VOX_NOMAIN, E_VOX_EOF, E_VOX_CHUNK = -10, -11, -12, -13, -14
E_VOX_MATERIAL, E_VOX_NOMATL, E_VOX_NOMODEL, E_IO, E_SCENE, E_NOSCENE, E_NOISE = -15, -16, -17, -18, -20, -21, -30

# vxrt_image
SAMPLED_COLOR, NORMAL_DEPTH, ALBEDO_NODE, ACCUM_COLOR, DENOISED = range(5)
# render flags
TRACE, TEMPORAL, DENOISE, ALL, TIMED = 1, 2, 4, 7, 8
DENOISE_INTERIOR, DENOISE_EDGE = 16, 32   # the denoise stage in two launches around a halo exchange (vxrt.h)
FEATURE_VARIANTS = 1                      # vxrt_build_features: tracers 2 / 3 / 5 and the wide scene records are in the library

NOISE_LEN = 512 * 128 * 128
DEFAULT_NOISE_SEED = 0x5EED0001


class Uniforms(C.Structure):
    """vxrt_uniforms == Uniforms, src/context.rs:425-469 (148 bytes)."""
    _fields_ = [
        ("camera_origin", C.c_float * 4), ("camera_right", C.c_float * 4), ("camera_up", C.c_float * 4),
        ("camera_forward", C.c_float * 4), ("light", C.c_float * 4), ("global_time", C.c_float),
        ("still_sample", C.c_uint32), ("frame_number", C.c_uint32), ("emit_strength", C.c_float),
        ("sun_strength", C.c_float), ("sun_size", C.c_float), ("sun_yaw", C.c_float), ("sun_pitch", C.c_float),
        ("sun_color", C.c_float * 4), ("sky_color", C.c_float * 4), ("specularity", C.c_float),
    ]

    @classmethod
    def default(cls):
        u = cls()
        lib().vxrt_default_uniforms(C.byref(u))
        return u


class TemporalUniforms(C.Structure):
    """vxrt_temporal == TemporalUniforms, src/context.rs:502-515."""
    _fields_ = [("sample_blending", C.c_float), ("maximum_blending", C.c_float),
                ("blending_distance_cutoff", C.c_float)]

    @classmethod
    def default(cls):
        t = cls()
        lib().vxrt_default_temporal(C.byref(t))
        return t


class DenoiseUniforms(C.Structure):
    """vxrt_denoise == DenoiseUniforms, src/context.rs:304-314."""
    _fields_ = [("radius", C.c_uint32), ("sigma_distance", C.c_float), ("sigma_range", C.c_float),
                ("albedo_factor", C.c_float)]

    @classmethod
    def default(cls):
        d = cls()
        lib().vxrt_default_denoise(C.byref(d))
        return d


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("device", C.c_int32), ("max_bounces", C.c_uint32),
                ("noise_seed", C.c_uint32), ("noise", C.c_void_p), ("rank", C.c_uint32), ("nranks", C.c_uint32),
                ("band_rows", C.c_uint32), ("frames_in_flight", C.c_uint32), ("tracer", C.c_uint32),
                ("frames_per_launch", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("rays", C.c_uint64), ("pixels", C.c_uint64), ("trace_ms", C.c_double),
                ("temporal_ms", C.c_double), ("denoise_ms", C.c_double), ("timed_frames", C.c_uint64),
                ("timed_launches", C.c_uint64), ("scene_bytes", C.c_uint64), ("noise_bytes", C.c_uint64), ("local_rows", C.c_uint32),
                ("octree_depth", C.c_uint32), ("octree_nodes", C.c_uint64), ("wide_nodes", C.c_uint64), ("scene_format", C.c_uint32),
                ("node_order", C.c_uint32), ("queue_bytes", C.c_uint64),
                ("queue_overflow_paths", C.c_uint64), ("halo_pack_ms", C.c_double), ("halo_unpack_ms", C.c_double),
                ("halo_exchanges", C.c_uint64), ("cull_box_valid", C.c_uint32), ("cull_box_min", C.c_float * 3),
                ("cull_box_max", C.c_float * 3), ("frame_lane_launches", C.c_uint32), ("split_launches", C.c_uint32)]


OPT_DENOISE_MODE, OPT_TAIL_CAPACITY, OPT_SCENE_FORMAT, OPT_HALO_ROWS, OPT_SKY_CULL, OPT_FRAME_LANES = 1, 2, 3, 4, 5, 6
(OPT_TILE_ORDER, OPT_TILE_SPREAD, OPT_TRACE_BLOCKS, OPT_TAIL_FROM, OPT_TAIL_SPLIT, OPT_HOST_SCENE_BUILD, OPT_TRACER_OVERRIDE,
 OPT_TRACE_SPLIT, OPT_PATH_BLOCKS, OPT_SHADE_BLOCKS, OPT_RAYS_PER_WAVE, OPT_NODE_ORDER, OPT_HEAD_STAGGER, OPT_LONG_TILES, OPT_FUSED_TAIL,
 OPT_TRACE_PRIORITY, OPT_XCD_AFFINITY) = range(7, 24)      # include/vxrt_debug.h (the options of experiments)
TILE_SPREAD_AUTO = 0xffffffff


class Tuning(C.Structure):
    """vxrt_tuning: an (option, value) pair for vxrt_create_tuned."""
    _fields_ = [("option", C.c_uint32), ("value", C.c_uint32)]


# The library reads no environment variable.  The A/B scripts and the parity tests of the scheduling variants, which used to steer
# it through VXRT_* variables, say what they want through Context(tuning=...) — or, to keep their command lines, switch this
# translation on (enable_env_knobs(): tests/conftest.py; VXRT_ENV_KNOBS=1: scripts/*.sh): environment name -> create-time option.
ENV_KNOBS = {"VXRT_SKY_CULL": OPT_SKY_CULL, "VXRT_HALO_ROWS": OPT_HALO_ROWS, "VXRT_TRACE_VARIANT": OPT_TRACER_OVERRIDE,
             "VXRT_PATH_BLOCKS": OPT_PATH_BLOCKS, "VXRT_TAIL_FROM": OPT_TAIL_FROM, "VXRT_TAIL_SPLIT": OPT_TAIL_SPLIT,
             "VXRT_TAIL_CAPACITY": OPT_TAIL_CAPACITY, "VXRT_WIDE": OPT_SCENE_FORMAT, "VXRT_SHADE_BLOCKS": OPT_SHADE_BLOCKS,
             "VXRT_RAYS_PER_WAVE": OPT_RAYS_PER_WAVE, "VXRT_TRACE_BLOCKS": OPT_TRACE_BLOCKS, "VXRT_FRAME_LANES": OPT_FRAME_LANES,
             "VXRT_SPREAD": OPT_TILE_SPREAD, "VXRT_TILE_ORDER": OPT_TILE_ORDER, "VXRT_TRACE_SPLIT": OPT_TRACE_SPLIT,
             "VXRT_HOST_BUILD": OPT_HOST_SCENE_BUILD, "VXRT_NODE_ORDER": OPT_NODE_ORDER, "VXRT_HEAD_STAGGER": OPT_HEAD_STAGGER, "VXRT_LONG_TILES": OPT_LONG_TILES, "VXRT_FUSED_TAIL": OPT_FUSED_TAIL,
             "VXRT_TRACE_PRIORITY": OPT_TRACE_PRIORITY, "VXRT_XCD_AFFINITY": OPT_XCD_AFFINITY}
_env_knobs_enabled = os.environ.get("VXRT_ENV_KNOBS") == "1"      # the explicit opt-in of the A/B scripts (scripts/*.sh)


def enable_env_knobs(on=True):
    """Tests and A/B scripts: let Context() translate the VXRT_* variables of ENV_KNOBS (and VXRT_INFLIGHT / VXRT_BATCH) into
    vxrt_create_tuned options.  Off by default: a host process does not inherit behaviour from its environment."""
    global _env_knobs_enabled
    _env_knobs_enabled = bool(on)


def tuning_from_env(environ=None):
    """[(option, value)] for the VXRT_* variables set in `environ`."""
    environ = os.environ if environ is None else environ
    out = []
    for name, opt in ENV_KNOBS.items():
        v = environ.get(name)
        if v is None or v == "":
            continue
        n = int(v, 0)
        if opt == OPT_TILE_SPREAD and n < 0:
            n = TILE_SPREAD_AUTO
        out.append((opt, max(n, 0) & 0xffffffff))
    return out


class HaloInfo(C.Structure):
    """vxrt_halo_info (include/vxrt.h)."""
    _fields_ = [("rows", C.c_uint32), ("slots", C.c_uint32), ("bytes_per_pixel", C.c_uint32), ("interior_tile_rows", C.c_uint32),
                ("edge_tile_rows", C.c_uint32), ("max_rows", C.c_uint32), ("message_bytes", C.c_uint64)]


class Camera:
    """Camera, src/camera.rs:5-9.  Default = the reference's start camera, src/context.rs:618-622."""

    def __init__(self, position=(0.0, 0.0, -2.0), direction=(0.0, 0.0, 1.0),
                 fov=float(np.float32(70.0) * (np.float32(np.pi) / np.float32(180.0)))):
        self.position = np.asarray(position, np.float32)
        self.direction = np.asarray(direction, np.float32)
        self.fov = float(np.float32(fov))

    def axis_scaled(self, width, height):
        """Camera::axis_scaled (src/camera.rs:19-28), evaluated by the library's host code."""
        r, u, f = (np.zeros(3, np.float32) for _ in range(3))
        _check(lib().vxrt_camera_axis_scaled(_p(self.position), _p(self.direction), C.c_float(self.fov),
                                             C.c_uint32(width), C.c_uint32(height), _p(r), _p(u), _p(f)),
               "vxrt_camera_axis_scaled")
        return r, u, f


_LOADED = {}     # path -> CDLL: every library this process has loaded (each keeps its own state; handles are never unloaded)


def _load(path):
    path = os.path.abspath(path)
    L = _LOADED.get(path)
    if L is None:
        L = C.CDLL(path)
        L.vxrt_last_error.restype = C.c_char_p
        L.vxrt_status_string.restype = C.c_char_p
        L.vxrt_abi_version.restype = C.c_uint32
        L.vxrt_build_features.restype = C.c_uint32
        L.vxrt_status_string.argtypes = [C.c_int]
        _LOADED[path] = L
    return L


def lib():
    """Load libvxrt.so, building it with hipcc first if it is missing or was built from other sources than the tree holds
    (compared by content hash, _build.needs_build: a stale binary must never run silently).  No fallback: without the
    library there is no product."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("VXRT_LIB")  # A/B builds of the library (scripts/ab_build.sh); the product build otherwise
        if not path:
            path = _build.build()
        _LIB = _load(path)
    return _LIB


def use_library(path=None):
    """Tests: make `path` (None = the product build, or VXRT_LIB) the library new contexts and the module-level helpers use.
    A context keeps the library it was created in.  Returns the path now in use."""
    global _LIB
    if path is None:
        _LIB = None
        lib()
    else:
        _LIB = _load(path)
    return _LIB._name


def variants_library():
    """Path of the -DVXRT_VARIANTS=1 build (tracers 2 / 3 / 5, the wide scene records: the schedules and the scene format that
    measured slower and stay out of the product), built first if it is missing or stale.  Test infrastructure: the parity cases of
    those variants load it beside the product library."""
    return _build.build(variants=True)


def build_features():
    """vxrt_build_features: FEATURE_* bits of the loaded library."""
    return int(lib().vxrt_build_features())


def has_variants():
    return bool(build_features() & FEATURE_VARIANTS)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(status, where):
    if status != 0:
        raise VxrtError(status, where, (lib().vxrt_last_error() or b"").decode(errors="replace"))


# ---- host-only helpers (no GPU needed) -------------------------------------------------------------------
def vox_to_voxels(data: bytes):
    """vox::parse + Context::voxels_from_vox -> (pos int16[n,3], mrgb uint8[n,4], (sx,sy,sz))."""
    buf = np.frombuffer(data, np.uint8)
    n = C.c_size_t(0)
    size = np.zeros(3, np.uint32)
    _check(lib().vxrt_vox_to_voxels(_p(buf), C.c_size_t(len(data)), None, None, C.c_size_t(0), C.byref(n), _p(size)),
           "vxrt_vox_to_voxels")
    pos = np.zeros((n.value, 3), np.int16)
    mrgb = np.zeros((n.value, 4), np.uint8)
    _check(lib().vxrt_vox_to_voxels(_p(buf), C.c_size_t(len(data)), _p(pos), _p(mrgb), C.c_size_t(n.value),
                                    C.byref(n), _p(size)), "vxrt_vox_to_voxels")
    return pos, mrgb, tuple(int(s) for s in size)


def build_octree(pos, mrgb):
    """Context::create_octree -> (int32 words, depth)."""
    pos = np.ascontiguousarray(pos, np.int16)
    mrgb = np.ascontiguousarray(mrgb, np.uint8)
    n = C.c_size_t(0)
    depth = C.c_uint32(0)
    _check(lib().vxrt_build_octree(_p(pos), _p(mrgb), C.c_size_t(len(pos)), None, C.c_size_t(0), C.byref(n),
                                   C.byref(depth)), "vxrt_build_octree")
    words = np.zeros(n.value, np.int32)
    _check(lib().vxrt_build_octree(_p(pos), _p(mrgb), C.c_size_t(len(pos)), _p(words), C.c_size_t(n.value),
                                   C.byref(n), C.byref(depth)), "vxrt_build_octree")
    return words, int(depth.value)


def build_records(pos, mrgb):
    """The device scene formats of a voxel list, built on the host (vxrt_build_records) ->
    (svo uint32[n,2] = masks, base; wide uint32[m,4] = mask lo, mask hi, base, top; leaves int32[k]; depth)."""
    pos = np.ascontiguousarray(pos, np.int16)
    mrgb = np.ascontiguousarray(mrgb, np.uint8)
    ns, nw, nl, depth = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
    call = lambda svo, wide, leaves: _check(lib().vxrt_build_records(  # noqa: E731
        _p(pos), _p(mrgb), C.c_size_t(len(pos)), _p(svo), C.c_size_t(0 if svo is None else len(svo)), C.byref(ns), _p(wide),
        C.c_size_t(0 if wide is None else len(wide)), C.byref(nw), _p(leaves), C.c_size_t(0 if leaves is None else len(leaves)),
        C.byref(nl), C.byref(depth)), "vxrt_build_records")
    call(None, None, None)
    svo, wide, leaves = np.zeros((ns.value, 2), np.uint32), np.zeros((nw.value, 4), np.uint32), np.zeros(nl.value, np.int32)
    call(svo, wide, leaves)
    return svo, wide, leaves, int(depth.value)


def halo_rows_for_motion(cam_a, cam_b, width, height, near, band_rows):
    """vxrt_halo_rows_for_motion: cam_a / cam_b are Camera objects with the same fov (see distributed.halo_rows_for_motion)."""
    rows = C.c_uint32(0)
    _check(lib().vxrt_halo_rows_for_motion(_p(cam_a.position), _p(cam_a.direction), _p(cam_b.position), _p(cam_b.direction), C.c_float(cam_a.fov),
                                           C.c_uint32(width), C.c_uint32(height), C.c_float(near), C.c_uint32(band_rows), C.byref(rows)),
           "vxrt_halo_rows_for_motion")
    return int(rows.value)


def noise_table(seed=DEFAULT_NOISE_SEED, n=NOISE_LEN):
    out = np.zeros(n, np.float32)
    _check(lib().vxrt_noise_table(C.c_uint32(seed), _p(out), C.c_size_t(n)), "vxrt_noise_table")
    return out


def blue_noise(seed=DEFAULT_NOISE_SEED, size=128, first_layer=0, layers=512, device=0):
    """Void-and-cluster blue noise made on the GPU (include/vxrt_bluenoise.h) -> float32[layers, size, size]: the kind
    of table the reference loads from its missing resources/blue-noise-128.zip (src/context.rs:1016-1040)."""
    out = np.zeros((layers, size, size), np.float32)
    _check(lib().vxrt_blue_noise(C.c_int32(device), C.c_uint32(seed), C.c_uint32(size), C.c_uint32(first_layer),
                                 C.c_uint32(layers), _p(out)), "vxrt_blue_noise")
    return out


def load_blue_noise(path):
    """Context::load_blue_noise (src/context.rs:1042-1085): -> (image size, float32 pixels of all images appended)."""
    size, layers = C.c_uint32(0), C.c_uint32(0)
    enc = os.fsencode(path)
    _check(lib().vxrt_noise_zip_read(enc, None, C.c_size_t(0), C.byref(size), C.byref(layers)), "vxrt_noise_zip_read")
    out = np.zeros(layers.value * size.value * size.value, np.float32)
    _check(lib().vxrt_noise_zip_read(enc, _p(out), C.c_size_t(out.size), C.byref(size), C.byref(layers)), "vxrt_noise_zip_read")
    return int(size.value), out


def save_blue_noise(path, table, size=128):
    """Writes a table in the reference's archive format (one stored entry per layer)."""
    t = np.ascontiguousarray(table, np.float32).reshape(-1, size, size)
    _check(lib().vxrt_noise_zip_write(os.fsencode(path), _p(t), C.c_uint32(size), C.c_uint32(len(t))), "vxrt_noise_zip_write")


VOX_ALL_MODELS, VOX_LENIENT_MATERIALS, VOX_REBASE = 1, 2, 4


def vox_scene_to_voxels(data: bytes, flags=VOX_ALL_MODELS):
    """Whole MagicaVoxel scenes (every shape instance of the scene graph, see vxrt.h) ->
    (pos int16[n,3], mrgb uint8[n,4], (bounds_min, bounds_max)) in the renderer's axes."""
    buf = np.frombuffer(data, np.uint8)
    n = C.c_size_t(0)
    lo, hi = np.zeros(3, np.int32), np.zeros(3, np.int32)
    _check(lib().vxrt_vox_scene_to_voxels(_p(buf), C.c_size_t(len(data)), C.c_uint32(flags), None, None, C.c_size_t(0),
                                          C.byref(n), _p(lo), _p(hi)), "vxrt_vox_scene_to_voxels")
    pos = np.zeros((n.value, 3), np.int16)
    mrgb = np.zeros((n.value, 4), np.uint8)
    _check(lib().vxrt_vox_scene_to_voxels(_p(buf), C.c_size_t(len(data)), C.c_uint32(flags), _p(pos), _p(mrgb),
                                          C.c_size_t(n.value), C.byref(n), _p(lo), _p(hi)), "vxrt_vox_scene_to_voxels")
    return pos, mrgb, (tuple(int(v) for v in lo), tuple(int(v) for v in hi))


def default_scene_voxels(seed=1):
    """Context::create_voxels (src/context.rs:838-910), the reference's start-up scene, seeded."""
    n = C.c_size_t(0)
    _check(lib().vxrt_default_scene_voxels(C.c_uint32(seed), None, None, C.c_size_t(0), C.byref(n)), "vxrt_default_scene_voxels")
    pos = np.zeros((n.value, 3), np.int16)
    mrgb = np.zeros((n.value, 4), np.uint8)
    _check(lib().vxrt_default_scene_voxels(C.c_uint32(seed), _p(pos), _p(mrgb), C.c_size_t(n.value), C.byref(n)),
           "vxrt_default_scene_voxels")
    return pos, mrgb


def menger_voxels(level, mrgb=(0, 0xb0, 0xd0, 0x60), clip=0, emissive_period=0):
    """Voxel list of the (clipped) level-`level` Menger sponge; the list form of Context.set_menger's scene."""
    m = np.asarray(mrgb, np.uint8)
    n = C.c_size_t(0)
    args = (C.c_uint32(level), C.c_uint32(clip), _p(m), C.c_uint32(emissive_period))
    _check(lib().vxrt_menger_voxels_ex(*args, None, None, C.c_size_t(0), C.byref(n)), "vxrt_menger_voxels_ex")
    pos = np.zeros((n.value, 3), np.int16)
    out = np.zeros((n.value, 4), np.uint8)
    _check(lib().vxrt_menger_voxels_ex(*args, _p(pos), _p(out), C.c_size_t(n.value), C.byref(n)), "vxrt_menger_voxels_ex")
    return pos, out


def detmath_probe(fn, x, y=None, device=0):
    names = {"sin": 0, "cos": 1, "exp": 2, "log": 3, "pow": 4, "sqrt": 5, "div": 6, "tan": 7, "chain": 8, "hemi_y": 9, "hemi_z": 10, "mul": 11, "sub": 12, "flip": 13, "min": 14, "max": 15, "max0": 16, "sign": 17, "clamp": 18, "min0": 19}
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    _check(lib().vxrt_detmath_probe(C.c_int32(device), C.c_int32(names[fn]), _p(x), _p(y), _p(out),
                                    C.c_size_t(x.size)), "vxrt_detmath_probe")
    return out


class PinnedImage:
    """A float32 image in pinned host memory (vxrt_host_alloc / vxrt_host_free): the destination vxrt_read_async wants.  `.array`
    is a numpy view of it; valid until close() / garbage collection."""

    def __init__(self, L, shape):
        self._L = L
        n = int(np.prod(shape)) * 4
        self._ptr = C.c_void_p()
        st = L.vxrt_host_alloc(C.c_size_t(n), C.byref(self._ptr))
        if st != 0:
            raise VxrtError(st, "vxrt_host_alloc", (L.vxrt_last_error() or b"").decode(errors="replace"))
        buf = (C.c_float * (n // 4)).from_address(self._ptr.value) if n else (C.c_float * 0)()
        self.array = np.frombuffer(buf, np.float32).reshape(shape)

    def close(self):
        if self._ptr:
            self.array = None
            self._L.vxrt_host_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """The render context: what `Context` is in the reference (src/context.rs:198-268), minus the window.

    One context = one GPU.  `rank`/`nranks` make it render only its interleaved row bands of the frame.
    """

    def __init__(self, width, height, device=0, max_bounces=3, noise=None, noise_seed=DEFAULT_NOISE_SEED, rank=0,
                 nranks=1, band_rows=16, frames_in_flight=1, tracer=0, frames_per_launch=1, tuning=None):
        """tuning: [(OPT_*, value)] applied by vxrt_create_tuned before anything is allocated (scheduling options for experiments
        and tests; the image never depends on them)."""
        self._h = C.c_void_p()
        self._L = lib()      # the library this context lives in (tests load the -DVXRT_VARIANTS=1 build beside the product: use_library)
        tuning = list(tuning or [])
        if _env_knobs_enabled:
            tuning = tuning_from_env() + tuning
            frames_in_flight = int(os.environ.get("VXRT_INFLIGHT", frames_in_flight))
            frames_per_launch = int(os.environ.get("VXRT_BATCH", frames_per_launch))
        self.width, self.height = int(width), int(height)
        self.camera = Camera()
        cfg = Config(self.width, self.height, int(device), int(max_bounces), int(noise_seed), None, int(rank),
                     int(nranks), int(band_rows), int(frames_in_flight), int(tracer), int(frames_per_launch))
        keep = None
        if noise is not None:
            keep = np.ascontiguousarray(noise, np.float32)
            if keep.size != NOISE_LEN:
                raise ValueError("noise table must hold 512*128*128 floats")
            cfg.noise = keep.ctypes.data
        pairs = (Tuning * max(len(tuning), 1))(*[Tuning(int(o), int(v)) for o, v in tuning])
        self._chk(self._L.vxrt_create_tuned(C.byref(cfg), pairs, C.c_size_t(len(tuning)), C.byref(self._h)), "vxrt_create_tuned")
        self.uniforms = Uniforms.default()
        self.temporal_uniforms = TemporalUniforms.default()
        self.denoise_uniforms = DenoiseUniforms.default()

    def _chk(self, status, where):
        if status != 0:
            raise VxrtError(status, where, (self._L.vxrt_last_error() or b"").decode(errors="replace"))

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vxrt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene ---------------------------------------------------------------------------------------
    def recreate_octree(self, pos, mrgb):
        """Context::recreate_octree(voxels): voxels as (position [i16;3], [material, r, g, b])."""
        pos = np.ascontiguousarray(pos, np.int16)
        mrgb = np.ascontiguousarray(mrgb, np.uint8)
        if pos.shape != (len(pos), 3) or mrgb.shape != (len(pos), 4):
            raise ValueError("pos must be [n,3] int16 and mrgb [n,4] uint8")
        self._chk(self._L.vxrt_set_voxels(self._h, _p(pos), _p(mrgb), C.c_size_t(len(pos))), "vxrt_set_voxels")

    def set_menger(self, level, clip=0, mrgb=(0, 0xb0, 0xd0, 0x60), emissive_period=0):
        """Procedural Menger sponge built straight into the device scene format (BASELINE config 5)."""
        m = np.asarray(mrgb, np.uint8)
        self._chk(self._L.vxrt_set_menger(self._h, C.c_uint32(level), C.c_uint32(clip), _p(m), C.c_uint32(emissive_period)),
               "vxrt_set_menger")

    def read_scene(self):
        """Test hook (vxrt_debug_read_scene): the device's scene -> (svo uint32[n,2] = masks, base; leaves int32[k])."""
        ns, nl = C.c_size_t(0), C.c_size_t(0)
        self._chk(self._L.vxrt_debug_read_scene(self._h, None, C.c_size_t(0), C.byref(ns), None, C.c_size_t(0), C.byref(nl)), "vxrt_debug_read_scene")
        svo, leaves = np.zeros((ns.value, 2), np.uint32), np.zeros(nl.value, np.int32)
        self._chk(self._L.vxrt_debug_read_scene(self._h, _p(svo), C.c_size_t(len(svo)), C.byref(ns), _p(leaves), C.c_size_t(len(leaves)), C.byref(nl)),
               "vxrt_debug_read_scene")
        return svo, leaves

    def load_vox(self, path, flags=0):
        """vox::load + voxels_from_vox + recreate_octree (src/context.rs:1817-1821); flags: VOX_* for whole scenes."""
        if flags == 0:
            self._chk(self._L.vxrt_load_vox(self._h, os.fsencode(path)), "vxrt_load_vox")
        else:
            with open(path, "rb") as f:
                pos, mrgb, _ = vox_scene_to_voxels(f.read(), flags)
            self.recreate_octree(pos, mrgb)

    def set_noise(self, table):
        """Replaces the noise table (512*128*128 floats), e.g. with blue_noise() or load_blue_noise()."""
        t = np.ascontiguousarray(table, np.float32).reshape(-1)
        if t.size != NOISE_LEN:
            raise ValueError("noise table must hold 512*128*128 floats")
        self._chk(self._L.vxrt_set_noise(self._h, _p(t)), "vxrt_set_noise")

    def load_vox_bytes(self, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        self._chk(self._L.vxrt_load_vox_memory(self._h, _p(buf), C.c_size_t(len(data))), "vxrt_load_vox_memory")

    # -- frame ---------------------------------------------------------------------------------------
    def resize(self, width, height):
        self._chk(self._L.vxrt_resize(self._h, C.c_uint32(width), C.c_uint32(height)), "vxrt_resize")
        self.width, self.height = int(width), int(height)

    def update_bindings(self):
        """Push camera + parameter blocks (Context::update_bindings, src/context.rs:2136-2162).  A block that has not changed since it
        was last pushed is not pushed again (four calls through ctypes are ~10 us: 2 % of a rank's 20-frame block on 8 GPUs)."""
        cam = self.camera
        pos, dirn = np.asarray(cam.position, np.float32), np.asarray(cam.direction, np.float32)
        state = (pos.tobytes(), dirn.tobytes(), cam.fov, bytes(self.uniforms), bytes(self.temporal_uniforms), bytes(self.denoise_uniforms))
        last = getattr(self, "_pushed", None)
        if last is None or state[:3] != last[:3]:
            self._chk(self._L.vxrt_set_camera(self._h, _p(pos), _p(dirn), C.c_float(cam.fov)), "vxrt_set_camera")
        if last is None or state[3] != last[3]:
            self._chk(self._L.vxrt_set_scene_params(self._h, C.byref(self.uniforms)), "vxrt_set_scene_params")
        if last is None or state[4] != last[4]:
            self._chk(self._L.vxrt_set_temporal(self._h, C.byref(self.temporal_uniforms)), "vxrt_set_temporal")
        if last is None or state[5] != last[5]:
            self._chk(self._L.vxrt_set_denoise(self._h, C.byref(self.denoise_uniforms)), "vxrt_set_denoise")
        self._pushed = state

    def render(self, flags=ALL):
        """Context::render(): frame_number += 1, voxels -> temporal -> denoise, history hand-over."""
        self.update_bindings()
        self._chk(self._L.vxrt_render(self._h, C.c_uint32(flags)), "vxrt_render")

    def render_frames(self, flags, count):
        """`count` frames with the current camera and parameters in one call (vxrt_render_frames)."""
        self.update_bindings()
        self._chk(self._L.vxrt_render_frames(self._h, C.c_uint32(flags), C.c_uint32(count)), "vxrt_render_frames")

    def cast_rays(self, origins, dirs):
        """Test hook (vxrt_debug_cast_rays): the kernels' cast_bounded_ray for given rays -> (hit bool[n], time, node int32, normal[n,3])."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(o)
        hit, time, node, normal = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32)
        self._chk(self._L.vxrt_debug_cast_rays(self._h, _p(o), _p(d), C.c_size_t(n), _p(hit), _p(time), _p(node), _p(normal)), "vxrt_debug_cast_rays")
        return hit.astype(bool), time, node, normal

    def path_log(self, x, y):
        """Test hook (vxrt_debug_path_log): the casts of pixel (x, y) of the next frame -> float32[casts, 12]
        (origin, direction, hit, time, bits(leaf word), normal)."""
        self.update_bindings()
        log, n = np.zeros((32, 12), np.float32), C.c_int32(0)
        self._chk(self._L.vxrt_debug_path_log(self._h, C.c_int32(x), C.c_int32(y), _p(log), C.byref(n)), "vxrt_debug_path_log")
        return log[:n.value]

    def render_path(self, flags, positions, directions, fov=None):
        """Frames along a camera path (vxrt_render_path): frame k is seen from positions[k] towards directions[k]."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        if len(pos) != len(dirs):
            raise ValueError("one direction per position")
        self.update_bindings()
        fov = self.camera.fov if fov is None else float(np.float32(fov))
        self._chk(self._L.vxrt_render_path(self._h, C.c_uint32(flags), C.c_uint32(len(pos)), _p(pos), _p(dirs), C.c_float(fov)), "vxrt_render_path")
        if len(pos):
            self.camera = Camera(pos[-1], dirs[-1], fov)

    def render_spp(self, flags, spp):
        """One displayed frame of `spp` samples per pixel (vxrt_render_spp): spp trace frames averaged, then temporal / denoise."""
        self.update_bindings()
        self._chk(self._L.vxrt_render_spp(self._h, C.c_uint32(flags), C.c_uint32(spp)), "vxrt_render_spp")

    def render_stage(self, flags):
        """vxrt_render without re-pushing parameters (multi-GPU: DENOISE after the halo exchange)."""
        self._chk(self._L.vxrt_render(self._h, C.c_uint32(flags)), "vxrt_render")

    def culled_pixels(self):
        """vxrt_debug_culled_pixels: primary rays of the next frame (camera as set) that the sky cull decides without a walk."""
        n = C.c_uint64(0)
        self.update_bindings()
        self._chk(self._L.vxrt_debug_culled_pixels(self._h, C.byref(n)), "vxrt_debug_culled_pixels")
        return int(n.value)

    def set_option(self, option, value):
        """vxrt_set_option: OPT_DENOISE_MODE (0 exact, 1 tolerant), OPT_TAIL_CAPACITY (records per queue shard; 0 = automatic)."""
        self._chk(self._L.vxrt_set_option(self._h, C.c_int(option), C.c_uint32(value)), "vxrt_set_option")

    def sync(self):
        self._chk(self._L.vxrt_sync(self._h), "vxrt_sync")

    def reset_history(self):
        self._chk(self._L.vxrt_reset_history(self._h), "vxrt_reset_history")

    def set_frame_number(self, n):
        self._chk(self._L.vxrt_set_frame_number(self._h, C.c_uint32(n)), "vxrt_set_frame_number")

    # -- outputs -------------------------------------------------------------------------------------
    def local_rows(self):
        n = C.c_uint32(0)
        self._chk(self._L.vxrt_local_rows(self._h, C.byref(n), None), "vxrt_local_rows")
        rows = np.zeros(n.value, np.uint32)
        if n.value:
            self._chk(self._L.vxrt_local_rows(self._h, C.byref(n), _p(rows)), "vxrt_local_rows")
        return rows

    def read(self, which):
        """-> float32[local_rows, width, 4] (whole frame for a single-GPU context)."""
        n = C.c_uint32(0)
        self._chk(self._L.vxrt_local_rows(self._h, C.byref(n), None), "vxrt_local_rows")
        out = np.zeros((n.value, self.width, 4), np.float32)
        self._chk(self._L.vxrt_read(self._h, C.c_int(which), _p(out), C.c_size_t(out.nbytes)), "vxrt_read")
        return out

    def read_into(self, which, arr):
        """vxrt_read into a caller's float32 array of the image's size (no allocation per call)."""
        self._chk(self._L.vxrt_read(self._h, C.c_int(which), _p(arr), C.c_size_t(arr.nbytes)), "vxrt_read")

    def read_async(self, which, dst, slot=0):
        """vxrt_read_async: snapshot image `which` as the stages enqueued so far leave it and start its transfer into `dst` (a
        PinnedImage, or any C-contiguous float32 array of the image's size) without waiting; read_wait(slot) waits for it."""
        arr = dst.array if isinstance(dst, PinnedImage) else dst
        self._chk(self._L.vxrt_read_async(self._h, C.c_int(which), _p(arr), C.c_size_t(arr.nbytes), C.c_uint32(slot)), "vxrt_read_async")

    def read_wait(self, slot=0):
        self._chk(self._L.vxrt_read_wait(self._h, C.c_uint32(slot)), "vxrt_read_wait")

    def pinned_image(self):
        """A pinned host buffer of this context's image size (vxrt_host_alloc) for read_async."""
        n = C.c_uint32(0)
        self._chk(self._L.vxrt_local_rows(self._h, C.byref(n), None), "vxrt_local_rows")
        return PinnedImage(self._L, (n.value, self.width, 4))

    def touch_map(self, enable=True):
        """vxrt_debug_touch_map (-DVXRT_VARIANTS=1 library): mark every 64-byte line of the scene that the frames from now on read."""
        self._chk(self._L.vxrt_debug_touch_map(self._h, C.c_uint32(1 if enable else 0)), "vxrt_debug_touch_map")

    def touch_count(self, reset=True):
        """vxrt_debug_touch_count -> dict: lines / bytes of node records and leaf words touched since the map was cleared."""
        out = (C.c_uint64 * 6)()
        self._chk(self._L.vxrt_debug_touch_count(self._h, out, C.c_uint32(1 if reset else 0)), "vxrt_debug_touch_count")
        n64, l64, n128, l128, nt, lt = (int(v) for v in out)
        return {"node_lines_64": n64, "leaf_lines_64": l64, "node_lines_128": n128, "leaf_lines_128": l128, "node_lines_total": nt, "leaf_lines_total": lt,
                "unique_bytes_64": 64 * (n64 + l64), "unique_bytes_128": 128 * (n128 + l128), "scene_bytes": 64 * (nt + lt)}

    def device_image(self, which):
        ptr = C.c_void_p()
        nbytes = C.c_size_t(0)
        self._chk(self._L.vxrt_device_image(self._h, C.c_int(which), C.byref(ptr), C.byref(nbytes)), "vxrt_device_image")
        return ptr.value, nbytes.value

    def stats(self):
        s = Stats()
        self._chk(self._L.vxrt_get_stats(self._h, C.byref(s)), "vxrt_get_stats")
        return s

    def reset_stats(self):
        self._chk(self._L.vxrt_reset_stats(self._h), "vxrt_reset_stats")

    # -- multi-GPU halo (include/vxrt.h "halo") --------------------------------------------------------
    def _push_denoise(self):
        # the halo's row count follows the denoise radius: the library must know the radius the caller has set
        self._chk(self._L.vxrt_set_denoise(self._h, C.byref(self.denoise_uniforms)), "vxrt_set_denoise")

    def halo_info(self):
        self._push_denoise()
        info = HaloInfo()
        self._chk(self._L.vxrt_halo_info_get(self._h, C.byref(info)), "vxrt_halo_info_get")
        return info

    def halo_bytes(self):
        self._push_denoise()
        n = C.c_size_t(0)
        self._chk(self._L.vxrt_halo_bytes(self._h, C.byref(n)), "vxrt_halo_bytes")
        return n.value

    def halo_pack(self, dev_to_prev, dev_to_next):
        """One pack launch on the context's stream; returns at once (order the communication with stream_wait_context)."""
        self._chk(self._L.vxrt_halo_pack(self._h, C.c_void_p(dev_to_prev), C.c_void_p(dev_to_next)), "vxrt_halo_pack")

    def halo_unpack(self, dev_from_prev, dev_from_next):
        """One unpack launch on the context's stream; returns at once."""
        self._chk(self._L.vxrt_halo_unpack(self._h, C.c_void_p(dev_from_prev), C.c_void_p(dev_from_next)), "vxrt_halo_unpack")

    def stream_wait_context(self, stream):
        """`stream` (a raw hipStream_t, e.g. torch.cuda.Stream.cuda_stream; 0 = the default stream) waits for what the context has enqueued."""
        self._chk(self._L.vxrt_stream_wait_context(self._h, C.c_void_p(stream)), "vxrt_stream_wait_context")

    def context_wait_stream(self, stream):
        """The context's stream waits for what `stream` has enqueued so far."""
        self._chk(self._L.vxrt_context_wait_stream(self._h, C.c_void_p(stream)), "vxrt_context_wait_stream")

    def halo_export(self, dev_to_prev, dev_to_next):
        """Synchronous halo_pack."""
        self._chk(self._L.vxrt_halo_export(self._h, C.c_void_p(dev_to_prev), C.c_void_p(dev_to_next)), "vxrt_halo_export")

    def halo_import(self, dev_from_prev, dev_from_next):
        """Synchronous halo_unpack."""
        self._chk(self._L.vxrt_halo_import(self._h, C.c_void_p(dev_from_prev), C.c_void_p(dev_from_next)),
               "vxrt_halo_import")
