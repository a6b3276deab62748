"""ORACLE — test infrastructure only.

ctypes front end of oracle/_build/liboracle.so, the CPU restatement of the reference's hot path
(see oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  Parity status: the shaders' restatement is pinned by the reference's
compiled modules executed by ospirv.cpp (spirv_dispatch below; tests/test_oracle_spirv_exec.py), up to driver-defined operations;
the host-side functions are unpinned by any reference output.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

NOISE_LEN = 512 * 128 * 128
NOISE_SEED = 0x5EED0001


def build(force=False):
    """Compile the oracle with g++ (make).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


class Uniforms(C.Structure):
    """Uniforms, src/context.rs:425-469 (148 bytes, std140 offsets of shaders/voxels.comp:28-49)."""
    _fields_ = [
        ("camera_origin", C.c_float * 4), ("camera_right", C.c_float * 4), ("camera_up", C.c_float * 4),
        ("camera_forward", C.c_float * 4), ("light", C.c_float * 4), ("global_time", C.c_float),
        ("still_sample", C.c_uint32), ("frame_number", C.c_uint32), ("emit_strength", C.c_float),
        ("sun_strength", C.c_float), ("sun_size", C.c_float), ("sun_yaw", C.c_float), ("sun_pitch", C.c_float),
        ("sun_color", C.c_float * 4), ("sky_color", C.c_float * 4), ("specularity", C.c_float),
    ]

    @classmethod
    def default(cls):
        """Uniforms::default(), src/context.rs:471-498."""
        u = cls()
        u.emit_strength, u.sun_strength, u.sun_size = 4.0, 4.0, 0.05
        u.sun_yaw, u.sun_pitch = 1.32, 1.0
        u.sun_color[:] = [1.0, 1.0, 1.0, 0.0]
        u.sky_color[:] = [0.45, 0.6, 0.65, 0.0]
        u.specularity = 0.0
        return u

    def set_camera(self, position, basis9):
        self.camera_origin[:] = [float(v) for v in position] + [0.0]
        self.camera_right[:] = [float(v) for v in basis9[0:3]] + [0.0]
        self.camera_up[:] = [float(v) for v in basis9[3:6]] + [0.0]
        self.camera_forward[:] = [float(v) for v in basis9[6:9]] + [0.0]

    def camera16(self):
        return np.array(list(self.camera_origin) + list(self.camera_right) + list(self.camera_up)
                        + list(self.camera_forward), dtype=np.float32)


assert C.sizeof(Uniforms) == 148


class Temporal(C.Structure):
    """TemporalUniforms, src/context.rs:502-525."""
    _fields_ = [("sample_blending", C.c_float), ("maximum_blending", C.c_float),
                ("blending_distance_cutoff", C.c_float)]

    @classmethod
    def default(cls):
        return cls(0.5, 0.98, 1e-2)


class Denoise(C.Structure):
    """DenoiseUniforms, src/context.rs:304-325."""
    _fields_ = [("radius", C.c_uint32), ("sigma_distance", C.c_float), ("sigma_range", C.c_float),
                ("albedo_factor", C.c_float)]

    @classmethod
    def default(cls):
        return cls(0, 2.0, 1.5, 1.0)


def _declare(L):
    L.orc_voxels_from_vox.restype = C.c_long
    L.orc_create_octree.restype = C.c_long
    L.orc_trace.restype = C.c_longlong
    L.orc_trace_menger.restype = C.c_longlong
    L.orc_menger_lazy_nodes.restype = C.c_longlong
    L.orc_default_scene.restype = C.c_long
    L.orc_parse_raw_f32img.restype = C.c_long
    return L


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


_ALT_PATH = os.path.join(_HERE, "_build", "liboracle_alt.so")
_alt = None
ALT_LIBM, ALT_NORMALIZE, ALT_SAMPLER, ALT_INVERSE = 1, 2, 4, 8     # oracle/ovec.h: which driver-defined choice is made the other way
ALT_ALL = 15


class alt_builtins:
    """with oracle.alt_builtins(mask): every call of this module goes to liboracle_alt.so (`make -C oracle alt`), the same restatement
    with the choices GLSL / Vulkan leave to the driver (U4, U5, U6) made the other way where `mask` says so (oracle/ovec.h).  Only
    tests/test_oracle_builtin_sensitivity.py uses it: how far can an image move between two conforming implementations?"""

    def __init__(self, mask):
        self.mask = int(mask)

    def __enter__(self):
        global _lib, _alt
        if _alt is None:
            subprocess.check_call(["make", "-s", "-C", _HERE, "alt"])
            _alt = _declare(C.CDLL(_ALT_PATH))
        self._saved = lib()
        _alt.orc_set_alt(C.c_int(self.mask))
        _lib = _alt
        return self

    def __exit__(self, *exc):
        global _lib
        _alt.orc_set_alt(C.c_int(0))
        _lib = self._saved


_FMA_PATH = os.path.join(_HERE, "_build", "liboracle_fma.so")
_fma = None


class contracted:
    """with oracle.contracted(): every call goes to liboracle_fma.so (`make -C oracle fma`): the same restatement compiled with
    -ffp-contract=fast, i.e. a * b + c fused wherever the compiler can — the freedom a Vulkan driver has with the reference's modules,
    which carry no NoContraction decoration (U9).  Only tests/test_oracle_builtin_sensitivity.py uses it."""

    def __enter__(self):
        global _lib, _fma
        if _fma is None:
            subprocess.check_call(["make", "-s", "-C", _HERE, "fma"])
            _fma = _declare(C.CDLL(_FMA_PATH))
        self._saved = lib()
        _lib = _fma
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def voxels_from_vox(data: bytes):
    """vox::parse + Context::voxels_from_vox -> (pos int16[n,3], mrgb uint8[n,4], size(x,y,z))."""
    buf = np.frombuffer(data, dtype=np.uint8)
    size = np.zeros(3, np.uint32)
    n = lib().orc_voxels_from_vox(_p(buf), C.c_size_t(len(data)), None, None, C.c_size_t(0), _p(size))
    if n < 0:
        raise OracleError(n)
    pos = np.zeros((n, 3), np.int16)
    mrgb = np.zeros((n, 4), np.uint8)
    lib().orc_voxels_from_vox(_p(buf), C.c_size_t(len(data)), _p(pos), _p(mrgb), C.c_size_t(n), _p(size))
    return pos, mrgb, tuple(int(s) for s in size)


def default_scene(seed):
    """Context::create_voxels (src/context.rs:838-910) with the seeded draws documented in ovox.cpp."""
    n = lib().orc_default_scene(C.c_uint32(seed), None, None, C.c_size_t(0))
    pos = np.zeros((n, 3), np.int16)
    mrgb = np.zeros((n, 4), np.uint8)
    lib().orc_default_scene(C.c_uint32(seed), _p(pos), _p(mrgb), C.c_size_t(n))
    return pos, mrgb


def blue_noise_layer(seed, layer, size=128):
    """One layer of the void-and-cluster table specified in include/vxrt_bluenoise.h -> float32[size, size]."""
    out = np.zeros((size, size), np.float32)
    rc = lib().orc_blue_noise_layer(C.c_uint32(seed), C.c_uint32(layer), C.c_int(size), _p(out))
    if rc != 0:
        raise OracleError(rc)
    return out


def parse_raw_f32img(data: bytes):
    """parse_raw_f32img (src/context.rs:1087-1116): BE u32 w, BE u32 h, w*h BE f32 -> float32[h, w]."""
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_uint32(0), C.c_uint32(0)
    out = np.zeros(max(len(data) // 4, 1), np.float32)
    n = lib().orc_parse_raw_f32img(_p(buf), C.c_size_t(len(data)), _p(out), C.c_size_t(len(out)), C.byref(w), C.byref(h))
    if n < 0:
        raise OracleError(n)
    return out[:n].reshape(h.value, w.value)


def default_palette():
    out = np.zeros(256, np.uint32)
    lib().orc_default_palette(_p(out))
    return out


def voxel_depth(pos):
    pos = np.ascontiguousarray(pos, np.int16)
    return lib().orc_voxel_depth(_p(pos), C.c_size_t(len(pos)))


def create_octree(pos, mrgb):
    """Context::create_octree -> int32 buffer (5-word header + 8 words per node)."""
    pos = np.ascontiguousarray(pos, np.int16)
    mrgb = np.ascontiguousarray(mrgb, np.uint8)
    n = lib().orc_create_octree(_p(pos), _p(mrgb), C.c_size_t(len(pos)), None, C.c_size_t(0))
    if n < 0:
        raise OracleError(n)
    out = np.zeros(n, np.int32)
    lib().orc_create_octree(_p(pos), _p(mrgb), C.c_size_t(len(pos)), _p(out), C.c_size_t(n))
    return out


def camera_axis_scaled(position, direction, fov, width, height):
    """Camera::axis_scaled -> float32[9] = right, up, forward_ray."""
    out = np.zeros(9, np.float32)
    pos = np.asarray(position, np.float32)
    d = np.asarray(direction, np.float32)
    lib().orc_camera_axis_scaled(_p(pos), _p(d), C.c_float(fov), C.c_uint32(width), C.c_uint32(height), _p(out))
    return out


def noise_table(seed=NOISE_SEED, n=NOISE_LEN):
    out = np.zeros(n, np.float32)
    lib().orc_noise_table(C.c_uint32(seed), _p(out), C.c_size_t(n))
    return out


def trace(octree, noise, uniforms, width, height, max_bounces=3, crop=None, nthreads=None):
    """voxels.comp over crop=(x0,y0,x1,y1) of a width x height frame.
    Returns (color, normal_depth, albedo) float32[h,w,4] and the ray count."""
    del width, height  # pixel coordinates are frame-absolute; the frame size only enters via the camera basis
    x0, y0, x1, y1 = crop
    h, w = y1 - y0, x1 - x0
    color = np.zeros((h, w, 4), np.float32)
    nd = np.zeros((h, w, 4), np.float32)
    alb = np.zeros((h, w, 4), np.float32)
    nthreads = nthreads or os.cpu_count() or 1
    rays = lib().orc_trace(_p(octree), _p(noise), C.byref(uniforms), C.c_int(max_bounces), C.c_int(x0), C.c_int(y0),
                           C.c_int(x1), C.c_int(y1), _p(color), _p(nd), _p(alb), C.c_int(nthreads))
    return color, nd, alb, int(rays)


def _menger_args(level, clip, mrgb, emissive_period):
    return C.c_uint32(level), C.c_uint32(clip), _p(np.ascontiguousarray(mrgb, np.uint8)), C.c_uint32(emissive_period)


def trace_menger(level, clip, mrgb, emissive_period, noise, uniforms, max_bounces, crop, nthreads=None):
    """trace() over the implicit octree of the procedural Menger scene (oprocedural.cpp; BASELINE config 5 at any size)."""
    x0, y0, x1, y1 = crop
    h, w = y1 - y0, x1 - x0
    color = np.zeros((h, w, 4), np.float32)
    nd = np.zeros((h, w, 4), np.float32)
    alb = np.zeros((h, w, 4), np.float32)
    nthreads = nthreads or min(os.cpu_count() or 1, 32)
    rays = lib().orc_trace_menger(*_menger_args(level, clip, mrgb, emissive_period), _p(noise), C.byref(uniforms), C.c_int(max_bounces),
                                  C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), _p(color), _p(nd), _p(alb), C.c_int(nthreads))
    return color, nd, alb, int(rays)


def cast_rays_menger(level, clip, mrgb, emissive_period, origins, dirs, max_distance=float(1 << 30), nthreads=None):
    origins = np.ascontiguousarray(origins, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n = len(origins)
    hit = np.zeros(n, np.uint8)
    time = np.zeros(n, np.float32)
    node = np.zeros(n, np.int32)
    normal = np.zeros((n, 3), np.float32)
    iters = np.zeros(n, np.int32)
    nthreads = nthreads or min(os.cpu_count() or 1, 32)
    lib().orc_cast_rays_menger(*_menger_args(level, clip, mrgb, emissive_period), _p(origins), _p(dirs), C.c_size_t(n),
                               C.c_float(max_distance), _p(hit), _p(time), _p(node), _p(normal), _p(iters), C.c_int(nthreads))
    return hit.astype(bool), time, node, normal, iters


def dda_menger(level, clip, origins, dirs, nthreads=None):
    """Independent binary64 DDA over the Menger voxel PREDICATE (no octree, no grid) -> hit, t, entry axis, cell."""
    origins = np.ascontiguousarray(origins, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n = len(origins)
    hit = np.zeros(n, np.uint8)
    time = np.zeros(n, np.float64)
    axis = np.zeros(n, np.int32)
    cell = np.zeros((n, 3), np.int32)
    nthreads = nthreads or os.cpu_count() or 1
    lib().orc_dda_menger(C.c_uint32(level), C.c_uint32(clip), _p(origins), _p(dirs), C.c_size_t(n), _p(hit), _p(time), _p(axis),
                         _p(cell), C.c_int(nthreads))
    return hit.astype(bool), time, axis, cell


def menger_cells(level, clip, mrgb, emissive_period, cells):
    """The procedural scene's voxel predicate: (solid bool[n], leaf word int32[n]) for integer cells int32[n,3]."""
    cells = np.ascontiguousarray(cells, np.int32)
    solid = np.zeros(len(cells), np.uint8)
    word = np.zeros(len(cells), np.int32)
    lib().orc_menger_cells(*_menger_args(level, clip, mrgb, emissive_period), _p(cells), C.c_size_t(len(cells)), _p(solid), _p(word))
    return solid.astype(bool), word


def menger_depth(level, clip):
    return int(lib().orc_menger_depth(C.c_uint32(level), C.c_uint32(clip)))


def menger_lazy_nodes(level, clip, mrgb, emissive_period):
    return int(lib().orc_menger_lazy_nodes(*_menger_args(level, clip, mrgb, emissive_period)))


def path_log(octree, noise, uniforms, max_bounces, x, y):
    """The casts of one pixel's path (orc_trace_pixel_log) -> float32[casts, 12] = origin, direction, hit, time, bits(leaf word), normal."""
    log = np.zeros((32, 12), np.float32)
    n = lib().orc_trace_pixel_log(_p(octree), _p(noise), C.byref(uniforms), C.c_int(max_bounces), C.c_int(x), C.c_int(y), _p(log))
    return log[:n]


def cast_rays(octree, origins, dirs, max_distance=float(1 << 30)):
    origins = np.ascontiguousarray(origins, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n = len(origins)
    hit = np.zeros(n, np.uint8)
    time = np.zeros(n, np.float32)
    node = np.zeros(n, np.int32)
    normal = np.zeros((n, 3), np.float32)
    iters = np.zeros(n, np.int32)
    lib().orc_cast_rays(_p(octree), _p(origins), _p(dirs), C.c_size_t(n), C.c_float(max_distance), _p(hit), _p(time),
                        _p(node), _p(normal), _p(iters))
    return hit.astype(bool), time, node, normal, iters


def temporal(sampled_color, new_nd, old_color, old_nd, cam16, old_cam16, tu, has_history, nthreads=None):
    h, w = sampled_color.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    nthreads = nthreads or os.cpu_count() or 1
    args = [np.ascontiguousarray(a, np.float32) for a in (sampled_color, new_nd, old_color, old_nd, cam16, old_cam16)]
    lib().orc_temporal(*[_p(a) for a in args], C.byref(tu), C.c_int(int(has_history)), C.c_int(w), C.c_int(h),
                       _p(out), C.c_int(nthreads))
    return out


def denoise(colors, normals_depths, albedo, cam16, du, nthreads=None):
    h, w = colors.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    nthreads = nthreads or os.cpu_count() or 1
    args = [np.ascontiguousarray(a, np.float32) for a in (colors, normals_depths, albedo, cam16)]
    lib().orc_denoise(*[_p(a) for a in args], C.byref(du), C.c_int(w), C.c_int(h), _p(out), C.c_int(nthreads))
    return out


def dda_cast(grid, base, origins, dirs):
    """Independent dense-grid DDA (odda.cpp). grid uint8[nx,ny,nz], base int[3] (integer cell of grid[0,0,0])."""
    grid = np.ascontiguousarray(grid, np.uint8)
    dims = np.array(grid.shape, np.int32)
    base = np.asarray(base, np.int32)
    origins = np.ascontiguousarray(origins, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n = len(origins)
    hit = np.zeros(n, np.uint8)
    time = np.zeros(n, np.float64)
    axis = np.zeros(n, np.int32)
    cell = np.zeros((n, 3), np.int32)
    lib().orc_dda_cast(_p(grid), _p(dims), _p(base), _p(origins), _p(dirs), C.c_size_t(n), _p(hit), _p(time), _p(axis),
                       _p(cell))
    return hit.astype(bool), time, axis, cell


def cpu_rs_render(coords_u16, rgb_u8, cam_pos, basis9, width, height, time=0.0):
    """CpuBackend::render (src/cpu.rs:32-72) -> (pixels u8[h,w,3], hit_time, hit_normal, hit_value)."""
    coords = np.ascontiguousarray(coords_u16, np.uint16)
    rgb = np.ascontiguousarray(rgb_u8, np.uint8)
    pixels = np.zeros((height, width, 3), np.uint8)
    ht = np.zeros((height, width), np.float32)
    hn = np.zeros((height, width, 3), np.float32)
    hv = np.zeros((height, width), np.int32)
    pos = np.asarray(cam_pos, np.float32)
    b9 = np.asarray(basis9, np.float32)
    lib().orc_cpu_rs_render(_p(coords), _p(rgb), C.c_size_t(len(coords)), _p(pos), _p(b9), C.c_int(width),
                            C.c_int(height), C.c_float(time), _p(pixels), _p(ht), _p(hn), _p(hv))
    return pixels, ht, hn, hv


class CpuRsBackend:
    """CpuBackend (src/cpu.rs:9-72): from_voxels once, render per frame over `nthreads` host threads (the reference: rayon)."""

    def __init__(self, coords_u16, rgb_u8):
        coords = np.ascontiguousarray(coords_u16, np.uint16)
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        lib().orc_cpu_rs_create.restype = C.c_void_p
        self._h = C.c_void_p(lib().orc_cpu_rs_create(_p(coords), _p(rgb), C.c_size_t(len(coords))))

    def render(self, cam_pos, basis9, width, height, time=0.0, nthreads=None):
        pixels = np.zeros((height, width, 3), np.uint8)
        pos = np.asarray(cam_pos, np.float32)
        b9 = np.asarray(basis9, np.float32)
        nthreads = nthreads or os.cpu_count() or 1
        lib().orc_cpu_rs_render_frame(self._h, _p(pos), _p(b9), C.c_int(width), C.c_int(height), C.c_float(time), _p(pixels),
                                      None, None, None, C.c_int(nthreads))
        return pixels

    def close(self):
        if self._h:
            lib().orc_cpu_rs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def detmath(fn, x, y=None):
    names = {"sin": 0, "cos": 1, "exp": 2, "log": 3, "pow": 4, "sqrt": 5, "div": 6, "tan": 7, "hemi_y": 9, "hemi_z": 10, "mul": 11, "sub": 12, "flip": 13, "min": 14, "max": 15, "max0": 16, "sign": 17, "clamp": 18, "min0": 19, "exp_unfused": 20}
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    lib().orc_detmath(C.c_int(names[fn]), _p(x), _p(y), _p(out), C.c_size_t(x.size))
    return out


# ---- the SPIR-V interpreter (ospirv.cpp): the reference's COMPILED shaders, executed instruction by instruction -------------------
class SpvBinding(C.Structure):
    _fields_ = [("set", C.c_uint32), ("binding", C.c_uint32), ("kind", C.c_uint32), ("pad", C.c_uint32), ("data", C.c_void_p),
                ("bytes", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("ox", C.c_uint32), ("oy", C.c_uint32),
                ("cw", C.c_uint32), ("ch", C.c_uint32)]


SPV_POISON = 1     # fill every Function variable with a NaN pattern at each function entry (does an output depend on an undefined read?)


class SpirvError(RuntimeError):
    pass


def spirv_buffer(binding, array):
    """A uniform / storage buffer binding over `array` (kept alive by the caller)."""
    a = np.ascontiguousarray(array)
    return (binding, 0, a, None)


def spirv_image(binding, array, size=None, origin=(0, 0), sampled=False):
    """A storage (or sampled) rgba32f image: `array` float32[h, w, 4] holds the window at `origin` of an image of `size` = (width, height)
    (default: the array is the whole image)."""
    a = np.ascontiguousarray(array, np.float32)
    assert a.ndim == 3 and a.shape[2] == 4
    return (binding, 2 if sampled else 1, a, (size or (a.shape[1], a.shape[0]), origin))


def spirv_sampler(binding):
    return (binding, 3, None, None)


def spirv_dispatch(module_bytes, bindings, x0, y0, x1, y1, flags=0, nthreads=None):
    """Runs the GLCompute entry point of a SPIR-V module for the invocation ids [x0, x1) x [y0, y1).  `bindings`: what spirv_buffer /
    spirv_image / spirv_sampler return (descriptor set 0); images are written in place.  Returns the instructions interpreted."""
    if len(module_bytes) % 4 or len(module_bytes) < 20:
        raise SpirvError("not a SPIR-V module (not a whole number of words)")
    words = np.frombuffer(module_bytes, dtype="<u4")
    arr = (SpvBinding * len(bindings))()
    keep = []
    for k, (binding, kind, a, img) in enumerate(bindings):
        b = arr[k]
        b.set, b.binding, b.kind = 0, binding, kind
        if a is not None:
            if not a.flags.writeable and kind == 1:
                raise ValueError("a storage image must be writeable")
            keep.append(a)
            b.data, b.bytes = a.ctypes.data, a.nbytes
        if img is not None:
            (b.width, b.height), (b.ox, b.oy) = img
            b.ch, b.cw = a.shape[0], a.shape[1]
    L = lib()
    L.orc_spirv_error.restype = C.c_char_p
    n = C.c_uint64(0)
    rc = L.orc_spirv_dispatch(_p(words), C.c_size_t(len(words)), arr, C.c_int(len(bindings)), C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(x1),
                              C.c_uint32(y1), C.c_uint32(flags), C.c_int(nthreads or os.cpu_count() or 1), C.byref(n))
    if rc != 0:
        raise SpirvError(L.orc_spirv_error().decode())
    return n.value
