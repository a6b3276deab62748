"""Builds libvxrt.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvxrt.so")
# the default library: tracers 1 (all-in-one kernel) and 4 (head + compacted tail) over the 8-byte scene records
SOURCES = ["api_context.hip", "api_scene.hip", "api_trace.hip", "api_frame.hip", "api_halo.hip", "api_host.cpp", "api_debug.hip",
           "trace.hip", "trace_tail.hip", "post.hip", "halo.hip", "noise.hip", "scene_device.hip", "scene_host.cpp", "scene_procedural.cpp",
           "noise_zip.cpp", "vox_scene.cpp"]
# -DVXRT_VARIANTS=1 (scripts/test_variants.sh): the schedules and the scene format that measured slower and are kept for comparison —
# tracers 2 (wavefront), 3 (ray queues), 5 (per-lane path refill) and the wide records (two tree levels per 16-byte record)
VARIANT_SOURCES = ["trace_wavefront.hip", "trace_paths.hip", "trace_pool.hip", "trace_dda.hip", "trace_fused.hip"]
HEADERS = ["ctx.h", "halo_view.h", "kernels.h", "trace_common.h", "trace_block.h", "trace_tail_body.h", "ray_queue.h", "walk_wide.h", "scene_host.h", "vx_vec.h", os.path.join("..", "..", "include", "vxrt.h"),
           os.path.join("..", "..", "include", "vxrt_host.h"), os.path.join("..", "..", "include", "vxrt_debug.h"),
           os.path.join("..", "..", "include", "vxrt_detmath.h"),
           os.path.join("..", "..", "include", "vxrt_bluenoise.h")]

# -ffp-contract=off / no fast-math / IEEE divide+sqrt / denormals kept: include/vxrt_detmath.h
# -fno-slp-vectorize: the SLP vectoriser pairs the x/y components of the 3-vector arithmetic into v_pk_add_f32 / v_pk_mul_f32;
# on gfx950 those cost more issue time than the two scalar-per-lane instructions they replace (measured: 21.8 -> 23.3 Gray/s)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-Wall", "-Wextra",
         "-Wno-unused-parameter"]


VARIANTS_LIB = os.path.join(HERE, "libvxrt_variants.so")   # the -DVXRT_VARIANTS=1 build (never the product; VXRT_LIB points tests at it)
OBJ_DIR = os.path.join(HERE, "_obj")                         # per-source objects, keyed by content (not tracked, does not travel)


def _stamp(lib):
    """What a library was built from (travels with it to the GPU box; modification times do not survive the trip)."""
    return lib + ".srchash"


STAMP = _stamp(LIB)


def _sources(variants):
    return SOURCES + (VARIANT_SOURCES if variants else [])


def _flags(extra_flags, variants):
    return FLAGS + (["-DVXRT_VARIANTS=1"] if variants else []) + list(extra_flags)


def _headers_digest():
    import hashlib
    h = hashlib.sha256()
    for name in sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def source_hash(extra_flags=(), variants=False):
    """sha256 over the sources, headers and flags that make the library."""
    import hashlib
    h = hashlib.sha256()
    h.update(_headers_digest().encode())
    for name in sorted(_sources(variants)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(_flags(extra_flags, variants)).encode())
    return h.hexdigest()


def needs_build(extra_flags=(), variants=False, out=None):
    """True when the library is missing or was built from other sources / flags than the tree holds now (by content)."""
    lib = out or (VARIANTS_LIB if variants else LIB)
    if not os.path.exists(lib) or not os.path.exists(_stamp(lib)):
        return True
    try:
        return open(_stamp(lib)).read().strip() != source_hash(extra_flags, variants)
    except OSError:
        return True


def _compile_one(job):
    import hashlib
    hipcc, flags, name, hdr = job
    src = os.path.join(CSRC, name)
    with open(src, "rb") as f:
        key = hashlib.sha256(hdr.encode() + b"\0" + name.encode() + b"\0" + f.read() + b"\0" + " ".join(flags).encode()).hexdigest()[:24]
    obj = os.path.join(OBJ_DIR, f"{os.path.splitext(name)[0]}.{key}.o")
    if not os.path.exists(obj):
        tmp = f"{obj}.{os.getpid()}.tmp"
        subprocess.check_call([hipcc] + [f for f in flags if f != "-shared"] + ["-x", "hip", "-c", src, "-o", tmp])
        os.replace(tmp, obj)
    return obj


def build(force=False, verbose=False, extra_flags=(), variants=False, out=None):
    """Compiles every source to an object (in parallel; objects are kept by content hash, so only what changed is recompiled) and
    links the library.  Ranks that start together (torchrun) serialise on a lock file and re-check after taking it; the library
    is written under a per-process name and renamed into place."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    extra_flags = list(extra_flags) + os.environ.get("VXRT_HIPCC_FLAGS", "").split()
    lib = out or (VARIANTS_LIB if variants else LIB)     # out: an A/B build under another name (scripts/ab_build.sh; VXRT_LIB selects it)
    if not force and not needs_build(extra_flags, variants, out):
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build(extra_flags, variants, out):   # another process built it while this one waited
            return lib
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = _flags(extra_flags, variants)
        hdr = _headers_digest()
        jobs = [(hipcc, flags, name, hdr) for name in _sources(variants)]
        if verbose:
            print(f"{hipcc} {' '.join(flags)} -c <{len(jobs)} sources> ; link -> {lib}")
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            objs = list(pool.map(_compile_one, jobs))
        keep = set(objs)
        for f in os.listdir(OBJ_DIR):   # objects of older source versions
            path = os.path.join(OBJ_DIR, f)
            if f.endswith(".o") and path not in keep and (time.time() - os.path.getmtime(path)) > 6 * 3600:
                os.remove(path)
        tmp = f"{lib}.{os.getpid()}.tmp"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp, "-lz"])
        os.replace(tmp, lib)      # other processes keep the library they mapped; new ones see a whole file
        with open(_stamp(lib), "w") as f:
            f.write(source_hash(extra_flags, variants) + "\n")
    return lib


TOOL = os.path.join(HERE, "vxrt_render")
TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "vxrt_render.cpp")


def build_tool(force=False):
    """The C++ headless driver (tools/vxrt_render.cpp over include/vxrt.hpp), linked against libvxrt.so."""
    build()
    if not force and os.path.exists(TOOL) and os.path.getmtime(TOOL) >= max(os.path.getmtime(TOOL_SRC), os.path.getmtime(LIB)):
        return TOOL
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", TOOL_SRC, "-o", TOOL, "-L" + HERE, "-lvxrt",
                           "-Wl,-rpath,$ORIGIN"])
    return TOOL


MULTI_TOOL = os.path.join(HERE, "vxrt_multi")
MULTI_TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "vxrt_multi.cpp")


def build_multi_tool(force=False):
    """The multi-GPU C++ host (tools/vxrt_multi.cpp): one thread per rank over include/vxrt.hpp, the halo over RCCL.  Host code only:
    g++ against the HIP runtime API and librccl of /opt/rocm (-D__HIP_PLATFORM_AMD__ is what hip_runtime_api.h asks a host
    compiler for)."""
    build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hdr = os.path.join(os.path.dirname(HERE), "include", "vxrt.hpp")
    if not force and os.path.exists(MULTI_TOOL) and os.path.getmtime(MULTI_TOOL) >= max(os.path.getmtime(MULTI_TOOL_SRC), os.path.getmtime(LIB),
                                                                                       os.path.getmtime(hdr)):
        return MULTI_TOOL
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           MULTI_TOOL_SRC, "-o", MULTI_TOOL, "-L" + HERE, "-lvxrt", "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
                           "-pthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return MULTI_TOOL


if __name__ == "__main__":
    import sys
    build(force="-f" in sys.argv, verbose=True)
    build_tool(force="-f" in sys.argv)
    build_multi_tool(force="-f" in sys.argv)
