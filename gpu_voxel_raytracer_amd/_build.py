"""Builds libvxrt.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvxrt.so")
SOURCES = ["vxrt_api.hip", "trace.hip", "trace_wavefront.hip", "trace_tail.hip", "trace_paths.hip", "post.hip", "noise.hip", "scene_device.hip", "scene_host.cpp", "scene_procedural.cpp",
           "noise_zip.cpp", "vox_scene.cpp"]
HEADERS = ["kernels.h", "trace_common.h", "walk_wide.h", "scene_host.h", "vx_vec.h", os.path.join("..", "..", "include", "vxrt.h"),
           os.path.join("..", "..", "include", "vxrt_detmath.h"),
           os.path.join("..", "..", "include", "vxrt_bluenoise.h")]

# -ffp-contract=off / no fast-math / IEEE divide+sqrt / denormals kept: include/vxrt_detmath.h
# -fno-slp-vectorize: the SLP vectoriser pairs the x/y components of the 3-vector arithmetic into v_pk_add_f32 / v_pk_mul_f32;
# on gfx950 those cost more issue time than the two scalar-per-lane instructions they replace (measured: 21.8 -> 23.3 Gray/s)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-Wall", "-Wextra",
         "-Wno-unused-parameter"]


STAMP = LIB + ".srchash"   # what the library was built from (travels with it to the GPU box; modification times do not survive the trip)


def source_hash(extra_flags=()):
    """sha256 over the sources, headers and flags that make libvxrt.so."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def needs_build(extra_flags=()):
    """True when libvxrt.so is missing or was built from other sources / flags than the tree holds now (by content)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != source_hash(extra_flags)
    except OSError:
        return True


def build(force=False, verbose=False, extra_flags=()):
    extra_flags = list(extra_flags) + os.environ.get("VXRT_HIPCC_FLAGS", "").split()
    if not force and not needs_build(extra_flags):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + extra_flags + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)      # other processes keep the library they mapped; new ones see a whole file
    with open(STAMP, "w") as f:
        f.write(source_hash(extra_flags) + "\n")
    return LIB


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


if __name__ == "__main__":
    import sys
    build(force="-f" in sys.argv, verbose=True)
    build_tool(force="-f" in sys.argv)
