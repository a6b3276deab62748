"""Builds libvxrt.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvxrt.so")
SOURCES = ["vxrt_api.hip", "trace.hip", "trace_wavefront.hip", "trace_tail.hip", "trace_paths.hip", "post.hip", "noise.hip", "scene_host.cpp", "scene_procedural.cpp",
           "noise_zip.cpp", "vox_scene.cpp"]
HEADERS = ["kernels.h", "trace_common.h", "scene_host.h", "vx_vec.h", os.path.join("..", "..", "include", "vxrt.h"),
           os.path.join("..", "..", "include", "vxrt_detmath.h"),
           os.path.join("..", "..", "include", "vxrt_bluenoise.h")]

# -ffp-contract=off / no fast-math / IEEE divide+sqrt / denormals kept: include/vxrt_detmath.h
# -fno-slp-vectorize: the SLP vectoriser pairs the x/y components of the 3-vector arithmetic into v_pk_add_f32 / v_pk_mul_f32;
# on gfx950 those cost more issue time than the two scalar-per-lane instructions they replace (measured: 21.8 -> 23.3 Gray/s)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-Wall", "-Wextra",
         "-Wno-unused-parameter"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra_flags = list(extra_flags) + os.environ.get("VXRT_HIPCC_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra_flags + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
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
