"""CPU: libvxrt.so loads without a GPU and exports every function its three headers declare — include/vxrt.h (the contract),
vxrt_host.h (host-side helpers), vxrt_debug.h (test hooks and experiment options) — and the product never links the oracle."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT


HEADERS = ("vxrt.h", "vxrt_host.h", "vxrt_debug.h")


def declared_functions(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(vxrt_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_the_contract_header_holds_the_contract_only():
    """include/vxrt.h is what a host binds to render — at most 40 entry points, no test hook, no experiment's
    option; the helpers that need no context live in vxrt_host.h, everything else in vxrt_debug.h; no symbol is declared twice."""
    contract, host, debug = (declared_functions((h,)) for h in HEADERS)
    assert len(contract) <= 40, len(contract)
    for must in ("vxrt_create", "vxrt_destroy", "vxrt_resize", "vxrt_set_voxels", "vxrt_load_vox", "vxrt_set_camera", "vxrt_set_scene_params",
                 "vxrt_render", "vxrt_render_frames", "vxrt_render_path", "vxrt_render_spp", "vxrt_sync", "vxrt_read", "vxrt_read_async",
                 "vxrt_read_wait", "vxrt_get_stats", "vxrt_halo_pack", "vxrt_halo_unpack", "vxrt_last_error", "vxrt_set_noise"):
        assert must in contract, must
    assert not [n for n in contract if n.startswith("vxrt_debug_")]
    for lab in ("vxrt_create_tuned", "vxrt_detmath_probe", "vxrt_set_frame_number", "vxrt_build_features", "vxrt_build_records"):
        assert lab in debug and lab not in contract, lab
    assert [n for n in debug if n.startswith("vxrt_debug_")]
    for tool in ("vxrt_vox_to_voxels", "vxrt_build_octree", "vxrt_camera_axis_scaled", "vxrt_blue_noise", "vxrt_noise_zip_write"):
        assert tool in host and tool not in contract, tool
    assert not (set(contract) & set(host)) and not (set(contract) & set(debug)) and not (set(host) & set(debug))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vxrt.h")).read(), flags=re.S)
    opts = [int(v) for v in re.findall(r"VXRT_OPT_[A-Z_]+\s*=\s*(\d+)", text)]
    assert opts and max(opts) <= 6, opts                   # the scheduling options of experiments (7 ..) are vxrt_debug.h's
    dbg = open(os.path.join(ROOT, "include", "vxrt_debug.h")).read()
    assert [int(v) for v in re.findall(r"#define VXRT_OPT_[A-Z_]+ \(\(vxrt_option\)(\d+)\)", dbg)] == list(range(7, 24))


def test_every_declared_symbol_is_exported(H):
    names = declared_functions()
    assert len(names) >= 30 and "vxrt_render" in names and "vxrt_create" in names
    lib = H.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.vxrt_abi_version() == 6
    assert lib.vxrt_status_string(-13) == b"unexpected end of file"


def test_no_torch_types_and_c_linkage():
    from gpu_voxel_raytracer_amd import _build
    out = subprocess.run(["nm", "-D", "--defined-only", _build.LIB], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    for n in declared_functions():
        assert n in exported, n                       # unmangled => extern "C"
    assert not any("torch" in s or "at::" in s or "c10" in s for s in exported)


def test_product_does_not_touch_the_oracle():
    # the product package, the C++ host tool, the headers and the measurement scripts: none of them may reach the oracle
    for top in ("gpu_voxel_raytracer_amd", "tools", "include", "scripts"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f
                    assert not re.search(r'#include\s+"[^"]*oracle/', text), f
    # bench.py: only inside its cpu_baseline*() legs
    import ast
    bench = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(bench)
    legs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.startswith("cpu_baseline")]
    assert {n.name for n in legs} == {"cpu_baseline", "cpu_baseline_cpu_rs"}
    lines = bench.splitlines()
    for n in legs:
        for i in range(n.lineno - 1, n.end_lineno):
            lines[i] = ""
    outside = "\n".join(lines)
    assert "oracle" not in outside.replace("oracle/", "").replace("CPU oracle", "") or "import oracle" not in outside
    assert "from oracle" not in outside and "import oracle" not in outside
    from gpu_voxel_raytracer_amd import _build
    ldd = subprocess.run(["ldd", _build.LIB], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd


def test_null_and_invalid_arguments_return_errors(H):
    lib = H.lib()
    assert lib.vxrt_create(None, None) == H.E_INVALID
    assert lib.vxrt_render(None, 1) == H.E_INVALID
    assert lib.vxrt_read(None, 0, None, 0) == H.E_INVALID
    assert lib.vxrt_destroy(None) == 0
    assert lib.vxrt_read_async(None, 0, None, ctypes.c_size_t(0), 0) == H.E_INVALID
    assert lib.vxrt_read_wait(None, 0) == H.E_INVALID
    assert lib.vxrt_host_alloc(ctypes.c_size_t(64), None) == H.E_INVALID
    assert lib.vxrt_host_free(None) == 0
    assert lib.vxrt_debug_touch_map(None, 1) == H.E_INVALID
    n = ctypes.c_size_t(0)
    assert lib.vxrt_vox_to_voxels(None, 0, None, None, 0, ctypes.byref(n), None) == H.E_INVALID
    assert lib.vxrt_menger_voxels(12, None, None, None, 0, ctypes.byref(n)) == H.E_INVALID


def test_cpp_header_compiles_and_links():
    """include/vxrt.hpp (the C++ host mirror) and tools/vxrt_render.cpp build against libvxrt.so with g++ alone."""
    from gpu_voxel_raytracer_amd import _build
    tool = _build.build_tool(force=True)
    assert os.path.exists(tool)
    out = subprocess.run([tool], capture_output=True, text=True)
    assert out.returncode == 2 and "usage:" in out.stderr


def test_the_library_reads_no_environment():
    """A host process must not inherit behaviour from its environment: the product library neither imports getenv nor holds the name of
    a tuning variable (the A/B knobs arrive through vxrt_create_tuned / vxrt_set_option; host.py translates VXRT_* variables for the
    tests and scripts that ask it to)."""
    from gpu_voxel_raytracer_amd import _build
    und = subprocess.run(["nm", "-D", "--undefined-only", _build.LIB], capture_output=True, text=True).stdout
    assert "getenv" not in und
    blob = open(_build.LIB, "rb").read()
    from gpu_voxel_raytracer_amd import host
    for name in list(host.ENV_KNOBS) + ["VXRT_INFLIGHT", "VXRT_BATCH"]:
        assert name.encode() + b"\0" not in blob, name
    for src in os.listdir(os.path.join(ROOT, "gpu_voxel_raytracer_amd", "csrc")):
        if not src.endswith((".hip", ".h", ".cpp")):
            continue
        assert "getenv" not in open(os.path.join(ROOT, "gpu_voxel_raytracer_amd", "csrc", src), errors="replace").read(), src


def test_create_tuned_validates_its_options(H):
    lib = H.lib()
    cfg = H.Config(64, 64, 0, 3, 1, None, 0, 1, 16, 1, 0, 1)
    h = ctypes.c_void_p()
    assert lib.vxrt_create_tuned(ctypes.byref(cfg), None, ctypes.c_size_t(2), ctypes.byref(h)) == H.E_INVALID


def test_multi_gpu_cpp_host_compiles_and_links():
    """tools/vxrt_multi.cpp — the RCCL host a Rust host would mirror: builds with g++ against libvxrt.so,
    the HIP runtime API and librccl, and every halo entry point it needs is in the C ABI (no torch, no Python)."""
    from gpu_voxel_raytracer_amd import _build
    tool = _build.build_multi_tool(force=True)
    assert os.path.exists(tool)
    out = subprocess.run([tool], capture_output=True, text=True)
    assert out.returncode == 2 and "usage:" in out.stderr and "--transport rccl|copy" in out.stderr
    ldd = subprocess.run(["ldd", tool], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libvxrt" in ldd and "torch" not in ldd and "python" not in ldd
    und = subprocess.run(["nm", "-D", "--undefined-only", tool], capture_output=True, text=True).stdout
    for sym in ("ncclCommInitAll", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "vxrt_halo_pack", "vxrt_halo_unpack",
                "vxrt_stream_wait_context", "vxrt_context_wait_stream"):
        assert sym in und, sym


def test_header_is_plain_c_and_a_c99_client_links(tmp_path):
    """include/vxrt.h is a C header (the boundary a Rust / C host binds): it passes `gcc -std=c11 -pedantic` as C, and a C99 program
    that includes it links against libvxrt.so and sees the structs at the sizes the reference's Rust structs have (Uniforms: 148 B,
    src/context.rs:425-469)."""
    from gpu_voxel_raytracer_amd import _build
    for h in HEADERS:
        hdr = os.path.join(ROOT, "include", h)
        chk = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
        assert chk.returncode == 0 and not chk.stderr.strip(), chk.stderr
    src = tmp_path / "client.c"
    src.write_text('#include "vxrt.h"\n#include <stdio.h>\n'
                   'int main(void) { vxrt_uniforms u; vxrt_default_uniforms(&u);\n'
                   '  printf("%u %u %u %u %.2f\\n", vxrt_abi_version(), (unsigned)sizeof(vxrt_uniforms), (unsigned)sizeof(vxrt_temporal), (unsigned)sizeof(vxrt_denoise), u.sun_size);\n'
                   '  return vxrt_create(0, 0) == VXRT_E_INVALID ? 0 : 1; }\n')
    exe = tmp_path / "client"
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                            "-L" + os.path.dirname(_build.LIB), "-lvxrt", "-Wl,-rpath," + os.path.dirname(_build.LIB)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and run.stdout.split() == ["6", "148", "12", "16", "0.05"], run.stdout


def test_variants_library_loads_beside_the_product(H):
    """The -DVXRT_VARIANTS=1 build (tracers 2 / 3 / 5, the wide scene records) is test infrastructure: the parity cases of those variants
    load it BESIDE the product library in the same process (conftest.require_variants).  Here, without a GPU: it builds, exports the
    same C ABI, says what it is, and switching between the two leaves the product the one in use."""
    from conftest import require_variants
    product = H.lib()
    assert not H.has_variants()
    path = H.variants_library()
    assert os.path.basename(path) == "libvxrt_variants.so"
    require_variants(H, tracer=3)
    assert H.has_variants() and H.lib() is not product
    missing = [n for n in declared_functions() if not hasattr(H.lib(), n)]
    assert not missing, missing
    assert H.lib().vxrt_abi_version() == product.vxrt_abi_version() == 6
    require_variants(H, tracer=4)            # a default-library case after it does not switch anything by itself ...
    assert H.has_variants()
    H.use_library(None)                      # ... the autouse fixture of conftest.py does, after every test
    assert H.lib() is product and not H.has_variants()
