#!/usr/bin/env python3
"""Writes tests/golden/spirv_pins.json: what the reference's COMPILED compute shaders (shaders/{voxels,temporal,denoise}.comp.spv —
the binaries src/context/shader.rs:6-45 loads and dispatches) say about constants, workgroup size, the GLSL.std.450 built-ins used,
and the operand trees of a list of expressions.  The shaders are read as data by tests/golden/spirv_reader.py; nothing is executed.

Run in the build container (needs /root/reference):   python tests/golden/make_spirv_fixture.py
tests/test_oracle_spirv_pin.py checks the oracle's restatement against the fixture (and, when the reference is mounted, that the
fixture is what this script produces).

The fixture holds facts ABOUT the binaries — a list of constants, instruction counts, the association order of the listed
expressions, the order in which four functions call others (the order of the rand() draws) and the order of the comparisons and returns of
cast_bounded_ray's loop — not the shaders' text: declarations, bindings, the bodies of the branches and everything not listed are not in it."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import spirv_reader as R  # noqa: E402

REFERENCE_SHADERS = "/root/reference/shaders"
OUT = os.path.join(HERE, "spirv_pins.json")

# (function prefix, variable, which store: index into the stores to that variable in instruction order)
EXPRESSIONS = {
    "voxels": [
        ("main", "random_index", 0), ("main", "sun_dir", 0), ("main", "ray_dir", 0), ("main", "hit_pos", 0),
        ("main", "reflect_dir", 0), ("main", "blending_factor", 1), ("main", "blending_factor", 2), ("main", "ray_origin", 1),
        ("main", "up_dir", 0), ("main", "right_dir", 0), ("main", "dx", 0), ("main", "light_dir", 0),
        ("main", "sample_color", 1), ("main", "sample_color", 2), ("main", "sample_color", 4), ("main", "sample_color", 5),
        ("main", "sun_power", 0), ("main", "out_color", 0),
        ("ray_cube_intersection", "signum", 0), ("ray_cube_intersection", "entry_planes", 0), ("ray_cube_intersection", "exit_planes", 0),
        ("ray_cube_intersection", "entries", 0), ("ray_cube_intersection", "exits", 0), ("ray_cube_intersection", "entry", 0),
        ("ray_cube_intersection", "exit", 0),
        ("octant_center", "delta", 0), ("current_octant", "dx", 0),
        ("cast_bounded_ray", "dir_mask", 0), ("cast_bounded_ray", "ray_inv_dir", 0), ("cast_bounded_ray", "time", 0),
        ("cast_bounded_ray", "value", 0), ("cast_bounded_ray", "hit", 0), ("cast_bounded_ray", "distances", 0),
        ("cast_bounded_ray", "max_dist", 0), ("cast_bounded_ray", "normal", 0), ("cast_bounded_ray", "t_mid", 0),
        ("cast_bounded_ray", "directional_octant", 0), ("cast_bounded_ray", "mid_intersect", 0), ("cast_bounded_ray", "next_time", 0),
        ("cast_bounded_ray", "next_octant", 0), ("cast_bounded_ray", "time", 1), ("cast_bounded_ray", "size", 1),
        ("cast_bounded_ray", "size", 2), ("cast_bounded_ray", "center", 2),
        ("node_emmitance", "e", 0), ("rand", "random_index", 0),
        ("random_hemisphere", "phi", 0), ("random_hemisphere", "reflect_dir[0]", 0), ("random_hemisphere", "plane_radius", 0),
        ("random_hemisphere", "reflect_dir[1]", 0), ("random_hemisphere", "reflect_dir[2]", 0), ("random_hemisphere", "reflect_dir", 0),
    ],
    "temporal": [
        ("main", "world_pos", 0), ("main", "old_world_to_screen", 0), ("main", "old_screen", 0), ("main", "old_screen", 1),
        ("main", "tex_coord", 0), ("main", "old_ray_dir", 0), ("main", "old_position", 0), ("main", "camera_dir", 0),
        ("main", "bias", 0), ("main", "dist", 0), ("main", "same_position", 0), ("main", "next_blending", 0),
    ],
    "denoise": [
        ("main", "ray_dir", 0), ("main", "sigma_distance_2", 0), ("main", "sigma_range_2", 0), ("main", "depth_bias", 0),
        ("main", "depth_delta", 0), ("main", "material_delta", 0), ("main", "factor_range", 0), ("main", "factor_distance", 0),
        ("main", "factor", 0), ("main", "normalization", 1), ("main", "sum", 1), ("main", "out_color", 2),
    ],
}
RETURNS = {"voxels": ["octant_center", "current_octant", "node_color", "node_emmitance", "rand"]}
# (function prefix, after the store to, up to the store to): the conditional branches and the values chosen between the two
DECISIONS = {"voxels": [("cast_bounded_ray", "plane", "transition")]}
# functions whose sequence of calls (in instruction order = source order: glslang does not reorder) is kept: the order of the rand() draws
CALLS = {"voxels": ["main", "cast_bounded_ray", "cast_ray", "random_hemisphere"]}
# functions whose comparisons and returns are kept in instruction order: which exit test of the loop comes first
LANDMARKS = {"voxels": ["cast_bounded_ray"], "temporal": ["main"], "denoise": ["main"]}
LANDMARK_OPS = ("SGreaterThanEqual", "FOrdGreaterThan", "SLessThan", "SGreaterThan", "IEqual", "INotEqual", "FOrdLessThanEqual", "LogicalNot",
                "ReturnValue", "LoopMerge", "SLessThanEqual", "FOrdLessThan", "FOrdGreaterThanEqual")


def find_function(m, prefix):
    names = [n for n in m.functions() if n == prefix or n.startswith(prefix + "(")]
    if len(names) != 1:
        raise KeyError(prefix)
    return names[0]


def decision_chain(m, function, after, until):
    """Between the store to `after` and the store to `until`: the condition of every OpBranchConditional and every value stored
    to a compiler temporary, in instruction order — how a nested ?: was lowered, i.e. which test comes first and what it picks."""
    out, on = [], False
    for ins in m.functions()[function]:
        if ins.name == "Store" and m.names.get(ins.words[0]) == after:
            on = True
            continue
        if not on:
            continue
        if ins.name == "Store" and m.names.get(ins.words[0]) == until:
            break
        if ins.name == "BranchConditional":
            out.append(["if", m.tree(ins.words[0])])
        elif ins.name == "Store" and m.names.get(ins.words[0], "") == "":
            d = m.defs.get(ins.words[1])
            if d is not None and d.name == "Load" and m.names.get(d.words[2], "") == "":
                continue                                   # a temporary copied into the enclosing temporary
            out.append(["pick", m.tree(ins.words[1])])
    return out


def landmarks(m, function):
    out = []
    for ins in m.functions()[function]:
        if ins.name not in LANDMARK_OPS:
            continue
        if ins.name == "LoopMerge":
            out.append(["loop"])
        elif ins.name == "ReturnValue":
            out.append(["return", m.tree(ins.words[0])])
        else:
            out.append([ins.name] + [m.tree(w) for w in ins.words[2:]])
    return out


def pins_of(name):
    path = os.path.join(REFERENCE_SHADERS, f"{name}.comp.spv")
    data = open(path, "rb").read()
    m = R.Module(data)
    out = {
        "binary": {"file": f"shaders/{name}.comp.spv", "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                   "spirv_version": f"{(m.version >> 16) & 0xff}.{(m.version >> 8) & 0xff}", "generator": hex(m.generator)},
        "local_size": m.local_size,
        "float_constants": [float(f"{v:.9g}") for v in m.float_constants()],
        "int_constants": m.int_constants(),
        "ext_insts": m.ext_inst_counts(),
        "expressions": {}, "returns": {}, "decisions": {}, "calls": {}, "landmarks": {},
    }
    for prefix, var, k in EXPRESSIONS.get(name, []):
        fn = find_function(m, prefix)
        stores = [t for p, t in m.stores(fn) if p == var]
        out["expressions"][f"{prefix}/{var}#{k}"] = stores[k]
    for prefix in RETURNS.get(name, []):
        out["returns"][prefix] = m.returns(find_function(m, prefix))
    for prefix in CALLS.get(name, []):
        out["calls"][prefix] = [m.names.get(i.words[2], "?").split("(")[0] for i in m.functions()[find_function(m, prefix)] if i.name == "FunctionCall"]
    for prefix in LANDMARKS.get(name, []):
        out["landmarks"][prefix] = landmarks(m, find_function(m, prefix))
    for prefix, after, until in DECISIONS.get(name, []):
        out["decisions"][f"{prefix}/{until}"] = decision_chain(m, find_function(m, prefix), after, until)
    return out


def make():
    return {name: pins_of(name) for name in ("voxels", "temporal", "denoise")}


if __name__ == "__main__":
    with open(OUT, "w") as f:
        json.dump(make(), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
