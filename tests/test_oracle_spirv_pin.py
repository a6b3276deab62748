"""CPU: a STATIC pin of the oracle against what the reference actually dispatches — its compiled shaders
(shaders/{voxels,temporal,denoise}.comp.spv, loaded by src/context/shader.rs:6-45).  tests/golden/spirv_reader.py reads the
binaries as data (nothing is executed) and tests/golden/make_spirv_fixture.py keeps the extracted facts in
tests/golden/spirv_pins.json: workgroup size, constants, which GLSL.std.450 built-ins are used, and the operand tree (association
order) of a list of expressions.  This test checks the ORACLE'S SOURCE against those facts mechanically: the C++ expression the
oracle evaluates (taken from oracle/oshaders.cpp by the name of the variable it assigns) is parsed into the same tree form and must
equal the compiled shader's tree, up to the rewrites listed in `canon` (commutative operand order of + and *, scalar * vector being
one operation, type conversions, pow(x, 2) -> x * x which is choice U2 of DESIGN.md).

What this pins: constants, operation order, select order, which built-in sits where.  What it cannot pin: the value any built-in
returns (exp, log, pow, sin, cos, normalize, inverse(): driver-defined — U5/U6), the sampler (U4), undefined behaviour (U1, U3, U7).
This is the static half; tests/test_oracle_spirv_exec.py is the dynamic one — it EXECUTES the same modules (oracle/ospirv.cpp) and holds
the oracle to their outputs bit for bit."""
import ast
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, have_reference

PINS = json.load(open(os.path.join(GOLDEN, "spirv_pins.json")))
ORACLE_SRC = open(os.path.join(ROOT, "oracle", "oshaders.cpp")).read()


# ---- S-expressions ----------------------------------------------------------------------------------------------------------
def parse_sexpr(text):
    tokens = re.findall(r"\(|\)|[^\s()]+", text)
    pos = 0

    def read():
        nonlocal pos
        t = tokens[pos]
        pos += 1
        if t == "(":
            out = []
            while tokens[pos] != ")":
                out.append(read())
            pos += 1
            return out
        return t
    tree = read()
    assert pos == len(tokens), text
    return tree


def number(tok):
    try:
        return float(np.float32(float(tok)))
    except (TypeError, ValueError):
        return None


COMMUTATIVE = {"FAdd", "FMul", "IAdd", "IMul", "Dot", "BitwiseOr", "BitwiseXor", "BitwiseAnd"}


def canon(t):
    """Rewrites that do not change a value: see the module docstring."""
    if isinstance(t, str):
        t = re.sub(r"%\d+\.", "", t)                      # uniform block members: %89.sun_size -> sun_size
        n = number(t)
        return f"{n:.9g}" if n is not None else t
    op, args = t[0], [canon(a) for a in t[1:]]
    if op in ("ConvertSToF", "ConvertUToF", "Bitcast"):   # int <-> float / signedness conversions of the same value
        return args[0]
    if op == "shuffle" and args[0] == args[1] and [str(i) for i in range(len(args) - 2)] == args[2:]:
        return args[0]                                    # v.xyz
    if op in ("CompositeConstruct", "vec") and len(args) > 1 and all(a == args[0] for a in args):
        return args[0]                                    # a splat is the scalar it repeats
    if op == "VectorTimesScalar":
        op = "FMul"
    if op == "Pow" and args[1] == "2":                    # U2: pow(v, 2) evaluated as v * v
        op, args = "FMul", [args[0], args[0]]
    if op == "Distance":                                  # distance(a, b) = length(a - b)
        op, args = "Length", [["FSub", args[0], args[1]]]
    op = {"IAdd": "FAdd", "IMul": "FMul", "ISub": "FSub"}.get(op, op)   # the C++ side is parsed without types: + is + (operand types are C++'s)
    if op == "FMul" and len(args) == 2 and all(isinstance(a, str) and number(a) is not None for a in args):
        return f"{float(np.float32(number(args[0])) * np.float32(number(args[1]))):.9g}"   # glslang folds a product of literals (2 * pi)
    if op in COMMUTATIVE:
        args = sorted(args, key=json.dumps)
    return [op] + args


# ---- the oracle's C++ expressions ------------------------------------------------------------------------------------------------
CALLS = {"dot": "Dot", "normalize": "Normalize", "cross": "Cross", "reflect": "Reflect", "length": "Length", "vx_max": "FMax",
         "vx_min": "FMin", "vx_exp": "Exp", "vx_log": "Log", "vx_abs": "FAbs", "vabs": "FAbs", "vx_pow": "Pow", "vx_sqrt": "Sqrt",
         "vx_cos": "Cos", "vx_sin": "Sin", "vx_clamp": "FClamp", "vx_f2i": "ConvertFToS", "vsign": "FSign", "v3": "CompositeConstruct",
         "rand": "call:rand"}
BINOPS = {ast.Add: "FAdd", ast.Sub: "FSub", ast.Mult: "FMul", ast.Div: "FDiv", ast.BitXor: "BitwiseXor", ast.BitOr: "BitwiseOr",
          ast.BitAnd: "BitwiseAnd", ast.Mod: "UMod"}
COMPARES = {ast.Lt: "FOrdLessThan", ast.NotEq: "INotEqual", ast.Gt: "FOrdGreaterThan"}


def cxx_tree(expr, rename=None, integer=False):
    """A C++ arithmetic expression of the oracle as the same kind of tree.  `rename`: oracle identifier -> shader name."""
    rename = rename or {}
    text = re.sub(r"(?<![\w.])(\d+\.?\d*(?:e[+-]?\d+)?)f\b", r"\1", expr)        # 1e4f -> 1e4
    text = re.sub(r"\b0x([0-9a-fA-F]+)\b", lambda m: str(int(m.group(1), 16)), text)
    text = text.replace("(float)", "+").replace("->", ".").replace("rng.rand()", "rand()")
    text = re.sub(r"\b(\d+)u\b", r"\1", text)
    text = " ".join(text.split())

    def walk(n):
        if isinstance(n, ast.BinOp):
            op = BINOPS[type(n.op)]
            if integer:
                op = {"FAdd": "IAdd", "FMul": "IMul"}.get(op, op)
            return [op, walk(n.left), walk(n.right)]
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.USub):
            if isinstance(n.operand, ast.Constant):
                return repr(-n.operand.value)
            return ["FNegate", walk(n.operand)]
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.UAdd):              # a (float) cast
            return walk(n.operand)
        if isinstance(n, ast.Call):
            name = n.func.attr if isinstance(n.func, ast.Attribute) else n.func.id
            return [CALLS[name]] + [walk(a) for a in n.args]
        if isinstance(n, ast.Compare):
            return [COMPARES[type(n.ops[0])], walk(n.left), walk(n.comparators[0])]
        if isinstance(n, ast.Attribute):                                            # u.sun_size, d.x
            base = walk(n.value)
            if n.attr in "xyz" and len(n.attr) == 1:
                return rename.get(f"{base}.{n.attr}", f"{rename.get(base, base)}[{'xyz'.index(n.attr)}]")
            return rename.get(n.attr, n.attr) if base in ("u", "tu", "du") else f"{base}.{n.attr}"
        if isinstance(n, ast.Name):
            return rename.get(n.id, n.id)
        if isinstance(n, ast.Constant):
            return repr(n.value)
        raise AssertionError(ast.dump(n))
    return walk(ast.parse(text, mode="eval").body)


def oracle_rhs(variable, occurrence=0, declared=True):
    """Right-hand side of the (occurrence-th) assignment to `variable` in oracle/oshaders.cpp."""
    pat = (r"(?:float|V3|bool|uint32_t|int32_t|const\s+\w+)\s+" if declared else r"(?<![\w.>])") + re.escape(variable) + r"\s*=\s*([^;]+);"
    found = re.findall(pat, ORACLE_SRC)
    assert len(found) > occurrence, (variable, len(found))
    return found[occurrence]


def same(expr_tree, pin_text):
    a, b = canon(expr_tree), canon(parse_sexpr(pin_text) if "(" in pin_text else pin_text)
    assert a == b, f"\noracle : {json.dumps(a)}\nshader : {json.dumps(b)}"


# ---- the pins ---------------------------------------------------------------------------------------------------------------
# key in the fixture -> (oracle variable, which assignment, declared with a type?, renames oracle -> shader, substitutions applied
# to the oracle's text first: intermediates the oracle names and the shader does not)
DENOISE = {
    "main/sigma_distance_2#0": ("sigma_distance_2", 0, True, {}, {}),
    "main/sigma_range_2#0": ("sigma_range_2", 0, True, {}, {}),
    "main/depth_bias#0": ("depth_bias", 0, True, {"cn": "center_normal_depth"}, {}),
    "main/depth_delta#0": ("depth_delta", 0, True, {"cd": "center_normal_depth[3]", "wd": "window_normal_depth[3]"}, {}),
    "main/factor_range#0": ("factor_range", 0, True, {}, {"bd": "(depth_bias * depth_delta)"}),
    "main/factor_distance#0": ("factor_distance", 0, True, {}, {}),
    "main/factor#0": ("factor", 0, True, {}, {}),
}
TEMPORAL = {
    "main/world_pos#0": ("world_pos", 0, True, {"cam_o": "camera_origin"}, {}),
    "main/old_ray_dir#0": ("old_ray_dir", 0, True, {"ocr": "old_camera_right", "ocu": "old_camera_up", "ocf": "old_camera_forward",
                                                    "sx": "old_screen[0]", "sy": "old_screen[1]"}, {}),
    "main/old_position#0": ("old_position", 0, True, {"oco": "old_camera_origin"}, {}),
    "main/camera_dir#0": ("camera_dir", 0, True, {"cam_o": "camera_origin"}, {}),
    "main/bias#0": ("bias", 0, True, {}, {}),
    "main/dist#0": ("dist", 0, True, {}, {}),
    "main/same_position#0": ("same_position", 0, True, {}, {}),
    "main/next_blending#0": ("next_blending", 0, True, {}, {}),
}
SUN = "(sun_strength * sun_color)"      # the oracle hoists SUN_COLOR = sun_color.xyz * sun_strength (voxels.comp:6) into `sun_color`
VOXELS = {
    "main/sun_dir#0": ("sun_dir", 0, True, {}, {}),
    "main/hit_pos#0": ("hit_pos", 0, True, {}, {"h.time": "time"}),
    "main/reflect_dir#0": ("reflect_dir", 0, True, {}, {}),
    "main/blending_factor#1": ("blending_factor", 1, False, {}, {}),
    "main/blending_factor#2": ("blending_factor", 2, False, {}, {}),
    "main/ray_origin#1": ("ray_origin", 1, False, {}, {}),
    "main/up_dir#0": ("up_dir", 0, True, {}, {}),
    "main/right_dir#0": ("right_dir", 0, True, {}, {}),
    "main/dx#0": ("dx", 1, True, {}, {}),
    "main/light_dir#0": ("light_dir", 0, True, {}, {}),
    "main/sample_color#1": ("sample_color", 1, False, {}, {}),
    "main/sample_color#2": ("sample_color", 2, False, {}, {"sun_color": SUN}),
    "main/sample_color#4": ("sample_color", 4, False, {"sky": "sky_color"}, {"sun_color": SUN}),
    "main/sample_color#5": ("sample_color", 5, False, {"sky": "sky_color"}, {}),
    "main/sun_power#0": ("sun_power", 0, True, {}, {}),
    "cast_bounded_ray/t_mid#0": ("t_mid", 0, True, {}, {}),
    "cast_bounded_ray/max_dist#0": ("max_dist", 0, True, {"distances.x": "distances[0]"}, {}),
    "cast_bounded_ray/next_time#0": ("next_time", 0, True, {}, {}),
    "cast_bounded_ray/distances#0": ("distances", 0, True, {"oc": "octant_center"}, {}),
    "cast_bounded_ray/time#0": ("time", 0, True, {}, {}),
    "random_hemisphere/phi#0": ("phi", 0, True, {}, {}),
    "random_hemisphere/plane_radius#0": ("plane_radius", 0, True, {"d": "reflect_dir"}, {}),
    "random_hemisphere/reflect_dir[1]#0": ("d.y", 0, False, {}, {}),
    "random_hemisphere/reflect_dir[2]#0": ("d.z", 0, False, {}, {}),
    "random_hemisphere/reflect_dir#0": ("d", 0, False, {"d": "reflect_dir"}, {}),
}


def check(shader, table):
    for key, (var, k, declared, rename, subst) in table.items():
        rhs = oracle_rhs(var, k, declared)
        for name, text in subst.items():
            rhs = re.sub(r"(?<![\w.])" + re.escape(name) + r"\b", text, rhs)
        try:
            same(cxx_tree(rhs, rename), PINS[shader]["expressions"][key])
        except AssertionError as e:
            raise AssertionError(f"{shader} {key}: oracle `{' '.join(rhs.split())}`{e}") from None


def test_denoise_expressions_have_the_compiled_shaders_order():
    """denoise.comp:39-40, 49, 66-80: the four range terms are summed left to right and divided once; exp(-range - distance)."""
    check("denoise", DENOISE)
    assert canon(parse_sexpr(PINS["denoise"]["expressions"]["main/material_delta#0"])) == \
        ["Select", ["INotEqual", ["ShiftRightArithmetic", "center_material", "24"], ["ShiftRightArithmetic", "window_material", "24"]], "1", "0"]
    assert "(cmat >> 24) != (wmat >> 24) ? 1.0f : 0.0f" in ORACLE_SRC
    # the filtered colour: out += sum / normalization, then mix(out, albedo * out, albedo_factor)   denoise.comp:88-90
    assert canon(parse_sexpr(PINS["denoise"]["expressions"]["main/out_color#2"])) == ["FMix", "out_color", ["FMul", "center_albedo", "out_color"], "albedo_factor"]


def test_temporal_expressions_have_the_compiled_shaders_order():
    """temporal.comp:60-65, 94-124: reprojection arithmetic, the acceptance test dist < (bias * cutoff) * depth, the blending update."""
    check("temporal", TEMPORAL)
    # tex_coord = (old_screen.xy + vec2(0.5, -0.5)) * vec2(1 / size.x, -1 / size.y)   temporal.comp:88-89 — the oracle's two scalars
    assert canon(parse_sexpr(PINS["temporal"]["expressions"]["main/tex_coord#0"])) == \
        ["FMul", ["CompositeConstruct", ["FDiv", "1", "size[0]"], ["FDiv", "-1", "size[1]"]], ["FAdd", "old_screen", ["vec", "0.5", "-0.5"]]]
    assert "(sx + 0.5f) * (1.0f / (float)width)" in ORACLE_SRC and "(sy + -0.5f) * (-1.0f / (float)height)" in ORACLE_SRC
    # the perspective divide touches x and y only (temporal.comp:85) and the matrix is inverted by the built-in (U5)
    assert PINS["temporal"]["expressions"]["main/old_world_to_screen#0"] == "(MatrixInverse old_screen_to_world)"
    assert "FDiv (shuffle old_screen old_screen 0 1) (CompositeConstruct old_screen[2] old_screen[2])" in PINS["temporal"]["expressions"]["main/old_screen#1"]
    assert "sx = sx / sz; sy = sy / sz;" in ORACLE_SRC


def test_voxels_expressions_have_the_compiled_shaders_order():
    """voxels.comp:191-197, 277-287, 296, 326-388: the shading sums and products, the hemisphere sample, the mid-plane times."""
    check("voxels", VOXELS)
    ret = PINS["voxels"]["returns"]
    # node_emmitance: (vec3(r, g, b) * (e * emit_strength)) / 255   voxels.comp:260-266
    same(cxx_tree("((e * emit_strength) * v3(r, g, b)) / 255.0f"), ret["node_emmitance"][0])
    assert "return ((e * emit_strength) * v3(r, g, b)) / 255.0f;" in ORACLE_SRC
    same(cxx_tree("v3(r, g, b) / 255.0f"), ret["node_color"][0])
    assert "return v3(r, g, b) / 255.0f;" in ORACLE_SRC
    # octant_center: center + (delta - 0.5) * (0.5 * size)   voxels.comp:92-95
    same(cxx_tree("center + (0.5f * size) * (delta - 0.5f)"), ret["octant_center"][0])
    assert "return center + (0.5f * size) * (delta - v3s(0.5f));" in ORACLE_SRC
    # the ray / cube slab test   voxels.comp:73-90
    check("voxels", {"ray_cube_intersection/entry_planes#0": ("entry_planes", 0, True, {}, {}),
                     "ray_cube_intersection/exit_planes#0": ("exit_planes", 0, True, {}, {}),
                     "ray_cube_intersection/entries#0": ("entries", 0, True, {}, {}),
                     "ray_cube_intersection/exits#0": ("exits", 0, True, {}, {}),
                     "ray_cube_intersection/entry#0": ("*entry", 0, False, {}, {}),
                     "ray_cube_intersection/exit#0": ("*exit", 0, False, {}, {})})
    e = PINS["voxels"]["expressions"]
    # the noise index: x % 128 + (y % 128) * 128 + ((frame % 512) * 128) * 128, advanced by 16384 mod 8388608   voxels.comp:268-275
    assert canon(parse_sexpr(e["rand/random_index#0"])) == ["UMod", ["FAdd", "16384", "random_index"], "8388608"]   # canon() spells integer + as FAdd
    assert "index = (index + BLUE_NOISE_SIZE * BLUE_NOISE_SIZE) % BLUE_NOISE_BUFFER_SIZE;" in ORACLE_SRC


def test_select_order_of_the_transition_and_the_locked_planes():
    """voxels.comp:196-200: mid_intersect is a SELECT (mix with a bvec3: no arithmetic), and transition tests x first, then y, then
    z — the order that breaks ties between equal plane times."""
    d = PINS["voxels"]["decisions"]["cast_bounded_ray/transition"]
    assert d == [["if", "plane[0]"], ["pick", "4"], ["if", "plane[1]"], ["pick", "2"], ["pick", "(Select plane[2] 1 0)"]]
    assert re.search(r"transition = \(mid_intersect\.x == next_time\) \? 4 : \(\(mid_intersect\.y == next_time\) \? 2 : \(\(mid_intersect\.z == next_time\) \? 1 : 0\)\);",
                     ORACLE_SRC)
    assert PINS["voxels"]["expressions"]["cast_bounded_ray/mid_intersect#0"] == "(Select locked (vec 1.07374182e+09 1.07374182e+09 1.07374182e+09) t_mid)"
    assert PINS["voxels"]["expressions"]["cast_bounded_ray/dir_mask#0"] == \
        "(Bitcast (BitwiseOr (BitwiseOr (Select (FOrdLessThan ray_dir[0] 0) 4 0) (Select (FOrdLessThan ray_dir[1] 0) 2 0)) (Select (FOrdLessThan ray_dir[2] 0) 1 0)))"
    assert "(ray_dir.x < 0.0f ? 4 : 0) | (ray_dir.y < 0.0f ? 2 : 0) | (ray_dir.z < 0.0f ? 1 : 0)" in ORACLE_SRC
    # current_octant: strict > (ties to the low side)   voxels.comp:119-125
    assert PINS["voxels"]["expressions"]["current_octant/dx#0"] == "(Bitcast (Select (FOrdGreaterThan delta[0] 0) 4 0))"
    assert "uint32_t dx = delta.x > 0.0f ? 4 : 0;" in ORACLE_SRC


def oracle_function_body(signature):
    """Text of the oracle function whose definition starts with `signature`, braces balanced."""
    i = ORACLE_SRC.index(signature)
    j = ORACLE_SRC.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(ORACLE_SRC[k], 0)
        if depth == 0:
            return ORACLE_SRC[j:k + 1]
        k += 1


def test_call_order_and_loop_exits():
    """The ORDER of things with side effects: the rand() draws of a hit (voxels.comp:326-371: the specular test first, then the three
    components of rand_dir, dx, dy, and random_hemisphere's two after the sun ray), and the exit tests of cast_bounded_ray's loop
    (voxels.comp:163-177, 204-234: trip cap, then distance, then the leaf test; has_next's three conditions; the pop loop's two)."""
    v = PINS["voxels"]
    assert v["calls"]["main"] == ["cast_ray", "node_color", "node_emmitance", "rand", "rand", "rand", "rand", "rand", "rand", "cast_ray",
                                  "random_hemisphere", "node_color"]
    assert v["calls"]["random_hemisphere"] == ["rand", "rand"] and v["calls"]["cast_ray"] == ["cast_bounded_ray"]
    body = oracle_function_body("static int trace_pixel(")
    tokens = re.findall(r"\b(cast_bounded_ray|node_color|node_emmitance|rng\.rand|random_hemisphere)\(", body)
    assert [{"cast_bounded_ray": "cast_ray", "rng.rand": "rand"}.get(t, t) for t in tokens] == v["calls"]["main"]
    hemi = oracle_function_body("static V3 random_hemisphere(")
    assert re.findall(r"rng\.rand\(\)", hemi) == ["rng.rand()"] * 2 and hemi.index("phi") < hemi.index("d.x = ")      # phi's draw first
    assert v["calls"]["cast_bounded_ray"] == ["ray_cube_intersection", "current_octant", "octant_center", "octant_center", "current_octant",
                                               "ray_cube_intersection", "octant_center", "ray_cube_intersection"]
    marks = [m for m in v["landmarks"]["cast_bounded_ray"] if m[0] != "FOrdLessThan"]     # without dir_mask's three sign tests (pinned above)
    assert marks == [
        ["LogicalNot", "intersect"], ["return", "False"], ["loop"],
        ["SGreaterThanEqual", "iterations", "2048"], ["return", "True"],
        ["FOrdGreaterThan", "time", "max_distance"], ["return", "False"],
        ["SLessThan", "value", "0"], ["return", "True"],
        ["INotEqual", "(BitwiseAnd directional_octant 4)", "0"], ["INotEqual", "(BitwiseAnd directional_octant 2)", "0"],
        ["INotEqual", "(BitwiseAnd directional_octant 1)", "0"],
        ["FOrdLessThanEqual", "next_time", "exit"], ["INotEqual", "transition", "0"], ["IEqual", "(BitwiseAnd directional_octant transition)", "0"],
        ["SGreaterThan", "value", "0"], ["loop"], ["IEqual", "top", "0"], ["return", "False"], ["IEqual", "node", "-1"], ["return", marks[-1][1]]]
    walk = oracle_function_body("bool cast_bounded_ray(const Scene& sc")
    order = ["if (!intersect) return false", "iterations >= 2048", "time > max_distance", "value < 0",
             "next_time <= exit && transition != 0 && (directional_octant & transition) == 0", "value > 0", "top == 0", "node == -1"]
    at = [walk.index(t) for t in order]
    assert at == sorted(at)


def test_comparisons_of_temporal_and_denoise():
    """Which comparison is strict: temporal.comp:92 accepts texture coordinates in [0, 1] INCLUSIVE on both sides and only for
    depth >= 0 (:68, :119), accepts a history texel when dist < (bias cutoff) depth (:113); denoise.comp:51-57 loops dy, dx over
    [-r, r] inclusive and takes a tap when 0 <= n < size on both axes; radius == 0 bypasses the sums (:89)."""
    t = PINS["temporal"]["landmarks"]["main"]
    assert t == [["FOrdGreaterThanEqual", "depth", "0"], ["FOrdLessThanEqual", "0", "tex_coord[0]"], ["FOrdLessThanEqual", "tex_coord[0]", "1"],
                 ["FOrdLessThanEqual", "0", "tex_coord[1]"], ["FOrdLessThanEqual", "tex_coord[1]", "1"],
                 ["FOrdLessThan", "dist", "(FMul (FMul bias %290.blending_distance_cutoff) depth)"], ["FOrdGreaterThanEqual", "depth", "0"]]
    body = oracle_function_body("void orc_temporal(")
    order = ["if (depth >= 0.0f && has_history)", "0.0f <= tu_ && tu_ <= 1.0f && 0.0f <= tv_ && tv_ <= 1.0f",
             "dist < (bias * tu->blending_distance_cutoff) * depth", "depth >= 0.0f ? vmix("]
    at = [body.index(x) for x in order]
    assert at == sorted(at)
    d = [m for m in PINS["denoise"]["landmarks"]["main"] if m[0] != "INotEqual"]
    assert d == [["loop"], ["SLessThanEqual", "dy", "r"], ["loop"], ["SLessThanEqual", "dx", "r"], ["SLessThanEqual", "0", "nx"],
                 ["SLessThan", "nx", "size[0]"], ["SLessThanEqual", "0", "ny"], ["SLessThan", "ny", "size[1]"], ["IEqual", "%68.radius", "0"]]
    body = oracle_function_body("void orc_denoise(")
    order = ["for (int dy = -r; dy <= r; dy++)", "for (int dx = -r; dx <= r; dx++)", "0 <= nx && nx < width && 0 <= ny && ny < height",
             "du->radius == 0 ? cc : sum / normalization"]
    at = [body.index(x) for x in order]
    assert at == sorted(at)


def test_constants_workgroups_and_builtins():
    v, t, d = PINS["voxels"], PINS["temporal"], PINS["denoise"]
    assert v["local_size"] == t["local_size"] == d["local_size"] == [16, 16, 1]
    f32 = lambda x: float(f"{np.float32(x):.9g}")                                    # noqa: E731
    # voxels.comp: the iteration cap, the bounce offset, ALMOST_INFINITY, /255, 2 pi, the noise geometry, the miss node
    assert {2048, 16384, 8388608, 128, 512, 255, 0xffffff, 1 << 30, -(1 << 31)} <= set(v["int_constants"])
    assert {f32(1e-5), f32(2.0 ** 30), 255.0, f32(2 * np.float32(3.14159265358979)), 0.5, 2.0} <= set(v["float_constants"])
    assert "1e-5f * normal" in ORACLE_SRC and "iterations >= 2048" in ORACLE_SRC and "(2.0f * 3.14159265358979f)" in ORACLE_SRC
    assert re.search(r"ALMOST_INFINITY\s*=\s*1073741824\.0f", open(os.path.join(ROOT, "oracle", "oracle.h")).read() + ORACLE_SRC)
    # denoise.comp: the 1e4 weights; temporal.comp: the half-texel offsets
    assert 10000.0 in d["float_constants"] and {0.5, -0.5, -1.0} <= set(t["float_constants"])
    assert ORACLE_SRC.count("1e4f") == 3
    # which built-ins the driver is asked for, and how often (the oracle defines each of them itself: U5 / U6)
    assert v["ext_insts"] == {"Cos": 4, "Cross": 2, "FAbs": 1, "FMax": 8, "FMin": 5, "FSign": 2, "Normalize": 8, "Pow": 2, "Reflect": 1, "Sin": 3, "Sqrt": 1}
    assert t["ext_insts"] == {"Distance": 1, "FClamp": 1, "FMax": 1, "FMix": 1, "MatrixInverse": 1, "Normalize": 3}
    assert d["ext_insts"] == {"Exp": 1, "FAbs": 2, "FMax": 1, "FMix": 1, "Log": 2, "Normalize": 1, "Pow": 3}
    # no fused multiply-add, no inversesqrt, no exp2 / log2 in any of the three: the oracle may not use them either
    for p in (v, t, d):
        assert not {"Fma", "InverseSqrt", "Exp2", "Log2"} & set(p["ext_insts"])


@pytest.mark.skipif(not have_reference(), reason="/root/reference not mounted (GPU box)")
def test_fixture_is_what_the_script_extracts_from_the_reference():
    import sys
    sys.path.insert(0, GOLDEN)
    import make_spirv_fixture as M
    assert json.loads(json.dumps(M.make(), sort_keys=True)) == PINS


def test_reader_rejects_what_is_not_spirv():
    import sys
    sys.path.insert(0, GOLDEN)
    import spirv_reader as R
    for bad in (b"", b"\x00" * 20, b"\x03\x02\x23\x07" + b"\x00" * 15, b"\x03\x02\x23\x07" + b"\x00" * 16 + b"\x05\x00\x09\x00"):
        with pytest.raises(ValueError):
            R.Module(bad)
