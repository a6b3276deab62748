"""Instruction budget of the octree walk from the compiler's gfx950 assembly.

    python scripts/isa_budget.py [--kernel bounce_kernel] [--source trace_tail.hip] [--out profiles/r04/walkf_step_isa.md] [-D...]

Compiles one kernel file of gpu_voxel_raytracer_amd/csrc with the product's flags + `-gline-tables-only -S --cuda-device-only`
(hipcc cross-compiles; no GPU needed), reads the .s, attributes every instruction of the named kernel to the line of
`walkf_step` (csrc/trace_common.h) it was inlined from — the `.loc` comments carry the inline chain — and writes:
  * registers, spills, LDS of the kernel;
  * per REGION of walkf_step (loop head + exits, the sibling search every trip executes, descend only, pop only, what descend and pop
    share, advance) the number of VALU / SALU / LDS / VMEM / branch / s_waitcnt instructions: the static price of a wave-trip that
    takes that branch (a wave whose lanes split over the branches executes them all);
  * the annotated listing of the walk (instruction, region, source line).
Static counts: an instruction inside a divergent region is counted once however many lanes are active.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu_voxel_raytracer_amd", "csrc")
sys.path.insert(0, ROOT)


def product_flags():
    from gpu_voxel_raytracer_amd import _build
    return [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]


def classify(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return "other"


LOC = re.compile(r"\.loc\s+(\d+)\s+(\d+)\s+\d+.*?;\s*(.*)$")
CHAIN = re.compile(r"([^\s\[\]@]+):(\d+):\d+")


def walk_regions(src_lines):
    """line number -> region of walkf_step, found by the text of the function (not by fixed line numbers)."""
    start = next(i for i, l in enumerate(src_lines, 1) if "int walkf_step(" in l)
    end = next(i for i in range(start, len(src_lines) + 1) if src_lines[i - 1].startswith("}"))
    region, out = "head", {}
    depth_marks = []
    for i in range(start, end + 1):
        l = src_lines[i - 1]
        if "const f3 tm = " in l:
            region = "search"
        elif "if (is_child || !has_next) {" in l:
            region = "shared"
        elif re.search(r"if \(is_child\) \{\s*// voxels.comp:205-214", l):
            region = "descend"
        elif re.search(r"\} else \{\s*// voxels.comp:225-234", l):
            region = "pop"
        elif "const float size = __builtin_ldexpf" in l:
            region = "shared"
        elif re.search(r"if \(is_child\) \{\s*// voxels.comp:216-221", l):
            region = "descend"
        elif re.search(r"\} else \{\s*// voxels.comp:236-242", l):
            region = "pop"
        elif "w.exit = vx_min3(w.ex.x" in l:
            region = "shared"
        elif re.search(r"\} else \{\s*// voxels.comp:222-224", l):
            region = "advance"
        out[i] = region
    return start, end, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="bounce_kernel")
    ap.add_argument("--source", default="trace_tail.hip")
    ap.add_argument("--out", default="")
    ap.add_argument("--keep", default="", help="keep the .s here")
    ap.add_argument("-D", action="append", default=[], dest="defs")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        s_path = args.keep or os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc"] + product_flags() + ["-D" + d for d in args.defs] + ["-gline-tables-only", "-x", "hip", "--cuda-device-only", "-S",
                                                                                        os.path.join(CSRC, args.source), "-o", s_path]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(s_path).read().splitlines()
    common = open(os.path.join(CSRC, "trace_common.h")).read().splitlines()
    w0, w1, regions = walk_regions(common)

    # the kernel's body: from its label to its .end_amdhsa_kernel / s_endpgm
    label = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:\s*(;.*)?$" % args.kernel, l))
    name = text[label].split(":")[0]
    body_end = next(i for i in range(label, len(text)) if text[i].strip().startswith(".Lfunc_end"))
    meta = {}
    for i, l in enumerate(text):
        if l.strip().startswith(".name:") and name in l:
            for m in text[i:i + 25]:
                mm = re.match(r"\s*\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):\s*(\d+)", m)
                if mm:
                    meta[mm.group(1)] = int(mm.group(2))
    cur = ("?", 0, [])
    rows = []   # (text, class, region, attributed line, innermost)
    for l in text[label + 1:body_end]:
        m = LOC.search(l)
        if m:
            chain = [(os.path.basename(f), int(n)) for f, n in CHAIN.findall(m.group(3))]
            cur = (chain[0] if chain else ("?", 0), chain)
            continue
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            if t.endswith(":") and not t.startswith("."):
                pass
            if re.match(r"^\.LBB\d+_\d+:", t):
                rows.append((t, "label", "", 0, None))
            continue
        op = t.split()[0]
        chain = cur[1] if len(cur) > 1 else []
        line = next((n for f, n in chain if f == "trace_common.h" and w0 <= n <= w1), 0)
        region = regions.get(line, "") if line else ""
        where = next(((f, n) for f, n in chain if f in ("trace_common.h", args.source)), cur[0])
        rows.append((t.split(";")[0].rstrip(), classify(op), region, line, where))

    classes = ["VALU", "SALU", "LDS", "VMEM", "branch", "wait", "nop"]
    by_region = collections.OrderedDict((r, collections.Counter()) for r in ("head", "search", "shared", "descend", "pop", "advance"))
    by_line = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    for t, c, region, line, where in rows:
        if c == "label":
            continue
        total[c] += 1
        if region:
            by_region[region][c] += 1
            by_line[line][c] += 1
    # how many copies of the walk loop the kernel holds (one per cast site the compiler inlined it into): the six operations of
    # `tm = (center - o) * inv` appear once per copy
    tm_line = next(i for i, l in enumerate(common, 1) if w0 <= i <= w1 and "const f3 tm = " in l)
    copies = max(1, by_line[tm_line]["VALU"] // 6)
    out = []
    out.append(f"# `walkf_step` in `{args.kernel}` — gfx950 instruction budget\n")
    out.append(f"Made by `python scripts/isa_budget.py --kernel {args.kernel} --source {args.source}" + "".join(f" -D{d}" for d in args.defs) +
               "` (hipcc of this image, the product's flags + `-gline-tables-only -S --cuda-device-only`; static counts from the assembly, "
               "each instruction attributed to the `walkf_step` line of `csrc/trace_common.h` it was inlined from through its `.loc` chain).\n")
    out.append(f"Kernel `{name}`: " + ", ".join(f"{k} {v}" for k, v in sorted(meta.items())) + f"; {sum(total.values())} instructions in all "
               f"({total['VALU']} VALU, {total['SALU']} SALU, {total['LDS']} LDS, {total['VMEM']} VMEM).\n")
    out.append(f"The kernel holds **{copies} cop{'y' if copies == 1 else 'ies'}** of the walk loop (one per cast site it is inlined into); the tables below "
               f"sum over the copies — divide by {copies} for the price of one trip.\n")
    out.append("## Per region of a trip (voxels.comp:163-246)\n")
    out.append("| region | " + " | ".join(classes) + " | what it is |")
    out.append("|---|" + "---|" * (len(classes) + 1))
    what = {"head": "iteration cap, distance check, leaf test (the two exits): every trip",
            "search": "mid-plane times, `directional`, next sibling, `has_next`, `is_child` (voxels.comp:191-204): every trip",
            "shared": "what descend and pop have in common: node size, centre from the integer path coordinates, `exit`, the new record",
            "descend": "frame store, child record load, path coordinates, slab test by selection, octant of the entry point",
            "pop": "highest level that can still advance (clz), frame load, the six plane times recomputed",
            "advance": "step to the sibling (voxels.comp:222-224)"}
    for r, cnt in by_region.items():
        out.append(f"| {r} | " + " | ".join(str(cnt[c]) for c in classes) + f" | {what[r]} |")
    trip = {k: sum(by_region[r][k] for r in ("head", "search")) for k in classes}
    out.append("")
    v = {r: by_region[r]["VALU"] // copies for r in by_region}
    out.append(f"One copy: a wave-trip executes head + search ({v['head'] + v['search']} VALU, {trip['SALU'] // copies} SALU) and then every branch at least one of "
               f"its lanes takes: advance + {v['advance']}, descend + {v['descend'] + v['shared']} (descend {v['descend']} + shared {v['shared']}), "
               f"pop + {v['pop'] + v['shared']}; all three: {sum(v.values())} VALU "
               f"(advance only: {v['head'] + v['search'] + v['advance']}; descend only: {v['head'] + v['search'] + v['shared'] + v['descend']}; "
               f"pop only: {v['head'] + v['search'] + v['shared'] + v['pop']}).\n")
    out.append("## Per source line\n")
    out.append("| line | region | VALU | SALU | LDS | VMEM | source |")
    out.append("|---|---|---|---|---|---|---|")
    for line in range(w0, w1 + 1):
        cnt = by_line.get(line)
        if cnt:
            src = common[line - 1].strip().replace("|", "\\|")
            out.append(f"| {line} | {regions[line]} | {cnt['VALU']} | {cnt['SALU']} | {cnt['LDS']} | {cnt['VMEM']} | `{src[:150]}` |")
    out.append("\n## Annotated listing (the blocks that hold `walkf_step`)\n")
    out.append("```")
    # print the basic blocks that contain at least one walk instruction, whole
    blocks, curb = [], []
    for r in rows:
        if r[1] == "label":
            if curb:
                blocks.append(curb)
            curb = [r]
        else:
            curb.append(r)
    if curb:
        blocks.append(curb)
    for b in blocks:
        if not any(r[2] for r in b):
            continue
        for t, c, region, line, where in b:
            if c == "label":
                out.append(t)
            else:
                tag = f"{region}:{line}" if region else (f"{where[0]}:{where[1]}" if where else "")
                out.append(f"    {t:<70} ; {c:<6} {tag}")
    out.append("```")
    md = "\n".join(out) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(md)
    summary = {r: dict(c) for r, c in by_region.items()}
    print(f"{args.kernel}: " + ", ".join(f"{k} {v}" for k, v in sorted(meta.items())))
    for r, c in by_region.items():
        print(f"  {r:8s} " + " ".join(f"{k} {c[k]:3d}" for k in classes))
    return summary


if __name__ == "__main__":
    main()
