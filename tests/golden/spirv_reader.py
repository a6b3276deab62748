"""A SPIR-V *reader* (no external package): word stream -> instruction list -> the facts the oracle's restatement can be checked
against.  Test infrastructure.  It reads the reference's compiled shaders as data (shaders/*.comp.spv, what
src/context/shader.rs:6-45 hands to wgpu); nothing here executes, translates or re-emits them.

What it extracts (make_spirv_fixture.py writes these into tests/golden/spirv_pins.json):
  * execution mode LocalSize, the module's float / integer constants, which GLSL.std.450 extended instructions are used;
  * for a LISTED (function, variable) pair: the operand tree of every value stored to that variable, as an S-expression over debug
    names (OpName) — i.e. the association order and the select order the compiler front end (glslang, no optimiser) recorded.
"""
import struct

MAGIC = 0x07230203

# the opcodes the three compute shaders use (SPIR-V 1.0 unified specification, section 3.32)
OPS = {
    1: "Undef", 3: "Source", 4: "SourceExtension", 5: "Name", 6: "MemberName", 11: "ExtInstImport", 12: "ExtInst", 14: "MemoryModel",
    15: "EntryPoint", 16: "ExecutionMode", 17: "Capability", 19: "TypeVoid", 20: "TypeBool", 21: "TypeInt", 22: "TypeFloat",
    23: "TypeVector", 24: "TypeMatrix", 25: "TypeImage", 26: "TypeSampler", 27: "TypeSampledImage", 28: "TypeArray",
    29: "TypeRuntimeArray", 30: "TypeStruct", 32: "TypePointer", 33: "TypeFunction", 41: "ConstantTrue", 42: "ConstantFalse",
    43: "Constant", 44: "ConstantComposite", 54: "Function", 55: "FunctionParameter", 56: "FunctionEnd", 57: "FunctionCall",
    59: "Variable", 61: "Load", 62: "Store", 65: "AccessChain", 71: "Decorate", 72: "MemberDecorate", 77: "VectorExtractDynamic",
    79: "VectorShuffle", 80: "CompositeConstruct", 81: "CompositeExtract", 82: "CompositeInsert", 84: "Transpose", 86: "SampledImage",
    87: "ImageSampleImplicitLod", 88: "ImageSampleExplicitLod", 98: "ImageRead", 99: "ImageWrite", 100: "Image", 103: "ImageQuerySizeLod",
    104: "ImageQuerySize", 109: "ConvertFToU", 110: "ConvertFToS", 111: "ConvertSToF", 112: "ConvertUToF", 124: "Bitcast",
    126: "SNegate", 127: "FNegate", 128: "IAdd", 129: "FAdd", 130: "ISub", 131: "FSub", 132: "IMul", 133: "FMul", 134: "UDiv",
    135: "SDiv", 136: "FDiv", 137: "UMod", 138: "SRem", 139: "SMod", 140: "FRem", 141: "FMod", 142: "VectorTimesScalar",
    143: "MatrixTimesScalar", 144: "VectorTimesMatrix", 145: "MatrixTimesVector", 146: "MatrixTimesMatrix", 148: "Dot",
    154: "Any", 155: "All", 156: "IsNan", 157: "IsInf", 164: "LogicalEqual", 165: "LogicalNotEqual", 166: "LogicalOr",
    167: "LogicalAnd", 168: "LogicalNot", 169: "Select", 170: "IEqual", 171: "INotEqual", 172: "UGreaterThan", 173: "SGreaterThan",
    174: "UGreaterThanEqual", 175: "SGreaterThanEqual", 176: "ULessThan", 177: "SLessThan", 178: "ULessThanEqual",
    179: "SLessThanEqual", 180: "FOrdEqual", 181: "FUnordEqual", 182: "FOrdNotEqual", 183: "FUnordNotEqual", 184: "FOrdLessThan",
    185: "FUnordLessThan", 186: "FOrdGreaterThan", 187: "FUnordGreaterThan", 188: "FOrdLessThanEqual", 189: "FUnordLessThanEqual",
    190: "FOrdGreaterThanEqual", 191: "FUnordGreaterThanEqual", 194: "ShiftRightLogical", 195: "ShiftRightArithmetic",
    196: "ShiftLeftLogical", 197: "BitwiseOr", 198: "BitwiseXor", 199: "BitwiseAnd", 200: "Not", 245: "Phi", 246: "LoopMerge",
    247: "SelectionMerge", 248: "Label", 249: "Branch", 250: "BranchConditional", 251: "Switch", 252: "Kill", 253: "Return",
    254: "ReturnValue", 255: "Unreachable",
}
# opcodes with (result type, result id) as their first two operands and value operands after them
VALUE_OPS = {n for n in OPS.values()} - {
    "Undef", "Source", "SourceExtension", "Name", "MemberName", "ExtInstImport", "MemoryModel", "EntryPoint", "ExecutionMode", "Capability",
    "TypeVoid", "TypeBool", "TypeInt", "TypeFloat", "TypeVector", "TypeMatrix", "TypeImage", "TypeSampler", "TypeSampledImage", "TypeArray",
    "TypeRuntimeArray", "TypeStruct", "TypePointer", "TypeFunction", "Function", "FunctionEnd", "Store", "Decorate", "MemberDecorate",
    "ImageWrite", "LoopMerge", "SelectionMerge", "Label", "Branch", "BranchConditional", "Switch", "Kill", "Return", "ReturnValue",
    "Unreachable"}

# GLSL.std.450 extended instruction numbers (the extended instruction set's specification, section 2)
GLSL450 = {
    1: "Round", 2: "RoundEven", 3: "Trunc", 4: "FAbs", 5: "SAbs", 6: "FSign", 7: "SSign", 8: "Floor", 9: "Ceil", 10: "Fract",
    11: "Radians", 12: "Degrees", 13: "Sin", 14: "Cos", 15: "Tan", 16: "Asin", 17: "Acos", 18: "Atan", 25: "Atan2", 26: "Pow",
    27: "Exp", 28: "Log", 29: "Exp2", 30: "Log2", 31: "Sqrt", 32: "InverseSqrt", 33: "Determinant", 34: "MatrixInverse",
    37: "FMin", 38: "UMin", 39: "SMin", 40: "FMax", 41: "UMax", 42: "SMax", 43: "FClamp", 44: "UClamp", 45: "SClamp", 46: "FMix",
    48: "Step", 49: "SmoothStep", 50: "Fma", 66: "Length", 67: "Distance", 68: "Cross", 69: "Normalize", 70: "FaceForward",
    71: "Reflect", 72: "Refract",
}


def _string(words):
    return struct.pack("<%dI" % len(words), *words).split(b"\0")[0].decode("utf-8", "replace")


class Inst:
    __slots__ = ("op", "name", "words")

    def __init__(self, op, words):
        self.op, self.name, self.words = op, OPS.get(op, f"Op{op}"), words   # words: the operands (without the first word)


class Module:
    def __init__(self, data: bytes):
        if len(data) % 4 or len(data) < 20:
            raise ValueError("not a SPIR-V word stream")
        w = struct.unpack("<%dI" % (len(data) // 4), data)
        if w[0] != MAGIC:
            raise ValueError("bad magic number")
        self.version, self.generator, self.bound = w[1], w[2], w[3]
        self.insts = []
        i = 5
        while i < len(w):
            count, op = w[i] >> 16, w[i] & 0xffff
            if count == 0 or i + count > len(w):
                raise ValueError("truncated instruction")
            self.insts.append(Inst(op, w[i + 1:i + count]))
            i += count
        self.names, self.member_names, self.types, self.consts, self.defs = {}, {}, {}, {}, {}
        self.local_size, self.ext_sets = None, {}
        for ins in self.insts:
            o = ins.words
            if ins.name == "Name":
                self.names[o[0]] = _string(o[1:])
            elif ins.name == "MemberName":
                self.member_names[(o[0], o[1])] = _string(o[2:])
            elif ins.name == "ExtInstImport":
                self.ext_sets[o[0]] = _string(o[1:])
            elif ins.name == "ExecutionMode" and o[1] == 17:      # LocalSize
                self.local_size = list(o[2:5])
            elif ins.name.startswith("Type"):
                self.types[o[0]] = ins
            elif ins.name in ("Constant", "ConstantComposite", "ConstantTrue", "ConstantFalse"):
                self.consts[o[1]] = ins
            if ins.name in VALUE_OPS and len(o) >= 2:
                self.defs[o[1]] = ins

    # ---- constants ----------------------------------------------------------------------------------------------------
    def scalar(self, cid):
        """Python value of an OpConstant (float, int) or None."""
        ins = self.consts.get(cid)
        if ins is None or ins.name != "Constant":
            return {"ConstantTrue": True, "ConstantFalse": False}.get(ins.name) if ins is not None else None
        t = self.types[ins.words[0]]
        if t.name == "TypeFloat":
            return struct.unpack("<f", struct.pack("<I", ins.words[2]))[0]
        if t.name == "TypeInt":
            v = ins.words[2]
            return v - (1 << 32) if (t.words[2] and v >= 1 << 31) else v
        return None

    def float_constants(self):
        return sorted({self.scalar(c) for c, i in self.consts.items() if i.name == "Constant" and self.types[i.words[0]].name == "TypeFloat"})

    def int_constants(self):
        return sorted({self.scalar(c) for c, i in self.consts.items() if i.name == "Constant" and self.types[i.words[0]].name == "TypeInt"})

    def ext_inst_counts(self):
        out = {}
        for ins in self.insts:
            if ins.name == "ExtInst" and self.ext_sets.get(ins.words[2], "").startswith("GLSL.std.450"):
                n = GLSL450.get(ins.words[3], f"ext{ins.words[3]}")
                out[n] = out.get(n, 0) + 1
        return dict(sorted(out.items()))

    # ---- functions and expression trees -----------------------------------------------------------------------------------
    def functions(self):
        """{name: [instructions of the body]}"""
        out, cur, name = {}, None, None
        for ins in self.insts:
            if ins.name == "Function":
                name, cur = self.names.get(ins.words[1], f"fn{ins.words[1]}"), []
            elif ins.name == "FunctionEnd":
                out[name] = cur
                cur = None
            elif cur is not None:
                cur.append(ins)
        return out

    def _pointer(self, pid, temps):
        """A pointer operand as text: a named variable, or base.member / base[index] for an access chain."""
        ins = self.defs.get(pid)
        if ins is None:
            return self.names.get(pid, f"%{pid}")
        if ins.name == "Variable" or ins.name == "FunctionParameter":
            return self.names.get(pid) or f"%{pid}"
        if ins.name == "AccessChain":
            base = self._pointer(ins.words[2], temps)
            base_type = self._pointee_type(ins.words[2])
            for idx in ins.words[3:]:
                v = self.scalar(idx)
                member = None
                if base_type is not None and base_type.name == "TypeStruct" and v is not None:
                    member = self.member_names.get((base_type.words[0], v))
                    base_type = self.types.get(base_type.words[1 + v])
                elif base_type is not None and base_type.name in ("TypeVector", "TypeArray", "TypeRuntimeArray", "TypeMatrix"):
                    base_type = self.types.get(base_type.words[1])
                base += f".{member}" if member else f"[{v if v is not None else self.tree(idx, temps)}]"
            return base
        return f"%{pid}"

    def _pointee_type(self, pid):
        ins = self.defs.get(pid)
        if ins is None:
            return None
        ptr = self.types.get(ins.words[0])
        return self.types.get(ptr.words[2]) if ptr is not None and ptr.name == "TypePointer" else None

    def tree(self, vid, temps=None, depth=0):
        """The value `vid` as an S-expression over names, constants and opcode names."""
        temps = temps or {}
        if depth > 60:
            return "..."
        if vid in self.consts:
            ins = self.consts[vid]
            if ins.name == "ConstantComposite":
                return "(" + " ".join(["vec"] + [self.tree(c, temps, depth + 1) for c in ins.words[2:]]) + ")"
            v = self.scalar(vid)
            return repr(v) if not isinstance(v, float) else f"{v:.9g}"
        ins = self.defs.get(vid)
        if ins is None:
            return self.names.get(vid, f"%{vid}")
        o = ins.words
        if ins.name == "Load":
            ptr = o[2]
            if ptr in temps:                       # a compiler temporary (function-call argument copy): what was stored into it
                return temps[ptr]
            return self._pointer(ptr, temps)
        if ins.name == "ExtInst":
            n = GLSL450.get(o[3], f"ext{o[3]}")
            return "(" + " ".join([n] + [self.tree(a, temps, depth + 1) for a in o[4:]]) + ")"
        if ins.name == "FunctionCall":
            args = [temps.get(a) or self._pointer(a, temps) for a in o[3:]]
            return "(" + " ".join(["call:" + self.names.get(o[2], f"fn{o[2]}").split("(")[0]] + args) + ")"
        if ins.name == "CompositeExtract":
            return "(" + " ".join(["extract", self.tree(o[2], temps, depth + 1)] + [str(i) for i in o[3:]]) + ")"
        if ins.name == "VectorShuffle":
            return "(" + " ".join(["shuffle", self.tree(o[2], temps, depth + 1), self.tree(o[3], temps, depth + 1)] + [str(i) for i in o[4:]]) + ")"
        if ins.name in ("Variable", "FunctionParameter", "AccessChain"):
            return self._pointer(vid, temps)
        return "(" + " ".join([ins.name] + [self.tree(a, temps, depth + 1) for a in o[2:]]) + ")"

    def stores(self, function):
        """[(pointer text, value tree)] of every OpStore in `function`, in order; stores into unnamed / `param` temporaries (glslang's
        copies of function-call arguments) are folded into the trees that read them."""
        body = self.functions()[function]
        temps, out = {}, []
        for ins in body:
            if ins.name != "Store":
                continue
            ptr, val = ins.words[0], ins.words[1]
            d = self.defs.get(ptr)
            text = self.tree(val, temps)
            if d is not None and d.name == "Variable" and (self.names.get(ptr, "") == "" or self.names.get(ptr, "").startswith("param")):
                temps[ptr] = text
                continue
            out.append((self._pointer(ptr, temps), text))
        return out

    def returns(self, function):
        """Value trees of the function's OpReturnValue instructions, in order."""
        body = self.functions()[function]
        temps = {}
        out = []
        for ins in body:
            if ins.name == "Store":
                d = self.defs.get(ins.words[0])
                if d is not None and d.name == "Variable" and (self.names.get(ins.words[0], "") == "" or self.names.get(ins.words[0], "").startswith("param")):
                    temps[ins.words[0]] = self.tree(ins.words[1], temps)
            elif ins.name == "ReturnValue":
                out.append(self.tree(ins.words[0], temps))
        return out


def load(path):
    with open(path, "rb") as f:
        return Module(f.read())
