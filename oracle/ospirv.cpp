// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// ospirv.cpp: an INTERPRETER for the three compute shaders the reference dispatches, taken in the form it actually hands to the GPU —
// shaders/{voxels,temporal,denoise}.comp.spv (SPIR-V 1.0 from glslang, unoptimised; loaded by src/context/shader.rs:6-45).  It turns
// "the oracle restates the shaders" from a reading of GLSL text into an executed comparison: the same inputs through (a) the compiled
// reference shader, instruction by instruction, and (b) oracle/oshaders.cpp must give the same bits (tests/test_oracle_spirv_exec.py),
// and the outputs of (a) are what tests/golden/spirv_exec/ holds for the GPU box, where the reference does not exist.
//
// What is executed FROM THE MODULE: every core instruction — control flow (branches, loops, phis, calls, returns), loads and stores
// through access chains with the module's own Offset / ArrayStride decorations, integer and bit arithmetic, conversions, comparisons,
// selects, composite shuffles, and the IEEE binary32 + - * / in the order the module gives them (this file is compiled with
// -ffp-contract=off, no fast-math).  Nothing of the reference runs natively: the module is data to this evaluator, every memory access is
// bounds-checked against the buffers the caller bound, and a trip budget ends a runaway loop.
//
// What SPIR-V leaves to the implementation is BOUND to the choices the oracle documents (oracle/oshaders.cpp header) —
// the same functions, called from here:
//   U2  GLSL.std.450 Pow whose exponent is the CONSTANT 2.0 -> x * x;
//   U4  OpImageSampleExplicitLod -> orc::Tex::sample (bilinear, 8 fractional weight bits, clamp to edge);
//   U5  MatrixInverse -> orc::affine_inverse on the columns (R,0) (U,0) (F,0) (O,1) (binary64 adjugate, rounded once);
//   U6  Sin Cos Pow Exp Log Sqrt -> include/vxrt_detmath.h; Normalize = v / sqrt(dot); Length, Distance, Cross, Reflect, FMix, FClamp,
//       FMin, FMax, FSign, FAbs as in oracle/ovec.h;
//   U7  OpConvertFToS / OpConvertFToU -> saturating (vx_f2i);
//   U8  the association of OpDot and OpMatrixTimesVector (SPIR-V fixes no order): left to right, ((x + y) + z) + w, as ovec.h's dot;
//   U9  contraction: the modules carry no NoContraction decoration, so a driver may fuse a * b + c; here, as in the oracle and the
//       kernels, never (tests/test_oracle_builtin_sensitivity.py measures what fusing moves).
// Undefined reads (U1: a Function variable read before it is written): memory is zeroed when an invocation starts and persists across
// calls; with flag ORC_SPV_POISON every Function variable is filled with a NaN pattern at each function entry instead — outputs that do
// not change between the two modes do not depend on an undefined read.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "oracle.h"

namespace {
using namespace orc;

enum Kind : uint8_t { K_NONE, K_VOID, K_BOOL, K_INT, K_FLOAT, K_VECTOR, K_MATRIX, K_ARRAY, K_RUNTIME_ARRAY, K_STRUCT, K_POINTER, K_IMAGE,
                      K_SAMPLER, K_SAMPLED_IMAGE, K_FUNCTION };

struct Type {
    Kind kind = K_NONE;
    bool is_signed = false;
    uint32_t elem = 0;       // vector / matrix column / array element / pointee type
    uint32_t count = 0;      // vector size, matrix columns, array length
    uint32_t storage = 0;    // pointer: storage class
    uint32_t sampled = 0;    // image: 1 sampled, 2 storage
    std::vector<uint32_t> members;
    uint32_t array_stride = 0;
    std::vector<uint32_t> member_offset;   // decorated (Offset); 0xffffffff = none
    uint32_t words = 0;      // flattened scalar count (0 for runtime arrays, images, ...)
    uint32_t packed = 0;     // bytes in this interpreter's own layout (Function / Private memory)
};

struct Ptr { uint8_t* p = nullptr; uint8_t* lo = nullptr; uint8_t* hi = nullptr; uint32_t type = 0; bool deco = false; };

struct Val {
    union { uint32_t u[16]; float f[16]; int32_t i[16]; };
    Ptr ptr;
    int img = -1;            // index into the bindings for image / sampler / sampled image values
    Val() { memset(u, 0, sizeof u); }
};

struct Inst { uint16_t op; uint32_t first; uint16_t n; };   // operands: words[first .. first + n)

struct Function { uint32_t id = 0; size_t entry = 0; std::vector<uint32_t> params; std::vector<uint32_t> vars; };

struct Global { uint32_t id, storage, type; int binding_index = -1; uint32_t builtin = 0xffffffffu; size_t scratch = 0; };

struct Module {
    std::vector<uint32_t> words;
    std::vector<Inst> insts;
    uint32_t bound = 0;
    std::vector<Type> types;
    std::vector<uint8_t> is_const;
    std::vector<Val> consts;
    std::vector<size_t> label_pc;
    std::vector<int> function_of;            // id -> index into functions
    std::vector<Function> functions;
    std::vector<Global> globals;
    std::vector<size_t> var_scratch;          // Function / Private variable id -> offset in a thread's scratch
    std::vector<uint32_t> var_init;           // variable id -> initializer id (0 = none)
    std::vector<uint32_t> set_of, binding_of, builtin_of;
    std::vector<uint32_t> type_of;            // result id -> result type id (0: not a value)
    uint32_t glsl_set = 0, entry = 0;
    size_t scratch_bytes = 0;
    std::string error;
};

struct Binding { uint32_t set, binding, kind; uint8_t* data; uint64_t bytes; uint32_t width, height, ox, oy, cw, ch; };

constexpr uint32_t kNone = 0xffffffffu;

bool fail(Module& m, const std::string& why) { if (m.error.empty()) m.error = why; return false; }

// ---- module set-up --------------------------------------------------------------------------------------------------------------
bool layout_type(Module& m, uint32_t id) {
    Type& t = m.types[id];
    switch (t.kind) {
        case K_BOOL: case K_INT: case K_FLOAT: t.words = 1; t.packed = 4; break;
        case K_VECTOR: case K_MATRIX: case K_ARRAY:
            t.words = m.types[t.elem].words * t.count; t.packed = m.types[t.elem].packed * t.count; break;
        case K_STRUCT:
            t.words = 0; t.packed = 0;
            for (uint32_t mem : t.members) { t.words += m.types[mem].words; t.packed += m.types[mem].packed; }
            break;
        default: t.words = 0; t.packed = 0; break;
    }
    return true;
}

// opcodes whose first two operands are (result type, result id)
bool has_result_type(unsigned op) {
    return op == 1 || op == 12 || (op >= 41 && op <= 44) || op == 54 || op == 55 || op == 57 || op == 59 || op == 61 || op == 65 || op == 77 ||
           (op >= 79 && op <= 84) || (op >= 86 && op <= 88) || op == 98 || op == 100 || op == 103 || op == 104 || (op >= 109 && op <= 112) ||
           op == 124 || (op >= 126 && op <= 148) || (op >= 154 && op <= 157) || (op >= 164 && op <= 200) || op == 245;
}

bool parse(Module& m, const uint32_t* w, size_t n) {
    if (n < 5 || w[0] != 0x07230203u) return fail(m, "not a SPIR-V module");
    m.words.assign(w, w + n);
    m.bound = w[3];
    if (m.bound == 0 || m.bound > (1u << 16)) return fail(m, "unreasonable id bound");
    m.types.assign(m.bound, Type());
    m.is_const.assign(m.bound, 0);
    m.consts.assign(m.bound, Val());
    m.label_pc.assign(m.bound, size_t(-1));
    m.function_of.assign(m.bound, -1);
    m.var_scratch.assign(m.bound, size_t(-1));
    m.var_init.assign(m.bound, 0);
    m.set_of.assign(m.bound, kNone); m.binding_of.assign(m.bound, kNone); m.builtin_of.assign(m.bound, kNone);
    m.type_of.assign(m.bound, 0);
    for (size_t i = 5; i < n;) {
        const uint32_t count = w[i] >> 16, op = w[i] & 0xffffu;
        if (count == 0 || i + count > n) return fail(m, "truncated instruction");
        m.insts.push_back(Inst{uint16_t(op), uint32_t(i + 1), uint16_t(count - 1)});
        i += count;
    }
    auto idok = [&](uint32_t id) { return id < m.bound; };
    int cur_fn = -1;
    for (size_t pc = 0; pc < m.insts.size(); pc++) {
        const Inst& in = m.insts[pc];
        const uint32_t* o = &m.words[in.first];
        auto need = [&](unsigned k) { return in.n >= k; };
        if (has_result_type(in.op)) {
            if (!need(2) || !idok(o[0]) || !idok(o[1])) return fail(m, "malformed value instruction");
            m.type_of[o[1]] = o[0];
        }
        switch (in.op) {
            case 11: {  // ExtInstImport
                if (!need(2) || !idok(o[0])) return fail(m, "bad ExtInstImport");
                const char* s = reinterpret_cast<const char*>(o + 1);
                // the name is bounded by the INSTRUCTION (in.n words, the first of them the result id), not by a NUL that a truncated
                // module may lack (strncmp read up to 8 bytes past a one-word name at the module's end)
                const size_t avail = (size_t(in.n) - 1) * 4;
                if (avail >= 12 && memcmp(s, "GLSL.std.450", 12) == 0) m.glsl_set = o[0];
                break;
            }
            case 15: if (!need(2) || !idok(o[1])) return fail(m, "bad EntryPoint"); if (o[0] == 5) m.entry = o[1]; break;   // GLCompute
            case 71: {  // Decorate
                if (!need(2) || !idok(o[0])) return fail(m, "bad Decorate");
                if (o[1] == 33 && need(3)) m.binding_of[o[0]] = o[2];
                if (o[1] == 34 && need(3)) m.set_of[o[0]] = o[2];
                if (o[1] == 11 && need(3)) m.builtin_of[o[0]] = o[2];
                if (o[1] == 6 && need(3)) m.types[o[0]].array_stride = o[2];
                break;
            }
            case 72: {  // MemberDecorate
                if (!need(3) || !idok(o[0]) || o[1] >= 1024u) return fail(m, "bad MemberDecorate");
                if (o[2] == 35 && need(4)) {
                    Type& t = m.types[o[0]];
                    if (t.member_offset.size() <= o[1]) t.member_offset.resize(o[1] + 1, kNone);
                    t.member_offset[o[1]] = o[3];
                }
                break;
            }
            case 19: case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 32: case 33: {
                if (!need(1) || !idok(o[0])) return fail(m, "bad type");
                if (m.types[o[0]].kind != K_NONE) return fail(m, "type id defined twice");
                Type& t = m.types[o[0]];
                const std::vector<uint32_t> keep_off = t.member_offset;   // decorations come before the types
                const uint32_t keep_stride = t.array_stride;
                t = Type(); t.member_offset = keep_off; t.array_stride = keep_stride;
                switch (in.op) {
                    case 19: t.kind = K_VOID; break;
                    case 20: t.kind = K_BOOL; break;
                    case 21: if (!need(3) || o[1] != 32) return fail(m, "only 32-bit integers"); t.kind = K_INT; t.is_signed = o[2] != 0; break;
                    case 22: if (!need(2) || o[1] != 32) return fail(m, "only 32-bit floats"); t.kind = K_FLOAT; break;
                    // a type refers to types DEFINED BEFORE it (no forward references in these modules): the type graph has no cycles
                    case 23: {
                        if (!need(3) || !idok(o[1]) || o[1] == o[0] || o[2] < 2 || o[2] > 4) return fail(m, "bad vector");
                        const Kind ek = m.types[o[1]].kind;
                        if (ek != K_BOOL && ek != K_INT && ek != K_FLOAT) return fail(m, "vector of a non-scalar");
                        t.kind = K_VECTOR; t.elem = o[1]; t.count = o[2];
                        break;
                    }
                    case 24:
                        if (!need(3) || !idok(o[1]) || o[1] == o[0] || o[2] < 2 || o[2] > 4 || m.types[o[1]].kind != K_VECTOR) return fail(m, "bad matrix");
                        t.kind = K_MATRIX; t.elem = o[1]; t.count = o[2];
                        break;
                    case 25: if (!need(8)) return fail(m, "bad image"); t.kind = K_IMAGE; t.sampled = o[6]; break;
                    case 26: t.kind = K_SAMPLER; break;
                    case 27: t.kind = K_SAMPLED_IMAGE; break;
                    case 28: {
                        if (!need(3) || !idok(o[1]) || o[1] == o[0] || !idok(o[2]) || !m.is_const[o[2]] || m.types[o[1]].kind == K_NONE) return fail(m, "bad array");
                        t.kind = K_ARRAY; t.elem = o[1]; t.count = m.consts[o[2]].u[0];
                        if (t.count == 0 || t.count > (1u << 26)) return fail(m, "bad array length");
                        break;
                    }
                    case 29: if (!need(2) || !idok(o[1]) || o[1] == o[0] || m.types[o[1]].kind == K_NONE) return fail(m, "bad runtime array"); t.kind = K_RUNTIME_ARRAY; t.elem = o[1]; break;
                    case 30:
                        t.kind = K_STRUCT;
                        for (unsigned k = 1; k < in.n; k++) { if (!idok(o[k]) || o[k] == o[0] || m.types[o[k]].kind == K_NONE) return fail(m, "bad struct"); t.members.push_back(o[k]); }
                        t.member_offset.resize(t.members.size(), kNone);
                        break;
                    case 32: if (!need(3) || !idok(o[2]) || o[2] == o[0] || m.types[o[2]].kind == K_NONE) return fail(m, "bad pointer"); t.kind = K_POINTER; t.storage = o[1]; t.elem = o[2]; break;
                    case 33: t.kind = K_FUNCTION; break;
                }
                layout_type(m, o[0]);
                break;
            }
            case 41: case 42: case 43: case 44: case 1: {   // ConstantTrue / False / Constant / ConstantComposite / Undef
                if (!need(2) || !idok(o[0]) || !idok(o[1])) return fail(m, "bad constant");
                Val v;
                const Type& t = m.types[o[0]];
                if (t.words > 16) return fail(m, "constant wider than 16 words");
                if (in.op == 41) v.u[0] = 1;
                if (in.op == 43) { if (!need(3)) return fail(m, "bad constant"); v.u[0] = o[2]; }
                if (in.op == 44) {
                    unsigned at = 0;
                    for (unsigned k = 2; k < in.n; k++) {
                        if (!idok(o[k]) || !m.is_const[o[k]]) return fail(m, "constant composite of a non-constant");
                        const unsigned ew = m.types[m.type_of[o[k]]].words;
                        if (at + ew > 16) return fail(m, "constant wider than 16 words");
                        memcpy(v.u + at, m.consts[o[k]].u, 4 * ew);
                        at += ew;
                    }
                }
                m.consts[o[1]] = v;
                m.is_const[o[1]] = 1;
                break;
            }
            case 54: {  // Function
                if (!need(4) || !idok(o[1])) return fail(m, "bad Function");
                m.function_of[o[1]] = int(m.functions.size());
                Function f; f.id = o[1]; f.entry = pc + 1;
                m.functions.push_back(f);
                cur_fn = int(m.functions.size()) - 1;
                break;
            }
            case 55: if (cur_fn < 0 || !need(2) || !idok(o[1])) return fail(m, "bad FunctionParameter"); m.functions[cur_fn].params.push_back(o[1]); m.functions[cur_fn].entry = pc + 1; break;
            case 56: cur_fn = -1; break;
            case 248: if (!need(1) || !idok(o[0])) return fail(m, "bad Label"); m.label_pc[o[0]] = pc; break;
            case 59: {  // Variable
                if (!need(3) || !idok(o[0]) || !idok(o[1])) return fail(m, "bad Variable");
                const Type& pt = m.types[o[0]];
                if (pt.kind != K_POINTER) return fail(m, "variable of a non-pointer type");
                if (in.n >= 4) { if (!idok(o[3])) return fail(m, "bad initializer"); m.var_init[o[1]] = o[3]; }
                if (o[2] == 7 || o[2] == 6 || o[2] == 1) {   // Function, Private, Input: this interpreter's own memory
                    const uint32_t bytes = m.types[pt.elem].packed;
                    if (bytes == 0) return fail(m, "variable of an unsized type");
                    m.var_scratch[o[1]] = m.scratch_bytes;
                    m.scratch_bytes += (bytes + 15u) & ~15u;
                    if (o[2] == 7) { if (cur_fn < 0) return fail(m, "Function variable outside a function"); m.functions[cur_fn].vars.push_back(o[1]); }
                }
                if (o[2] != 7) m.globals.push_back(Global{o[1], o[2], pt.elem, -1, kNone, 0});
                break;
            }
            default: break;
        }
    }
    if (m.entry == 0 || m.function_of[m.entry] < 0) return fail(m, "no GLCompute entry point");
    if (m.scratch_bytes > (1u << 20)) return fail(m, "too much private memory");
    return true;
}

// ---- one invocation -----------------------------------------------------------------------------------------------------------------
struct Thread {
    Module& m;
    const std::vector<Binding>& bindings;
    std::vector<Val> vals;
    std::vector<uint8_t> scratch;
    uint64_t budget = 0, executed = 0;
    uint32_t flags = 0;
    int depth = 0;
    std::string error;

    Thread(Module& mod, const std::vector<Binding>& b) : m(mod), bindings(b), vals(mod.bound), scratch(mod.scratch_bytes + 16) {}

    bool fail(const std::string& why) { if (error.empty()) error = why; return false; }
    const uint32_t* ops(const Inst& in) const { return &m.words[in.first]; }

    // a value operand: constants live in the module, everything else in this thread
    const Val& get(uint32_t id) { return m.is_const[id] ? m.consts[id] : vals[id]; }

    bool load_rec(uint32_t type, const Ptr& p, uint8_t* addr, uint32_t* out, unsigned& at) {
        const Type& t = m.types[type];
        switch (t.kind) {
            case K_BOOL: case K_INT: case K_FLOAT:
                if (addr < p.lo || addr + 4 > p.hi) return fail("load outside the bound memory");
                if (at >= 16) return fail("value wider than 16 words");
                memcpy(out + at++, addr, 4);
                return true;
            case K_VECTOR:
                for (uint32_t k = 0; k < t.count; k++) if (!load_rec(t.elem, p, addr + 4 * k, out, at)) return false;
                return true;
            case K_MATRIX: case K_ARRAY: {
                const uint32_t stride = (p.deco && t.array_stride) ? t.array_stride : m.types[t.elem].packed;
                for (uint32_t k = 0; k < t.count; k++) if (!load_rec(t.elem, p, addr + size_t(stride) * k, out, at)) return false;
                return true;
            }
            case K_STRUCT: {
                uint32_t off = 0;
                for (size_t k = 0; k < t.members.size(); k++) {
                    const uint32_t mo = (p.deco && t.member_offset[k] != kNone) ? t.member_offset[k] : off;
                    if (!load_rec(t.members[k], p, addr + mo, out, at)) return false;
                    off += m.types[t.members[k]].packed;
                }
                return true;
            }
            default: return fail("load of an unsupported type");
        }
    }
    bool store_rec(uint32_t type, const Ptr& p, uint8_t* addr, const uint32_t* in, unsigned& at) {
        const Type& t = m.types[type];
        switch (t.kind) {
            case K_BOOL: case K_INT: case K_FLOAT:
                if (p.deco) return fail("store to a read-only buffer");
                if (addr < p.lo || addr + 4 > p.hi) return fail("store outside the bound memory");
                if (at >= 16) return fail("value wider than 16 words");
                memcpy(addr, in + at++, 4);
                return true;
            case K_VECTOR:
                for (uint32_t k = 0; k < t.count; k++) if (!store_rec(t.elem, p, addr + 4 * k, in, at)) return false;
                return true;
            case K_MATRIX: case K_ARRAY: {
                const uint32_t stride = m.types[t.elem].packed;
                for (uint32_t k = 0; k < t.count; k++) if (!store_rec(t.elem, p, addr + size_t(stride) * k, in, at)) return false;
                return true;
            }
            case K_STRUCT: {
                uint32_t off = 0;
                for (size_t k = 0; k < t.members.size(); k++) {
                    if (!store_rec(t.members[k], p, addr + off, in, at)) return false;
                    off += m.types[t.members[k]].packed;
                }
                return true;
            }
            default: return fail("store of an unsupported type");
        }
    }

    // flattened word offset and type of a composite's constituent
    bool constituent(uint32_t& type, unsigned& word, uint32_t index) {
        const Type& t = m.types[type];
        switch (t.kind) {
            case K_VECTOR: case K_MATRIX: case K_ARRAY:
                if (index >= t.count) return fail("composite index out of range");
                word += m.types[t.elem].words * index; type = t.elem; return true;
            case K_STRUCT:
                if (index >= t.members.size()) return fail("member index out of range");
                for (uint32_t k = 0; k < index; k++) word += m.types[t.members[k]].words;
                type = t.members[index]; return true;
            default: return fail("constituent of a scalar");
        }
    }

    uint32_t operand_type(uint32_t id) const { return m.type_of[id]; }
    unsigned operand_words(uint32_t id) const { return m.types[m.type_of[id]].words; }

    const Binding* image_of(const Val& v) {
        if (v.img < 0 || size_t(v.img) >= bindings.size()) { fail("image operand without a binding"); return nullptr; }
        return &bindings[size_t(v.img)];
    }
    bool texel_address(const Binding& b, int x, int y, float** out) {
        if (x < 0 || y < 0 || uint32_t(x) >= b.width || uint32_t(y) >= b.height) return fail("image coordinate outside the image");
        if (uint32_t(x) < b.ox || uint32_t(y) < b.oy || uint32_t(x) - b.ox >= b.cw || uint32_t(y) - b.oy >= b.ch)
            return fail("image coordinate outside the window the caller bound");
        *out = reinterpret_cast<float*>(b.data) + 4 * (size_t(uint32_t(y) - b.oy) * b.cw + (uint32_t(x) - b.ox));
        return true;
    }

    bool ext_inst(const Inst& in, const uint32_t* o, Val& r, const Type& rt);
    bool call(int fn_index, Val* result);
    bool run(uint32_t gx, uint32_t gy);
};

float f_min(float a, float b) { return vx_min(a, b); }
float f_max(float a, float b) { return vx_max(a, b); }

bool Thread::ext_inst(const Inst& in, const uint32_t* o, Val& r, const Type& rt) {
    const uint32_t which = o[3];
    const unsigned n = rt.words;
    auto arg = [&](unsigned k) -> const Val& { return get(o[4 + k]); };
    const unsigned nargs = in.n - 4;
    auto want = [&](unsigned k) { return nargs == k ? true : fail("extended instruction with an unexpected operand count"); };
    auto as3 = [](const Val& v) { return v3(v.f[0], v.f[1], v.f[2]); };
    auto put3 = [&](V3 v) { r.f[0] = v.x; r.f[1] = v.y; r.f[2] = v.z; };
    switch (which) {
        case 4: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_abs(arg(0).f[k]); return true;             // FAbs
        case 6: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_sign(arg(0).f[k]); return true;            // FSign
        case 13: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_sin(arg(0).f[k]); return true;
        case 14: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_cos(arg(0).f[k]); return true;
        case 26: {  // Pow; U2: a constant exponent of 2 is a product
            if (!want(2)) return false;
            const uint32_t e = o[5];
            bool square = m.is_const[e] != 0;
            for (unsigned k = 0; square && k < n; k++) square = m.consts[e].f[k] == 2.0f;
            for (unsigned k = 0; k < n; k++) r.f[k] = square ? arg(0).f[k] * arg(0).f[k] : vx_pow(arg(0).f[k], arg(1).f[k]);
            return true;
        }
        case 27: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_exp(arg(0).f[k]); return true;
        case 28: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_log(arg(0).f[k]); return true;
        case 31: if (!want(1)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_sqrt(arg(0).f[k]); return true;
        case 34: {  // MatrixInverse (U5): only the matrix temporal.comp:75-82 builds has a defined meaning here
            if (!want(1) || n != 16) return fail("MatrixInverse of something other than a mat4");
            const float* c = arg(0).f;   // column-major
            float inv[12];
            affine_inverse(c + 0, c + 4, c + 8, c + 12, inv);
            for (int col = 0; col < 4; col++) {
                for (int row = 0; row < 3; row++) r.f[4 * col + row] = inv[4 * row + col];
                r.f[4 * col + 3] = col == 3 ? 1.0f : 0.0f;
            }
            return true;
        }
        case 37: if (!want(2)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = f_min(arg(0).f[k], arg(1).f[k]); return true;
        case 40: if (!want(2)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = f_max(arg(0).f[k], arg(1).f[k]); return true;
        case 43: if (!want(3)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_clamp(arg(0).f[k], arg(1).f[k], arg(2).f[k]); return true;
        case 46: if (!want(3)) return false; for (unsigned k = 0; k < n; k++) r.f[k] = vx_mix(arg(0).f[k], arg(1).f[k], arg(2).f[k]); return true;
        case 66: if (!want(1)) return false; r.f[0] = length(as3(arg(0))); return true;                                               // Length (vec3)
        case 67: if (!want(2)) return false; r.f[0] = length(as3(arg(0)) - as3(arg(1))); return true;                                  // Distance
        case 68: if (!want(2) || n != 3) return fail("Cross of non-vec3"); put3(cross(as3(arg(0)), as3(arg(1)))); return true;
        case 69: if (!want(1) || n != 3) return fail("Normalize of non-vec3"); put3(normalize(as3(arg(0)))); return true;
        case 71: if (!want(2) || n != 3) return fail("Reflect of non-vec3"); put3(reflect(as3(arg(0)), as3(arg(1)))); return true;
        default: return fail("GLSL.std.450 instruction " + std::to_string(which) + " is not one the three shaders use");
    }
}

bool Thread::call(int fn_index, Val* result) {
    if (++depth > 32) return fail("call depth");
    const Function& fn = m.functions[size_t(fn_index)];
    if (flags & 1u) {   // ORC_SPV_POISON
        for (uint32_t v : fn.vars) {
            const uint32_t bytes = m.types[m.types[m.type_of[v]].elem].packed;
            uint8_t* p = scratch.data() + m.var_scratch[v];
            for (uint32_t k = 0; k + 4 <= bytes; k += 4) { const uint32_t poison = 0x7fc0deadu; memcpy(p + k, &poison, 4); }
        }
    }
    size_t pc = fn.entry;
    uint32_t cur_label = 0, prev_label = 0;
    for (;;) {
        if (pc >= m.insts.size()) return fail("ran off the end of the module");
        if (++executed > budget) return fail("instruction budget exhausted (runaway loop?)");
        const Inst& in = m.insts[pc++];
        const uint32_t* o = ops(in);
        auto idok = [&](uint32_t id) { return id < m.bound; };
        // result type / id of value instructions
        const uint32_t rtype = in.n >= 2 ? o[0] : 0, rid = in.n >= 2 ? o[1] : 0;
        auto check_value = [&](unsigned operands) {
            if (in.n < 2 + operands || !idok(rtype) || !idok(rid)) return fail("malformed value instruction");
            for (unsigned k = 0; k < operands; k++) if (!idok(o[2 + k])) return fail("operand id out of range");
            if (m.types[rtype].words > 16) return fail("result wider than 16 words");
            return true;
        };
        switch (in.op) {
            case 248: if (!idok(o[0])) return fail("bad label"); prev_label = cur_label; cur_label = o[0]; break;
            case 246: case 247: case 8: case 317: break;    // LoopMerge, SelectionMerge, Line, NoLine
            case 249: if (in.n < 1 || !idok(o[0]) || m.label_pc[o[0]] == size_t(-1)) return fail("bad branch"); pc = m.label_pc[o[0]]; break;
            case 250: {
                if (in.n < 3 || !idok(o[0]) || !idok(o[1]) || !idok(o[2])) return fail("bad conditional branch");
                const uint32_t t = (get(o[0]).u[0] & 1u) ? o[1] : o[2];
                if (m.label_pc[t] == size_t(-1)) return fail("branch to a non-label");
                pc = m.label_pc[t];
                break;
            }
            case 251: {  // Switch
                if (in.n < 2 || !idok(o[0]) || !idok(o[1])) return fail("bad switch");
                uint32_t t = o[1];
                for (unsigned k = 2; k + 1 < in.n; k += 2) if (get(o[0]).u[0] == o[k]) { t = o[k + 1]; break; }
                if (!idok(t) || m.label_pc[t] == size_t(-1)) return fail("switch to a non-label");
                pc = m.label_pc[t];
                break;
            }
            case 253: depth--; return true;                                     // Return
            case 254: if (in.n < 1 || !idok(o[0])) return fail("bad return"); if (result) *result = get(o[0]); depth--; return true;
            case 255: return fail("OpUnreachable reached");
            case 252: return fail("OpKill in a compute shader");
            case 56: return fail("function without a return");
            case 59: {  // Variable (Function storage): its memory is static (GLSL has no recursion)
                if (in.n < 3 || !idok(o[0]) || !idok(o[1]) || m.var_scratch[o[1]] == size_t(-1)) return fail("bad variable");
                const uint32_t pointee = m.types[o[0]].elem;
                Val& v = vals[o[1]];
                uint8_t* p = scratch.data() + m.var_scratch[o[1]];
                v.ptr = Ptr{p, p, p + m.types[pointee].packed, pointee, false};
                if (m.var_init[o[1]]) { unsigned at = 0; if (!store_rec(pointee, v.ptr, p, get(m.var_init[o[1]]).u, at)) return false; }
                break;
            }
            case 61: {  // Load
                if (!check_value(1)) return false;
                const Val& src = get(o[2]);
                Val& r = vals[rid];
                const Type& t = m.types[rtype];
                if (t.kind == K_IMAGE || t.kind == K_SAMPLER || t.kind == K_SAMPLED_IMAGE) { r.img = src.img; break; }
                if (src.ptr.p == nullptr) return fail("load through a null pointer");
                Val tmp; unsigned at = 0;
                if (!load_rec(rtype, src.ptr, src.ptr.p, tmp.u, at)) return false;
                memcpy(r.u, tmp.u, sizeof r.u);
                break;
            }
            case 62: {  // Store
                if (in.n < 2 || !idok(o[0]) || !idok(o[1])) return fail("bad store");
                const Val& dst = get(o[0]);
                if (dst.ptr.p == nullptr) return fail("store through a null pointer");
                unsigned at = 0;
                Val tmp = get(o[1]);
                if (!store_rec(dst.ptr.type, dst.ptr, dst.ptr.p, tmp.u, at)) return false;
                break;
            }
            case 65: {  // AccessChain
                if (!check_value(1)) return false;
                Ptr p = get(o[2]).ptr;
                if (p.p == nullptr) return fail("access chain on a null pointer");
                for (unsigned k = 3; k < in.n; k++) {
                    if (!idok(o[k])) return fail("bad index");
                    const uint32_t idx = get(o[k]).u[0];
                    const Type& t = m.types[p.type];
                    switch (t.kind) {
                        case K_STRUCT: {
                            if (!m.is_const[o[k]] || idx >= t.members.size()) return fail("bad member index");
                            uint32_t off = 0;
                            if (p.deco && t.member_offset[idx] != kNone) off = t.member_offset[idx];
                            else for (uint32_t j = 0; j < idx; j++) off += m.types[t.members[j]].packed;
                            p.p += off; p.type = t.members[idx];
                            break;
                        }
                        case K_ARRAY: case K_MATRIX: case K_RUNTIME_ARRAY: {
                            if (t.kind != K_RUNTIME_ARRAY && idx >= t.count) return fail("array index " + std::to_string(idx) + " out of range (" + std::to_string(t.count) + ")");
                            const uint32_t stride = (p.deco && t.array_stride) ? t.array_stride : m.types[t.elem].packed;
                            if (stride == 0) return fail("array without a stride");
                            if (uint64_t(idx) * stride > uint64_t(p.hi - p.lo)) return fail("index " + std::to_string(idx) + " beyond the bound buffer");
                            p.p += size_t(idx) * stride; p.type = t.elem;
                            break;
                        }
                        case K_VECTOR:
                            if (idx >= t.count) return fail("vector index out of range");
                            p.p += 4 * idx; p.type = t.elem;
                            break;
                        default: return fail("access chain into a scalar");
                    }
                }
                vals[rid].ptr = p;
                break;
            }
            case 57: {  // FunctionCall
                if (!check_value(1) || m.function_of[o[2]] < 0) return fail("call of a non-function");
                const int callee = m.function_of[o[2]];
                const Function& cf = m.functions[size_t(callee)];
                if (cf.params.size() != size_t(in.n - 3)) return fail("argument count");
                for (size_t k = 0; k < cf.params.size(); k++) { if (!idok(o[3 + k])) return fail("bad argument"); vals[cf.params[k]] = get(o[3 + k]); }
                Val ret;
                if (!call(callee, &ret)) return false;
                vals[rid] = ret;
                break;
            }
            case 245: {  // Phi
                if (!check_value(0)) return false;
                bool found = false;
                for (unsigned k = 2; k + 1 < in.n; k += 2) {
                    if (o[k + 1] == prev_label) { if (!idok(o[k])) return fail("bad phi"); Val v = get(o[k]); vals[rid] = v; found = true; break; }
                }
                if (!found) return fail("phi without an entry for the block it was reached from");
                break;
            }
            case 12: {  // ExtInst
                if (!check_value(2)) return false;
                if (o[2] != m.glsl_set) return fail("extended instruction of an unknown set");
                for (unsigned k = 4; k < in.n; k++) if (!idok(o[k])) return fail("bad operand");
                Val r;
                if (!ext_inst(in, o, r, m.types[rtype])) return false;
                vals[rid] = r;
                break;
            }
            // ---- images ----
            case 86: case 100: {  // SampledImage (image, sampler) / Image (sampled image): the handle of the image
                if (!check_value(1)) return false;
                vals[rid].img = get(o[2]).img;
                break;
            }
            case 104: case 103: {  // ImageQuerySize / ImageQuerySizeLod
                if (!check_value(1)) return false;
                const Binding* b = image_of(get(o[2]));
                if (!b) return false;
                vals[rid].u[0] = b->width; vals[rid].u[1] = b->height;
                break;
            }
            case 98: {  // ImageRead
                if (!check_value(2)) return false;
                const Binding* b = image_of(get(o[2]));
                if (!b) return false;
                if (b->kind != 1) return fail("imageLoad from something that is not a storage image");
                const Val& c = get(o[3]);
                float* t;
                if (!texel_address(*b, c.i[0], c.i[1], &t)) return false;
                memcpy(vals[rid].f, t, 16);
                break;
            }
            case 99: {  // ImageWrite
                if (in.n < 3 || !idok(o[0]) || !idok(o[1]) || !idok(o[2])) return fail("bad image write");
                const Binding* b = image_of(get(o[0]));
                if (!b) return false;
                if (b->kind != 1) return fail("imageStore to something that is not a storage image");
                const Val& c = get(o[1]);
                float* t;
                if (!texel_address(*b, c.i[0], c.i[1], &t)) return false;
                memcpy(t, get(o[2]).f, 16);
                break;
            }
            case 88: {  // ImageSampleExplicitLod (U4)
                if (!check_value(2)) return false;
                const Binding* b = image_of(get(o[2]));
                if (!b) return false;
                if (b->kind != 2 || b->ox != 0 || b->oy != 0 || b->cw != b->width || b->ch != b->height) return fail("texture() needs a whole sampled image");
                if (in.n < 6 || o[4] != 2u || !idok(o[5]) || get(o[5]).f[0] != 0.0f) return fail("texture() with anything but Lod 0");
                const Val& c = get(o[3]);
                Tex tex{reinterpret_cast<const float*>(b->data), int(b->width), int(b->height)};
                float out[4];
                tex.sample(c.f[0], c.f[1], out);
                memcpy(vals[rid].f, out, 16);
                break;
            }
            // ---- composites ----
            case 80: {  // CompositeConstruct
                if (!check_value(0)) return false;
                Val r; unsigned at = 0;
                const Type& t = m.types[rtype];
                for (unsigned k = 2; k < in.n; k++) {
                    if (!idok(o[k])) return fail("bad constituent");
                    // a vector may be built from scalars and shorter vectors: the operand's width is what is left to fill, bounded by its definition
                    unsigned w;
                    if (t.kind == K_VECTOR) {
                        // width of operand k: scalars and vectors of the same component type; known from the defining instruction's result type
                        w = operand_words(o[k]);
                    } else if (t.kind == K_STRUCT) w = m.types[t.members[k - 2]].words;
                    else w = m.types[t.elem].words;
                    if (w == 0 || at + w > 16) return fail("composite wider than 16 words");
                    memcpy(r.u + at, get(o[k]).u, 4 * w);
                    at += w;
                }
                if (at != t.words) return fail("composite construct does not fill its type");
                vals[rid] = r;
                break;
            }
            case 81: {  // CompositeExtract
                if (!check_value(1)) return false;
                uint32_t type = operand_type(o[2]); unsigned word = 0;
                if (type == 0) return fail("extract from a value of unknown type");
                for (unsigned k = 3; k < in.n; k++) if (!constituent(type, word, o[k])) return false;
                Val r; const Val src = get(o[2]);
                const unsigned w = m.types[type].words;
                if (word + w > 16) return fail("extract beyond the value");
                memcpy(r.u, src.u + word, 4 * w);
                vals[rid] = r;
                break;
            }
            case 82: {  // CompositeInsert (object, composite, indexes)
                if (!check_value(2)) return false;
                uint32_t type = rtype; unsigned word = 0;
                for (unsigned k = 4; k < in.n; k++) if (!constituent(type, word, o[k])) return false;
                Val r = get(o[3]);
                const unsigned w = m.types[type].words;
                if (word + w > 16) return fail("insert beyond the value");
                memcpy(r.u + word, get(o[2]).u, 4 * w);
                vals[rid] = r;
                break;
            }
            case 79: {  // VectorShuffle
                if (!check_value(2)) return false;
                const unsigned n1 = operand_words(o[2]), n2 = operand_words(o[3]);
                const Val a = get(o[2]), b = get(o[3]);
                Val r;
                for (unsigned k = 4; k < in.n; k++) {
                    const uint32_t c = o[k];
                    if (k - 4 >= 16) return fail("shuffle wider than 16");
                    if (c == 0xffffffffu) r.u[k - 4] = 0;
                    else if (c < n1) r.u[k - 4] = a.u[c];
                    else if (c < n1 + n2) r.u[k - 4] = b.u[c - n1];
                    else return fail("shuffle component out of range");
                }
                vals[rid] = r;
                break;
            }
            case 77: {  // VectorExtractDynamic
                if (!check_value(2)) return false;
                const uint32_t idx = get(o[3]).u[0];
                if (idx >= operand_words(o[2])) return fail("dynamic vector index out of range");
                Val r; r.u[0] = get(o[2]).u[idx];
                vals[rid] = r;
                break;
            }
            case 83: if (!check_value(1)) return false; { Val r = get(o[2]); vals[rid] = r; } break;   // CopyObject
            case 1: if (!check_value(0)) return false; { Val r; if (flags & 1u) for (auto& x : r.u) x = 0x7fc0deadu; vals[rid] = r; } break;   // Undef in a function body
            default: {
                // ---- component-wise arithmetic ----
                if (!check_value(0)) return false;
                const Type& rt = m.types[rtype];
                const unsigned n = rt.words;
                if (n == 0) return fail("opcode " + std::to_string(in.op) + " is not one the three shaders use");
                Val r;
                auto A = [&](unsigned k) -> const Val& { return get(o[2 + k]); };
                auto operands = [&](unsigned k) { if (in.n != 2 + k) return fail("operand count of opcode " + std::to_string(in.op)); for (unsigned j = 0; j < k; j++) if (!idok(o[2 + j])) return fail("operand id"); return true; };
#define UN(expr)  { if (!operands(1)) return false; const Val a = A(0); for (unsigned k = 0; k < n; k++) { expr; } }
#define BIN(expr) { if (!operands(2)) return false; const Val a = A(0), b = A(1); for (unsigned k = 0; k < n; k++) { expr; } }
                switch (in.op) {
                    case 127: UN(r.f[k] = -a.f[k]) break;
                    case 126: UN(r.u[k] = 0u - a.u[k]) break;
                    case 129: BIN(r.f[k] = a.f[k] + b.f[k]) break;
                    case 131: BIN(r.f[k] = a.f[k] - b.f[k]) break;
                    case 133: BIN(r.f[k] = a.f[k] * b.f[k]) break;
                    case 136: BIN(r.f[k] = a.f[k] / b.f[k]) break;
                    case 128: BIN(r.u[k] = a.u[k] + b.u[k]) break;
                    case 130: BIN(r.u[k] = a.u[k] - b.u[k]) break;
                    case 132: BIN(r.u[k] = a.u[k] * b.u[k]) break;
                    case 134: BIN(if (b.u[k] == 0) return fail("integer division by zero"); r.u[k] = a.u[k] / b.u[k]) break;
                    case 137: BIN(if (b.u[k] == 0) return fail("integer modulo by zero"); r.u[k] = a.u[k] % b.u[k]) break;
                    case 135: BIN(if (b.i[k] == 0 || (a.i[k] == INT32_MIN && b.i[k] == -1)) return fail("signed division overflow"); r.i[k] = a.i[k] / b.i[k]) break;
                    case 138: BIN(if (b.i[k] == 0 || (a.i[k] == INT32_MIN && b.i[k] == -1)) return fail("signed remainder overflow"); r.i[k] = a.i[k] % b.i[k]) break;
                    case 142: { if (!operands(2)) return false; const Val a = A(0), b = A(1); for (unsigned k = 0; k < n; k++) r.f[k] = a.f[k] * b.f[0]; } break;   // VectorTimesScalar
                    case 145: {  // MatrixTimesVector (U8): columns of the matrix times the vector's components, summed left to right
                        if (!operands(2)) return false;
                        const Val a = A(0), b = A(1);
                        const unsigned cols = operand_words(o[3]);
                        if (cols * n > 16 || cols < 2) return fail("matrix times vector shape");
                        for (unsigned row = 0; row < n; row++) {
                            float s = a.f[row] * b.f[0];
                            for (unsigned c = 1; c < cols; c++) s = s + a.f[c * n + row] * b.f[c];
                            r.f[row] = s;
                        }
                        break;
                    }
                    case 148: {  // Dot (U8)
                        if (!operands(2)) return false;
                        const Val a = A(0), b = A(1);
                        const unsigned len = operand_words(o[2]);
                        float s = a.f[0] * b.f[0];
                        for (unsigned c = 1; c < len; c++) s = s + a.f[c] * b.f[c];
                        r.f[0] = s;
                        break;
                    }
                    case 109: UN(r.u[k] = !(a.f[k] >= 0.0f) ? 0u : (a.f[k] >= 4294967296.0f ? 0xffffffffu : uint32_t(a.f[k]))) break;   // ConvertFToU (U7)
                    case 110: UN(r.i[k] = vx_f2i(a.f[k])) break;                                                                      // ConvertFToS (U7)
                    case 111: UN(r.f[k] = float(a.i[k])) break;
                    case 112: UN(r.f[k] = float(a.u[k])) break;
                    case 124: UN(r.u[k] = a.u[k]) break;   // Bitcast
                    case 194: BIN(r.u[k] = a.u[k] >> (b.u[k] & 31u)) break;
                    case 195: BIN(r.i[k] = a.i[k] >> (b.u[k] & 31u)) break;
                    case 196: BIN(r.u[k] = a.u[k] << (b.u[k] & 31u)) break;
                    case 197: BIN(r.u[k] = a.u[k] | b.u[k]) break;
                    case 198: BIN(r.u[k] = a.u[k] ^ b.u[k]) break;
                    case 199: BIN(r.u[k] = a.u[k] & b.u[k]) break;
                    case 200: UN(r.u[k] = ~a.u[k]) break;
                    case 164: BIN(r.u[k] = (a.u[k] & 1u) == (b.u[k] & 1u)) break;
                    case 165: BIN(r.u[k] = (a.u[k] & 1u) != (b.u[k] & 1u)) break;
                    case 166: BIN(r.u[k] = (a.u[k] | b.u[k]) & 1u) break;
                    case 167: BIN(r.u[k] = a.u[k] & b.u[k] & 1u) break;
                    case 168: UN(r.u[k] = (a.u[k] & 1u) ^ 1u) break;
                    case 154: { if (!operands(1)) return false; const Val a = A(0); uint32_t any = 0; for (unsigned c = 0; c < operand_words(o[2]); c++) any |= a.u[c] & 1u; r.u[0] = any; } break;
                    case 155: { if (!operands(1)) return false; const Val a = A(0); uint32_t all = 1; for (unsigned c = 0; c < operand_words(o[2]); c++) all &= a.u[c] & 1u; r.u[0] = all; } break;
                    case 156: UN(r.u[k] = a.f[k] != a.f[k]) break;
                    case 157: UN(r.u[k] = vx_abs(a.f[k]) == __builtin_inff()) break;
                    case 169: {  // Select: a scalar or a per-component condition
                        if (!operands(3)) return false;
                        const Val c = A(0), a = A(1), b = A(2);
                        const bool per = operand_words(o[2]) == n && n > 1;
                        for (unsigned k = 0; k < n; k++) r.u[k] = (c.u[per ? k : 0] & 1u) ? a.u[k] : b.u[k];
                        break;
                    }
                    case 170: BIN(r.u[k] = a.u[k] == b.u[k]) break;
                    case 171: BIN(r.u[k] = a.u[k] != b.u[k]) break;
                    case 172: BIN(r.u[k] = a.u[k] > b.u[k]) break;
                    case 173: BIN(r.u[k] = a.i[k] > b.i[k]) break;
                    case 174: BIN(r.u[k] = a.u[k] >= b.u[k]) break;
                    case 175: BIN(r.u[k] = a.i[k] >= b.i[k]) break;
                    case 176: BIN(r.u[k] = a.u[k] < b.u[k]) break;
                    case 177: BIN(r.u[k] = a.i[k] < b.i[k]) break;
                    case 178: BIN(r.u[k] = a.u[k] <= b.u[k]) break;
                    case 179: BIN(r.u[k] = a.i[k] <= b.i[k]) break;
                    case 180: BIN(r.u[k] = a.f[k] == b.f[k]) break;
                    case 182: BIN(r.u[k] = a.f[k] < b.f[k] || a.f[k] > b.f[k]) break;
                    case 184: BIN(r.u[k] = a.f[k] < b.f[k]) break;
                    case 186: BIN(r.u[k] = a.f[k] > b.f[k]) break;
                    case 188: BIN(r.u[k] = a.f[k] <= b.f[k]) break;
                    case 190: BIN(r.u[k] = a.f[k] >= b.f[k]) break;
                    case 181: BIN(r.u[k] = !(a.f[k] < b.f[k] || a.f[k] > b.f[k])) break;
                    case 183: BIN(r.u[k] = !(a.f[k] == b.f[k])) break;
                    case 185: BIN(r.u[k] = !(a.f[k] >= b.f[k])) break;
                    case 187: BIN(r.u[k] = !(a.f[k] <= b.f[k])) break;
                    case 189: BIN(r.u[k] = !(a.f[k] > b.f[k])) break;
                    case 191: BIN(r.u[k] = !(a.f[k] < b.f[k])) break;
                    default: return fail("opcode " + std::to_string(in.op) + " is not one the three shaders use");
                }
#undef UN
#undef BIN
                vals[rid] = r;
                break;
            }
        }
    }
}

bool Thread::run(uint32_t gx, uint32_t gy) {
    memset(scratch.data(), 0, scratch.size());
    depth = 0;
    for (const Global& g : m.globals) {
        Val& v = vals[g.id];
        v = Val();
        const Type& t = m.types[g.type];
        if (g.storage == 6 || g.storage == 1) {   // Private, Input
            uint8_t* p = scratch.data() + m.var_scratch[g.id];
            v.ptr = Ptr{p, p, p + t.packed, g.type, false};
            if (g.storage == 1) {
                if (m.builtin_of[g.id] != 28u || t.words != 3) return fail("an input other than gl_GlobalInvocationID");
                const uint32_t id3[3] = {gx, gy, 0u};
                memcpy(p, id3, 12);
            } else if (m.var_init[g.id]) {
                unsigned at = 0;
                if (!store_rec(g.type, v.ptr, p, get(m.var_init[g.id]).u, at)) return false;
            }
            continue;
        }
        int found = -1;
        for (size_t k = 0; k < bindings.size(); k++)
            if (bindings[k].set == m.set_of[g.id] && bindings[k].binding == m.binding_of[g.id]) found = int(k);
        if (found < 0) return fail("nothing bound at set " + std::to_string(m.set_of[g.id]) + " binding " + std::to_string(m.binding_of[g.id]));
        const Binding& b = bindings[size_t(found)];
        if (g.storage == 2 || g.storage == 12) {   // Uniform, StorageBuffer: the caller's bytes with the module's layout
            if (b.kind != 0) return fail("a buffer variable bound to an image");
            v.ptr = Ptr{b.data, b.data, b.data + b.bytes, g.type, true};
        } else if (g.storage == 0) {                // UniformConstant: image, sampler
            if (t.kind == K_IMAGE) { if (b.kind != (t.sampled == 2 ? 1u : 2u)) return fail("image kind does not match binding " + std::to_string(b.binding)); }
            else if (t.kind == K_SAMPLER) { if (b.kind != 3) return fail("sampler binding"); }
            else return fail("unsupported UniformConstant variable");
            v.img = found;
        } else {
            return fail("unsupported storage class " + std::to_string(g.storage));
        }
    }
    return call(m.function_of[m.entry], nullptr);
}

thread_local std::string g_spv_error;

}  // namespace

extern "C" {
// One binding of a dispatch.  kind 0: uniform / storage buffer (data, bytes); 1: storage image rgba32f; 2: sampled image rgba32f;
// 3: sampler (no data).  Images: width x height is the image the shader sees (imageSize / textureSize); data holds the window
// [ox, ox + cw) x [oy, oy + ch) of it, row-major, 4 floats per texel — an access outside the window is an error, not a guess.
struct OrcSpvBinding { uint32_t set, binding, kind, pad; void* data; uint64_t bytes; uint32_t width, height, ox, oy, cw, ch; };

enum { ORC_SPV_POISON = 1 };   // flags; bits 8 and up: the instruction budget of one invocation in millions (0: 200)
}  // extern "C"

namespace {
int dispatch(const uint32_t* words, size_t nwords, const OrcSpvBinding* bind, int nbind, uint32_t x0, uint32_t y0, uint32_t x1,
             uint32_t y1, uint32_t flags, int nthreads, uint64_t* executed) {
    Module m;
    if (!parse(m, words, nwords)) { g_spv_error = m.error; return -1; }
    std::vector<Binding> b;
    for (int k = 0; k < nbind; k++) {
        const OrcSpvBinding& s = bind[k];
        if (s.kind > 3) { g_spv_error = "unknown binding kind"; return -1; }
        if ((s.kind == 1 || s.kind == 2) && (s.data == nullptr || uint64_t(s.cw) * s.ch * 16 != s.bytes || s.ox + s.cw > s.width || s.oy + s.ch > s.height)) {
            g_spv_error = "image binding " + std::to_string(s.binding) + ": window and byte count disagree";
            return -1;
        }
        if (s.kind == 0 && (s.data == nullptr || s.bytes == 0)) { g_spv_error = "empty buffer binding"; return -1; }
        b.push_back(Binding{s.set, s.binding, s.kind, static_cast<uint8_t*>(s.data), s.bytes, s.width, s.height, s.ox, s.oy, s.cw, s.ch});
    }
    if (x1 < x0 || y1 < y0) { g_spv_error = "empty range"; return -1; }
    nthreads = nthreads < 1 ? 1 : (nthreads > 256 ? 256 : nthreads);
    std::atomic<uint32_t> next_row{y0};
    std::atomic<uint64_t> total{0};
    std::atomic<bool> failed{false};
    std::string first_error;
    std::mutex lock;
    auto work = [&]() {
      try {
        Thread t(m, b);
        t.flags = flags;
        for (;;) {
            const uint32_t y = next_row.fetch_add(1);
            if (y >= y1 || failed.load()) break;
            for (uint32_t x = x0; x < x1; x++) {
                t.budget = t.executed + ((flags >> 8) ? uint64_t(flags >> 8) * 1000000ull : 200000000ull);
                if (!t.run(x, y)) {
                    std::lock_guard<std::mutex> g(lock);
                    if (!failed.exchange(true)) first_error = "invocation (" + std::to_string(x) + ", " + std::to_string(y) + "): " + t.error;
                    break;
                }
            }
        }
        total += t.executed;
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(lock);
        if (!failed.exchange(true)) first_error = std::string("exception: ") + e.what();
      }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < nthreads; k++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (executed) *executed = total.load();
    if (failed.load()) { g_spv_error = first_error; return -1; }
    return 0;
}
}  // namespace

extern "C" {

// Runs the module's GLCompute entry point once per invocation id (x, y, 0), x0 <= x < x1, y0 <= y < y1 (the workgroup shape does not
// matter to shaders without shared memory or barriers: these three have neither).  0, or -1 with orc_spirv_error() set.
// executed (optional): instructions interpreted, summed over the invocations.
int orc_spirv_dispatch(const uint32_t* words, size_t nwords, const OrcSpvBinding* bind, int nbind, uint32_t x0, uint32_t y0, uint32_t x1,
                       uint32_t y1, uint32_t flags, int nthreads, uint64_t* executed) {
    g_spv_error.clear();
    try {
        return dispatch(words, nwords, bind, nbind, x0, y0, x1, y1, flags, nthreads, executed);
    } catch (const std::exception& e) {     // out of memory on a hostile module: an error, not an abort through the C boundary
        g_spv_error = std::string("exception: ") + e.what();
        return -1;
    }
}

const char* orc_spirv_error(void) { return g_spv_error.c_str(); }

}  // extern "C"
