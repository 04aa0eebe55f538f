// smr_plan.cpp -- host-side planner of libstrided_hip.so.
//
// GPU analogue of the reference's planning chain
//   _mapreduce_fuse!  (src/mapreduce.jl:98-117)   -> canonicalise(): drop/sort/flip/fuse/dedupe
//   _mapreduce_order! (src/mapreduce.jl:119-139)  -> canonical dim order (destination-stride
//                                                    sorted) + kernel-family classification
//   _mapreduce_block!/_computeblocks (:142-180, :452-500)
//                                                 -> plan_tiles(): LDS tile shape instead of
//                                                    L1 cache blocks
// The CPU heuristics (importance digits, 32 KiB blocks, 64 B lines) are NOT transliterated:
// on CDNA4 the quantities that matter are the unit-stride axis of every operand (coalescing),
// the LDS budget and the number of workgroups.  The faithful restatement of the CPU planner
// lives in oracle/ as test infrastructure.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "smr_internal.h"

namespace smr {

Options& options() {
    static Options o = [] {
        Options d;
        if (const char* e = std::getenv("SMR_JIT")) d.jit = std::atoll(e);  // SMR_JIT=0: always interpret
        return d;
    }();
    return o;
}

constexpr i64 FLAT_GROUP_MAX = 64;

static int nextpow2_log(i64 v) {
    int l = 0;
    while (((i64)1 << l) < v) ++l;
    return l;
}

// ---- f-program validation / recognition -------------------------------------------------------
static int check_prog(const ProgD& p, int M, int& maxdepth) {
    int sp = 0;
    maxdepth = 0;
    for (int pc = 0; pc < p.len; ++pc) {
        int op = p.code[2 * pc], imm = p.code[2 * pc + 1];
        if (op == SMR_OP_ARG) {
            if (imm < 1 || imm >= M) return -1;
            ++sp;
        } else if (op == SMR_OP_CONST) {
            if (imm >= p.nconst) return -1;
            ++sp;
        } else if (op >= SMR_OP_NEG && op <= SMR_OP_WIDEN) {
            if (sp < 1) return -1;
        } else if (op >= SMR_OP_ADD && op <= SMR_OP_NE) {
            if (sp < 2) return -1;
            --sp;
        } else if (op == SMR_OP_SELECT) {
            if (sp < 3) return -1;
            sp -= 2;
        } else {
            return -1;
        }
        maxdepth = std::max(maxdepth, sp);
    }
    return sp == 1 ? 0 : -1;
}

// Julia types every integer operation by its operands (Int32 * Int32 wraps at 32 bits, Int8 - Int8 at 8, a literal is Int64), the
// integer class computes everything in one 64-bit domain.  The two agree exactly when no value that Julia would have wrapped at a
// narrower width is ever OBSERVED at a wider one.  Per stack slot: (`bits`, `sgn`) = Julia's type of the value, `cong` = the device
// value is congruent to Julia's modulo 2^cong (64: identical; otherwise cong >= bits).  Ring operations (+ - * neg abs2) keep
// congruences modulo the width of their result, order and equality (min max abs < <= == select's condition ...) need identical
// operands, and the destination (width wd) observes the low wd bits.
// Round 5: where an observer needs more than the slot offers, the value is re-wrapped to its Julia type right after the instruction
// that produced it (SMR_OP_WRAP_*: sign- or zero-extension of the low bits, one or two VALU instructions) -- lazily, so
// `Int32 a .* b .+ c` into an Int32 destination stays the three-instruction program it was, `Int32 a .* b` into an Int64 destination
// gets one wrap at the end, and UInt8 `min(a - b, c)` one after the subtraction.  (Before: both were refused with SMR_EUNSUPPORTED,
// ADVICE r3 / VERDICT r4 "missing #4".)  Returns false only for what cannot be typed statically: arithmetic on the result of an
// ifelse whose branches have different types.
static bool int_class_fit_julia(const smr_problem* p, const bool* isbool, ProgD& prog, int* nwraps) {
    auto width = [](int dt) { return 8 << ((dt - SMR_I8) & 3); };  // I8 I16 I32 I64 U8 U16 U32 U64
    struct Slot {
        int bits;
        bool sgn;
        int cong;
        int prod;  // the instruction that produced the value
        bool amb;  // one of two differently typed ifelse branches: Julia's type depends on the data
    };
    Slot st[STACK + 4];
    int sp = 0, wrap_after[SMR_MAXPROG] = {0}, nw = 0;
    auto promote = [](const Slot& a, const Slot& b, int& bits, bool& sgn) {
        if (a.bits == 1) { bits = b.bits; sgn = b.sgn; return; }
        if (b.bits == 1) { bits = a.bits; sgn = a.sgn; return; }
        if (a.bits != b.bits) { const Slot& w = a.bits > b.bits ? a : b; bits = w.bits; sgn = w.sgn; return; }
        bits = a.bits; sgn = a.sgn && b.sgn;
    };
    auto exact = [&](Slot& s) {  // cong < 64 implies 8 <= bits <= cong < 64
        if (s.cong == 64) return;
        static const int code[2][3] = {{SMR_OP_WRAP_U8, SMR_OP_WRAP_U16, SMR_OP_WRAP_U32}, {SMR_OP_WRAP_I8, SMR_OP_WRAP_I16, SMR_OP_WRAP_I32}};
        wrap_after[s.prod] = code[s.sgn ? 1 : 0][s.bits == 8 ? 0 : s.bits == 16 ? 1 : 2];
        s.cong = 64;
        ++nw;
    };
    for (int pc = 0; pc < prog.len; ++pc) {
        const int op = prog.code[2 * pc], imm = prog.code[2 * pc + 1];
        switch (op) {
            case SMR_OP_ARG:
                if (isbool[imm]) st[sp++] = {1, false, 64, pc, false};
                else st[sp++] = {width(p->ops[imm].dtype), p->ops[imm].dtype < SMR_U8, 64, pc, false};
                break;
            case SMR_OP_CONST: st[sp++] = {64, true, 64, pc, false}; break;
            case SMR_OP_CONJ: case SMR_OP_REAL: st[sp - 1].prod = pc; break;
            case SMR_OP_IMAG: st[sp - 1] = {64, true, 64, pc, false}; break;
            case SMR_OP_WIDEN:
                exact(st[sp - 1]);
                st[sp - 1] = {64, true, 64, pc, false};
                break;
            case SMR_OP_NEG: case SMR_OP_ABS2: {
                Slot& a = st[sp - 1];
                if (a.amb) return false;
                a.prod = pc;
                if (a.bits == 1) {  // Bool: -true is an Int, abs2(true) is true
                    if (op == SMR_OP_NEG) { a.bits = 64; a.sgn = true; }
                    break;
                }
                a.cong = std::min(a.cong, a.bits);
                break;
            }
            case SMR_OP_ABS: {
                Slot& a = st[sp - 1];
                if (a.amb) return false;
                exact(a);
                a.prod = pc;
                if (a.bits > 1 && a.sgn) a.cong = a.bits;  // abs(typemin) wraps at the operand's width
                break;
            }
            case SMR_OP_ADD: case SMR_OP_SUB: case SMR_OP_MUL: {
                Slot b = st[--sp];
                Slot& a = st[sp - 1];
                if (a.amb || b.amb) return false;
                int w;
                bool sg;
                promote(a, b, w, sg);
                if (a.bits == 1 && b.bits == 1) {  // Bool * Bool is a Bool, Bool (+,-) Bool an Int
                    if (op != SMR_OP_MUL) { w = 64; sg = true; }
                }
                // an operand Julia converts to a wider type first must be identical, not merely congruent at its own width
                if (a.cong < w) exact(a);
                if (b.cong < w) { exact(b); }
                a = {w, sg, w == 1 ? 64 : std::min({a.cong, b.cong, w}), pc, false};
                break;
            }
            case SMR_OP_MIN: case SMR_OP_MAX: {
                Slot b = st[--sp];
                Slot& a = st[sp - 1];
                exact(a);
                exact(b);
                int w;
                bool sg;
                promote(a, b, w, sg);
                // min(::Int8, ::UInt32) promotes to UInt32 and throws for a negative value in Julia; the oracle keeps the mathematical
                // value.  Either way it is not a value of the promoted type: never re-wrapped, no arithmetic on it.
                a = {w, sg, 64, pc, a.amb || b.amb || (!sg && ((a.sgn && a.bits > 1) || (b.sgn && b.bits > 1)))};
                break;
            }
            case SMR_OP_LT: case SMR_OP_LE: case SMR_OP_GT: case SMR_OP_GE: case SMR_OP_EQ: case SMR_OP_NE: {
                Slot b = st[--sp];
                Slot& a = st[sp - 1];
                exact(a);
                exact(b);
                a = {1, false, 64, pc, false};  // Bool
                break;
            }
            case SMR_OP_SELECT: {
                Slot c = st[--sp], b = st[--sp];
                Slot& a = st[sp - 1];
                exact(a);
                const bool same = b.bits == c.bits && b.sgn == c.sgn && !b.amb && !c.amb;  // (a data-dependent type is never re-wrapped: it stays identical)
                if (!same) { exact(b); exact(c); }
                int w;
                bool sg;
                promote(b, c, w, sg);
                a = {w, sg, std::min(b.cong, c.cong), pc, !same || b.amb || c.amb};
                break;
            }
            default: return false;
        }
    }
    const int wd = width(p->ops[0].dtype);
    const bool ring = p->redop == SMR_RED_NONE || p->redop == SMR_RED_ADD || p->redop == SMR_RED_MUL;
    if (!(ring ? st[0].cong >= wd : st[0].cong == 64)) exact(st[0]);
    if (nwraps) *nwraps = nw;
    if (nw == 0) return true;
    if (prog.len + nw > SMR_MAXPROG) return false;
    uint8_t code[2 * SMR_MAXPROG];
    int n = 0;
    for (int pc = 0; pc < prog.len; ++pc) {
        code[2 * n] = prog.code[2 * pc];
        code[2 * n + 1] = prog.code[2 * pc + 1];
        ++n;
        if (wrap_after[pc]) {
            code[2 * n] = (uint8_t)wrap_after[pc];
            code[2 * n + 1] = 0;
            ++n;
        }
    }
    std::memcpy(prog.code, code, sizeof(uint8_t) * 2 * (size_t)n);
    prog.len = n;
    return true;
}

static void recognise(Canon& c) {
    const ProgD& p = c.prog;
    auto op = [&](int i) { return (int)p.code[2 * i]; };
    auto im = [&](int i) { return (int)p.code[2 * i + 1]; };
    auto isarg = [&](int i, int k) { return op(i) == SMR_OP_ARG && im(i) == k; };
    auto isconst = [&](int i) { return op(i) == SMR_OP_CONST; };
    auto cre = [&](int i) { return p.consts[2 * im(i)]; };
    auto cim = [&](int i) { return p.consts[2 * im(i) + 1]; };
    c.fkind = FK_PROG;
    const int n = p.len, nin = c.M - 1;
    if (n == 1 && isarg(0, 1) && nin == 1) { c.fkind = FK_IDENT; return; }
    if (n == 3 && isarg(0, 1) && isarg(1, 2) && op(2) == SMR_OP_ADD && nin == 2) { c.fkind = FK_ADD2; return; }
    if (n == 5 && isarg(0, 1) && isarg(1, 2) && op(2) == SMR_OP_ADD && isarg(3, 3) && op(4) == SMR_OP_ADD && nin == 3) {
        c.fkind = FK_ADD3; return;
    }
    if (n == 7 && isarg(0, 1) && isarg(1, 2) && op(2) == SMR_OP_ADD && isarg(3, 3) && op(4) == SMR_OP_ADD &&
        isarg(5, 4) && op(6) == SMR_OP_ADD && nin == 4) {
        c.fkind = FK_ADD4; return;
    }
    if (n == 3 && nin == 1 && op(2) == SMR_OP_MUL &&
        ((isarg(0, 1) && isconst(1)) || (isconst(0) && isarg(1, 1)))) {
        int ci = isconst(0) ? 0 : 1;
        c.fkind = FK_SCALE; c.fc[0] = cre(ci); c.fc[1] = cim(ci); return;
    }
    if (n == 5 && nin == 2 && isarg(0, 1) && isarg(1, 2) && op(2) == SMR_OP_ADD && isconst(3) && op(4) == SMR_OP_DIV) {
        c.fkind = FK_SYM; c.fc[0] = cre(3); c.fc[1] = cim(3); return;
    }
    if (n == 5 && nin == 2 && isconst(0) && isarg(1, 1) && op(2) == SMR_OP_MUL && isarg(3, 2) && op(4) == SMR_OP_ADD) {
        c.fkind = FK_AXPY; c.fc[0] = cre(0); c.fc[1] = cim(0); return;
    }
    if (n == 7 && nin == 2 && isconst(0) && isarg(1, 1) && op(2) == SMR_OP_MUL && isconst(3) && isarg(4, 2) &&
        op(5) == SMR_OP_MUL && op(6) == SMR_OP_ADD) {
        c.fkind = FK_AXPBY; c.fc[0] = cre(0); c.fc[1] = cim(0); c.fc[2] = cre(3); c.fc[3] = cim(3); return;
    }
    if (n == 2 && nin == 1 && isarg(0, 1) && op(1) == SMR_OP_ABS2) { c.fkind = FK_ABS2; return; }
    if (n == 3 && nin == 2 && isarg(0, 1) && isarg(1, 2) && op(2) == SMR_OP_MUL) { c.fkind = FK_MUL2; return; }
    // a .* exp.(c .* a) .+ sin.(a .* a)   (real types)
    if (n == 11 && nin == 1 && (c.ct == SMR_F32 || c.ct == SMR_F64) && isarg(0, 1) && isconst(1) && isarg(2, 1) &&
        op(3) == SMR_OP_MUL && op(4) == SMR_OP_EXP && op(5) == SMR_OP_MUL && isarg(6, 1) && isarg(7, 1) &&
        op(8) == SMR_OP_MUL && op(9) == SMR_OP_SIN && op(10) == SMR_OP_ADD && cim(1) == 0.0) {
        c.fkind = FK_EXPR5; c.fc[0] = cre(1); return;
    }
}

// ---- canonicalisation ---------------------------------------------------------------------------
int canonicalise(const smr_problem* p0, Canon& c) {
    if (!p0) return set_error(SMR_EINVAL, "null problem");
    // Bool operands are UInt8 operands for every kernel; only the typing of an integer f-program tells them apart
    smr_problem pcopy = *p0;
    bool isbool[MAXM] = {false};
    if (pcopy.M >= 1 && pcopy.M <= MAXM)
        for (int k = 0; k < pcopy.M; ++k)
            if (pcopy.ops[k].dtype == SMR_BOOL) {
                isbool[k] = true;
                pcopy.ops[k].dtype = SMR_U8;
            }
    const smr_problem* p = &pcopy;
    const int N0 = p->N, M0 = p->M;
    if (N0 < 1 || N0 > MAXN) return set_error(SMR_EINVAL, "rank N out of range 1..8");
    if (M0 < 1 || M0 > MAXM) return set_error(SMR_EINVAL, "operand count M out of range 1..8");
    for (int i = 0; i < N0; ++i)
        if (p->dims[i] < 1) return set_error(SMR_EINVAL, "every dim must be >= 1 (zero-size is handled by the caller, src/mapreduce.jl:48,88)");
    for (int k = 0; k < M0; ++k) {
        if (!p->ops[k].base) return set_error(SMR_EINVAL, "null operand base pointer");
        if (dtype_size(p->ops[k].dtype) == 0) return set_error(SMR_EINVAL, "bad operand dtype");
    }
    if (p->redop < SMR_RED_NONE || p->redop > SMR_RED_OR) return set_error(SMR_EINVAL, "bad redop");
    if (p->initop < SMR_INIT_NONE || p->initop > SMR_INIT_CONJ) return set_error(SMR_EINVAL, "bad initop");
    if (p->redop == SMR_RED_NONE && p->initop != SMR_INIT_NONE)
        return set_error(SMR_EINVAL, "initop requires a reduction op (src/mapreduce.jl:310-316)");

    // program
    ProgD& prog = c.prog;
    std::memset(&prog, 0, sizeof(prog));
    if (!p->fprog || p->fprog_len == 0) {
        if (M0 < 2) return set_error(SMR_EINVAL, "identity program needs an input");
        prog.len = 1;
        prog.code[0] = SMR_OP_ARG;
        prog.code[1] = 1;
    } else {
        if (p->fprog_len < 0 || p->fprog_len > SMR_MAXPROG) return set_error(SMR_EINVAL, "f-program too long");
        prog.len = p->fprog_len;
        std::memcpy(prog.code, p->fprog, (size_t)(2 * p->fprog_len));
    }
    if (p->nconsts < 0 || p->nconsts > SMR_MAXCONST) return set_error(SMR_EINVAL, "too many constants");
    if (p->nconsts > 0 && !p->fconsts) return set_error(SMR_EINVAL, "null fconsts");
    prog.nconst = p->nconsts;
    for (int i = 0; i < 2 * p->nconsts; ++i) prog.consts[i] = p->fconsts[i];
    int depth = 0;
    if (check_prog(prog, std::max(M0, 1), depth) != 0) return set_error(SMR_EINVAL, "malformed f-program");
    if (depth > STACK) return set_error(SMR_EUNSUPPORTED, "f-program needs more than 8 stack slots");

    // compute class (Julia promote_type over the operand eltypes, restricted to the four
    // float classes of the reference tests)
    bool dbl = false, cplx = false, anyint = false;
    for (int k = 0; k < M0; ++k) {
        int dt = p->ops[k].dtype;
        if (dt == SMR_F64 || dt == SMR_C64) dbl = true;
        if (dt == SMR_C32 || dt == SMR_C64) cplx = true;
        if (dt >= SMR_I8) { anyint = true; dbl = true; }
    }
    for (int i = 0; i < prog.nconst; ++i)
        if (prog.consts[2 * i + 1] != 0.0) cplx = true;
    for (int pc = 0; pc < prog.len; ++pc)
        if (prog.code[2 * pc] == SMR_OP_WIDEN) dbl = true;  // a 64-bit scalar takes part (strided_hip.h)
    c.ct = cplx ? (dbl ? SMR_C64 : SMR_C32) : (dbl ? SMR_F64 : SMR_F32);
    // The integer class (round 3): every operand has an integer eltype and f stays inside the integers -> wrapping
    // 64-bit arithmetic like Julia's (typeof(op(...)) is the accumulator type, src/mapreduce.jl:55-72); narrower
    // destinations truncate on store.  UInt64 takes part in ring operations only (no order on the device).
    {
        bool allint = true, has_u64 = false, has_signed = false, has_const = false, eqne = false;
        for (int k = 0; k < M0; ++k) {
            if (p->ops[k].dtype < SMR_I8) allint = false;
            if (p->ops[k].dtype == SMR_U64) has_u64 = true;
            if (p->ops[k].dtype >= SMR_I8 && p->ops[k].dtype <= SMR_I64) has_signed = true;
        }
        bool closed = true, ordered = p->redop == SMR_RED_MIN || p->redop == SMR_RED_MAX;
        for (int pc = 0; pc < prog.len && closed; ++pc) {
            const int op = prog.code[2 * pc];
            switch (op) {
                case SMR_OP_ARG: case SMR_OP_NEG: case SMR_OP_ABS2: case SMR_OP_CONJ: case SMR_OP_REAL: case SMR_OP_IMAG:
                case SMR_OP_ADD: case SMR_OP_SUB: case SMR_OP_MUL: case SMR_OP_SELECT: case SMR_OP_WIDEN: break;
                case SMR_OP_ABS: case SMR_OP_MIN: case SMR_OP_MAX: case SMR_OP_LT: case SMR_OP_LE: case SMR_OP_GT: case SMR_OP_GE:
                    ordered = true;
                    break;
                // == and !=: equality of bit patterns IS Julia's equality as long as nothing signed can meet a UInt64 -- Julia
                // compares a UInt64 with a signed value mathematically (-1 != 0xffff...ff), a 64-bit domain compares the patterns.
                // Signed values come from signed operands and from constants (integer literals are Int64: UInt8 - 10 is an Int64).
                case SMR_OP_EQ: case SMR_OP_NE:
                    eqne = true;
                    break;
                case SMR_OP_CONST: {
                    has_const = true;
                    const double re = prog.consts[2 * prog.code[2 * pc + 1]], im = prog.consts[2 * prog.code[2 * pc + 1] + 1];
                    // integer-valued (or a +-Inf seed of a min / max reduction, which saturates to typemax / typemin)
                    if (im != 0.0 || !(re == std::floor(re) || std::isinf(re)) || (std::fabs(re) > 9223372036854775808.0 && !std::isinf(re))) closed = false;
                    break;
                }
                default: closed = false;
            }
        }
        if (p->initop == SMR_INIT_SCALE || p->initop == SMR_INIT_CONST) {
            const double re = p->initarg[0], im = p->initarg[1];
            if (im != 0.0 || re != std::floor(re) || std::fabs(re) > 9223372036854775808.0) closed = false;
        }
        if (eqne && has_u64 && (has_signed || has_const)) ordered = true;  // (round 5: all-unsigned equality tests are exact and stay on the device)
        if (allint && closed && !(has_u64 && ordered)) {
            if (!int_class_fit_julia(p, isbool, prog, &c.int_wraps))
                return set_error(SMR_EUNSUPPORTED,
                                 "integer f-program that cannot be typed statically (arithmetic on an ifelse whose branches have different integer "
                                 "types), or too long once narrow intermediate results are re-wrapped to their Julia types");
            c.ct = SMR_I64;
        }
    }
    c.redop = p->redop;
    c.initop = p->initop;
    c.initarg[0] = p->initarg[0];
    c.initarg[1] = p->initarg[1];
    c.bitcopy = false;
    if (anyint && c.ct == SMR_I64 && p->redop == SMR_RED_NONE && M0 == 2 && p->ops[0].dtype == p->ops[1].dtype && prog.len == 1 &&
        prog.code[0] == SMR_OP_ARG) {
        c.bitcopy = true;  // a pure move stays a bit copy of opaque elements (any width, vectorised)
        c.ct = SMR_F64;
    }
    if (anyint && c.ct != SMR_I64 && !c.bitcopy) {
        bool pure = p->redop == SMR_RED_NONE && M0 == 2 && p->ops[0].dtype == p->ops[1].dtype && prog.len == 1 &&
                    prog.code[0] == SMR_OP_ARG;
        if (pure) c.bitcopy = true;
        // Integer data that is not merely moved is computed in Float64 (the four float classes are the device
        // scope): exact for every value of an 8/16/32-bit type and for sums / counts below
        // 2^53, which is what the reference's integer tests need (counting reductions, test/othertests.jl:116,123).
        // 64-bit integer INPUTS can hold values a double cannot: refuse them instead of rounding silently -- the
        // reference-side binding falls back to the CPU method (Julia's wrapping Int64 arithmetic).
        if (!pure)
            for (int k = 1; k < M0; ++k)
                if (p->ops[k].dtype == SMR_I64 || p->ops[k].dtype == SMR_U64)
                    return set_error(SMR_EUNSUPPORTED,
                                     "64-bit integer inputs outside the integer class (all operands integer, f built from + - * neg abs abs2 min max "
                                     "comparisons select and integer constants): the arithmetic would run in Float64, which is exact only below 2^53");
    }

    // working copies
    int N = 0;
    i64 dims[MAXN];
    i64 str[MAXM][MAXN];
    int M = M0;
    for (int i = 0; i < N0; ++i) {
        if (p->dims[i] == 1) continue;  // size-1 dims carry no information
        dims[N] = p->dims[i];
        for (int k = 0; k < M0; ++k) str[k][N] = p->ops[k].strides[i];
        ++N;
    }
    i64 off[MAXM];
    void* base[MAXM];
    int dtype[MAXM], conj[MAXM];
    for (int k = 0; k < M0; ++k) {
        off[k] = p->ops[k].offset;
        base[k] = p->ops[k].base;
        dtype[k] = p->ops[k].dtype;
        conj[k] = (p->ops[k].conj && (dtype[k] == SMR_C32 || dtype[k] == SMR_C64)) ? 1 : 0;
    }

    // dedupe identical inputs (capturestridedargs does not: src/broadcast.jl:41-46), and
    // drop inputs the program never reads
    {
        int remap[MAXM];
        bool used[MAXM] = {false};
        for (int pc = 0; pc < prog.len; ++pc)
            if (prog.code[2 * pc] == SMR_OP_ARG) used[prog.code[2 * pc + 1]] = true;
        int newM = 1;
        int keep[MAXM];
        keep[0] = 0;
        for (int k = 1; k < M; ++k) {
            remap[k] = -1;
            if (!used[k]) continue;
            for (int j = 1; j < newM; ++j) {
                int q = keep[j];
                bool same = base[q] == base[k] && off[q] == off[k] && dtype[q] == dtype[k] && conj[q] == conj[k];
                for (int i = 0; i < N && same; ++i) same = str[q][i] == str[k][i];
                if (same) { remap[k] = j; break; }
            }
            if (remap[k] < 0) { keep[newM] = k; remap[k] = newM; ++newM; }
        }
        for (int pc = 0; pc < prog.len; ++pc)
            if (prog.code[2 * pc] == SMR_OP_ARG) prog.code[2 * pc + 1] = (uint8_t)remap[prog.code[2 * pc + 1]];
        i64 s2[MAXM][MAXN], o2[MAXM];
        void* b2[MAXM];
        int d2[MAXM], c2[MAXM];
        for (int j = 0; j < newM; ++j) {
            int q = keep[j];
            for (int i = 0; i < N; ++i) s2[j][i] = str[q][i];
            o2[j] = off[q]; b2[j] = base[q]; d2[j] = dtype[q]; c2[j] = conj[q];
            c.orig[j] = q;
        }
        M = newM;
        for (int j = 0; j < M; ++j) {
            for (int i = 0; i < N; ++i) str[j][i] = s2[j][i];
            off[j] = o2[j]; base[j] = b2[j]; dtype[j] = d2[j]; conj[j] = c2[j];
        }
    }

    // a pure map must not alias destination elements (the reference would serialise the
    // writes; on a GPU that is a race)
    if (p->redop == SMR_RED_NONE)
        for (int i = 0; i < N; ++i)
            if (str[0][i] == 0) return set_error(SMR_EUNSUPPORTED, "map into a destination with a zero stride");

    // flip dims so that destination strides (kept dims) / first-input strides (reduced dims)
    // are positive; iteration direction is irrelevant for map and for associative reductions
    for (int i = 0; i < N; ++i) {
        i64 lead = str[0][i];
        if (lead == 0)
            for (int k = 1; k < M && lead == 0; ++k) lead = str[k][i];
        if (lead < 0)
            for (int k = 0; k < M; ++k) {
                off[k] += (dims[i] - 1) * str[k][i];
                str[k][i] = -str[k][i];
            }
    }

    // sort: kept dims by destination stride, then reduced dims by smallest input stride
    std::vector<int> perm(N);
    for (int i = 0; i < N; ++i) perm[i] = i;
    auto minin = [&](int i) {
        i64 m = INT64_MAX;
        for (int k = 1; k < M; ++k)
            if (str[k][i] != 0) m = std::min<i64>(m, std::llabs(str[k][i]));
        return m;
    };
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        bool ra = str[0][a] == 0, rb = str[0][b] == 0;
        if (ra != rb) return !ra;
        if (!ra) return str[0][a] < str[0][b];
        return minin(a) < minin(b);
    });
    i64 sd[MAXN], ss[MAXM][MAXN];
    for (int i = 0; i < N; ++i) {
        sd[i] = dims[perm[i]];
        for (int k = 0; k < M; ++k) ss[k][i] = str[k][perm[i]];
    }
    // fuse adjacent dims that are jointly contiguous in every operand (same rule as
    // src/mapreduce.jl:103-115, applied after sorting so it fires more often); never across
    // the kept/reduced boundary
    int NF = 0;
    i64 fd[MAXN], fs[MAXM][MAXN];
    for (int i = 0; i < N; ++i) {
        bool merged = false;
        if (NF > 0) {
            bool kb = fs[0][NF - 1] != 0, ka = ss[0][i] != 0;
            bool ok = (kb == ka);
            for (int k = 0; k < M && ok; ++k) ok = ss[k][i] == fd[NF - 1] * fs[k][NF - 1];
            if (ok) {
                fd[NF - 1] *= sd[i];
                merged = true;
            }
        }
        if (!merged) {
            fd[NF] = sd[i];
            for (int k = 0; k < M; ++k) fs[k][NF] = ss[k][i];
            ++NF;
        }
    }
    if (NF == 0) {  // every dim had size 1: a single element
        NF = 1;
        fd[0] = 1;
        for (int k = 0; k < M; ++k) fs[k][0] = (k == 0 && p->redop == SMR_RED_NONE) ? 1 : 0;
        if (p->redop != SMR_RED_NONE) fs[0][0] = 0;
    }
    c.N = NF;
    c.M = M;
    c.NK = 0;
    c.total = 1;
    c.nout = 1;
    for (int i = 0; i < NF; ++i) {
        c.dims[i] = fd[i];
        c.total *= fd[i];
        if (fs[0][i] != 0) { c.NK = i + 1; c.nout *= fd[i]; }
    }
    if (p->redop == SMR_RED_NONE) { c.NK = NF; }
    c.mixed = false;
    for (int k = 0; k < M; ++k) {
        for (int i = 0; i < NF; ++i) c.strides[k][i] = fs[k][i];
        for (int i = NF; i < MAXN; ++i) c.strides[k][i] = 0;
        c.offsets[k] = off[k];
        c.base[k] = base[k];
        c.dtype[k] = dtype[k];
        c.conj[k] = conj[k];
        c.esize[k] = dtype_size(dtype[k]);
        if (dtype[k] != c.ct) c.mixed = true;
    }
    for (int i = NF; i < MAXN; ++i) c.dims[i] = 1;
    if (c.bitcopy) c.mixed = false;

    // algorithmic bytes: every distinct buffer once
    {
        std::map<void*, i64> foot;
        for (int k = 0; k < M; ++k) {
            i64 n = 1;
            for (int i = 0; i < NF; ++i)
                if (c.strides[k][i] != 0) n *= c.dims[i];
            i64 bytes = n * c.esize[k];
            auto it = foot.find(c.base[k]);
            if (it == foot.end()) foot[c.base[k]] = bytes;
            else it->second = std::max(it->second, bytes);
        }
        c.algbytes = 0;
        for (auto& kv : foot) c.algbytes += kv.second;
    }
    recognise(c);
    return SMR_OK;
}

// ---- tile planning (FAM_TILED) -------------------------------------------------------------------
// fast (unit-stride) axis of operand k, or -1
static int fast_axis(const Canon& c, int k) {
    for (int i = 0; i < c.N; ++i)
        if (c.dims[i] > 1 && (c.strides[k][i] == 1 || c.strides[k][i] == -1)) return i;
    return -1;
}

// ---- locality-aware tile order ------------------------------------------------------------------------
// When several inputs are differently permuted views of ONE buffer (B .= (A .+ A')./2, the 4-way
// permuted sum), the tiles T, pi(T), pi^2(T), ... read the same regions of that buffer.  MI355X has
// eight XCDs with private L2s and dispatches workgroup b to XCD b mod 8, so in natural tile order
// every XCD fetches every region it touches by itself: the buffer crosses the fabric once per
// permutation.  Here the tiles are grouped into super-tiles whose extents are invariant under the
// permutations, the super-tiles into orbits of the permutation group, and the orbit-major list is
// cut into eight contiguous runs, one per XCD (workgroup 8*slot + x executes entry slot of run x):
// the members of an orbit meet in one L2 at about the same time.
static void plan_tile_order(const Canon& c, TilePlan& t, const int* lg) {
    constexpr int NX = 8;  // XCDs
    if (t.grid < 2 * NX || t.grid > ((i64)1 << 22)) return;
    // generators: input k reads through strides that are a permutation of input j's
    std::vector<std::array<int, MAXN>> gens;
    for (int j = 1; j < c.M; ++j)
        for (int k = j + 1; k < c.M; ++k) {
            if (c.esize[j] != c.esize[k]) continue;
            if ((char*)c.base[j] + c.offsets[j] * c.esize[j] != (char*)c.base[k] + c.offsets[k] * c.esize[k]) continue;
            std::array<int, MAXN> pi;
            bool used[MAXN] = {false}, ok = true, ident = true;
            for (int d = 0; d < c.N && ok; ++d) {
                int hit = -1;
                if (c.strides[j][d] == c.strides[k][d] && !used[d]) hit = d;  // prefer fixed points
                for (int e = 0; e < c.N && hit < 0; ++e)
                    if (!used[e] && c.strides[j][e] == c.strides[k][d] && c.dims[e] == c.dims[d]) hit = e;
                if (hit < 0) ok = false;
                else {
                    used[hit] = true;
                    pi[d] = hit;
                    if (hit != d) ident = false;
                }
            }
            if (ok && !ident) gens.push_back(pi);
        }
    if (gens.empty()) return;
    // super-tile extents: equal along every cycle of every generator
    // (measured: widening the super-tiles to a full 128-B line along every permuted dim does not help)
    int E[MAXN];
    for (int d = 0; d < c.N; ++d) E[d] = lg[d];
    for (bool changed = true; changed;) {
        changed = false;
        for (const auto& pi : gens)
            for (int d = 0; d < c.N; ++d) {
                const int m = std::max(E[d], E[pi[d]]);
                if (E[d] != m || E[pi[d]] != m) {
                    E[d] = E[pi[d]] = m;
                    changed = true;
                }
            }
    }
    i64 sg[MAXN], ns = 1;
    for (int d = 0; d < c.N; ++d) {
        sg[d] = (c.dims[d] + ((i64)1 << E[d]) - 1) >> E[d];
        ns *= sg[d];
    }
    // orbits of super-tile coordinates (breadth first), largest orbits first
    std::vector<char> seen((size_t)ns, 0);
    std::vector<std::vector<uint32_t>> orbits;
    auto decode = [&](i64 id, i64* sc) {
        for (int d = 0; d < c.N; ++d) {
            sc[d] = id % sg[d];
            id /= sg[d];
        }
    };
    auto encode = [&](const i64* sc) {
        i64 id = 0;
        for (int d = c.N - 1; d >= 0; --d) id = id * sg[d] + sc[d];
        return id;
    };
    for (i64 s0 = 0; s0 < ns; ++s0) {
        if (seen[(size_t)s0]) continue;
        std::vector<uint32_t> orb{(uint32_t)s0};
        seen[(size_t)s0] = 1;
        for (size_t head = 0; head < orb.size(); ++head) {
            i64 sc[MAXN], sp[MAXN];
            decode(orb[head], sc);
            for (const auto& pi : gens) {
                // input k at box coordinate sc touches what input j touches at sp, sp[pi[d]] = sc[d]
                for (int d = 0; d < c.N; ++d) sp[pi[d]] = sc[d];
                const i64 id = encode(sp);
                if (!seen[(size_t)id]) {
                    seen[(size_t)id] = 1;
                    orb.push_back((uint32_t)id);
                }
            }
        }
        orbits.push_back(std::move(orb));
    }
    std::stable_sort(orbits.begin(), orbits.end(), [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) { return a.size() > b.size(); });
    // tiles of a super-tile, natural order; linear tile id as the kernel decodes it
    i64 tmul[MAXN], acc = 1;
    for (int d = 0; d < c.N; ++d) {
        tmul[d] = acc;
        acc *= t.ntiles[d];
    }
    std::vector<uint32_t> list;
    list.reserve((size_t)t.grid);
    const bool interleave = false;  // members of an orbit one after the other (interleaving them measured the same)
    for (const auto& orb : orbits) {
        std::vector<std::vector<uint32_t>> per;  // tiles of each member super-tile
        for (uint32_t sid : orb) {
            i64 sc[MAXN], lo[MAXN], n[MAXN], cnt = 1;
            decode(sid, sc);
            for (int d = 0; d < c.N; ++d) {
                lo[d] = sc[d] << (E[d] - lg[d]);
                n[d] = std::min<i64>((i64)1 << (E[d] - lg[d]), t.ntiles[d] - lo[d]);
                cnt *= n[d];
            }
            per.emplace_back();
            for (i64 q = 0; q < cnt; ++q) {
                i64 r = q, id = 0;
                for (int d = 0; d < c.N; ++d) {
                    id += (lo[d] + r % n[d]) * tmul[d];
                    r /= n[d];
                }
                per.back().push_back((uint32_t)id);
            }
        }
        if (!interleave) {
            for (const auto& v : per) list.insert(list.end(), v.begin(), v.end());
        } else {
            size_t mx = 0;
            for (const auto& v : per) mx = std::max(mx, v.size());
            for (size_t q = 0; q < mx; ++q)
                for (const auto& v : per)
                    if (q < v.size()) list.push_back(v[q]);
        }
    }
    if ((i64)list.size() != t.grid) return;  // cannot happen; keep the natural order if it does
    const i64 cs = (t.grid + NX - 1) / NX;
    t.ord.assign((size_t)(cs * NX), 0xffffffffu);
    for (i64 x = 0; x < NX; ++x)
        for (i64 slot = 0; slot < cs; ++slot) {
            const i64 at = x * cs + slot;
            if (at < t.grid) t.ord[(size_t)(slot * NX + x)] = list[(size_t)at];
        }
    t.ord_groups = (int)orbits.size();
}


// Block order for inputs that are DISTINCT arrays with different unit axes (no orbit structure to exploit): the
// tiles that run at the same time form compact blocks (blk tiles along every tiled dim) instead of a slab that is
// long along dim 0 only, so that every operand -- whatever its unit axis -- has its 32-/64-byte runs completed to
// longer contiguous pieces by tiles that are in flight together (DRAM row locality; option "tile_block").
// blk < 0: run-balanced blocks -- along every dim the block is as many tiles long as it takes for the operand whose unit
// axis that dim is to see ~256 contiguous bytes (a 16 x 8 x 8 x 4 Float64 tile: 2 x 4 x 4 x 8 tiles), about 256 tiles in all.
static void plan_block_order(const Canon& c, TilePlan& t, const int* lg, int blk, bool xcd_runs) {
    if (blk == 0 || blk == 1 || t.grid < 64 || t.grid > ((i64)1 << 22)) return;
    i64 nb[MAXN], blocks = 1, tmul[MAXN], acc = 1, bd[MAXN];
    const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
    for (int d = 0; d < c.N; ++d) bd[d] = blk > 0 ? blk : 1;
    if (blk < 0) {
        i64 prod = 1;
        for (int d = 0; d < c.N; ++d) {
            bool unit = false;
            for (int k = 0; k < c.M; ++k)
                if (c.strides[k][d] == 1 || c.strides[k][d] == -1) unit = true;
            if (unit) bd[d] = std::max<i64>(1, 256 / (((i64)es) << lg[d]));
            bd[d] = std::min<i64>(bd[d], t.ntiles[d]);
            prod *= bd[d];
        }
        for (int guard = 0; prod < 192 && guard < 32; ++guard) {  // fill up to about 256 tiles: grow the shortest extents first
            int best = -1;
            for (int d = 0; d < c.N; ++d)
                if (bd[d] < t.ntiles[d] && (best < 0 || (bd[d] << lg[d]) < (bd[best] << lg[best]))) best = d;
            if (best < 0) break;
            prod = prod / bd[best] * std::min<i64>(bd[best] * 2, t.ntiles[best]);
            bd[best] = std::min<i64>(bd[best] * 2, t.ntiles[best]);
        }
    }
    for (int d = 0; d < c.N; ++d) {
        tmul[d] = acc;
        acc *= t.ntiles[d];
        nb[d] = (t.ntiles[d] + bd[d] - 1) / bd[d];
        blocks *= nb[d];
    }
    std::vector<uint32_t> list;
    list.reserve((size_t)t.grid);
    for (i64 b = 0; b < blocks; ++b) {
        i64 bc[MAXN], r = b, lo[MAXN], n[MAXN], cnt = 1;
        for (int d = 0; d < c.N; ++d) {
            bc[d] = r % nb[d];
            r /= nb[d];
            lo[d] = bc[d] * bd[d];
            n[d] = std::min<i64>(bd[d], t.ntiles[d] - lo[d]);
            cnt *= n[d];
        }
        for (i64 q = 0; q < cnt; ++q) {
            i64 rr = q, id = 0;
            for (int d = 0; d < c.N; ++d) {
                id += (lo[d] + rr % n[d]) * tmul[d];
                rr /= n[d];
            }
            list.push_back((uint32_t)id);
        }
    }
    if ((i64)list.size() != t.grid) return;
    if (xcd_runs) {  // experiment: one contiguous run of the list per XCD (a block then meets in ONE L2)
        const i64 cs = (t.grid + 7) / 8;
        t.ord.assign((size_t)(cs * 8), 0xffffffffu);
        for (i64 x = 0; x < 8; ++x)
            for (i64 slot = 0; slot < cs; ++slot)
                if (x * cs + slot < t.grid) t.ord[(size_t)(slot * 8 + x)] = list[(size_t)(x * cs + slot)];
    } else {
        while (list.size() % 8) list.push_back(0xffffffffu);
        t.ord = std::move(list);
    }
    t.ord_groups = (int)blocks;
}

// ---- orbit planning (FAM_ORBIT) ----------------------------------------------------------------------
// B .= (A .+ A')./2, the 4-way permuted sum, ... : every input is a permuted view of ONE buffer.  The
// classic tiled kernel reads that buffer once per view (through L1: 4 x 8 MiB for the 4-way sum at 32^4);
// here the workgroup that owns a tile also owns its images under the permutation group, loads the buffer
// on those tiles once (natural order, full vectors) and serves every view from LDS.  Measured on MI355X
// (tools/c3_proto.hip, 4-way sum f64): 32^4 6.6 -> 4.7 us, 64^4 83 -> 46 us, 128^4 1.73 -> 1.41 ms.
static bool plan_orbit(const Canon& c, OrbitPlan& o) {
    const Options& opt = options();
    if (!opt.orbit) return false;
    if (c.redop != SMR_RED_NONE || c.mixed || c.bitcopy) return false;
    if (c.M < 3 || c.N < 2 || c.strides[0][0] != 1) return false;
    const int es = dtype_size(c.ct);
    // one buffer, one element type
    for (int k = 2; k < c.M; ++k) {
        if (c.dtype[k] != c.dtype[1]) return false;
        if ((char*)c.base[k] + c.offsets[k] * c.esize[k] != (char*)c.base[1] + c.offsets[1] * c.esize[1]) return false;
    }
    if (c.dtype[0] != c.dtype[1]) return false;
    // identity view: an input with exactly the destination's strides
    o.k0 = -1;
    for (int k = 1; k < c.M && o.k0 < 0; ++k) {
        bool same = true;
        for (int d = 0; d < c.N; ++d) same = same && c.strides[k][d] == c.strides[0][d];
        if (same) o.k0 = k;
    }
    if (o.k0 < 0) return false;
    const i64* s = c.strides[o.k0];
    for (int d = 0; d < c.N; ++d) {
        if (s[d] <= 0) return false;
        for (int e = 0; e < d; ++e)
            if (s[e] == s[d]) return false;
    }
    // pi_k: strides of input k = strides of the identity view permuted
    for (int k = 1; k < c.M; ++k) {
        bool used[MAXN] = {false};
        for (int d = 0; d < c.N; ++d) {
            int hit = -1;
            for (int e = 0; e < c.N && hit < 0; ++e)
                if (!used[e] && s[e] == c.strides[k][d] && c.dims[e] == c.dims[d]) hit = e;
            if (hit < 0) return false;
            used[hit] = true;
            o.pdim[k][d] = hit;
        }
    }
    // the group they generate (closure under composition), identity first
    std::vector<std::array<int, MAXN>> G;
    {
        std::array<int, MAXN> id;
        for (int d = 0; d < MAXN; ++d) id[d] = d;
        G.push_back(id);
        for (size_t head = 0; head < G.size(); ++head)
            for (int k = 1; k < c.M; ++k) {
                std::array<int, MAXN> h = id;
                for (int d = 0; d < c.N; ++d) h[d] = o.pdim[k][G[head][d]];  // pi_k o g
                if (std::find(G.begin(), G.end(), h) == G.end()) {
                    if ((int)G.size() >= MAXG) return false;
                    G.push_back(h);
                }
            }
    }
    o.ng = (int)G.size();
    if (o.ng < 2) return false;
    for (int a = 0; a < o.ng; ++a) {
        for (int d = 0; d < MAXN; ++d) o.gdim[a][d] = G[a][d];
        for (int k = 1; k < c.M; ++k) {
            std::array<int, MAXN> h = G[0];
            for (int d = 0; d < c.N; ++d) h[d] = o.pdim[k][G[a][d]];
            o.slot[a][k] = (int)(std::find(G.begin(), G.end(), h) - G.begin());
        }
        o.slot[a][0] = a;
    }
    // the unit class: dims that are the unit-stride axis of some view
    bool unit[MAXN] = {false};
    int nu = 0;
    for (int a = 0; a < o.ng; ++a)
        if (!unit[G[a][0]]) {
            unit[G[a][0]] = true;
            ++nu;
        }
    // tile edge: the largest power of two that divides the unit dims, fits LDS and leaves enough orbits
    const i64 n0 = c.dims[0];
    i64 others = 1;
    for (int d = 0; d < c.N; ++d)
        if (!unit[d]) others *= c.dims[d];
    const int vmax = std::max(1, 16 / es);
    int vlog = 0;
    while ((1 << vlog) < vmax) ++vlog;
    // 16-byte accesses need aligned operands and vector-multiple strides; element-wise tiles stay <= 1024 elements
    bool vec_ok = vmax > 1;
    for (int d = 1; d < c.N; ++d)
        if (s[d] % vmax) vec_ok = false;
    if ((((uintptr_t)c.base[0] + (uintptr_t)(c.offsets[0] * c.esize[0])) | ((uintptr_t)c.base[o.k0] + (uintptr_t)(c.offsets[o.k0] * c.esize[o.k0]))) % 16) vec_ok = false;
    const int maxbits = (vmax > 1 && !vec_ok) ? 10 : 12;
    int best = -1;
    for (int l = maxbits / nu; l >= 1; --l) {
        if (n0 & (((i64)1 << l) - 1)) continue;
        // LDS: the launch pads a group of order 3 to four slots; "max_lds_bytes" can raise (never lower) the 128 KiB cap
        const size_t slots = o.ng == 3 ? 4 : (size_t)o.ng;
        if (slots * ((size_t)es << (l * nu)) > std::min<size_t>((size_t)160 * 1024, std::max<size_t>((size_t)opt.max_lds_bytes, (size_t)128 * 1024))) continue;
        if (((i64)es << l) < opt.orbit_minrun || l < vlog) continue;  // runs of at least 32 bytes (option orbit_minrun)
        if (l * nu < 8) continue;                        // at least 256 elements per tile (128 lanes x 16 B)
        if (opt.orbit_lg >= 0) {
            if (l == opt.orbit_lg) best = l;
            continue;
        }
        i64 tiles = others;
        for (int u = 0; u < nu; ++u) tiles *= n0 >> l;
        best = l;  // the smallest admissible edge when no edge yields orbit_min orbits
        if (tiles / o.ng >= opt.orbit_min) break;
        // a handful of big workgroups loses against the classic kernel's many small ones (measured: 24^4 f32,
        // 20 orbits of 8^4: 3.96 vs 3.38 us)
        if (l == 1 || (((i64)es << (l - 1)) < opt.orbit_minrun) || l - 1 < vlog || (l - 1) * nu < 8)
            if (tiles / o.ng < opt.orbit_few) best = -1;
    }
    if (best < 0) return false;
    o.tilelog = best * nu;
    if (o.tilelog < 8) return false;  // at least 256 elements per tile (128 lanes x 16 B); smaller: classic kernel
    o.ntiles_total = 1;
    for (int d = 0; d < c.N; ++d) {
        o.lg[d] = unit[d] ? best : 0;
        o.ntiles[d] = c.dims[d] >> o.lg[d];
        o.ntiles_total *= o.ntiles[d];
    }
    for (int d = c.N; d < MAXN; ++d) {
        o.lg[d] = 0;
        o.ntiles[d] = 1;
    }
    if (o.ntiles_total > ((i64)1 << 22)) return false;
    // 16-byte accesses: every non-unit stride a multiple of the vector (alignment is re-checked at launch)
    o.vec = vec_ok ? vmax : 1;
    o.lds_bytes = (o.ng == 3 ? 4 : (size_t)o.ng) * ((size_t)es << o.tilelog);
    {  // 32-bit byte offsets inside a tile
        long double span = 0;
        for (int d = 0; d < c.N; ++d) span += (long double)(((i64)1 << o.lg[d]) - 1) * (long double)s[d] * es;
        if (span >= 2147483648.0L) return false;
        // slot origins travel as 32-bit element offsets
        long double last = 0;
        for (int d = 0; d < c.N; ++d) last += (long double)(c.dims[d] - 1) * (long double)s[d];
        if (last >= 4294967295.0L) return false;
    }

    // orbits of tile coordinates; one root per orbit, listed super-cell by super-cell (2 tiles along every
    // tiled dim: the partner halves of 64-B runs then meet in one XCD's L2), XCD-contiguous runs
    i64 tmul[MAXN], acc = 1;
    for (int d = 0; d < c.N; ++d) {
        tmul[d] = acc;
        acc *= o.ntiles[d];
    }
    std::vector<char> seen((size_t)o.ntiles_total, 0);
    std::vector<uint32_t> list;
    std::vector<uint32_t> cell_of;  // super-cell (linear index, dim 0 fastest) every list entry came from
    i64 ncell[MAXN], cells = 1;
    int sub[MAXN];
    for (int d = 0; d < c.N; ++d) {
        sub[d] = (o.lg[d] > 0 && o.ntiles[d] > 1) ? (int)std::max<i64>(1, std::min<i64>(opt.orbit_group, o.ntiles[d])) : 1;
        ncell[d] = (o.ntiles[d] + sub[d] - 1) / sub[d];
        cells *= ncell[d];
    }
    int nsub = 1;
    for (int d = 0; d < c.N; ++d) nsub *= sub[d];
    for (i64 cell = 0; cell < cells; ++cell) {
        i64 cc[MAXN], r = cell;
        for (int d = 0; d < c.N; ++d) {
            cc[d] = r % ncell[d];
            r /= ncell[d];
        }
        // skewed enumeration (option orbit_skew): consecutive super-cells step along the DIAGONAL of the tiled dims, so
        // that the orbits in flight together differ in every coordinate -- at power-of-two sizes the lines of one cube
        // share a handful of L2 sets (strides 1 KiB / 128 KiB / 16 MiB), and neighbours along one dim share them too
        if (opt.orbit_skew) {
            int first = -1;
            for (int d = 0; d < c.N; ++d)
                if (sub[d] > 1 || o.lg[d] > 0) {
                    if (first < 0) first = d;
                    else cc[d] = (cc[d] + cc[first] * opt.orbit_skew) % ncell[d];
                }
        }
        for (int q = 0; q < nsub; ++q) {
            i64 t[MAXN];
            int qq = q;
            bool inside = true;
            for (int d = 0; d < c.N; ++d) {
                t[d] = cc[d] * sub[d] + qq % sub[d];
                qq /= sub[d];
                if (t[d] >= o.ntiles[d]) inside = false;
            }
            if (!inside) continue;
            i64 root = -1;
            for (int a = 0; a < o.ng; ++a) {
                i64 id = 0;
                for (int d = 0; d < c.N; ++d) id += t[d] * tmul[G[a][d]];
                if (root < 0 || id < root) root = id;
            }
            if (seen[(size_t)root]) continue;
            seen[(size_t)root] = 1;
            list.push_back((uint32_t)root);
            cell_of.push_back((uint32_t)cell);
        }
    }
    o.norbits = (int)list.size();
    o.list = list;
    // ---- workgroups: the tiles of an orbit fill the LDS slots of one workgroup; orbits with fewer distinct tiles share one ------------
    const int NS = o.ng == 3 ? 4 : o.ng;  // a group of order 3 runs in four slots (the fourth repeats the first)
    o.nslots = NS;
    struct Wg {
        uint32_t tile[MAXG];
        uint64_t map;
    };
    auto field = [](int g, int k) { return (unsigned)((g * 8 + k) * 2); };  // k = input index - 1
    std::vector<Wg> wgs;
    std::vector<char> wg_full;      // the workgroup holds ONE orbit of |G| distinct tiles, slot g = g_g . t (PAIR form)
    std::vector<uint32_t> cell_wg;  // super-cell of every workgroup (orbit_deal = 1)
    // pending[s]: orbits of s distinct tiles waiting for company -- (tiles, per-tile reads as indices into the orbit's own tiles)
    struct Part {
        uint32_t tile[MAXG];
        int rd[MAXG][MAXM];
        int n;
    };
    std::vector<Part> pending[MAXG + 1];
    auto emit_packed = [&](int sdist, uint32_t cell) {
        std::vector<Part>& pp = pending[sdist];
        if (pp.empty()) return;
        Wg w;
        w.map = 0;
        int base = 0;
        for (const Part& q : pp) {
            for (int j = 0; j < q.n; ++j) {
                w.tile[base + j] = q.tile[j];
                for (int k = 1; k < c.M; ++k) w.map |= (uint64_t)(base + q.rd[j][k]) << field(base + j, k - 1);
            }
            base += q.n;
        }
        for (int g = base; g < NS; ++g) {  // an incomplete last workgroup: the free slots repeat slot 0 (identical values are stored twice)
            w.tile[g] = w.tile[0];
            for (int k = 1; k < c.M; ++k) w.map |= ((w.map >> field(0, k - 1)) & 3u) << field(g, k - 1);
        }
        wgs.push_back(w);
        wg_full.push_back(0);
        cell_wg.push_back(cell);
        pp.clear();
    };
    for (size_t i = 0; i < list.size(); ++i) {
        i64 tc[MAXN], id = list[i];
        for (int d = 0; d < c.N; ++d) {
            tc[d] = id % o.ntiles[d];
            id /= o.ntiles[d];
        }
        uint32_t ids[MAXG];
        for (int a = 0; a < o.ng; ++a) {
            i64 v = 0;
            for (int d = 0; d < c.N; ++d) v += tc[d] * tmul[G[a][d]];
            ids[a] = (uint32_t)v;
        }
        int first[MAXG], ndist = 0, dix[MAXG];  // dix[a]: index of g_a . t among the distinct tiles
        for (int a = 0; a < o.ng; ++a) {
            dix[a] = -1;
            for (int b = 0; b < a && dix[a] < 0; ++b)
                if (ids[b] == ids[a]) dix[a] = dix[b];
            if (dix[a] < 0) {
                dix[a] = ndist;
                first[ndist++] = a;
            }
        }
        if (ndist == o.ng || !opt.orbit_pack || NS / ndist < 2) {
            Wg w;
            w.map = 0;
            for (int g = 0; g < NS; ++g) {
                const int a = g < o.ng ? g : 0;
                w.tile[g] = ids[a];
                for (int k = 1; k < c.M; ++k) w.map |= (uint64_t)o.slot[a][k] << field(g, k - 1);
            }
            wgs.push_back(w);
            wg_full.push_back(ndist == o.ng && NS == o.ng ? 1 : 0);
            cell_wg.push_back(cell_of[i]);
            continue;
        }
        Part q;
        q.n = ndist;
        for (int j = 0; j < ndist; ++j) {
            q.tile[j] = ids[first[j]];
            for (int k = 1; k < c.M; ++k) q.rd[j][k] = dix[o.slot[first[j]][k]];
        }
        pending[ndist].push_back(q);
        if ((int)pending[ndist].size() * ndist + ndist > NS) emit_packed(ndist, cell_of[i]);
    }
    for (int sd = 1; sd <= MAXG; ++sd) emit_packed(sd, cell_wg.empty() ? 0u : cell_wg.back());
    constexpr int NX = 8;
    // ---- PAIR form: two slot sets per workgroup, unit-axis neighbours first ----------------------------------------------------------
    o.pair_ok = false;
    {
        int ntd = 0;
        bool cubes = true;
        for (int d = 0; d < c.N; ++d)
            if (o.lg[d] > 0) {
                ++ntd;
                if (o.lg[d] != 2) cubes = false;
            }
        if (opt.orbit_pair && o.ng == 4 && NS == 4 && ntd == 4 && cubes && o.tilelog == 8 && es == 8 && o.vec == 2 && o.lg[0] == 2 && opt.orbit_deal != 1 && wgs.size() >= 16) {
            std::vector<int> owner((size_t)o.ntiles_total, -1);
            for (size_t i = 0; i < wgs.size(); ++i)
                if (wg_full[i])
                    for (int g = 0; g < NS; ++g) owner[wgs[i].tile[g]] = (int)i;
            std::vector<char> used(wgs.size(), 0);
            std::vector<uint32_t> pt;
            std::vector<uint64_t> pm;
            auto push_set = [&](const uint32_t* tile, uint64_t map) {
                for (int g = 0; g < NS; ++g) pt.push_back(tile[g]);
                pm.push_back(map);
            };
            uint64_t stdmap = 0;
            for (int g = 0; g < NS; ++g)
                for (int k = 1; k < c.M; ++k) stdmap |= (uint64_t)o.slot[g][k] << field(g, k - 1);
            // the slot sets of an orbit re-based at tile `id`: slot g = g_g . id (any tile of a full orbit can be its slot 0)
            auto rebased = [&](uint32_t id0, uint32_t* bt) {
                i64 nc[MAXN], id = id0;
                for (int d = 0; d < c.N; ++d) {
                    nc[d] = id % o.ntiles[d];
                    id /= o.ntiles[d];
                }
                for (int g = 0; g < NS; ++g) {
                    i64 v = 0;
                    for (int d = 0; d < c.N; ++d) v += nc[d] * tmul[G[g][d]];
                    bt[g] = (uint32_t)v;
                }
            };
            // super-cell by super-cell, in the cell's own tile order: the tile with an even unit-axis coordinate and its neighbour, both
            // taken as slot 0 of their orbits -- the eight pairs of a super-cell then run back to back on one XCD and cover its four
            // rotated images completely (pairing an orbit's smallest tile with whatever sits next to it scattered the pairs: 4.72 us
            // per launch against 4.27 with this list, tools/orbit16_probe.hip)
            for (i64 cell = 0; cell < cells; ++cell) {
                i64 cc[MAXN], r = cell;
                for (int d = 0; d < c.N; ++d) {
                    cc[d] = r % ncell[d];
                    r /= ncell[d];
                }
                for (int q = 0; q < nsub; ++q) {
                    i64 t[MAXN];
                    int qq = q;
                    bool inside = true;
                    for (int d = 0; d < c.N; ++d) {
                        t[d] = cc[d] * sub[d] + qq % sub[d];
                        qq /= sub[d];
                        if (t[d] >= o.ntiles[d]) inside = false;
                    }
                    if (!inside || (t[0] & 1) || t[0] + 1 >= o.ntiles[0]) continue;
                    i64 id = 0;
                    for (int d = 0; d < c.N; ++d) id += t[d] * tmul[d];
                    const int i = owner[(size_t)id], j = owner[(size_t)id + 1];  // (tmul[0] == 1: the unit-axis neighbour is id + 1)
                    if (i < 0 || j < 0 || i == j || used[(size_t)i] || used[(size_t)j]) continue;
                    uint32_t ta[MAXG], tb[MAXG];
                    rebased((uint32_t)id, ta);
                    rebased((uint32_t)id + 1, tb);
                    push_set(ta, stdmap);
                    push_set(tb, stdmap);
                    used[(size_t)i] = used[(size_t)j] = 1;
                }
            }
            std::vector<size_t> loose;
            for (size_t i = 0; i < wgs.size(); ++i)
                if (!used[i]) loose.push_back(i);
            for (size_t q = 0; q < loose.size(); q += 2) {  // what found no neighbour (diagonals, packed orbits): any two
                const Wg& a = wgs[loose[q]];
                const Wg& b = wgs[q + 1 < loose.size() ? loose[q + 1] : loose[q]];  // an odd one out runs twice (same values stored twice)
                push_set(a.tile, a.map);
                push_set(b.tile, b.map);
            }
            // one contiguous run of the list per XCD, as below
            const size_t np = pm.size() / 2, pcs = (np + NX - 1) / NX;
            o.ptile.assign(pcs * NX * 8, 0xffffffffu);
            o.pmap.assign(pcs * NX * 2, 0);
            for (size_t x = 0; x < (size_t)NX; ++x)
                for (size_t sl = 0; sl < pcs; ++sl)
                    if (x * pcs + sl < np) {
                        const size_t src = x * pcs + sl, dst = sl * NX + x;
                        for (int q = 0; q < 8; ++q) o.ptile[dst * 8 + q] = pt[src * 8 + q];
                        o.pmap[dst * 2] = pm[src * 2];
                        o.pmap[dst * 2 + 1] = pm[src * 2 + 1];
                    }
            o.pair_ok = true;
        }
    }
    auto place = [&](size_t pos, const Wg& w) {
        for (int g = 0; g < NS; ++g) o.wtile[pos * NS + g] = w.tile[g];
        o.wmap[pos] = w.map;
    };
    if (opt.orbit_deal == 1) {
        // experiment (round 5): super-cell c runs on XCD c mod 8 -- consecutive super-cells step along dim 0, so at any time the eight
        // XCDs work on eight neighbours along the buffer's unit axis (whole DRAM pages chip-wide) while the partner halves of every
        // line still meet inside one XCD's L2
        std::vector<size_t> perx[NX];
        for (size_t i = 0; i < wgs.size(); ++i) perx[cell_wg[i] % NX].push_back(i);
        size_t cs2 = 0;
        for (int x = 0; x < NX; ++x) cs2 = std::max(cs2, perx[x].size());
        o.wtile.assign(cs2 * NX * NS, 0xffffffffu);
        o.wmap.assign(cs2 * NX, 0);
        for (int x = 0; x < NX; ++x)
            for (size_t sl = 0; sl < perx[x].size(); ++sl) place(sl * NX + x, wgs[perx[x][sl]]);
        return true;
    }
    const size_t cs = (wgs.size() + NX - 1) / NX;
    o.wtile.assign(cs * NX * NS, 0xffffffffu);
    o.wmap.assign(cs * NX, 0);
    for (size_t x = 0; x < (size_t)NX; ++x)
        for (size_t sl = 0; sl < cs; ++sl)
            if (x * cs + sl < wgs.size()) place(sl * NX + x, wgs[x * cs + sl]);
    return true;
}

// the dim along which operand k's memory continues after a (short) dim 0: stride == extent of dim 0, or -1
static int continuation_of_dim0(const Canon& c, int k) {
    if (c.strides[k][0] != 1) return -1;
    for (int d = 1; d < c.N; ++d)
        if (c.dims[d] > 1 && std::llabs(c.strides[k][d]) == c.dims[0]) return d;
    return -1;
}

// A short common unit axis (a physical index of 2..4, the colour channel of an image) with operands that continue
// along DIFFERENT dims behind it -- permutedims of (2,128,128) to (1,3,2) -- is a transposition one level up: rows
// of dims[0] elements are all the STREAM family would move.  Such inputs are staged like transposed ones.
static bool short_dim0_transposition(const Canon& c, int k, int es) {
    if (c.N < 3 || c.dims[0] * es >= 64 || c.strides[0][0] != 1 || c.strides[k][0] != 1) return false;
    const int d0 = continuation_of_dim0(c, 0), dk = continuation_of_dim0(c, k);
    return d0 > 0 && dk > 0 && d0 != dk;
}

// The dim along which operand k's memory advances least: its unit-stride dim, or -- a stepped range, A[1:3:end, :] seen through a
// permutation -- a dim whose step is 2..4 elements: lanes laid along it still share cache lines (a third of each at step 3), where
// lanes laid along the destination's dim 0 would touch one line each (round 4; `step-3 cols transpose-add` 0.6 TB/s in r2 / r3).
static int near_axis(const Canon& c, int k) {
    int q = fast_axis(c, k);
    if (q >= 0) return q;
    i64 best = 5;
    for (int i = 0; i < c.N; ++i) {
        const i64 s = std::llabs(c.strides[k][i]);
        if (c.dims[i] > 1 && s >= 2 && s < best) {
            best = s;
            q = i;
        }
    }
    return q;
}

static bool plan_tiles(const Canon& c, TilePlan& t) {
    if (c.redop != SMR_RED_NONE) return false;
    if (c.N < 2 || c.strides[0][0] != 1) return false;
    const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
    // which inputs need staging: unit-stride axis exists and differs from the destination's
    bool axis_used[MAXN] = {false};
    axis_used[0] = true;
    int nst = 0;
    t.staged[0] = -1;
    for (int k = 1; k < c.M; ++k) {
        int q = near_axis(c, k);
        t.staged[k] = -1;
        if (q > 0 && c.strides[k][0] != 0 && std::llabs(c.strides[k][0]) != 1) {
            t.staged[k] = nst++;
            axis_used[q] = true;
        } else if (q == 0 && short_dim0_transposition(c, k, es)) {
            t.staged[k] = nst++;
            axis_used[continuation_of_dim0(c, 0)] = true;
            axis_used[continuation_of_dim0(c, k)] = true;
        } else if (q > 0 && c.strides[k][0] == 0) {
            // broadcast along dim 0 with another unit axis: reads are wave-uniform along
            // dim 0 anyway; leave it direct
        }
    }
    if (nst == 0) return false;
    t.nstaged = nst;
    int axes[MAXN], na = 0;
    for (int i = 0; i < c.N; ++i)
        if (axis_used[i]) axes[na++] = i;
    // target contiguous run per axis
    const Options& o = options();
    i64 runbytes = (na == 2) ? 256 : (na == 3 ? 128 : 64);
    // The tiled kernel is specialised for 1024-element tiles (256 lanes x 4 elements): measured
    // on MI355X, 1024-element tiles beat 4096-element ones at 32^4 (more workgroups in flight)
    // and tie at 128^4.
    // Three or more distinct unit axes and a big problem: 4096-element tiles on 1024 threads (the
    // same 4 elements per lane) keep every operand's contiguous run at >= 64 B.  Measured on the
    // 4-way permuted sum: 32^4 8.0 vs 8.4 us, 64^4 141 vs 175 us, 128^4 3.24 vs 4.64 ms (needs at
    // least one big tile per CU).
    int tl_cap = 10;
    bool big_transpose = false;
    if (o.tile_log2 == 12) tl_cap = 12;
    // true-HBM sized transposes (two unit axes, >= 1 GiB moved, 8-/16-byte elements): 128 x 32 tiles on 1024 lanes --
    // 1-KiB non-temporal write runs along the destination's unit axis, 256-byte read runs along the source's.
    // Measured (tools/perm_tiles2.py, Float64): permutedims! 128^4 32x32 940, 64x64 798, 128x32 792 us; transposes
    // 16384^2 840 / 778 / 719 us, 8192^2 270 (218 persistent) / 215 / 206 us; sizes that are not powers of two
    // (4000^2, 12000^2) tie.  Float32 stays with 32 x 32 (128^4: 451 vs 480 / 535 us)
    else if (o.tile_log2 == 0 && na == 2 && nst == 1 && es >= 8 && (long double)c.total * es * 2 >= 1073741824.0L) {
        // ... only when the 128 x 32 tile is (nearly) filled: on 96^4 / 144^4 the 128-wide tile idles 25 / 44 % of its
        // lanes (round 3: 144^4 3.47 TB/s with it)
        auto util = [&](int d, i64 e) { return (long double)c.dims[d] / (long double)(((c.dims[d] + e - 1) / e) * e); };
        if (util(axes[0], 128) * util(axes[1], 32) >= 0.9L) {
            tl_cap = 12;
            runbytes = 32 * es;
            big_transpose = true;
        }
    }
    else if (o.tile_log2 == 0 && na >= 3 && (size_t)nst * 4096 * es <= (size_t)128 * 1024 && c.total >= (i64)4096 * 256) {
        bool fits = true;  // every axis must be able to reach its share of the 12 bits
        int bits = 0;
        for (int a = 0; a < na; ++a) bits += std::min(nextpow2_log(c.dims[axes[a]]), 12);
        if (bits < 12) fits = false;
        // one round of big tiles (<= 256 CUs, 1024 lanes each) or a long persistent work list: big;
        // in between (measured 48^4: 25.8 vs 32.0 us, 64^4: 73.5 vs 83.3 us) several small workgroups per
        // CU overlap better than a few rounds of one big workgroup per CU
        const i64 nbig = c.total >> 12;
        // ... when the inputs are views of ONE buffer (they hit in L2).  DISTINCT arrays stream from HBM / the
        // Infinity Cache and want the longer runs of the big tiles much earlier (round 3, tools/cliff_ab.py, Float64:
        // A .+ perm(C) .* perm'(D) 48^4 35.7 -> 26.6 us, 64^4 117.8 -> 105.5 us; four arrays with four unit axes:
        // 64^4 177 -> 157 us, 96^4 888 -> 755 us together with the block order below, 48^4 stays with small tiles)
        bool one_buffer = false;
        for (int j = 1; j < c.M; ++j)
            for (int k = j + 1; k < c.M; ++k)
                if ((char*)c.base[j] + c.offsets[j] * c.esize[j] == (char*)c.base[k] + c.offsets[k] * c.esize[k]) one_buffer = true;
        if (one_buffer) {
            if (nbig > 256 && nbig < 8192) fits = false;
        } else if (nbig > 256 && nbig < (na >= 4 ? 2048 : 257)) {
            fits = false;
        }
        if (fits) tl_cap = 12;
    }
    if ((size_t)nst * ((size_t)1 << tl_cap) * es > std::max<size_t>((size_t)o.max_lds_bytes, tl_cap == 12 ? (size_t)128 * 1024 : 0)) return false;
    int lg[MAXN] = {0};
    int total = 0;
    // grow the axes round-robin towards the run target
    int want = std::max(1, nextpow2_log(std::max<i64>(1, runbytes / es)));
    bool forced = false;
    for (int i = 0; i < c.N; ++i)
        if (o.tile_lg[i] >= 0) forced = true;
    if (forced) {  // tuning override: exact per-dim log2 extents
        for (int i = 0; i < c.N; ++i) {
            lg[i] = (int)std::max<i64>(0, o.tile_lg[i]);
            while (lg[i] > 0 && ((i64)1 << (lg[i] - 1)) >= c.dims[i]) --lg[i];
            total += lg[i];
        }
        tl_cap = total;
    }
    if (!forced && tl_cap == 12) {
        // big tiles: give the destination axis a full 128-B line first (stores and the direct
        // inputs then move whole lines; measured on the 4-way sum at 32^4: 16x8x8x4 7.7 us,
        // 16x8x4x8 7.6 us, 8x8x8x8 8.0 us), the other axes share the rest
        const int want0 = big_transpose ? 7 : std::max(1, nextpow2_log(std::max<i64>(1, 128 / es)));
        while (lg[0] < want0 && ((i64)1 << lg[0]) < c.dims[0] && total < tl_cap) {
            ++lg[0];
            ++total;
        }
    }
    bool grew = !forced;
    while (grew && total < tl_cap) {
        grew = false;
        for (int a = 0; a < na && total < tl_cap; ++a) {
            int i = axes[a];
            if (i == 0 && tl_cap == 12 && lg[0] >= want) continue;  // already served above
            if (lg[i] < want && ((i64)1 << lg[i]) < c.dims[i]) {
                ++lg[i];
                ++total;
                grew = true;
            }
        }
    }
    // an operand whose unit axis is shorter than the run target (3 x H x W images, physical indices of 2..4 in front):
    // its memory run continues along the dim whose stride equals that extent -- grow that one before the
    // destination-order widening below (permutedims (3,1000,700) -> (700,1000,3): 256 x 4 tiles read 24-byte
    // runs, 32 x 8 x 4 tiles read 192-byte runs)
    if (!forced) {
        for (int k = 0; k < c.M; ++k) {
            if (k > 0 && t.staged[k] < 0) continue;
            const int q = (k == 0) ? 0 : fast_axis(c, k);
            if (q < 0 || ((i64)1 << lg[q]) < c.dims[q] || c.dims[q] >= ((i64)1 << want)) continue;
            int v = -1;
            for (int d = 0; d < c.N; ++d)
                if (d != q && std::llabs(c.strides[k][d]) == c.dims[q]) v = d;
            if (v < 0) continue;
            while (total < tl_cap && (c.dims[q] << lg[v]) < ((i64)1 << want) && ((i64)1 << lg[v]) < c.dims[v]) {
                ++lg[v];
                ++total;
            }
        }
    }
    // small problems / short axes: widen the tile with the remaining dims (destination order)
    // until a block has at least 1024 elements of work
    int minlog = forced ? 0 : tl_cap;
    for (int i = 0; i < c.N && total < minlog; ++i)
        while (total < minlog && ((i64)1 << lg[i]) < c.dims[i]) {
            ++lg[i];
            ++total;
        }
    if (total != 10 && total != 12) return false;  // smaller problems go to the generic family
    t.nt = 0;
    t.tilelog = total;
    for (int i = 0; i < c.N; ++i)
        if (lg[i] > 0) {
            if (t.nt >= 5) return false;
            t.tdim[t.nt] = i;
            t.tlog[t.nt] = lg[i];
            ++t.nt;
        }
    if (t.tdim[0] != 0) return false;
    // enumeration order per staged input: its own stride order over the tiled dims
    for (int k = 1; k < c.M; ++k) {
        int idx[MAXN];
        for (int j = 0; j < t.nt; ++j) idx[j] = j;
        if (t.staged[k] >= 0)
            std::stable_sort(idx, idx + t.nt, [&](int a, int b) {
                i64 sa = std::llabs(c.strides[k][t.tdim[a]]), sb = std::llabs(c.strides[k][t.tdim[b]]);
                // zero strides last: they do not move the address
                if ((sa == 0) != (sb == 0)) return sb == 0;
                return sa < sb;
            });
        for (int j = 0; j < t.nt; ++j) t.order[k][j] = idx[j];
    }
    for (int j = 0; j < t.nt; ++j) t.order[0][j] = j;
    t.threads = (total == 12) ? 1024 : 256;  // (2048 elements on 512 lanes measured: never the best)
    t.grid = 1;
    for (int i = 0; i < c.N; ++i) {
        i64 e = (i64)1 << lg[i];
        t.ntiles[i] = (c.dims[i] + e - 1) / e;
        t.grid *= t.ntiles[i];
    }
    t.lds_bytes = (size_t)nst * ((size_t)1 << total) * es;
    if (t.grid > 0x7fffffffLL) return false;
    // Order of the grid dims (which tile coordinate varies fastest with the workgroup id).  Canonical order = by destination
    // stride: consecutive workgroups continue the destination's memory.  For HBM-sized transposing copies whose input's unit axis is
    // cut into several tiles (permutedims!(B, A, (4,3,2,1)) of 128^4: 128 x 32 tiles, four tiles along A's rows of 1 KiB) the tile
    // index along THAT axis goes second: the ~512 workgroups resident at once then cover whole rows of the input as well as
    // contiguous runs of the destination, instead of quarter rows of four different input rows -- tools/xpose_proto.hip: 843 -> 723
    // us for the same tile shape.  In the library (tools/gorder_ab.py, profiles/r05_gorder_ab.txt, Float64): 144^4 (4,3,2,1) 1336 ->
    // 1182 us, (256,128,128,64) reversed 879 -> 811, (64,64,64,64,16) reversed 834 -> 791, Float32 128^4 457 -> 434; neutral for 2-D
    // transposes and when the input's axis is one tile; so: every transposing copy of >= 512 MiB with one staged input and rank >= 3.
    for (int i = 0; i < MAXN; ++i) t.gorder[i] = i < c.N ? i : -1;
    {
        const i64 mode = o.tiled_gorder;
        const bool big = c.algbytes >= ((i64)512 << 20);
        if ((mode > 0 || (mode < 0 && big && na == 2 && nst == 1)) && c.N >= 3) {
            // staged inputs' unit axes that are tiled AND split into several tiles, moved to position 1 (behind the fastest grid dim)
            int order[MAXN], n = 0, first = -1;
            for (int d = 0; d < c.N; ++d)
                if (t.ntiles[d] > 1) {
                    first = d;
                    break;
                }
            if (first >= 0) {
                order[n++] = first;
                for (int k = 1; k < c.M; ++k) {
                    if (t.staged[k] < 0) continue;
                    const int q = fast_axis(c, k);
                    if (q >= 0 && q != first && lg[q] > 0 && t.ntiles[q] > 1) {
                        bool have = false;
                        for (int i = 0; i < n; ++i) have = have || order[i] == q;
                        if (!have) order[n++] = q;
                    }
                }
                for (int d = 0; d < c.N; ++d) {
                    bool have = false;
                    for (int i = 0; i < n; ++i) have = have || order[i] == d;
                    if (!have) order[n++] = d;
                }
                for (int i = 0; i < c.N; ++i) t.gorder[i] = order[i];
            }
        }
    }
    t.ord.clear();
    t.ord_groups = 0;
    if (o.tile_order) plan_tile_order(c, t, lg);
    // 128 x 32 transposing tiles run one-shot: measured round 3 (tools/perm_block_ab.py, Float64) permutedims! 128^4
    // 826 -> 791 us, (2,3,4,1) 866 -> 745 us, (3,4,1,2) 801 -> 699 us, transpose 16384^2 859 -> 719 us, 8192^2 / 12000^2 tie
    t.no_persist = big_transpose;
    if (t.ord.empty() && (na >= 3 || (o.tile_block_min_axes <= 2 && na >= 2)) && o.tile_block != 0 && (o.tile_block > 0 || o.tile_block == -2 || t.grid >= 1024)) {
        // one contiguous run of the list per XCD while the operands fit the Infinity Cache (a block's partner pieces meet in ONE
        // L2: 48^4 45.0 -> 38.5 us); round-robin once they stream from HBM (64^4: 157 vs 175 us)
        const bool xcd_runs = o.tile_block_xcd > 0 || (o.tile_block_xcd < 0 && c.algbytes <= ((i64)256 << 20));
        plan_block_order(c, t, lg, o.tile_block > 0 ? (int)o.tile_block : (o.tile_block == -2 ? -1 : 4), xcd_runs);
        t.no_persist = t.no_persist || !t.ord.empty();  // measured: the one-shot form wins on block-ordered lists (128^4: 2715 vs 2773 us)
    }
    if (!t.ord.empty())  // work lists hold linear tile ids in canonical grid order
        for (int i = 0; i < MAXN; ++i) t.gorder[i] = i < c.N ? i : -1;
    return true;
}

// ---- flat planning (FAM_FLAT) -----------------------------------------------------------------------------
// A unary transposing map in which one side's memory run consists of short leading dims with extents that are not powers of
// two (an image's 3 channels, a physical index of 3 in a tensor network): smr_k_flat.hip addresses that side through the
// flattened run.  side = 0: the destination is flat, 1: the input.
// The FLAT forms take ONE input with a layout of its own (it crosses LDS); any further input must have exactly the destination's
// strides (C .= beta .* C .+ alpha .* permutedims(A, p)).  Returns that input's operand index, or -1.
static int flat_transposed_input(const Canon& c) {
    int kt = -1;
    for (int k = 1; k < c.M; ++k) {
        bool same = true;
        for (int d = 0; d < c.N; ++d)
            if (c.strides[k][d] != c.strides[0][d]) same = false;
        if (same) continue;
        if (kt >= 0) return -1;
        kt = k;
    }
    return kt;
}

static bool plan_flat_side(const Canon& c, FlatPlan& f, int side, int es) {
    const int kf = side == 0 ? 0 : f.kt, kl = side == 0 ? f.kt : 0;
    const i64* sf = c.strides[kf];
    const i64* sl = c.strides[kl];
    // leading dim of the flat side
    int lead = -1;
    for (int d = 0; d < c.N; ++d)
        if (sf[d] == 1 && c.dims[d] > 1) lead = d;
    if (lead < 0) return false;
    const i64 e0 = c.dims[lead];
    // long or power-of-two leading dims: the tiled family's ground -- except (round 6, option flat_wide) whole rows of 65..128
    // elements whose byte length is not a multiple of the 128-byte line: there every 256-byte tile row of TILED is three partial
    // lines, while rows taken whole make the tile one contiguous, line-aligned piece of this side's memory
    // Measured (tools/ragged_family_ab.py, profiles/r06_flat_wide_ab.txt): this form's phases are loops of dependent loads, so at
    // 11 MiB it loses ((7200,100) 5.4 against 4.4 us) and it must not take leads of 65 elements away from the two-sided form
    // ((257,129,65) 13.3 against 9.9 us); on tall matrices of 32 MiB and more (per array) it wins: (100,100000) Float32 23.2 -> 15.1 us,
    // Float64 34.9 -> 28.5, (100000,100) Float32 21.9 -> 17.4.  flat_wide = 2 forces the form wherever it applies (tests).
    const i64 fw = options().flat_wide;
    const bool wide = fw && e0 * es >= 128 && e0 <= 128 && (e0 * es) % 128 != 0 && (fw >= 2 || (c.N == 2 && c.total * es >= ((i64)32 << 20)));
    if ((e0 * es >= 128 && !wide) || (e0 & (e0 - 1)) == 0) return false;
    // unit axis of the line side: not the lead ...
    int q = -1;
    for (int d = 0; d < c.N; ++d)
        if (d != lead && sl[d] == 1 && c.dims[d] * es >= 64) q = d;   // line-side runs of at least 64 bytes
    // ... or the lead itself, continued by a dim of stride e0: a transposition of e0-element groups ((3,W,H) -> (3,H,W))
    f.lshare = false;
    if (q < 0 && sl[lead] == 1) {
        for (int d = 0; d < c.N; ++d)
            if (d != lead && sl[d] == e0 && sf[d] != e0 && c.dims[d] * e0 * es >= 64) q = d;
        f.lshare = q >= 0;
    }
    if (q < 0) return false;
    if (!f.lshare && (sl[lead] == 1 || sl[lead] == -1)) return false;
    for (int d = 0; d < MAXN; ++d) f.ingroup[d] = false;
    f.ingroup[lead] = true;
    i64 R = e0;
    int p = -1;
    // take further contiguous dims whole while the run stays short, then tile the next one
    for (;;) {
        int nxt = -1;
        for (int d = 0; d < c.N; ++d)
            if (!f.ingroup[d] && d != q && sf[d] == R && c.dims[d] > 1) nxt = d;
        if (nxt < 0) break;
        if (!f.lshare && R * es < 192 && R * c.dims[nxt] <= FLAT_GROUP_MAX && R * c.dims[nxt] * es <= 512) {
            f.ingroup[nxt] = true;
            R *= c.dims[nxt];
            continue;
        }
        p = nxt;
        break;
    }
    if (R > (wide ? 128 : 64)) return false;
    int tplog = 0;
    if (p >= 0)
        while ((R << tplog) * es < 256 && ((i64)1 << tplog) < c.dims[p] && (R << (tplog + 1)) <= 128) ++tplog;
    const i64 L = R << tplog;
    // planar <-> interleaved (NCHW <-> NHWC with 3 channels): the flat side continues along q ITSELF
    f.fuse = p < 0 && sf[q] == R && !f.lshare;
    if (f.lshare && p < 0) return false;
    const int vmax = std::max(1, 16 / es);
    if (!f.fuse) {
        if (L * es < 96 || L > 128) return false;      // no run worth flattening
        if (L / vmax < 2) return false;
    }
    f.dir = side;
    f.R = (int)R;
    f.tplog = tplog;
    f.tqlog = es >= 8 ? 5 : 6;
    if (f.fuse) {  // the whole R x TQ tile is one run: up to 2048 elements per workgroup (3 channels: 512 pixels)
        f.tqlog = 5;
        while (f.tqlog < 9 && (R << (f.tqlog + 1)) <= 2048 && ((i64)1 << f.tqlog) < c.dims[q]) ++f.tqlog;
    }
    f.p = p;
    f.q = q;
    // line-side offsets of the leading index r (mixed radix over the group dims in flat-side stride order)
    std::vector<int> order;
    for (int d = 0; d < c.N; ++d)
        if (f.ingroup[d]) order.push_back(d);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return sf[x] < sf[y]; });
    for (i64 r = 0; r < R; ++r) {
        i64 rem = r, off = 0;
        for (int d : order) {
            off += (rem % c.dims[d]) * sl[d];
            rem /= c.dims[d];
        }
        if (off > 2147483647LL || off < -2147483647LL) return false;
        f.roff[r] = (int32_t)off;
    }
    return true;
}

static bool plan_flat(const Canon& c, FlatPlan& f) {
    const Options& o = options();
    if (!o.flat || c.redop != SMR_RED_NONE || c.M < 2 || c.mixed || c.N < 2) return false;
    const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
    if (es < 4) return false;
    f.kt = flat_transposed_input(c);
    if (f.kt < 0) return false;
    if (c.total < 65536) return false;  // small boxes: generic / tiled are fine and launch-bound anyway
    for (int d = 0; d < c.N; ++d)
        if (c.strides[0][d] <= 0) return false;
    return plan_flat_side(c, f, 0, es) || plan_flat_side(c, f, 1, es);
}

// Batched FLAT form: see FlatBPlan (smr_internal.h).  Conditions: one input; the destination's first g >= 2 dims are dense
// (strides 1, n0, n0 n1, ...) with P = n0 ... n_{g-1} <= 512 elements and at most 4 KiB; the input's strides over those dims are a
// dense layout of the same P elements in another order; dim g has stride P in the destination (consecutive blocks are adjacent
// there).  When the input's blocks follow one another the same way and the dims behind agree, a chunk is contiguous on both sides;
// otherwise (blocks of at least 256 bytes) the input side is read block by block.
static bool plan_flatb(const Canon& c, FlatBPlan& f) {
    const Options& o = options();
    f.on = false;
    if (!o.flatb || !o.flat || c.redop != SMR_RED_NONE || c.M != 2 || c.mixed || c.N < 3) return false;
    const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
    if (es < 4 || c.total < 65536) return false;
    if (c.strides[0][0] != 1) return false;
    // the longest dense destination prefix that stays within the size limits
    int g = 0;
    i64 P = 1;
    while (g < c.N - 1 && c.strides[0][g] == P && P * c.dims[g] <= FLATB_MAXP && P * c.dims[g] * es <= 4096) {
        P *= c.dims[g];
        ++g;
    }
    for (; g >= 2; P /= c.dims[g - 1], --g) {
        // the input over dims 0..g-1: a dense permutation of the same block?
        int ord[MAXN];
        for (int i = 0; i < g; ++i) ord[i] = i;
        std::sort(ord, ord + g, [&](int a, int b) { return c.strides[1][a] < c.strides[1][b]; });
        i64 run = 1;
        bool dense = true, moved = false;
        for (int i = 0; i < g && dense; ++i) {
            if (c.strides[1][ord[i]] != run) dense = false;
            run *= c.dims[ord[i]];
            if (ord[i] != i) moved = true;
        }
        if (!dense || !moved) continue;
        // blocks whose extents are all powers of two are TILED's home ground ((2,128,2,128,8) permutations: 2.6 us there, 4.0-5.5 here)
        bool pow2 = true;
        for (int i = 0; i < g; ++i)
            if (c.dims[i] & (c.dims[i] - 1)) pow2 = false;
        if (pow2) continue;
        // the destination's blocks follow one another along dim g; the input's blocks may sit anywhere (a batch grid that is permuted
        // as well: (9,11,300,300) -> (11,9,300',300)): its side then moves in whole blocks, which must not be tiny
        if (c.strides[0][g] != P) continue;
        bool same = c.strides[1][g] == P;
        for (int d = g + 1; d < c.N; ++d)
            if (c.strides[0][d] != c.strides[1][d]) same = false;
        if (!same && (P * es < 256 || c.strides[1][g] < P)) continue;
        f.g = g;
        f.P = (int)P;
        // K blocks per workgroup: about 4096 elements (32 KiB of Float64 in LDS), an even count so that 16-byte vectors line up
        i64 K = std::max<i64>(1, 4096 / P);
        K = std::min<i64>(K, c.dims[g]);
        if (K > 1) K &= ~(i64)1;
        f.K = (int)K;
        for (i64 r = 0; r < P; ++r) {
            i64 rem = r, off = 0;
            for (int d = 0; d < g; ++d) {
                off += (rem % c.dims[d]) * c.strides[1][d];
                rem /= c.dims[d];
            }
            f.srcoff[r] = (uint16_t)off;
        }
        f.on = true;
        return true;
    }
    return false;
}

// Two-sided FLAT form: the unit-stride dims of BOTH sides are short (under 256 bytes) and at least one of them is not a
// power of two (or under 32 bytes): no power-of-two tile fits either side -- (17,33,65,31) reversed runs a 32 x 32 tile over a
// 31 x 17 face with 51 % of its lanes.  Each side's run = its leading dims taken whole while the run stays short + a tile of the
// next contiguous dim, about flat2_bytes long; the runs share no dim (smr_k_flat.hip, flat2_body).
static bool plan_flat2(const Canon& c, Flat2Plan& f) {
    const Options& o = options();
    f.on = false;
    if (!o.flat2 || !o.flat || c.redop != SMR_RED_NONE || c.M < 2 || c.mixed || c.N < 2) return false;
    const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
    if (es < 4 || c.total < 65536) return false;
    f.kt = flat_transposed_input(c);
    if (f.kt < 0) return false;
    const i64* st[2] = {c.strides[0], c.strides[f.kt]};  // side 0 = destination, side 1 = the transposed input
    int lead[2] = {-1, -1};
    for (int s = 0; s < 2; ++s)
        for (int d = 0; d < c.N; ++d) {
            if (st[s][d] <= 0) return false;
            if (st[s][d] == 1 && c.dims[d] > 1) lead[s] = d;
        }
    if (lead[0] < 0 || lead[1] < 0 || lead[0] == lead[1]) return false;
    // Round 4: a LONG unit-stride dim that 32-wide power-of-two tiles fit badly (100 -> 3 x 32 + 4: a quarter of the last tile's lanes,
    // 257 -> 8 x 32 + 1) is cut EVENLY instead -- its side's run is then a tile of the unit dim itself (R = 1, p = the lead,
    // TP = extent / ceil(extent / ~48)): (100,90,80) -> tiles of 34 x 30, no lane idles.
    // Measured (tools/flat2_long_ab.py, profiles/r04_flat2_long_ab.txt, Float64): (257,129,65) 12.3-14.0 -> 9.5-10.1 us, (17,33,65,31)
    // (3,2,0,1) 9.0 -> 6.5 us, (999,1001) 7.2 -> 6.4 us; but (100,90,80) (6 MB) 5.5 -> 6.4 us and well-filled tiles ((200,300,70),
    // (1400,1500), (4000,4100), every power of two) stay ahead in TILED -- so: the padded tiles would be under flat2_long % full and
    // the array has at least 8 MiB, or the OTHER side's lead is short and awkward anyway.
    bool awkward = false, cutlead[2] = {false, false}, novec = false, odd_short = false;
    long double fill = 1;
    const i64 vlen = std::max<i64>(1, 16 / es);
    for (int s = 0; s < 2; ++s) {
        const i64 e = c.dims[lead[s]];
        if (e * es >= o.flat2_lead_bytes) {
            if (!o.flat2_long) return false;
            cutlead[s] = true;
            fill *= (long double)e / (long double)((e + 31) / 32 * 32);
            // (round 6: TILED keeps 16-byte accesses at element alignment -- (999,1001) Float64 5.7 us there against 6.45 here, Float32 4.6
            // against 6.1 -- but at 64 MiB this form is ahead again: (2049,2051) Float64 14.6 against 18.4 us)
            if (e % vlen && (!o.tiled_uavec || c.total * es >= ((i64)32 << 20))) novec = true;  // TILED then moves 4- / 8-byte elements one by one as well, in padded tiles: (999,1001) 7.2 -> 6.4 us
            continue;
        }
        if ((e & (e - 1)) != 0) awkward = odd_short = true;
        else if (e * es < 32) awkward = true;
    }
    if (cutlead[0] || cutlead[1]) {
        // next to a cut lead only a short lead that is NOT a power of two counts ((2,2,256,2,2,256) and (64,2,64,2,16) permutations with a
        // 16-byte lead beside a 2-KiB one: TILED 3.2-3.9 us, this form 6.9-7.1)
        const bool anysize = o.flat2_long >= 100;  // tests / experiments: wherever the form applies
        // (round 5: ... and under 1 GiB: at HBM size the padded 32 x 32 tiles win again -- permutedims!(4,3,2,1) of 136^4 Float64, 72 %
        // fill: 1562 us in this form, 1373 us in TILED, profiles/r05_pow2.txt)
        awkward = odd_short || (fill * 100 < (long double)o.flat2_long && (anysize || (c.total * es >= ((i64)8 << 20) && c.total * es < ((i64)1 << 30)))) ||
                  (novec && (anysize || c.total * es >= ((i64)3 << 20)));
    }
    if (!awkward) return false;
    bool used[MAXN];
    for (int d = 0; d < MAXN; ++d) used[d] = f.ingroup[0][d] = f.ingroup[1][d] = false;
    used[lead[0]] = used[lead[1]] = true;
    const i64 target = std::max<i64>(64, o.flat2_bytes);
    for (int s = 0; s < 2; ++s) {
        if (cutlead[s]) {
            const i64 e = c.dims[lead[s]];
            i64 tp = std::max<i64>(8, std::min<i64>(target / es, 128));
            const i64 nt = (e + tp - 1) / tp;
            f.R[s] = 1;
            f.p[s] = lead[s];
            f.TP[s] = (int)((e + nt - 1) / nt);
            continue;
        }
        f.ingroup[s][lead[s]] = true;
        i64 R = c.dims[lead[s]];
        f.p[s] = -1;
        f.TP[s] = 1;
        for (;;) {
            int nxt = -1;
            for (int d = 0; d < c.N; ++d)
                if (!used[d] && st[s][d] == R && c.dims[d] > 1) nxt = d;
            if (nxt < 0) break;
            used[nxt] = true;
            if (R * c.dims[nxt] <= 64 && R * c.dims[nxt] * es <= target) {
                f.ingroup[s][nxt] = true;
                R *= c.dims[nxt];
                continue;
            }
            i64 tp = std::max<i64>(1, std::min<i64>(std::min<i64>(target / (R * es), 128 / R), c.dims[nxt]));
            const i64 nt = (c.dims[nxt] + tp - 1) / tp;
            tp = (c.dims[nxt] + nt - 1) / nt;  // evened out over the extent
            f.p[s] = nxt;
            f.TP[s] = (int)tp;
            break;
        }
        if (R > 64) return false;
        f.R[s] = (int)R;
    }
    // both sides continue along the same dim behind their leads ((5,N,7) -> (7,N,5)): the destination got it; the input's run is
    // its lead alone, but input memory is still walked contiguously (Flat2Plan::shared)
    f.shared = f.p[1] < 0 && f.p[0] >= 0 && st[1][f.p[0]] == f.R[1] && f.TP[0] > 1;
    if (f.shared) {  // the tile is R[0] x TP[0] x R[1]: a longer tile along the shared dim (16 KiB, at most 512 positions of the destination run)
        const i64 dimp = c.dims[f.p[0]];
        i64 tp = std::max<i64>(1, std::min<i64>(std::min<i64>(512 / f.R[0], 16384 / ((i64)f.R[0] * f.R[1] * es)), dimp));
        const i64 nt = (dimp + tp - 1) / tp;
        f.TP[0] = (int)((dimp + nt - 1) / nt);
    }
    // enough tiles for the device: shorten the longer run while it stays above 128 bytes
    auto ntiles = [&]() {
        i64 n = 1;
        for (int d = 0; d < c.N; ++d) {
            if (f.ingroup[0][d] || f.ingroup[1][d]) continue;
            if (d == f.p[0]) n *= (c.dims[d] + f.TP[0] - 1) / f.TP[0];
            else if (d == f.p[1]) n *= (c.dims[d] + f.TP[1] - 1) / f.TP[1];
            else n *= c.dims[d];
        }
        return n;
    };
    for (int guard = 0; guard < 16 && ntiles() < 1024; ++guard) {
        const int s = ((i64)f.R[0] * f.TP[0] >= (i64)f.R[1] * f.TP[1]) ? 0 : 1;
        int pick = -1;
        for (int t : {s, 1 - s})
            if (pick < 0 && f.TP[t] > 1 && (i64)f.R[t] * ((f.TP[t] + 1) / 2) * es >= 128) pick = t;
        if (pick < 0) break;
        f.TP[pick] = (f.TP[pick] + 1) / 2;
    }
    if (f.TP[0] <= 1) f.shared = false;  // one position along the shared dim: the two orders coincide
    if ((i64)f.R[0] * f.TP[0] > (f.shared ? 512 : 128) || (i64)f.R[1] * f.TP[1] > 128) return false;
    if ((i64)f.R[0] * f.TP[0] * es < 64 && (i64)f.R[1] * f.TP[1] * es < 64) return false;  // nothing gained over the generic kernel
    if (ntiles() > 0x7fffffffLL) return false;
    // offsets on the other side of the leading index r of each run (mixed radix over the group dims in this side's stride order)
    for (int s = 0; s < 2; ++s) {
        std::vector<int> order;
        for (int d = 0; d < c.N; ++d)
            if (f.ingroup[s][d]) order.push_back(d);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return st[s][x] < st[s][y]; });
        for (i64 r = 0; r < f.R[s]; ++r) {
            i64 rem = r, off = 0;
            for (int d : order) {
                off += (rem % c.dims[d]) * st[1 - s][d];
                rem /= c.dims[d];
            }
            if (off > 2147483647LL) return false;
            f.roff[s][r] = (int32_t)off;
        }
    }
    f.on = true;
    return true;
}

// ---- family selection -------------------------------------------------------------------------------
int make_plan(const smr_problem* p, Plan& plan) {
    int rc = canonicalise(p, plan.c);
    if (rc) return rc;
    const Canon& c = plan.c;
    const Options& o = options();
    int fam = FAM_GENERIC;
    plan.flat2.on = false;
    plan.flatb.on = false;
    // FLAT: the one-sided form first -- unless its line side is under 8 elements long, where the two-sided form's bigger tiles win
    // (ComplexF64 (6,64,64,64,5) reversed 94 -> 40 us, (4,300,300,3) 16.5 -> 7.6; with 24-element lines the one-sided form's
    // vector accesses are 1.5x ahead; profiles/r03_flat2_ab.txt).  flat2 = 2: the two-sided form wherever it applies (experiments)
    // force_family = FAM_FLAT (tests, experiments) tries the FLAT planners like the default does and leaves the rest of the decision alone
    bool flat_ok = c.redop == SMR_RED_NONE && (o.force_family == 0 || o.force_family == FAM_FLAT);
    // plan_orbit builds the orbit list: run it once per plan, whoever asks first
    int orbit_state = -1;
    auto orbit_ok = [&]() {
        if (orbit_state < 0) orbit_state = plan_orbit(c, plan.orbit) ? 1 : 0;
        return orbit_state == 1;
    };
    // several inputs that are permuted views of ONE buffer: the orbit kernel reads the buffer once, the n-ary FLAT forms would read it per view
    if (flat_ok && c.M > 2 && o.force_family == 0 && orbit_ok()) flat_ok = false;
    if (flat_ok && plan_flatb(c, plan.flatb)) {
        plan.family = FAM_FLAT;
        describe(plan);
        return SMR_OK;
    }
    const bool one = flat_ok && plan_flat(c, plan.flat);
    const bool two_first = flat_ok && (o.flat2 >= 2 || !one || (!plan.flat.fuse && !plan.flat.lshare && c.dims[plan.flat.q] < 8));
    if (two_first && plan_flat2(c, plan.flat2)) {
        fam = FAM_FLAT;
    } else if (one) {
        fam = FAM_FLAT;
    } else if (c.redop == SMR_RED_NONE) {
        bool stream = true;
        for (int k = 0; k < c.M; ++k) {
            i64 s = c.strides[k][0];
            if (!(s == 1 || (k > 0 && s == 0))) stream = false;
        }
        if (stream && c.N >= 2 && o.force_family != FAM_STREAM) {
            // short rows pack (256 >> txlog) entries of dim 1 into a workgroup; when dim 1 is short too (a permutation
            // of a 4x4x4x... tensor) most lanes idle -- 8 of 256 for rows of 4 Float64 x 4 -- and the per-element
            // decode of the GENERIC family wins (measured 4^8 Float64: 9.9 us -> tools/perf_sanity.py)
            const int es0 = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
            const i64 vmax = std::max<i64>(1, 16 / es0);
            const i64 n0v = (c.dims[0] % vmax == 0) ? c.dims[0] / vmax : c.dims[0];
            if (n0v <= 128) {
                int txlog = 0;
                while (((i64)1 << txlog) < n0v) ++txlog;
                const i64 used = n0v * std::min<i64>(c.dims[1], (i64)256 >> txlog);
                if (used * 4 < 256) stream = false;
            }
        }
        if (stream && o.force_family != FAM_STREAM) {
            const int es0 = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
            bool transposed = false;
            for (int k = 1; k < c.M; ++k) transposed = transposed || short_dim0_transposition(c, k, es0);
            if (transposed && plan_tiles(c, plan.tile)) {
                stream = false;
                fam = FAM_TILED;
            }
        }
        if (fam == FAM_TILED) {
        } else if (stream) {
            fam = FAM_STREAM;
        } else if (o.force_family != FAM_TILED && o.force_family != FAM_GENERIC && orbit_ok()) {
            fam = FAM_ORBIT;
        } else if (plan_tiles(c, plan.tile)) {
            fam = FAM_TILED;
        } else if (c.dims[0] >= 64) {
            // strided along dim 0 (A[1:2:end, :], a destination that is itself a strided view):
            // still one row segment per workgroup with the outer dims decoded once per
            // workgroup; element-wise (non-vector) accesses with the operand's own stride
            fam = FAM_STREAM;
        }
    } else {
        fam = (c.NK == 0) ? FAM_REDUCE_ALL : FAM_REDUCE_PART;
        if (fam == FAM_REDUCE_ALL && c.N > 1) {
            // a complete reduction whose dims do not fuse into one run (a sub-box, a permuted sub-box):
            // the ROW form with a single destination element walks it without a per-element index
            // decomposition (measured on a 1000x1000x256 box of a 1024x1024x256 array: 0.59 -> 5 TB/s)
            bool row = true, any = false;
            for (int k = 1; k < c.M; ++k) {
                if (c.strides[k][0] == 1) any = true;
                else if (c.strides[k][0] != 0) row = false;
            }
            if (row && any) fam = FAM_REDUCE_PART;
        }
    }
    if (o.force_family == FAM_GENERIC && c.redop == SMR_RED_NONE) fam = FAM_GENERIC;
    if (o.force_family == FAM_TILED && c.redop == SMR_RED_NONE && fam != FAM_TILED && plan_tiles(c, plan.tile)) fam = FAM_TILED;
    plan.family = fam;

    if (fam == FAM_STREAM) {
        // vector width: 16 B per lane when every unit-stride operand stays 16-B aligned
        const int es = c.bitcopy ? c.esize[0] : dtype_size(c.ct);
        int v = (c.mixed || es >= 16) ? 1 : 16 / es;
        for (int k = 0; k < c.M; ++k)
            if (c.strides[k][0] != 1 && c.strides[k][0] != 0) v = 1;  // strided form
        while (v > 1) {
            bool ok = c.dims[0] % v == 0;
            for (int k = 0; k < c.M && ok; ++k) {
                if (c.strides[k][0] == 0) continue;
                if (((uintptr_t)c.base[k] + (uintptr_t)(c.offsets[k] * c.esize[k])) % (size_t)(v * es)) ok = false;
                for (int i = 1; i < c.N && ok; ++i)
                    if (c.strides[k][i] % v) ok = false;
            }
            if (ok) break;
            v >>= 1;
        }
        plan.vec = v;
        plan.vec_ua = false;
        // Round 5: rows that are not whole aligned vectors (odd lengths, odd row strides, views starting inside a vector) still move
        // as 16-byte vectors -- at element alignment, plus one partial vector per row (smr_k_stream.hip: UVec) -- when the elements
        // are 4 or 8 bytes and a row holds at least four vectors.
        {
            const int vmax = (c.mixed || es >= 16) ? 1 : 16 / es;
            bool unit = true;
            for (int k = 0; k < c.M; ++k)
                if (c.strides[k][0] != 1 && c.strides[k][0] != 0) unit = false;
            // (short rows with a long tail lose: rows of 63 Float32 = 15 vectors + 3 single elements 24.9 -> 27.1 us; rows of 17
            // Float64 = 8 vectors + 1 element gain 9 %: a tail of one element, or at least 32 vectors per row.  tools/stream_ua_ab.py,
            // profiles/r05_stream_ua_ab.txt: (257,129,65) (0,2,1) 12.3 -> 9.5 us, (999,1001) axpy 4.7 -> 3.1 us, (1001,999,5) f32 10.9 -> 8.0)
            if (o.stream_ua && v < vmax && unit && es >= 4 && c.dims[0] >= 4 * vmax && (c.dims[0] % vmax <= 1 || c.dims[0] >= 32 * vmax)) {
                plan.vec = vmax;
                plan.vec_ua = true;
            }
        }
    }
    if (fam == FAM_REDUCE_ALL || fam == FAM_REDUCE_PART) {
        const int es = dtype_size(c.ct);
        if (fam == FAM_REDUCE_ALL) {
            i64 per_block = 256 * 16;
            i64 nb = (c.total + per_block - 1) / per_block;
            nb = std::max<i64>(1, std::min<i64>(nb, std::max<i64>(1, o.reduce_blocks)));
            plan.red_blocks = (int)nb;
            plan.scratch_bytes = (size_t)nb * es;
        } else {
            const i64 red = c.total / std::max<i64>(1, c.nout);
            // no reduced dim at all (an accumulating map, dest[I] = op(dest[I], f(...)), over up to MAXN kept
            // dims): nothing to index at position NK -- the general form handles it
            const bool nored = c.NK >= c.N;
            const i64 L0 = nored ? 1 : c.dims[c.NK];  // inner reduced dim
            const i64 Q = red / L0;                    // outer reduced index (dims NK+1..)
            // which vectorisable form applies?
            bool row = !nored, col = !nored && c.strides[0][0] == 1, any_row = false, any_col = false;
            for (int k = 1; k < c.M && !nored; ++k) {
                const i64 sr = c.strides[k][c.NK], sc = c.strides[k][0];
                if (sr == 1) any_row = true;
                else if (sr != 0) row = false;
                if (sc == 1) any_col = true;
                else if (sc != 0) col = false;
            }
            row = row && any_row;
            col = col && any_col;
            const int vmax = std::max(1, 16 / es);
            auto p2ceil = [](i64 v) {
                int l = 0;
                while (((i64)1 << l) < v) ++l;
                return l;
            };
            // number of chunks an index range of `n` is cut into when about `want` are asked for: all of n when that is
            // within 2x (a trailing dim of 7 cut 6 ways would leave chunks of 2,2,2,1 and two idle workgroups), else a
            // count that leaves no chunk empty
            auto even_cut = [](i64 want, i64 n) -> i64 {
                if (n <= 1 || want <= 1) return 1;
                if (n <= 2 * want) return n;
                const i64 per = (n + want - 1) / want;
                return (n + per - 1) / per;
            };
            if (o.reduce_part_kind >= 0) {  // tuning / testing override
                row = row && o.reduce_part_kind == 1;
                col = col && o.reduce_part_kind == 2;
            }
            if (col) {
                // COL: a workgroup = TX lanes along kept dim 0 (vmax elements each) x TY rows of the
                // reduced space; LDS tree over the rows
                plan.part_kind = 2;
                const i64 K0 = c.dims[0];
                int txlog = std::min(8, p2ceil((K0 + vmax - 1) / vmax));
                // rows of the reduced space per workgroup: fewer when the reduction is short (sum over a trailing dim of 7:
                // every lane then walks its 7 rows itself instead of 8 lanes sharing them through LDS)
                txlog = std::min<int>(txlog, std::max<int>((int)o.reduce_col_txlog, 8 - p2ceil(std::max<i64>(1, red / 8))));
                // too few workgroups along the kept dims: narrower row segments (down to 128 bytes) give 2-4x as many and more
                // rows to each -- when that makes the split unnecessary it saves the partials and the second launch
                // (sum(A; dims=2) of 512x384x64 f32: 16.0 -> 12.4 us, 256^3: 18.4 -> 14.7 us, tools/reduce_sweep.py)
                auto kb_at = [&](int t) { return ((K0 + (((i64)vmax) << t) - 1) / (((i64)vmax) << t)) * (c.nout / K0); };
                if (o.reduce_col_narrow && kb_at(txlog) < o.reduce_part_wgs) {
                    int fill = -1;  // the widest segment that still puts a workgroup on every CU (round 6)
                    const int t0 = txlog;
                    for (int t = txlog - 1; t >= 3 && (((i64)vmax << t) * es >= 128); --t) {
                        if ((red >> (8 - t)) < 8) break;  // fewer than 8 rows of the reduced space per lane row
                        if (fill < 0 && kb_at(t) >= 256) fill = t;
                        if (kb_at(t) >= o.reduce_part_wgs) {
                            txlog = t;
                            break;
                        }
                    }
                    // no width reaches the target, but one fills the device without a split: the second launch of a split costs more
                    // than a thinner first one (sum(A; dims=(3,4)) of (100,90,80,7) Float32: 9000 outputs, 560 rows each: 71 workgroups cut
                    // 7 ways + a second pass 8.9 us, 282 workgroups of 8 lanes x 32 rows in one launch: see profiles/r06_sum_cases.txt)
                    if (txlog == t0 && fill >= 0 && o.reduce_col_narrow >= 1 && kb_at(t0) < 256) txlog = fill;
                }
                plan.part_txlog = txlog;
                const int tylog = 8 - txlog;
                plan.part_g0log = std::min(tylog, p2ceil(L0));
                plan.part_g1log = tylog - plan.part_g0log;
                i64 kb = ((K0 + (((i64)vmax) << txlog) - 1) / (((i64)vmax) << txlog)) * (c.nout / K0);
                i64 rows_per_wg = (i64)1 << tylog, y0 = (i64)1 << plan.part_g0log, y1 = (i64)1 << plan.part_g1log;
                // exact lane map (round 6): a row of 100 Float32 is 25 vectors -- 32 lanes leave 22 % of the workgroup idle, 25 lanes x 10
                // rows leave 2 % (and 10 rows = 4000 contiguous bytes per step).  Taken when it fills at least 3 % more lanes; the
                // row may be cut into up to four even segments.  Valid for the vector width assumed here (the launch checks alignment).
                plan.part_col_tx = 0;
                if (o.reduce_col_exact) {
                    bool vdiv = vmax > 1 && K0 % vmax == 0;
                    for (int k = 1; k < c.M && vdiv; ++k)
                        if (c.strides[k][0] == 1)
                            for (int d = 1; d < c.N; ++d)
                                if (c.strides[k][d] % vmax) vdiv = false;
                    const i64 vv = vdiv ? vmax : 1;
                    const i64 n0v = (K0 + vv - 1) / vv, per2 = vv << txlog, nk2 = (K0 + per2 - 1) / per2;
                    double best = (double)K0 / (double)(nk2 * per2);
                    i64 btx = 0;
                    // (cutting the row into more segments just to put a workgroup on every CU and avoid the split loses: rows of 100 as
                    // 4 x 7 lanes x 36 rows, sum over dims (2,4) of (100,90,80,7) Float32 11.4 us against 9.2 with the split)
                    for (i64 sg = 1; sg <= 4; ++sg) {
                        const i64 tx = (n0v + sg - 1) / sg;
                        if (tx > 256 || tx * vv * es < 64 || (tx & (tx - 1)) == 0) continue;
                        const i64 ty = 256 / tx;
                        if (red < 8 * ty) continue;  // short reductions keep the few-rows rule above
                        const double util = (double)n0v / (double)(sg * tx) * (double)(tx * ty) / 256.0;
                        if (util > best + 0.03) {
                            best = util;
                            btx = tx;
                            if (util >= 0.9) break;  // the fewest segments that fill the workgroup: longer contiguous pieces per row
                        }
                    }
                    if (btx) {
                        plan.part_col_tx = (int)btx;
                        plan.part_col_v = (int)vv;
                        rows_per_wg = 256 / btx;
                        // rows along the inner reduced dim: the largest divisor of the row count that it can fill
                        y0 = 1;
                        for (i64 dv = 1; dv <= rows_per_wg; ++dv)
                            if (rows_per_wg % dv == 0 && dv <= std::max<i64>(1, L0)) y0 = dv;
                        y1 = rows_per_wg / y0;
                        plan.part_col_y0 = (int)y0;
                        plan.part_col_y1 = (int)y1;
                        kb = ((n0v + btx - 1) / btx) * (c.nout / K0);
                    }
                }
                i64 split = 1;
                // half the ROW form's target: every workgroup leaves TX*V partials per chunk, and the sweep has 512 ahead of 1024-4096
                const i64 target = std::max<i64>(1, o.reduce_part_wgs / 2);
                if (kb < target && red >= rows_per_wg * 16) split = std::max<i64>(1, std::min<i64>(target / kb, red / (rows_per_wg * 8)));
                split = std::min<i64>(split, 4096);
                // cut the outer reduced index first, the inner reduced dim with what is left (a short Q -- a trailing
                // dim of 7 -- used to forbid any cut but 2 along L0: 160 workgroups for 19 MiB)
                plan.part_qsplit = (int)even_cut(split, Q / y1);
                plan.part_xsplit = (int)std::max<i64>(1, std::min<i64>(split / plan.part_qsplit, L0 / (4 * y0)));
                plan.part_split = plan.part_xsplit * plan.part_qsplit;
                plan.part_tr = (int)rows_per_wg;
            } else if (row) {
                // ROW: G consecutive lanes per output, vector loads along the inner reduced dim
                plan.part_kind = 1;
                int glog = 8;
                // (no floor on the lanes per output -- rounds 1-3 had 16: sum(A; dims=1) of a 3 x N array then ran on 1 lane in 16
                // and took 200 us for 33 MB (now 15.4), of 100 x 50400 f32 7.6 us (now 4.75 with 4 lanes of 6 vectors each), of
                // 32 x 200000 21.5 (now 5.6); tools/reduce_sweep.py, profiles/r03_reduce_sweep.txt)
                const int gfloor = o.reduce_row_floor >= 0 ? (int)o.reduce_row_floor : 0;
                while (glog > gfloor && red < ((i64)vmax << glog) * 4) --glog;
                // a very short inner dim with kept dim 0 right behind it in memory (sum over the channels AND the rows of a
                // 3 x W x H image): lanes along the outputs read neighbouring segments; lanes along the outer reduced index -- what
                // the rule above picks when the whole reduction is long -- would each fetch 12 bytes from a line of their own
                bool dense0 = o.reduce_row_dense && c.NK >= 1 && Q > 1 && c.dims[0] >= 256 && L0 * es <= 64;
                for (int k = 1; k < c.M && dense0; ++k)
                    if (c.strides[k][c.NK] == 1 && c.strides[k][0] != L0) dense0 = false;
                if (dense0) {
                    glog = 8;
                    while (glog > 0 && L0 < ((i64)vmax << glog) * 4) --glog;
                }
                plan.part_g0log = std::min(glog, p2ceil((L0 + vmax - 1) / vmax));
                plan.part_g1log = glog - plan.part_g0log;
                const i64 groups = (c.nout + (256 >> glog) - 1) / (256 >> glog);
                i64 split = 1;
                if (groups < o.reduce_part_wgs && red >= ((i64)vmax << glog) * 16) split = std::max<i64>(1, std::min<i64>(o.reduce_part_wgs / groups, red / (((i64)vmax << glog) * 8)));
                split = std::min<i64>(split, 4096);
                plan.part_qsplit = (int)even_cut(split, Q >> plan.part_g1log);
                plan.part_xsplit = (int)std::max<i64>(1, std::min<i64>(split / plan.part_qsplit, L0 / (((i64)vmax * 4) << plan.part_g0log)));
                plan.part_split = plan.part_xsplit * plan.part_qsplit;
                plan.part_tr = 1 << glog;
            } else {
                // general form: lanes cooperating per output: more when few outputs / long reductions
                int tr = 1;
                // the inputs' fastest-varying dim is a reduced one -> lanes along it coalesce
                bool red_fast = false;
                for (int k = 1; k < c.M; ++k)
                    for (int i = c.NK; i < c.N; ++i)
                        if (std::llabs(c.strides[k][i]) == 1) red_fast = true;
                if (red_fast || c.nout < 256 * 256) {
                    while (tr < 256 && tr * 2 <= red && (tr < 64 || c.nout * tr < 256 * 1024)) tr <<= 1;
                }
                if (red_fast && tr < 16 && red >= 16) tr = 16;
                plan.part_tr = tr;
                // few destination elements, long reductions: cut the reduced range so that ~2048
                // workgroups are in flight, partials folded by a second launch (also keeps the
                // serial per-lane accumulation short, which is what bounds the rounding error)
                const i64 groups = (c.nout + (256 / tr) - 1) / (256 / tr);
                i64 split = 1;
                if (groups < 1024 && red >= (i64)tr * 256) {
                    split = std::min<i64>(2048 / std::max<i64>(1, groups), red / ((i64)tr * 64));
                    split = std::max<i64>(1, std::min<i64>(split, 4096));
                }
                plan.part_split = (int)split;
            }
            if (plan.part_split > 1) {
                // chunk partials, then RED_SHARDS shard partials per output (two-level in-launch fold, smr_k_reduce.hip)
                plan.scratch_bytes = (size_t)c.nout * ((size_t)plan.part_split + RED_SHARDS) * es;
                plan.red_blocks = plan.part_split;  // > 1: the API allocates the partials buffer
            }
        }
    }
    describe(plan);
    return SMR_OK;
}

void describe(Plan& plan) {
    const Canon& c = plan.c;
    static const char* fam[] = {"auto", "generic", "stream", "tiled", "reduce_all", "reduce_part", "orbit", "flat"};
    static const char* fk[] = {"prog", "ident", "add2", "add3", "add4", "scale", "sym", "axpy", "axpby", "abs2", "mul2", "expr5"};
    static const char* ct[] = {"f32", "f64", "c32", "c64"};
    char buf[1024];
    int n = std::snprintf(buf, sizeof buf, "family=%s ct=%s%s f=%s N=%d M=%d dims=", fam[plan.family], c.ct == SMR_I64 ? "i64" : ct[c.ct & 3],
                          c.bitcopy ? "(bitcopy)" : (c.mixed ? "(mixed)" : ""), fk[c.fkind], c.N, c.M);
    for (int i = 0; i < c.N; ++i) n += std::snprintf(buf + n, sizeof buf - n, "%s%lld", i ? "x" : "", (long long)c.dims[i]);
    if (plan.family == FAM_TILED) {
        n += std::snprintf(buf + n, sizeof buf - n, " tile=");
        for (int j = 0; j < plan.tile.nt; ++j)
            n += std::snprintf(buf + n, sizeof buf - n, "%sd%d:%d", j ? "," : "", plan.tile.tdim[j], 1 << plan.tile.tlog[j]);
        n += std::snprintf(buf + n, sizeof buf - n, " staged=%d lds=%zu grid=%lld threads=%d", plan.tile.nstaged,
                           plan.tile.lds_bytes, (long long)plan.tile.grid, plan.tile.threads);
        if (!plan.tile.ord.empty()) n += std::snprintf(buf + n, sizeof buf - n, " order=orbits:%d", plan.tile.ord_groups);
    } else if (plan.family == FAM_ORBIT) {
        const OrbitPlan& ob = plan.orbit;
        n += std::snprintf(buf + n, sizeof buf - n, " tile=");
        bool first = true;
        for (int d = 0; d < c.N; ++d)
            if (ob.lg[d] > 0) {
                n += std::snprintf(buf + n, sizeof buf - n, "%sd%d:%d", first ? "" : ",", d, 1 << ob.lg[d]);
                first = false;
            }
        n += std::snprintf(buf + n, sizeof buf - n, " group=%d orbits=%d lds=%zu grid=%zu", ob.ng, ob.norbits, ob.lds_bytes, ob.wmap.size());
        if (ob.pair_ok) n += std::snprintf(buf + n, sizeof buf - n, " pair_grid=%zu", ob.pmap.size() / 2);
    } else if (plan.family == FAM_FLAT && plan.flatb.on) {
        n += std::snprintf(buf + n, sizeof buf - n, " batched block=%d(d0..d%d) blocks_per_wg=%d", plan.flatb.P, plan.flatb.g - 1, plan.flatb.K);
    } else if (plan.family == FAM_FLAT && plan.flat2.on) {
        const Flat2Plan& f2 = plan.flat2;
        n += std::snprintf(buf + n, sizeof buf - n, " two-sided dest_run=%dx%d(d%d) input_run=%dx%d(d%d)%s", f2.R[0], f2.TP[0], f2.p[0], f2.R[1], f2.TP[1], f2.p[1],
                           f2.shared ? " shared-dim" : "");
    } else if (plan.family == FAM_FLAT) {
        const FlatPlan& fp = plan.flat;
        n += std::snprintf(buf + n, sizeof buf - n, " flat_side=%s run=%dx%d(d%d)%s line=d%d:%d", fp.dir == 0 ? "dest" : "input", fp.R, 1 << fp.tplog, fp.p,
                           fp.fuse ? "+line" : (fp.lshare ? " shared-lead" : ""), fp.q, 1 << fp.tqlog);
    } else if (plan.family == FAM_STREAM) {
        n += std::snprintf(buf + n, sizeof buf - n, " vec=%d%s", plan.vec, plan.vec_ua ? "(element-aligned+tail)" : "");
    } else if (plan.family == FAM_REDUCE_ALL) {
        n += std::snprintf(buf + n, sizeof buf - n, " blocks=%d", plan.red_blocks);
    } else if (plan.family == FAM_REDUCE_PART) {
        static const char* kinds[] = {"general", "row", "col"};
        n += std::snprintf(buf + n, sizeof buf - n, " nout=%lld form=%s lanes_per_out=%d split=%d", (long long)c.nout, kinds[plan.part_kind],
                           plan.part_tr, plan.part_split);
        if (plan.part_kind == 2 && plan.part_col_tx) n += std::snprintf(buf + n, sizeof buf - n, " lanes=%dx%d", plan.part_col_tx, 256 / plan.part_col_tx);
    }
    if (c.int_wraps) n += std::snprintf(buf + n, sizeof buf - n, " int_wraps=%d", c.int_wraps);
    std::snprintf(buf + n, sizeof buf - n, " algbytes=%lld", (long long)c.algbytes);
    plan.desc = buf;
}

}  // namespace smr
