// strided_oracle.cpp -- CPU restatement of Strided.jl's map/reduce engine.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the "port" CPU baseline.
// Nothing under strided.jl_amd/ (the product) may include, link or call it; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle.so.
//
// PARITY PINNING: the reference is pure Julia and there is no Julia runtime in the build
// container or on the GPU box, and the reference's test-suite stores no
// golden vectors (every assertion is differential against Base Julia at run time).  This
// oracle is therefore "parity unpinned" against reference *executions*; it is pinned
// instead (tests/test_oracle_*.py) against
//   (i)  NumPy restatements of the very Base-Julia relations the reference tests assert
//        (test/othertests.jl:1-128), on committed fixtures (tests/golden/), and
//   (ii) the planner known-answers hand-traced from the reference code.
//
// Every function cites the reference lines it follows (paths relative to /root/reference).
// ONE deliberate deviation from the letter of the reference: src/mapreduce.jl:409 (`init_i &= stride_i_1 > 0`) is read as `!= 0`
// -- with a reversed (negative-stride) destination the literal line leaves block-size-dependent elements uninitialised; see
// Kernel::blockloop, oracle_set_literal_409 and tests/test_oracle_numpy.py::test_initop_with_a_reversed_destination.
// Indices here are 0-based; the reference is 1-based.

#include "../include/strided_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

typedef int64_t i64;
constexpr int MAXN = SMR_MAXN;
constexpr int MAXM = SMR_MAXM;

thread_local std::string g_err;
std::atomic<long> g_guard_hits{0};  // how often _computeblocks' termination guard fired (tests/golden: the case the reference is suspected to spin on)
std::atomic<bool> g_literal_409{false};  // see Kernel::blockloop (process-wide: the oracle's worker threads read it; atomic, so toggling it is no data race)
int fail(int code, const std::string& m) {
    g_err = m;
    return code;
}

int dtype_size(int dt) {
    switch (dt) {
        case SMR_F32: return 4;
        case SMR_F64: return 8;
        case SMR_C32: return 8;
        case SMR_C64: return 16;
        case SMR_I8: case SMR_U8: case SMR_BOOL: return 1;
        case SMR_I16: case SMR_U16: return 2;
        case SMR_I32: case SMR_U32: return 4;
        case SMR_I64: case SMR_U64: return 8;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// complex arithmetic with Julia's plain formulas (Base complex.jl: * is the 4-multiply form)
// ---------------------------------------------------------------------------------------------
template <class R>
struct cx {
    R re, im;
};
template <class T> struct traits;
template <> struct traits<float> { typedef float real; static constexpr bool is_cx = false; };
template <> struct traits<double> { typedef double real; static constexpr bool is_cx = false; };
template <> struct traits<cx<float>> { typedef float real; static constexpr bool is_cx = true; };
template <> struct traits<cx<double>> { typedef double real; static constexpr bool is_cx = true; };
// the integer class: Julia's Int64 arithmetic (wrapping two's complement; built with -fwrapv).  Selected when every
// operand has an integer eltype and f stays inside the integers -- src/mapreduce.jl:55-72 takes typeof(op(...)) as the
// accumulator type, and `+`/`*` of Int64 wrap in Julia.
typedef long long ix64;
template <> struct traits<ix64> { typedef ix64 real; static constexpr bool is_cx = false; };
// double -> real type of a class (saturating for the integer class: the +-Inf seeds of min / max become typemax / typemin)
template <class R> inline R rcast(double d) { return R(d); }
template <> inline ix64 rcast<ix64>(double d) {
    return d >= 9223372036854775807.0 ? 9223372036854775807LL : (d <= -9223372036854775808.0 ? (-9223372036854775807LL - 1) : (ix64)d);
}

template <class R> inline R re_of(R x) { return x; }
template <class R> inline R re_of(cx<R> x) { return x.re; }
template <class R> inline R im_of(R) { return R(0); }
template <class R> inline R im_of(cx<R> x) { return x.im; }
template <class T> inline T make(typename traits<T>::real re, typename traits<T>::real im);
template <> inline float make<float>(float re, float) { return re; }
template <> inline double make<double>(double re, double) { return re; }
template <> inline cx<float> make<cx<float>>(float re, float im) { return {re, im}; }
template <> inline cx<double> make<cx<double>>(double re, double im) { return {re, im}; }
template <> inline ix64 make<ix64>(ix64 re, ix64) { return re; }

template <class R> inline cx<R> operator+(cx<R> a, cx<R> b) { return {a.re + b.re, a.im + b.im}; }
template <class R> inline cx<R> operator-(cx<R> a, cx<R> b) { return {a.re - b.re, a.im - b.im}; }
template <class R> inline cx<R> operator*(cx<R> a, cx<R> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class R> inline cx<R> operator/(cx<R> a, cx<R> b) {
    std::complex<R> q = std::complex<R>(a.re, a.im) / std::complex<R>(b.re, b.im);
    return {q.real(), q.imag()};
}
template <class R> inline cx<R> operator-(cx<R> a) { return {-a.re, -a.im}; }

template <class R> inline R conj_of(R x) { return x; }
template <class R> inline cx<R> conj_of(cx<R> x) { return {x.re, -x.im}; }

// ---------------------------------------------------------------------------------------------
// f-program evaluation (the CaptureArgs functor call, src/broadcast.jl:86-98: arguments are
// consumed depth-first, operators applied left to right exactly as written)
// ---------------------------------------------------------------------------------------------
struct Prog {
    int len = 0;
    uint8_t code[2 * SMR_MAXPROG];
    int nconst = 0;
    double consts[2 * SMR_MAXCONST];
    // integer class only (julia_int_types below): the width Julia's type of the value produced at pc has when it is narrower than
    // 64 bits (0: no wrap), and its signedness.  Julia types every operation by its operands -- Int32 * Int32 is an Int32 and wraps
    // at 32 bits, Int8(1) - Int8(2) compared as UInt8 is 255 -- and the oracle follows it operation by operation.
    uint8_t wrap_bits[SMR_MAXPROG] = {};
    bool wrap_signed[SMR_MAXPROG] = {};
};

template <class T> struct ops;  // scalar math per compute type

template <class R>
struct real_ops {
    typedef R T;
    static T un(int op, T a) {
        switch (op) {
            case SMR_OP_NEG: return -a;
            case SMR_OP_ABS: return std::fabs(a);
            case SMR_OP_ABS2: return a * a;
            case SMR_OP_CONJ: return a;
            case SMR_OP_REAL: return a;
            case SMR_OP_IMAG: return R(0);
            case SMR_OP_SQRT: return std::sqrt(a);
            case SMR_OP_EXP: return std::exp(a);
            case SMR_OP_LOG: return std::log(a);
            case SMR_OP_SIN: return std::sin(a);
            case SMR_OP_COS: return std::cos(a);
            case SMR_OP_TANH: return std::tanh(a);
            case SMR_OP_INV: return R(1) / a;
            case SMR_OP_ROUND32: return (R)(float)a;
        }
        return a;
    }
    static T bin(int op, T a, T b) {
        switch (op) {
            case SMR_OP_ADD: return a + b;
            case SMR_OP_SUB: return a - b;
            case SMR_OP_MUL: return a * b;
            case SMR_OP_DIV: return a / b;
            // Julia's min / max (Base/math.jl): NaN if either argument is NaN, min(-0.0, 0.0) = -0.0,
            // max(-0.0, 0.0) = 0.0
            case SMR_OP_MIN: return (a != a) ? a : ((b != b) ? b : ((b < a) ? b : ((a < b) ? a : (std::signbit(a) ? a : b))));
            case SMR_OP_MAX: return (a != a) ? a : ((b != b) ? b : ((a < b) ? b : ((b < a) ? a : (std::signbit(a) ? b : a))));
            case SMR_OP_LT: return a < b ? R(1) : R(0);
            case SMR_OP_LE: return a <= b ? R(1) : R(0);
            case SMR_OP_GT: return a > b ? R(1) : R(0);
            case SMR_OP_GE: return a >= b ? R(1) : R(0);
            case SMR_OP_EQ: return a == b ? R(1) : R(0);
            case SMR_OP_NE: return a != b ? R(1) : R(0);
        }
        return a;
    }
    static bool truthy(T a) { return a != R(0); }
};
template <> struct ops<ix64> {
    typedef unsigned long long U;
    static ix64 un(int op, ix64 a) {
        switch (op) {
            case SMR_OP_NEG: return (ix64)(U(0) - (U)a);
            case SMR_OP_ABS: return a < 0 ? (ix64)(U(0) - (U)a) : a;  // abs(typemin(Int64)) == typemin(Int64) in Julia
            case SMR_OP_ABS2: return (ix64)((U)a * (U)a);
            case SMR_OP_IMAG: return 0;
        }
        return a;
    }
    static ix64 bin(int op, ix64 a, ix64 b) {
        switch (op) {
            case SMR_OP_ADD: return (ix64)((U)a + (U)b);
            case SMR_OP_SUB: return (ix64)((U)a - (U)b);
            case SMR_OP_MUL: return (ix64)((U)a * (U)b);
            case SMR_OP_MIN: return b < a ? b : a;
            case SMR_OP_MAX: return a < b ? b : a;
            case SMR_OP_LT: return a < b ? 1 : 0;
            case SMR_OP_LE: return a <= b ? 1 : 0;
            case SMR_OP_GT: return a > b ? 1 : 0;
            case SMR_OP_GE: return a >= b ? 1 : 0;
            case SMR_OP_EQ: return a == b ? 1 : 0;
            case SMR_OP_NE: return a != b ? 1 : 0;
        }
        return a;
    }
    static bool truthy(ix64 a) { return a != 0; }
};
template <> struct ops<float> : real_ops<float> {};
template <> struct ops<double> : real_ops<double> {};

template <class R>
struct cx_ops {
    typedef cx<R> T;
    typedef std::complex<R> S;
    static T from(S s) { return {s.real(), s.imag()}; }
    static S to(T a) { return S(a.re, a.im); }
    static T un(int op, T a) {
        switch (op) {
            case SMR_OP_NEG: return -a;
            case SMR_OP_ABS: return {std::hypot(a.re, a.im), R(0)};
            case SMR_OP_ABS2: return {a.re * a.re + a.im * a.im, R(0)};
            case SMR_OP_CONJ: return {a.re, -a.im};
            case SMR_OP_REAL: return {a.re, R(0)};
            case SMR_OP_IMAG: return {a.im, R(0)};
            case SMR_OP_SQRT: return from(std::sqrt(to(a)));
            case SMR_OP_EXP: return from(std::exp(to(a)));
            case SMR_OP_LOG: return from(std::log(to(a)));
            case SMR_OP_SIN: return from(std::sin(to(a)));
            case SMR_OP_COS: return from(std::cos(to(a)));
            case SMR_OP_TANH: return from(std::tanh(to(a)));
            case SMR_OP_INV: return T{R(1), R(0)} / a;
            case SMR_OP_ROUND32: return T{(R)(float)a.re, (R)(float)a.im};
        }
        return a;
    }
    static T bin(int op, T a, T b) {
        switch (op) {
            case SMR_OP_ADD: return a + b;
            case SMR_OP_SUB: return a - b;
            case SMR_OP_MUL: return a * b;
            case SMR_OP_DIV: return a / b;
            case SMR_OP_MIN: return (b.re < a.re) ? b : a;
            case SMR_OP_MAX: return (a.re < b.re) ? b : a;
            case SMR_OP_LT: return {a.re < b.re ? R(1) : R(0), R(0)};
            case SMR_OP_LE: return {a.re <= b.re ? R(1) : R(0), R(0)};
            case SMR_OP_GT: return {a.re > b.re ? R(1) : R(0), R(0)};
            case SMR_OP_GE: return {a.re >= b.re ? R(1) : R(0), R(0)};
            case SMR_OP_EQ: return {(a.re == b.re && a.im == b.im) ? R(1) : R(0), R(0)};
            case SMR_OP_NE: return {(a.re != b.re || a.im != b.im) ? R(1) : R(0), R(0)};
        }
        return a;
    }
    static bool truthy(T a) { return a.re != R(0); }
};
template <> struct ops<cx<float>> : cx_ops<float> {};
template <> struct ops<cx<double>> : cx_ops<double> {};

template <class T>
inline T eval_prog(const Prog& p, const T* args /* args[k] = value of input k (1-based) */) {
    typedef typename traits<T>::real R;
    T st[SMR_MAXPROG];
    int sp = 0;
    for (int pc = 0; pc < p.len; ++pc) {
        int op = p.code[2 * pc], imm = p.code[2 * pc + 1];
        if (op == SMR_OP_ARG) {
            st[sp++] = args[imm];
        } else if (op == SMR_OP_CONST) {
            st[sp++] = make<T>(rcast<R>(p.consts[2 * imm]), rcast<R>(p.consts[2 * imm + 1]));
        } else if (op < 32) {
            st[sp - 1] = ops<T>::un(op, st[sp - 1]);
        } else if (op < 64) {
            T b = st[--sp];
            st[sp - 1] = ops<T>::bin(op, st[sp - 1], b);
        } else {  // SELECT
            T c = st[--sp];
            T b = st[--sp];
            st[sp - 1] = ops<T>::truthy(st[sp - 1]) ? b : c;
        }
        if constexpr (std::is_same<T, ix64>::value) {  // the value in Julia's (narrower) type of this operation
            const int w = p.wrap_bits[pc];
            if (w) {
                const unsigned long long m = (1ull << w) - 1, u = (unsigned long long)st[sp - 1] & m;
                st[sp - 1] = (p.wrap_signed[pc] && (u >> (w - 1))) ? (ix64)(u | ~m) : (ix64)u;
            }
        }
    }
    return st[0];
}

// Validates stack discipline; returns max input index used (or <0 on error).
int check_prog(const Prog& p, int M) {
    int sp = 0, maxarg = 0;
    for (int pc = 0; pc < p.len; ++pc) {
        int op = p.code[2 * pc], imm = p.code[2 * pc + 1];
        if (op == SMR_OP_ARG) {
            if (imm < 1 || imm >= M) return -1;
            maxarg = std::max(maxarg, imm);
            ++sp;
        } else if (op == SMR_OP_CONST) {
            if (imm >= p.nconst) return -1;
            ++sp;
        } else if (op >= 8 && op <= SMR_OP_WIDEN) {
            if (sp < 1) return -1;
        } else if (op >= 32 && op <= SMR_OP_NE) {
            if (sp < 2) return -1;
            --sp;
        } else if (op == SMR_OP_SELECT) {
            if (sp < 3) return -1;
            sp -= 2;
        } else {
            return -1;
        }
    }
    return sp == 1 ? maxarg : -1;
}

// ---------------------------------------------------------------------------------------------
// element access: ParentIndex get/set applies the view's op (identity / conj) on load and
// store (src/mapreduce.jl:276-278; StridedViews.jl semantics)
// ---------------------------------------------------------------------------------------------
template <class T>
inline T load_as(const void* base, i64 idx, int dt) {
    typedef typename traits<T>::real R;
    if constexpr (std::is_same<T, ix64>::value) {  // sign / zero extension
        switch (dt) {
            case SMR_I8: return ((const int8_t*)base)[idx];
            case SMR_U8: case SMR_BOOL: return ((const uint8_t*)base)[idx];
            case SMR_I16: return ((const int16_t*)base)[idx];
            case SMR_U16: return ((const uint16_t*)base)[idx];
            case SMR_I32: return ((const int32_t*)base)[idx];
            case SMR_U32: return ((const uint32_t*)base)[idx];
            case SMR_I64: return ((const long long*)base)[idx];
            case SMR_U64: return (ix64)((const unsigned long long*)base)[idx];
        }
        return 0;
    } else
    switch (dt) {
        case SMR_F32: return make<T>(R(((const float*)base)[idx]), R(0));
        case SMR_F64: return make<T>(R(((const double*)base)[idx]), R(0));
        case SMR_C32: return make<T>(R(((const float*)base)[2 * idx]), R(((const float*)base)[2 * idx + 1]));
        case SMR_C64: return make<T>(R(((const double*)base)[2 * idx]), R(((const double*)base)[2 * idx + 1]));
        case SMR_I8: return make<T>(R(((const int8_t*)base)[idx]), R(0));
        case SMR_U8: case SMR_BOOL: return make<T>(R(((const uint8_t*)base)[idx]), R(0));
        case SMR_I16: return make<T>(R(((const int16_t*)base)[idx]), R(0));
        case SMR_U16: return make<T>(R(((const uint16_t*)base)[idx]), R(0));
        case SMR_I32: return make<T>(R(((const int32_t*)base)[idx]), R(0));
        case SMR_U32: return make<T>(R(((const uint32_t*)base)[idx]), R(0));
        case SMR_I64: return make<T>(R(((const int64_t*)base)[idx]), R(0));
        case SMR_U64: return make<T>(R(((const uint64_t*)base)[idx]), R(0));
    }
    return make<T>(R(0), R(0));
}
template <class T>
inline void store_as(void* base, i64 idx, int dt, T v) {
    if constexpr (std::is_same<T, ix64>::value) {  // truncation = the wrapped value in the narrower type
        switch (dt) {
            case SMR_I8: case SMR_U8: case SMR_BOOL: ((uint8_t*)base)[idx] = (uint8_t)v; break;
            case SMR_I16: case SMR_U16: ((uint16_t*)base)[idx] = (uint16_t)v; break;
            case SMR_I32: case SMR_U32: ((uint32_t*)base)[idx] = (uint32_t)v; break;
            case SMR_I64: case SMR_U64: ((long long*)base)[idx] = v; break;
        }
        return;
    } else {
    auto re = re_of(v);
    auto im = im_of(v);
    switch (dt) {
        case SMR_F32: ((float*)base)[idx] = (float)re; break;
        case SMR_F64: ((double*)base)[idx] = (double)re; break;
        case SMR_C32: ((float*)base)[2 * idx] = (float)re; ((float*)base)[2 * idx + 1] = (float)im; break;
        case SMR_C64: ((double*)base)[2 * idx] = (double)re; ((double*)base)[2 * idx + 1] = (double)im; break;
        case SMR_I8: ((int8_t*)base)[idx] = (int8_t)std::llrint((double)re); break;
        case SMR_U8: case SMR_BOOL: ((uint8_t*)base)[idx] = (uint8_t)std::llrint((double)re); break;
        case SMR_I16: ((int16_t*)base)[idx] = (int16_t)std::llrint((double)re); break;
        case SMR_U16: ((uint16_t*)base)[idx] = (uint16_t)std::llrint((double)re); break;
        case SMR_I32: ((int32_t*)base)[idx] = (int32_t)std::llrint((double)re); break;
        case SMR_U32: ((uint32_t*)base)[idx] = (uint32_t)std::llrint((double)re); break;
        case SMR_I64: ((int64_t*)base)[idx] = (int64_t)std::llrint((double)re); break;
        case SMR_U64: ((uint64_t*)base)[idx] = (uint64_t)std::llrint((double)re); break;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// planner helpers
// ---------------------------------------------------------------------------------------------
// indexorder, src/mapreduce.jl:427-441
void indexorder(const i64* strides, int N, i64* out) {
    for (int i = 0; i < N; ++i) {
        i64 si = std::llabs(strides[i]);
        if (si == 0) {
            out[i] = 1;
            continue;
        }
        i64 k = 1;
        for (int j = 0; j < N; ++j)
            if (strides[j] != 0 && std::llabs(strides[j]) < si) ++k;
        out[i] = k;
    }
}

// _length, src/mapreduce.jl:443-447
i64 length_nonzero(const i64* dims, const i64* strides, int N) {
    i64 l = 1;
    for (int i = 0; i < N; ++i) l *= (strides[i] == 0 ? 1 : dims[i]);
    return l;
}

// _lastargmax, src/mapreduce.jl:452-460 (ties -> LAST index)
int lastargmax(const i64* t, int N) {
    int i = 0;
    for (int j = 1; j < N; ++j)
        if (t[j] >= t[i]) i = j;
    return i;
}

constexpr i64 MINTHREADLENGTH = 1 << 15;  // src/mapreduce.jl:141
constexpr i64 BLOCKMEMORYSIZE = 1 << 15;  // src/mapreduce.jl:462
constexpr i64 CACHELINE = 64;             // src/mapreduce.jl:502

// totalmemoryregion, src/mapreduce.jl:503-520
i64 totalmemoryregion(const i64* dims, int N, const i64 (*bytestrides)[MAXN], int M, int first = 0) {
    i64 region = 0;
    for (int k = 0; k < M; ++k) {
        i64 contig = 0, nblocks = 1;
        for (int i = first; i < N; ++i) {
            i64 d = dims[i], s = bytestrides[k][i];
            if (s < CACHELINE)
                contig += (d - 1) * s;
            else
                nblocks *= d;
        }
        // Julia div() truncates toward zero
        i64 lines = contig / CACHELINE + 1;
        region += CACHELINE * lines * nblocks;
    }
    return region;
}

// _computeblocks, src/mapreduce.jl:463-500 (recursion on the tail = `first` index here)
void computeblocks(const i64* dims, const i64* costs, const i64 (*bytestrides)[MAXN],
                   const i64 (*strideorders)[MAXN], int N, int M, int first, i64* blocks) {
    if (first >= N) return;
    if (totalmemoryregion(dims, N, bytestrides, M, first) <= BLOCKMEMORYSIZE) {
        for (int i = first; i < N; ++i) blocks[i] = dims[i];
        return;
    }
    i64 minorder = INT64_MAX;
    for (int k = 0; k < M; ++k)
        for (int i = first; i < N; ++i) minorder = std::min(minorder, strideorders[k][i]);
    bool allfirst = true;
    for (int k = 0; k < M; ++k)
        if (strideorders[k][first] != minorder) allfirst = false;
    if (allfirst) {  // :477-483
        blocks[first] = dims[first];
        computeblocks(dims, costs, bytestrides, strideorders, N, M, first + 1, blocks);
        return;
    }
    i64 minbs = INT64_MAX;
    for (int k = 0; k < M; ++k)
        for (int i = first; i < N; ++i) minbs = std::min(minbs, bytestrides[k][i]);
    if (minbs > BLOCKMEMORYSIZE) {  // :485-487
        for (int i = first; i < N; ++i) blocks[i] = 1;
        return;
    }
    i64 b[MAXN];
    for (int i = 0; i < N; ++i) b[i] = dims[i];
    i64 w[MAXN];
    auto pick = [&]() {
        for (int i = first; i < N; ++i) w[i - first] = (b[i] - 1) * costs[i];
        return first + lastargmax(w, N - first);
    };
    // Termination guard (not in the reference): `costs` come from the SIGNED minimum stride (:137),
    // so with reversed ranges (negative strides) every candidate weight can be <= 0 and the
    // arg-max lands on a dim whose block is already 1 -- the reference's loops would then spin
    // forever (:491-498).  Blocking only affects the traversal order, so the oracle stops there.
    while (totalmemoryregion(b, N, bytestrides, M, first) >= 2 * BLOCKMEMORYSIZE) {  // :491-494
        int i = pick();
        if (b[i] <= 1) {
            g_guard_hits.fetch_add(1);
            break;
        }
        b[i] = (b[i] + 1) >> 1;
    }
    while (totalmemoryregion(b, N, bytestrides, M, first) > BLOCKMEMORYSIZE) {  // :495-498
        int i = pick();
        if (b[i] <= 1) {
            g_guard_hits.fetch_add(1);
            break;
        }
        b[i] = b[i] - 1;
    }
    for (int i = first; i < N; ++i) blocks[i] = b[i];
}

// The lowered state handed from _mapreduce_order! to _mapreduce_block! and the kernel.
struct Lowered {
    int N = 0, M = 0;
    i64 fused[MAXN];       // dims after _mapreduce_fuse!
    int g = 0;             // bits per importance digit
    i64 importance[MAXN];  // per fused dim
    int perm[MAXN];        // loop order (0-based indices into the fused dims)
    i64 dims[MAXN];        // permuted dims
    i64 strides[MAXM][MAXN];
    i64 offsets[MAXM];
    i64 costs[MAXN];
    i64 blocks[MAXN];
    int esize[MAXM];
};

// _mapreduce_fuse! (src/mapreduce.jl:98-117) + _mapreduce_order! (:119-139) + the planning
// half of _mapreduce_block! (:144-146)
int lower(const smr_problem* p, Lowered& L) {
    int N = p->N, M = p->M;
    if (N < 1 || N > MAXN || M < 1 || M > MAXM) return fail(SMR_EINVAL, "bad N/M");
    L.N = N;
    L.M = M;
    i64 dims[MAXN];
    for (int i = 0; i < N; ++i) {
        dims[i] = p->dims[i];
        if (dims[i] < 1) return fail(SMR_EINVAL, "dims must be >= 1 at the funnel");
    }
    // fuse: for i = N:-1:2, merge dim i into i-1 when contiguous in EVERY operand
    for (int i = N - 1; i >= 1; --i) {
        bool merge = true;
        for (int k = 0; k < M; ++k)
            if (p->ops[k].strides[i] != dims[i - 1] * p->ops[k].strides[i - 1]) {
                merge = false;
                break;
            }
        if (merge) {
            dims[i - 1] *= dims[i];
            dims[i] = 1;
        }
    }
    for (int i = 0; i < N; ++i) L.fused[i] = dims[i];
    // order
    int g = 0;
    {
        uint64_t v = (uint64_t)(M + 1);
        while (v) {
            ++g;
            v >>= 1;
        }  // 64 - leading_zeros(M+1)
    }
    L.g = g;
    i64 ord[MAXN];
    for (int i = 0; i < N; ++i) L.importance[i] = 0;
    for (int k = 0; k < M; ++k) {
        indexorder(p->ops[k].strides, N, ord);
        for (int i = 0; i < N; ++i) {
            i64 term = (i64)1 << (g * (N - ord[i]));
            L.importance[i] += (k == 0 ? 2 : 1) * term;
        }
    }
    for (int i = 0; i < N; ++i)
        if (dims[i] <= 1) L.importance[i] = 0;
    // TupleTools.sortperm(importance; rev=true): stable, descending (tie order is not
    // observable through any reference test; it only changes the loop order).
    for (int i = 0; i < N; ++i) L.perm[i] = i;
    std::stable_sort(L.perm, L.perm + N, [&](int a, int b) { return L.importance[a] > L.importance[b]; });
    for (int i = 0; i < N; ++i) L.dims[i] = dims[L.perm[i]];
    for (int k = 0; k < M; ++k) {
        for (int i = 0; i < N; ++i) L.strides[k][i] = p->ops[k].strides[L.perm[i]];
        L.offsets[k] = p->ops[k].offset;
        L.esize[k] = dtype_size(p->ops[k].dtype);
        if (!L.esize[k]) return fail(SMR_EINVAL, "bad dtype");
    }
    for (int i = 0; i < N; ++i) {
        i64 m = L.strides[0][i];
        for (int k = 1; k < M; ++k) m = std::min(m, L.strides[k][i]);
        L.costs[i] = (m == 0) ? 1 : (m << 1);  // :137
    }
    // blocks
    i64 bytestrides[MAXM][MAXN], strideorders[MAXM][MAXN];
    for (int k = 0; k < M; ++k) {
        for (int i = 0; i < N; ++i) bytestrides[k][i] = L.esize[k] * L.strides[k][i];
        indexorder(L.strides[k], N, strideorders[k]);
    }
    computeblocks(L.dims, L.costs, bytestrides, strideorders, N, M, 0, L.blocks);
    return SMR_OK;
}

// ---------------------------------------------------------------------------------------------
// the kernel: _mapreduce_kernel!, src/mapreduce.jl:229-425 (shape: SURVEY App. C)
// ---------------------------------------------------------------------------------------------
template <class T>
struct RedOp {
    int op;
    inline T operator()(T a, T b) const {
        switch (op) {
            case SMR_RED_ADD: return ops<T>::bin(SMR_OP_ADD, a, b);
            case SMR_RED_MUL: return ops<T>::bin(SMR_OP_MUL, a, b);
            case SMR_RED_MIN: return ops<T>::bin(SMR_OP_MIN, a, b);
            case SMR_RED_MAX: return ops<T>::bin(SMR_OP_MAX, a, b);
            case SMR_RED_AND: return make<T>((ops<T>::truthy(a) && ops<T>::truthy(b)) ? 1 : 0, 0);
            case SMR_RED_OR: return make<T>((ops<T>::truthy(a) || ops<T>::truthy(b)) ? 1 : 0, 0);
        }
        return b;
    }
};
template <class T>
struct InitOp {
    int op;
    T beta;
    inline T operator()(T x) const {
        typedef typename traits<T>::real R;
        switch (op) {
            case SMR_INIT_IDENTITY: return x;
            case SMR_INIT_ZERO: return make<T>(R(0), R(0));
            case SMR_INIT_SCALE: return ops<T>::bin(SMR_OP_MUL, x, beta);
            case SMR_INIT_CONST: return beta;
            case SMR_INIT_CONJ: return conj_of(x);
        }
        return x;
    }
};

struct Operands {
    int M;
    void* base[MAXM];
    int dtype[MAXM];
    int conj[MAXM];
    int esize[MAXM];
};

// Recognised shapes of f get a tight native inner loop (what Julia's specialisation on the
// closure type gives the reference); everything else runs through eval_prog.
enum FKind { F_GENERIC = 0, F_IDENT, F_ADDN, F_SYM };
struct FSpec {
    FKind kind = F_GENERIC;
    int nadd = 0;      // F_ADDN: number of inputs 1..nadd added left to right
    double c = 0;      // F_SYM: (a1 + a2) / c
};
FSpec classify(const Prog& p) {
    FSpec s;
    auto op = [&](int i) { return (int)p.code[2 * i]; };
    auto im = [&](int i) { return (int)p.code[2 * i + 1]; };
    if (p.len == 1 && op(0) == SMR_OP_ARG && im(0) == 1) {
        s.kind = F_IDENT;
        return s;
    }
    // ARG1 ARG2 ADD [ARG3 ADD ...]
    if (p.len >= 3 && (p.len % 2) == 1 && op(0) == SMR_OP_ARG && im(0) == 1) {
        bool ok = true;
        int n = 1;
        for (int i = 1; i + 1 < p.len && ok; i += 2) {
            ++n;
            ok = op(i) == SMR_OP_ARG && im(i) == n && op(i + 1) == SMR_OP_ADD;
        }
        if (ok) {
            s.kind = F_ADDN;
            s.nadd = n;
            return s;
        }
    }
    // ARG1 ARG2 ADD CONSTk DIV  (real constant)
    if (p.len == 5 && op(0) == SMR_OP_ARG && im(0) == 1 && op(1) == SMR_OP_ARG && im(1) == 2 &&
        op(2) == SMR_OP_ADD && op(3) == SMR_OP_CONST && op(4) == SMR_OP_DIV &&
        p.consts[2 * im(3) + 1] == 0.0) {
        s.kind = F_SYM;
        s.c = p.consts[2 * im(3)];
        return s;
    }
    return s;
}

template <class T>
struct Kernel {
    typedef typename traits<T>::real R;
    const Lowered& L;   // dims/strides/blocks (dims & offsets overridden per task)
    const Operands& A;
    const Prog& prog;
    FSpec fs;
    int redop, initop;
    T beta;
    bool uniform;  // all operands have dtype == T's dtype (fast loads)
    int N, M;
    i64 dims[MAXN], blocks[MAXN];
    i64 I[MAXM];    // running linear indices (0-based)
    i64 d[MAXN];    // current block extents

    inline T getA(int k, i64 idx) const {
        T v = uniform ? ((const T*)A.base[k])[idx] : load_as<T>(A.base[k], idx, A.dtype[k]);
        return A.conj[k] ? conj_of(v) : v;
    }
    inline void setA0(i64 idx, T v) const {
        if (A.conj[0]) v = conj_of(v);
        if (uniform)
            ((T*)A.base[0])[idx] = v;
        else
            store_as<T>(A.base[0], idx, A.dtype[0], v);
    }
    inline T fcall(const i64* Ik) const {
        T args[MAXM];
        for (int k = 1; k < M; ++k) args[k] = getA(k, Ik[k]);
        return eval_prog<T>(prog, args);
    }

    // innermost @simd loop, src/mapreduce.jl:318-337
    void inner1() {
        const i64 n = d[0];
        const i64 s0 = L.strides[0][0];
        i64 J[MAXM];
        for (int k = 0; k < M; ++k) J[k] = I[k];
        if (redop == SMR_RED_NONE) {
            if (uniform && !A.conj[0] && fs.kind != F_GENERIC) {
                bool plain = true;
                for (int k = 1; k < M; ++k) plain = plain && !A.conj[k];
                if (plain && native_map(n)) return;
            }
            for (i64 j = 0; j < n; ++j) {
                setA0(J[0], fcall(J));
                for (int k = 0; k < M; ++k) J[k] += L.strides[k][0];
            }
        } else {
            RedOp<T> op{redop};
            if (s0 == 0) {  // hoisted accumulator, :320-327
                T a = getA(0, J[0]);
                for (i64 j = 0; j < n; ++j) {
                    a = op(a, fcall(J));
                    for (int k = 1; k < M; ++k) J[k] += L.strides[k][0];
                }
                setA0(J[0], a);
            } else {
                for (i64 j = 0; j < n; ++j) {
                    setA0(J[0], op(getA(0, J[0]), fcall(J)));
                    for (int k = 0; k < M; ++k) J[k] += L.strides[k][0];
                }
            }
        }
    }

    // Native inner loops for the recognised f shapes (pure map, same dtype, no conj).
    bool native_map(i64 n) {
        T* dst = (T*)A.base[0] + I[0];
        const i64 sd = L.strides[0][0];
        if (fs.kind == F_IDENT && M >= 2) {
            const T* a = (const T*)A.base[1] + I[1];
            const i64 sa = L.strides[1][0];
            if (sd == 1 && sa == 1)
                for (i64 j = 0; j < n; ++j) dst[j] = a[j];
            else
                for (i64 j = 0; j < n; ++j) dst[j * sd] = a[j * sa];
            return true;
        }
        if (fs.kind == F_ADDN && fs.nadd + 1 == M) {
            const T* a[MAXM];
            i64 sa[MAXM];
            for (int k = 1; k < M; ++k) {
                a[k] = (const T*)A.base[k] + I[k];
                sa[k] = L.strides[k][0];
            }
            if (M == 5) {
                for (i64 j = 0; j < n; ++j)
                    dst[j * sd] = ((a[1][j * sa[1]] + a[2][j * sa[2]]) + a[3][j * sa[3]]) + a[4][j * sa[4]];
            } else {
                for (i64 j = 0; j < n; ++j) {
                    T acc = a[1][j * sa[1]];
                    for (int k = 2; k < M; ++k) acc = acc + a[k][j * sa[k]];
                    dst[j * sd] = acc;
                }
            }
            return true;
        }
        if (fs.kind == F_SYM && M == 3) {
            if constexpr (!traits<T>::is_cx) {
                const T* a = (const T*)A.base[1] + I[1];
                const T* b = (const T*)A.base[2] + I[2];
                const i64 sa = L.strides[1][0], sb = L.strides[2][0];
                const T c = (T)fs.c;
                for (i64 j = 0; j < n; ++j) dst[j * sd] = (a[j * sa] + b[j * sb]) / c;
                return true;
            }
        }
        return false;
    }

    // loops j_2 .. j_N around inner1 (src/mapreduce.jl:339-349), level = dim index
    void inner(int level) {
        if (level == 0) {
            inner1();
            return;
        }
        for (i64 j = 0; j < d[level]; ++j) {
            inner(level - 1);
            for (int k = 0; k < M; ++k) I[k] += L.strides[k][level];
        }
        for (int k = 0; k < M; ++k) I[k] -= d[level] * L.strides[k][level];
    }

    // init sub-nest: every destination element of the block once, reduced dims collapsed
    // (d'_i = stride_i_1 == 0 ? 1 : d_i), src/mapreduce.jl:351-375
    void initnest(int level) {
        InitOp<T> io{initop, beta};
        const i64 dp = (L.strides[0][level] == 0) ? 1 : d[level];
        for (i64 j = 0; j < dp; ++j) {
            if (level == 0)
                setA0(I[0], io(getA(0, I[0])));
            else
                initnest(level - 1);
            I[0] += L.strides[0][level];
        }
        I[0] -= dp * L.strides[0][level];
    }

    // block loops J_level (src/mapreduce.jl:385-394 / 403-414); `init_above` = init_{level+1}
    void blockloop(int level, bool init_above) {
        bool init = init_above;  // init_i = init_{i+1}, re-read at every entry
        for (i64 Jb = 0; Jb < dims[level]; Jb += blocks[level]) {
            d[level] = std::min(blocks[level], dims[level] - Jb);
            if (level == 0) {
                if (initop != SMR_INIT_NONE && init) initnest(N - 1);
                inner(N - 1);
            } else {
                blockloop(level - 1, init);
            }
            // :409 reads `init_i &= stride_i_1 > 0`.  Taken literally, a NEGATIVE destination stride along a kept dim switches initop off for
            // every block after the first along that dim: those destination elements are accumulated onto their stale contents, and
            // which elements they are depends on the block sizes (the cache-size constant) and on the thread count (every task starts
            // with init = true again).  No reference test has a reversed destination with an initop; `!= 0` is what the line is for
            // (a reduced dim, stride 0, must not be initialised twice).  The oracle follows the intent; the literal reading is kept
            // selectable (oracle_set_literal_409) so that tests/test_oracle_numpy.py can show the difference against NumPy.
            init = init && (g_literal_409.load(std::memory_order_relaxed) ? L.strides[0][level] > 0 : L.strides[0][level] != 0);
            for (int k = 0; k < M; ++k) I[k] += d[level] * L.strides[k][level];
        }
        for (int k = 0; k < M; ++k) I[k] -= dims[level] * L.strides[k][level];
    }

    void run(const i64* taskdims, const i64* taskoffsets) {
        N = L.N;
        M = L.M;
        for (int i = 0; i < N; ++i) {
            dims[i] = taskdims[i];
            blocks[i] = L.blocks[i];
        }
        for (int k = 0; k < M; ++k) I[k] = taskoffsets[k];
        blockloop(N - 1, true);
    }
};

// ---------------------------------------------------------------------------------------------
// _mapreduce_threaded!, src/mapreduce.jl:195-227
// ---------------------------------------------------------------------------------------------
struct Box {
    i64 dims[MAXN];
    i64 offsets[MAXM];
    int taskindex;
};

template <class Leaf>
void threaded(const Lowered& L, const i64* dims, const i64* offsets, const i64* costs, int nthreads,
              i64 spacing, int taskindex, Leaf&& leaf, bool spawn) {
    const int N = L.N, M = L.M;
    i64 prod = 1;
    for (int i = 0; i < N; ++i) prod *= dims[i];
    auto run_leaf = [&]() {
        Box b;
        for (int i = 0; i < N; ++i) b.dims[i] = dims[i];
        for (int k = 0; k < M; ++k) b.offsets[k] = offsets[k];
        b.offsets[0] = offsets[0] + spacing * (taskindex - 1);
        b.taskindex = taskindex;
        leaf(b);
    };
    if (nthreads == 1 || prod <= MINTHREADLENGTH) {
        run_leaf();
        return;
    }
    i64 w[MAXN];
    for (int i = 0; i < N; ++i) w[i] = (dims[i] - 1) * costs[i];
    int i = lastargmax(w, N);
    if (costs[i] == 0 || dims[i] <= std::min<i64>(L.blocks[i], 1024)) {
        run_leaf();
        return;
    }
    i64 di = dims[i], ndi = di >> 1;
    int nn = nthreads >> 1;
    i64 d1[MAXN], d2[MAXN], o2[MAXM];
    for (int j = 0; j < N; ++j) d1[j] = d2[j] = dims[j];
    d1[i] = ndi;
    d2[i] = di - ndi;
    for (int k = 0; k < M; ++k) o2[k] = offsets[k] + ndi * L.strides[k][i];
    if (spawn) {
        std::thread t([&]() { threaded(L, d1, offsets, costs, nn, spacing, taskindex, leaf, spawn); });
        threaded(L, d2, o2, costs, nthreads - nn, spacing, taskindex + nn, leaf, spawn);
        t.join();
    } else {
        threaded(L, d1, offsets, costs, nn, spacing, taskindex, leaf, spawn);
        threaded(L, d2, o2, costs, nthreads - nn, spacing, taskindex + nn, leaf, spawn);
    }
}

// ---------------------------------------------------------------------------------------------
// _mapreduce_block!, src/mapreduce.jl:142-180
// ---------------------------------------------------------------------------------------------
template <class T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int v = SMR_F32; };
template <> struct dtype_of<double> { static constexpr int v = SMR_F64; };
template <> struct dtype_of<cx<float>> { static constexpr int v = SMR_C32; };
template <> struct dtype_of<cx<double>> { static constexpr int v = SMR_C64; };
template <> struct dtype_of<ix64> { static constexpr int v = SMR_I64; };

template <class T>
int run_typed(const smr_problem* p, const Lowered& L, const Prog& prog, int nthreads) {
    typedef typename traits<T>::real R;
    Operands A;
    A.M = L.M;
    bool uniform = true;
    for (int k = 0; k < L.M; ++k) {
        A.base[k] = p->ops[k].base;
        A.dtype[k] = p->ops[k].dtype;
        A.conj[k] = p->ops[k].conj && (A.dtype[k] == SMR_C32 || A.dtype[k] == SMR_C64);
        A.esize[k] = L.esize[k];
        if (A.dtype[k] != dtype_of<T>::v) uniform = false;
    }
    T beta = make<T>(rcast<R>(p->initarg[0]), rcast<R>(p->initarg[1]));
    FSpec fs = classify(prog);
    auto make_kernel = [&](const Operands& ops_) {
        Kernel<T> K{L, ops_, prog, fs, p->redop, p->initop, beta, uniform};
        return K;
    };
    i64 prod = 1;
    for (int i = 0; i < L.N; ++i) prod *= L.dims[i];

    if (nthreads == 1 || prod <= MINTHREADLENGTH) {  // :151-152
        Kernel<T> K = make_kernel(A);
        K.run(L.dims, L.offsets);
        return SMR_OK;
    }
    if (p->redop != SMR_RED_NONE && length_nonzero(L.dims, L.strides[0], L.N) == 1) {
        // complete reduction, :153-170
        i64 spacing = std::max<i64>(1, 64 / (i64)sizeof(T));
        std::vector<T> threadedout((size_t)(spacing * nthreads));
        // a = arrays[1][ParentIndex(1)]
        T a = load_as<T>(A.base[0], L.offsets[0], A.dtype[0]);
        if (A.conj[0]) a = conj_of(a);
        if (p->initop != SMR_INIT_NONE) a = InitOp<T>{p->initop, beta}(a);
        // _init_reduction!, :182-187
        T neutral;
        switch (p->redop) {
            case SMR_RED_ADD: neutral = make<T>(R(0), R(0)); break;
            case SMR_RED_MUL: neutral = make<T>(R(1), R(0)); break;
            case SMR_RED_AND: neutral = make<T>(R(1), R(0)); break;  // true, :188
            case SMR_RED_OR: neutral = make<T>(R(0), R(0)); break;   // false, :189
            default: neutral = a; break;  // min / max: fill with a
        }
        for (auto& v : threadedout) v = neutral;
        Operands A2 = A;
        A2.base[0] = threadedout.data();
        A2.dtype[0] = dtype_of<T>::v;
        A2.conj[0] = 0;
        bool uniform2 = true;
        for (int k = 0; k < L.M; ++k)
            if (A2.dtype[k] != dtype_of<T>::v) uniform2 = false;
        Lowered L2 = L;
        i64 offs[MAXM];
        for (int k = 0; k < L.M; ++k) offs[k] = L.offsets[k];
        offs[0] = 0;
        auto leaf = [&](const Box& b) {
            Kernel<T> K{L2, A2, prog, fs, p->redop, SMR_INIT_NONE, beta, uniform2};
            K.run(b.dims, b.offsets);
        };
        threaded(L2, L.dims, offs, L.costs, nthreads, spacing, 1, leaf, true);
        RedOp<T> op{p->redop};
        for (int i = 0; i < nthreads; ++i) a = op(a, threadedout[(size_t)(i * spacing)]);  // :167-169
        if (A.conj[0]) a = conj_of(a);
        store_as<T>(A.base[0], L.offsets[0], A.dtype[0], a);
        return SMR_OK;
    }
    // :171-177 -- reduction dims must never be split
    i64 costs[MAXN];
    for (int i = 0; i < L.N; ++i) costs[i] = L.costs[i] * (L.strides[0][i] != 0 ? 1 : 0);
    auto leaf = [&](const Box& b) {
        Kernel<T> K = make_kernel(A);
        K.run(b.dims, b.offsets);
    };
    threaded(L, L.dims, L.offsets, costs, nthreads, 0, 1, leaf, true);
    return SMR_OK;
}

// Julia's integer typing of the f-program, operation by operation (Base promotion rules: the wider type wins, equal widths of
// mixed signedness give the unsigned type, Bool yields to every integer type and Bool (+,-,*) Bool is an Int, an integer literal is
// an Int64).  Fills prog.wrap_bits / wrap_signed for every operation whose Julia type is narrower than 64 bits.
void julia_int_types(const smr_problem* p, Prog& prog) {
    struct Ty {
        int bits;  // 1 = Bool, 8, 16, 32, 64
        bool sgn;
    };
    auto of_dtype = [](int dt) {
        Ty t;
        t.bits = dt == SMR_BOOL ? 1 : 8 << ((dt - SMR_I8) & 3);
        t.sgn = dt < SMR_U8;
        return t;
    };
    auto promote = [](Ty a, Ty b) {
        if (a.bits == 1) return b;
        if (b.bits == 1) return a;
        if (a.bits != b.bits) return a.bits > b.bits ? a : b;
        Ty t = a;
        t.sgn = a.sgn && b.sgn;
        return t;
    };
    Ty st[SMR_MAXPROG];
    int sp = 0;
    for (int pc = 0; pc < prog.len; ++pc) {
        const int op = prog.code[2 * pc], imm = prog.code[2 * pc + 1];
        bool arith = false;
        if (op == SMR_OP_ARG) {
            st[sp++] = of_dtype(p->ops[imm].dtype);
        } else if (op == SMR_OP_CONST) {
            st[sp++] = Ty{64, true};
        } else if (op < 32) {
            Ty& a = st[sp - 1];
            if (op == SMR_OP_WIDEN) a = Ty{64, true};
            if (op == SMR_OP_NEG || op == SMR_OP_ABS || op == SMR_OP_ABS2) {
                if (a.bits == 1 && op == SMR_OP_NEG) a = Ty{64, true};  // -true is an Int
                arith = true;
            }
        } else if (op < 64) {
            const Ty b = st[--sp];
            Ty& a = st[sp - 1];
            if (op >= SMR_OP_LT && op <= SMR_OP_NE) {
                a = Ty{1, false};
            } else {
                const bool bools = a.bits == 1 && b.bits == 1;
                a = promote(a, b);
                if (op == SMR_OP_ADD || op == SMR_OP_SUB || op == SMR_OP_MUL) {
                    if (bools && op != SMR_OP_MUL) a = Ty{64, true};  // Bool (+,-) Bool is an Int, Bool * Bool a Bool (base/bool.jl)
                    arith = true;
                }
            }
        } else {
            const Ty c = st[--sp], b = st[--sp];
            st[sp - 1] = promote(b, c);
        }
        const Ty r = st[sp - 1];
        prog.wrap_bits[pc] = (uint8_t)((arith && r.bits > 1 && r.bits < 64) ? r.bits : 0);
        prog.wrap_signed[pc] = r.sgn;
    }
}

// the integer class applies when every operand is an integer type and f is closed over the integers (the same rule as
// the device planner, csrc/smr_plan.cpp: canonicalise)
bool integer_class(const smr_problem* p, const Prog& prog) {
    bool has_u64 = false, has_signed = false, has_const = false, eqne = false;
    for (int k = 0; k < p->M; ++k) {
        if (p->ops[k].dtype < SMR_I8) return false;
        if (p->ops[k].dtype == SMR_U64) has_u64 = true;
        if (p->ops[k].dtype >= SMR_I8 && p->ops[k].dtype <= SMR_I64) has_signed = true;
    }
    bool ordered = p->redop == SMR_RED_MIN || p->redop == SMR_RED_MAX;
    for (int pc = 0; pc < prog.len; ++pc) {
        const int op = prog.code[2 * pc];
        switch (op) {
            case SMR_OP_ARG: case SMR_OP_NEG: case SMR_OP_ABS2: case SMR_OP_CONJ: case SMR_OP_REAL: case SMR_OP_IMAG:
            case SMR_OP_ADD: case SMR_OP_SUB: case SMR_OP_MUL: case SMR_OP_SELECT: case SMR_OP_WIDEN: break;
            case SMR_OP_ABS: case SMR_OP_MIN: case SMR_OP_MAX: case SMR_OP_LT: case SMR_OP_LE: case SMR_OP_GT: case SMR_OP_GE:
                ordered = true;
                break;
            // == and !=: Julia compares a UInt64 with a signed value mathematically, a 64-bit signed domain compares bit patterns; among
            // unsigned values (no signed operand, no literal: literals are Int64) the two are the same
            case SMR_OP_EQ: case SMR_OP_NE:
                eqne = true;
                break;
            case SMR_OP_CONST: {
                has_const = true;
                const double re = prog.consts[2 * prog.code[2 * pc + 1]], im = prog.consts[2 * prog.code[2 * pc + 1] + 1];
                if (im != 0.0 || !(re == std::floor(re) || std::isinf(re)) || (std::fabs(re) > 9223372036854775808.0 && !std::isinf(re))) return false;
                break;
            }
            default: return false;
        }
    }
    if (p->initop == SMR_INIT_SCALE || p->initop == SMR_INIT_CONST) {
        const double re = p->initarg[0], im = p->initarg[1];
        if (im != 0.0 || re != std::floor(re) || std::fabs(re) > 9223372036854775808.0) return false;
    }
    if (eqne && has_u64 && (has_signed || has_const)) ordered = true;
    return !(has_u64 && ordered);
}

int compute_class(const smr_problem* p) {
    // Julia's promote_type over the operand eltypes restricted to the four float classes.
    bool dbl = false, cplx = false;
    for (int k = 0; k < p->M; ++k) {
        int dt = p->ops[k].dtype;
        if (dt == SMR_F64 || dt == SMR_C64) dbl = true;
        if (dt == SMR_C32 || dt == SMR_C64) cplx = true;
        if (dt >= SMR_I8) dbl = true;  // Int with Float32 would stay Float32 in Julia, but
                                       // counting reductions need exact integers: use f64
    }
    for (int i = 0; i < p->nconsts; ++i)
        if (p->fconsts[2 * i + 1] != 0.0) cplx = true;
    for (int pc = 0; pc < p->fprog_len; ++pc)
        if (p->fprog[2 * pc] == SMR_OP_WIDEN) dbl = true;  // a Float64 scalar among Float32 arrays: Julia computes in Float64
    return cplx ? (dbl ? SMR_C64 : SMR_C32) : (dbl ? SMR_F64 : SMR_F32);
}

// Pure copy of non-float data: moved as opaque bytes through the same loop nest.
int run_bitcopy(const smr_problem* p, const Lowered& L) {
    const int es = L.esize[0];
    // lowered dims/strides/offsets; iterate the box in the planned order (any order is
    // equivalent for a pure move)
    i64 idx[MAXN] = {0};
    const char* src = (const char*)p->ops[1].base;
    char* dst = (char*)p->ops[0].base;
    i64 I0 = L.offsets[0], I1 = L.offsets[1];
    while (true) {
        std::memcpy(dst + I0 * es, src + I1 * es, (size_t)es);
        int i = 0;
        for (; i < L.N; ++i) {
            I0 += L.strides[0][i];
            I1 += L.strides[1][i];
            if (++idx[i] < L.dims[i]) break;
            I0 -= L.dims[i] * L.strides[0][i];
            I1 -= L.dims[i] * L.strides[1][i];
            idx[i] = 0;
        }
        if (i == L.N) break;
    }
    return SMR_OK;
}

int load_prog(const smr_problem* p, Prog& prog) {
    if (p->fprog == nullptr || p->fprog_len == 0) {
        prog.len = 1;
        prog.code[0] = SMR_OP_ARG;
        prog.code[1] = 1;
    } else {
        if (p->fprog_len > SMR_MAXPROG) return fail(SMR_EINVAL, "program too long");
        prog.len = p->fprog_len;
        std::memcpy(prog.code, p->fprog, (size_t)(2 * p->fprog_len));
    }
    if (p->nconsts > SMR_MAXCONST) return fail(SMR_EINVAL, "too many constants");
    prog.nconst = p->nconsts;
    for (int i = 0; i < 2 * p->nconsts; ++i) prog.consts[i] = p->fconsts[i];
    if (check_prog(prog, std::max(p->M, 2)) < 0) return fail(SMR_EINVAL, "malformed f-program");
    return SMR_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported test API (host pointers)
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

// 1: read src/mapreduce.jl:409 literally (see Kernel::blockloop); default 0
void oracle_set_literal_409(int on) { g_literal_409.store(on != 0); }
// number of times the termination guard of _computeblocks fired since the last reset (reset != 0 clears it after reading)
long oracle_guard_hits(int reset) {
    const long n = g_guard_hits.load();
    if (reset) g_guard_hits.store(0);
    return n;
}

// Full reference path on host memory: _mapreduce_fuse! -> ... -> _mapreduce_kernel!
int oracle_mapreduce(const smr_problem* p, int nthreads) {
    if (!p) return fail(SMR_EINVAL, "null problem");
    Prog prog;
    int rc = load_prog(p, prog);
    if (rc) return rc;
    Lowered L;
    rc = lower(p, L);
    if (rc) return rc;
    if (nthreads < 1) nthreads = 1;
    // integer data: only pure moves between equal dtypes
    bool anyint = false;
    for (int k = 0; k < p->M; ++k)
        if (p->ops[k].dtype >= SMR_I8) anyint = true;
    if (anyint && p->redop == SMR_RED_NONE && p->M == 2 && p->ops[0].dtype == p->ops[1].dtype &&
        prog.len == 1 && prog.code[0] == SMR_OP_ARG)
        return run_bitcopy(p, L);
    if (integer_class(p, prog)) {
        julia_int_types(p, prog);
        return run_typed<ix64>(p, L, prog, nthreads);
    }
    switch (compute_class(p)) {
        case SMR_F32: return run_typed<float>(p, L, prog, nthreads);
        case SMR_F64: return run_typed<double>(p, L, prog, nthreads);
        case SMR_C32: return run_typed<cx<float>>(p, L, prog, nthreads);
        case SMR_C64: return run_typed<cx<double>>(p, L, prog, nthreads);
    }
    return fail(SMR_EINVAL, "bad compute class");
}

// Planner introspection for the known-answer tests.
typedef struct oracle_plan_info {
    int32_t N, M, g, _pad;
    int64_t fused[SMR_MAXN];
    int64_t importance[SMR_MAXN];
    int32_t perm[SMR_MAXN]; /* 0-based */
    int64_t dims[SMR_MAXN];
    int64_t strides[SMR_MAXM][SMR_MAXN];
    int64_t costs[SMR_MAXN];
    int64_t blocks[SMR_MAXN];
} oracle_plan_info;

int oracle_plan(const smr_problem* p, oracle_plan_info* out) {
    Lowered L;
    int rc = lower(p, L);
    if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    out->N = L.N;
    out->M = L.M;
    out->g = L.g;
    for (int i = 0; i < L.N; ++i) {
        out->fused[i] = L.fused[i];
        out->importance[i] = L.importance[i];
        out->perm[i] = L.perm[i];
        out->dims[i] = L.dims[i];
        out->costs[i] = L.costs[i];
        out->blocks[i] = L.blocks[i];
    }
    for (int k = 0; k < L.M; ++k)
        for (int i = 0; i < L.N; ++i) out->strides[k][i] = L.strides[k][i];
    return SMR_OK;
}

// The leaf boxes _mapreduce_threaded! would hand to tasks (map / partial-reduction mode).
// boxes: maxboxes x (N dims followed by M offsets).  Returns the number of boxes.
int oracle_threaded_boxes(const smr_problem* p, int nthreads, int64_t* boxes, int maxboxes) {
    Lowered L;
    int rc = lower(p, L);
    if (rc) return rc;
    i64 costs[MAXN];
    for (int i = 0; i < L.N; ++i) costs[i] = L.costs[i] * (L.strides[0][i] != 0 ? 1 : 0);
    int n = 0;
    auto leaf = [&](const Box& b) {
        if (n < maxboxes) {
            int64_t* o = boxes + (size_t)n * (L.N + L.M);
            for (int i = 0; i < L.N; ++i) o[i] = b.dims[i];
            for (int k = 0; k < L.M; ++k) o[L.N + k] = b.offsets[k];
        }
        ++n;
    };
    threaded(L, L.dims, L.offsets, costs, nthreads, 0, 1, leaf, false);
    return n;
}

void oracle_indexorder(const int64_t* strides, int n, int64_t* out) { indexorder(strides, n, out); }

int oracle_host_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
