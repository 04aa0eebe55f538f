// smr_device.h -- device-side building blocks: element types, loads/stores with the
// StridedView `op` (identity/conj) applied, the fused elementwise functors and the
// wave-uniform f-program interpreter.  gfx950 only (wave64).
//
// Compiled with -ffp-contract=off: Julia never contracts (a+b)/2 or c*x+y into FMAs, and
// the map results must match the reference bit-for-bit for + - * / (SURVEY section 7).
#pragma once

#ifndef SMR_JIT
#include <hip/hip_runtime.h>
#endif

#include "smr_internal.h"

namespace smr {

#define SMR_DEV __device__ __forceinline__

// ---- device-side wall-clock stamps (debug build: make stamp -> libstrided_hip_stamp.so, -DSMR_STAMP=1) --------
// Profiler-independent timing of a launch: every wave records s_memrealtime (100 MHz, one clock for the whole
// device) at entry and after its last store has been acknowledged, into its own 16-byte slot of the region the
// launcher reserved for this launch (smr_set_option "stamp_base" / "stamp_cap" / "stamp_used"; tools/device_span.py
// turns the slots into first-start, last-end, cadence and inter-launch gap).  The product build compiles none of it.
#ifndef SMR_STAMP
#define SMR_STAMP 0
#endif
#if SMR_STAMP && !defined(SMR_JIT)
#define SMR_STAMP_PARAM , unsigned long long* smr_stamps
#define SMR_STAMP_ARG(grid, block) , ::smr::stamp_next((size_t)(grid) * (((size_t)(block) + 63) / 64))
#define SMR_STAMP_BEGIN const unsigned long long smr_t0 = (unsigned long long)wall_clock64();
#define SMR_STAMP_END ::smr::stamp_end(smr_stamps, smr_t0);
#else
#define SMR_STAMP_PARAM
#define SMR_STAMP_ARG(grid, block)
#define SMR_STAMP_BEGIN
#define SMR_STAMP_END
#endif
#if SMR_STAMP && !defined(SMR_JIT)
__device__ __forceinline__ void stamp_end(unsigned long long* stamps, unsigned long long t0) {
    if (!stamps) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = (unsigned long long)wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6)) * 2;
        o[0] = t0;
        o[1] = t1;
    }
}
#endif

template <class R>
struct alignas(2 * sizeof(R)) cplx {
    R re, im;
};
typedef cplx<float> c32;
typedef cplx<double> c64;

// opaque movers for bit copies of 1- and 2-byte integers
struct b8 {
    uint8_t v;
};
struct b16 {
    uint16_t v;
};

template <class T> struct tr;
template <> struct tr<float> { typedef float real; static constexpr bool cx = false; static constexpr bool arith = true; static constexpr int dt = SMR_F32; };
template <> struct tr<double> { typedef double real; static constexpr bool cx = false; static constexpr bool arith = true; static constexpr int dt = SMR_F64; };
template <> struct tr<c32> { typedef float real; static constexpr bool cx = true; static constexpr bool arith = true; static constexpr int dt = SMR_C32; };
template <> struct tr<c64> { typedef double real; static constexpr bool cx = true; static constexpr bool arith = true; static constexpr int dt = SMR_C64; };
// the integer compute class: 64-bit two's-complement, wrapping (Julia's Int64 arithmetic); compiled with -fwrapv
typedef long long ix64;
template <> struct tr<ix64> { typedef ix64 real; static constexpr bool cx = false; static constexpr bool arith = true; static constexpr int dt = SMR_I64; };
template <> struct tr<b8> { typedef float real; static constexpr bool cx = false; static constexpr bool arith = false; static constexpr int dt = SMR_U8; };
template <> struct tr<b16> { typedef float real; static constexpr bool cx = false; static constexpr bool arith = false; static constexpr int dt = SMR_U16; };

// ---- construction / parts ----------------------------------------------------------------
// 16-/8-/4-byte vector store, plain or non-temporal.  A run-time switch between the two must not be written
// as if (nts) store_nt else store: the optimiser hoists/sinks the pair into ONE store and drops `nt`
// (seen in the ISA).  store_vec() keeps them apart with asm statements; loops over several stores
// branch once and bracket the non-temporal block with nt_block_guard().
// Store policy of a launch (kernel argument `nts`): 0 plain, 1 non-temporal, 2 agent-scope WRITE-THROUGH.
// Policy 2 ("self-released" launches, round 5): every store carries sc1 -- it is written through the XCD's L2 to the memory side, the
// level all eight XCDs share -- and the wave waits for the acknowledgements before it ends (self_release_wait).  When such a kernel
// has completed, nothing it wrote sits dirty in an L2: the release fence of its dispatch packet (a write-back of all eight L2s by the
// packet processor) has nothing left to do, and a recorded sequence drops it (smr_seq.cpp).  A later reader still acquires.
template <bool NT, class VT>
SMR_DEV void store_vec_ct(char* p, const VT& v) {
    // (the vector types carry 4-byte alignment: callers may hand in element-aligned addresses -- STREAM's UVec -- and the hardware
    // takes dwordx2 / dwordx4 accesses at any dword address; the instruction selected is the same)
    if constexpr (NT && sizeof(VT) == 16) {
        typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
        __builtin_nontemporal_store(*reinterpret_cast<const u4a*>(&v), reinterpret_cast<u4a*>(p));
    } else if constexpr (NT && sizeof(VT) == 8) {
        typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(4)));
        __builtin_nontemporal_store(*reinterpret_cast<const u2a*>(&v), reinterpret_cast<u2a*>(p));
    } else if constexpr (NT && sizeof(VT) == 4) {
        __builtin_nontemporal_store(*reinterpret_cast<const uint32_t*>(&v), reinterpret_cast<uint32_t*>(p));
    } else {
        *reinterpret_cast<VT*>(p) = v;
    }
}
// is there a write-through form for this vector type?  (launchers offer policy 2 only where every store of the kernel has one)
template <class VT>
struct has_wt_store {
    static constexpr bool value = sizeof(VT) == 16 || sizeof(VT) == 8 || sizeof(VT) == 4;
};
template <class VT>
SMR_DEV void store_vec_wt(char* p, const VT& v) {
    if constexpr (sizeof(VT) == 16) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 x = *reinterpret_cast<const u4*>(&v);
        // The s_nop belongs to the store: on gfx940+ a VALU instruction that writes a data VGPR of a store of more than 64 bits needs
        // two wait states behind it (the store is still reading the register).  For its own stores the compiler inserts them; an
        // inline-asm statement is opaque to its hazard recogniser, and the register allocator does reuse the data registers for the
        // next address right away (seen in the ISA: wrong sums until the s_nop was added).
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    } else if constexpr (sizeof(VT) == 8) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 x = *reinterpret_cast<const u2*>(&v);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    } else if constexpr (sizeof(VT) == 4) {
        const uint32_t x = *reinterpret_cast<const uint32_t*>(&v);
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    } else {
        *reinterpret_cast<VT*>(p) = v;  // (never offered: has_wt_store)
    }
}
// every store of this wave has been acknowledged by the memory side (policy 2: the last thing a wave does)
SMR_DEV void self_release_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
SMR_DEV void nt_block_guard() { asm volatile("; nt stores" ::: "memory"); }

// vector load, plain or non-temporal by a compile-time switch (a run-time switch belongs around the whole loop:
// a diamond per load is folded like the stores, and guarding each one keeps the loads of a batch apart)
template <int I>
struct IntC {
    static constexpr int value = I;
};
template <bool B>
struct BoolC {
    static constexpr bool value = B;
};
template <bool NT, class VT>
SMR_DEV VT load_vec_ct(const void* p) {
    if constexpr (NT && sizeof(VT) == 16) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        VT v;
        *reinterpret_cast<u4*>(&v) = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
        return v;
    } else {
        return *reinterpret_cast<const VT*>(p);
    }
}

template <class VT>
SMR_DEV void store_vec(char* p, const VT& v, int nts) {
    if (nts == 2) {
        store_vec_wt<VT>(p, v);
    } else if (nts) {
        nt_block_guard();
        store_vec_ct<true>(p, v);
        nt_block_guard();
    } else {
        store_vec_ct<false>(p, v);
    }
}

template <class T> struct is_int_class { static constexpr bool value = false; };
template <> struct is_int_class<ix64> { static constexpr bool value = true; };
// double -> the real type of a compute class; saturating for the integer class (the +-Inf seeds of min / max
// reductions become typemax / typemin, as _init_reduction! would give them)
template <class R> SMR_DEV R rcast(double d) { return R(d); }
template <> SMR_DEV ix64 rcast<ix64>(double d) {
    return d >= 9223372036854775807.0 ? 9223372036854775807LL : (d <= -9223372036854775808.0 ? (-9223372036854775807LL - 1) : (ix64)d);
}
template <class T> SMR_DEV T mk(typename tr<T>::real re, typename tr<T>::real im);
template <> SMR_DEV ix64 mk<ix64>(ix64 re, ix64) { return re; }
template <> SMR_DEV float mk<float>(float re, float) { return re; }
template <> SMR_DEV double mk<double>(double re, double) { return re; }
template <> SMR_DEV c32 mk<c32>(float re, float im) { return c32{re, im}; }
template <> SMR_DEV c64 mk<c64>(double re, double im) { return c64{re, im}; }

SMR_DEV float re_(float x) { return x; }
SMR_DEV double re_(double x) { return x; }
template <class R> SMR_DEV R re_(cplx<R> x) { return x.re; }
SMR_DEV ix64 re_(ix64 x) { return x; }
SMR_DEV ix64 im_(ix64) { return 0; }
SMR_DEV ix64 cj(ix64 x) { return x; }
SMR_DEV float im_(float) { return 0.f; }
SMR_DEV double im_(double) { return 0.; }
template <class R> SMR_DEV R im_(cplx<R> x) { return x.im; }

SMR_DEV float cj(float x) { return x; }
SMR_DEV double cj(double x) { return x; }
template <class R> SMR_DEV cplx<R> cj(cplx<R> x) { return cplx<R>{x.re, -x.im}; }
SMR_DEV b8 cj(b8 x) { return x; }
SMR_DEV b16 cj(b16 x) { return x; }

// ---- arithmetic (complex: Julia's plain 4-multiply product, Base complex.jl) -------------
template <class R> SMR_DEV cplx<R> operator+(cplx<R> a, cplx<R> b) { return cplx<R>{a.re + b.re, a.im + b.im}; }
template <class R> SMR_DEV cplx<R> operator-(cplx<R> a, cplx<R> b) { return cplx<R>{a.re - b.re, a.im - b.im}; }
template <class R> SMR_DEV cplx<R> operator-(cplx<R> a) { return cplx<R>{-a.re, -a.im}; }
template <class R> SMR_DEV cplx<R> operator*(cplx<R> a, cplx<R> b) {
    return cplx<R>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class R> SMR_DEV cplx<R> operator/(cplx<R> a, cplx<R> b) {
    // scaled (Smith) division: avoids overflow of |b|^2
    if (fabs(b.re) >= fabs(b.im)) {
        R r = b.im / b.re, den = b.re + b.im * r;
        return cplx<R>{(a.re + a.im * r) / den, (a.im - a.re * r) / den};
    } else {
        R r = b.re / b.im, den = b.re * r + b.im;
        return cplx<R>{(a.re * r + a.im) / den, (a.im * r - a.re) / den};
    }
}

template <class T> struct mathx;  // unary / binary ops of the f-program per compute type

template <class R>
struct mathx_real {
    static SMR_DEV R un(int op, R a) {
        switch (op) {
            case SMR_OP_NEG: return -a;
            case SMR_OP_ABS: return fabs(a);
            case SMR_OP_ABS2: return a * a;
            case SMR_OP_CONJ: return a;
            case SMR_OP_REAL: return a;
            case SMR_OP_IMAG: return R(0);
            case SMR_OP_SQRT: return sqrt(a);
            case SMR_OP_EXP: return exp(a);
            case SMR_OP_LOG: return log(a);
            case SMR_OP_SIN: return sin(a);
            case SMR_OP_COS: return cos(a);
            case SMR_OP_TANH: return tanh(a);
            case SMR_OP_INV: return R(1) / a;
            case SMR_OP_ROUND32: return (R)(float)a;
        }
        return a;
    }
    static SMR_DEV R bin(int op, R a, R b) {
        switch (op) {
            case SMR_OP_ADD: return a + b;
            case SMR_OP_SUB: return a - b;
            case SMR_OP_MUL: return a * b;
            case SMR_OP_DIV: return a / b;
            // Julia's min / max (Base/math.jl): NaN if either argument is NaN; min(-0.0, 0.0) = -0.0, max(-0.0, 0.0) = 0.0
            case SMR_OP_MIN: return (a != a) ? a : ((b != b) ? b : ((b < a) ? b : ((a < b) ? a : (__builtin_signbit(a) ? a : b))));
            case SMR_OP_MAX: return (a != a) ? a : ((b != b) ? b : ((a < b) ? b : ((b < a) ? a : (__builtin_signbit(a) ? b : a))));
            case SMR_OP_LT: return a < b ? R(1) : R(0);
            case SMR_OP_LE: return a <= b ? R(1) : R(0);
            case SMR_OP_GT: return a > b ? R(1) : R(0);
            case SMR_OP_GE: return a >= b ? R(1) : R(0);
            case SMR_OP_EQ: return a == b ? R(1) : R(0);
            case SMR_OP_NE: return a != b ? R(1) : R(0);
        }
        return a;
    }
    static SMR_DEV bool truthy(R a) { return a != R(0); }
};
template <> struct mathx<float> : mathx_real<float> {};
template <> struct mathx<double> : mathx_real<double> {};

// integer class: every operation is closed over Int64 and wraps (unsigned arithmetic underneath: no UB);
// the planner admits only these opcodes (csrc/smr_plan.cpp: canonicalise)
template <> struct mathx<ix64> {
    typedef unsigned long long U;
    static SMR_DEV ix64 un(int op, ix64 a) {
        switch (op) {
            case SMR_OP_NEG: return (ix64)(U(0) - (U)a);
            case SMR_OP_ABS: return a < 0 ? (ix64)(U(0) - (U)a) : a;  // abs(typemin) = typemin, like Julia
            case SMR_OP_ABS2: return (ix64)((U)a * (U)a);
            case SMR_OP_IMAG: return 0;
            // Julia's narrow integer types, where their wrapping would be observed (inserted by the planner, csrc/smr_plan.cpp)
            case SMR_OP_WRAP_I8: return (ix64)(signed char)a;
            case SMR_OP_WRAP_I16: return (ix64)(short)a;
            case SMR_OP_WRAP_I32: return (ix64)(int)a;
            case SMR_OP_WRAP_U8: return a & 0xff;
            case SMR_OP_WRAP_U16: return a & 0xffff;
            case SMR_OP_WRAP_U32: return a & 0xffffffffLL;
        }
        return a;  // conj, real, round32 / widen (never emitted for integers)
    }
    static SMR_DEV ix64 bin(int op, ix64 a, ix64 b) {
        switch (op) {
            case SMR_OP_ADD: return (ix64)((U)a + (U)b);
            case SMR_OP_SUB: return (ix64)((U)a - (U)b);
            case SMR_OP_MUL: return (ix64)((U)a * (U)b);
            case SMR_OP_DIV: return b == 0 ? 0 : ((a == (-9223372036854775807LL - 1) && b == -1) ? a : a / b);  // not admitted by the planner (Julia's `/` leaves the integers)
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
    static SMR_DEV bool truthy(ix64 a) { return a != 0; }
};

template <class R>
struct mathx_cx {
    typedef cplx<R> T;
    static SMR_DEV T csqrt(T a) {
        R r = hypot(a.re, a.im);
        if (r == R(0)) return T{R(0), a.im};
        if (a.re >= R(0)) {
            R t = sqrt((r + a.re) * R(0.5));
            return T{t, a.im / (t + t)};
        }
        R t = sqrt((r - a.re) * R(0.5));
        return T{fabs(a.im) / (t + t), copysign(t, a.im)};
    }
    static SMR_DEV T un(int op, T a) {
        switch (op) {
            case SMR_OP_NEG: return -a;
            case SMR_OP_ABS: return T{hypot(a.re, a.im), R(0)};
            case SMR_OP_ABS2: return T{a.re * a.re + a.im * a.im, R(0)};
            case SMR_OP_CONJ: return T{a.re, -a.im};
            case SMR_OP_REAL: return T{a.re, R(0)};
            case SMR_OP_IMAG: return T{a.im, R(0)};
            case SMR_OP_SQRT: return csqrt(a);
            case SMR_OP_EXP: {
                R e = exp(a.re);
                return T{e * cos(a.im), e * sin(a.im)};
            }
            case SMR_OP_LOG: return T{log(hypot(a.re, a.im)), atan2(a.im, a.re)};
            case SMR_OP_SIN: return T{sin(a.re) * cosh(a.im), cos(a.re) * sinh(a.im)};
            case SMR_OP_COS: return T{cos(a.re) * cosh(a.im), -sin(a.re) * sinh(a.im)};
            case SMR_OP_TANH: {
                R x2 = a.re + a.re, y2 = a.im + a.im;
                R den = cosh(x2) + cos(y2);
                return T{sinh(x2) / den, sin(y2) / den};
            }
            case SMR_OP_INV: return T{R(1), R(0)} / a;
            case SMR_OP_ROUND32: return T{(R)(float)a.re, (R)(float)a.im};
        }
        return a;
    }
    static SMR_DEV T bin(int op, T a, T b) {
        switch (op) {
            case SMR_OP_ADD: return a + b;
            case SMR_OP_SUB: return a - b;
            case SMR_OP_MUL: return a * b;
            case SMR_OP_DIV: return a / b;
            case SMR_OP_MIN: return (b.re < a.re) ? b : a;
            case SMR_OP_MAX: return (a.re < b.re) ? b : a;
            case SMR_OP_LT: return T{a.re < b.re ? R(1) : R(0), R(0)};
            case SMR_OP_LE: return T{a.re <= b.re ? R(1) : R(0), R(0)};
            case SMR_OP_GT: return T{a.re > b.re ? R(1) : R(0), R(0)};
            case SMR_OP_GE: return T{a.re >= b.re ? R(1) : R(0), R(0)};
            case SMR_OP_EQ: return T{(a.re == b.re && a.im == b.im) ? R(1) : R(0), R(0)};
            case SMR_OP_NE: return T{(a.re != b.re || a.im != b.im) ? R(1) : R(0), R(0)};
        }
        return a;
    }
    static SMR_DEV bool truthy(T a) { return a.re != R(0); }
};
template <> struct mathx<c32> : mathx_cx<float> {};
template <> struct mathx<c64> : mathx_cx<double> {};

// ---- loads / stores ------------------------------------------------------------------------
// Typed fast path: the operand's dtype is the compute type.
template <class T>
SMR_DEV T ld(const void* base, i64 idx) {
    return ((const T*)base)[idx];
}
template <class T>
SMR_DEV void st(void* base, i64 idx, T v) {
    ((T*)base)[idx] = v;
}
// Mixed path: convert from/to the operand's own dtype (wave-uniform switch).
template <class T>
SMR_DEV T ld_as(const void* base, i64 idx, int dt) {
    typedef typename tr<T>::real R;
    if constexpr (is_int_class<T>::value) {  // sign / zero extension; floating operands never reach this class
        switch (dt) {
            case SMR_I8: return ((const int8_t*)base)[idx];
            case SMR_U8: return ((const uint8_t*)base)[idx];
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
        case SMR_F32: return mk<T>(R(((const float*)base)[idx]), R(0));
        case SMR_F64: return mk<T>(R(((const double*)base)[idx]), R(0));
        case SMR_C32: { c32 v = ((const c32*)base)[idx]; return mk<T>(R(v.re), R(v.im)); }
        case SMR_C64: { c64 v = ((const c64*)base)[idx]; return mk<T>(R(v.re), R(v.im)); }
        case SMR_I8: return mk<T>(R(((const int8_t*)base)[idx]), R(0));
        case SMR_U8: return mk<T>(R(((const uint8_t*)base)[idx]), R(0));
        case SMR_I16: return mk<T>(R(((const int16_t*)base)[idx]), R(0));
        case SMR_U16: return mk<T>(R(((const uint16_t*)base)[idx]), R(0));
        case SMR_I32: return mk<T>(R(((const int32_t*)base)[idx]), R(0));
        case SMR_U32: return mk<T>(R(((const uint32_t*)base)[idx]), R(0));
        case SMR_I64: return mk<T>(R(((const long long*)base)[idx]), R(0));
        case SMR_U64: return mk<T>(R(((const unsigned long long*)base)[idx]), R(0));
    }
    return mk<T>(R(0), R(0));
}
template <class T>
SMR_DEV void st_as(void* base, i64 idx, int dt, T v) {
    if constexpr (is_int_class<T>::value) {  // truncation = the wrapped value in the narrower type
        switch (dt) {
            case SMR_I8: case SMR_U8: ((uint8_t*)base)[idx] = (uint8_t)v; break;
            case SMR_I16: case SMR_U16: ((uint16_t*)base)[idx] = (uint16_t)v; break;
            case SMR_I32: case SMR_U32: ((uint32_t*)base)[idx] = (uint32_t)v; break;
            case SMR_I64: case SMR_U64: ((long long*)base)[idx] = v; break;
        }
        return;
    } else {
    auto re = re_(v);
    auto im = im_(v);
    switch (dt) {
        case SMR_F32: ((float*)base)[idx] = (float)re; break;
        case SMR_F64: ((double*)base)[idx] = (double)re; break;
        case SMR_C32: ((c32*)base)[idx] = c32{(float)re, (float)im}; break;
        case SMR_C64: ((c64*)base)[idx] = c64{(double)re, (double)im}; break;
        case SMR_I8: ((int8_t*)base)[idx] = (int8_t)llrint((double)re); break;
        case SMR_U8: ((uint8_t*)base)[idx] = (uint8_t)llrint((double)re); break;
        case SMR_I16: ((int16_t*)base)[idx] = (int16_t)llrint((double)re); break;
        case SMR_U16: ((uint16_t*)base)[idx] = (uint16_t)llrint((double)re); break;
        case SMR_I32: ((int32_t*)base)[idx] = (int32_t)llrint((double)re); break;
        case SMR_U32: ((uint32_t*)base)[idx] = (uint32_t)llrint((double)re); break;
        case SMR_I64: ((long long*)base)[idx] = (long long)llrint((double)re); break;
        case SMR_U64: {
            // llrint saturates at 2^63: values of the upper half are shifted into range first
            const double d = (double)re;
            ((unsigned long long*)base)[idx] = d >= 9223372036854775808.0 ? (unsigned long long)llrint(d - 9223372036854775808.0) + 9223372036854775808ull
                                                                          : (unsigned long long)llrint(d);
            break;
        }
    }
    }
}

// Operand table passed to every kernel by value.
struct OpTab {
    void* base[MAXM];   // base pointer with the element offset already applied
    int32_t dtype[MAXM];
    int32_t conj[MAXM];
};

template <class T, bool MIXED>
SMR_DEV T load_op(const OpTab& t, int k, i64 idx) {
    T v;
    if constexpr (MIXED)
        v = ld_as<T>(t.base[k], idx, t.dtype[k]);
    else
        v = ld<T>(t.base[k], idx);
    if constexpr (tr<T>::cx) {
        if (t.conj[k]) v = cj(v);
    }
    return v;
}
template <class T, bool MIXED>
SMR_DEV void store_op(const OpTab& t, i64 idx, T v) {
    if constexpr (tr<T>::cx) {
        if (t.conj[0]) v = cj(v);
    }
    if constexpr (MIXED)
        st_as<T>(t.base[0], idx, t.dtype[0], v);
    else
        st<T>(t.base[0], idx, v);
}

// ---- functors: f(a[0..NIN-1]) ---------------------------------------------------------------
// NIN < 0 means "runtime" (interpreter).  Constants come from Canon::fc.
template <class T> struct FIdent {
    static constexpr int NIN = 1;
    SMR_DEV T operator()(const T* a) const { return a[0]; }
};
template <class T> struct FAdd2 {
    static constexpr int NIN = 2;
    SMR_DEV T operator()(const T* a) const { return a[0] + a[1]; }
};
template <class T> struct FAdd3 {
    static constexpr int NIN = 3;
    SMR_DEV T operator()(const T* a) const { return (a[0] + a[1]) + a[2]; }
};
template <class T> struct FAdd4 {
    static constexpr int NIN = 4;
    SMR_DEV T operator()(const T* a) const { return ((a[0] + a[1]) + a[2]) + a[3]; }
};
template <class T> struct FScale {
    static constexpr int NIN = 1;
    T c;
    SMR_DEV T operator()(const T* a) const { return a[0] * c; }
};
template <class T> struct FSym {
    static constexpr int NIN = 2;
    T c;
    SMR_DEV T operator()(const T* a) const { return mathx<T>::bin(SMR_OP_DIV, a[0] + a[1], c); }
};
template <class T> struct FAxpy {
    static constexpr int NIN = 2;
    T c;
    SMR_DEV T operator()(const T* a) const { return c * a[0] + a[1]; }
};
template <class T> struct FAxpby {
    static constexpr int NIN = 2;
    T c, d;
    SMR_DEV T operator()(const T* a) const { return c * a[0] + d * a[1]; }
};
template <class T> struct FAbs2 {
    static constexpr int NIN = 1;
    SMR_DEV T operator()(const T* a) const { return mathx<T>::un(SMR_OP_ABS2, a[0]); }
};
template <class T> struct FMul2 {
    static constexpr int NIN = 2;
    SMR_DEV T operator()(const T* a) const { return a[0] * a[1]; }
};
template <class T> struct FExpr5 {  // a*exp(c*a) + sin(a*a), real types only
    static constexpr int NIN = 1;
    T c;
    SMR_DEV T operator()(const T* a) const {
        T x = a[0];
        if constexpr (is_int_class<T>::value) return x;  // never selected for the integer class
        else return x * exp(c * x) + sin(x * x);
    }
};

// Bytecode interpreter.  The program is wave-uniform, so every branch below is a scalar
// branch; the value stack lives in registers and is rotated on push/pop so that every
// register index is static (runtime-indexed arrays would be demoted to scratch memory).
template <class T> struct FProg {
    static constexpr int NIN = -1;
    ProgD p;
    SMR_DEV T operator()(const T* a) const {
        typedef typename tr<T>::real R;
        T s0 = mk<T>(R(0), R(0)), s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0, s6 = s0, s7 = s0;
        const int n = p.len;
        for (int pc = 0; pc < n; ++pc) {
            const int op = p.code[2 * pc], imm = p.code[2 * pc + 1];
            if (op <= SMR_OP_CONST) {
                T v;
                if (op == SMR_OP_ARG) {
                    v = a[0];
#pragma unroll
                    for (int k = 1; k < MAXIN; ++k)
                        if (imm == k + 1) v = a[k];
                } else {
                    v = mk<T>(rcast<R>(p.consts[2 * imm]), rcast<R>(p.consts[2 * imm + 1]));
                }
                s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
            } else if (op < 32) {
                s0 = mathx<T>::un(op, s0);
            } else if (op < 64) {
                s0 = mathx<T>::bin(op, s1, s0);
                s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
            } else {  // SELECT: a ? b : c with a=s2, b=s1, c=s0
                s0 = mathx<T>::truthy(s2) ? s1 : s0;
                s1 = s3; s2 = s4; s3 = s5; s4 = s6; s5 = s7;
            }
        }
        return s0;
    }
};

// reduction operator / initop (wave-uniform switches)
template <class T>
SMR_DEV T red_apply(int op, T a, T b) {
    if constexpr (tr<T>::arith) {
        switch (op) {
            case SMR_RED_ADD: return a + b;
            case SMR_RED_MUL: return a * b;
            case SMR_RED_MIN: return mathx<T>::bin(SMR_OP_MIN, a, b);
            case SMR_RED_MAX: return mathx<T>::bin(SMR_OP_MAX, a, b);
            case SMR_RED_AND: return mk<T>((mathx<T>::truthy(a) && mathx<T>::truthy(b)) ? typename tr<T>::real(1) : typename tr<T>::real(0), typename tr<T>::real(0));
            case SMR_RED_OR: return mk<T>((mathx<T>::truthy(a) || mathx<T>::truthy(b)) ? typename tr<T>::real(1) : typename tr<T>::real(0), typename tr<T>::real(0));
        }
    }
    return b;
}
template <class T>
SMR_DEV T init_apply(int op, T x, T beta) {
    typedef typename tr<T>::real R;
    switch (op) {
        case SMR_INIT_ZERO: return mk<T>(R(0), R(0));
        case SMR_INIT_SCALE: return x * beta;
        case SMR_INIT_CONST: return beta;
        case SMR_INIT_CONJ: return cj(x);
    }
    return x;
}

// wave64 shuffle of arbitrary POD (by 32-bit words)
template <class T>
SMR_DEV T shfl_xor_any(T v, int mask) {
    static_assert(sizeof(T) % 4 == 0, "word multiple");
    union {
        T t;
        int w[sizeof(T) / 4];
    } u;
    u.t = v;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) u.w[i] = __shfl_xor(u.w[i], mask, 64);
    return u.t;
}

}  // namespace smr
