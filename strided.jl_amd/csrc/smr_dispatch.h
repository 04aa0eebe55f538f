// smr_dispatch.h -- compile-time dispatch helpers shared by the kernel translation units.
//
// Every kernel family is a template over <compute type T, functor F, MIXED>.  Each family's
// .hip file is compiled once per compute type (-DSMR_CT=0..3 and 7 = the integer class) so the instantiation matrix
// builds in parallel; the per-type entry points are the explicit specialisations of
// launch_<family>_ct<CT>.
#pragma once

#ifndef SMR_JIT
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <vector>
#endif

#include "smr_device.h"

namespace smr {

template <int CT> struct ct_type;
template <> struct ct_type<SMR_F32> { typedef float type; };
template <> struct ct_type<SMR_F64> { typedef double type; };
template <> struct ct_type<SMR_C32> { typedef c32 type; };
template <> struct ct_type<SMR_C64> { typedef c64 type; };
template <> struct ct_type<SMR_I64> { typedef ix64 type; };  // the integer class (smr_device.h)

#ifndef SMR_JIT
template <class T> inline T hostmk(double re, double im);
template <> inline float hostmk<float>(double re, double) { return (float)re; }
template <> inline double hostmk<double>(double re, double) { return re; }
template <> inline c32 hostmk<c32>(double re, double im) { return c32{(float)re, (float)im}; }
template <> inline c64 hostmk<c64>(double re, double im) { return c64{re, im}; }
template <> inline ix64 hostmk<ix64>(double re, double) {
    return re >= 9223372036854775807.0 ? 9223372036854775807LL : (re <= -9223372036854775808.0 ? (-9223372036854775807LL - 1) : (ix64)re);
}

// Host-side stand-in for the functor that smr_jit.cpp generates from the f-program: selects the
// jit_launch() path inside the launchers (no device code is instantiated for it).
template <class T> struct FJitTag {
    static constexpr int NIN = -1;
};
template <class F> struct is_jit {
    static constexpr bool value = false;
};
template <class T> struct is_jit<FJitTag<T>> {
    static constexpr bool value = true;
};
template <class T> inline const char* tname();
template <> inline const char* tname<float>() { return "float"; }
template <> inline const char* tname<double>() { return "double"; }
template <> inline const char* tname<c32>() { return "smr::c32"; }
template <> inline const char* tname<c64>() { return "smr::c64"; }
template <> inline const char* tname<ix64>() { return "smr::ix64"; }
template <> inline const char* tname<b8>() { return "smr::b8"; }
template <> inline const char* tname<b16>() { return "smr::b16"; }

// f-programs without a native functor: runtime-compiled when possible, interpreted otherwise
template <class T, class Fn>
int with_prog(const Canon& c, Fn&& fn) {
    if (options().jit) {
        const int rc = fn(FJitTag<T>{});
        if (rc != SMR_JIT_UNAVAILABLE) return rc;
        if (jit_dry_run()) return set_error(SMR_EUNSUPPORTED, "runtime compilation is unavailable (hiprtc missing or the generated source failed to compile)");
    }
    return fn(FProg<T>{c.prog});
}

constexpr unsigned fbit(int k) { return 1u << k; }
constexpr unsigned FMASK_ALL = 0xffffffffu;

// Calls fn(functor) with the natively compiled functor for c.fkind when `mask` allows it,
// otherwise with the bytecode interpreter.
template <class T, class Fn>
int with_functor(const Canon& c, unsigned mask, Fn&& fn) {
    const int k = (mask & fbit(c.fkind)) ? c.fkind : FK_PROG;
    switch (k) {
        case FK_IDENT: return fn(FIdent<T>{});
        case FK_ADD2: return fn(FAdd2<T>{});
        case FK_ADD3: return fn(FAdd3<T>{});
        case FK_ADD4: return fn(FAdd4<T>{});
        case FK_SCALE: return fn(FScale<T>{hostmk<T>(c.fc[0], c.fc[1])});
        case FK_SYM:
            if constexpr (!is_int_class<T>::value) return fn(FSym<T>{hostmk<T>(c.fc[0], c.fc[1])});
            break;
        case FK_AXPY: return fn(FAxpy<T>{hostmk<T>(c.fc[0], c.fc[1])});
        case FK_AXPBY: return fn(FAxpby<T>{hostmk<T>(c.fc[0], c.fc[1]), hostmk<T>(c.fc[2], c.fc[3])});
        case FK_ABS2: return fn(FAbs2<T>{});
        case FK_MUL2: return fn(FMul2<T>{});
        case FK_EXPR5:
            if constexpr (!tr<T>::cx && !is_int_class<T>::value) return fn(FExpr5<T>{hostmk<T>(c.fc[0], 0)});
            break;
        default: break;
    }
    return with_prog<T>(c, fn);
}

// Per-type launch entry points (explicitly specialised in the -DSMR_CT objects).
template <int CT> int launch_generic_map_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_stream_map_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_tiled_map_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_reduce_all_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_reduce_part_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_orbit_map_ct(const Plan&, void* const*, hipStream_t);
template <int CT> int launch_flat_map_ct(const Plan&, void* const*, hipStream_t);

// Fills the kernel operand table: base pointers with the element offset folded in.
inline OpTab make_optab(const Canon& c, void* const* bases) {
    OpTab t;
    for (int k = 0; k < MAXM; ++k) {
        t.base[k] = nullptr;
        t.dtype[k] = 0;
        t.conj[k] = 0;
    }
    for (int k = 0; k < c.M; ++k) {
        char* b = (char*)(bases ? bases[c.orig[k]] : c.base[k]);
        t.base[k] = b + c.offsets[k] * (i64)c.esize[k];
        t.dtype[k] = c.dtype[k];
        t.conj[k] = c.conj[k];
    }
    return t;
}

// Every kernel launch of the library goes through SMR_LAUNCH.  take_launch_flags() hands out -- ONCE, to the first launch of
// the execution in progress -- the AQL ordering the overlap window decided on (smr_api.cpp: hipExtAnyOrderLaunch = the
// dispatch packet goes out without the barrier bit, so its waves may start while earlier, independent launches of the
// same stream are still draining); every later launch of the same execution (a folding pass) is ordered as usual.
// While a sequence records (smr_seq.cpp) nothing is launched: the launch is appended to the recorder with its arguments packed the
// way the kernarg segment holds them (every argument at its natural alignment, in order).
template <class T> inline void pack_arg(std::vector<unsigned char>& b, const T& v) {
    const size_t off = (b.size() + alignof(T) - 1) & ~(alignof(T) - 1);
    b.resize(off + sizeof(T));
    std::memcpy(b.data() + off, &v, sizeof(T));
}
template <class... A> inline void record_launch(std::vector<RecLaunch>* rec, const void* fn, unsigned grid, unsigned block, size_t lds, const A&... a) {
    RecLaunch r;
    r.hostfn = fn;
    r.grid = grid;
    r.block = block;
    r.lds = (unsigned)lds;
    (void)std::initializer_list<int>{(pack_arg(r.args, a), 0)...};
    take_slice_mark(r);
    rec->push_back(std::move(r));
}
#define SMR_LAUNCH(kern, grid, block, lds, s, ...)                                                                     \
    do {                                                                                                               \
        if (auto* smr_rec_ = ::smr::recorder()) {                                                                      \
            ::smr::record_launch(smr_rec_, (const void*)(kern), (grid).x, (block).x, (size_t)(lds), __VA_ARGS__);      \
            break;                                                                                                     \
        }                                                                                                              \
        ::smr::count_launch();                                                                                        \
        const unsigned smr_lf_ = ::smr::take_launch_flags();                                                           \
        if (smr_lf_) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)(lds), s, nullptr, nullptr, smr_lf_, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                               \
    } while (0)

// A failed earlier HIP call (e.g. an attribute query) must not be mistaken for a launch failure.
inline void clear_sticky_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, what);
    return SMR_OK;
}

#endif  // !SMR_JIT

}  // namespace smr
