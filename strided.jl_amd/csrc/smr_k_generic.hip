// smr_k_generic.hip -- family GENERIC: one thread per destination element, linear index
// decomposed over the canonical dims (destination-fastest), arbitrary signed / zero strides.
// This is the always-correct fallback of the map path; it is what the reference's
// _mapreduce_kernel! loop nest (src/mapreduce.jl:229-349) computes, without assuming anything
// about unit strides.  Compiled once per compute type (-DSMR_CT=n).
#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif

namespace smr {

struct GenArgs {
    OpTab ops;
    int32_t N, M;
    i64 total;
    i64 dims[MAXN];
    i64 strides[MAXM][MAXN];
};

template <class T, class F, bool MIXED>
SMR_DEV void generic_map_body(const GenArgs a, F f) {
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    const i64 step = (i64)gridDim.x * 256;
    const bool small = a.total <= 0x7fffffffLL;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < a.total; i += step) {
        i64 off[MAXM];
#pragma unroll
        for (int k = 0; k < MAXM; ++k) off[k] = 0;
        if (small) {
            uint32_t rem = (uint32_t)i;
#pragma unroll
            for (int d = 0; d < MAXN; ++d) {
                if (d < a.N) {
                    const uint32_t dd = (uint32_t)a.dims[d];
                    const uint32_t q = rem / dd;
                    const i64 c = (i64)(rem - q * dd);
                    rem = q;
#pragma unroll
                    for (int k = 0; k < MAXM; ++k)
                        if (k < a.M) off[k] += c * a.strides[k][d];
                }
            }
        } else {
            i64 rem = i;
#pragma unroll
            for (int d = 0; d < MAXN; ++d) {
                if (d < a.N) {
                    const i64 q = rem / a.dims[d];
                    const i64 c = rem - q * a.dims[d];
                    rem = q;
#pragma unroll
                    for (int k = 0; k < MAXM; ++k)
                        if (k < a.M) off[k] += c * a.strides[k][d];
                }
            }
        }
        T in[MAXIN];
#pragma unroll
        for (int k = 0; k < MAXIN; ++k) {
            in[k] = T{};
            if (k < nin) in[k] = load_op<T, MIXED>(a.ops, k + 1, off[k + 1]);
        }
        store_op<T, MIXED>(a.ops, off[0], f(in));
    }
}

#ifndef SMR_JIT
template <class T, class F, bool MIXED>
__global__ void __launch_bounds__(256) k_generic_map(GenArgs a, F f) {
    generic_map_body<T, F, MIXED>(a, f);
}

template <class T, class F, bool MIXED>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    GenArgs a;
    a.ops = make_optab(c, bases);
    a.N = c.N;
    a.M = c.M;
    a.total = c.total;
    for (int i = 0; i < MAXN; ++i) a.dims[i] = c.dims[i];
    for (int k = 0; k < MAXM; ++k)
        for (int i = 0; i < MAXN; ++i) a.strides[k][i] = (k < c.M) ? c.strides[k][i] : 0;
    i64 blocks = (c.total + 255) / 256;
    blocks = std::min<i64>(blocks, 256 * 32);
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "generic";
        l.tname = tname<T>();
        l.argtype = "smr::GenArgs";
        l.entry = std::string("smr::generic_map_body<") + tname<T>() + ", smr::FJit, " + (MIXED ? "true" : "false") + ">(a, smr::FJit{kc});";
        l.grid = (unsigned)blocks;
        l.block = 256;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        SMR_LAUNCH((k_generic_map<T, F, MIXED>), dim3((unsigned)blocks), dim3(256), 0, s, a, f);
        return check_launch("k_generic_map");
    }
}

template <>
int launch_generic_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.bitcopy) {
#if SMR_CT == SMR_F32
        // opaque element moves by size; only this object carries the narrow movers
        switch (c.esize[0]) {
            case 1: return go<b8, FIdent<b8>, false>(plan, bases, s, FIdent<b8>{});
            case 2: return go<b16, FIdent<b16>, false>(plan, bases, s, FIdent<b16>{});
            case 4: return go<float, FIdent<float>, false>(plan, bases, s, FIdent<float>{});
            case 8: return go<double, FIdent<double>, false>(plan, bases, s, FIdent<double>{});
            default: return go<c64, FIdent<c64>, false>(plan, bases, s, FIdent<c64>{});
        }
#else
        return set_error(SMR_EINVAL, "bitcopy is dispatched through the f32 object");
#endif
    }
    if (c.mixed) return with_prog<T>(c, [&](auto f) { return go<T, decltype(f), true>(plan, bases, s, f); });
    return with_functor<T>(c, fbit(FK_IDENT), [&](auto f) { return go<T, decltype(f), false>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
