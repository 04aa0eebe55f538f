// smr_k_reduce.hip -- the reduce path: dest[I] = op(initop(dest[I]), op_over_reduced_dims f(...))
//
// Semantics follow _mapreduce_kernel! with op !== nothing (reference src/mapreduce.jl:313-327)
// and the initop-once rule (:351-382, :403-409): `initop` is applied exactly once to every
// destination element, then the mapped values are accumulated into it.  On the GPU this is
//     dest = op(initop(dest_old), partial)     in the epilogue,
// where `partial` is a deterministic tree reduction (per-lane accumulators -> wave64
// __shfl_xor butterflies -> LDS across the workgroup's waves -> one partial per workgroup ->
// a fold of the partials).  No float atomics, so results are run-to-run identical;
// this mirrors the reference's per-task partial slots + serial fold (:153-170).
// The fold of a split reduction runs inside the SAME launch (Options::reduce_single): every workgroup publishes
// its partials, bumps the arrival counter of its output group, and the workgroup that arrives last folds the group's
// partials in slot order -- which workgroup that is varies from run to run, what it computes does not.  (The second
// launch this replaces cost more than the 19-48 MiB reductions it served: a kernel boundary is 1.2-2 us.)
//   REDUCE_ALL : complete reduction (destination is one element)
//   REDUCE_PART: some dims kept; TR lanes cooperate per destination element
#include "smr_dispatch.h"

#ifndef SMR_RED_U
#define SMR_RED_U 4  // loads in flight per lane in the ROW / COL forms
#endif
#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif

namespace smr {

struct RedArgs {
    OpTab ops;
    int32_t N, NK, M, redop, initop, linear, tr, trlog;
    i64 total, nout, nred;
    i64 dims[MAXN];
    i64 strides[MAXM][MAXN];
    double beta[2];
    void* partials;
    int32_t nparts;
    int32_t nsplit;   // REDUCE_PART: the reduced range is cut into nsplit chunks (one workgroup each)
    int32_t ngroups;  // REDUCE_PART: workgroups along the output index
    i64 chunk;        // reduced elements per chunk (multiple of tr)
    // vectorised forms (ROW / COL): reduced space = inner dim NK (extent L0) x outer index q in [0, Q)
    int32_t ctx, cty, cy0, cy1;  // COL: lanes along kept dim 0 x rows (cty = cy0 rows along the inner reduced dim x cy1 along the outer index); ctx * cty <= 256
    uint32_t ctx_m, cy0_m;       //      ceil(2^16 / ctx), ceil(2^16 / cy0): n / d = (n * m) >> 16 for n < 256, d <= 256
    int32_t g0log, g1log, txlog, xsplit, qsplit, ntl;  // ntl: non-temporal loads in REDUCE_ALL (Options::nt_load)
    i64 L0, Q, xchunk, qchunk;
    i64 nkb0;         // COL: workgroups along kept dim 0
    unsigned* counters;  // one-launch split reductions: arrival counter per output group (zero between launches)
    int32_t single;      // 1: the last workgroup to arrive folds the partials; 0: a second launch does
    // two-level arrival (round 4): a device-scope ticket costs ~12 ns and tickets on ONE counter serialise -- 512 chunks on one
    // counter are 6 us.  With nshard > 0 chunk sp arrives at counter [group * nshard + sp % nshard]; the last arrival of a shard
    // folds that shard's partials into ONE shard partial (partials2[o * nshard + shard]) and arrives at the group's second-level
    // counter [ngroups * nshard + group]; the last of the nshard shards folds the shard partials and writes the destination.
    // Fold order is fixed by the lane layout, not by who arrives when: results are reproducible run to run.
    int32_t nshard;
    int32_t cpad;        // counter stride in the two-level form: 32 (one counter per 128-byte line) whenever the counters fit that way
    void* partials2;
};

template <class T>
SMR_DEV T neutral(int op) {
    typedef typename tr<T>::real R;
    switch (op) {
        case SMR_RED_MUL: return mk<T>(R(1), R(0));
        case SMR_RED_AND: return mk<T>(R(1), R(0));
        case SMR_RED_MIN: return mk<T>(rcast<R>(__builtin_huge_val()), R(0));   // integer class: typemax / typemin
        case SMR_RED_MAX: return mk<T>(rcast<R>(-__builtin_huge_val()), R(0));
    }
    return mk<T>(R(0), R(0));
}

// reduce `v` over the `width` (power of two <= 64) consecutive lanes of a wave
template <class T>
SMR_DEV T wave_reduce(T v, int op, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) v = red_apply<T>(op, v, shfl_xor_any(v, m));
    return v;
}

template <class T, bool MIXED>
SMR_DEV void epilogue(const RedArgs& a, i64 off0, T acc) {
    typedef typename tr<T>::real R;
    const T beta = mk<T>(rcast<R>(a.beta[0]), rcast<R>(a.beta[1]));
    T old;
    if (a.initop == SMR_INIT_ZERO || a.initop == SMR_INIT_CONST) {
        old = init_apply<T>(a.initop, T{}, beta);  // does not depend on what the destination held: no load
    } else {
        old = load_op<T, MIXED>(a.ops, 0, off0);
        if (a.initop != SMR_INIT_NONE) old = init_apply<T>(a.initop, old, beta);
    }
    store_op<T, MIXED>(a.ops, off0, red_apply<T>(a.redop, old, acc));
}

// offsets of box element `i` (decomposed over dims [d0, d1)) added into off[]
// (a real loop over the dims, not an unrolled one: unrolled, every call site carried eight copies of the 64-bit
// division -- the ROW kernel was 15,800 ISA lines, more than the instruction cache holds)
SMR_DEV void decompose(const RedArgs& a, i64 i, int d0, int d1, i64* off) {
    i64 rem = i;
#pragma nounroll
    for (int d = d0; d < d1; ++d) {
        i64 c;
        if (d == d1 - 1) {
            c = rem;
        } else {
            const i64 q = rem / a.dims[d];
            c = rem - q * a.dims[d];
            rem = q;
        }
#pragma unroll
        for (int k = 0; k < MAXM; ++k)
            if (k < a.M) off[k] += c * a.strides[k][d];
    }
}

// One-launch split reductions.  The L2s of the eight XCDs are not coherent with each other and a CU's L1 is never
// refreshed by another CU's stores, so partials that are folded inside the launch travel write-through: agent-scope
// relaxed atomic stores (global_store ... sc1, at most 8 bytes each) and agent-scope loads in the folding workgroup, ordered
// by "every wave waits for its stores' acknowledgements -> barrier -> one relaxed agent-scope ticket".  No release /
// acquire fences: buffer_wbl2 / buffer_inv cost 1.7 us each on this part, as much as the launch they would save.
// The ordering above is a property of the gfx9 memory pipeline (vmcnt counts stores as well as loads, sc1 stores write through the
// per-XCD L2): on targets that count stores separately (vscnt, gfx10+) the last workgroup could read stale partials.  This file is
// built for gfx950 only (csrc/Makefile); a port must turn reduce_single off or replace the protocol by release / acquire.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "smr_k_reduce.hip: the one-launch fold relies on gfx942 / gfx950 store ordering (s_waitcnt vmcnt covers stores, sc1 write-through)"
#endif
template <class T>
SMR_DEV void put_partial(const RedArgs& a, i64 idx, T v) {
    T* p = (T*)a.partials + idx;
    if (!a.single) {
        *p = v;
        return;
    }
    if constexpr (sizeof(T) == 4) {
        unsigned u;
        __builtin_memcpy(&u, &v, 4);
        __hip_atomic_store((unsigned*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        unsigned long long u[sizeof(T) / 8];
        __builtin_memcpy(u, &v, sizeof(T));
#pragma unroll
        for (unsigned j = 0; j < sizeof(T) / 8; ++j) __hip_atomic_store((unsigned long long*)p + j, u[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <class T>
SMR_DEV void put_shard_partial(const RedArgs& a, i64 idx, T v) {  // always write-through: only the one-launch form has shards
    RedArgs b = a;
    b.partials = a.partials2;
    put_partial<T>(b, idx, v);
}
template <class T>
SMR_DEV T get_partial_from(const RedArgs& a, const void* base, i64 idx) {
    const T* p = (const T*)base + idx;
    if (!a.single) return *p;
    T v;
    if constexpr (sizeof(T) == 4) {
        const unsigned u = __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_memcpy(&v, &u, 4);
    } else {
        unsigned long long u[sizeof(T) / 8];
#pragma unroll
        for (unsigned j = 0; j < sizeof(T) / 8; ++j) u[j] = __hip_atomic_load((const unsigned long long*)p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_memcpy(&v, u, sizeof(T));
    }
    return v;
}
template <class T>
SMR_DEV T get_partial(const RedArgs& a, i64 idx) { return get_partial_from<T>(a, a.partials, idx); }
// Called by every thread of a workgroup after its partials were stored; true in the workgroup that arrived last of the
// `nparts` sharing counter `group` (it may then read all their partials).  The last one also zeroes the counter for the
// next launch of this plan.
SMR_DEV bool arrive_last(const RedArgs& a, i64 group, int nparts) {
    __shared__ int s_last;
    __syncthreads();  // s_last may still be read by a previous arrival of this workgroup (two-level form)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.counters + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)(nparts - 1);
        if (last) __hip_atomic_store(a.counters + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// ---- complete reduction -------------------------------------------------------------------------
template <class T, int V>
struct alignas(sizeof(T) * V) RVec {
    T v[V];
};

// fold of the per-workgroup partials of a complete reduction (one workgroup, slot order fixed by the thread layout)
template <class T, bool MIXED>
SMR_DEV void fold_all(const RedArgs& a, int redop, T* wsum) {
    T v = neutral<T>(redop);
    for (int i = threadIdx.x; i < a.nparts; i += 256) v = red_apply<T>(redop, v, get_partial<T>(a, i));
    v = wave_reduce(v, redop, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = red_apply<T>(redop, red_apply<T>(redop, wsum[0], wsum[1]), red_apply<T>(redop, wsum[2], wsum[3]));
        epilogue<T, MIXED>(a, 0, v);
    }
}

template <class T, class F, bool MIXED, int V, int OPC>
SMR_DEV void reduce_all_impl(const RedArgs& a, F f) {
    const int redop = (OPC >= 0) ? OPC : a.redop;  // compile-time for the sum: no op switch inside the loops
    __shared__ T wsum[4];
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    constexpr int ACC = 4;
    T acc[ACC];
#pragma unroll
    for (int j = 0; j < ACC; ++j) acc[j] = neutral<T>(redop);
    const i64 nthreads = (i64)gridDim.x * 256;
    const i64 t0 = (i64)blockIdx.x * 256 + threadIdx.x;
    if constexpr (V > 1) {
        // linear, unit-stride (or broadcast) inputs: 16-byte loads, ACC vectors in flight
        typedef RVec<T, V> VT;
        const i64 nvec = a.total / V;
        // a wave reads ACC consecutive 1-KiB rows (one contiguous 4-KiB piece) per iteration
        // (measured on 4 GiB: 6.3 TB/s, against 6.1 TB/s for grid-strided single rows)
        const i64 lane_ = threadIdx.x & 63;
        const i64 nrows = (nvec + 63) >> 6;
        // one run-time branch around the loop selects plain or non-temporal loads (a diamond per load costs 10 %: the loads of a batch no longer issue together)
        auto sweep = [&](auto NTL) {
            for (i64 row = (t0 >> 6) * ACC; row < nrows; row += (nthreads >> 6) * ACC) {
                const i64 i = row * 64 + lane_;
                VT x[ACC][MAXIN];
#pragma unroll
                for (int j = 0; j < ACC; ++j) {
                    const i64 ii = i + j * 64;
                    if (ii < nvec) {
#pragma unroll
                        for (int k = 0; k < MAXIN; ++k)
                            if (k < nin) {
                                if (a.strides[k + 1][0] == 0) {
                                    const T s = load_op<T, false>(a.ops, k + 1, 0);
#pragma unroll
                                    for (int e = 0; e < V; ++e) x[j][k].v[e] = s;
                                } else {
                                    x[j][k] = load_vec_ct<decltype(NTL)::value, VT>((const T*)a.ops.base[k + 1] + ii * V);
                                    if constexpr (tr<T>::cx) {
                                        if (a.ops.conj[k + 1]) {
#pragma unroll
                                            for (int e = 0; e < V; ++e) x[j][k].v[e] = cj(x[j][k].v[e]);
                                        }
                                    }
                                }
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < ACC; ++j) {
                    const i64 ii = i + j * 64;
                    if (ii < nvec) {
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            T in[MAXIN];
#pragma unroll
                            for (int k = 0; k < MAXIN; ++k) {
                                in[k] = T{};
                                if (k < nin) in[k] = x[j][k].v[e];
                            }
                            acc[j] = red_apply<T>(redop, acc[j], f(in));
                        }
                    }
                }
            }
        };
        if (a.ntl) {
            nt_block_guard();
            sweep(BoolC<true>{});
            nt_block_guard();
        } else {
            sweep(BoolC<false>{});
        }
    } else {
        for (i64 i = t0; i < a.total; i += nthreads * ACC) {
#pragma unroll
            for (int j = 0; j < ACC; ++j) {
                const i64 ii = i + j * nthreads;
                if (ii < a.total) {
                    i64 off[MAXM];
#pragma unroll
                    for (int k = 0; k < MAXM; ++k) off[k] = 0;
                    if (a.linear) {
#pragma unroll
                        for (int k = 1; k < MAXM; ++k)
                            if (k < a.M) off[k] = ii * a.strides[k][0];
                    } else {
                        decompose(a, ii, 0, a.N, off);
                    }
                    T in[MAXIN];
#pragma unroll
                    for (int k = 0; k < MAXIN; ++k) {
                        in[k] = T{};
                        if (k < nin) in[k] = load_op<T, MIXED>(a.ops, k + 1, off[k + 1]);
                    }
                    acc[j] = red_apply<T>(redop, acc[j], f(in));
                }
            }
        }
    }
#pragma unroll
    for (int w = ACC / 2; w > 0; w >>= 1) {  // pairwise tree over the per-lane accumulators
#pragma unroll
        for (int j = 0; j < w; ++j) acc[j] = red_apply<T>(redop, acc[j], acc[j + w]);
    }
    T v = acc[0];
    v = wave_reduce(v, redop, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = red_apply<T>(redop, red_apply<T>(redop, wsum[0], wsum[1]), red_apply<T>(redop, wsum[2], wsum[3]));
        if (gridDim.x == 1)
            epilogue<T, MIXED>(a, 0, v);
        else
            put_partial<T>(a, blockIdx.x, v);
    }
    if (gridDim.x > 1 && a.single) {
        if (!arrive_last(a, 0, (int)gridDim.x)) return;
        fold_all<T, MIXED>(a, redop, wsum);
    }
}
template <class T, class F, bool MIXED, int V>
SMR_DEV void reduce_all_body(const RedArgs a, F f) {
    if (a.redop == SMR_RED_ADD)
        reduce_all_impl<T, F, MIXED, V, SMR_RED_ADD>(a, f);
    else
        reduce_all_impl<T, F, MIXED, V, -1>(a, f);
}

template <class T, bool MIXED>
__global__ void __launch_bounds__(256) k_reduce_final(RedArgs a) {
    __shared__ T wsum[4];
    fold_all<T, MIXED>(a, a.redop, wsum);
}

// fold of the nsplit partials of outputs [obase, obase + ocount) by one workgroup: 2^lpolog consecutive lanes per output
// walk its partials (slot order fixed by the lane layout), wave butterfly, epilogue.  Used by the workgroup that arrived
// last (one-launch form) and, one output group per workgroup, by nothing else: the two-launch form has its own kernel.
// MODE 0: all nsplit partials of every output -> destination.  MODE 1: the partials of shard `sh` (chunks sh, sh + nshard, ...) ->
// the output's shard partial.  MODE 2: the nshard shard partials -> destination.
template <class T, bool MIXED, int MODE = 0>
SMR_DEV void fold_part(const RedArgs& a, int redop, i64 obase, int ocount, int sh = 0) {
    const int n = MODE == 0 ? a.nsplit : (MODE == 1 ? (a.nsplit - sh + a.nshard - 1) / a.nshard : a.nshard);
    int lpolog = 0;
    while (lpolog < 6 && (ocount << (lpolog + 1)) <= 256 && (2 << lpolog) <= n) ++lpolog;
    const int lpo = 1 << lpolog;
    const int l = threadIdx.x & (lpo - 1);
    const int per = 256 >> lpolog;
    for (int o0 = 0; o0 < ocount; o0 += per) {  // uniform trip count: the butterfly below needs whole lane groups
        const int oo = o0 + (threadIdx.x >> lpolog);
        const bool live = oo < ocount;
        const i64 o = obase + oo;
        T acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = neutral<T>(redop);
        if (live) {
            for (int i = l; i < n; i += 4 * lpo) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ii = i + j * lpo;
                    if (ii < n) {
                        T x;
                        if constexpr (MODE == 0) x = get_partial<T>(a, o * a.nsplit + ii);
                        else if constexpr (MODE == 1) x = get_partial<T>(a, o * a.nsplit + sh + (i64)ii * a.nshard);
                        else x = get_partial_from<T>(a, a.partials2, o * a.nshard + ii);
                        acc[j] = red_apply<T>(redop, acc[j], x);
                    }
                }
            }
        }
        T v = red_apply<T>(redop, red_apply<T>(redop, acc[0], acc[1]), red_apply<T>(redop, acc[2], acc[3]));
        v = wave_reduce(v, redop, lpo);
        if (live && l == 0) {
            if constexpr (MODE == 1) {
                put_shard_partial<T>(a, o * a.nshard + sh, v);
            } else {
                i64 ooff[MAXM];
#pragma unroll
                for (int k = 0; k < MAXM; ++k) ooff[k] = 0;
                decompose(a, o, 0, a.NK, ooff);
                epilogue<T, MIXED>(a, ooff[0], v);
            }
        }
    }
}
// The in-launch fold of output group `group` (outputs [obase, obase + ocount)) after this workgroup -- chunk `sp` -- has stored its
// partials: one level of tickets, or two (RedArgs::nshard).
template <class T, bool MIXED>
SMR_DEV void fold_in_launch(const RedArgs& a, int redop, i64 group, i64 sp, i64 obase, int ocount) {
    if (a.nshard <= 0) {
        if (!arrive_last(a, group, a.nsplit)) return;
        fold_part<T, MIXED, 0>(a, redop, obase, ocount);
        return;
    }
    const int sh = (int)(sp % a.nshard);
    const int members = (a.nsplit - sh + a.nshard - 1) / a.nshard;
    if (!arrive_last(a, (group * a.nshard + sh) * a.cpad, members)) return;
    fold_part<T, MIXED, 1>(a, redop, obase, ocount, sh);
    if (!arrive_last(a, ((i64)a.ngroups * a.nshard + group) * a.cpad, a.nshard)) return;
    fold_part<T, MIXED, 2>(a, redop, obase, ocount);
}

// ---- partial reduction ----------------------------------------------------------------------------
// 256 threads = (256/TR) destination elements x TR lanes.  Lanes of one destination element
// are consecutive threads, so when the inputs' unit-stride axis is a reduced dim the loads
// coalesce; with TR == 1 consecutive threads own consecutive destination elements instead.
template <class T, class F, bool MIXED, int OPC>
SMR_DEV void reduce_part_impl(const RedArgs& a, F f) {
    const int redop = (OPC >= 0) ? OPC : a.redop;  // compile-time for the sum: no op switch inside the loops
    __shared__ T xbuf[256];
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    const int tr_ = a.tr;
    const int ob = 256 >> a.trlog;
    const int rl = threadIdx.x & (tr_ - 1);
    const int ol = threadIdx.x >> a.trlog;
    const i64 sp = (i64)blockIdx.x / a.ngroups;  // which chunk of the reduced range
    const i64 og = (i64)blockIdx.x - sp * a.ngroups;
    const i64 o = og * ob + ol;
    const bool live = o < a.nout;
    const i64 rbeg = sp * a.chunk;
    const i64 rend = (rbeg + a.chunk < a.nred) ? rbeg + a.chunk : a.nred;
    i64 ooff[MAXM];
#pragma unroll
    for (int k = 0; k < MAXM; ++k) ooff[k] = 0;
    if (live) decompose(a, o, 0, a.NK, ooff);
    T acc = neutral<T>(redop);
    if (live) {
        for (i64 r = rbeg + rl; r < rend; r += tr_) {
            i64 off[MAXM];
#pragma unroll
            for (int k = 0; k < MAXM; ++k) off[k] = ooff[k];
            decompose(a, r, a.NK, a.N, off);
            T in[MAXIN];
#pragma unroll
            for (int k = 0; k < MAXIN; ++k) {
                in[k] = T{};
                if (k < nin) in[k] = load_op<T, MIXED>(a.ops, k + 1, off[k + 1]);
            }
            acc = red_apply<T>(redop, acc, f(in));
        }
    }
    if (tr_ > 1) {
        const int w = tr_ < 64 ? tr_ : 64;
        acc = wave_reduce(acc, redop, w);
        if (tr_ > 64) {  // lanes of one output span several waves
            xbuf[threadIdx.x] = acc;
            __syncthreads();
            if (rl == 0) {
                for (int j = 64; j < tr_; j += 64) acc = red_apply<T>(redop, acc, xbuf[threadIdx.x + j]);
            }
        }
    }
    if (live && rl == 0) {
        if (a.nsplit == 1)
            epilogue<T, MIXED>(a, ooff[0], acc);
        else
            put_partial<T>(a, o * a.nsplit + sp, acc);
    }
    if (a.nsplit > 1 && a.single) {
        const i64 left = a.nout - og * ob;
        fold_in_launch<T, MIXED>(a, redop, og, sp, og * ob, (int)(left < ob ? left : ob));
    }
}
template <class T, class F, bool MIXED>
SMR_DEV void reduce_part_body(const RedArgs a, F f) {
    if (a.redop == SMR_RED_ADD)
        reduce_part_impl<T, F, MIXED, SMR_RED_ADD>(a, f);
    else
        reduce_part_impl<T, F, MIXED, -1>(a, f);
}


// ---- partial reduction, vectorised forms -----------------------------------------------------------------
// Both walk the reduced space as (inner dim NK) x (outer index q over dims NK+1..N-1): the outer
// offsets are decomposed once per q, the inner dim advances by plain stride additions -- no
// per-element index division (the general form above pays one 64-bit decompose per element).
//
// ROW: every input is unit-stride (or broadcast) along the inner reduced dim.  G = G0 x G1 consecutive
// lanes cooperate on one destination element: G0 lanes walk the inner dim with V-element vector
// loads, G1 lanes take different q.  sum(A; dims=1) of a column-major matrix is the model case.
template <class T, class F, bool MIXED, int V, int OPC>
SMR_DEV void reduce_row_impl(const RedArgs& a, F f) {
    const int redop = (OPC >= 0) ? OPC : a.redop;  // compile-time for the sum: no op switch inside the loops
    __shared__ T xbuf[256];
    typedef RVec<T, V> VT;
    constexpr int U = SMR_RED_U;  // vectors in flight per lane
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    const int glog = a.g0log + a.g1log;
    const int G0 = 1 << a.g0log, G1 = 1 << a.g1log;
    const int gl = threadIdx.x & ((1 << glog) - 1);
    const int ol = threadIdx.x >> glog;
    const int l0 = gl & (G0 - 1), l1 = gl >> a.g0log;
    const i64 sp = (i64)blockIdx.x / a.ngroups;
    const i64 og = (i64)blockIdx.x - sp * a.ngroups;
    const i64 o = og * (256 >> glog) + ol;
    const bool live = o < a.nout;
    const i64 sx = sp % a.xsplit, sq = sp / a.xsplit;
    const i64 xbeg = sx * a.xchunk, xend = (xbeg + a.xchunk < a.L0) ? xbeg + a.xchunk : a.L0;
    const i64 qbeg = sq * a.qchunk, qend = (qbeg + a.qchunk < a.Q) ? qbeg + a.qchunk : a.Q;
    i64 ooff[MAXM];
#pragma unroll
    for (int k = 0; k < MAXM; ++k) ooff[k] = 0;
    if (live) decompose(a, o, 0, a.NK, ooff);
    T acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = neutral<T>(redop);
    // U loads in flight per lane: along the inner dim when it is long enough, else over q
    const bool unroll_q = (xend - xbeg) <= (i64)G0 * V;
    auto row_step = [&](const i64 (&off)[U][MAXM], const i64 (&xs)[U], const bool (&ok)[U]) {
        VT in[U][MAXIN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
#pragma unroll
                for (int k = 0; k < MAXIN; ++k)
                    if (k < nin) {
                        if (a.strides[k + 1][a.NK] == 0) {
                            const T sv = load_op<T, MIXED>(a.ops, k + 1, off[u][k + 1]);
#pragma unroll
                            for (int e = 0; e < V; ++e) in[u][k].v[e] = sv;
                        } else if constexpr (MIXED || V == 1) {
                            in[u][k].v[0] = load_op<T, MIXED>(a.ops, k + 1, off[u][k + 1] + xs[u]);
                        } else {
                            in[u][k] = *reinterpret_cast<const VT*>((const T*)a.ops.base[k + 1] + off[u][k + 1] + xs[u]);
                            if constexpr (tr<T>::cx) {
                                if (a.ops.conj[k + 1]) {
#pragma unroll
                                    for (int e = 0; e < V; ++e) in[u][k].v[e] = cj(in[u][k].v[e]);
                                }
                            }
                        }
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    T arg[MAXIN];
#pragma unroll
                    for (int k = 0; k < MAXIN; ++k) {
                        arg[k] = T{};
                        if (k < nin) arg[k] = in[u][k].v[e];
                    }
                    acc[u] = red_apply<T>(redop, acc[u], f(arg));
                }
            }
        }
    };
    if (live && !unroll_q) {
        for (i64 q = qbeg + l1; q < qend; q += G1) {
            i64 off[U][MAXM];
#pragma unroll
            for (int k = 0; k < MAXM; ++k) off[0][k] = ooff[k];
            if (a.N > a.NK + 1) decompose(a, q, a.NK + 1, a.N, off[0]);
#pragma unroll
            for (int u = 1; u < U; ++u)
#pragma unroll
                for (int k = 0; k < MAXM; ++k) off[u][k] = off[0][k];
            for (i64 x0 = xbeg + (i64)l0 * V; x0 < xend; x0 += (i64)G0 * V * U) {
                i64 xs[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xs[u] = x0 + (i64)u * G0 * V;
                    ok[u] = xs[u] < xend;
                }
                row_step(off, xs, ok);
            }
        }
    } else if (live) {
        const i64 x = xbeg + (i64)l0 * V;
        if (x < xend) {
            for (i64 q0 = qbeg + l1; q0 < qend; q0 += (i64)G1 * U) {
                i64 off[U][MAXM], xs[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const i64 q = q0 + (i64)u * G1;
                    ok[u] = q < qend;
                    xs[u] = x;
#pragma unroll
                    for (int k = 0; k < MAXM; ++k) off[u][k] = ooff[k];
                    if (ok[u] && a.N > a.NK + 1) decompose(a, q, a.NK + 1, a.N, off[u]);
                }
                row_step(off, xs, ok);
            }
        }
    }
    // pairwise over the U accumulators (U = 4: (a0 + a1) + (a2 + a3))
#pragma unroll
    for (int w = 1; w < U; w <<= 1)
#pragma unroll
        for (int u = 0; u + w < U; u += 2 * w) acc[u] = red_apply<T>(redop, acc[u], acc[u + w]);
    T v = acc[0];
    const int G = 1 << glog;
    v = wave_reduce(v, redop, G < 64 ? G : 64);
    if (G > 64) {  // lanes of one output span several waves
        xbuf[threadIdx.x] = v;
        __syncthreads();
        if (gl == 0)
            for (int j = 64; j < G; j += 64) v = red_apply<T>(redop, v, xbuf[threadIdx.x + j]);
    }
    if (live && gl == 0) {
        if (a.nsplit == 1)
            epilogue<T, MIXED>(a, ooff[0], v);
        else
            put_partial<T>(a, o * a.nsplit + sp, v);
    }
    if (a.nsplit > 1 && a.single) {
        const i64 ob = 256 >> glog, left = a.nout - og * ob;
        fold_in_launch<T, MIXED>(a, redop, og, sp, og * ob, (int)(left < ob ? left : ob));
    }
}
template <class T, class F, bool MIXED, int V>
SMR_DEV void reduce_row_body(const RedArgs a, F f) {
    if (a.redop == SMR_RED_ADD)
        reduce_row_impl<T, F, MIXED, V, SMR_RED_ADD>(a, f);
    else
        reduce_row_impl<T, F, MIXED, V, -1>(a, f);
}

// COL: every input is unit-stride (or broadcast) along kept dim 0.  A workgroup = TX lanes along
// dim 0 (V destination elements each, vector loads) x TY rows of the reduced space (Y0 rows along
// the inner reduced dim x Y1 along q); the rows are folded through LDS.  sum(A; dims=2) of a
// column-major matrix is the model case.
template <class T, class F, bool MIXED, int V, int OPC>
SMR_DEV void reduce_col_impl(const RedArgs& a, F f) {
    const int redop = (OPC >= 0) ? OPC : a.redop;  // compile-time for the sum: no op switch inside the loops
    __shared__ T xbuf[256 * V];
    typedef RVec<T, V> VT;
    constexpr int U = SMR_RED_U;
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    // lanes along kept dim 0 x rows: powers of two, or sized to the row (25 x 10 for rows of 100 Float32; lanes past TX * TY idle)
    const int TX = a.ctx, TY = a.cty;
    const int ty = (int)((threadIdx.x * a.ctx_m) >> 16), tx = (int)threadIdx.x - ty * TX;
    const int Y0 = a.cy0, Y1 = a.cy1;
    const int t1 = (int)(((uint32_t)ty * a.cy0_m) >> 16), t0 = ty - t1 * Y0;
    const i64 sp = (i64)blockIdx.x / a.ngroups;
    const i64 kb = (i64)blockIdx.x - sp * a.ngroups;
    const i64 krest = kb / a.nkb0;
    const i64 c0 = kb - krest * a.nkb0;
    const i64 i0 = (c0 * TX + tx) * V;
    const bool live = ty < TY && i0 < a.dims[0];
    const i64 sx = sp % a.xsplit, sq = sp / a.xsplit;
    const i64 jbeg = sx * a.xchunk, jend = (jbeg + a.xchunk < a.L0) ? jbeg + a.xchunk : a.L0;
    const i64 qbeg = sq * a.qchunk, qend = (qbeg + a.qchunk < a.Q) ? qbeg + a.qchunk : a.Q;
    i64 ooff[MAXM];
#pragma unroll
    for (int k = 0; k < MAXM; ++k) ooff[k] = (k < a.M) ? i0 * a.strides[k][0] : 0;
    if (a.NK > 1) decompose(a, krest, 1, a.NK, ooff);
    T acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = neutral<T>(redop);
    // U loads in flight per lane: along the inner reduced dim when it is long enough, else over q
    const bool unroll_q = (jend - jbeg) <= (i64)Y0;
    auto col_step = [&](const i64 (&off)[U][MAXM], const bool (&ok)[U]) {
        VT in[U][MAXIN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
#pragma unroll
                for (int k = 0; k < MAXIN; ++k)
                    if (k < nin) {
                        const i64 oj = off[u][k + 1];
                        if (a.strides[k + 1][0] == 0) {
                            const T sv = load_op<T, MIXED>(a.ops, k + 1, oj);
#pragma unroll
                            for (int e = 0; e < V; ++e) in[u][k].v[e] = sv;
                        } else if constexpr (MIXED || V == 1) {
                            in[u][k].v[0] = load_op<T, MIXED>(a.ops, k + 1, oj);
                        } else {
                            in[u][k] = *reinterpret_cast<const VT*>((const T*)a.ops.base[k + 1] + oj);
                            if constexpr (tr<T>::cx) {
                                if (a.ops.conj[k + 1]) {
#pragma unroll
                                    for (int e = 0; e < V; ++e) in[u][k].v[e] = cj(in[u][k].v[e]);
                                }
                            }
                        }
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    T arg[MAXIN];
#pragma unroll
                    for (int k = 0; k < MAXIN; ++k) {
                        arg[k] = T{};
                        if (k < nin) arg[k] = in[u][k].v[e];
                    }
                    acc[e] = red_apply<T>(redop, acc[e], f(arg));
                }
            }
        }
    };
    if (live && !unroll_q) {
        for (i64 q = qbeg + t1; q < qend; q += Y1) {
            i64 offq[MAXM];
#pragma unroll
            for (int k = 0; k < MAXM; ++k) offq[k] = ooff[k];
            if (a.N > a.NK + 1) decompose(a, q, a.NK + 1, a.N, offq);
            for (i64 j0 = jbeg + t0; j0 < jend; j0 += (i64)Y0 * U) {
                i64 off[U][MAXM];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const i64 j = j0 + (i64)u * Y0;
                    ok[u] = j < jend;
#pragma unroll
                    for (int k = 0; k < MAXM; ++k) off[u][k] = (k < a.M) ? offq[k] + j * a.strides[k][a.NK] : 0;
                }
                col_step(off, ok);
            }
        }
    } else if (live) {
        const i64 j = jbeg + t0;
        if (j < jend) {
            for (i64 q0 = qbeg + t1; q0 < qend; q0 += (i64)Y1 * U) {
                i64 off[U][MAXM];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const i64 q = q0 + (i64)u * Y1;
                    ok[u] = q < qend;
#pragma unroll
                    for (int k = 0; k < MAXM; ++k) off[u][k] = (k < a.M) ? ooff[k] + j * a.strides[k][a.NK] : 0;
                    if (ok[u] && a.N > a.NK + 1) decompose(a, q, a.NK + 1, a.N, off[u]);
                }
                col_step(off, ok);
            }
        }
    }
    // fold the TY rows: LDS tree, halving the number of active rows
    // (any row count: a step folds rows [h, cur) onto [0, cur - h); fixed order, whatever the count)
#pragma unroll
    for (int e = 0; e < V; ++e) xbuf[threadIdx.x * V + e] = acc[e];
    __syncthreads();
    int hh = 1;
    while (hh < TY) hh <<= 1;
    int cur = TY;
    for (int h = hh >> 1; h > 0; h >>= 1) {
        if (ty < h && ty + h < cur) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                acc[e] = red_apply<T>(redop, acc[e], xbuf[(threadIdx.x + h * TX) * V + e]);
                xbuf[threadIdx.x * V + e] = acc[e];
            }
        }
        cur = cur < h ? cur : h;
        __syncthreads();
    }
    if (live && ty == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (a.nsplit == 1) {
                epilogue<T, MIXED>(a, ooff[0] + e * a.strides[0][0], acc[e]);
            } else {
                const i64 o = i0 + e + krest * a.dims[0];
                put_partial<T>(a, o * a.nsplit + sp, acc[e]);
            }
        }
    }
    if (a.nsplit > 1 && a.single) {
        const i64 per = (i64)TX * V, ibeg = c0 * per, left = a.dims[0] - ibeg;
        fold_in_launch<T, MIXED>(a, redop, kb, sp, krest * a.dims[0] + ibeg, (int)(left < per ? left : per));
    }
}
template <class T, class F, bool MIXED, int V>
SMR_DEV void reduce_col_body(const RedArgs a, F f) {
    if (a.redop == SMR_RED_ADD)
        reduce_col_impl<T, F, MIXED, V, SMR_RED_ADD>(a, f);
    else
        reduce_col_impl<T, F, MIXED, V, -1>(a, f);
}

// second pass of a split partial reduction: LPO = 2^lpolog consecutive lanes fold the nsplit partials
// of one output (strided walk + wave butterfly); LPO = 1 when there are many outputs and few partials
template <class T, bool MIXED>
__global__ void __launch_bounds__(256) k_reduce_part_final(RedArgs a) {
    const int lpolog = a.trlog;  // reused: log2 lanes per output of THIS pass (<= 6)
    const int lpo = 1 << lpolog;
    const i64 o = ((i64)blockIdx.x * 256 + threadIdx.x) >> lpolog;
    const int l = threadIdx.x & (lpo - 1);
    const bool live = o < a.nout;
    T acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = neutral<T>(a.redop);
    if (live) {
        // eight partials per lane and round, loaded before the first is used (guarded loads one by one were eight serial round trips to
        // the memory side -- the partials were written by the launch before -- and made this pass as long as the main one: sum over
        // dims (2,3,4) of (100,90,80,7) Float32, 512 chunks: 5.4 us of the 10.8); partial i still goes to accumulator (i / lpo) % 4
        const T* p = (const T*)a.partials + o * a.nsplit;
        const int last = a.nsplit - 1;
        auto walk = [&](auto nb) {
            constexpr int NB = decltype(nb)::value;
            for (int i = l; i < a.nsplit; i += NB * lpo) {
                T x[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int ii = i + j * lpo;
                    x[j] = p[ii < last ? ii : last];
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int ii = i + j * lpo;
                    const T t = red_apply<T>(a.redop, acc[j & 3], x[j]);
                    if (ii < a.nsplit) acc[j & 3] = t;
                }
            }
        };
        if (a.nsplit > 4 * lpo) {
            walk(IntC<8>{});
        } else {  // a handful of partials (one round of at most four per lane): guarded loads, as before round 6 -- measured 0.3 us ahead
            for (int i = l; i < a.nsplit; i += 4 * lpo) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j * lpo < a.nsplit) acc[j] = red_apply<T>(a.redop, acc[j], p[i + j * lpo]);
            }
        }
    }
    T v = red_apply<T>(a.redop, red_apply<T>(a.redop, acc[0], acc[1]), red_apply<T>(a.redop, acc[2], acc[3]));
    v = wave_reduce(v, a.redop, lpo);
    if (live && l == 0) {
        i64 ooff[MAXM];
#pragma unroll
        for (int k = 0; k < MAXM; ++k) ooff[k] = 0;
        decompose(a, o, 0, a.NK, ooff);
        epilogue<T, MIXED>(a, ooff[0], v);
    }
}

#ifndef SMR_JIT
template <class T, class F, bool MIXED, int V>
__global__ void __launch_bounds__(256) k_reduce_all(RedArgs a, F f) {
    reduce_all_body<T, F, MIXED, V>(a, f);
}
template <class T, class F, bool MIXED>
__global__ void __launch_bounds__(256) k_reduce_part(RedArgs a, F f) {
    reduce_part_body<T, F, MIXED>(a, f);
}
template <class T, class F, bool MIXED, int V>
__global__ void __launch_bounds__(256) k_reduce_row(RedArgs a, F f) {
    reduce_row_body<T, F, MIXED, V>(a, f);
}
template <class T, class F, bool MIXED, int V>
__global__ void __launch_bounds__(256) k_reduce_col(RedArgs a, F f) {
    reduce_col_body<T, F, MIXED, V>(a, f);
}

// ---- launchers ----------------------------------------------------------------------------------------
template <class T, class F, bool MIXED, int V>
static int launch_all(const Canon& c, const RedArgs& a, int blocks, hipStream_t s, F f) {
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "reduce";
        l.tname = tname<T>();
        l.argtype = "smr::RedArgs";
        l.entry = std::string("smr::reduce_all_body<") + tname<T>() + ", smr::FJit, " + (MIXED ? "true" : "false") + ", " +
                  std::to_string(V) + ">(a, smr::FJit{kc});";
        l.grid = (unsigned)blocks;
        l.block = 256;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        SMR_LAUNCH((k_reduce_all<T, F, MIXED, V>), dim3(blocks), dim3(256), 0, s, a, f);
        return check_launch("k_reduce_all");
    }
}

static void fill_args(const Plan& plan, void* const* bases, RedArgs& a) {
    const Canon& c = plan.c;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    a.N = c.N;
    a.NK = c.NK;
    a.M = c.M;
    a.redop = c.redop;
    a.initop = c.initop;
    a.total = c.total;
    a.nout = c.nout;
    a.nred = c.total / c.nout;
    a.beta[0] = c.initarg[0];
    a.beta[1] = c.initarg[1];
    for (int i = 0; i < MAXN; ++i) a.dims[i] = (i < c.N) ? c.dims[i] : 1;
    for (int k = 0; k < MAXM; ++k)
        for (int i = 0; i < MAXN; ++i) a.strides[k][i] = (k < c.M && i < c.N) ? c.strides[k][i] : 0;
    a.partials = plan.scratch;
    a.counters = plan.scratch ? (unsigned*)((char*)plan.scratch + plan.counter_off) : nullptr;
    a.single = 0;  // set by go_all / go_part when the number of partials per output is at most Options::reduce_single
    // non-temporal loads: 64 MiB 18.5 -> 16.5 us, 512 MiB 96 -> 92 us, 4 GiB 700 +- 15 us either way (tools/reduce_nt.py)
    a.ntl = (options().nt_load > 0 || (options().nt_load < 0 && (long double)c.total * c.esize[1] < 2147483648.0L)) ? 1 : 0;
}

template <class T, class F, bool MIXED>
static int go_all(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    RedArgs a;
    fill_args(plan, bases, a);
    a.linear = (c.N == 1) ? 1 : 0;
    int blocks = plan.red_blocks;
    if (blocks > 1 && !plan.scratch) blocks = 1;
    a.nparts = blocks;
    // up to 64 partials are folded inside the launch (1 MiB: 4.9 -> 4.1 us; tools/reduce_all_sweep.py) -- the arrivals of ONE
    // counter serialise, so larger reductions keep the second launch (8 MiB with 256 workgroups: 6.6 vs 5.4 us)
    const i64 rs = options().reduce_single;
    a.single = (blocks > 1 && rs > 0 && blocks <= std::max<i64>(rs, 64)) ? 1 : 0;
    // vector path: one fused dim, every input unit stride / broadcast, 16-B aligned
    constexpr int VMAX = (sizeof(T) >= 16) ? 1 : (int)(16 / sizeof(T));
    bool vec = !MIXED && VMAX > 1 && c.N == 1 && (c.total % VMAX == 0) && c.total >= 4096;
    for (int k = 1; k < c.M && vec; ++k) {
        if (c.strides[k][0] == 0) continue;
        if (c.strides[k][0] != 1) vec = false;
        if (((uintptr_t)a.ops.base[k]) % 16) vec = false;
    }
    int rc = SMR_OK;
    bool done = false;
    if constexpr (!MIXED && VMAX > 1) {
        if (vec) {
            rc = launch_all<T, F, false, VMAX>(c, a, blocks, s, f);
            done = true;
        }
    }
    if (!done) rc = launch_all<T, F, MIXED, 1>(c, a, blocks, s, f);
    if (rc) return rc;
    if (blocks > 1 && !a.single && !jit_no_launch()) {
        clear_sticky_error();
    SMR_LAUNCH((k_reduce_final<T, MIXED>), dim3(1), dim3(256), 0, s, a);
        rc = check_launch("k_reduce_final");
    }
    return rc;
}

// launches one of the three partial-reduction bodies (KIND 0 general, 1 ROW, 2 COL)
template <class T, class F, bool MIXED, int KIND, int V>
static int launch_part(const Canon& c, const RedArgs& a, i64 blocks, hipStream_t s, F f) {
    if constexpr (is_jit<F>::value) {
        static const char* body[] = {"reduce_part_body", "reduce_row_body", "reduce_col_body"};
        JitLaunch l;
        l.family = "reduce";
        l.tname = tname<T>();
        l.argtype = "smr::RedArgs";
        l.entry = std::string("smr::") + body[KIND] + "<" + tname<T>() + ", smr::FJit, " + (MIXED ? "true" : "false") +
                  (KIND ? ", " + std::to_string(V) : std::string()) + ">(a, smr::FJit{kc});";
        l.grid = (unsigned)blocks;
        l.block = 256;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        if constexpr (KIND == 0) SMR_LAUNCH((k_reduce_part<T, F, MIXED>), dim3((unsigned)blocks), dim3(256), 0, s, a, f);
        if constexpr (KIND == 1) SMR_LAUNCH((k_reduce_row<T, F, MIXED, V>), dim3((unsigned)blocks), dim3(256), 0, s, a, f);
        if constexpr (KIND == 2) SMR_LAUNCH((k_reduce_col<T, F, MIXED, V>), dim3((unsigned)blocks), dim3(256), 0, s, a, f);
        return check_launch("k_reduce_part");
    }
}

template <class T, class F, bool MIXED>
static int go_part(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    RedArgs a;
    fill_args(plan, bases, a);
    const bool have_scratch = plan.scratch != nullptr;
    int nsplit = (plan.part_split > 1 && have_scratch) ? plan.part_split : 1;
    a.nsplit = nsplit;
    a.single = (nsplit > 1 && nsplit <= options().reduce_single) ? 1 : 0;
    a.nshard = 0;
    a.partials2 = nullptr;
    // more chunks than one counter takes cheaply: two levels of tickets (shards of about sqrt(nsplit), at most RED_SHARDS -- the
    // plan's scratch holds RED_SHARDS shard partials per output behind the chunk partials)
    const bool tree = nsplit > options().reduce_single && options().reduce_tree > 0 && nsplit <= options().reduce_tree;
    if (tree) {
        int ns = 2;
        while (ns < RED_SHARDS && ns * ns < nsplit) ns <<= 1;
        a.single = 1;
        a.nshard = ns;
        a.partials2 = (char*)plan.scratch + (size_t)c.nout * (size_t)plan.part_split * sizeof(T);
    }
    a.cpad = 1;
    auto counters_fit = [&](i64 groups) {
        const i64 n = groups * (a.nshard > 0 ? a.nshard + 1 : 1);
        if (a.nshard > 0 && n * 32 <= RED_COUNTERS) a.cpad = 32;
        return n <= RED_COUNTERS;
    };
    a.xsplit = a.qsplit = 1;
    i64 blocks = 0;
    int rc;
    constexpr int VMAX = (MIXED || sizeof(T) >= 16) ? 1 : (int)(16 / sizeof(T));
    if (plan.part_kind == 0) {
        a.tr = plan.part_tr;
        a.trlog = 0;
        while ((1 << a.trlog) < a.tr) ++a.trlog;
        const int ob = 256 / a.tr;
        const i64 groups = (c.nout + ob - 1) / ob;
        a.ngroups = (int32_t)groups;
        if (!counters_fit(groups)) a.single = a.nshard = 0;
        a.chunk = ((a.nred + nsplit - 1) / nsplit + a.tr - 1) / a.tr * a.tr;
        blocks = groups * nsplit;
        if (blocks > 0x7fffffffLL || groups > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "reduce grid too large");
        rc = launch_part<T, F, MIXED, 0, 1>(c, a, blocks, s, f);
    } else {
        a.g0log = plan.part_g0log;
        a.g1log = plan.part_g1log;
        a.txlog = plan.part_txlog;
        a.L0 = c.dims[c.NK];
        a.Q = a.nred / a.L0;
        if (nsplit > 1) {
            a.xsplit = plan.part_xsplit;
            a.qsplit = plan.part_qsplit;
        }
        a.qchunk = (a.Q + a.qsplit - 1) / a.qsplit;
        // vector width: the vector axis must divide, every vector-loaded operand must be aligned
        const int vax = plan.part_kind == 1 ? c.NK : 0;  // axis the vectors run along
        bool vec = VMAX > 1 && (c.dims[vax] % VMAX == 0);
        for (int k = 1; k < c.M && vec; ++k) {
            if (c.strides[k][vax] != 1) continue;
            if (((uintptr_t)a.ops.base[k]) % 16) vec = false;
            for (int d = 0; d < c.N; ++d)
                if (d != vax && (c.strides[k][d] % VMAX)) vec = false;
        }
        const int V = vec ? VMAX : 1;
        if (plan.part_kind == 1) {
            const i64 unit = ((i64)V) << a.g0log;
            a.xchunk = ((a.L0 + a.xsplit - 1) / a.xsplit + unit - 1) / unit * unit;
            const i64 groups = (c.nout + (256 >> (a.g0log + a.g1log)) - 1) / (256 >> (a.g0log + a.g1log));
            a.ngroups = (int32_t)groups;
            if (!counters_fit(groups)) a.single = a.nshard = 0;
            blocks = groups * nsplit;
            if (blocks > 0x7fffffffLL || groups > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "reduce grid too large");
            rc = SMR_OK;
            bool done = false;
            if constexpr (VMAX > 1) {
                if (vec) {
                    rc = launch_part<T, F, MIXED, 1, VMAX>(c, a, blocks, s, f);
                    done = true;
                }
            }
            if (!done) rc = launch_part<T, F, MIXED, 1, 1>(c, a, blocks, s, f);
        } else {
            a.xchunk = (a.L0 + a.xsplit - 1) / a.xsplit;
            a.ctx = 1 << a.txlog;
            a.cty = 256 >> a.txlog;
            a.cy0 = 1 << a.g0log;
            a.cy1 = 1 << a.g1log;
            if (plan.part_col_tx > 0 && plan.part_col_v == V) {  // exact lane map (smr_plan.cpp), planned for this vector width
                a.ctx = plan.part_col_tx;
                a.cty = 256 / a.ctx;
                a.cy0 = plan.part_col_y0;
                a.cy1 = plan.part_col_y1;
            }
            a.ctx_m = (65536u + (uint32_t)a.ctx - 1u) / (uint32_t)a.ctx;
            a.cy0_m = (65536u + (uint32_t)a.cy0 - 1u) / (uint32_t)a.cy0;
            const i64 per = (i64)V * a.ctx;
            a.nkb0 = (c.dims[0] + per - 1) / per;
            const i64 groups = a.nkb0 * (c.nout / c.dims[0]);
            a.ngroups = (int32_t)groups;
            if (!counters_fit(groups)) a.single = a.nshard = 0;
            blocks = groups * nsplit;
            if (blocks > 0x7fffffffLL || groups > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "reduce grid too large");
            rc = SMR_OK;
            bool done = false;
            if constexpr (VMAX > 1) {
                if (vec) {
                    rc = launch_part<T, F, MIXED, 2, VMAX>(c, a, blocks, s, f);
                    done = true;
                }
            }
            if (!done) rc = launch_part<T, F, MIXED, 2, 1>(c, a, blocks, s, f);
        }
    }
    if (rc || nsplit == 1 || a.single || jit_no_launch()) return rc;
    // lanes per output of the folding pass: as many as there are partials (up to a wave), fewer
    // when there are plenty of outputs anyway
    int lpolog = 0;
    while (lpolog < 6 && (4 << lpolog) < nsplit && (c.nout << lpolog) < 256 * 1024) ++lpolog;
    a.trlog = lpolog;
    const i64 fthreads = c.nout << lpolog;
    SMR_LAUNCH((k_reduce_part_final<T, MIXED>), dim3((unsigned)((fthreads + 255) / 256)), dim3(256), 0, s, a);
    return check_launch("k_reduce_part_final");
}

template <>
int launch_reduce_all_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.mixed) return with_prog<T>(c, [&](auto f) { return go_all<T, decltype(f), true>(plan, bases, s, f); });
    const unsigned mask = fbit(FK_IDENT) | fbit(FK_ABS2) | fbit(FK_MUL2);
    return with_functor<T>(c, mask, [&](auto f) { return go_all<T, decltype(f), false>(plan, bases, s, f); });
}

template <>
int launch_reduce_part_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.mixed) return with_prog<T>(c, [&](auto f) { return go_part<T, decltype(f), true>(plan, bases, s, f); });
    const unsigned mask = fbit(FK_IDENT) | fbit(FK_MUL2);
    return with_functor<T>(c, mask, [&](auto f) { return go_part<T, decltype(f), false>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
