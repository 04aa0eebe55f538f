// smr_k_stream.hip -- family STREAM: every operand is unit-stride (or broadcast, stride 0)
// along the destination's fast dim.  Contiguous broadcasts, axpy-type updates, the
// compute-bound README map.  16-byte vector accesses, U vectors in flight per lane, outer
// dims resolved once per workgroup with scalar arithmetic.
#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif

namespace smr {

struct StreamArgs {
    OpTab ops;
    int32_t N, M;
    i64 n0v;   // whole vectors along dim 0
    i64 n0t;   // column slots along dim 0 = n0v, + 1 when the rows end in a partial vector (round 5)
    int32_t tail, pad2;  // elements of that partial vector (0 < tail < V), 0: rows are whole vectors
    i64 rows;  // product of the outer dims
    i64 bpr;   // workgroups per row
    int32_t txlog, nts;   // nts: non-temporal stores (big streaming outputs, Options::nt_store); txlog 8: a workgroup covers U x 256 vectors of ONE row; < 8 (short rows): 2^txlog lanes
                          // along dim 0 x (256 >> txlog) x U entries of dim 1 -- then `rows` counts dims >= 2
    int32_t ncc, packed;  // packed (round 4): rows of 129 .. 128*U vectors -- the packed form with 256 lanes along dim 0, ncc column chunks per row and U / ncc entries
                          // of dim 1 per workgroup (the one-row form leaves (U - ncc) / U of every lane's slots empty: rows of 257 Float64 moved 2 KiB per workgroup)
    i64 dims[MAXN];
    i64 strides[MAXM][MAXN];
};

template <class T, int V>
struct alignas(sizeof(T) * V) Vec {
    T v[V];
};
// the same 16 bytes with ELEMENT alignment: what the global accesses use.  The hardware takes a dwordx4 access at any dword address
// (the compiler emits it for align 4 / 8: checked in the ISA), so rows that start at odd element offsets -- odd row lengths, views
// that begin in the middle of a vector -- move as vectors too, plus one partial vector per row (round 5; before: 8-byte accesses,
// (257,129,65) (0,2,1) 2.66 TB/s)
template <class T, int V>
struct alignas(sizeof(T)) UVec {
    T v[V];
};

template <class T, class F, bool MIXED, int V, int U, int FORM>  // FORM 0: one row segment per workgroup; 1: packed short rows; 2: packed rows of several column chunks
SMR_DEV void stream_map_body(const StreamArgs a, F f) {
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    i64 row = 0, cb = blockIdx.x;
    if (a.rows > 1) {
        row = cb / a.bpr;
        cb -= row * a.bpr;
    }
    constexpr bool flat = FORM == 0;       // one row segment per workgroup (compile-time: the classic form stays lean)
    const int txlog = (FORM != 1) ? 8 : a.txlog;
    const int ncc = (FORM == 2) ? a.ncc : 1, rpw = (FORM == 2) ? U / ncc : U;  // packed form: column chunks per row, rows per lane
    const int dfirst = flat ? 1 : 2;       // first dim resolved per workgroup (scalar arithmetic)
    const int tx = threadIdx.x & ((1 << txlog) - 1), ty = threadIdx.x >> txlog, TY = 256 >> txlog;
    i64 roff[MAXM];
#pragma unroll
    for (int k = 0; k < MAXM; ++k) roff[k] = 0;
    if (a.rows > 1) {
        i64 rem = row;
#pragma unroll
        for (int d = 1; d < MAXN; ++d) {
            if (d >= dfirst && d < a.N) {
                const i64 q = rem / a.dims[d];
                const i64 c = rem - q * a.dims[d];
                rem = q;
#pragma unroll
                for (int k = 0; k < MAXM; ++k)
                    if (k < a.M) roff[k] += c * a.strides[k][d];
            }
        }
    }
    typedef Vec<T, V> VT;
    typedef UVec<T, V> GT;  // global-memory view of a vector
    VT in[U][MAXIN];
    bool part[U];           // this slot is the partial vector at the end of its row
    i64 col[U];
    i64 joff[U][MAXM];  // offset of this lane's dim-1 entry (short-row form), 0 otherwise
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        i64 j = 0;
        if (flat) {
            col[u] = (cb * U + u) * 256 + threadIdx.x;
            live[u] = col[u] < a.n0t;
        } else {
            if constexpr (FORM == 2) {
                const int cc = u % ncc, jj = u / ncc;
                col[u] = tx + ((i64)cc << 8);
                j = cb * rpw + jj;
                live[u] = col[u] < a.n0t && j < a.dims[1] && jj < rpw;
            } else {
                col[u] = tx;
                j = (cb * U + u) * TY + ty;
                live[u] = tx < a.n0t && j < a.dims[1];
            }
        }
#pragma unroll
        for (int k = 0; k < MAXM; ++k) joff[u][k] = (k < a.M) ? (flat ? roff[k] : roff[k] + j * a.strides[k][1]) : 0;
        part[u] = V > 1 && a.tail != 0 && col[u] == a.n0v;
        if (live[u]) {
#pragma unroll
            for (int k = 0; k < MAXIN; ++k) {
                if (k < nin) {
                    if (a.strides[k + 1][0] == 0) {
                        const T s = load_op<T, MIXED>(a.ops, k + 1, joff[u][k + 1]);
#pragma unroll
                        for (int e = 0; e < V; ++e) in[u][k].v[e] = s;
                    } else if constexpr (MIXED || V == 1) {
                        in[u][k].v[0] = load_op<T, MIXED>(a.ops, k + 1, joff[u][k + 1] + col[u] * a.strides[k + 1][0]);
                    } else {
                        const T* src = (const T*)a.ops.base[k + 1] + joff[u][k + 1] + col[u] * V;
                        if (!part[u]) {
                            const GT g = *reinterpret_cast<const GT*>(src);
#pragma unroll
                            for (int e = 0; e < V; ++e) in[u][k].v[e] = g.v[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < V; ++e) in[u][k].v[e] = e < a.tail ? src[e] : T{};
                        }
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
    }
    VT out[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (live[u]) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                T x[MAXIN];
#pragma unroll
                for (int k = 0; k < MAXIN; ++k) {
                    x[k] = T{};
                    if (k < nin) x[k] = in[u][k].v[e];
                }
                out[u].v[e] = f(x);
            }
            if constexpr (MIXED || V == 1) {
                store_op<T, MIXED>(a.ops, joff[u][0] + col[u] * a.strides[0][0], out[u].v[0]);
            } else {
                if constexpr (tr<T>::cx) {
                    if (a.ops.conj[0]) {
#pragma unroll
                        for (int e = 0; e < V; ++e) out[u].v[e] = cj(out[u].v[e]);
                    }
                }
            }
        }
    }
    if constexpr (!(MIXED || V == 1)) {
        // one wave-uniform branch around all vector stores (see smr_device.h:store_vec)
        auto put = [&](auto NT) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (live[u] && !part[u]) {
                    GT g;
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] = out[u].v[e];
                    store_vec_ct<decltype(NT)::value, GT>(reinterpret_cast<char*>((T*)a.ops.base[0] + joff[u][0] + col[u] * V), g);
                }
        };
        if (a.nts) {
            nt_block_guard();
            put(BoolC<true>{});
            nt_block_guard();
        } else {
            put(BoolC<false>{});
        }
        if (a.tail != 0) {  // the partial vector at the end of a row: element by element
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (live[u] && part[u]) {
                    T* dst = (T*)a.ops.base[0] + joff[u][0] + col[u] * V;
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (e < a.tail) dst[e] = out[u].v[e];
                }
        }
    }
}

#ifndef SMR_JIT
template <class T, class F, bool MIXED, int V, int U, int FORM>
__global__ void __launch_bounds__(256) k_stream_map(StreamArgs a, F f SMR_STAMP_PARAM) {
    SMR_STAMP_BEGIN
    stream_map_body<T, F, MIXED, V, U, FORM>(a, f);
    SMR_STAMP_END
}

template <class T, class F, bool MIXED, int V, int UOVR = 0>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    // vectors in flight per lane: measured on 256-512 MiB maps, 2 x 16 B beats 4 (2-3 %) and 8 (5-8 % worse)
    constexpr int U = UOVR ? UOVR : ((sizeof(T) * V >= 16) ? 2 : 8);
    if constexpr (is_jit<F>::value && UOVR == 0) {  // experiment (runtime-compiled functors only): option stream_u
        if (options().stream_u == 2) return go<T, F, MIXED, V, 2>(plan, bases, s, f);
        if (options().stream_u == 8) return go<T, F, MIXED, V, 8>(plan, bases, s, f);
    }
    StreamArgs a;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    a.N = c.N;
    a.M = c.M;
    a.n0v = c.dims[0] / V;
    a.tail = (int32_t)(c.dims[0] % V);
    a.n0t = a.n0v + (a.tail ? 1 : 0);
    {
        // Vector stores of this family are non-temporal at EVERY size (nt_stream_min = 0).  Measured
        // (profiles/r02_nt_store_ab.txt): configs[4] 8192^2 f32 91.8 -> 80.2 us, 2048^2 copy 5.38 -> 3.88 us, and a
        // non-temporal producer never slowed the kernel that reads its output (producer -> consumer pairs at 32^4,
        // 64^4, 2048^2: 9.88 / 101.9 / 12.7 -> 9.55 / 91.5 / 10.6 us) -- the dirty lines leave during the kernel
        // instead of at its end.  "nt_stream_min" raises the threshold for callers who want small outputs cached.
        const Options& o = options();
        const i64 outbytes = c.nout * (i64)c.esize[0];
        a.nts = (o.nt_store > 0 || (o.nt_store < 0 && outbytes >= o.nt_stream_min)) ? 1 : 0;
    }
    // short rows (sub-boxes): pack (256 >> txlog) entries of dim 1 into a workgroup instead of leaving
    // most of its lanes idle (measured on 100-element rows: 0.93 -> TB/s below)
    a.txlog = 8;
    a.ncc = 1;
    a.packed = 0;
    if (c.N >= 2 && a.n0t <= 128) {
        a.txlog = 0;
        while ((1 << a.txlog) < a.n0t) ++a.txlog;
        a.packed = 1;
    } else if (c.N >= 2 && U >= 2 && a.n0t <= 128 * U && options().stream_pack_rows) {
        // measured (tools/stream_pack_ab.py, profiles/r04_stream_pack_ab.txt): one or two column chunks per row win (rows of 257 Float64
        // 13.5 -> 12.0 us, 300 / 400 11.5 -> 8.4 us, Float32 rows of 700 / 1000 11.9 -> 8.7 / 12.2 -> 10.7 us), three lose
        // (513, 561: 15.3 -> 18.0 us), and so do problems that are left with fewer than ~1000 workgroups ((257,33,31): 3.3 -> 4.4 us)
        const int ncc = (int)((a.n0t + 255) / 256);
        i64 nrows = 1;
        for (int i = 1; i < c.N; ++i) nrows *= c.dims[i];
        if (ncc <= 2 && nrows / (U / ncc) >= 1024) {
            a.ncc = ncc;
            a.packed = 1;
        }
    }
    a.rows = 1;
    for (int i = (a.packed ? 2 : 1); i < c.N; ++i) a.rows *= c.dims[i];
    const i64 rows_per_wg = (i64)(256 >> a.txlog) * (U / a.ncc);
    a.bpr = !a.packed ? (a.n0t + 256 * U - 1) / (256 * U) : (c.dims[1] + rows_per_wg - 1) / rows_per_wg;
    for (int i = 0; i < MAXN; ++i) a.dims[i] = (i < c.N) ? c.dims[i] : 1;
    for (int k = 0; k < MAXM; ++k)
        for (int i = 0; i < MAXN; ++i) a.strides[k][i] = (k < c.M && i < c.N) ? c.strides[k][i] : 0;
    const i64 grid = a.bpr * a.rows;
    if (grid > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "stream grid too large");
    const int form = !a.packed ? 0 : (a.txlog < 8 ? 1 : 2);
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "stream";
        l.tname = tname<T>();
        l.argtype = "smr::StreamArgs";
        l.entry = std::string("smr::stream_map_body<") + tname<T>() + ", smr::FJit, " + (MIXED ? "true" : "false") + ", " +
                  std::to_string(V) + ", " + std::to_string(U) + ", " + std::to_string(form) + ">(a, smr::FJit{kc});";
        l.grid = (unsigned)grid;
        l.block = 256;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        if (form == 0)
            SMR_LAUNCH((k_stream_map<T, F, MIXED, V, U, 0>), dim3((unsigned)grid), dim3(256), 0, s, a, f SMR_STAMP_ARG(grid, 256));
        else if (form == 1)
            SMR_LAUNCH((k_stream_map<T, F, MIXED, V, U, 1>), dim3((unsigned)grid), dim3(256), 0, s, a, f SMR_STAMP_ARG(grid, 256));
        else
            SMR_LAUNCH((k_stream_map<T, F, MIXED, V, U, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, f SMR_STAMP_ARG(grid, 256));
        return check_launch("k_stream_map");
    }
}

template <class T, class F>
static int go_vec(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    constexpr int VMAX = (sizeof(T) >= 16) ? 1 : (int)(16 / sizeof(T));
    if constexpr (VMAX > 1) {
        if (plan.vec == VMAX) {
            // the plan chose vectors for the pointers it was created with; re-check rebound ones (element-aligned vectors --
            // Plan::vec_ua -- take any element address)
            bool aligned = true;
            if (bases && !plan.vec_ua) {
                const OpTab tab = make_optab(plan.c, bases);
                for (int k = 0; k < plan.c.M; ++k)
                    if (plan.c.strides[k][0] != 0 && ((uintptr_t)tab.base[k]) % 16) aligned = false;
            }
            if (aligned) return go<T, F, false, VMAX>(plan, bases, s, f);
        }
    }
    return go<T, F, false, 1>(plan, bases, s, f);
}

template <>
int launch_stream_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.bitcopy) {
#if SMR_CT == SMR_F32
        switch (c.esize[0]) {
            case 1: return go_vec<b8>(plan, bases, s, FIdent<b8>{});
            case 2: return go_vec<b16>(plan, bases, s, FIdent<b16>{});
            case 4: return go_vec<float>(plan, bases, s, FIdent<float>{});
            case 8: return go_vec<double>(plan, bases, s, FIdent<double>{});
            default: return go_vec<c64>(plan, bases, s, FIdent<c64>{});
        }
#else
        return set_error(SMR_EINVAL, "bitcopy is dispatched through the f32 object");
#endif
    }
    if (c.mixed) return with_prog<T>(c, [&](auto f) { return go<T, decltype(f), true, 1>(plan, bases, s, f); });
    return with_functor<T>(c, FMASK_ALL, [&](auto f) { return go_vec<T>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
