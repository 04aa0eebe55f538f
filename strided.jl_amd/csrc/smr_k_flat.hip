// smr_k_flat.hip -- family FLAT (round 3): transposing unary maps in which ONE side's contiguous memory run is a short group of
// leading dims whose extents are not powers of two -- permutedims (640,480,3) <-> (3,480,640) (image planes <-> interleaved
// channels), tensor-network shapes with physical dims of 3 (100,3,100,3,10), ... -- while the other side is unit-stride along a
// long dim.  The power-of-two tiles of the TILED family idle a quarter of their lanes on an extent of 3 and fall back to 8-byte
// accesses (1.1-2.0 TB/s, profiles/r02_perf_sanity.txt).  Here the flat side is addressed through its FLATTENED run: the leading
// dims d_0..d_{g-1} (extents multiply to R) are taken whole, the next contiguous dim p contributes a power-of-two tile TP, and
// j = r + R * jp enumerates L = R * TP consecutive elements of memory (16-byte vectors whenever L * sizeof(T) allows); the line
// side walks its own unit axis q in 16-byte vectors (tile TQ).  The L x TQ tile crosses LDS once (row pitch padded by one vector).
// DIR 0: the destination is the flat side; DIR 1: the input is.  Reference semantics: a pure map! with one input
// (src/mapreduce.jl:38-53 -> _mapreduce_kernel! :229-349), any unary f.
#ifndef SMR_JIT
#include <cstdio>
#include <cstdlib>
#endif

#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif

namespace smr {

constexpr int FLAT_MAXR = 64;    // two-sided form: entries per offset table
constexpr int FLAT_MAXR1 = 128;  // one-sided form (round 6: whole rows of up to 128 elements whose byte length is not a multiple of the 128-byte line)

struct FlatArgs {
    OpTab ops;
    int32_t N, dir, R, TPlog, TQ, L;      // L = R << TPlog; TQ: tile of the line side (a vector multiple, not necessarily a power of two)
    int32_t p, q;                         // tiled group dim of the flat side (-1: none), unit axis of the line side
    int32_t nouter, conjv;                // dims handled by the block index besides p and q; conjv: any conj flag set
    int32_t nin, kt;                      // inputs of f; kt: the ONE input with the other layout (crosses LDS) -- every other input has the
                                          // destination's strides and is read in phase 2 at the destination's offsets
    int32_t fuse, lshare;                  // fuse: the flat side continues along q itself (stride of q = R): the whole R x TQ tile is ONE run;
                                          // lshare: the LINE side is unit-stride along the shared lead and continues along q (stride R): a
                                          // transposition of R-element groups ((3,W,H) -> (3,H,W)); its rows are runs of R * TQ elements
    uint32_t ntp, ntq;                    // tiles along p / q
    uint32_t magicR, magicLv, magicXV, magicRow;  // floor(2^32 / d) + 1 for d = R, L / VF, TQ / VL, R * TQ / VL
    i64 dimp, dimq;                       // extents of p (1 when p < 0) and q
    i64 sfq, slp;                         // flat-side stride of q, line-side stride of p
    i64 sfp;                              // flat-side stride of p (= R)
    int32_t roff[FLAT_MAXR1];             // line-side element offset of the leading index r
    int32_t odim[MAXN];                   // outer dims (neither in the group nor p nor q)
    i64 oext[MAXN], osf[MAXN], osl[MAXN]; // their extents and flat- / line-side strides
};

template <class T, int V>
struct alignas(sizeof(T) * V) FVec {
    T v[V];
};

// n / d for n < 65536, d < 65536 with magic = floor(2^32 / d) + 1
SMR_DEV uint32_t fdiv16(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

// Phase 2 of every form: V consecutive destination elements at element offset `off`.  tv[] = what the transposed input (operand kt)
// contributed through LDS; the other inputs share the destination's layout and are loaded here, at the same offset.
// (C .= beta .* C .+ alpha .* permutedims(A, p) -- TensorOperations' tensoradd! -- is the model case.)
template <class T, class F, int V>
SMR_DEV void flat_emit(const OpTab& ops, int nin_rt, int kt, bool anyconj, i64 off, const T (&tv)[V], F& f) {
    const int nin = (F::NIN >= 0) ? F::NIN : nin_rt;
    FVec<T, V> dv[MAXIN];
#pragma unroll
    for (int k = 0; k < MAXIN; ++k)
        if (k < nin && k + 1 != kt) dv[k] = *reinterpret_cast<const FVec<T, V>*>((const T*)ops.base[k + 1] + off);
    FVec<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        T arg[MAXIN];
#pragma unroll
        for (int k = 0; k < MAXIN; ++k) {
            arg[k] = T{};
            if (k < nin) {
                T v = (k + 1 == kt) ? tv[e] : dv[k].v[e];
                if constexpr (tr<T>::cx) {
                    if (anyconj && ops.conj[k + 1]) v = cj(v);
                }
                arg[k] = v;
            }
        }
        T r = f(arg);
        if constexpr (tr<T>::cx) {
            if (anyconj && ops.conj[0]) r = cj(r);
        }
        o.v[e] = r;
    }
    *reinterpret_cast<FVec<T, V>*>((T*)ops.base[0] + off) = o;
}

// VL / VF: elements per access on the line / flat side (1 or 16 bytes' worth)
template <class T, class F, int DIR, int VL, int VF>
SMR_DEV void flat_map_body(const FlatArgs a, F f) {
    constexpr int PAD = (16 / (int)sizeof(T)) > 0 ? (16 / (int)sizeof(T)) : 1;
    const int TQ = a.TQ;                   // tile of the line side (runtime: ~32 / 64 elements, up to 512 in the fused form)
    const int PITCH = TQ + PAD;
    const int XV = TQ / VL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_flat[];
    T* lds = reinterpret_cast<T*>(smem_flat);
    const uint32_t tid = threadIdx.x;
    // ---- tile origin: block -> (tq, tp, outer index) -------------------------------------------------------------
    uint32_t b = blockIdx.x;
    const uint32_t tq = b % a.ntq;
    b /= a.ntq;
    const uint32_t tp = b % a.ntp;
    b /= a.ntp;
    i64 bf = 0, bl = 0;  // element offsets of the tile origin on the flat / line side
    for (int i = 0; i < a.nouter; ++i) {
        const uint32_t e = (uint32_t)a.oext[i];
        const uint32_t c = b % e;
        b /= e;
        bf += (i64)c * a.osf[i];
        bl += (i64)c * a.osl[i];
    }
    const i64 q0 = (i64)tq * TQ, p0 = (i64)tp << a.TPlog;
    bf += q0 * a.sfq + p0 * a.sfp;
    bl += q0 * (a.lshare ? a.R : 1) + p0 * a.slp;
    const int nq = (int)((a.dimq - q0 < TQ) ? (a.dimq - q0) : TQ);                       // valid columns
    const i64 vp = (a.dimp - p0 < ((i64)1 << a.TPlog)) ? (a.dimp - p0) : ((i64)1 << a.TPlog);
    const int nj = (int)(vp * a.R);                                                        // valid flat run (a contiguous prefix)
    const T* src = (const T*)a.ops.base[a.kt];
    const bool anyconj = a.conjv != 0;

    // ---- phase 1: global -> LDS[j][x] --------------------------------------------------------------------------------
    if constexpr (DIR == 0) {
        // the input is the line side: 16-byte vectors along q
        const int nvec = a.L * XV;
        if (a.lshare) {
            // rows jp of R * nq contiguous elements t = x * R + r
            const int rowv = (a.R * TQ) / VL, nt = a.R * nq, njp = nj / a.R;
            for (int v = (int)tid; v < rowv * njp; v += 256) {
                const int jp = (int)fdiv16((uint32_t)v, a.magicRow), t0 = (v - jp * rowv) * VL;
                if (t0 < nt) {
                    const FVec<T, VL> t = *reinterpret_cast<const FVec<T, VL>*>(src + bl + (i64)jp * a.slp + t0);
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const int x = (int)fdiv16((uint32_t)(t0 + e), a.magicR), r = t0 + e - x * a.R;
                        lds[(r + a.R * jp) * PITCH + x] = t.v[e];
                    }
                }
            }
        } else
        {   // four vectors per lane and round, loaded before the first LDS write (round 6; a slot outside the tile re-reads the
            // tile's first vector and drops it)
            constexpr int UB = 4;
            const i64 safe = bl + a.roff[0];
            for (int v0 = (int)tid; v0 < nvec; v0 += 256 * UB) {
                FVec<T, VL> t[UB];
                int at[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int v = v0 + u * 256;
                    const int j = (int)fdiv16((uint32_t)v, a.magicXV), x = (v - j * XV) * VL;
                    const bool ok = v < nvec && j < nj && x < nq;
                    const uint32_t jp = fdiv16((uint32_t)(ok ? j : 0), a.magicR);
                    const int r = (ok ? j : 0) - (int)jp * a.R;
                    at[u] = ok ? j * PITCH + x : -1;
                    t[u] = *reinterpret_cast<const FVec<T, VL>*>(src + (ok ? bl + a.roff[r] + (i64)jp * a.slp + x : safe));
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (at[u] >= 0) {
#pragma unroll
                        for (int e = 0; e < VL; ++e) lds[at[u] + e] = t[u].v[e];
                    }
            }
        }
    } else {
        // the input is the flat side: 16-byte vectors along the flattened run
        const int LV = a.L / VF;
        const int nvec = LV * TQ;
        if (a.fuse) {  // planar <-> interleaved: element t = x * R + r of the tile, all of it contiguous
            const int nt = a.R * nq;
            for (int t0 = (int)tid * VF; t0 < nt; t0 += 256 * VF) {
                const FVec<T, VF> t = *reinterpret_cast<const FVec<T, VF>*>(src + bf + t0);
#pragma unroll
                for (int e = 0; e < VF; ++e) {
                    const int x = (int)fdiv16((uint32_t)(t0 + e), a.magicR), r = t0 + e - x * a.R;
                    lds[r * PITCH + x] = t.v[e];
                }
            }
        } else
        {   // (four vectors per lane and round, as above)
            constexpr int UB = 4;
            for (int v0 = (int)tid; v0 < nvec; v0 += 256 * UB) {
                FVec<T, VF> t[UB];
                int at[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int v = v0 + u * 256;
                    const int x = (int)fdiv16((uint32_t)v, a.magicLv), j = (v - x * LV) * VF;
                    const bool ok = v < nvec && j < nj && x < nq;
                    at[u] = ok ? j * PITCH + x : -1;
                    t[u] = *reinterpret_cast<const FVec<T, VF>*>(src + (ok ? bf + (i64)x * a.sfq + j : bf));
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (at[u] >= 0) {
#pragma unroll
                        for (int e = 0; e < VF; ++e) lds[at[u] + e * PITCH] = t[u].v[e];
                    }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: LDS -> f -> global ------------------------------------------------------------------------------------
    if constexpr (DIR == 0) {
        const int LV = a.L / VF;
        const int nvec = LV * TQ;
        if (a.fuse) {
            const int nt = a.R * nq;
            for (int t0 = (int)tid * VF; t0 < nt; t0 += 256 * VF) {
                T tv[VF];
#pragma unroll
                for (int e = 0; e < VF; ++e) {
                    const int x = (int)fdiv16((uint32_t)(t0 + e), a.magicR), r = t0 + e - x * a.R;
                    tv[e] = lds[r * PITCH + x];
                }
                flat_emit<T, F, VF>(a.ops, a.nin, a.kt, anyconj, bf + t0, tv, f);
            }
        } else
        for (int v = (int)tid; v < nvec; v += 256) {
            const int x = (int)fdiv16((uint32_t)v, a.magicLv), j = (v - x * LV) * VF;
            if (j < nj && x < nq) {
                T tv[VF];
#pragma unroll
                for (int e = 0; e < VF; ++e) tv[e] = lds[(j + e) * PITCH + x];
                flat_emit<T, F, VF>(a.ops, a.nin, a.kt, anyconj, bf + (i64)x * a.sfq + j, tv, f);
            }
        }
    } else {
        const int nvec = a.L * XV;
        if (a.lshare) {
            const int rowv = (a.R * TQ) / VL, nt = a.R * nq, njp = nj / a.R;
            for (int v = (int)tid; v < rowv * njp; v += 256) {
                const int jp = (int)fdiv16((uint32_t)v, a.magicRow), t0 = (v - jp * rowv) * VL;
                if (t0 < nt) {
                    T tv[VL];
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const int x = (int)fdiv16((uint32_t)(t0 + e), a.magicR), r = t0 + e - x * a.R;
                        tv[e] = lds[(r + a.R * jp) * PITCH + x];
                    }
                    flat_emit<T, F, VL>(a.ops, a.nin, a.kt, anyconj, bl + (i64)jp * a.slp + t0, tv, f);
                }
            }
        } else
        for (int v = (int)tid; v < nvec; v += 256) {
            const int j = (int)fdiv16((uint32_t)v, a.magicXV), x = (v - j * XV) * VL;
            if (j < nj && x < nq) {
                const uint32_t jp = fdiv16((uint32_t)j, a.magicR);
                const int r = j - (int)jp * a.R;
                T tv[VL];
#pragma unroll
                for (int e = 0; e < VL; ++e) tv[e] = lds[j * PITCH + x + e];
                flat_emit<T, F, VL>(a.ops, a.nin, a.kt, anyconj, bl + a.roff[r] + (i64)jp * a.slp + x, tv, f);
            }
        }
    }
}

// ---- two-sided form ------------------------------------------------------------------------------------------------
// Both sides' memory runs are short groups of leading dims (plan_flat2): run s = R[s] leading elements x a tile TP[s] of the next
// contiguous dim p[s], L[s] = R[s] * TP[s] <= 128 consecutive elements of side s's memory; side 0 = destination, 1 = input.
// A tile is the L[0] x L[1] product.  Phase 1 enumerates it input-run-major (lanes along the input's memory), phase 2
// destination-run-major; LDS[b][a] with an odd row pitch.  No lane is spent on padding: extents of 5, 7, 17, 31 fill the tile.
struct Flat2Args {
    OpTab ops;
    int32_t R[2], TP[2], L[2];
    int32_t nouter, conjv;
    int32_t nin, kt;                      // as in FlatArgs
    int32_t shared;                       // the input would continue along p[0] too (stride R[1]): phase 1 walks the destination run tile-index-major
    uint32_t magicTP0;
    uint32_t ntp[2];                      // tiles along p[0] / p[1]
    uint32_t magicR[2], magicL[2];        // floor(2^32 / d) + 1 for d = R[s], L[s]
    uint32_t magicI[2];                   // ... for d = (L[s] + 2) / 2: pair slots per run (PAIR form)
    i64 dimp[2];                          // extents of p[s] (1 when there is none)
    i64 spo[2];                           // stride of p[s] on the OTHER side
    int32_t roff[2][FLAT_MAXR];           // offset on the other side of leading index r of run s
    i64 oext[MAXN], os0[MAXN], os1[MAXN]; // outer dims: extents, destination / input strides
};

// PAIR (round 4): both phases move two consecutive elements of a run per lane wherever the pair starts on an address that is a multiple
// of 2 * sizeof(T) -- the runs start at odd offsets as often as not (a row of 257 elements shifts the next row's parity), so every row
// gets its pairs from ITS parity: slot i of a row holds elements 2i - p and 2i - p + 1 (p = parity of the row's first element), the
// first / last slot may hold a single element.  Halves the global-memory instructions of 4- and 8-byte element types.
template <class T, class F, bool PAIR = false>
SMR_DEV void flat2_body(const Flat2Args a, F f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_flat[];
    T* lds = reinterpret_cast<T*>(smem_flat);
    const uint32_t tid = threadIdx.x;
    uint32_t b = blockIdx.x;
    const uint32_t t0 = b % a.ntp[0];
    b /= a.ntp[0];
    const uint32_t t1 = b % a.ntp[1];
    b /= a.ntp[1];
    i64 bd = 0, bs = 0;  // element offsets of the tile origin in the destination / the input
    for (int i = 0; i < a.nouter; ++i) {
        const uint32_t e = (uint32_t)a.oext[i];
        const uint32_t c = b % e;
        b /= e;
        bd += (i64)c * a.os0[i];
        bs += (i64)c * a.os1[i];
    }
    const i64 p0 = (i64)t0 * a.TP[0], p1 = (i64)t1 * a.TP[1];
    bd += p0 * a.R[0] + p1 * a.spo[1];
    bs += p1 * a.R[1] + p0 * a.spo[0];
    const int n0 = (int)(((a.dimp[0] - p0 < a.TP[0]) ? (a.dimp[0] - p0) : a.TP[0]) * a.R[0]);  // valid prefix of the destination run
    const int n1 = (int)(((a.dimp[1] - p1 < a.TP[1]) ? (a.dimp[1] - p1) : a.TP[1]) * a.R[1]);  // ... of the input run
    const int L0 = a.L[0], L1 = a.L[1], PITCH = L0 | 1;
    const T* src = (const T*)a.ops.base[a.kt];
    const bool anyconj = a.conjv != 0;
    // the other side's offset of every position of the two runs, once per workgroup (the loops below then cost one LDS read per
    // element instead of two multiply-highs and a table load from the kernel arguments)
    // Phase 1 visits the destination run in the order x' -> column xcol[x']: the identity, or -- `shared`: both sides continue
    // along the SAME dim behind their leads, (5,N,7) -> (7,N,5) -- tile index fastest, so that consecutive (x', y) are consecutive
    // in the input's memory (r1 + R[1] * jd) although the input's own run is only its lead.
    i64* offs = reinterpret_cast<i64*>(smem_flat + (((size_t)L1 * PITCH * sizeof(T) + 15) & ~(size_t)15));  // input offset of x'
    i64* offd = offs + L0;                                                                                  // destination offset of y
    int* xcol = reinterpret_cast<int*>(offd + L1);
    for (int t = (int)tid; t < L0; t += 256) {
        int jp, r;
        if (a.shared) {
            r = (int)fdiv16((uint32_t)t, a.magicTP0);
            jp = t - r * a.TP[0];
        } else if (a.R[0] == 1) {  // (the magic number of 1 does not fit 32 bits) a run that is a tile of the unit dim itself
            jp = t;
            r = 0;
        } else {
            jp = (int)fdiv16((uint32_t)t, a.magicR[0]);
            r = t - jp * a.R[0];
        }
        offs[t] = (i64)a.roff[0][r] + (i64)jp * a.spo[0];
        xcol[t] = r + jp * a.R[0];
    }
    for (int y = 255 - (int)tid; y < L1; y += 256) {  // (the other end of the workgroup: L0 + L1 <= 256 is the common case)
        const int jp = a.R[1] == 1 ? y : (int)fdiv16((uint32_t)y, a.magicR[1]), r = y - jp * a.R[1];
        offd[y] = (i64)a.roff[1][r] + (i64)jp * a.spo[1];
    }
    __syncthreads();
    if constexpr (PAIR) {
        const int I1 = (L1 + 2) >> 1, I0 = (L0 + 2) >> 1;
        {   // phase 1: lanes along the pair slots of the input run
            const int dx = (int)fdiv16(256u, a.magicI[1]), di = 256 - dx * I1;
            int x = (int)fdiv16(tid, a.magicI[1]), i = (int)tid - x * I1;
            while (x < L0) {
                const int xc = xcol[x];
                if (xc < n0) {
                    const T* row = src + bs + offs[x];
                    const int p = (int)(((uintptr_t)row / sizeof(T)) & 1u);
                    const int y = 2 * i - p;
                    const bool v0 = y >= 0 && y < n1, v1 = y + 1 < n1;
                    if (v0 && v1) {
                        const FVec<T, 2> v = *reinterpret_cast<const FVec<T, 2>*>(row + y);
                        lds[y * PITCH + xc] = v.v[0];
                        lds[(y + 1) * PITCH + xc] = v.v[1];
                    } else if (v0) {
                        lds[y * PITCH + xc] = row[y];
                    } else if (v1) {
                        lds[(y + 1) * PITCH + xc] = row[y + 1];
                    }
                }
                x += dx;
                i += di;
                if (i >= I1) {
                    i -= I1;
                    ++x;
                }
            }
        }
        __syncthreads();
        {   // phase 2: lanes along the pair slots of the destination run
            const int dy = (int)fdiv16(256u, a.magicI[0]), di = 256 - dy * I0;
            int y = (int)fdiv16(tid, a.magicI[0]), i = (int)tid - y * I0;
            while (y < n1) {
                const i64 o = bd + offd[y];
                const int p = (int)(((uintptr_t)((const T*)a.ops.base[0] + o) / sizeof(T)) & 1u);
                const int x = 2 * i - p;
                const bool v0 = x >= 0 && x < n0, v1 = x + 1 < n0;
                if (v0 && v1) {
                    const T tv[2] = {lds[y * PITCH + x], lds[y * PITCH + x + 1]};
                    flat_emit<T, F, 2>(a.ops, a.nin, a.kt, anyconj, o + x, tv, f);
                } else if (v0) {
                    const T tv[1] = {lds[y * PITCH + x]};
                    flat_emit<T, F, 1>(a.ops, a.nin, a.kt, anyconj, o + x, tv, f);
                } else if (v1) {
                    const T tv[1] = {lds[y * PITCH + x + 1]};
                    flat_emit<T, F, 1>(a.ops, a.nin, a.kt, anyconj, o + x + 1, tv, f);
                }
                y += dy;
                i += di;
                if (i >= I0) {
                    i -= I0;
                    ++y;
                }
            }
        }
        return;
    }
    // phase 1: (x along the destination run) x (y along the input run), lanes along y; (x, y) advance by 256 positions per step
    {
        // four positions per lane and round, all four loads issued before the first LDS write (round 6: one load, its wait, one
        // write per round was a chain of dependent memory round trips -- 8 of them for a 31 x 65 tile); a position outside the tile
        // re-reads the tile's first element and drops it
        const int dx = (int)fdiv16(256u, a.magicL[1]), dy = 256 - dx * L1;
        int x = (int)fdiv16(tid, a.magicL[1]), y = (int)tid - x * L1;
        constexpr int UB = 4;
        const i64 safe = bs + offs[0];
        while (x < L0) {
            T v[UB];
            int at[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int xs = x < L0 ? x : 0;
                const int xc = xcol[xs];
                const bool ok = x < L0 && y < n1 && xc < n0;
                at[u] = ok ? y * PITCH + xc : -1;
                v[u] = src[ok ? bs + y + offs[xs] : safe];
                x += dx;
                y += dy;
                if (y >= L1) {
                    y -= L1;
                    ++x;
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (at[u] >= 0) lds[at[u]] = v[u];
        }
    }
    __syncthreads();
    // phase 2: lanes along x
    {
        const int dy = (int)fdiv16(256u, a.magicL[0]), dx = 256 - dy * L0;
        int y = (int)fdiv16(tid, a.magicL[0]), x = (int)tid - y * L0;
        while (y < n1) {
            if (x < n0) {
                const T tv[1] = {lds[y * PITCH + x]};
                flat_emit<T, F, 1>(a.ops, a.nin, a.kt, anyconj, bd + x + offd[y], tv, f);
            }
            y += dy;
            x += dx;
            if (x >= L0) {
                x -= L0;
                ++y;
            }
        }
    }
}

// ---- batched form (round 4) ----------------------------------------------------------------------------------------------
// FlatBPlan (smr_internal.h): blocks of P elements that are contiguous on BOTH sides, in different element orders, one behind the
// other along the batch dim -- batched transposes of small matrices.  A workgroup owns K consecutive blocks = K * P consecutive
// elements of either side: phase 1 streams them from the input into LDS (16-byte vectors when the chunk is aligned), phase 2 writes
// the destination's K * P consecutive elements, each fetched from LDS at [block * P + srcoff[r]].  Both sides move at streaming
// speed whatever the block's shape; (9,11,N) -> (11,9,N) ran at 1.4 TB/s in the two-sided form (99-element tiles, element-wise).
struct FlatBArgs {
    OpTab ops;
    int32_t P, K, nouter, conjv;
    uint32_t magicP, nchunk;              // floor(2^32 / P) + 1; chunks of K blocks along the batch dim
    i64 nb;                               // extent of the batch dim
    i64 sstr;                             // input stride of the batch dim (P: the input's blocks are adjacent too)
    i64 oext[MAXN], ostr[MAXN], ostr1[MAXN];  // the dims behind it: extents, destination strides, input strides
    uint16_t srcoff[FLATB_MAXP];
};

template <class T, class F>
SMR_DEV void flatb_body(const FlatBArgs a, F f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_flat[];
    T* lds = reinterpret_cast<T*>(smem_flat);
    uint16_t* tab = reinterpret_cast<uint16_t*>(smem_flat + (((size_t)a.K * a.P * sizeof(T) + 15) & ~(size_t)15));
    const uint32_t tid = threadIdx.x;
    uint32_t b = blockIdx.x;
    const uint32_t ch = b % a.nchunk;
    b /= a.nchunk;
    i64 base = (i64)ch * a.K * a.P, base1 = (i64)ch * a.K * a.sstr;
    for (int i = 0; i < a.nouter; ++i) {
        const uint32_t e = (uint32_t)a.oext[i];
        const uint32_t c = b % e;
        b /= e;
        base += (i64)c * a.ostr[i];
        base1 += (i64)c * a.ostr1[i];
    }
    const i64 left = a.nb - (i64)ch * a.K;
    const int n = (int)((left < a.K ? left : a.K) * a.P);  // elements of this chunk
    for (int r = (int)tid; r < a.P; r += 256) tab[r] = a.srcoff[r];
    const T* src = (const T*)a.ops.base[1] + base1;
    constexpr int V = (16 / (int)sizeof(T)) > 1 ? (16 / (int)sizeof(T)) : 1;
    const bool adjacent = a.sstr == a.P;  // the chunk is contiguous in the input as well
    const bool vec = V > 1 && (n % V) == 0 && ((uintptr_t)((T*)a.ops.base[0] + base) % 16) == 0;
    if (vec && adjacent && ((uintptr_t)src % 16) == 0) {
        for (int t = (int)tid * V; t < n; t += 256 * V) *reinterpret_cast<FVec<T, V>*>(lds + t) = *reinterpret_cast<const FVec<T, V>*>(src + t);
    } else if (adjacent) {
        for (int t = (int)tid; t < n; t += 256) lds[t] = src[t];
    } else {  // block by block: P consecutive elements each
        for (int t = (int)tid; t < n; t += 256) {
            const uint32_t blk = fdiv16((uint32_t)t, a.magicP), r = (uint32_t)t - blk * (uint32_t)a.P;
            lds[t] = src[(i64)blk * a.sstr + r];
        }
    }
    __syncthreads();
    const bool anyconj = a.conjv != 0;
    if (vec) {
        for (int t = (int)tid * V; t < n; t += 256 * V) {
            T tv[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const uint32_t blk = fdiv16((uint32_t)(t + e), a.magicP), r = (uint32_t)(t + e) - blk * (uint32_t)a.P;
                tv[e] = lds[blk * (uint32_t)a.P + tab[r]];
            }
            flat_emit<T, F, V>(a.ops, 1, 1, anyconj, base + t, tv, f);
        }
    } else {
        for (int t = (int)tid; t < n; t += 256) {
            const uint32_t blk = fdiv16((uint32_t)t, a.magicP), r = (uint32_t)t - blk * (uint32_t)a.P;
            const T tv[1] = {lds[blk * (uint32_t)a.P + tab[r]]};
            flat_emit<T, F, 1>(a.ops, 1, 1, anyconj, base + t, tv, f);
        }
    }
}

#ifndef SMR_JIT
template <class T, class F, int DIR, int VL, int VF>
__global__ void __launch_bounds__(256) k_flat_map(const FlatArgs a, F f) {
    flat_map_body<T, F, DIR, VL, VF>(a, f);
}
template <class T, class F, bool PAIR>
__global__ void __launch_bounds__(256) k_flat2_map(const Flat2Args a, F f) {
    flat2_body<T, F, PAIR>(a, f);
}

template <class T, class F>
__global__ void __launch_bounds__(256) k_flatb_map(const FlatBArgs a, F f) {
    flatb_body<T, F>(a, f);
}

template <class T, class F>
static int gob(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const FlatBPlan& fp = plan.flatb;
    FlatBArgs a;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    a.P = fp.P;
    a.K = fp.K;
    a.magicP = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)fp.P + 1);
    a.nb = c.dims[fp.g];
    a.nchunk = (uint32_t)((a.nb + fp.K - 1) / fp.K);
    i64 blocks = a.nchunk;
    for (int d = fp.g + 1; d < c.N; ++d) {
        a.oext[a.nouter] = c.dims[d];
        a.ostr[a.nouter] = c.strides[0][d];
        a.ostr1[a.nouter] = c.strides[1][d];
        blocks *= c.dims[d];
        ++a.nouter;
    }
    a.sstr = c.strides[1][fp.g];
    if (c.conj[0] || c.conj[1]) a.conjv = 1;
    std::memcpy(a.srcoff, fp.srcoff, sizeof(uint16_t) * (size_t)fp.P);
    if (blocks > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "flat plan: too many tiles");
    const size_t lds = (((size_t)fp.K * fp.P * sizeof(T) + 15) & ~(size_t)15) + (((size_t)fp.P * sizeof(uint16_t) + 15) & ~(size_t)15);
    const unsigned grid = (unsigned)blocks;
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "flat";
        l.tname = tname<T>();
        l.argtype = "smr::FlatBArgs";
        l.entry = std::string("smr::flatb_body<") + tname<T>() + ", smr::FJit>(a, smr::FJit{kc});";
        l.grid = grid;
        l.block = 256;
        l.lds = lds;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(plan.c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k_flatb_map<T, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
        }
        SMR_LAUNCH((k_flatb_map<T, F>), dim3(grid), dim3(256), lds, s, a, f);
        return check_launch("k_flatb_map");
    }
}

template <class T, class F>
static int go2(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const Flat2Plan& fp = plan.flat2;
    Flat2Args a;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    i64 blocks = 1;
    for (int t = 0; t < 2; ++t) {
        a.R[t] = fp.R[t];
        a.TP[t] = fp.TP[t];
        a.L[t] = fp.R[t] * fp.TP[t];
        a.dimp[t] = fp.p[t] >= 0 ? c.dims[fp.p[t]] : 1;
        a.spo[t] = fp.p[t] >= 0 ? c.strides[t == 0 ? fp.kt : 0][fp.p[t]] : 0;
        a.ntp[t] = (unsigned)((a.dimp[t] + fp.TP[t] - 1) / fp.TP[t]);
        a.magicR[t] = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)a.R[t] + 1);
        a.magicL[t] = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)a.L[t] + 1);
        a.magicI[t] = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)((a.L[t] + 2) / 2) + 1);
        for (int r = 0; r < fp.R[t]; ++r) a.roff[t][r] = fp.roff[t][r];
        blocks *= a.ntp[t];
    }
    a.nin = c.M - 1;
    a.kt = fp.kt;
    for (int k = 0; k < c.M; ++k)
        if (c.conj[k]) a.conjv = 1;
    for (int d = 0; d < c.N; ++d) {
        if (d == fp.p[0] || d == fp.p[1] || fp.ingroup[0][d] || fp.ingroup[1][d]) continue;
        a.oext[a.nouter] = c.dims[d];
        a.os0[a.nouter] = c.strides[0][d];
        a.os1[a.nouter] = c.strides[fp.kt][d];
        blocks *= c.dims[d];
        ++a.nouter;
    }
    if (blocks > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "flat plan: too many tiles");
    const size_t lds = (((size_t)a.L[1] * (size_t)(a.L[0] | 1) * sizeof(T) + 15) & ~(size_t)15) + (size_t)(a.L[0] + a.L[1]) * sizeof(i64) +
                       (size_t)a.L[0] * sizeof(int);
    a.shared = (fp.shared && fp.TP[0] > 1) ? 1 : 0;
    a.magicTP0 = fp.TP[0] > 1 ? (uint32_t)(((uint64_t)1 << 32) / (uint32_t)fp.TP[0] + 1) : 0u;
    const unsigned grid = (unsigned)blocks;
    // pairs: 4- / 8-byte elements, runs long enough to hold a few, the two-element slots of the tile within 16-bit divisions, and every
    // input that shares the destination's layout congruent with it modulo the pair size (it is read at the destination's offsets)
    // Measured (tools/flat2_pair_ab.py, profiles/r04_flat2_pair_ab.txt): 4-byte elements with both runs made of short leading dims gain 5-20 %
    // ((6,64,64,64,5) 22.4 -> 18.5 us, (40,50,60,36) 12.4 -> 10.8, (12,5000,30,10) 37.8 -> 32.8); 8-byte elements and cut leads (R = 1: every
    // other row of an odd extent starts misaligned) lose 0-18 % -- flat2_pair = 2 forces pairs wherever they are possible
    const int pmode = (int)options().flat2_pair;
    bool pair = pmode != 0 && sizeof(T) <= 8 && a.L[0] >= 8 && a.L[1] >= 8 &&
                (pmode >= 2 || (sizeof(T) == 4 && a.R[0] > 1 && a.R[1] > 1 && c.total * (i64)sizeof(T) >= ((i64)8 << 20)));
    for (int k = 1; k < c.M && pair; ++k)
        if (k != fp.kt && ((uintptr_t)a.ops.base[k] % (2 * sizeof(T))) != ((uintptr_t)a.ops.base[0] % (2 * sizeof(T)))) pair = false;
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "flat";
        l.tname = tname<T>();
        l.argtype = "smr::Flat2Args";
        l.entry = std::string("smr::flat2_body<") + tname<T>() + ", smr::FJit, " + (pair ? "true" : "false") + ">(a, smr::FJit{kc});";
        l.grid = grid;
        l.block = 256;
        l.lds = lds;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(plan.c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        if (pair)
            SMR_LAUNCH((k_flat2_map<T, F, true>), dim3(grid), dim3(256), lds, s, a, f);
        else
            SMR_LAUNCH((k_flat2_map<T, F, false>), dim3(grid), dim3(256), lds, s, a, f);
        return check_launch("k_flat2_map");
    }
}

template <class T, class F, int DIR, int VL, int VF>
static int go3(const Plan& plan, hipStream_t s, F f, const FlatArgs& a, size_t lds, unsigned grid) {
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "flat";
        l.tname = tname<T>();
        l.argtype = "smr::FlatArgs";
        l.entry = std::string("smr::flat_map_body<") + tname<T>() + ", smr::FJit, " + std::to_string(DIR) + ", " + std::to_string(VL) + ", " +
                  std::to_string(VF) + ">(a, smr::FJit{kc});";
        l.grid = grid;
        l.block = 256;
        l.lds = lds;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(plan.c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        SMR_LAUNCH((k_flat_map<T, F, DIR, VL, VF>), dim3(grid), dim3(256), lds, s, a, f);
        return check_launch("k_flat_map");
    }
}

template <class T, class F>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const FlatPlan& fp = plan.flat;
    constexpr int VMAX = (16 / (int)sizeof(T)) > 1 ? (16 / (int)sizeof(T)) : 1;
    // tile of the line side: about 1 << tqlog elements, evened out over the extent (100 -> 4 tiles of 26, not 3 of 32 + 1 of 4),
    // a multiple of the vector width
    int TQ = 1 << fp.tqlog;
    {
        const i64 nt = (c.dims[fp.q] + TQ - 1) / TQ;
        TQ = (int)((c.dims[fp.q] + nt - 1) / nt);
        TQ = (TQ + VMAX - 1) / VMAX * VMAX;
    }
    FlatArgs a;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    a.N = c.N;
    a.dir = fp.dir;
    a.R = fp.R;
    a.TPlog = fp.tplog;
    a.L = fp.R << fp.tplog;
    a.TQ = TQ;
    a.p = fp.p;
    a.q = fp.q;
    const int kf = fp.dir == 0 ? 0 : fp.kt, kl = fp.dir == 0 ? fp.kt : 0;  // operand index of the flat / line side
    a.nin = c.M - 1;
    a.kt = fp.kt;
    a.dimp = fp.p >= 0 ? c.dims[fp.p] : 1;
    a.dimq = c.dims[fp.q];
    a.sfq = c.strides[kf][fp.q];
    a.sfp = fp.R;
    a.slp = fp.p >= 0 ? c.strides[kl][fp.p] : 0;
    a.ntp = (unsigned)((a.dimp + ((i64)1 << fp.tplog) - 1) >> fp.tplog);
    a.ntq = (unsigned)((a.dimq + TQ - 1) / TQ);
    a.conjv = 0;
    for (int k = 0; k < c.M; ++k)
        if (c.conj[k]) a.conjv = 1;
    a.fuse = fp.fuse ? 1 : 0;
    a.lshare = fp.lshare ? 1 : 0;
    for (int r = 0; r < fp.R; ++r) a.roff[r] = fp.roff[r];
    i64 blocks = (i64)a.ntp * a.ntq;
    for (int d = 0; d < c.N; ++d) {
        if (d == fp.p || d == fp.q || fp.ingroup[d]) continue;
        a.odim[a.nouter] = d;
        a.oext[a.nouter] = c.dims[d];
        a.osf[a.nouter] = c.strides[kf][d];
        a.osl[a.nouter] = c.strides[kl][d];
        blocks *= c.dims[d];
        ++a.nouter;
    }
    if (blocks > 0x7fffffffLL) return set_error(SMR_EUNSUPPORTED, "flat plan: too many tiles");
    // vector widths: 16 bytes where extents, strides and base addresses allow
    int vl = VMAX, vf = VMAX;
    auto aligned = [&](int k) { return (((uintptr_t)a.ops.base[k]) % 16) == 0; };
    if (fp.lshare) {
        // line-side rows are runs of R * (valid columns) elements starting at multiples of R * TQ
        if ((fp.R * c.dims[fp.q]) % VMAX || !aligned(kl)) vl = 1;
        for (int d = 0; d < c.N; ++d)
            if (d != fp.q && !fp.ingroup[d] && c.strides[kl][d] % VMAX) vl = 1;
    } else {
        if (c.dims[fp.q] % VMAX || !aligned(kl)) vl = 1;
        for (int d = 0; d < c.N; ++d)
            if (d != fp.q && c.strides[kl][d] % VMAX) vl = 1;
    }
    if (fp.fuse) {
        // the run is R * (valid columns): whole vectors when R * extent(q) is a vector multiple (tile origins are q0 * R)
        if ((fp.R * c.dims[fp.q]) % VMAX || !aligned(kf)) vf = 1;
        for (int d = 0; d < c.N; ++d)
            if (!fp.ingroup[d] && d != fp.q && c.strides[kf][d] % VMAX) vf = 1;
    } else {
        if (a.L % VMAX || !aligned(kf)) vf = 1;
        if ((fp.R * (fp.p >= 0 ? c.dims[fp.p] : 1)) % VMAX) vf = 1;  // a ragged last tile must end on a vector boundary
        for (int d = 0; d < c.N; ++d)
            if (!fp.ingroup[d] && d != fp.p && c.strides[kf][d] % VMAX) vf = 1;
    }
    // inputs that share the destination's layout are read with the destination's vector width
    for (int k = 1; k < c.M; ++k)
        if (k != fp.kt && !aligned(k)) (fp.dir == 0 ? vf : vl) = 1;
    a.magicR = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)fp.R + 1);
    a.magicRow = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)((fp.R * TQ) / vl) + 1);
    a.magicXV = (TQ / vl) > 1 ? (uint32_t)(((uint64_t)1 << 32) / (uint32_t)(TQ / vl) + 1) : 0u;
    a.magicLv = (a.L / vf) > 1 ? (uint32_t)(((uint64_t)1 << 32) / (uint32_t)(a.L / vf) + 1) : 0u;  // unused in the fused form
    constexpr int PAD = (16 / (int)sizeof(T)) > 0 ? (16 / (int)sizeof(T)) : 1;
    const size_t lds = (size_t)a.L * (TQ + PAD) * sizeof(T);
    const unsigned grid = (unsigned)blocks;
    auto run = [&](auto DIRc) -> int {
        constexpr int DIR = decltype(DIRc)::value;
        if constexpr (VMAX > 1) {
            if (vl > 1 && vf > 1) return go3<T, F, DIR, VMAX, VMAX>(plan, s, f, a, lds, grid);
            if (vl > 1) return go3<T, F, DIR, VMAX, 1>(plan, s, f, a, lds, grid);
            if (vf > 1) return go3<T, F, DIR, 1, VMAX>(plan, s, f, a, lds, grid);
        }
        return go3<T, F, DIR, 1, 1>(plan, s, f, a, lds, grid);
    };
    return fp.dir == 0 ? run(std::integral_constant<int, 0>{}) : run(std::integral_constant<int, 1>{});
}

template <>
int launch_flat_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    const bool two = plan.flat2.on;
    if (plan.flatb.on) {
        if (c.bitcopy) {
#if SMR_CT == SMR_F32
            switch (c.esize[0]) {
                case 4: return gob<float, FIdent<float>>(plan, bases, s, FIdent<float>{});
                case 8: return gob<double, FIdent<double>>(plan, bases, s, FIdent<double>{});
                case 16: return gob<c64, FIdent<c64>>(plan, bases, s, FIdent<c64>{});
                default: return set_error(SMR_EINVAL, "flat plan: 1- / 2-byte moves take the generic family");
            }
#else
            return set_error(SMR_EINVAL, "bitcopy is dispatched through the f32 object");
#endif
        }
        return with_functor<T>(c, fbit(FK_IDENT) | fbit(FK_SCALE), [&](auto f) { return gob<T, decltype(f)>(plan, bases, s, f); });
    }
    if (c.bitcopy) {
#if SMR_CT == SMR_F32
        switch (c.esize[0]) {
            case 4: return two ? go2<float, FIdent<float>>(plan, bases, s, FIdent<float>{}) : go<float, FIdent<float>>(plan, bases, s, FIdent<float>{});
            case 8: return two ? go2<double, FIdent<double>>(plan, bases, s, FIdent<double>{}) : go<double, FIdent<double>>(plan, bases, s, FIdent<double>{});
            case 16: return two ? go2<c64, FIdent<c64>>(plan, bases, s, FIdent<c64>{}) : go<c64, FIdent<c64>>(plan, bases, s, FIdent<c64>{});
            default: return set_error(SMR_EINVAL, "flat plan: 1- / 2-byte moves take the generic family");
        }
#else
        return set_error(SMR_EINVAL, "bitcopy is dispatched through the f32 object");
#endif
    }
    const unsigned mask = fbit(FK_IDENT) | fbit(FK_SCALE) | fbit(FK_ADD2) | fbit(FK_AXPY) | fbit(FK_AXPBY);
    if (two) return with_functor<T>(c, mask, [&](auto f) { return go2<T, decltype(f)>(plan, bases, s, f); });
    return with_functor<T>(c, mask, [&](auto f) { return go<T, decltype(f)>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
