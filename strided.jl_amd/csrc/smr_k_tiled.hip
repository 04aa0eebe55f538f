// smr_k_tiled.hip -- family TILED: fused N-ary map whose operands have DIFFERENT unit-stride
// axes (permutedims!, adjoint!, B .= (A .+ A')./2, the 4-way permuted sum).
//
// MI355X design (not the reference's L1-blocked loop nest, src/mapreduce.jl:385-401):
//   * a workgroup owns one N-d tile whose extents are powers of two and which is long enough
//     along EVERY operand's unit-stride axis (1024 elements on 256 lanes; 4096 on 1024 lanes
//     for big problems with three or more distinct unit axes: always 4 elements per lane);
//   * phase A: every global load of the tile is issued up front: transposed inputs in THEIR
//     OWN stride order (consecutive lanes walk that input's unit-stride axis -> coalesced,
//     16 bytes per lane), direct inputs in destination order;
//   * phase B: transposed inputs are scattered into an LDS tile laid out in DESTINATION
//     order, XOR-swizzled with masks searched on the host per plan (choose_swizzle: GF(2) rank
//     test of every staged operand's write pattern and of the read pattern) so that neither the
//     strided writes nor the linear reads of a lane group collide on a bank;
//   * phase C: linear (conflict-free) LDS reads, f applied in registers, 16-byte coalesced
//     stores.
// Variants: the classic form runs one tile per workgroup (tiled_map_body; MODE selects at compile
// time which rarely needed features it carries); long work lists use the persistent,
// software-pipelined form (tiled_map_pipe_body).  Tiles are visited in the planner's order
// (smr_plan.cpp: plan_tile_order -- orbit-major when several inputs are permuted views of one
// buffer, so that they meet in one XCD's L2).
//
// Index arithmetic.  Tile extents are powers of two, so the position e of an element inside a
// tile (in any operand's enumeration order) is a bit string, and both its global byte offset
// and its swizzled LDS index are ADD/XOR-linear in those bits.  A lane owns NREP vectors of V
// consecutive elements: e = ((r*T + tid) << vlog) + h.  The host precomputes
//     lane table  (device memory, one row per lane and operand): byte offset + LDS index of
//                 the tid bits -- ONE 8-byte vector load per operand, issued first thing;
//     Gr[r], Lr[r], Lh[h] (kernel arguments): contribution of repeat index r / sub-element h,
// so the kernel contains no index arithmetic beyond one add (address) and one xor (LDS) per
// access.  Tile coordinates come from multiply-shift division of the workgroup id.  The kernel
// arguments are built once per plan and cached (only the operand addresses are patched per call).
// Measured history (32^4 f64, permutedims! / 4-way sum, us per launch): generic per-dim decode
// with dependent kernarg loads 11.5 / 28.5 (~2000 scalar instructions per wave: bound by the
// CU's single scalar unit) -> per-bit tables 6.0 / 14.6 -> 16-B vectors + repeat tables
// 4.4 / 9.9 -> all loads first, branch-free variants 4.1 / 8.9 -> lane tables 3.4 / 8.4 ->
// 4096-element tiles on 1024 lanes, searched swizzle, orbit-major order 3.4 / 6.85 -> compile-time
// modes 3.4 / 6.6 (a plain 16 MiB copy: 3.2; a 5-stream contiguous add: 5.4).
#ifndef SMR_JIT
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif
// 1: compute the per-lane rows (bit slices of the lane id, fold swizzle) instead of loading them from the
// device table.  Measured on MI355X (tools/perm_ab.py, round 2): SLOWER where it was meant to help -- permutedims!
// 32^4 f64 3.57 vs 3.32 us, f32 2.79 vs 2.57 us, transpose 8192^2 225 vs 209 us (only permutedims! 128^4 f64
// gained, 866 vs 935 us) -- the table row is one coalesced 8-byte load that overlaps the kernel-argument loads,
// the arithmetic needs ~30 more scalar argument words and ~15 VALU per operand.  Kept as a build-time experiment.
#ifndef SMR_TILED_BITS
#define SMR_TILED_BITS 0
#endif
#ifndef SMR_TILED_NTL
#define SMR_TILED_NTL 0
#endif

namespace smr {

constexpr int MAXT = 5;
constexpr int NG = 4;   // grid dims decoded branch-free; further ones in a (rare) loop
constexpr int EPL = 4;  // elements per lane (tile elements / workgroup size)
constexpr int NORD16 = 512;  // tile-order entries that fit into the kernel arguments

template <bool WIDE> struct off_t_of { typedef uint32_t type; };
template <> struct off_t_of<true> { typedef i64 type; };

// one row of the per-lane table
template <bool WIDE> struct LaneRow { uint32_t g; uint32_t l; };
template <> struct alignas(16) LaneRow<true> { i64 g; uint32_t l; uint32_t pad; };

// Wave-uniform description of one operand (a few wide scalar loads).
template <bool WIDE>
struct OpDesc {
    typedef typename off_t_of<WIDE>::type O;
    void* base;            // element offset already applied
    uint32_t tstep32[NG];  // byte offset of one tile step along grid dim g (when base32)
    O Gr[EPL];             // byte offset of repeat index r (own order if staged, else dst order)
    uint32_t Lr[EPL];      // staged: swizzled LDS index of repeat index r (own order)
    uint32_t Lh[EPL];      // staged: ... of sub-element h
    int32_t dtype, conj;
};

template <bool WIDE>
struct TiledArgs {
    // header + tile decode first: they arrive with the first batch of scalar loads
    int32_t M, ng, tilelog, nstaged, base32, nt, ordmode, nwork;  // nwork: entries of the work list (persistent form)
    int32_t nts;           // nts: non-temporal stores (small destinations: see Options::nt_store)
    uint32_t blk0;         // first workgroup of a block-range slice (smr_seq.cpp); 0 for a whole launch.  One-shot form only
    int32_t pad_[2];
    int32_t staged[MAXM];  // [1 + i]: LDS slot of input i or -1 ([0] unused)
    uint32_t ntiles[MAXN], div_m[MAXN], div_s[MAXN], last_ragged[MAXN];
    const LaneRow<WIDE>* lanetab;  // [(operand k) * T + tid], k = 0 destination
    const uint32_t* ordtab;        // ordmode 2: tile executed by workgroup b (0xffffffff = none)
    OpDesc<WIDE> op[MAXM];         // [0] destination, [1 + i] input i
    uint32_t Lrd[EPL], Lhd[EPL];   // destination-order LDS index of repeat r / sub-element h
    i64 tstep[MAXM][MAXN];         // 64-bit tile steps (only read when !base32)
    // edge tiles only
    uint32_t gflip, gflip_pad;        // bit g: grid coordinate g counts DOWN (ragged dims: their last, partly filled tiles start first)
    int32_t ej_g[MAXT];               // per tiled dim j: its grid slot (-1: none), the index of its LAST tile and the valid elements in
    uint32_t ej_ntm1[MAXT], ej_last[MAXT];  // that tile (round 6: one batch of scalar loads instead of a dependent chain through tgrid / gdims / glog)
    int32_t tgrid[MAXT], tlog[MAXT];  // grid dim / log2 extent of tiled dim j
    int32_t esh[MAXM][MAXT];          // bit position of tiled dim j in operand k's enumeration
    i64 gdims[MAXN];
    int32_t glog[MAXN];
    uint32_t ord16[NORD16 / 2];  // ordmode 1: the same as 16-bit entries inside the kernel arguments
    // BITS form (narrow offsets, variants 0 / 1 / 2): the per-lane table rows are COMPUTED -- bit slices of the lane
    // id -- instead of loaded, which takes one dependent memory round trip out of the prologue
    int32_t bpos[MAXM][MAXT], blen[MAXM][MAXT], blsh[MAXM][MAXT];  // slice p of operand k's enumeration (blen 0 = unused)
    uint32_t bstr[MAXM][MAXT];                                     // its byte stride
    uint32_t fs1, fs2, fmask, fpad;                                // fold swizzle l ^ (((l >> fs1) ^ (l >> fs2)) & fmask)
};

SMR_DEV uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t s) { return (__umulhi(m, n) + n) >> s; }

template <class T, int V>
struct alignas(sizeof(T) * V) TVec {
    T v[V];
};

// global-memory view of a lane's vector: element alignment only (round 6: odd extents and odd row strides keep 16-byte accesses;
// the hardware takes dwordx2 / dwordx4 at any dword address, as in smr_k_stream.hip)
template <class T, int V>
struct alignas(sizeof(T)) TUVec {
    T v[V];
};

template <class T, bool MIXED>
SMR_DEV T load_at(const char* p, int dtype, int conj) {
    T v;
    if constexpr (MIXED)
        v = ld_as<T>(p, 0, dtype);
    else
        v = *reinterpret_cast<const T*>(p);
    if constexpr (tr<T>::cx) {
        if (conj) v = cj(v);
    }
    return v;
}
template <class T, bool MIXED>
SMR_DEV void store_at(char* p, int dtype, int conj, T v) {
    if constexpr (tr<T>::cx) {
        if (conj) v = cj(v);
    }
    if constexpr (MIXED)
        st_as<T>(p, 0, dtype, v);
    else
        *reinterpret_cast<T*>(p) = v;
}

// V > 1 implies !MIXED && !WIDE and every direct operand unit-stride along dim 0 (launcher).
// MODE bit 0: bounds checks for ragged tiles, bit 1: tile-order lookup, bit 2: general tile origins
// (more than 4 grid dims, or origins beyond 4 GiB / negative steps).  The plain variant (0) carries
// none of it: code size and every extra scalar wait are part of the latency of a ~3.5 us launch
// (measured: permutedims! 3.48 -> 3.43 us, 4-way sum 6.84 -> 6.59 us for dropping bit 2 alone).
// Bit 3: partial vectors (the extent of a vector axis is not a multiple of V: the last vector of a row is moved element by element).
// Instantiated: 0 (plain), 1 (plain + bounds checks on edge tiles), 9 (1 + partial vectors), 2 (orbit order), 7 (everything).
template <class T, class F, bool MIXED, bool WIDE, int V, int MODE, int THRLOG>
SMR_DEV void tiled_map_body(const TiledArgs<WIDE> a, F f) {
    constexpr bool EDGE = (MODE & 1) != 0;
    constexpr bool ORD = (MODE & 2) != 0;
    constexpr bool GENORG = (MODE & 4) != 0;
    constexpr bool PV = (MODE & 8) != 0 || MODE == 7;  // partial vectors at the end of a row (extent of a vector axis not a multiple of V)
    typedef typename off_t_of<WIDE>::type O;
    typedef TVec<T, V> VT;
    typedef TUVec<T, V> GT;  // the same vector in global memory
    constexpr int VLOG = (V == 1) ? 0 : (V == 2 ? 1 : 2);
    constexpr int NREP = EPL / V;
    constexpr int NT = 1 << THRLOG;
    constexpr int NIN_STATIC = F::NIN;
    constexpr int NINMAX = (NIN_STATIC >= 0) ? NIN_STATIC : MAXIN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int nin = (NIN_STATIC >= 0) ? NIN_STATIC : a.M - 1;
    const uint32_t tid = threadIdx.x;

    // ---- per-lane rows (byte offset + swizzled LDS index of the lane's first element, per operand) ----
    // tables in device memory (first memory instructions of the kernel) or, BITS, bit slices of the lane id
    constexpr bool BITS = SMR_TILED_BITS && !WIDE && (MODE & 4) == 0;
    LaneRow<WIDE> row[NINMAX + 1];
#pragma unroll
    for (int k = 0; k <= NINMAX; ++k) {
        row[k].g = 0;
        row[k].l = 0;
        if constexpr (BITS) {
            if (NIN_STATIC >= 0 || k <= nin) {
                const uint32_t e0 = tid << VLOG;
                uint32_t g = 0, idx = 0;
#pragma unroll
                for (int p = 0; p < MAXT; ++p) {
                    const uint32_t c = __builtin_amdgcn_ubfe(e0, (uint32_t)a.bpos[k][p], (uint32_t)a.blen[k][p]);
                    g += c * a.bstr[k][p];
                    idx |= c << a.blsh[k][p];
                }
                row[k].g = g;
                row[k].l = idx ^ (((idx >> a.fs1) ^ (idx >> a.fs2)) & a.fmask);
            }
        } else {
            if (k <= nin) row[k] = a.lanetab[k * NT + tid];
        }
    }

    // ---- which tile ---------------------------------------------------------------------------------
    uint32_t b = blockIdx.x + a.blk0;
    if constexpr (ORD) {
        // locality-aware tile order (smr_plan.cpp: plan_tile_order).  The in-kernarg lookup is
        // issued unconditionally so that it travels with the first batch of scalar loads.
        const uint32_t pair = a.ord16[(b & (NORD16 - 1)) >> 1];
        if (a.ordmode == 1) {
            const uint32_t v = (b & 1u) ? (pair >> 16) : (pair & 0xffffu);
            b = (v == 0xffffu) ? 0xffffffffu : v;
        } else if (a.ordmode == 2) {
            b = a.ordtab[b];
        }
        if (b == 0xffffffffu) return;  // padding workgroup (whole workgroup: no barrier is skipped)
    }
    uint32_t tc[MAXN];
#pragma unroll
    for (int g = 0; g < NG; ++g) {  // unused grid dims are padded with ntiles = 1
        const uint32_t q = fastdiv(b, a.div_m[g], a.div_s[g]);
        tc[g] = b - q * a.ntiles[g];
        b = q;
    }
    if constexpr (EDGE && !GENORG) {
        // ragged grid dims run slowest and backwards: the workgroups of the partly filled last tiles -- the ones with bounds code on
        // their path -- are the FIRST to start, and their longer lives end with everybody else's instead of after them
        const uint32_t fl = a.gflip;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if ((fl >> g) & 1u) tc[g] = a.ntiles[g] - 1u - tc[g];
    }
#pragma unroll
    for (int g = NG; g < MAXN; ++g) tc[g] = 0;
    if (GENORG && a.ng > NG) {
#pragma unroll
        for (int g = NG; g < MAXN; ++g) {
            const uint32_t q = fastdiv(b, a.div_m[g], a.div_s[g]);
            tc[g] = b - q * a.ntiles[g];
            b = q;
        }
    }
    // EDGE = false is instantiated for problems without any ragged dim: no bounds code at all
    // (code size is part of the latency of a ~4 us launch)
    bool edge = false;
    uint32_t lim[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) lim[j] = 0x7fffffffu;
    // Everything the bounds checks need from the kernel arguments is fetched HERE, unconditionally, with the first batch of scalar
    // loads (round 6).  Fetched where it was used -- inside `if (edge)` -- it was a chain of ~15 dependent scalar-memory round trips
    // in the workgroups of a ragged last tile, 1-2 us each: they finished that much after everybody else and set the launch's span
    // (transposes of 7200 x 100 Float64: 4.4 us against 3.0 us for 7200 x 128, with a quarter of the workgroups nearly empty).
    constexpr int KP = (NINMAX < 3 ? NINMAX : 3) + 1;  // operands whose enumeration shifts are pinned (destination + up to 3 inputs)
    int gj[MAXT];
    uint32_t ntm1[MAXT], lastn[MAXT], tlg[MAXT], eshp[KP][MAXT];
    if constexpr (EDGE) {
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            gj[j] = __builtin_amdgcn_readfirstlane(a.ej_g[j]);
            ntm1[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ej_ntm1[j]);
            lastn[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ej_last[j]);
            tlg[j] = (uint32_t)__builtin_amdgcn_readfirstlane(a.tlog[j]);
            asm volatile("" : "+s"(gj[j]), "+s"(ntm1[j]), "+s"(lastn[j]), "+s"(tlg[j]));
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                eshp[k][j] = (uint32_t)__builtin_amdgcn_readfirstlane(a.esh[k][j]);
                asm volatile("" : "+s"(eshp[k][j]));
            }
        }
        uint32_t emin = 0xffffffffu;  // 0 iff some grid coordinate sits on a ragged last tile
#pragma unroll
        for (int g = 0; g < MAXN; ++g)
            if (g < NG || (GENORG && a.ng > NG)) emin = min(emin, tc[g] ^ a.last_ragged[g]);
        edge = emin == 0;
        if (edge) {
#pragma unroll
            for (int j = 0; j < MAXT; ++j) {
                uint32_t t = 0xffffffffu;
#pragma unroll
                for (int gg = 0; gg < MAXN; ++gg)
                    if (gg == gj[j]) t = tc[gg];
                if (gj[j] >= 0 && t == ntm1[j]) lim[j] = lastn[j];
            }
        }
    }
    const int ntd = a.nt;
    // how many of the V elements e .. e + V - 1 of operand k's enumeration lie inside the array (they differ in the coordinate at
    // bit 0): V or 0, or -- extent of the vector axis not a multiple of V -- something between in the last vector of a row
    auto in_count = [&](int k, uint32_t e) -> uint32_t {
        uint32_t cnt = V;
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
            if (j < ntd) {
                uint32_t sh = 0, tl = 0;
                if constexpr (EDGE) {
                    tl = tlg[j];
                    sh = (uint32_t)a.esh[k][j];
#pragma unroll
                    for (int kk = 0; kk < KP; ++kk)
                        if (kk == k) sh = eshp[kk][j];
                }
                const uint32_t cj = (e >> sh) & ((1u << tl) - 1u);
                if (cj >= lim[j]) cnt = 0;
                else if (PV && V > 1 && sh == 0) cnt = min(cnt, lim[j] - cj);
            }
        return cnt;
    };
    auto tile_base = [&](int k) -> char* {
        if (!GENORG || a.base32) {  // every tile origin of every operand is below 4 GiB, steps non-negative
            uint32_t o = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) o += tc[g] * a.op[k].tstep32[g];
            return (char*)a.op[k].base + o;
        }
        i64 o = 0;
#pragma unroll
        for (int g = 0; g < MAXN; ++g)
            if (g < NG || a.ng > NG) o += (i64)tc[g] * a.tstep[k][g];
        return (char*)a.op[k].base + o;
    };

    // The body below exists twice in the bounds-checking variants (round 6): a workgroup of whole tiles runs the copy WITHOUT any
    // bounds code (EDG = false) -- the ~90 extra instructions in front of its loads were 0.3 us per launch on whole tiles --, a
    // workgroup on a ragged last tile the checked one.  `edge` is uniform over the workgroup: both copies keep their barrier.
    auto body = [&](auto edge_c) {
        constexpr bool EDG = EDGE && decltype(edge_c)::value;
        // ---- phase A: issue EVERY global load of the tile before anything waits ------------------------
        VT x[NINMAX > 0 ? NINMAX : 1][NREP];
        uint32_t cnl[NINMAX > 0 ? NINMAX : 1][NREP];  // valid elements of input i's repeat r in ITS order (bounds-checking vector variants)
        uint32_t cntd[NREP];  // valid elements of repeat r in destination order (edge tiles: 0 .. V)
    #pragma unroll
        for (int r = 0; r < NREP; ++r) {
            cntd[r] = V;
            if constexpr (EDG) cntd[r] = in_count(0, (((uint32_t)r << THRLOG) | tid) << VLOG);
        }
    #pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
    #pragma unroll
            for (int r = 0; r < NREP; ++r)
    #pragma unroll
                for (int h = 0; h < V; ++h) x[i][r].v[h] = T{};
            if (i < nin) {
                const OpDesc<WIDE>& d = a.op[i + 1];
                const char* bp = tile_base(i + 1);
                const bool stg = a.staged[i + 1] >= 0;
    #pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    uint32_t cn = cntd[r];
                    if (EDG && stg) cn = in_count(i + 1, (((uint32_t)r << THRLOG) | tid) << VLOG);
                    if constexpr (EDG && V > 1) {
                        // Bounds-checking vector variant: the load itself is UNCONDITIONAL -- a lane outside the array reads the operand's
                        // first vector (always there: the launcher requires an extent of at least V along the vector axis) and drops it.
                        // Guarded by a branch per load, the compiler put an s_waitcnt vmcnt(0) in front of every one of them: a tile's
                        // loads went out one memory round trip after the other, in EVERY workgroup of a ragged problem (+0.5 us on whole
                        // tiles, profiles/r06_ragged_tiles.txt).  Partial vectors are patched up below, after all loads have been issued.
                        const bool full = cn == (uint32_t)V;
                        const char* p = full ? bp + (O)(row[i + 1].g + d.Gr[r]) : (const char*)d.base;
                        const GT gv = *reinterpret_cast<const GT*>(p);
    #pragma unroll
                        for (int h = 0; h < V; ++h) x[i][r].v[h] = gv.v[h];  // (nothing may USE the value here: the next load must leave first;
                        cnl[i][r] = cn;                                       //  what a lane outside the array holds is never stored)
                    } else if (cn) {
                        const char* p = bp + (O)(row[i + 1].g + d.Gr[r]);
                        if constexpr (V == 1) {
                            x[i][r].v[0] = load_at<T, MIXED>(p, d.dtype, d.conj);
                        } else {
    #if SMR_TILED_NTL == 2  // experiment (A/B build): system-scope loads (sc0 sc1: served below the L2, which keeps its contents)
                            {
                                constexpr int NQ = (int)(sizeof(VT) / 8);
                                uint64_t qw[NQ > 0 ? NQ : 1];
    #pragma unroll
                                for (int w = 0; w < NQ; ++w)
                                    qw[w] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __builtin_memcpy(&x[i][r], qw, sizeof(VT));
                            }
    #elif SMR_TILED_NTL  // experiment (A/B build): non-temporal loads
                            x[i][r] = load_vec_ct<true, VT>(p);
    #else
                            {
                                const GT gv = *reinterpret_cast<const GT*>(p);
    #pragma unroll
                                for (int h = 0; h < V; ++h) x[i][r].v[h] = gv.v[h];
                            }
    #endif
                            if constexpr (tr<T>::cx) {
                                if (d.conj) {
    #pragma unroll
                                    for (int h = 0; h < V; ++h) x[i][r].v[h] = cj(x[i][r].v[h]);
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (EDG && V > 1 && tr<T>::cx) {  // conjugated inputs of the bounds-checking vector variants: behind all loads
    #pragma unroll
            for (int i = 0; i < NINMAX; ++i)
                if (i < nin && a.op[i + 1].conj) {
    #pragma unroll
                    for (int r = 0; r < NREP; ++r)
    #pragma unroll
                        for (int h = 0; h < V; ++h) x[i][r].v[h] = cj(x[i][r].v[h]);
                }
        }
        if constexpr (EDG && PV && V > 1) {  // the partial vector at the end of a row: element by element, behind the vector loads
    #pragma unroll
            for (int i = 0; i < NINMAX; ++i)
                if (i < nin) {
                    const OpDesc<WIDE>& d = a.op[i + 1];
                    const char* bp = tile_base(i + 1);
    #pragma unroll
                    for (int r = 0; r < NREP; ++r)
                        if (cnl[i][r] > 0 && cnl[i][r] < (uint32_t)V) {
                            const char* p = bp + (O)(row[i + 1].g + d.Gr[r]);
    #pragma unroll
                            for (int h = 0; h < V; ++h)
                                if ((uint32_t)h < cnl[i][r]) x[i][r].v[h] = load_at<T, MIXED>(p + h * sizeof(T), d.dtype, d.conj);
                        }
                }
        }
        char* bp0 = tile_base(0);

        // ---- phase B: staged inputs -> LDS, destination order, XOR-swizzled -----------------------------
    #pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
            if (i < nin && a.staged[i + 1] >= 0) {
                const OpDesc<WIDE>& d = a.op[i + 1];
                T* L = lds + ((size_t)a.staged[i + 1] << a.tilelog);
    #pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    bool ok = true;
                    if constexpr (EDG && !(V > 1)) {  // (the vector variants park every lane's vector: a lane outside the array owns its own LDS slots)
                        ok = in_count(i + 1, (((uint32_t)r << THRLOG) | tid) << VLOG) > 0;
                    }
                    if (ok) {
    #pragma unroll
                        for (int h = 0; h < V; ++h) L[row[i + 1].l ^ d.Lr[r] ^ d.Lh[h]] = x[i][r].v[h];
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase C: read back in destination order, apply f, store ---------------------------------------
    #pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
            if (i < nin && a.staged[i + 1] >= 0) {
                const T* L = lds + ((size_t)a.staged[i + 1] << a.tilelog);
    #pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    if ((EDG && V > 1) || cntd[r]) {
    #pragma unroll
                        for (int h = 0; h < V; ++h) x[i][r].v[h] = L[row[0].l ^ a.Lrd[r] ^ a.Lhd[h]];
                    }
                }
            }
        }
        VT out[NREP];
    #pragma unroll
        for (int r = 0; r < NREP; ++r) {
            if ((EDG && V > 1) || cntd[r]) {
    #pragma unroll
                for (int h = 0; h < V; ++h) {
                    T arg[MAXIN];
    #pragma unroll
                    for (int i = 0; i < MAXIN; ++i) {
                        arg[i] = T{};
                        if (i < NINMAX) arg[i] = x[i < NINMAX ? i : 0][r].v[h];
                    }
                    out[r].v[h] = f(arg);
                }
                if constexpr (V == 1) {
                    store_at<T, MIXED>(bp0 + (O)(row[0].g + a.op[0].Gr[r]), a.op[0].dtype, a.op[0].conj, out[r].v[0]);
                } else {
                    if constexpr (tr<T>::cx) {
                        if (a.op[0].conj) {
    #pragma unroll
                            for (int h = 0; h < V; ++h) out[r].v[h] = cj(out[r].v[h]);
                        }
                    }
                }
            }
        }
        if constexpr (V > 1) {
            // one wave-uniform branch around all vector stores (see smr_device.h:store_vec)
            auto put = [&](auto NT) {
    #pragma unroll
                for (int r = 0; r < NREP; ++r)
                    if (cntd[r] == (uint32_t)V) {
                        GT gv;
    #pragma unroll
                        for (int h = 0; h < V; ++h) gv.v[h] = out[r].v[h];
                        store_vec_ct<decltype(NT)::value, GT>(bp0 + (O)(row[0].g + a.op[0].Gr[r]), gv);
                    }
            };
            // the partial vector at the end of a row (edge tiles, extent of the vector axis not a multiple of V): element by element
            auto put_partial_vecs = [&](auto WT) {
                if constexpr (EDG && PV) {
    #pragma unroll
                    for (int r = 0; r < NREP; ++r)
                        if (cntd[r] > 0 && cntd[r] < (uint32_t)V) {
    #pragma unroll
                            for (int h = 0; h < V; ++h)
                                if ((uint32_t)h < cntd[r]) {
                                    char* p = bp0 + (O)(row[0].g + a.op[0].Gr[r]) + h * sizeof(T);
                                    if constexpr (decltype(WT)::value) {
                                        TVec<T, 1> one;
                                        one.v[0] = out[r].v[h];
                                        store_vec_wt<TVec<T, 1>>(p, one);
                                    } else {
                                        *reinterpret_cast<T*>(p) = out[r].v[h];
                                    }
                                }
                        }
                }
            };
            if (a.nts == 2) {  // agent-scope write-through ("self-released" launch: smr_device.h)
                if constexpr (has_wt_store<VT>::value) {
    #pragma unroll
                    for (int r = 0; r < NREP; ++r)
                        if (cntd[r] == (uint32_t)V) store_vec_wt<VT>(bp0 + (O)(row[0].g + a.op[0].Gr[r]), out[r]);
                    if constexpr (has_wt_store<TVec<T, 1>>::value) put_partial_vecs(BoolC<true>{});
                    self_release_wait();
                }
            } else if (a.nts) {
                nt_block_guard();
                put(BoolC<true>{});
                nt_block_guard();
                put_partial_vecs(BoolC<false>{});
            } else {
                put(BoolC<false>{});
                put_partial_vecs(BoolC<false>{});
            }
        }
    };
    if constexpr (EDGE) {
        if (edge) body(BoolC<true>{});
        else body(BoolC<false>{});
    } else {
        body(BoolC<false>{});
    }
}


// ---- persistent, software-pipelined form ---------------------------------------------------------------
// For grids larger than the machine holds at once.  The classic form runs one tile per workgroup
// and its phases (table rows -> global loads -> LDS exchange -> stores) strictly one after the
// other; with one 1024-lane workgroup per CU nothing overlaps, and every workgroup re-reads its
// 40 KiB of lane-table rows (measured at 128^4: 24 % of all L2 read requests, 7.4 us per tile,
// 3 TB/s of HBM traffic).  Here a workgroup stays resident, reads its table rows ONCE and walks the
// work list with stride gridDim.x; the global loads of tile i+1 are issued right after tile i's
// staged values have gone to LDS, so they are in flight during tile i's LDS reads, f and stores.
// Only instantiated without bounds checks (no ragged tiled dim).
template <class T, class F, bool MIXED, bool WIDE, int V, int THRLOG>
SMR_DEV void tiled_map_pipe_body(const TiledArgs<WIDE> a, F f) {
    typedef typename off_t_of<WIDE>::type O;
    typedef TVec<T, V> VT;
    constexpr int NREP = EPL / V;
    constexpr int NT = 1 << THRLOG;
    constexpr int NIN_STATIC = F::NIN;
    constexpr int NINMAX = (NIN_STATIC >= 0) ? NIN_STATIC : MAXIN;
    constexpr int NX = NINMAX > 0 ? NINMAX : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int nin = (NIN_STATIC >= 0) ? NIN_STATIC : a.M - 1;
    const uint32_t tid = threadIdx.x;

    constexpr bool BITS = SMR_TILED_BITS && !WIDE;
    constexpr int VLOGP = (V == 1) ? 0 : (V == 2 ? 1 : 2);
    LaneRow<WIDE> row[NINMAX + 1];
#pragma unroll
    for (int k = 0; k <= NINMAX; ++k) {
        row[k].g = 0;
        row[k].l = 0;
        if constexpr (BITS) {
            if (NIN_STATIC >= 0 || k <= nin) {
                const uint32_t e0 = tid << VLOGP;
                uint32_t g = 0, idx = 0;
#pragma unroll
                for (int p = 0; p < MAXT; ++p) {
                    const uint32_t c = __builtin_amdgcn_ubfe(e0, (uint32_t)a.bpos[k][p], (uint32_t)a.blen[k][p]);
                    g += c * a.bstr[k][p];
                    idx |= c << a.blsh[k][p];
                }
                row[k].g = g;
                row[k].l = idx ^ (((idx >> a.fs1) ^ (idx >> a.fs2)) & a.fmask);
            }
        } else {
            if (k <= nin) row[k] = a.lanetab[k * NT + tid];
        }
    }

    // work-list entry -> tile id (0xffffffff: none, and none after it for this workgroup)
    auto tile_of = [&](uint32_t e) -> uint32_t {
        if (e >= (uint32_t)a.nwork) return 0xffffffffu;
        return a.ordmode == 2 ? a.ordtab[e] : e;
    };
    // tile id -> per-operand tile origins
    auto origins = [&](uint32_t b, char* (&bp)[NINMAX + 1]) {
        uint32_t tc[MAXN];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t q = fastdiv(b, a.div_m[g], a.div_s[g]);
            tc[g] = b - q * a.ntiles[g];
            b = q;
        }
#pragma unroll
        for (int g = NG; g < MAXN; ++g) tc[g] = 0;
        if (a.ng > NG) {
#pragma unroll
            for (int g = NG; g < MAXN; ++g) {
                const uint32_t q = fastdiv(b, a.div_m[g], a.div_s[g]);
                tc[g] = b - q * a.ntiles[g];
                b = q;
            }
        }
#pragma unroll
        for (int k = 0; k <= NINMAX; ++k) {
            bp[k] = nullptr;
            if (k <= nin) {
                if (a.base32) {
                    uint32_t o = 0;
#pragma unroll
                    for (int g = 0; g < NG; ++g) o += tc[g] * a.op[k].tstep32[g];
                    bp[k] = (char*)a.op[k].base + o;
                } else {
                    i64 o = 0;
#pragma unroll
                    for (int g = 0; g < MAXN; ++g)
                        if (g < NG || a.ng > NG) o += (i64)tc[g] * a.tstep[k][g];
                    bp[k] = (char*)a.op[k].base + o;
                }
            }
        }
    };
    auto issue_loads = [&](char* const (&bp)[NINMAX + 1], VT (&x)[NX][NREP]) {
#pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
            if (i < nin) {
                const OpDesc<WIDE>& d = a.op[i + 1];
#pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    const char* p = bp[i + 1] + (O)(row[i + 1].g + d.Gr[r]);
                    if constexpr (V == 1) {
                        x[i][r].v[0] = load_at<T, MIXED>(p, d.dtype, d.conj);
                    } else {
                        x[i][r] = *reinterpret_cast<const VT*>(p);
                        if constexpr (tr<T>::cx) {
                            if (d.conj) {
#pragma unroll
                                for (int h = 0; h < V; ++h) x[i][r].v[h] = cj(x[i][r].v[h]);
                            }
                        }
                    }
                }
            }
        }
    };

    uint32_t e = blockIdx.x;
    uint32_t tile = tile_of(e);
    if (tile == 0xffffffffu) return;
    char* bp[NINMAX + 1];
    VT x[NX][NREP], xn[NX][NREP];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int r = 0; r < NREP; ++r)
#pragma unroll
            for (int h = 0; h < V; ++h) {
                x[i][r].v[h] = T{};
                xn[i][r].v[h] = T{};
            }
    origins(tile, bp);
    issue_loads(bp, x);
    while (true) {
        char* const bp0 = bp[0];
        // staged inputs of the current tile -> LDS (destination order, swizzled)
#pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
            if (i < nin && a.staged[i + 1] >= 0) {
                const OpDesc<WIDE>& d = a.op[i + 1];
                T* L = lds + ((size_t)a.staged[i + 1] << a.tilelog);
#pragma unroll
                for (int r = 0; r < NREP; ++r)
#pragma unroll
                    for (int h = 0; h < V; ++h) L[row[i + 1].l ^ d.Lr[r] ^ d.Lh[h]] = x[i][r].v[h];
            }
        }
        __syncthreads();
        // next tile: its loads fly while this one is finished
        e += gridDim.x;
        const uint32_t next = tile_of(e);
        const bool more = next != 0xffffffffu;
        if (more) {
            origins(next, bp);
            issue_loads(bp, xn);
        }
        // current tile: LDS -> registers (destination order), f, store
#pragma unroll
        for (int i = 0; i < NINMAX; ++i) {
            if (i < nin && a.staged[i + 1] >= 0) {
                const T* L = lds + ((size_t)a.staged[i + 1] << a.tilelog);
#pragma unroll
                for (int r = 0; r < NREP; ++r)
#pragma unroll
                    for (int h = 0; h < V; ++h) x[i][r].v[h] = L[row[0].l ^ a.Lrd[r] ^ a.Lhd[h]];
            }
        }
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            VT out;
#pragma unroll
            for (int h = 0; h < V; ++h) {
                T arg[MAXIN];
#pragma unroll
                for (int i = 0; i < MAXIN; ++i) {
                    arg[i] = T{};
                    if (i < NINMAX) arg[i] = x[i < NINMAX ? i : 0][r].v[h];
                }
                out.v[h] = f(arg);
            }
            char* p = bp0 + (O)(row[0].g + a.op[0].Gr[r]);
            if constexpr (V == 1) {
                store_at<T, MIXED>(p, a.op[0].dtype, a.op[0].conj, out.v[0]);
            } else {
                if constexpr (tr<T>::cx) {
                    if (a.op[0].conj) {
#pragma unroll
                        for (int h = 0; h < V; ++h) out.v[h] = cj(out.v[h]);
                    }
                }
                store_vec<VT>(p, out, a.nts);
            }
        }
        if (!more) break;
        __syncthreads();  // every lane has read its LDS values before the next tile overwrites them
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int r = 0; r < NREP; ++r) x[i][r] = xn[i][r];
    }
}

#ifndef SMR_JIT
template <class T, class F, bool MIXED, bool WIDE, int V, int MODE, int THRLOG>
__global__ void __launch_bounds__(1 << THRLOG) k_tiled_map(const TiledArgs<WIDE> a, F f SMR_STAMP_PARAM) {
    SMR_STAMP_BEGIN
    tiled_map_body<T, F, MIXED, WIDE, V, MODE, THRLOG>(a, f);
    SMR_STAMP_END
}

template <class T, class F, bool MIXED, bool WIDE, int V, int THRLOG>
__global__ void __launch_bounds__(1 << THRLOG) k_tiled_map_pipe(const TiledArgs<WIDE> a, F f SMR_STAMP_PARAM) {
    SMR_STAMP_BEGIN
    tiled_map_pipe_body<T, F, MIXED, WIDE, V, THRLOG>(a, f);
    SMR_STAMP_END
}

// ---- LDS swizzle ------------------------------------------------------------------------------------
// l' = l ^ XOR_{b >= w, bit b of l set} mask[b]: the low w index bits (the 128 B one LDS write
// group spans, in elements) are XORed with one w-bit mask per higher index bit.  A lane group's
// accesses are conflict-free iff the slot images of the index bits its lanes toggle are linearly
// independent over GF(2).  The masks are searched on the host (a small hill climb, once per plan)
// against the write pattern of every staged operand and the destination-order read pattern; the
// start point is the plain fold mask[b] = 1 << ((b - w) mod w).
struct Swizzle {
    int w;
    uint32_t mask[32];
    uint32_t apply(uint32_t l) const {
        uint32_t r = l;
        for (int b = w; b < 32; ++b)
            if ((l >> b) & 1u) r ^= mask[b];
        return r;
    }
};

static int gf2_rank(const uint32_t* v, int n) {
    uint32_t basis[32];
    int r = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t x = v[i];
        for (int j = 0; j < r; ++j)
            if ((x ^ basis[j]) < x) x ^= basis[j];
        if (x) {
            basis[r++] = x;
            for (int j = r - 1; j > 0 && basis[j] > basis[j - 1]; --j) std::swap(basis[j], basis[j - 1]);
        }
    }
    return r;
}

// the index bits one hardware lane group toggles + the width of the bank-slot space it lands in
struct LanePattern {
    int n;
    int bits[8];
    int slotbits;
};

static Swizzle choose_swizzle(int tilelog, int w, const std::vector<LanePattern>& pats, int* cost0 = nullptr, int* cost1 = nullptr) {
    Swizzle sw;
    sw.w = (w > 0 && tilelog > w) ? w : 32;
    for (int b = 0; b < 32; ++b) sw.mask[b] = 0;
    if (sw.w == 32) return sw;
    for (int b = w; b < tilelog; ++b) sw.mask[b] = 1u << ((b - w) % w);
    auto cost = [&](const Swizzle& s) {
        int c = 0;
        for (const LanePattern& p : pats) {
            uint32_t img[8];
            const uint32_t sm = (1u << p.slotbits) - 1u;
            for (int i = 0; i < p.n; ++i) {
                const int b = p.bits[i];
                uint32_t v = 1u << b;          // the bit itself (dropped below if outside the slot space)
                if (b >= w) v ^= s.mask[b];   // its swizzle mask (acts on the low w bits)
                img[i] = v & sm;
            }
            c += (1 << (p.n - gf2_rank(img, p.n))) - 1;
        }
        return c;
    };
    int best = cost(sw);
    if (cost0) *cost0 = best;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < 6000 && best > 0; ++it) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const int b = w + (int)((rng >> 33) % (uint64_t)(tilelog - w));
        const uint32_t m = (uint32_t)((rng >> 20) & ((1u << w) - 1u));
        Swizzle t = sw;
        t.mask[b] = m;
        const int c = cost(t);
        if (c <= best) {
            best = c;
            sw = t;
        }
    }
    if (cost1) *cost1 = best;
    return sw;
}

template <class T, class F, bool MIXED, bool WIDE, int V, int MODE, int THRLOG>
static int go3e(const Plan& plan, hipStream_t s, F f, const OpTab& tab, bool ua = false) {
    constexpr bool EDGE = (MODE & 1) != 0;
    typedef typename off_t_of<WIDE>::type O;
    constexpr int NREP = EPL / V;
    constexpr int NT = 1 << THRLOG;
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    int vlog = 0;
    while ((1 << vlog) < V) ++vlog;
    constexpr int variant = (WIDE ? 2 : 0) + (V > 1 ? 1 : 0);
    const size_t lds = (size_t)t.nstaged * ((size_t)1 << t.tilelog) * sizeof(T);
    const unsigned grid_ = t.ord.empty() ? (unsigned)t.grid : (unsigned)t.ord.size();
    // persistent, software-pipelined form when the work list is longer than the machine holds at once
    unsigned pgrid = 0;
    if constexpr (!EDGE) {
        const Options& o = options();
        if (o.tiled_persist && !t.no_persist && !ua) {  // (the persistent form keeps aligned vector accesses)
            static const int ncu = [] {
                int dev = 0, n = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                (void)hipGetLastError();
                return n;
            }();
            i64 wpc = std::min<i64>(2048 >> THRLOG, lds ? (i64)(160 * 1024 / lds) : 8);
            wpc = std::max<i64>(1, std::min<i64>(wpc, 4));  // measured: 4 workgroups per CU beat 8 and 2
            if (o.tiled_persist_wpc > 0) wpc = o.tiled_persist_wpc;
            const i64 cap = (i64)ncu * wpc / 8 * 8;
            if ((i64)grid_ >= cap * o.tiled_persist_min && cap >= 8) pgrid = (unsigned)cap;
        }
    }
    auto launch = [&](const TiledArgs<WIDE>& ka) -> int {
        auto b2s = [](bool b) { return b ? "true" : "false"; };
        if constexpr (is_jit<F>::value) {
            JitLaunch l;
            l.family = "tiled";
            l.tname = tname<T>();
            l.argtype = WIDE ? "smr::TiledArgs<true>" : "smr::TiledArgs<false>";
            if (pgrid)
                l.entry = std::string("smr::tiled_map_pipe_body<") + tname<T>() + ", smr::FJit, " + b2s(MIXED) + ", " + b2s(WIDE) + ", " +
                          std::to_string(V) + ", " + std::to_string(THRLOG) + ">(a, smr::FJit{kc});";
            else
                l.entry = std::string("smr::tiled_map_body<") + tname<T>() + ", smr::FJit, " + b2s(MIXED) + ", " + b2s(WIDE) + ", " +
                          std::to_string(V) + ", " + std::to_string(MODE) + ", " + std::to_string(THRLOG) + ">(a, smr::FJit{kc});";
            l.grid = pgrid ? pgrid : grid_;
            l.block = 1u << THRLOG;
            l.lds = lds;
            l.args = &ka;
            l.argsize = sizeof ka;
            return jit_launch(c, l, s);
        } else {
            if (jit_no_launch()) return SMR_OK;
            clear_sticky_error();
            if constexpr (!EDGE) {
                if (pgrid) {
                    auto kern = k_tiled_map_pipe<T, F, MIXED, WIDE, V, THRLOG>;
                    if (lds > 64 * 1024) {
                        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
                    }
                    SMR_LAUNCH(kern, dim3(pgrid), dim3(1u << THRLOG), lds, s, ka, f SMR_STAMP_ARG(pgrid, 1u << THRLOG));
                    return check_launch("k_tiled_map_pipe");
                }
            }
            auto kern = k_tiled_map<T, F, MIXED, WIDE, V, MODE, THRLOG>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
            }
            mark_sliceable(1, (unsigned)offsetof(TiledArgs<WIDE>, blk0), 0);  // a workgroup owns its tile: block ranges are independent
            if (ka.nts == 2) mark_self_released();
            SMR_LAUNCH(kern, dim3(grid_), dim3(1u << THRLOG), lds, s, ka, f SMR_STAMP_ARG(grid_, 1u << THRLOG));
            return check_launch("k_tiled_map");
        }
    };
    TiledArgs<WIDE> a;
    // non-temporal stores: measured faster or equal whenever a tile writes whole 128-byte lines (32^4 Float64
    // permutedims! 3.36 -> 2.76 us, 128^4 864 -> 818 us, never slower for a consumer kernel that follows);
    // partial lines must meet in L2 first, so short destination runs keep plain stores
    int nts_now = (options().nt_store > 0) ? 1 : 0;
    if (options().nt_store < 0) {
        const i64 run = std::min<i64>(c.dims[0], (i64)1 << t.tlog[0]) * c.esize[0];
        nts_now = (c.strides[0][0] == 1 && run >= 128) ? 1 : 0;
    }
    // write-through stores (policy 2): forced, or a launch recorded for a sequence (its packet then needs no release fence).  Only the
    // one-shot vector form: there every store of the kernel is one of the vector stores below.
    if constexpr (V > 1 && !MIXED && has_wt_store<TVec<T, V>>::value) {
        if (pgrid == 0 && (options().nt_store == 2 || want_self_release(plan))) nts_now = 2;
    }
    // the arguments depend on the plan only, except for the operand addresses: built once
    std::vector<unsigned char>& cached = plan.tiled_args[variant];
    if (cached.size() == sizeof a) {
        std::memcpy(&a, cached.data(), sizeof a);
        if constexpr ((MODE & 4) == 0) {
            if (!a.base32 || a.ng > NG) return go3e<T, F, MIXED, WIDE, V, 7, THRLOG>(plan, s, f, tab, ua);
        }
        for (int k = 0; k < c.M; ++k) a.op[k].base = tab.base[k];
        a.nts = nts_now;
        return launch(a);
    }
    std::memset(&a, 0, sizeof a);
    a.nwork = (int32_t)grid_;
    if (!t.ord.empty() && t.ord.size() <= (size_t)NORD16 && t.grid < 0xffff) {
        a.ordmode = 1;
        for (size_t i = 0; i < t.ord.size(); ++i) {
            const uint32_t v = t.ord[i] == 0xffffffffu ? 0xffffu : t.ord[i];
            a.ord16[i >> 1] |= v << (16 * (i & 1));
        }
    } else if (!t.ord.empty()) {
        a.ordmode = 2;
        if (!plan.ordtab && !jit_dry_run()) {
            void* dptr = nullptr;
            hipError_t e = hipMalloc(&dptr, t.ord.size() * sizeof(uint32_t));
            if (e != hipSuccess) return hip_error(e, "hipMalloc(tile order)");
            e = hipMemcpy(dptr, t.ord.data(), t.ord.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(dptr);
                return hip_error(e, "hipMemcpy(tile order)");
            }
            plan.ordtab = dptr;
        }
        a.ordtab = reinterpret_cast<const uint32_t*>(plan.ordtab);
    }
    a.M = c.M;
    a.nt = t.nt;
    a.tilelog = t.tilelog;
    a.nstaged = t.nstaged;
    int tlogdim[MAXN] = {0};
    int lsh[MAXT];
    int sh = 0;
    for (int j = 0; j < t.nt; ++j) {
        lsh[j] = sh;
        tlogdim[t.tdim[j]] = t.tlog[j];
        sh += t.tlog[j];
    }
    // LDS swizzle: slot width = the 128 B an LDS write group spans, in elements
    int w = 0;
    while ((sizeof(T) << w) < 128) ++w;
    std::vector<LanePattern> pats;
    if (sizeof(T) >= 4 && t.tilelog > w) {
        const int nread = sizeof(T) == 16 ? 4 : 5;  // ds_read_b32/b64: 32 lanes, b128: 16 lanes per pass
        for (int k = 1; k < c.M; ++k) {
            if (t.staged[k] < 0) continue;
            // operand k's enumeration: its bit (vlog + i) is lane bit i of the write group
            int bitpos[32], pos = 0;
            for (int jj = 0; jj < t.nt; ++jj) {
                const int j = t.order[k][jj];
                for (int bit = 0; bit < t.tlog[j]; ++bit) bitpos[pos++] = lsh[j] + bit;
            }
            LanePattern p;
            p.n = w;
            p.slotbits = w;
            for (int i2 = 0; i2 < w; ++i2) p.bits[i2] = bitpos[vlog + i2];
            pats.push_back(p);
        }
        LanePattern r;
        r.n = nread;
        r.slotbits = nread;
        for (int i2 = 0; i2 < nread; ++i2) r.bits[i2] = vlog + i2;
        pats.push_back(r);
    }
    int cost0 = 0, cost1 = 0;
    constexpr bool BITS = SMR_TILED_BITS && !WIDE && (MODE & 4) == 0;
    Swizzle swz = choose_swizzle(t.tilelog, w, pats, &cost0, &cost1);
    uint32_t fs1 = 31, fs2 = 31, fmask = 0;
    if (BITS && swz.w != 32) {
        // the kernel computes the swizzle itself: restrict it to XOR folds of the index onto its low w bits,
        // i ^ (((i >> s1) ^ (i >> s2)) & (2^w - 1)), s1 >= w; pick the fold with the fewest bank conflicts
        auto fold = [&](int s1, int s2) {
            Swizzle f;
            f.w = w;
            for (int b = 0; b < 32; ++b) {
                f.mask[b] = 0;
                if (b >= s1 && b < t.tilelog) f.mask[b] ^= (1u << (b - s1)) & ((1u << w) - 1u);
                if (b >= s2 && b < t.tilelog) f.mask[b] ^= (1u << (b - s2)) & ((1u << w) - 1u);
            }
            return f;
        };
        auto fcost = [&](const Swizzle& f) {
            int cst = 0;
            for (const LanePattern& pt : pats) {
                uint32_t img[8];
                const uint32_t sm = (1u << pt.slotbits) - 1u;
                for (int i2 = 0; i2 < pt.n; ++i2) {
                    const int b = pt.bits[i2];
                    uint32_t v = 1u << b;
                    if (b >= w) v ^= f.mask[b];
                    img[i2] = v & sm;
                }
                cst += (1 << (pt.n - gf2_rank(img, pt.n))) - 1;
            }
            return cst;
        };
        int bestc = 1 << 30;
        for (int s1 = w; s1 < t.tilelog; ++s1)
            for (int s2 = s1 + 1; s2 <= t.tilelog; ++s2) {  // s2 == tilelog: one-term fold
                const Swizzle f = fold(s1, s2 == t.tilelog ? 32 : s2);
                const int cst = fcost(f);
                if (cst < bestc) {
                    bestc = cst;
                    swz = f;
                    fs1 = (uint32_t)s1;
                    fs2 = s2 == t.tilelog ? 31u : (uint32_t)s2;
                    fmask = (1u << w) - 1u;
                }
            }
        cost1 = bestc;
    }
    if (std::getenv("SMR_DEBUG_SWIZZLE"))
        std::fprintf(stderr, "[smr] tiled swizzle: w=%d V=%d patterns=%zu conflict cost fold=%d searched=%d\n", w, V, pats.size(), cost0, cost1);
    // grid dims: canonical dims with more than one tile, in the order the planner chose (TilePlan::gorder: canonical order, or --
    // HBM-sized transposes -- the tile index along the INPUT's unit axis second: plan_tiles)
    int gof[MAXN], ng = 0;
    for (int d = 0; d < c.N; ++d) gof[d] = -1;
    for (int i = 0; i < c.N; ++i) {
        const int d = t.gorder[i];
        if (d >= 0 && d < c.N && gof[d] < 0 && t.ntiles[d] > 1) gof[d] = ng++;
    }
    for (int d = 0; d < c.N; ++d)  // (anything the planner's order left out: canonical order behind it)
        if (gof[d] < 0 && t.ntiles[d] > 1) gof[d] = ng++;
    a.gflip = 0;
    auto is_ragged = [&](int d) { return tlogdim[d] > 0 && (c.dims[d] & (((i64)1 << tlogdim[d]) - 1)) != 0; };
    int nragged = 0;
    for (int d = 0; d < c.N; ++d)
        if (gof[d] >= 0 && is_ragged(d)) ++nragged;
    // (measured with ONE ragged grid dim: transposes of (7200,104) 4.05 -> 3.23 us, (7168,100) 4.08 -> 3.72, (96,9000) 3.42 -> 3.21;
    // with two -- (100,90,80) permutes -- a tie or a loss: those keep the planner's order.  tiled_edge_first = 2: whenever it applies)
    // ... with two or more they keep the planner's order and only count backwards ((257,129,65) 8.2 -> 7.8 us, (1400,1500) 6.6 -> 6.5;
    // moved to the slow end as well: (300,301,35) 12.1 -> 13.1).  tiled_edge_first = 3: backwards only, 2: moved + backwards, always
    if (EDGE && (MODE & 4) == 0 && t.ord.empty() && (options().tiled_edge_first == 3 || (options().tiled_edge_first == 1 && nragged >= 2))) {
        for (int d = 0; d < c.N; ++d)
            if (gof[d] >= 0 && gof[d] < NG && is_ragged(d)) a.gflip |= 1u << gof[d];
    } else if (EDGE && (MODE & 4) == 0 && t.ord.empty() && (options().tiled_edge_first == 2 || (options().tiled_edge_first == 1 && nragged == 1))) {
        // ragged dims to the slow end of the grid (stable), counted backwards
        int order[MAXN], n2 = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int g = 0; g < ng; ++g)
                for (int d = 0; d < c.N; ++d)
                    if (gof[d] == g && (is_ragged(d) ? 1 : 0) == pass) order[n2++] = d;
        for (int i = 0; i < n2; ++i) {
            gof[order[i]] = i;
            if (is_ragged(order[i])) a.gflip |= 1u << i;
        }
    }
    for (int g = 0; g < MAXN; ++g) {
        a.ntiles[g] = 1;
        a.div_m[g] = 0;
        a.div_s[g] = 0;
        a.last_ragged[g] = 0xffffffffu;
        a.gdims[g] = 1;
        a.glog[g] = 0;
    }
    for (int d = 0; d < c.N; ++d) {
        const int g = gof[d];
        if (g < 0) continue;
        const uint32_t nt = (uint32_t)t.ntiles[d];
        a.ntiles[g] = nt;
        int l = 0;
        while ((1ull << l) < nt) ++l;
        a.div_m[g] = (uint32_t)((((1ull << 32) * ((1ull << l) - nt)) / nt) + 1);
        a.div_s[g] = (uint32_t)l;
        a.gdims[g] = c.dims[d];
        a.glog[g] = tlogdim[d];
        if (c.dims[d] & (((i64)1 << tlogdim[d]) - 1)) a.last_ragged[g] = nt - 1;
    }
    for (int j = 0; j < t.nt; ++j) {
        a.tlog[j] = t.tlog[j];
        int g = gof[t.tdim[j]];
        if (g < 0 && (c.dims[t.tdim[j]] & (((i64)1 << t.tlog[j]) - 1))) {
            // a tiled dim covered by a single, partly empty tile: give it a one-tile grid slot so
            // that every workgroup takes the bounds-checked path
            if (ng >= MAXN) return set_error(SMR_EUNSUPPORTED, "tiled: too many grid dims");
            g = ng++;
            a.gdims[g] = c.dims[t.tdim[j]];
            a.glog[g] = t.tlog[j];
            a.last_ragged[g] = 0;  // tc == 0 always
        }
        a.tgrid[j] = g;
    }
    for (int j = 0; j < MAXT; ++j) {
        a.ej_g[j] = -1;
        a.ej_ntm1[j] = 0;
        a.ej_last[j] = 0x7fffffffu;
        if (j < t.nt && a.tgrid[j] >= 0) {
            const i64 ext = c.dims[t.tdim[j]], tile = (i64)1 << t.tlog[j], nt = (ext + tile - 1) / tile;
            a.ej_g[j] = a.tgrid[j];
            a.ej_ntm1[j] = (uint32_t)(nt - 1);
            a.ej_last[j] = (uint32_t)(ext - (nt - 1) * tile);  // 1 .. tile (a whole last tile keeps every lane: lim = tile)
        }
    }
    a.ng = ng;
    // 32-bit tile-origin arithmetic when every operand's tile origins stay below 4 GiB
    bool base32 = ng <= NG;
    for (int k = 0; k < c.M && base32; ++k) {
        long double span = 0;
        for (int d = 0; d < c.N; ++d) {
            if (gof[d] < 0) continue;
            if (c.strides[k][d] < 0) base32 = false;
            span += (long double)c.strides[k][d] * c.esize[k] * (long double)(t.ntiles[d] - 1) * (long double)((i64)1 << tlogdim[d]);
        }
        if (span >= 4294967296.0L) base32 = false;
    }
    a.base32 = base32 ? 1 : 0;
    if constexpr ((MODE & 4) == 0) {
        // the lean variants assume 32-bit tile origins over at most 4 grid dims
        if (!base32 || ng > NG) return go3e<T, F, MIXED, WIDE, V, 7, THRLOG>(plan, s, f, tab, ua);
    }

    // per-lane table: built once per (plan, kernel variant), kept in device memory (not needed by the BITS form)
    const bool build_tab = !BITS && plan.lanetab[variant] == nullptr && !jit_dry_run();
    a.fs1 = fs1;
    a.fs2 = fs2;
    a.fmask = fmask;
    std::vector<LaneRow<WIDE>> rows;
    if (build_tab) rows.assign((size_t)c.M * NT, LaneRow<WIDE>{});

    for (int k = 0; k < MAXM; ++k) a.staged[k] = -1;
    for (int k = 0; k < c.M; ++k) {
        const bool own = k > 0 && t.staged[k] >= 0;
        OpDesc<WIDE>& d = a.op[k];
        const i64 es = c.esize[k];
        a.staged[k] = own ? t.staged[k] : -1;
        d.base = tab.base[k];
        d.dtype = tab.dtype[k];
        d.conj = tab.conj[k];
        for (int dd = 0; dd < c.N; ++dd)
            if (gof[dd] >= 0) {
                const i64 st = c.strides[k][dd] * ((i64)1 << tlogdim[dd]) * es;
                a.tstep[k][gof[dd]] = st;
                if (gof[dd] < NG) d.tstep32[gof[dd]] = (uint32_t)st;
            }
        // per-bit contributions in this operand's enumeration order
        i64 gbit[32] = {0};
        uint32_t lbit[32] = {0};
        int pos = 0;
        for (int jj = 0; jj < t.nt; ++jj) {
            const int j = own ? t.order[k][jj] : jj;
            a.esh[k][j] = pos;
            a.bpos[k][jj] = pos;
            a.blen[k][jj] = t.tlog[j];
            a.blsh[k][jj] = lsh[j];
            a.bstr[k][jj] = (uint32_t)(c.strides[k][t.tdim[j]] * es);
            for (int bit = 0; bit < t.tlog[j]; ++bit) {
                gbit[pos + bit] = c.strides[k][t.tdim[j]] * ((i64)1 << bit) * es;
                lbit[pos + bit] = swz.apply(1u << (lsh[j] + bit));
            }
            pos += t.tlog[j];
        }
        for (int h = 0; h < V; ++h) {
            uint32_t l = 0;
            for (int bit = 0; bit < vlog; ++bit)
                if ((h >> bit) & 1) l ^= lbit[bit];
            d.Lh[h] = l;
        }
        for (int r = 0; r < NREP; ++r) {
            i64 g = 0;
            uint32_t l = 0;
            for (int bit = 0; bit < 5; ++bit)
                if ((r >> bit) & 1) {
                    g += gbit[vlog + THRLOG + bit];
                    l ^= lbit[vlog + THRLOG + bit];
                }
            d.Gr[r] = (O)g;
            d.Lr[r] = l;
        }
        if (build_tab)
            for (int tid = 0; tid < NT; ++tid) {
                i64 g = 0;
                uint32_t l = 0;
                for (int bit = 0; bit < THRLOG; ++bit)
                    if ((tid >> bit) & 1) {
                        g += gbit[vlog + bit];
                        l ^= lbit[vlog + bit];
                    }
                rows[(size_t)k * NT + tid].g = (O)g;
                rows[(size_t)k * NT + tid].l = l;
            }
    }
    for (int r = 0; r < NREP; ++r) a.Lrd[r] = a.op[0].Lr[r];
    for (int h = 0; h < V; ++h) a.Lhd[h] = a.op[0].Lh[h];
    if (build_tab) {
        void* dptr = nullptr;
        const size_t bytes = rows.size() * sizeof(LaneRow<WIDE>);
        hipError_t e = hipMalloc(&dptr, bytes);
        if (e != hipSuccess) return hip_error(e, "hipMalloc(lane table)");
        // synchronous upload, once per plan and kernel variant (must not happen inside a stream
        // capture: execute a plan once before capturing it into a hipGraph)
        e = hipMemcpy(dptr, rows.data(), bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(dptr);
            return hip_error(e, "hipMemcpy(lane table)");
        }
        plan.lanetab[variant] = dptr;
    }
    a.lanetab = reinterpret_cast<const LaneRow<WIDE>*>(plan.lanetab[variant]);

    if (!jit_dry_run()) {
        cached.resize(sizeof a);
        std::memcpy(cached.data(), &a, sizeof a);
    }
    a.nts = nts_now;
    return launch(a);
}

template <class T, class F, bool MIXED, bool WIDE, int V, int THRLOG>
static int go3(const Plan& plan, hipStream_t s, F f, const OpTab& tab, bool ua = false) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    bool ragged = false;
    for (int j = 0; j < t.nt; ++j)
        if (c.dims[t.tdim[j]] & (((i64)1 << t.tlog[j]) - 1)) ragged = true;
    // ragged extents: the lean kernel plus bounds checks in the workgroups that sit on a last, partly filled tile (round 6; every
    // ragged problem used to take variant 7 -- lane tables from memory, 64-bit origins, order lookups -- and paid ~2 us for it:
    // transposes of 7200 x 100 Float64 5.4 us against 3.0 us for 7200 x 128, profiles/r06_ragged_tiles.txt)
    if (options().tiled_force_edge) ragged = true;  // experiment: the bounds-checking variant on whole tiles (what does the variant itself cost?)
    bool pv = false;  // does some operand's vector axis end in a partial vector?
    if constexpr (V > 1) {
        for (int k = 0; k < c.M; ++k) {
            const int j0 = (k > 0 && t.staged[k] >= 0) ? t.order[k][0] : 0;
            if (c.dims[t.tdim[j0]] % V) pv = true;
        }
        if (ragged && t.ord.empty() && pv) return go3e<T, F, MIXED, WIDE, V, 9, THRLOG>(plan, s, f, tab, ua);
    }
    if (ragged && t.ord.empty()) return go3e<T, F, MIXED, WIDE, V, 1, THRLOG>(plan, s, f, tab, ua);
    if (ragged) return go3e<T, F, MIXED, WIDE, V, 7, THRLOG>(plan, s, f, tab, ua);
    if (!t.ord.empty()) return go3e<T, F, MIXED, WIDE, V, 2, THRLOG>(plan, s, f, tab, ua);
    return go3e<T, F, MIXED, WIDE, V, 0, THRLOG>(plan, s, f, tab, ua);
}

// The same at element alignment (round 6): odd extents, odd row strides, views that begin inside a vector.  Every operand still runs
// along its unit axis; the one partial vector at the end of a row (extent not a multiple of V) is moved element by element by the
// workgroups of the ragged last tile.  Elements of 4 or 8 bytes.
template <class T>
static bool vector_ok_ua(const Plan& plan, const OpTab& tab, int V) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    int vlog = 0;
    while ((1 << vlog) < V) ++vlog;
    if (sizeof(T) < 4) return false;
    for (int k = 0; k < c.M; ++k) {
        const bool staged = k > 0 && t.staged[k] >= 0;
        const int j0 = staged ? t.order[k][0] : 0;
        const int d0 = t.tdim[j0];
        if (t.tlog[j0] < vlog) return false;
        if (c.strides[k][d0] != 1 || c.dims[d0] < V) return false;
        if (((uintptr_t)tab.base[k]) % sizeof(T)) return false;
    }
    return true;
}

// Can every operand be accessed V elements at a time (V * sizeof(T) <= 16 bytes)?
template <class T>
static bool vector_ok(const Plan& plan, const OpTab& tab, int V) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    int vlog = 0;
    while ((1 << vlog) < V) ++vlog;
    const size_t vb = (size_t)V * sizeof(T);
    for (int k = 0; k < c.M; ++k) {
        const bool staged = k > 0 && t.staged[k] >= 0;
        const int j0 = staged ? t.order[k][0] : 0;  // first axis of this operand's enumeration
        const int d0 = t.tdim[j0];
        if (t.tlog[j0] < vlog) return false;
        // a direct input that is not unit-stride along dim 0 (broadcast, odd stride) has no V-wide form
        if (c.strides[k][d0] != 1) return false;
        if (c.dims[d0] % V) return false;
        if (((uintptr_t)tab.base[k]) % vb) return false;
        for (int d = 0; d < c.N; ++d)
            if (d != d0 && (c.strides[k][d] % V)) return false;
    }
    return true;
}

template <class T, class F, bool MIXED, int THRLOG>
static int go_tl(const Plan& plan, hipStream_t s, F f, const OpTab& tab, bool narrow) {
    if (!narrow) return go3<T, F, MIXED, true, 1, THRLOG>(plan, s, f, tab);
    if constexpr (!MIXED && sizeof(T) < 16) {
        // a lane's 4 elements as 16-byte vectors (8-byte for 1/2-byte element types)
        constexpr int VMAX = (16 / sizeof(T)) > 4 ? 4 : (int)(16 / sizeof(T));
        if (options().tiled_vec && vector_ok<T>(plan, tab, VMAX)) return go3<T, F, false, false, VMAX, THRLOG>(plan, s, f, tab);
        if (options().tiled_vec && options().tiled_uavec && vector_ok_ua<T>(plan, tab, VMAX)) return go3<T, F, false, false, VMAX, THRLOG>(plan, s, f, tab, true);
    }
    return go3<T, F, MIXED, false, 1, THRLOG>(plan, s, f, tab);
}

// ---- XPOSE: the lean form of an HBM-sized transposing copy (round 5) -------------------------------------------------------------
// One staged input, 128 x 32 tiles, every tiled extent a whole number of tiles, 16-byte accesses: permutedims! / adjoint! / copy! of
// a permuted view / B .= c .* A' on arrays of 1 GiB and more.  The general kernel above carries lane tables (16 KiB of table rows per
// 1024-lane workgroup: a quarter more L2 read requests than the payload), swizzle masks, tile-order lookups and edge code through its
// prologue; at 65536 workgroups per launch that is 8 % of the run time (tools/xpose_proto.hip: 723 us against 782 us for the same
// tile shape and grid order at 128^4 Float64).  Here the lane's place in the tile is two integer divisions of its id, the tile's
// origin a handful of divisions of the workgroup id, every load of the tile is issued before the first LDS write, the LDS rows
// carry a pitch of TQ + 2 elements, stores are non-temporal.  Grid order = TilePlan::gorder.
struct XposeArgs {
    const char* src;
    char* dst;
    int32_t ng, pad;
    uint32_t ext[MAXN];     // extent of grid coordinate g (fastest first)
    i64 sstep[MAXN];        // byte step of the input / the destination per unit of grid coordinate g
    i64 dstep[MAXN];
    i64 src_row, dst_row;   // byte stride of the input along the destination's unit dim / of the destination along the input's unit dim
};

template <class T, class F, int V, int T0, int TQ>
__global__ void __launch_bounds__(1024) k_xpose_big(const XposeArgs a, F f) {
    typedef TVec<T, V> VT;
    constexpr int PITCH = TQ + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    T* lds = reinterpret_cast<T*>(smem_x);
    uint32_t b = blockIdx.x;
    i64 so = 0, dofs = 0;
#pragma unroll 1
    for (int g = 0; g < a.ng; ++g) {
        const uint32_t q = b / a.ext[g], r = b - q * a.ext[g];
        so += (i64)r * a.sstep[g];
        dofs += (i64)r * a.dstep[g];
        b = q;
    }
    const char* s0 = a.src + so;
    char* d0 = a.dst + dofs;
    const int tid = threadIdx.x;
    // load: lanes along the input's unit axis (TQ / V lanes per row), rows along the destination's unit dim
    constexpr int LPR = TQ / V, RPP = 1024 / LPR, NPASS = T0 / RPP;
    const int lc = tid % LPR, lr = tid / LPR;
    VT x[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) x[p] = *reinterpret_cast<const VT*>(s0 + (i64)(p * RPP + lr) * a.src_row + (i64)lc * (V * (int)sizeof(T)));
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        T* L = lds + (size_t)(p * RPP + lr) * PITCH + V * lc;  // row = position along the destination's unit dim
#pragma unroll
        for (int h = 0; h < V; ++h) L[h] = x[p].v[h];
    }
    __syncthreads();
    // store: lanes along the destination's unit dim (T0 / V lanes per row), rows along the input's unit axis
    constexpr int LPR2 = T0 / V, RPP2 = 1024 / LPR2, NPASS2 = TQ / RPP2;
    const int sc = tid % LPR2, sr = tid / LPR2;
#pragma unroll
    for (int p = 0; p < NPASS2; ++p) {
        const int iq = p * RPP2 + sr;
        VT out;
#pragma unroll
        for (int h = 0; h < V; ++h) {
            T arg[MAXIN];
#pragma unroll
            for (int i = 0; i < MAXIN; ++i) arg[i] = T{};
            arg[0] = lds[(size_t)(V * sc + h) * PITCH + iq];
            out.v[h] = f(arg);
        }
        store_vec_ct<true, VT>(d0 + (i64)iq * a.dst_row + (i64)sc * (V * (int)sizeof(T)), out);
    }
}

// SMR_OK: launched; SMR_EUNSUPPORTED (no error text set): not this case, the caller takes the general kernel
template <class T, class F>
static int try_xpose_big(const Plan& plan, hipStream_t s, F f, const OpTab& tab) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    constexpr int V = 16 / (int)sizeof(T) >= 1 ? 16 / (int)sizeof(T) : 1;
    constexpr int T0 = 128, TQ = 32;
    if constexpr (is_jit<F>::value || !tr<T>::arith || sizeof(T) < 8 || (F::NIN >= 0 && F::NIN != 1)) {
        return SMR_EUNSUPPORTED;
    } else {
        if (!options().tiled_xpose) return SMR_EUNSUPPORTED;
        // (rank 2 -- a plain matrix transpose, or a permutation that fuses to one -- stays with the general kernel: 16384^2 700 vs 708 us)
        if (c.M != 2 || c.mixed || c.N < 3 || c.N > MAXN || t.nt != 2 || t.tilelog != 12 || t.staged[1] < 0 || !t.ord.empty()) return SMR_EUNSUPPORTED;
        if (t.tdim[0] != 0 || t.tlog[0] != 7 || t.tlog[1] != 5) return SMR_EUNSUPPORTED;
        const int q = t.tdim[1];
        if (c.strides[0][0] != 1 || c.strides[1][q] != 1 || c.dims[0] % T0 || c.dims[q] % TQ) return SMR_EUNSUPPORTED;
        if (tab.conj[0] || tab.conj[1]) return SMR_EUNSUPPORTED;
        if (c.algbytes < ((i64)512 << 20)) return SMR_EUNSUPPORTED;
        const i64 es = (i64)sizeof(T);
        if ((((uintptr_t)tab.base[0]) | ((uintptr_t)tab.base[1])) % 16) return SMR_EUNSUPPORTED;
        for (int d = 0; d < c.N; ++d) {
            if (d != 0 && (c.strides[0][d] * es) % 16) return SMR_EUNSUPPORTED;
            if (d != q && (c.strides[1][d] * es) % 16) return SMR_EUNSUPPORTED;
        }
        XposeArgs a;
        std::memset(&a, 0, sizeof a);
        a.src = (const char*)tab.base[1];
        a.dst = (char*)tab.base[0];
        a.src_row = c.strides[1][0] * es;
        a.dst_row = c.strides[0][q] * es;
        i64 grid = 1;
        int ng = 0;
        for (int i = 0; i < c.N; ++i) {
            const int d = t.gorder[i] >= 0 && t.gorder[i] < c.N ? t.gorder[i] : -1;
            if (d < 0) return SMR_EUNSUPPORTED;
            const i64 tile = d == 0 ? T0 : (d == q ? TQ : 1);
            const i64 n = c.dims[d] / tile;
            if (n <= 1) continue;
            if (n > 0xffffffffLL) return SMR_EUNSUPPORTED;
            a.ext[ng] = (uint32_t)n;
            a.sstep[ng] = c.strides[1][d] * es * tile;
            a.dstep[ng] = c.strides[0][d] * es * tile;
            grid *= n;
            ++ng;
        }
        {   // gorder must be a permutation of the dims (every dim with more than one tile appears once)
            bool seen[MAXN] = {false};
            for (int i = 0; i < c.N; ++i) {
                if (seen[t.gorder[i]]) return SMR_EUNSUPPORTED;
                seen[t.gorder[i]] = true;
            }
        }
        a.ng = ng;
        if (grid < 1 || grid > 0x7fffffffLL) return SMR_EUNSUPPORTED;
        if (jit_no_launch()) return SMR_OK;  // (prepare mode: this form has no tables to build)
        const size_t lds = (size_t)T0 * (TQ + 2) * sizeof(T);
        auto kern = k_xpose_big<T, F, V, T0, TQ>;
        clear_sticky_error();
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
        }
        SMR_LAUNCH(kern, dim3((unsigned)grid), dim3(1024), lds, s, a, f);
        return check_launch("k_xpose_big");
    }
}

template <class T, class F, bool MIXED>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    const OpTab tab = make_optab(c, bases);
    if constexpr (!MIXED) {  // HBM-sized transposing copies: the lean form
        const int rc = try_xpose_big<T, F>(plan, s, f, tab);
        if (rc != SMR_EUNSUPPORTED) return rc;
    }
    // 32-bit within-tile byte offsets when every tiled stride is >= 0 and the tile spans < 4 GiB
    bool narrow = true;
    for (int k = 0; k < c.M && narrow; ++k) {
        long double span = 0;
        for (int j = 0; j < t.nt; ++j) {
            const i64 st = c.strides[k][t.tdim[j]];
            if (st < 0) narrow = false;
            span += (long double)st * (((i64)1 << t.tlog[j]) - 1) * c.esize[k];
        }
        if (span >= 4294967296.0L) narrow = false;
    }
    // always 4 elements per lane: 1024-element tiles on 256 lanes, 4096-element tiles on 1024 lanes.
    // Measured alternatives (32^4 f64): 2048/4096-element tiles on 256 lanes are 10-40 % slower.
    if (t.tilelog == 10) return go_tl<T, F, MIXED, 8>(plan, s, f, tab, narrow);
    if (t.tilelog == 12) return go_tl<T, F, MIXED, 10>(plan, s, f, tab, narrow);
    return set_error(SMR_EINVAL, "tiled: the planner must pick 1024- or 4096-element tiles");
}

template <>
int launch_tiled_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.bitcopy) {
#if SMR_CT == SMR_F32
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
    switch (c.fkind) {  // natively compiled functors of this family; everything else is compiled at run time
        case FK_IDENT: return go<T, FIdent<T>, false>(plan, bases, s, FIdent<T>{});
        case FK_ADD2: return go<T, FAdd2<T>, false>(plan, bases, s, FAdd2<T>{});
        case FK_ADD3: return go<T, FAdd3<T>, false>(plan, bases, s, FAdd3<T>{});
        case FK_ADD4: return go<T, FAdd4<T>, false>(plan, bases, s, FAdd4<T>{});
        case FK_SCALE: return go<T, FScale<T>, false>(plan, bases, s, FScale<T>{hostmk<T>(c.fc[0], c.fc[1])});
        case FK_SYM: return go<T, FSym<T>, false>(plan, bases, s, FSym<T>{hostmk<T>(c.fc[0], c.fc[1])});
        case FK_AXPY: return go<T, FAxpy<T>, false>(plan, bases, s, FAxpy<T>{hostmk<T>(c.fc[0], c.fc[1])});
        case FK_AXPBY:
            return go<T, FAxpby<T>, false>(plan, bases, s, FAxpby<T>{hostmk<T>(c.fc[0], c.fc[1]), hostmk<T>(c.fc[2], c.fc[3])});
        default: break;
    }
    return with_prog<T>(c, [&](auto f) { return go<T, decltype(f), false>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
