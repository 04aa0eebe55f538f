// smr_k_orbit.hip -- family ORBIT: fused N-ary map whose inputs are differently PERMUTED VIEWS OF ONE
// BUFFER (B .= (A .+ A')./2, the 4-way permuted sum of the reference's README, any symmetrisation).
//
// The classic tiled kernel (smr_k_tiled.hip) treats the views as independent operands: the buffer passes
// through L1 once per view, and with three or more distinct unit-stride axes most of those passes move
// 32-/64-byte runs.  Here the permutation group G the views generate (|G| <= 4) acts on the TILES: tile
// extents are equal along every cycle of G, so the image of a tile under g in G is again a tile.  A
// workgroup owns one orbit {g.t : g in G}:
//   * slot a of the LDS holds the buffer on tile g_a.t, loaded ONCE, in the buffer's natural order
//     (16 B per lane along the buffer's unit-stride axis -- every slot has the same shape, so one set
//     of per-lane offsets serves all of them);
//   * an output element of tile g_a.t reads view k from slot (pi_k o g_a) at the local coordinate
//     permuted by pi_k; the identity view comes straight from the lane's own registers;
//   * outputs leave in the same natural order (the destination has the identity view's strides).
// Every element of the buffer crosses L1 once instead of once per view, and an in-place update
// (destination = the buffer) is safe: an orbit is read completely before it is written, and orbits are
// disjoint.  Orbits are executed super-cell by super-cell in XCD-contiguous runs (smr_plan.cpp:
// plan_orbit), so the partner halves of 64-B runs meet in one XCD's L2.
// All index arithmetic is bit slicing of the lane id with kernel-argument shifts (no lane tables, no
// dependent memory round trip before the first global load).
// Measured on MI355X (4-way sum, Float64): 32^4 6.6 -> 4.7 us, 64^4 83 -> 46 us, 128^4 1.73 -> 1.41 ms
// (tools/c3_proto.hip is the design experiment this kernel follows).
#ifndef SMR_JIT
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3 or 7"
#endif

// 1: the per-lane offsets (byte offset in a tile + one LDS read index per permuted view) come from a small
// device table -- one vector load per lane, issued next to the origin row's scalar load -- instead of
// bit-slice arithmetic on kernel arguments (measured on the classic kernel: the table is the faster form)
#ifndef SMR_ORBIT_TABLE
#define SMR_ORBIT_TABLE 0  // measured: no difference (5.01 vs 5.00 us at 32^4) -> arithmetic, no table to upload
#endif

namespace smr {

constexpr int OMAXT = 4;  // tiled dims = dims of the unit class <= |G| <= 4

struct OrbitArgs {
    const char* src;  // the shared buffer, element offset applied
    char* dst;
    const uint32_t* list;  // per workgroup a row of 2 * NG words: the NG slot origins (element offsets; [0] = 0xffffffff: idle), then the
                           // 64-bit slot map -- field (g * 8 + k) * 2: the LDS slot an output of slot g reads input k from.  (The
                           // natively compiled kernels receive this pointer once more as their FIRST parameter: it is preloaded
                           // into SGPRs with the wave, and the row's load leaves before the kernel arguments have been fetched.)
    const uint32_t* lanetab;  // [(r * NT + tid) * rowlen]: byte offset, then the LDS read index of every non-own view
    int32_t nin, tilelog, ntlog, conj0, nts, nlist;  // nlist: rows of `list` (PIPE form)
    uint32_t swz_s1, swz_s2, swz_mask, pad1;
    // element enumeration inside a tile (natural order of the buffer); unused tiled dims have elen = 0
    int32_t esh[OMAXT], elen[OMAXT];
    uint32_t estride[OMAXT];       // byte stride of tiled dim j
    int32_t lsh[MAXIN][OMAXT];     // LDS bit position of tiled dim j as seen through view k
    uint32_t conjbit[MAXIN];       // 0x80000000 when view k is conjugated (complex types)
};

// What a wave needs before its first global load can leave: the natively compiled kernels receive these as LEADING SCALAR
// parameters, which gfx950 preloads into SGPRs with the wave (Makefile: -mllvm -amdgpu-kernarg-preload-count=16) -- the only
// memory round trip in front of the loads is then the origin row's.  (Left in the argument struct the compiler fetched them in
// two dependent batches of scalar loads: tools/orbit32_probe.hip, profiles/r06_orbit_phases.txt.)
struct OrbitHead {
    const uint32_t* list;  // = OrbitArgs::list
    const char* src;       // = OrbitArgs::src
    char* dst;             // = OrbitArgs::dst (left in the struct, its scalar load was sunk between the LDS reads and the adds)
    uint32_t eshp, elenp;  // esh[j] / elen[j] in byte j
    uint32_t estride[OMAXT];
    uint32_t ntlog;
};
static inline __host__ __device__ OrbitHead orbit_head(const OrbitArgs& a) {
    OrbitHead h;
    h.list = a.list;
    h.src = a.src;
    h.dst = a.dst;
    h.eshp = h.elenp = 0;
    for (int j = 0; j < OMAXT; ++j) {
        h.eshp |= (uint32_t)a.esh[j] << (8 * j);
        h.elenp |= (uint32_t)a.elen[j] << (8 * j);
        h.estride[j] = a.estride[j];
    }
    h.ntlog = (uint32_t)a.ntlog;
    return h;
}

template <class T, int V>
struct alignas(sizeof(T) * V) OVec {
    T v[V];
};

// conjugate iff bit == 0x80000000, without a branch (sign flip of the imaginary part)
SMR_DEV float ocj(float x, uint32_t) { return x; }
SMR_DEV double ocj(double x, uint32_t) { return x; }
SMR_DEV c32 ocj(c32 x, uint32_t bit) { return c32{x.re, __uint_as_float(__float_as_uint(x.im) ^ bit)}; }
SMR_DEV c64 ocj(c64 x, uint32_t bit) {
    return c64{x.re, __longlong_as_double(__double_as_longlong(x.im) ^ ((long long)bit << 32))};
}

// LDS access by absolute 32-bit address (base of the dynamic segment included), element as one 4- / 8- / 16-byte word: lets a kernel
// form every address of the exchange BEFORE its barrier (left to the compiler, base and scaling are added behind it, in front of the reads)
template <int BYTES> struct lds_word;
template <> struct lds_word<4> { typedef uint32_t type; };
template <> struct lds_word<8> { typedef unsigned long long type; };
template <> struct lds_word<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <class T>
SMR_DEV void lds_put(uint32_t addr, const T& v) {
    typedef typename lds_word<sizeof(T)>::type W;
    W w;
    __builtin_memcpy(&w, &v, sizeof(T));
    *(__attribute__((address_space(3))) W*)(uintptr_t)addr = w;
}
template <class T>
SMR_DEV T lds_get(uint32_t addr) {
    typedef typename lds_word<sizeof(T)>::type W;
    const W w = *(const __attribute__((address_space(3))) W*)(uintptr_t)addr;
    T v;
    __builtin_memcpy(&v, &w, sizeof(T));
    return v;
}

// NG = |G| (2 or 4; a group of order 3 is padded with a copy of slot 0), OWN0: view 0 is the identity view
// (its value is the lane's own register).  Apart from the rare > 4 grid dims there is no branch on a
// kernel argument before the stores: every scalar branch on a just-loaded argument is a serial
// scalar-cache round trip, and at 32^4 the whole launch lasts 4-5 us.
// PIPE: persistent form for orbits whose LDS footprint leaves one workgroup per CU -- the workgroup walks the
// list with stride gridDim.x (a multiple of 8: it stays on its XCD's run) and issues the loads of its next orbit
// right after the barrier, so that they fly during the exchange and the stores of the current one.
template <class T, class F, int V, int NREP, int NG, bool OWN0, bool PIPE = false>
SMR_DEV void orbit_map_body(const OrbitArgs a, const OrbitHead h, F f) {
    typedef OVec<T, V> VT;
    constexpr int NIN_STATIC = F::NIN;
    constexpr int NINMAX = (NIN_STATIC >= 0) ? NIN_STATIC : MAXIN;
    constexpr int NK = NINMAX > 0 ? NINMAX : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int nin = (NIN_STATIC >= 0) ? NIN_STATIC : a.nin;
    const uint32_t tid = threadIdx.x;
    // ---- slot origins: one wide scalar load of this workgroup's table row -----------------------------
    i64 org[NG];
    uint64_t smap;
    bool live;
    auto load_row = [&](uint32_t b, i64(&og)[NG], uint64_t& mp, bool& lv) {
        typedef uint32_t rowv __attribute__((ext_vector_type(2 * NG)));
        const rowv row = reinterpret_cast<const rowv*>(h.list)[b];
        uint32_t o32[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) o32[g] = row[g];
        mp = (uint64_t)row[NG] | ((uint64_t)row[NG + 1] << 32);
        // a padding workgroup (first word 0xffffffff) runs on slot origins 0 and only skips its stores: an early
        // exit here would keep every other kernel-argument load behind this row's round trip
        lv = o32[0] != 0xffffffffu;
#pragma unroll
        for (int g = 0; g < NG; ++g) og[g] = lv ? (i64)o32[g] * (i64)sizeof(T) : 0;
    };
    load_row(blockIdx.x, org, smap, live);

    // ---- per-lane byte offsets inside a tile + LDS read indices -----------------------------------------
    constexpr int NLR = NK - (OWN0 ? 1 : 0);      // views read from LDS
    constexpr int ROWLEN = (1 + NLR) <= 2 ? 2 : ((1 + NLR) <= 4 ? 4 : 8);
    (void)ROWLEN;
    uint32_t goff[NREP];
    uint32_t lr[NK][NREP];  // LDS index of the lane's first element seen through view k
#if SMR_ORBIT_TABLE
    {
        typedef uint32_t trow __attribute__((ext_vector_type(ROWLEN)));
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const trow t = reinterpret_cast<const trow*>(a.lanetab)[((uint32_t)r << h.ntlog) | tid];
            goff[r] = t[0];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                lr[k][r] = 0;
                if (!(OWN0 && k == 0)) lr[k][r] = t[1 + k - (OWN0 ? 1 : 0)];
            }
        }
    }
#else
    uint32_t cj[NREP][OMAXT];
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        const uint32_t e = (((uint32_t)r << h.ntlog) | tid) * V;
        uint32_t g = 0;
#pragma unroll
        for (int j = 0; j < OMAXT; ++j) {
            cj[r][j] = __builtin_amdgcn_ubfe(e, (h.eshp >> (8 * j)) & 0xffu, (h.elenp >> (8 * j)) & 0xffu);
            g += cj[r][j] * h.estride[j];
        }
        goff[r] = g;
    }
#endif

    // ---- load every slot (natural order) --------------------------------------------------------------
    VT x[NG][NREP];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < NREP; ++r) x[g][r] = *reinterpret_cast<const VT*>(h.src + org[g] + goff[r]);

    // ---- while the loads fly: everything the exchange needs from the kernel arguments ---------------
    // (left to itself the compiler sinks these scalar loads below the barrier: one more serial scalar-memory
    // round trip in a 5 us launch; the empty asm statements pin the values in registers here)
#if !SMR_ORBIT_TABLE
#pragma unroll
    for (int r = 0; r < NREP; ++r)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            uint32_t l = 0;
            if (!(OWN0 && k == 0)) {
#pragma unroll
                for (int j = 0; j < OMAXT; ++j) l |= cj[r][j] << a.lsh[k][j];
            }
            lr[k][r] = l;
        }
#endif
    uint32_t swz_s1 = a.swz_s1, swz_s2 = a.swz_s2, swz_mask = a.swz_mask;
    asm volatile("" : "+s"(swz_s1), "+s"(swz_s2), "+s"(swz_mask));
    uint32_t sbase[NG][NK], hbit[NK], cbit[NK];
    uint32_t tilelog = (uint32_t)a.tilelog;
    asm volatile("" : "+s"(tilelog));
    // the LDS slot every (output slot, view) pair reads from: fields of this workgroup's row (orbits that share a workgroup have
    // their own little maps: smr_plan.cpp, plan_orbit)
    auto slot_bases = [&](uint64_t mp) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                sbase[g][k] = ((uint32_t)(mp >> ((g * 8 + k) * 2)) & 3u) << tilelog;
                asm volatile("" : "+s"(sbase[g][k]));
            }
    };
    slot_bases(smap);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        hbit[k] = (uint32_t)a.lsh[k][0];
        cbit[k] = a.conjbit[k];
        asm volatile("" : "+s"(hbit[k]), "+s"(cbit[k]));
#pragma unroll
        for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(lr[k][r]));
    }
    uint32_t conj0 = a.conj0 ? 0x80000000u : 0u, nts_flag = (uint32_t)a.nts;
    asm volatile("" : "+s"(conj0), "+s"(nts_flag));
    auto swz = [&](uint32_t i) { return i ^ (((i >> swz_s1) ^ (i >> swz_s2)) & swz_mask); };
    // one-shot form: the swizzled LDS indices of the lane's writes and of its transposing reads are formed HERE, while the global
    // loads fly (left alone the compiler computes them behind the barrier: ~45 vector instructions on the critical path of a
    // launch whose waves live 2 us).  The persistent form recomputes them per orbit (its registers hold two orbits).
    constexpr int NPRE = PIPE ? 1 : NREP;
    uint32_t wi[NPRE][V], ri[NK][NPRE][V];
    if constexpr (!PIPE) {
#pragma unroll
        for (int r = 0; r < NREP; ++r)
#pragma unroll
            for (int hh = 0; hh < V; ++hh) {
                wi[r][hh] = swz(((((uint32_t)r << h.ntlog) | tid) * V) + hh);
                asm volatile("" : "+v"(wi[r][hh]));
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    ri[k][r][hh] = 0;
                    if (!(OWN0 && k == 0)) {
                        ri[k][r][hh] = swz(lr[k][r] | ((uint32_t)hh << hbit[k]));
                        asm volatile("" : "+v"(ri[k][r][hh]));
                    }
                }
            }
    }

    // one-shot form with few addresses (the 4-way sum: 24 of reads, 8 of writes): whole LDS addresses, formed here (round 6: the
    // base and the scaling used to be added behind the barrier -- ~50 vector instructions in front of the transposing reads)
    constexpr bool ABS = !PIPE && (NG * NLR * NREP * V <= 32) && (sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16);
    uint32_t pa[ABS ? NG : 1][NREP][V], la[ABS ? NG : 1][NK][NREP][V];
    if constexpr (ABS) {
        const uint32_t lbase = (uint32_t)(uintptr_t)smem_raw;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < NREP; ++r)
#pragma unroll
                for (int hh = 0; hh < V; ++hh) {
                    pa[g][r][hh] = lbase + ((((uint32_t)g) << tilelog) + wi[r][hh]) * (uint32_t)sizeof(T);
                    asm volatile("" : "+v"(pa[g][r][hh]));
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        la[g][k][r][hh] = 0;
                        if (!(OWN0 && k == 0)) {
                            la[g][k][r][hh] = lbase + (sbase[g][k] + ri[k][r][hh]) * (uint32_t)sizeof(T);
                            asm volatile("" : "+v"(la[g][k][r][hh]));
                        }
                    }
                }
    }

    uint32_t wg = blockIdx.x, tidp = tid;
    for (;;) {
        if constexpr (PIPE) {
            // loop-invariant LDS addresses (one per view x repeat x slot) would be hoisted and spill: recompute them
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(lr[k][r]));
#pragma unroll
            for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(goff[r]));
            asm volatile("" : "+v"(tidp));
        }
        // the NEXT orbit's origin row is requested before the slots are parked (unconditionally, on a clamped index:
        // no branch in front of the LDS stores), so that it has arrived when the barrier opens and the next loads
        // leave right behind it
        i64 norg[NG];
        uint64_t nmap = 0;
        bool nlive = false, more = false;
        if constexpr (PIPE) {
            wg += gridDim.x;
            more = wg < (uint32_t)a.nlist;
            load_row(more ? wg : (uint32_t)a.nlist - 1u, norg, nmap, nlive);
        }
        // ---- park the slots in LDS ---------------------------------------------------------------------------
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            T* L = lds + ((size_t)g << a.tilelog);
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                const uint32_t e = (((uint32_t)r << h.ntlog) | tidp) * V;
#pragma unroll
                for (int hh = 0; hh < V; ++hh) {
                    if constexpr (PIPE) L[swz(e + hh)] = x[g][r].v[hh];
                    else if constexpr (ABS) lds_put<T>(pa[g][r][hh], x[g][r].v[hh]);
                    else L[wi[r][hh]] = x[g][r].v[hh];
                }
            }
        }
        __syncthreads();
        VT xn[NG][NREP];
        if constexpr (PIPE) {
            if (more) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < NREP; ++r) xn[g][r] = *reinterpret_cast<const VT*>(h.src + norg[g] + goff[r]);
            }
        }

        // ---- outputs of every slot: all LDS reads of a repeat are issued before the first use ----------------
        // (the one-shot form batches the reads of all slots: one LDS latency per repeat; the persistent form has the
        // next orbit's loads in registers as well and batches per slot)
        constexpr int GB = (PIPE && NREP > 1) ? 1 : NG;
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
#pragma unroll
            for (int g0 = 0; g0 < NG; g0 += GB) {
                T val[GB][V][NK];
#pragma unroll
                for (int gi = 0; gi < GB; ++gi)
#pragma unroll
                    for (int h = 0; h < V; ++h)
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            const int g = g0 + gi;
                            if (OWN0 && k == 0) {
                                val[gi][h][k] = x[g][r].v[h];
                            } else {
                                // sub-element h moves along tiled dim 0 of the natural order
                                if constexpr (PIPE) {
                                    const uint32_t idx = lr[k][r] | ((uint32_t)h << hbit[k]);
                                    val[gi][h][k] = lds[sbase[g][k] + swz(idx)];
                                } else if constexpr (ABS) {
                                    val[gi][h][k] = lds_get<T>(la[g][k][r][h]);
                                } else {
                                    val[gi][h][k] = lds[sbase[g][k] + ri[k][r][h]];
                                }
                            }
                        }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads together: one LDS latency per batch, not one per output
#pragma unroll
                for (int gi = 0; gi < GB; ++gi) {
                    const int g = g0 + gi;
                    VT out;
#pragma unroll
                    for (int h = 0; h < V; ++h) {
                        T arg[MAXIN];
#pragma unroll
                        for (int k = 0; k < MAXIN; ++k) {
                            arg[k] = T{};
                            if (k < NK) {
                                T v = val[gi][h][k < NK ? k : 0];
                                if constexpr (tr<T>::cx) v = ocj(v, cbit[k < NK ? k : 0]);
                                arg[k] = v;
                            }
                        }
                        (void)nin;
                        T o = f(arg);
                        if constexpr (tr<T>::cx) o = ocj(o, conj0);
                        out.v[h] = o;
                    }
                    x[g][r] = out;
                }
                if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);  // one slot at a time: the registers hold two orbits
            }
        }
        if constexpr (!PIPE) {
            if (!live) return;
        }
        const bool nts = nts_flag != 0;
        if (!PIPE || live) {
            if (nts_flag == 2) {  // agent-scope write-through ("self-released" launch: smr_device.h)
                if constexpr (has_wt_store<VT>::value) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < NREP; ++r) store_vec_wt<VT>(h.dst + org[g] + goff[r], x[g][r]);
                }
            } else if (nts) {
                nt_block_guard();
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < NREP; ++r) store_vec_ct<true, VT>(h.dst + org[g] + goff[r], x[g][r]);
                nt_block_guard();
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < NREP; ++r) store_vec_ct<false, VT>(h.dst + org[g] + goff[r], x[g][r]);
            }
        }
        if constexpr (!PIPE) {
            if (nts_flag == 2) self_release_wait();
            return;
        } else {
            if (!more) {
                if (nts_flag == 2) self_release_wait();
                return;
            }
            __syncthreads();  // every lane has read its LDS values: the slots may be overwritten
            live = nlive;
            slot_bases(nmap);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                org[g] = norg[g];
#pragma unroll
                for (int r = 0; r < NREP; ++r) x[g][r] = xn[g][r];
            }
        }
    }
}

// ---- PAIR form (round 6): 4^4 cubes of 8-byte elements, |G| = 4, identity view first ---------------------------------------------
// tools/orbit16_probe.hip: the memory system serves this pattern by REQUESTS -- one per 32-byte run of a cube (TCP_TCC_READ_REQ =
// 262144 for the 8 MiB of the 4-way sum at 32^4, a quarter of what a line-wide kernel needs per request) -- and a lane quad that
// covers two neighbouring cubes' halves of one 64-byte run makes one request of two.  A workgroup of 256 lanes owns TWO slot sets
// (plan_orbit: where it can, set 1 is the orbit of set 0's unit-axis neighbour); lane bit 1 selects the set, so slot 0 of both moves
// as 64-byte runs -- 4.29 -> 4.13 us warm, 6.5 -> 5.9 us from HBM (the 16-cube form with two paired slots loses: one workgroup
// per CU waits for ALL its loads before the first exchange).  Every cube has its own 2-KiB LDS region; the in-region index folds its
// three high bits into bank bits 1..3 and set 1's regions are XORed with 17: the 32 lanes of a half-wave then hit 32 distinct
// 8-byte banks in the parking writes and in every transposing read (16 lanes of a set stay inside {bit 0 ^ bit 4 = const}).
// Table: per workgroup 4 entries (one per slot j) of 8 words -- origin of set 0 / set 1 (element offsets; first word 0xffffffff:
// idle), the own region words (16 bits per set: region << 8 | parity * 17), then for views 1..3 the region words of the cubes an
// output of slot j reads that view from.
template <class T, class F, int SMODE = -1>  // SMODE: store policy fixed at compile time (0 plain, 1 non-temporal, 2 write-through), -1: from the arguments
SMR_DEV void orbit_pair_body(const OrbitArgs a, const OrbitHead h, F f) {
    constexpr int V = 2;
    typedef OVec<T, V> VT;
    constexpr int NK = F::NIN;
    static_assert(NK >= 2 && NK <= 4 && sizeof(T) == 8, "PAIR form: 2..4 views of 8-byte elements");
    typedef uint32_t entv __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t b = (tid >> 1) & 1u;
    const uint32_t u = (tid & 1u) | ((tid >> 2) << 1);  // the lane's place in its cube (7 bits)
    const uint32_t e = u * V;
    entv ent[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ent[j] = reinterpret_cast<const entv*>(h.list)[blockIdx.x * 4u + (uint32_t)j];
    const bool live = ent[0][0] != 0xffffffffu;
    uint32_t cj[OMAXT], goff = 0;
#pragma unroll
    for (int j = 0; j < OMAXT; ++j) {
        cj[j] = __builtin_amdgcn_ubfe(e, (h.eshp >> (8 * j)) & 0xffu, (h.elenp >> (8 * j)) & 0xffu);
        goff += cj[j] * h.estride[j];
    }
    i64 org[4];
    VT x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t o32 = live ? (b ? ent[j][1] : ent[j][0]) : 0u;
        org[j] = (i64)o32 * (i64)sizeof(T);
        x[j] = *reinterpret_cast<const VT*>(h.src + org[j] + goff);
    }
    // ---- while the loads fly: LDS indices ----------------------------------------------------------------------------------------
    auto swz = [](uint32_t i) { return i ^ (((i >> 5) & 7u) << 1); };
    const uint32_t sh = b * 16u;
    uint32_t wi[V], ri[NK][V], own[4], src[4][NK];
#pragma unroll
    for (int hh = 0; hh < V; ++hh) wi[hh] = swz(e + (uint32_t)hh);
#pragma unroll
    for (int k = 1; k < NK; ++k) {
        uint32_t l = 0;
#pragma unroll
        for (int j = 0; j < OMAXT; ++j) l |= cj[j] << a.lsh[k][j];
#pragma unroll
        for (int hh = 0; hh < V; ++hh) ri[k][hh] = swz(l | ((uint32_t)hh << a.lsh[k][0]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        own[j] = (ent[j][2] >> sh) & 0xffffu;
#pragma unroll
        for (int k = 1; k < NK; ++k) src[j][k] = (ent[j][2 + k] >> sh) & 0xffffu;
    }
    uint32_t cbit[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) cbit[k] = a.conjbit[k];
    uint32_t conj0 = a.conj0 ? 0x80000000u : 0u, nts_flag = (uint32_t)a.nts;
    // everything above is pinned in registers HERE, while the global loads fly (left alone the compiler fetches the kernel arguments
    // it needs behind the barrier: two serial scalar-memory round trips in a launch whose waves live 2 us)
    asm volatile("" : "+s"(conj0), "+s"(nts_flag));
#pragma unroll
    for (int k = 0; k < NK; ++k) asm volatile("" : "+s"(cbit[k]));
    // the LDS addresses themselves (24 of reads, 8 of writes): formed here, not behind the barrier -- whole 32-bit LDS addresses (the
    // base of the dynamic segment included: added behind the barrier it cost 24 more vector instructions in front of the reads)
    const uint32_t lbase = (uint32_t)(uintptr_t)smem_raw;
    uint32_t wa[4][V], ra[4][V][NK];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < V; ++hh) {
            wa[j][hh] = lbase + (own[j] ^ wi[hh]) * (uint32_t)sizeof(T);
            asm volatile("" : "+v"(wa[j][hh]));
#pragma unroll
            for (int k = 1; k < NK; ++k) {
                ra[j][hh][k] = lbase + (src[j][k] ^ ri[k][hh]) * (uint32_t)sizeof(T);
                asm volatile("" : "+v"(ra[j][hh][k]));
            }
        }
    // ---- park, exchange ----------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < V; ++hh) lds_put<T>(wa[j][hh], x[j].v[hh]);
    __syncthreads();
    T val[4][V][NK];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < V; ++hh) {
            val[j][hh][0] = x[j].v[hh];
#pragma unroll
            for (int k = 1; k < NK; ++k) val[j][hh][k] = lds_get<T>(ra[j][hh][k]);
        }
    __builtin_amdgcn_sched_barrier(0);  // all LDS reads issued before the first use: one LDS latency, then per slot adds and its store
    if (!live) return;
    auto outputs = [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            VT out;
#pragma unroll
            for (int hh = 0; hh < V; ++hh) {
                T arg[MAXIN];
#pragma unroll
                for (int k = 0; k < MAXIN; ++k) {
                    arg[k] = T{};
                    if (k < NK) {
                        T v = val[j][hh][k < NK ? k : 0];
                        if constexpr (tr<T>::cx) v = ocj(v, cbit[k < NK ? k : 0]);
                        arg[k] = v;
                    }
                }
                T o = f(arg);
                if constexpr (tr<T>::cx) o = ocj(o, conj0);
                out.v[hh] = o;
            }
            char* p = h.dst + org[j] + goff;
            if constexpr (MODE == 2) {
                store_vec_wt<VT>(p, out);
            } else if constexpr (MODE == 1) {
                nt_block_guard();
                store_vec_ct<true, VT>(p, out);
                nt_block_guard();
            } else {
                store_vec_ct<false, VT>(p, out);
            }
        }
    };
    if constexpr (SMODE == 0) {
        outputs(IntC<0>{});
    } else if constexpr (SMODE == 1) {
        outputs(IntC<1>{});
    } else if constexpr (SMODE == 2) {
        if constexpr (has_wt_store<VT>::value) outputs(IntC<2>{});
        self_release_wait();
    } else if (nts_flag == 2) {
        if constexpr (has_wt_store<VT>::value) outputs(IntC<2>{});
        self_release_wait();
    } else if (nts_flag) {
        outputs(IntC<1>{});
    } else {
        outputs(IntC<0>{});
    }
}

#ifndef SMR_JIT
template <class T, class F, int SMODE>
__global__ void __launch_bounds__(256) k_orbit_pair(const uint32_t* list, const char* src, char* dst, uint32_t eshp, uint32_t elenp, uint32_t es0, uint32_t es1,
                                                    uint32_t es2, uint32_t es3, uint32_t ntlog, const OrbitArgs a, F f SMR_STAMP_PARAM) {
    SMR_STAMP_BEGIN
    OrbitHead h;
    h.list = list;
    h.src = src;
    h.dst = dst;
    h.eshp = eshp;
    h.elenp = elenp;
    h.estride[0] = es0;
    h.estride[1] = es1;
    h.estride[2] = es2;
    h.estride[3] = es3;
    h.ntlog = ntlog;
    orbit_pair_body<T, F, SMODE>(a, h, f);
    SMR_STAMP_END
}

template <class T, class F, int V, int NREP, int NG, bool OWN0, bool PIPE>
__global__ void __launch_bounds__(1024) k_orbit_map(const uint32_t* list, const char* src, char* dst, uint32_t eshp, uint32_t elenp, uint32_t es0, uint32_t es1,
                                                    uint32_t es2, uint32_t es3, uint32_t ntlog, const OrbitArgs a, F f SMR_STAMP_PARAM) {
    // the leading scalars = OrbitHead, preloaded into SGPRs: the origin row's load is the first thing the wave does, and the global
    // loads leave as soon as it is back (tools/orbit32_probe.hip: 4.76 -> 4.45 us at 32^4)
    SMR_STAMP_BEGIN
    OrbitHead h;
    h.list = list;
    h.src = src;
    h.dst = dst;
    h.eshp = eshp;
    h.elenp = elenp;
    h.estride[0] = es0;
    h.estride[1] = es1;
    h.estride[2] = es2;
    h.estride[3] = es3;
    h.ntlog = ntlog;
    orbit_map_body<T, F, V, NREP, NG, OWN0, PIPE>(a, h, f);
    SMR_STAMP_END
}

// XOR-fold swizzle l ^ (((l >> s1) ^ (l >> s2)) & mask): parameters picked per plan by counting the
// bank conflicts of the transposing LDS reads (a lane group of the hardware must hit distinct banks)
struct OSwz {
    uint32_t s1 = 31, s2 = 31, mask = 0;
    uint32_t operator()(uint32_t i) const { return i ^ (((i >> s1) ^ (i >> s2)) & mask); }
};

static OSwz choose_orbit_swizzle(const OrbitArgs& a, int nin, bool own0, int esize, int V, int NREP) {
    // lane group / bank-slot model: ds_read_b32/b64 are served in two halves of 32 lanes over 32 slots
    // of the access width; ds_read_b128 in groups of 16 lanes over 16 slots
    const int group = esize >= 16 ? 16 : 32;
    const uint32_t slots = (uint32_t)group;
    const uint32_t nt = 1u << a.ntlog;
    auto cost = [&](const OSwz& s) {
        long c = 0;
        for (int k = own0 ? 1 : 0; k < nin; ++k)
            for (int r = 0; r < NREP; ++r)
                for (int h = 0; h < V; ++h)
                    for (uint32_t w0 = 0; w0 < std::min(nt, 256u); w0 += group) {
                        uint32_t first[32];
                        int cnt[32];
                        int worst = 0;
                        for (uint32_t q = 0; q < slots; ++q) cnt[q] = 0;
                        for (int lane = 0; lane < group; ++lane) {
                            const uint32_t e = (((uint32_t)r << a.ntlog) | (w0 + lane)) * V + h;
                            uint32_t idx = 0;
                            for (int j = 0; j < OMAXT; ++j) idx |= ((e >> a.esh[j]) & ((1u << a.elen[j]) - 1u)) << a.lsh[k][j];
                            const uint32_t l = s(idx), q = l % slots;
                            if (cnt[q] == 0) {
                                first[q] = l;
                                cnt[q] = 1;
                            } else if (first[q] != l) {
                                ++cnt[q];
                            }
                        }
                        for (uint32_t q = 0; q < slots; ++q) worst = std::max(worst, cnt[q]);
                        c += worst - 1;
                    }
        return c;
    };
    OSwz best;
    long bc = cost(best);
    auto consider = [&](uint32_t s1, uint32_t s2) {
        OSwz t;
        t.s1 = s1;
        t.s2 = s2;
        t.mask = slots - 1;
        const long cc = cost(t);
        if (cc < bc) {
            bc = cc;
            best = t;
        }
    };
    for (uint32_t s1 = 2; s1 <= 6 && bc > 0; ++s1) {
        consider(s1, 31);  // one-term fold
        for (uint32_t s2 = s1 + 1; s2 <= 13 && bc > 0; ++s2) consider(s1, s2);
    }
    if (std::getenv("SMR_DEBUG_SWIZZLE"))
        std::fprintf(stderr, "[smr] orbit swizzle: s1=%u s2=%u mask=%u conflict cost %ld\n", best.s1, best.s2, best.mask, bc);
    return best;
}

template <class T, class F, int V, int NREP, int NG, bool OWN0, bool PIPE>
static int go4(const Plan& plan, hipStream_t s, F f, const OpTab& tab) {
    const Canon& c = plan.c;
    const OrbitPlan& o = plan.orbit;
    OrbitArgs a;
    std::vector<unsigned char>& cached = plan.tiled_args[V > 1 ? 1 : 0];
    if (cached.size() == sizeof a) {
        std::memcpy(&a, cached.data(), sizeof a);
    } else {
        std::memset(&a, 0, sizeof a);
        const i64 es = (i64)sizeof(T);
        a.nin = c.M - 1;
        a.tilelog = o.tilelog;
        int vlog = 0, nreplog = 0;
        while ((1 << vlog) < V) ++vlog;
        while ((1 << nreplog) < NREP) ++nreplog;
        a.ntlog = o.tilelog - vlog - nreplog;
        a.conj0 = tab.conj[0];
        // tiled dims in canonical (= natural) order
        int nt = 0, sh = 0, jof[MAXN];
        for (int d = 0; d < c.N; ++d) {
            jof[d] = -1;
            if (o.lg[d] == 0) continue;
            if (nt >= OMAXT) return set_error(SMR_EUNSUPPORTED, "orbit: too many tiled dims");
            jof[d] = nt;
            a.esh[nt] = sh;
            a.elen[nt] = o.lg[d];
            a.estride[nt] = (uint32_t)(c.strides[o.k0][d] * es);
            sh += o.lg[d];
            ++nt;
        }
        for (int k = 1; k < c.M; ++k) {
            for (int d = 0; d < c.N; ++d)
                if (jof[d] >= 0) a.lsh[k - 1][jof[d]] = a.esh[jof[o.pdim[k][d]]];
            a.conjbit[k - 1] = tab.conj[k] ? 0x80000000u : 0u;
        }
        if (o.nslots != NG) return set_error(SMR_EINVAL, "orbit: slot count mismatch");
        const OSwz sw = choose_orbit_swizzle(a, c.M - 1, OWN0, (int)sizeof(T), V, NREP);
        a.swz_s1 = sw.s1;
        a.swz_s2 = sw.s2;
        a.swz_mask = sw.mask;
        if (!plan.ordtab && !jit_dry_run()) {
            // one row per workgroup: the NG slot origins (element offset of the tile in each slot), then the slot map (plan_orbit)
            const size_t nwg = o.wmap.size();
            std::vector<uint32_t> rows(nwg * 2 * NG, 0u);
            for (size_t w = 0; w < nwg; ++w) {
                uint32_t* row = &rows[w * 2 * NG];
                if (o.wtile[w * NG] == 0xffffffffu) {
                    row[0] = 0xffffffffu;
                    continue;
                }
                for (int g = 0; g < NG; ++g) {
                    i64 id = o.wtile[w * NG + g], org = 0;
                    for (int d = 0; d < c.N; ++d) {
                        org += (id % o.ntiles[d]) * (c.strides[o.k0][d] << o.lg[d]);
                        id /= o.ntiles[d];
                    }
                    row[g] = (uint32_t)org;
                }
                row[NG] = (uint32_t)o.wmap[w];
                row[NG + 1] = (uint32_t)(o.wmap[w] >> 32);
            }
            void* dptr = nullptr;
            hipError_t e = hipMalloc(&dptr, rows.size() * sizeof(uint32_t));
            if (e != hipSuccess) return hip_error(e, "hipMalloc(orbit origins)");
            e = hipMemcpy(dptr, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(dptr);
                return hip_error(e, "hipMemcpy(orbit origins)");
            }
            plan.ordtab = dptr;
        }
        a.list = reinterpret_cast<const uint32_t*>(plan.ordtab);
        if (!plan.lanetab[V > 1 ? 1 : 0] && !jit_dry_run()) {
            // per-lane rows: {byte offset in a tile, LDS read index of every view read from LDS}
            const int nk = is_jit<F>::value ? c.M - 1 : ((F::NIN >= 0) ? F::NIN : MAXIN);  // = NK of the device functor
            const int nlr = nk - (OWN0 ? 1 : 0);
            const int rowlen = (1 + nlr) <= 2 ? 2 : ((1 + nlr) <= 4 ? 4 : 8);
            const uint32_t nt = 1u << a.ntlog;
            std::vector<uint32_t> rows((size_t)NREP * nt * rowlen, 0u);
            for (uint32_t r = 0; r < (uint32_t)NREP; ++r)
                for (uint32_t tid = 0; tid < nt; ++tid) {
                    const uint32_t e = ((r << a.ntlog) | tid) * V;
                    uint32_t* row = &rows[((size_t)(r << a.ntlog) | tid) * rowlen];
                    for (int j = 0; j < OMAXT; ++j) {
                        const uint32_t cj = (e >> a.esh[j]) & ((1u << a.elen[j]) - 1u);
                        row[0] += cj * a.estride[j];
                        for (int k = OWN0 ? 1 : 0; k < nk && k < c.M - 1; ++k) row[1 + k - (OWN0 ? 1 : 0)] |= cj << a.lsh[k][j];
                    }
                }
            void* dptr = nullptr;
            hipError_t e2 = hipMalloc(&dptr, rows.size() * sizeof(uint32_t));
            if (e2 != hipSuccess) return hip_error(e2, "hipMalloc(orbit lane table)");
            e2 = hipMemcpy(dptr, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e2 != hipSuccess) {
                (void)hipFree(dptr);
                return hip_error(e2, "hipMemcpy(orbit lane table)");
            }
            plan.lanetab[V > 1 ? 1 : 0] = dptr;
        }
        a.lanetab = reinterpret_cast<const uint32_t*>(plan.lanetab[V > 1 ? 1 : 0]);
        if (!jit_dry_run()) {
            cached.resize(sizeof a);
            std::memcpy(cached.data(), &a, sizeof a);
        }
    }
    a.src = (const char*)tab.base[o.k0];
    a.dst = (char*)tab.base[0];
    const Options& opt = options();
    // automatic: only tiles that write whole 128-byte lines (symmetrise 4000^2: 40.2 -> 38.5 us); the 32-/64-byte
    // runs of the 4-D orbits rely on line partners meeting in L2 and get slower (4.8 -> 6.3 us at 32^4)
    a.nts = (opt.nt_store > 0 || (opt.nt_store < 0 && plan.c.strides[0][0] == 1 && (sizeof(T) << o.lg[0]) >= 128)) ? 1 : 0;
    // write-through stores: forced (nt_store = 2), or a launch recorded for a sequence that fits the caches several times over
    // (want_self_release): its packet then needs no release fence
    const bool wt = has_wt_store<OVec<T, V>>::value && (opt.nt_store == 2 || want_self_release(plan));
    if (wt) a.nts = 2;
    const unsigned block = 1u << a.ntlog;
    size_t lds = (size_t)NG * (sizeof(T) << o.tilelog);
    if (opt.orbit_lds_min > 0) lds = std::max(lds, (size_t)opt.orbit_lds_min);  // experiment: fewer resident workgroups per CU
    a.nlist = (int32_t)o.wmap.size();
    unsigned grid = (unsigned)o.wmap.size();
    if (PIPE) {
        // as many workgroups as the machine holds at once, a multiple of 8 so that a workgroup stays on its XCD's run
        static const int ncu = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            (void)hipGetLastError();
            return n;
        }();
        const unsigned cap = (unsigned)ncu * (unsigned)std::max<size_t>(1, (160 * 1024) / lds) / 8u * 8u;
        if (cap >= 8) grid = std::min<unsigned>(grid, cap);
        if (opt.orbit_wgs >= 8) grid = std::min<unsigned>(grid, (unsigned)opt.orbit_wgs / 8u * 8u);
    }
    if constexpr (is_jit<F>::value) {
        JitLaunch l;
        l.family = "orbit";
        l.tname = tname<T>();
        l.argtype = "smr::OrbitArgs";
        l.entry = std::string("smr::orbit_map_body<") + tname<T>() + ", smr::FJit, " + std::to_string(V) + ", " + std::to_string(NREP) + ", " +
                  std::to_string(NG) + ", " + (OWN0 ? "true" : "false") + ", " + (PIPE ? "true" : "false") + ">(a, smr::orbit_head(a), smr::FJit{kc});";
        l.grid = grid;
        l.block = block;
        l.lds = lds;
        l.args = &a;
        l.argsize = sizeof a;
        return jit_launch(c, l, s);
    } else {
        if (jit_no_launch()) return SMR_OK;
        clear_sticky_error();
        auto kern = k_orbit_map<T, F, V, NREP, NG, OWN0, PIPE>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
        }
        if (!PIPE) mark_sliceable(2, 0u, (unsigned)(2 * NG * sizeof(uint32_t)));  // one table row per workgroup; the pointer is parameter 0
        if (a.nts == 2) mark_self_released();
        const OrbitHead h = orbit_head(a);
        SMR_LAUNCH(kern, dim3(grid), dim3(block), lds, s, h.list, h.src, h.dst, h.eshp, h.elenp, h.estride[0], h.estride[1], h.estride[2], h.estride[3], h.ntlog, a,
                   f SMR_STAMP_ARG(grid, block));
        return check_launch("k_orbit_map");
    }
}

// PAIR form (orbit_pair_body): builds the 8-word entries once per plan, launches 256 lanes per two slot sets
template <class T, class F>
static int go_pair(const Plan& plan, hipStream_t s, F f, const OpTab& tab) {
    const Canon& c = plan.c;
    const OrbitPlan& o = plan.orbit;
    constexpr int NS = 4, V = 2;
    OrbitArgs a;
    std::vector<unsigned char>& cached = plan.tiled_args[2];
    if (cached.size() == sizeof a) {
        std::memcpy(&a, cached.data(), sizeof a);
    } else {
        std::memset(&a, 0, sizeof a);
        const i64 es = (i64)sizeof(T);
        a.nin = c.M - 1;
        a.tilelog = o.tilelog;
        a.ntlog = 8;
        a.conj0 = tab.conj[0];
        int nt = 0, sh = 0, jof[MAXN];
        for (int d = 0; d < c.N; ++d) {
            jof[d] = -1;
            if (o.lg[d] == 0) continue;
            if (nt >= OMAXT) return set_error(SMR_EUNSUPPORTED, "orbit: too many tiled dims");
            jof[d] = nt;
            a.esh[nt] = sh;
            a.elen[nt] = o.lg[d];
            a.estride[nt] = (uint32_t)(c.strides[o.k0][d] * es);
            sh += o.lg[d];
            ++nt;
        }
        for (int k = 1; k < c.M; ++k) {
            for (int d = 0; d < c.N; ++d)
                if (jof[d] >= 0) a.lsh[k - 1][jof[d]] = a.esh[jof[o.pdim[k][d]]];
            a.conjbit[k - 1] = tab.conj[k] ? 0x80000000u : 0u;
        }
        if (!plan.lanetab[3] && !jit_dry_run()) {
            const size_t nwg = o.pmap.size() / 2;
            std::vector<uint32_t> rows(nwg * 4 * 8, 0u);
            auto region = [](int slot, int b) { return (uint32_t)(((2 * slot + b) << 8) | (b ? 17 : 0)); };
            for (size_t w = 0; w < nwg; ++w) {
                if (o.ptile[w * 8] == 0xffffffffu) {
                    rows[w * 32] = 0xffffffffu;
                    continue;
                }
                for (int j = 0; j < NS; ++j) {
                    uint32_t* en = &rows[(w * 4 + (size_t)j) * 8];
                    for (int b = 0; b < 2; ++b) {
                        i64 id = o.ptile[w * 8 + (size_t)b * 4 + (size_t)j], org = 0;
                        for (int d = 0; d < c.N; ++d) {
                            org += (id % o.ntiles[d]) * (c.strides[o.k0][d] << o.lg[d]);
                            id /= o.ntiles[d];
                        }
                        en[b] = (uint32_t)org;
                        en[2] |= region(j, b) << (16 * b);
                        const uint64_t mp = o.pmap[w * 2 + (size_t)b];
                        // view k of the kernel = input k + 1; input 1 is the identity view (the lane's own registers)
                        for (int k = 1; k < c.M - 1 && k < 4; ++k) en[2 + k] |= region((int)((mp >> ((j * 8 + k) * 2)) & 3u), b) << (16 * b);
                    }
                }
            }
            void* dptr = nullptr;
            hipError_t e = hipMalloc(&dptr, rows.size() * sizeof(uint32_t));
            if (e != hipSuccess) return hip_error(e, "hipMalloc(orbit pair table)");
            e = hipMemcpy(dptr, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(dptr);
                return hip_error(e, "hipMemcpy(orbit pair table)");
            }
            plan.lanetab[3] = dptr;
        }
        a.list = reinterpret_cast<const uint32_t*>(plan.lanetab[3]);
        if (!jit_dry_run()) {
            cached.resize(sizeof a);
            std::memcpy(cached.data(), &a, sizeof a);
        }
    }
    a.src = (const char*)tab.base[o.k0];
    a.dst = (char*)tab.base[0];
    const Options& opt = options();
    a.nts = opt.nt_store > 0 ? 1 : 0;  // (32- / 64-byte runs: partial lines meet in L2, plain stores -- as in the one-orbit form)
    if (has_wt_store<OVec<T, V>>::value && (opt.nt_store == 2 || want_self_release(plan))) a.nts = 2;
    a.nlist = (int32_t)(o.pmap.size() / 2);
    const unsigned grid = (unsigned)(o.pmap.size() / 2), block = 256;
    const size_t lds = 8 * (sizeof(T) << 8);
    if (jit_no_launch()) return SMR_OK;
    clear_sticky_error();
    mark_sliceable(2, 0u, (unsigned)(4 * 8 * sizeof(uint32_t)));  // one table row (four entries) per workgroup; the pointer is parameter 0
    if (a.nts == 2) mark_self_released();
    const OrbitHead h = orbit_head(a);
    // one kernel per store policy: no branch on a kernel argument between the exchange and the stores
    if (a.nts == 2)
        SMR_LAUNCH((k_orbit_pair<T, F, 2>), dim3(grid), dim3(block), lds, s, h.list, h.src, h.dst, h.eshp, h.elenp, h.estride[0], h.estride[1], h.estride[2], h.estride[3],
                   h.ntlog, a, f SMR_STAMP_ARG(grid, block));
    else if (a.nts)
        SMR_LAUNCH((k_orbit_pair<T, F, 1>), dim3(grid), dim3(block), lds, s, h.list, h.src, h.dst, h.eshp, h.elenp, h.estride[0], h.estride[1], h.estride[2], h.estride[3],
                   h.ntlog, a, f SMR_STAMP_ARG(grid, block));
    else
        SMR_LAUNCH((k_orbit_pair<T, F, 0>), dim3(grid), dim3(block), lds, s, h.list, h.src, h.dst, h.eshp, h.elenp, h.estride[0], h.estride[1], h.estride[2], h.estride[3],
                   h.ntlog, a, f SMR_STAMP_ARG(grid, block));
    return check_launch("k_orbit_pair");
}

// persistent pipelined form: when the LDS footprint leaves one workgroup per CU and every CU gets several orbits
template <class T, class F, int V, int NREP, int NG, bool OWN0>
static int go3(const Plan& plan, hipStream_t s, F f, const OpTab& tab) {
    const OrbitPlan& o = plan.orbit;
    const size_t lds = (size_t)NG * (sizeof(T) << o.tilelog);
    const i64 want = options().orbit_pipe;
    bool pipe = false;
    if constexpr (V * sizeof(T) == 16) {
        // at least 8 orbits per CU: with 4 (64^4) the pipelined ComplexF32 kernel -- 128 VGPRs, 20 bytes of scratch -- loses
        // (68.4 vs 47.6 us) and the Float64 one ties; from 80^4 on it gains 2-4 % (tools/orbit_cplx.py)
        pipe = want > 0 ? o.list.size() > 256 : (want < 0 && lds > 80 * 1024 && o.list.size() >= 8 * 256);
        if (pipe) return go4<T, F, V, NREP, NG, OWN0, true>(plan, s, f, tab);
    }
    return go4<T, F, V, NREP, NG, OWN0, false>(plan, s, f, tab);
}

template <class T, class F, int V, int NREP>
static int go2(const Plan& plan, hipStream_t s, F f, const OpTab& tab) {
    const OrbitPlan& o = plan.orbit;
    bool own0 = true;  // is input 1 the identity view?
    for (int d = 0; d < plan.c.N; ++d)
        if (o.pdim[1][d] != d) own0 = false;
    if constexpr (!is_jit<F>::value && V == 2 && NREP == 1 && sizeof(T) == 8) {
        if constexpr (F::NIN >= 2 && F::NIN <= 4) {
            // Write-through launches (store policy 2: recorded sequences, eager calls on library-owned streams) gain most -- every store
            // travels to the memory side on its own and 64-byte pieces pay: replay of the 4-way sum at 32^4 4.72 -> 4.31 us on one queue,
            // 3.66 -> 3.04 cut in two, bench step 5.65 -> 5.2 us -- plain-store launches through HIP 4.40 -> 4.27 us.  (With the first
            // work list -- an orbit's smallest tile paired with whatever sat next to it -- the form LOST through HIP, 4.72 us: the list
            // matters as much as the kernel, smr_plan.cpp.)  orbit_pair = 0 switches the form off (tests, tools/cold_orbit_sweep.py).
            if (o.pair_ok && own0 && o.ng == 4 && plan.c.M - 1 == F::NIN && options().orbit_pair >= 1)
                return go_pair<T, F>(plan, s, f, tab);
        }
    }
    if (o.ng == 2) return own0 ? go3<T, F, V, NREP, 2, true>(plan, s, f, tab) : go3<T, F, V, NREP, 2, false>(plan, s, f, tab);
    return own0 ? go3<T, F, V, NREP, 4, true>(plan, s, f, tab) : go3<T, F, V, NREP, 4, false>(plan, s, f, tab);
}

template <class T, class F>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const OrbitPlan& o = plan.orbit;
    const OpTab tab = make_optab(c, bases);
    // every view must still be a view of ONE buffer after rebinding
    for (int k = 1; k < c.M; ++k)
        if (tab.base[k] != tab.base[o.k0]) return set_error(SMR_EINVAL, "orbit plan: the inputs must stay views of one buffer");
    constexpr int VMAX = (16 / sizeof(T)) > 1 ? (int)(16 / sizeof(T)) : 1;
    bool vec = o.vec == VMAX && VMAX > 1;
    if (vec && ((((uintptr_t)tab.base[0]) | ((uintptr_t)tab.base[o.k0])) % 16)) vec = false;
    const int tile = 1 << o.tilelog;
    if constexpr (VMAX > 1) {
        if (vec) {
            const int nrep = std::max(1, tile / (VMAX * 1024));
            if (nrep == 1) return go2<T, F, VMAX, 1>(plan, s, f, tab);
            if (nrep == 2) return go2<T, F, VMAX, 2>(plan, s, f, tab);
            return set_error(SMR_EINVAL, "orbit: unexpected tile size");
        }
        // element-wise accesses (operands not 16-byte aligned): only instantiated for tiles of up to 1024 elements
        if (tile <= 1024) return go2<T, F, 1, 1>(plan, s, f, tab);
        return set_error(SMR_EUNSUPPORTED, "orbit plan: operands must be 16-byte aligned (rebind aligned pointers or create a new plan)");
    } else {
        const int nrep = std::max(1, tile / 1024);
        if (nrep == 1) return go2<T, F, 1, 1>(plan, s, f, tab);
        if (nrep == 2) return go2<T, F, 1, 2>(plan, s, f, tab);
        if (nrep == 4) return go2<T, F, 1, 4>(plan, s, f, tab);
        return set_error(SMR_EINVAL, "orbit: unexpected tile size");
    }
}

template <>
int launch_orbit_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    switch (c.fkind) {  // natively compiled functors of this family; everything else is compiled at run time
        case FK_ADD2: return go<T, FAdd2<T>>(plan, bases, s, FAdd2<T>{});
        case FK_ADD4: return go<T, FAdd4<T>>(plan, bases, s, FAdd4<T>{});
        case FK_SYM: return go<T, FSym<T>>(plan, bases, s, FSym<T>{hostmk<T>(c.fc[0], c.fc[1])});
        default: break;
    }
    return with_prog<T>(c, [&](auto f) { return go<T, decltype(f)>(plan, bases, s, f); });
}
#endif  // !SMR_JIT

}  // namespace smr
