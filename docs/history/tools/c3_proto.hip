// c3_proto.hip -- design experiment for the 4-way permuted sum (BASELINE configs[2]):
//     B[i] = ((A[i] + A[r i]) + A[r^2 i]) + A[r^3 i],   r(i0,i1,i2,i3) = (i3,i0,i1,i2),  n^4 Float64.
// The library's TILED kernel reads A four times through L1 (once per permuted view).  Here a workgroup
// owns the ORBIT {T, rT, r^2 T, r^3 T} of an index box T under the permutation group the views
// generate: it loads A on the four boxes ONCE (natural order, 16 B per lane), exchanges through LDS and
// writes B on the same four boxes -- every element of A crosses L1 once instead of four times.
// Root boxes: the boxes of the coarse cells (edge m = longest box edge) whose coarse coordinate is
// the lexicographically smallest of its rotations; cells with a symmetric coordinate are covered by more
// than one orbit (identical values written twice: 9 % extra work at 32^4, 0.4 % at 128^4).
// Also timed: plain copy, 4-input contiguous add (distinct arrays / the same array), and the box-shaped
// copy with the orbit kernel's access pattern but no exchange (cost of 32-/64-B runs alone).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/c3_proto.hip -o tools/bin/c3_proto
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct alignas(16) d2 {
    double v[2];
};

template <int SWZ>
__device__ __forceinline__ uint32_t swz(uint32_t i) {
    if constexpr (SWZ == 1) return i ^ ((i >> 5) & 31u);
    if constexpr (SWZ == 2) return i ^ (((i >> 5) ^ (i >> 10)) & 31u);
    if constexpr (SWZ == 3) return i ^ (((i >> 4) ^ (i >> 8)) & 31u);
    if constexpr (SWZ == 4) return i ^ (((i >> 4) ^ (i >> 8) ^ (i >> 12)) & 31u);
    return i;
}

template <bool NTS>
__device__ __forceinline__ void store2(double* p, d2 v) {
    if constexpr (NTS) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        dv2 t;
        t.x = v.v[0];
        t.y = v.v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<dv2*>(p));
    } else {
        *reinterpret_cast<d2*>(p) = v;
    }
}

// ---- orbit kernel ----------------------------------------------------------------------------------
// MODE 0: orbit sum; MODE 1: box copy (same global access pattern, no LDS); MODE 2: box read, linear
// write; MODE 3: linear read, box write (2/3: timing only).  SPLIT > 1: the orbit's outputs are divided
// over SPLIT workgroups, each of which still loads all four boxes.
template <bool NTL>
__device__ __forceinline__ d2 load2(const double* p) {
    if constexpr (NTL) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        dv2 t = __builtin_nontemporal_load(reinterpret_cast<const dv2*>(p));
        d2 r;
        r.v[0] = t.x;
        r.v[1] = t.y;
        return r;
    } else {
        return *reinterpret_cast<const d2*>(p);
    }
}

template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS, int MODE, int SPLIT = 1, bool NTL = false, int Q = 1>
__global__ void __launch_bounds__(Q << NTLOG) k_orbit(const double* __restrict__ A, double* __restrict__ B, int nlog, const uint32_t* __restrict__ boxes) {
    constexpr int LG[4] = {L0, L1, L2, L3};
    constexpr int TL = L0 + L1 + L2 + L3;
    constexpr int NT = 1 << NTLOG;
    constexpr int NREP = (1 << TL) / (2 * NT);
    static_assert(NREP >= 1, "box too small for the workgroup");
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const uint32_t tid = threadIdx.x & (NT - 1);
    const uint32_t sub = threadIdx.x >> NTLOG;
    double* lds = lds_all + ((size_t)sub << (TL + 2));
    uint32_t packed = boxes[(blockIdx.x / SPLIT) * Q + sub];
    const uint32_t part = blockIdx.x % SPLIT;
    const bool idle = packed == 0xffffffffu;
    if (Q == 1 && idle) return;
    if (idle) packed = 0;
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = ((packed >> (8 * d)) & 0xffu) << 2;  // origins in units of 4 elements

    d2 x[4][NREP];
    uint32_t goff[4][NREP];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // box k = r^k T: shape log sl[d] = LG[(d - k) & 3], origin o[(d - k) & 3]
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const uint32_t e = (((uint32_t)r << NTLOG) | tid) << 1;
            uint32_t off = 0;
            int sh = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int sl = LG[(d - k) & 3];
                const uint32_t j = (e >> sh) & ((1u << sl) - 1u);
                off += (o[(d - k) & 3] + j) << (nlog * d);
                sh += sl;
            }
            goff[k][r] = off;
            if constexpr (MODE == 3) off = ((blockIdx.x * 4u + k) << TL) + e;
            x[k][r] = load2<NTL>(A + off);
        }
    }
    if constexpr (MODE != 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                uint32_t off = goff[k][r];
                if constexpr (MODE == 2) off = ((blockIdx.x * 4u + k) << TL) + ((((uint32_t)r << NTLOG) | tid) << 1);
                store2<NTS>(B + off, x[k][r]);
            }
        return;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                const uint32_t e = (((uint32_t)r << NTLOG) | tid) << 1;
                lds[(k << TL) + swz<SWZ>(e)] = x[k][r].v[0];
                lds[(k << TL) + swz<SWZ>(e | 1u)] = x[k][r].v[1];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                if (SPLIT > 1 && (uint32_t)(k % SPLIT) != part) continue;
                if (Q > 1 && idle) continue;
                d2 out;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t e = ((((uint32_t)r << NTLOG) | tid) << 1) | (uint32_t)h;
                    uint32_t j[4];
                    int sh = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int sl = LG[(d - k) & 3];
                        j[d] = (e >> sh) & ((1u << sl) - 1u);
                        sh += sl;
                    }
                    double acc = x[k][r].v[h];
#pragma unroll
                    for (int m = 1; m < 4; ++m) {
                        // r^m i sits in box (k + m) & 3 at local coordinate jj[d] = j[(d - m) & 3]
                        uint32_t idx = 0;
                        int s2 = 0;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            idx |= j[(d - m) & 3] << s2;
                            s2 += LG[(d - m - k) & 3];
                        }
                        acc = acc + lds[(((k + m) & 3) << TL) + swz<SWZ>(idx)];
                    }
                    out.v[h] = acc;
                }
                store2<NTS>(B + goff[k][r], out);
            }
    }
}

// ---- streaming baselines -----------------------------------------------------------------------------
template <bool NTS>
__global__ void __launch_bounds__(256) k_copy(const double* __restrict__ A, double* __restrict__ B) {
    const size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    d2 a = *reinterpret_cast<const d2*>(A + i);
    d2 b = *reinterpret_cast<const d2*>(A + i + 512);
    store2<NTS>(B + i, a);
    store2<NTS>(B + i + 512, b);
}
__global__ void __launch_bounds__(256) k_add4(const double* __restrict__ A1, const double* __restrict__ A2, const double* __restrict__ A3,
                                              const double* __restrict__ A4, double* __restrict__ B) {
    const size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    d2 a[2], b[2], c[2], d[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        a[u] = *reinterpret_cast<const d2*>(A1 + i + 512 * u);
        b[u] = *reinterpret_cast<const d2*>(A2 + i + 512 * u);
        c[u] = *reinterpret_cast<const d2*>(A3 + i + 512 * u);
        d[u] = *reinterpret_cast<const d2*>(A4 + i + 512 * u);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        d2 o;
        o.v[0] = ((a[u].v[0] + b[u].v[0]) + c[u].v[0]) + d[u].v[0];
        o.v[1] = ((a[u].v[1] + b[u].v[1]) + c[u].v[1]) + d[u].v[1];
        *reinterpret_cast<d2*>(B + i + 512 * u) = o;
    }
}

// ---- host ----------------------------------------------------------------------------------------------
template <class L>
static float time_graph(hipStream_t st, int reps, L launch) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return best * 1000.f / reps;
}

static std::vector<uint32_t> root_boxes(int n, const int* lg) {
    const int ml = *std::max_element(lg, lg + 4);
    const int nc = n >> ml;  // coarse cells per dim
    std::vector<uint32_t> out;
    for (int h3 = 0; h3 < nc; ++h3)
        for (int h2 = 0; h2 < nc; ++h2)
            for (int h1 = 0; h1 < nc; ++h1)
                for (int h0 = 0; h0 < nc; ++h0) {
                    std::array<int, 4> h{h0, h1, h2, h3};
                    bool root = true;
                    std::array<int, 4> g = h;
                    for (int k = 1; k < 4; ++k) {
                        g = {g[3], g[0], g[1], g[2]};
                        if (g < h) root = false;
                    }
                    if (!root) continue;
                    int nb[4];
                    for (int d = 0; d < 4; ++d) nb[d] = 1 << (ml - lg[d]);
                    for (int u3 = 0; u3 < nb[3]; ++u3)
                        for (int u2 = 0; u2 < nb[2]; ++u2)
                            for (int u1 = 0; u1 < nb[1]; ++u1)
                                for (int u0 = 0; u0 < nb[0]; ++u0) {
                                    const int u[4] = {u0, u1, u2, u3};
                                    uint32_t p = 0;
                                    for (int d = 0; d < 4; ++d) p |= (uint32_t)(((h[d] << ml) + (u[d] << lg[d])) >> 2) << (8 * d);
                                    out.push_back(p);
                                }
                }
    // interleave so that consecutive workgroups (round-robin over the 8 XCDs) stay apart: not needed for
    // correctness; natural order kept
    return out;
}


// Orbit list grouped by 16^4 super-cells and laid out so that the 16 cube orbits of a super-cell orbit run
// on ONE XCD at about the same time (workgroup b runs on XCD b % 8): the half-line partners of every 64-B
// run then meet in that XCD's L2.  Cubic boxes only.  Entries 0xffffffff = idle workgroup.
static std::vector<uint32_t> grouped_boxes(int n, int ml, int group_log) {
    const int nc = n >> ml;
    auto canon = [&](std::array<int, 4> h) {
        std::array<int, 4> best = h, g = h;
        for (int k = 1; k < 4; ++k) {
            g = {g[3], g[0], g[1], g[2]};
            if (g < best) best = g;
        }
        return best;
    };
    std::vector<char> seen((size_t)nc * nc * nc * nc, 0);
    auto id = [&](const std::array<int, 4>& h) { return (size_t)h[0] + (size_t)nc * (h[1] + (size_t)nc * (h[2] + (size_t)nc * h[3])); };
    std::vector<uint32_t> list;
    const int gs = 1 << group_log, ng = (nc + gs - 1) / gs;
    for (int H3 = 0; H3 < ng; ++H3)
        for (int H2 = 0; H2 < ng; ++H2)
            for (int H1 = 0; H1 < ng; ++H1)
                for (int H0 = 0; H0 < ng; ++H0)
                    for (int u = 0; u < gs * gs * gs * gs; ++u) {
                        std::array<int, 4> h{H0 * gs + (u % gs), H1 * gs + (u / gs % gs), H2 * gs + (u / gs / gs % gs), H3 * gs + (u / gs / gs / gs)};
                        if (h[0] >= nc || h[1] >= nc || h[2] >= nc || h[3] >= nc) continue;
                        const std::array<int, 4> c = canon(h);
                        if (seen[id(c)]) continue;
                        seen[id(c)] = 1;
                        uint32_t p = 0;
                        for (int d = 0; d < 4; ++d) p |= (uint32_t)((c[d] << ml) >> 2) << (8 * d);
                        list.push_back(p);
                    }
    const size_t cs = (list.size() + 7) / 8;
    std::vector<uint32_t> out(cs * 8, 0xffffffffu);
    for (size_t x = 0; x < 8; ++x)
        for (size_t slot = 0; slot < cs; ++slot)
            if (x * cs + slot < list.size()) out[slot * 8 + x] = list[x * cs + slot];
    return out;
}

struct Ctx {
    int n, nlog;
    size_t N;
    double *dA, *dB, *dA2, *dA3, *dA4;
    std::vector<double> hA, want, got;
    hipStream_t st;
    int reps;
};

static bool check(Ctx& c, const char* name) {
    CK(hipMemcpy(c.got.data(), c.dB, c.N * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < c.N; ++i)
        if (std::memcmp(&c.got[i], &c.want[i], 8) != 0) ++bad;
    if (bad) printf("   !! %s: %zu mismatching elements\n", name, bad);
    return bad == 0;
}

template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS, int MODE, int SPLIT = 1, bool NTL = false, int GROUP = -1, int Q = 1>
static void run_orbit(Ctx& c) {
    const int lg[4] = {L0, L1, L2, L3};
    const int ml = *std::max_element(lg, lg + 4);
    if ((1 << ml) > c.n) return;
    std::vector<uint32_t> boxes = GROUP >= 0 ? grouped_boxes(c.n, ml, GROUP) : root_boxes(c.n, lg);
    while (boxes.size() % (8 * Q)) boxes.push_back(0xffffffffu);
    uint32_t* db;
    CK(hipMalloc(&db, boxes.size() * 4));
    CK(hipMemcpy(db, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    constexpr int TL = L0 + L1 + L2 + L3;
    const size_t lds = MODE == 0 ? ((size_t)4 * 8 << TL) * Q : 0;
    auto kern = k_orbit<L0, L1, L2, L3, NTLOG, SWZ, NTS, MODE, SPLIT, NTL, Q>;
    while (boxes.size() % (8 * Q)) boxes.push_back(0xffffffffu);
    unsigned grid = (unsigned)boxes.size() * SPLIT / Q;
    if (MODE >= 2) grid = std::min<unsigned>(grid, (unsigned)(c.N / (4u << TL)));
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(c.dB, 0xff, c.N * 8, c.st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Q << NTLOG), lds, c.st, c.dA, c.dB, c.nlog, db);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c.st));
    bool ok = true;
    if (MODE == 0) ok = check(c, "orbit");
    else if (MODE == 1) {
        CK(hipMemcpy(c.got.data(), c.dB, c.N * 8, hipMemcpyDeviceToHost));
        ok = std::memcmp(c.got.data(), c.hA.data(), c.N * 8) == 0;
        if (!ok) printf("   !! boxcopy mismatch\n");
    }
    const float us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(Q << NTLOG), lds, c.st, c.dA, c.dB, c.nlog, db); });
    static const char* mn[] = {"orbit", "boxcopy", "box->lin", "lin->box"};
    printf("n=%3d %-8s/%d grp %2d box %2dx%2dx%2dx%2d lanes %4d swz %d nts %d ntl %d wgs %6u lds %6zu : %9.2f us  %7.1f GB/s  (%.1f %% of 8 TB/s) %s\n", c.n,
           mn[MODE], SPLIT, GROUP, 1 << L0, 1 << L1, 1 << L2, 1 << L3, Q << NTLOG, SWZ, (int)NTS, (int)NTL, grid, lds, us,
           2.0 * c.N * 8 / us * 1e-3, 2.0 * c.N * 8 / us * 1e-3 / 80.0, ok ? "ok" : "WRONG");
    CK(hipFree(db));
}


// ---- library-like tiled kernel, fully specialised -------------------------------------------------------
// One destination tile per workgroup; A1 read in destination order, A2..A4 in their own memory order and
// exchanged through LDS (destination order, swizzled) -- the structure of smr_k_tiled.hip with every index
// a compile-time expression (no lane tables, no kernel-argument decode): a lower bound for that design.
template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS>
__global__ void __launch_bounds__(1 << NTLOG) k_tiled4(const double* __restrict__ A, double* __restrict__ B, int nlog) {
    constexpr int LG[4] = {L0, L1, L2, L3};
    constexpr int TL = L0 + L1 + L2 + L3;
    constexpr int NT = 1 << NTLOG;
    constexpr int NREP = (1 << TL) / (2 * NT);
    constexpr int LSH[4] = {0, L0, L0 + L1, L0 + L1 + L2};
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t tid = threadIdx.x;
    // tile coordinates: blockIdx decomposed over (n >> L_d) tiles per dim, dim 0 fastest
    uint32_t b = blockIdx.x, o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int tl = nlog - LG[d];
        o[d] = (b & ((1u << tl) - 1u)) << LG[d];
        b >>= tl;
    }
    d2 x[4][NREP];
    uint32_t doff[NREP];
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        const uint32_t e = (((uint32_t)r << NTLOG) | tid) << 1;
        uint32_t off = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) off += (o[d] + ((e >> LSH[d]) & ((1u << LG[d]) - 1u))) << (nlog * d);
        doff[r] = off;
        x[0][r] = *reinterpret_cast<const d2*>(A + off);
    }
    uint32_t lidx[3][NREP][2];
#pragma unroll
    for (int m = 1; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const uint32_t e = (((uint32_t)r << NTLOG) | tid) << 1;
            uint32_t off = 0, li = 0;
            int sh = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {  // memory coordinate d of A_m is destination coordinate q = (d - m) & 3
                const int q = (d - m) & 3;
                const uint32_t a = (e >> sh) & ((1u << LG[q]) - 1u);
                off += (o[q] + a) << (nlog * d);
                li |= a << LSH[q];
                sh += LG[q];
            }
            x[m][r] = *reinterpret_cast<const d2*>(A + off);
            lidx[m - 1][r][0] = swz<SWZ>(li);
            lidx[m - 1][r][1] = swz<SWZ>(li | (1u << LSH[(0 - m) & 3]));
        }
#pragma unroll
    for (int m = 1; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            lds[((m - 1) << TL) + lidx[m - 1][r][0]] = x[m][r].v[0];
            lds[((m - 1) << TL) + lidx[m - 1][r][1]] = x[m][r].v[1];
        }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        d2 out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t e = ((((uint32_t)r << NTLOG) | tid) << 1) | (uint32_t)h;
            double acc = x[0][r].v[h];
#pragma unroll
            for (int m = 1; m < 4; ++m) acc = acc + lds[((m - 1) << TL) + swz<SWZ>(e)];
            out.v[h] = acc;
        }
        store2<NTS>(B + doff[r], out);
    }
}

template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS>
static void run_tiled(Ctx& c) {
    constexpr int TL = L0 + L1 + L2 + L3;
    const int lg[4] = {L0, L1, L2, L3};
    for (int d = 0; d < 4; ++d)
        if ((1 << lg[d]) > c.n) return;
    const size_t lds = (size_t)3 * 8 << TL;
    const unsigned grid = (unsigned)(c.N >> TL);
    auto kern = k_tiled4<L0, L1, L2, L3, NTLOG, SWZ, NTS>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(c.dB, 0xff, c.N * 8, c.st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << NTLOG), lds, c.st, c.dA, c.dB, c.nlog);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c.st));
    const bool ok = check(c, "tiled4");
    const float us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << NTLOG), lds, c.st, c.dA, c.dB, c.nlog); });
    printf("n=%3d tiled4     tile %2dx%2dx%2dx%2d lanes %4d swz %d nts %d  wgs %6u lds %6zu : %9.2f us  %7.1f GB/s  (%.1f %% of 8 TB/s) %s\n", c.n, 1 << L0,
           1 << L1, 1 << L2, 1 << L3, 1 << NTLOG, SWZ, (int)NTS, grid, lds, us, 2.0 * c.N * 8 / us * 1e-3, 2.0 * c.N * 8 / us * 1e-3 / 80.0, ok ? "ok" : "WRONG");
}

// ---- the library's ORBIT kernel (smr_k_orbit.hip) restated stand-alone: every shift / stride / swizzle
// parameter is a kernel argument.  VAR selects experiments on its prologue.
struct RtArgs {
    const char* src;
    char* dst;
    const uint32_t* list;  // 4 element origins per workgroup
    int32_t nin, tilelog, ntlog, conj0, nts, pad0;
    uint32_t swz_s1, swz_s2, swz_mask, pad1;
    int32_t esh[4], elen[4];
    uint32_t estride[4];
    int32_t lsh[7][4];
    int32_t slot[4][7];
    uint32_t conjbit[7];
};

template <int VAR>
__global__ void __launch_bounds__(1024) k_orbit_rt(const RtArgs a, long long* __restrict__ stamps) {
    constexpr int NG = 4, NK = 4, V = 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t tid = threadIdx.x;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (VAR & 8) t0 = __builtin_readcyclecounter();
    typedef uint32_t rowv __attribute__((ext_vector_type(4)));
    const rowv row = reinterpret_cast<const rowv*>(a.list)[blockIdx.x];
    const bool live = row[0] != 0xffffffffu;
    long long org[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) org[g] = live ? (long long)row[g] * 8 : 0;
    const uint32_t e = tid * V;
    uint32_t cj[4], goff = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cj[j] = __builtin_amdgcn_ubfe(e, (uint32_t)a.esh[j], (uint32_t)a.elen[j]);
        goff += cj[j] * a.estride[j];
    }
    if (VAR & 8) t1 = __builtin_readcyclecounter();
    d2 x[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) x[g] = *reinterpret_cast<const d2*>(a.src + org[g] + goff);
    uint32_t lr[NK];
#pragma unroll
    for (int k = 1; k < NK; ++k) {
        uint32_t l = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) l |= cj[j] << a.lsh[k][j];
        lr[k] = l;
    }
    lr[0] = 0;
    uint32_t s1 = a.swz_s1, s2 = a.swz_s2, sm = a.swz_mask;
    uint32_t sbase[NG][NK], hbit[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        hbit[k] = (uint32_t)a.lsh[k][0];
#pragma unroll
        for (int g = 0; g < NG; ++g) sbase[g][k] = (uint32_t)a.slot[g][k] << a.tilelog;
    }
    if (VAR & 1) {
        asm volatile("" : "+s"(s1), "+s"(s2), "+s"(sm));
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            asm volatile("" : "+s"(hbit[k]), "+v"(lr[k]));
#pragma unroll
            for (int g = 0; g < NG; ++g) asm volatile("" : "+s"(sbase[g][k]));
        }
    }
    auto swzf = [&](uint32_t i) { return i ^ (((i >> s1) ^ (i >> s2)) & sm); };
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double* L = lds + ((size_t)g << a.tilelog);
        L[swzf(e)] = x[g].v[0];
        L[swzf(e + 1)] = x[g].v[1];
    }
    if (VAR & 8) t2 = __builtin_readcyclecounter();
    __syncthreads();
    if (VAR & 8) t3 = __builtin_readcyclecounter();
    double val[NG][V][NK];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int h = 0; h < V; ++h)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (k == 0) val[g][h][k] = x[g].v[h];
                else val[g][h][k] = lds[sbase[g][k] + swzf(lr[k] | ((uint32_t)h << hbit[k]))];
            }
    if (VAR & 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int h = 0; h < V; ++h) x[g].v[h] = ((val[g][h][0] + val[g][h][1]) + val[g][h][2]) + val[g][h][3];
    if (VAR & 8) t4 = __builtin_readcyclecounter();
    if (!live) return;
#pragma unroll
    for (int g = 0; g < NG; ++g) *reinterpret_cast<d2*>(a.dst + org[g] + goff) = x[g];
    if ((VAR & 8) && tid == 0) {
        long long* o = stamps + (size_t)blockIdx.x * 8;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
        o[3] = t3;
        o[4] = t4;
        o[5] = __builtin_readcyclecounter();
    }
}

template <int VAR>
static void run_orbit_rt(Ctx& c, int lgc, int ntlog) {
    const int ml = lgc;
    if ((1 << ml) > c.n) return;
    std::vector<uint32_t> boxes = grouped_boxes(c.n, ml, 1);
    std::vector<uint32_t> rows(boxes.size() * 4, 0xffffffffu);
    for (size_t w = 0; w < boxes.size(); ++w) {
        if (boxes[w] == 0xffffffffu) continue;
        uint32_t o[4];
        for (int d = 0; d < 4; ++d) o[d] = ((boxes[w] >> (8 * d)) & 0xffu) << 2;
        for (int k = 0; k < 4; ++k) {
            uint32_t off = 0;
            for (int d = 0; d < 4; ++d) off += o[(d - k) & 3] << (c.nlog * d);
            rows[w * 4 + k] = off;
        }
    }
    uint32_t* db;
    CK(hipMalloc(&db, rows.size() * 4));
    CK(hipMemcpy(db, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    long long* stamps;
    CK(hipMalloc(&stamps, boxes.size() * 64));
    CK(hipMemset(stamps, 0, boxes.size() * 64));
    RtArgs a;
    memset(&a, 0, sizeof a);
    a.src = (const char*)c.dA;
    a.dst = (char*)c.dB;
    a.list = db;
    a.nin = 4;
    a.tilelog = 4 * lgc;
    a.ntlog = ntlog;
    for (int j = 0; j < 4; ++j) {
        a.esh[j] = j * lgc;
        a.elen[j] = lgc;
        a.estride[j] = 8u << (c.nlog * j);
    }
    // view k reads the buffer at r^k(i): local coordinate j of the output sits at tiled dim (j + k) & 3
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) a.lsh[k][j] = ((j + k) & 3) * lgc;
    for (int g = 0; g < 4; ++g)
        for (int k = 0; k < 4; ++k) a.slot[g][k] = (g + k) & 3;
    a.swz_s1 = 4;
    a.swz_s2 = 8;
    a.swz_mask = 31;
    const size_t lds = (size_t)4 * 8 << a.tilelog;
    const unsigned grid = (unsigned)boxes.size(), block = 1u << ntlog;
    auto kern = k_orbit_rt<VAR>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(c.dB, 0xff, c.N * 8, c.st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, c.st, a, stamps);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c.st));
    const bool ok = check(c, "orbit_rt");
    const float us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, c.st, a, stamps); });
    printf("n=%3d orbit_rt var %2d cube %d lanes %4u wgs %6u lds %6zu : %9.2f us  %7.1f GB/s  (%.1f %% of 8 TB/s) %s\n", c.n, VAR, 1 << lgc, block, grid, lds, us,
           2.0 * c.N * 8 / us * 1e-3, 2.0 * c.N * 8 / us * 1e-3 / 80.0, ok ? "ok" : "WRONG");
    if (VAR & 8) {
        std::vector<long long> h(boxes.size() * 8);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        long long tmin = -1;
        for (size_t w = 0; w < boxes.size(); ++w)
            if (h[w * 8] && (tmin < 0 || h[w * 8] < tmin)) tmin = h[w * 8];
        double acc[6] = {0, 0, 0, 0, 0, 0}, last = 0;
        size_t cnt = 0;
        for (size_t w = 0; w < boxes.size(); ++w) {
            if (!h[w * 8]) continue;
            ++cnt;
            for (int q = 0; q < 6; ++q) acc[q] += (double)(h[w * 8 + q] - (q ? h[w * 8 + q - 1] : tmin));
            last = std::max(last, (double)(h[w * 8 + 5] - tmin));
        }
        printf("      cycles (mean over %zu workgroups): start-after-first %.0f | row+offsets %.0f | loads->LDS %.0f | barrier %.0f | LDS reads+adds %.0f | stores issued %.0f | last end %.0f\n",
               cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, last);
    }
    CK(hipFree(db));
    CK(hipFree(stamps));
}

int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {32, 64, 128};
    for (int n : sizes) {
        Ctx c;
        c.n = n;
        c.nlog = 0;
        while ((1 << c.nlog) < n) ++c.nlog;
        c.N = (size_t)n * n * n * n;
        c.reps = n <= 32 ? 200 : (n <= 64 ? 40 : 4);
        CK(hipStreamCreate(&c.st));
        c.hA.resize(c.N);
        c.want.resize(c.N);
        c.got.resize(c.N);
        uint64_t s = 1234;
        for (size_t i = 0; i < c.N; ++i) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            c.hA[i] = (double)((int64_t)(s >> 11)) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
        }
        const size_t n1 = n, n2 = n1 * n, n3 = n2 * n;
        for (size_t i3 = 0; i3 < n1; ++i3)
            for (size_t i2 = 0; i2 < n1; ++i2)
                for (size_t i1 = 0; i1 < n1; ++i1)
                    for (size_t i0 = 0; i0 < n1; ++i0) {
                        const double a = c.hA[i0 + n1 * i1 + n2 * i2 + n3 * i3];
                        const double b = c.hA[i3 + n1 * i0 + n2 * i1 + n3 * i2];
                        const double cc = c.hA[i2 + n1 * i3 + n2 * i0 + n3 * i1];
                        const double d = c.hA[i1 + n1 * i2 + n2 * i3 + n3 * i0];
                        volatile double t = a + b;
                        t = t + cc;
                        t = t + d;
                        c.want[i0 + n1 * i1 + n2 * i2 + n3 * i3] = t;
                    }
        CK(hipMalloc(&c.dA, c.N * 8));
        CK(hipMalloc(&c.dB, c.N * 8));
        CK(hipMemcpy(c.dA, c.hA.data(), c.N * 8, hipMemcpyHostToDevice));
        const bool distinct = n <= 64;
        if (distinct) {
            CK(hipMalloc(&c.dA2, c.N * 8));
            CK(hipMalloc(&c.dA3, c.N * 8));
            CK(hipMalloc(&c.dA4, c.N * 8));
            CK(hipMemcpy(c.dA2, c.hA.data(), c.N * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(c.dA3, c.hA.data(), c.N * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(c.dA4, c.hA.data(), c.N * 8, hipMemcpyHostToDevice));
        }
        const unsigned g = (unsigned)(c.N / 1024);
        float us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(k_copy<false>, dim3(g), dim3(256), 0, c.st, c.dA, c.dB); });
        printf("n=%3d copy                 : %9.2f us  %7.1f GB/s\n", n, us, 2.0 * c.N * 8 / us * 1e-3);
        us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(k_copy<true>, dim3(g), dim3(256), 0, c.st, c.dA, c.dB); });
        printf("n=%3d copy (nt stores)     : %9.2f us  %7.1f GB/s\n", n, us, 2.0 * c.N * 8 / us * 1e-3);
        us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(k_add4, dim3(g), dim3(256), 0, c.st, c.dA, c.dA, c.dA, c.dA, c.dB); });
        printf("n=%3d add4, same array x4  : %9.2f us  %7.1f GB/s (algorithmic 2N)\n", n, us, 2.0 * c.N * 8 / us * 1e-3);
        if (distinct) {
            us = time_graph(c.st, c.reps, [&] { hipLaunchKernelGGL(k_add4, dim3(g), dim3(256), 0, c.st, c.dA, c.dA2, c.dA3, c.dA4, c.dB); });
            printf("n=%3d add4, 4 arrays       : %9.2f us  %7.1f GB/s (5N bytes: %.1f GB/s)\n", n, us, 2.0 * c.N * 8 / us * 1e-3, 5.0 * c.N * 8 / us * 1e-3);
        }
        if (c.n <= 32) {
            run_orbit<2, 2, 2, 2, 7, 3, false, 0, 1, false, 1>(c);
            run_orbit_rt<0>(c, 2, 7);
            run_orbit_rt<1>(c, 2, 7);
            run_orbit_rt<2>(c, 2, 7);
            run_orbit_rt<3>(c, 2, 7);
            run_orbit_rt<8>(c, 2, 7);
            run_orbit<2, 2, 2, 2, 7, 3, false, 0, 1, false, 1>(c);
        } else {
            run_orbit<3, 3, 3, 3, 10, 3, false, 0, 1, false, 1>(c);
            run_orbit_rt<0>(c, 3, 10);
            run_orbit_rt<3>(c, 3, 10);
        }
        CK(hipFree(c.dA));
        CK(hipFree(c.dB));
        if (distinct) {
            CK(hipFree(c.dA2));
            CK(hipFree(c.dA3));
            CK(hipFree(c.dA4));
        }
        CK(hipStreamDestroy(c.st));
    }
    return 0;
}
