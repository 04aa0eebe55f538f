// c3_proto2.hip -- round-3 design experiments for the 4-way permuted sum (BASELINE configs[2]) and for where the
// time of a 3-5 us launch goes.  Stand-alone (no library): every shape is a template parameter.
//     B[i] = ((A[i] + A[r i]) + A[r^2 i]) + A[r^3 i],   r(i0,i1,i2,i3) = (i3,i0,i1,i2),  n^4 Float64.
// New against tools/c3_proto.hip (round 2):
//   * rotated NON-CUBIC boxes with a grouped work list: a workgroup owns the boxes T, rT, r^2T, r^3T of a
//     sub-box T (L0 x L1 x L2 x L3) of a coarse cube; the boxes have rotated shapes, so three of the four can keep
//     64-byte runs while one is cut along its unit axis (8x8x8x4: 140 workgroups at 32^4; 8x8x4x4: 280; 8x4x4x4: 560).
//     All sub-boxes of one coarse orbit sit next to each other on ONE XCD (half-line partners meet in its L2);
//   * Q orbits per workgroup (fewer, fatter workgroups to dispatch);
//   * MODE 4 / 5: the read side / the write side of the orbit kernel alone;
//   * device-side wall-clock stamps (s_memrealtime, 100 MHz): per wave start / end for every launch of a replayed
//     graph -> first start, last end, dispatch ramp, cadence and inter-launch gap without any profiler or host clock;
//   * the two kernels of the bench step in ONE graph as two independent branches (fork / join) against in-order.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/c3_proto2.hip -o tools/bin/c3_proto2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef unsigned long long u64;

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

// per-wave stamps: stamps[(blockIdx.x * WAVES + wave) * 2 + {0: start, 1: end}] (wall clock ticks)
__device__ __forceinline__ void stamp_end(u64* stamps, u64 t0, int waves) {
    if (!stamps) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's stores have been acknowledged
    const u64 t1 = (u64)wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        u64* o = stamps + ((size_t)blockIdx.x * waves + (threadIdx.x >> 6)) * 2;
        o[0] = t0;
        o[1] = t1;
    }
}

// ---- orbit kernel: MODE 0 sum; 4: loads + exchange + adds, stores only of impossible values (read side);
// 5: stores only (write side, values from the lane id) -------------------------------------------------------
template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS, int MODE, int Q, bool ST>
__global__ void __launch_bounds__(Q << NTLOG) k_orb(const double* __restrict__ A, double* __restrict__ B, int n, const uint32_t* __restrict__ boxes,
                                                   u64* __restrict__ stamps) {
    constexpr int LG[4] = {L0, L1, L2, L3};
    constexpr int TL = L0 + L1 + L2 + L3;
    constexpr int NT = 1 << NTLOG;
    constexpr int NREP = (1 << TL) / (2 * NT);
    static_assert(NREP >= 1, "box too small for the workgroup");
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const u64 t0 = ST ? (u64)wall_clock64() : 0;
    const uint32_t st[4] = {1u, (uint32_t)n, (uint32_t)n * (uint32_t)n, (uint32_t)n * (uint32_t)n * (uint32_t)n};
    const uint32_t tid = threadIdx.x & (NT - 1);
    const uint32_t sub = threadIdx.x >> NTLOG;
    double* lds = lds_all + ((size_t)sub << (TL + 2));
    uint32_t packed = boxes[blockIdx.x * Q + sub];
    const bool idle = packed == 0xffffffffu;
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
                off += (o[(d - k) & 3] + j) * st[d];
                sh += sl;
            }
            goff[k][r] = off;
            if constexpr (MODE != 5) x[k][r] = *reinterpret_cast<const d2*>(A + off);
            else {
                x[k][r].v[0] = (double)e;
                x[k][r].v[1] = (double)(e + 1);
            }
        }
    }
    if constexpr (MODE == 5) {
        if (!idle) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREP; ++r) store2<NTS>(B + goff[k][r], x[k][r]);
        }
        if constexpr (ST) stamp_end(stamps, t0, (Q << NTLOG) >> 6);
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
                if constexpr (MODE == 4) {
                    // read side only: a store that never happens for finite data keeps the loads alive
                    if (out.v[0] == 1.2345e300 && !idle) store2<NTS>(B + goff[k][r], out);
                } else {
                    if (!idle) store2<NTS>(B + goff[k][r], out);
                }
            }
        if constexpr (ST) stamp_end(stamps, t0, (Q << NTLOG) >> 6);
    }
}

// ---- streaming baselines -----------------------------------------------------------------------------
// MODE 0 copy, 4 read only, 5 write only
template <bool NTS, int MODE, bool ST>
__global__ void __launch_bounds__(256) k_copy(const double* __restrict__ A, double* __restrict__ B, u64* __restrict__ stamps) {
    const u64 t0 = ST ? (u64)wall_clock64() : 0;
    const size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    d2 a, b;
    if constexpr (MODE != 5) {
        a = *reinterpret_cast<const d2*>(A + i);
        b = *reinterpret_cast<const d2*>(A + i + 512);
    } else {
        a.v[0] = a.v[1] = b.v[0] = b.v[1] = (double)threadIdx.x;
    }
    if constexpr (MODE == 4) {
        if (a.v[0] == 1.2345e300 || b.v[1] == 1.2345e300) store2<NTS>(B + i, a);
    } else {
        store2<NTS>(B + i, a);
        store2<NTS>(B + i + 512, b);
    }
    if constexpr (ST) stamp_end(stamps, t0, 4);
}
template <bool ST>
__global__ void __launch_bounds__(256) k_empty(u64* __restrict__ stamps) {
    const u64 t0 = ST ? (u64)wall_clock64() : 0;
    if constexpr (ST) stamp_end(stamps, t0, 4);
}

// B = permutedims(A, (4,3,2,1)): 32 x 32 tiles over (d0, d3), LDS transpose, 256 lanes x 4 elements (a plain
// restatement of the shape of the library's TILED kernel for the fork/join experiment)
template <bool NTS>
__global__ void __launch_bounds__(256) k_perm4321(const double* __restrict__ A, double* __restrict__ B, int nlog, u64* __restrict__ stamps) {
    __shared__ double t[32][33];
    const uint32_t n = 1u << nlog;
    // block -> (tile of d0, tile of d3, i1, i2) of the SOURCE
    uint32_t b = blockIdx.x;
    const uint32_t nt = n >> 5;
    const uint32_t t0s = b % nt; b /= nt;
    const uint32_t t3s = b % nt; b /= nt;
    const uint32_t i1 = b % n, i2 = b / n;
    const uint32_t lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i3 = t3s * 32 + ly + 8 * r, i0 = t0s * 32 + lx;
        t[ly + 8 * r][lx] = A[(size_t)i0 + ((size_t)i1 << nlog) + ((size_t)i2 << (2 * nlog)) + ((size_t)i3 << (3 * nlog))];
    }
    __syncthreads();
    // B[j0,j1,j2,j3] = A[j3,j2,j1,j0]: j0 = i3, j1 = i2, j2 = i1, j3 = i0
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t j0 = t3s * 32 + lx, j3 = t0s * 32 + ly + 8 * r;
        const double v = t[lx][ly + 8 * r];
        double* p = B + (size_t)j0 + ((size_t)i2 << nlog) + ((size_t)i1 << (2 * nlog)) + ((size_t)j3 << (3 * nlog));
        if constexpr (NTS) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
    (void)stamps;
}

// ---- host ----------------------------------------------------------------------------------------------
static double g_tick_ns = 10.0;

struct Ctx {
    int n, nlog;
    size_t N;
    double *dA, *dB, *dC;
    std::vector<double> hA, want, got;
    hipStream_t st, st2;
    int reps;
    u64* dstamps;
    size_t stamp_cap;  // u64 words
    bool verify = true;
};

// Launch `reps` times inside one graph (each launch gets its own stamp region when `stamped`), replay, time with
// HIP events; with stamps: analyse the LAST replay.
struct SpanStats {
    float us_events = 0;
    double span = 0, cadence = 0, gap = 0, ramp50 = 0, ramp95 = 0, rampmax = 0, life = 0;
};
template <class L>
static SpanStats time_graph(Ctx& c, int reps, unsigned waves_total, bool stamped, L launch) {
    SpanStats out;
    hipGraph_t g;
    hipGraphExec_t ge;
    if (stamped && (size_t)reps * waves_total * 2 > c.stamp_cap) reps = (int)(c.stamp_cap / ((size_t)waves_total * 2));
    CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch(stamped ? c.dstamps + (size_t)i * waves_total * 2 : (u64*)nullptr);
    CK(hipStreamEndCapture(c.st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, c.st));
    CK(hipStreamSynchronize(c.st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int t = 0; t < 5; ++t) {
        if (stamped) CK(hipMemsetAsync(c.dstamps, 0, (size_t)reps * waves_total * 16, c.st));
        CK(hipEventRecord(e0, c.st));
        CK(hipGraphLaunch(ge, c.st));
        CK(hipEventRecord(e1, c.st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    out.us_events = best * 1000.f / reps;
    if (stamped) {
        std::vector<u64> h((size_t)reps * waves_total * 2);
        CK(hipMemcpy(h.data(), c.dstamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> first(reps), last(reps), spans, cads, gaps, r50, r95, rmax, life;
        for (int i = 0; i < reps; ++i) {
            u64 f = ~0ull, l = 0;
            std::vector<u64> starts;
            double lf = 0;
            size_t cnt = 0;
            for (unsigned w = 0; w < waves_total; ++w) {
                const u64 s = h[((size_t)i * waves_total + w) * 2], e = h[((size_t)i * waves_total + w) * 2 + 1];
                if (!s) continue;
                f = std::min(f, s);
                l = std::max(l, e);
                starts.push_back(s);
                lf += (double)(e - s);
                ++cnt;
            }
            if (!cnt) continue;
            first[i] = (double)f;
            last[i] = (double)l;
            std::sort(starts.begin(), starts.end());
            spans.push_back((double)(l - f));
            r50.push_back((double)(starts[starts.size() / 2] - f));
            r95.push_back((double)(starts[starts.size() * 95 / 100] - f));
            rmax.push_back((double)(starts.back() - f));
            life.push_back(lf / cnt);
        }
        for (int i = 10; i + 1 < reps; ++i) {  // skip the head of the graph
            cads.push_back(first[i + 1] - first[i]);
            gaps.push_back(first[i + 1] - last[i]);
        }
        auto med = [](std::vector<double> v) {
            if (v.empty()) return 0.0;
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        const double k = g_tick_ns * 1e-3;  // ticks -> us
        out.span = med(spans) * k;
        out.cadence = med(cads) * k;
        out.gap = med(gaps) * k;
        out.ramp50 = med(r50) * k;
        out.ramp95 = med(r95) * k;
        out.rampmax = med(rmax) * k;
        out.life = med(life) * k;
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return out;
}

static void print_span(const char* what, const SpanStats& plain, const SpanStats& st) {
    printf("      %-28s events %6.2f us (unstamped build of the graph) | stamped graph: events %6.2f, device cadence %6.2f = span %5.2f + gap %5.2f us;"
           " wave starts after first: p50 %4.2f p95 %4.2f max %4.2f us; mean wave lifetime %5.2f us\n",
           what, plain.us_events, st.us_events, st.cadence, st.span, st.gap, st.ramp50, st.ramp95, st.rampmax, st.life);
}

// Work list: roots = coarse cubes (edge 2^ml) whose coordinate is the lexicographically smallest of its rotations,
// visited super-cell by super-cell (gs coarse cubes per dim); every root contributes all its sub-boxes one after
// the other; the list is cut into 8 contiguous runs, run x executed by the workgroups b = 8 * slot + x (XCD x).
static int g_order = 0;  // 0: super-cell grouped, one contiguous run per XCD; 1: the same list dealt round-robin to the XCDs; 2: shuffled, runs per XCD
static std::vector<uint32_t> grouped_boxes(int n, const int* lg, int group_log, int Q) {
    const int ml = *std::max_element(lg, lg + 4);
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
    int nb[4];
    for (int d = 0; d < 4; ++d) nb[d] = 1 << (ml - lg[d]);
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
                        for (int u3 = 0; u3 < nb[3]; ++u3)
                            for (int u2 = 0; u2 < nb[2]; ++u2)
                                for (int u1 = 0; u1 < nb[1]; ++u1)
                                    for (int u0 = 0; u0 < nb[0]; ++u0) {
                                        const int uu[4] = {u0, u1, u2, u3};
                                        uint32_t p = 0;
                                        for (int d = 0; d < 4; ++d) p |= (uint32_t)(((c[d] << ml) + (uu[d] << lg[d])) >> 2) << (8 * d);
                                        list.push_back(p);
                                    }
                    }
    if (g_order == 2) {
        uint64_t rs = 88172645463325252ull;
        for (size_t i = list.size(); i > 1; --i) {
            rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
            std::swap(list[i - 1], list[rs % i]);
        }
    }
    // Q consecutive entries form one workgroup
    while (list.size() % Q) list.push_back(0xffffffffu);
    if (g_order == 1) {
        while (list.size() % (8 * Q)) list.push_back(0xffffffffu);
        return list;
    }
    const size_t nwg = list.size() / Q;
    const size_t cs = (nwg + 7) / 8;
    std::vector<uint32_t> out(cs * 8 * Q, 0xffffffffu);
    for (size_t x = 0; x < 8; ++x)
        for (size_t slot = 0; slot < cs; ++slot)
            if (x * cs + slot < nwg)
                for (int q = 0; q < Q; ++q) out[(slot * 8 + x) * Q + q] = list[(x * cs + slot) * Q + q];
    return out;
}

static bool check(Ctx& c, const char* name) {
    CK(hipMemcpy(c.got.data(), c.dB, c.N * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < c.N; ++i)
        if (std::memcmp(&c.got[i], &c.want[i], 8) != 0) ++bad;
    if (bad) printf("   !! %s: %zu mismatching elements\n", name, bad);
    return bad == 0;
}

template <int L0, int L1, int L2, int L3, int NTLOG, int SWZ, bool NTS, int MODE, int Q, int GROUP = 1>
static void run_orb(Ctx& c, bool stamps = false) {
    const int lg[4] = {L0, L1, L2, L3};
    const int ml = *std::max_element(lg, lg + 4);
    if ((1 << ml) > c.n) return;
    std::vector<uint32_t> boxes = grouped_boxes(c.n, lg, GROUP, Q);
    uint32_t* db;
    CK(hipMalloc(&db, boxes.size() * 4));
    CK(hipMemcpy(db, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    constexpr int TL = L0 + L1 + L2 + L3;
    const size_t lds = MODE != 5 ? ((size_t)4 * 8 << TL) * Q : 0;
    auto kern = k_orb<L0, L1, L2, L3, NTLOG, SWZ, NTS, MODE, Q, false>;
    auto kern_st = k_orb<L0, L1, L2, L3, NTLOG, SWZ, NTS, MODE, Q, true>;
    const unsigned grid = (unsigned)(boxes.size() / Q);
    if (lds > 64 * 1024) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)kern_st, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    CK(hipMemsetAsync(c.dB, 0xff, c.N * 8, c.st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Q << NTLOG), lds, c.st, c.dA, c.dB, c.n, db, (u64*)nullptr);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c.st));
    bool ok = true;
    if (MODE == 0 && c.verify) ok = check(c, "orbit");
    const unsigned waves = grid * ((Q << NTLOG) >> 6);
    const SpanStats p = time_graph(c, c.reps, waves, false, [&](u64* s) { hipLaunchKernelGGL(kern, dim3(grid), dim3(Q << NTLOG), lds, c.st, c.dA, c.dB, c.n, db, s); });
    static const char* mn[] = {"orbit", "?", "?", "?", "orbit-readside", "orbit-writeside"};
    char what[160];
    snprintf(what, sizeof what, "%-15s ord %d box %2dx%2dx%2dx%2d Q %d grp %d lanes %4d swz %d nts %d wgs %6u lds %6zu", mn[MODE], g_order, 1 << L0, 1 << L1, 1 << L2, 1 << L3, Q, 1 << GROUP,
             Q << NTLOG, SWZ, (int)NTS, grid, lds);
    printf("n=%3d %s : %9.2f us  %7.1f GB/s  (%.1f %% of 8 TB/s) %s\n", c.n, what, p.us_events, 2.0 * c.N * 8 / p.us_events * 1e-3, 2.0 * c.N * 8 / p.us_events * 1e-3 / 80.0,
           ok ? "ok" : "WRONG");
    if (stamps) {
        const SpanStats s = time_graph(c, c.reps, waves, true, [&](u64* sp) { hipLaunchKernelGGL(kern_st, dim3(grid), dim3(Q << NTLOG), lds, c.st, c.dA, c.dB, c.n, db, sp); });
        print_span(mn[MODE], p, s);
    }
    fflush(stdout);
    CK(hipFree(db));
}

template <bool NTS, int MODE>
static void run_copy(Ctx& c, const char* name, bool stamps) {
    const unsigned g = (unsigned)(c.N / 1024);
    const SpanStats p = time_graph(c, c.reps, g * 4, false, [&](u64* s) { hipLaunchKernelGGL((k_copy<NTS, MODE, false>), dim3(g), dim3(256), 0, c.st, c.dA, c.dB, s); });
    printf("n=%3d %-28s: %9.2f us  %7.1f GB/s\n", c.n, name, p.us_events, 2.0 * c.N * 8 / p.us_events * 1e-3);
    if (stamps) {
        const SpanStats s = time_graph(c, c.reps, g * 4, true, [&](u64* sp) { hipLaunchKernelGGL((k_copy<NTS, MODE, true>), dim3(g), dim3(256), 0, c.st, c.dA, c.dB, sp); });
        print_span(name, p, s);
    }
    fflush(stdout);
}

// the bench step (permutedims! into B, 4-way sum into C) as an in-order pair and as two branches of one graph
static void run_forkjoin(Ctx& c) {
    if (c.n != 32) return;
    const int lg[4] = {2, 2, 2, 2};
    std::vector<uint32_t> boxes = grouped_boxes(c.n, lg, 1, 1);
    uint32_t* db;
    CK(hipMalloc(&db, boxes.size() * 4));
    CK(hipMemcpy(db, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    auto korb = k_orb<2, 2, 2, 2, 7, 3, false, 0, 1, false>;
    const unsigned gorb = (unsigned)boxes.size(), gperm = (unsigned)(c.N / 1024);
    const size_t lds = (size_t)4 * 8 << 8;
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int reps = 200;
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < reps; ++i) {
            if (variant == 0) {  // in order on one stream
                hipLaunchKernelGGL(k_perm4321<true>, dim3(gperm), dim3(256), 0, c.st, c.dA, c.dC, c.nlog, (u64*)nullptr);
                hipLaunchKernelGGL(korb, dim3(gorb), dim3(128), lds, c.st, c.dA, c.dB, c.n, db, (u64*)nullptr);
            } else if (variant == 1) {  // fork / join every step
                CK(hipEventRecord(fork, c.st));
                CK(hipStreamWaitEvent(c.st2, fork, 0));
                hipLaunchKernelGGL(k_perm4321<true>, dim3(gperm), dim3(256), 0, c.st, c.dA, c.dC, c.nlog, (u64*)nullptr);
                hipLaunchKernelGGL(korb, dim3(gorb), dim3(128), lds, c.st2, c.dA, c.dB, c.n, db, (u64*)nullptr);
                CK(hipEventRecord(join, c.st2));
                CK(hipStreamWaitEvent(c.st, join, 0));
            } else {  // two independent chains, joined once at the end (each output has its own stream order)
                if (i == 0) {
                    CK(hipEventRecord(fork, c.st));
                    CK(hipStreamWaitEvent(c.st2, fork, 0));
                }
                hipLaunchKernelGGL(k_perm4321<true>, dim3(gperm), dim3(256), 0, c.st, c.dA, c.dC, c.nlog, (u64*)nullptr);
                hipLaunchKernelGGL(korb, dim3(gorb), dim3(128), lds, c.st2, c.dA, c.dB, c.n, db, (u64*)nullptr);
                if (i == reps - 1) {
                    CK(hipEventRecord(join, c.st2));
                    CK(hipStreamWaitEvent(c.st, join, 0));
                }
            }
        }
        CK(hipStreamEndCapture(c.st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, c.st));
        CK(hipStreamSynchronize(c.st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int t = 0; t < 5; ++t) {
            CK(hipEventRecord(e0, c.st));
            CK(hipGraphLaunch(ge, c.st));
            CK(hipEventRecord(e1, c.st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        static const char* vn[] = {"in order, one stream", "fork/join every step", "two chains, one join"};
        const double us = best * 1000.0 / reps;
        printf("n= 32 step (perm4321 + orbit sum) %-22s: %7.2f us per step  %7.1f GB/s (%.1f %% of 8 TB/s)\n", vn[variant], us, 4.0 * c.N * 8 / us * 1e-3,
               4.0 * c.N * 8 / us * 1e-3 / 80.0);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // correctness of the plain permute kernel
    hipLaunchKernelGGL(k_perm4321<true>, dim3(gperm), dim3(256), 0, c.st, c.dA, c.dC, c.nlog, (u64*)nullptr);
    CK(hipStreamSynchronize(c.st));
    std::vector<double> got(c.N);
    CK(hipMemcpy(got.data(), c.dC, c.N * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    const size_t n1 = c.n, n2 = n1 * n1, n3 = n2 * n1;
    for (size_t j3 = 0; j3 < n1; ++j3)
        for (size_t j2 = 0; j2 < n1; ++j2)
            for (size_t j1 = 0; j1 < n1; ++j1)
                for (size_t j0 = 0; j0 < n1; ++j0)
                    if (got[j0 + n1 * j1 + n2 * j2 + n3 * j3] != c.hA[j3 + n1 * j2 + n2 * j1 + n3 * j0]) ++bad;
    printf("      perm4321 check: %s\n", bad ? "WRONG" : "ok");
    fflush(stdout);
    CK(hipFree(db));
}

int main(int argc, char** argv) {
    std::vector<int> sizes;
    bool sweep = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "sweep")) sweep = true;
        else sizes.push_back(atoi(argv[i]));
    }
    if (sizes.empty()) sizes = {32, 64, 128};
    {
        int dev = 0, khz = 0;
        CK(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) g_tick_ns = 1e6 / khz;
        printf("wall clock rate %d kHz -> %.2f ns per tick\n", khz, g_tick_ns);
    }
    for (int n : sizes) {
        Ctx c;
        c.n = n;
        c.nlog = 0;
        while ((1 << c.nlog) < n) ++c.nlog;
        c.N = (size_t)n * n * n * n;
        c.reps = n <= 32 ? 200 : (n <= 64 ? 40 : 4);
        CK(hipStreamCreate(&c.st));
        CK(hipStreamCreate(&c.st2));
        c.stamp_cap = (size_t)16 << 20;  // 128 MiB of stamps
        CK(hipMalloc(&c.dstamps, c.stamp_cap * 8));
        c.verify = !sweep;
        if (sweep) {
            // the size sweep (any multiple of 8): timing only -- read side, write side and the whole orbit kernel on 8^4
            // cubes, with the work list grouped / dealt round-robin / shuffled
            CK(hipMalloc(&c.dA, c.N * 8));
            CK(hipMalloc(&c.dB, c.N * 8));
            CK(hipMemset(c.dA, 0, c.N * 8));
            c.reps = n <= 64 ? 20 : 3;
            run_copy<false, 0>(c, "copy", false);
            run_copy<false, 4>(c, "linear read only", false);
            run_copy<false, 5>(c, "linear write only", false);
            for (int ord = 0; ord < 3; ++ord) {
                g_order = ord;
                run_orb<3, 3, 3, 3, 10, 3, false, 0, 1>(c);
                run_orb<3, 3, 3, 3, 10, 3, false, 4, 1>(c);
                run_orb<3, 3, 3, 3, 10, 3, false, 5, 1>(c);
            }
            g_order = 0;
            run_orb<3, 3, 3, 3, 10, 3, true, 5, 1>(c);
            run_orb<3, 3, 3, 3, 10, 3, false, 5, 1, 2>(c);
            CK(hipFree(c.dA));
            CK(hipFree(c.dB));
            CK(hipFree(c.dstamps));
            CK(hipStreamDestroy(c.st));
            CK(hipStreamDestroy(c.st2));
            continue;
        }
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
        CK(hipMalloc(&c.dC, c.N * 8));
        CK(hipMemcpy(c.dA, c.hA.data(), c.N * 8, hipMemcpyHostToDevice));
        const bool st = n <= 32;  // stamps: the launch-bound size
        if (st) {
            const SpanStats p = time_graph(c, c.reps, 1024 * 4, false, [&](u64* sp) { hipLaunchKernelGGL(k_empty<false>, dim3(1024), dim3(256), 0, c.st, sp); });
            const SpanStats q = time_graph(c, c.reps, 1024 * 4, true, [&](u64* sp) { hipLaunchKernelGGL(k_empty<true>, dim3(1024), dim3(256), 0, c.st, sp); });
            printf("n=%3d empty kernel, 1024 x 256   : %9.2f us\n", n, p.us_events);
            print_span("empty", p, q);
        }
        run_copy<false, 0>(c, "copy", st);
        run_copy<true, 0>(c, "copy (nt stores)", st);
        run_copy<false, 4>(c, "linear read only", st);
        run_copy<false, 5>(c, "linear write only", st);
        run_copy<true, 5>(c, "linear write only (nt)", st);
        if (n == 32) {
            run_orb<2, 2, 2, 2, 7, 3, false, 0, 1>(c, true);
            run_orb<2, 2, 2, 2, 7, 3, false, 4, 1>(c, true);
            run_orb<2, 2, 2, 2, 7, 3, false, 5, 1>(c, true);
            run_orb<2, 2, 2, 2, 7, 3, false, 0, 2>(c, true);
            run_orb<2, 2, 2, 2, 6, 3, false, 0, 1>(c, true);   // one wave per orbit
            run_orb<2, 2, 2, 2, 6, 3, false, 0, 2>(c, true);
            run_orb<2, 2, 2, 2, 6, 3, false, 0, 4>(c);
            run_orb<3, 2, 2, 2, 7, 1, false, 0, 1>(c);
            run_orb<3, 2, 2, 2, 7, 3, false, 0, 1>(c);
            run_orb<3, 2, 2, 2, 8, 3, false, 0, 1>(c);
            run_orb<3, 2, 2, 2, 8, 2, false, 0, 1>(c);
            run_orb<2, 2, 2, 2, 7, 3, false, 0, 4>(c, true);
            run_orb<3, 2, 2, 2, 8, 1, false, 0, 1>(c, true);
            run_orb<3, 2, 2, 2, 8, 1, false, 0, 2>(c);
            run_orb<3, 3, 2, 2, 8, 1, false, 0, 1>(c, true);
            run_orb<3, 3, 2, 2, 9, 1, false, 0, 1>(c);
            run_orb<3, 2, 3, 2, 8, 1, false, 0, 1>(c);
            run_orb<3, 3, 3, 2, 9, 2, false, 0, 1>(c, true);
            run_orb<3, 3, 3, 2, 10, 2, false, 0, 1>(c);
            run_orb<3, 3, 3, 2, 10, 2, true, 0, 1>(c);
            run_orb<3, 3, 3, 3, 10, 3, false, 0, 1>(c, true);
            run_orb<3, 3, 3, 3, 10, 3, true, 0, 1>(c);
            run_orb<4, 2, 2, 2, 9, 1, false, 0, 1>(c);
            run_orb<4, 3, 2, 2, 9, 1, false, 0, 1>(c);
            run_orb<3, 3, 3, 2, 9, 2, false, 4, 1>(c);
            run_orb<3, 3, 3, 2, 9, 2, false, 5, 1>(c);
            run_orb<3, 3, 3, 2, 9, 2, true, 5, 1>(c);
            run_forkjoin(c);
        } else {
            run_orb<3, 3, 3, 3, 10, 3, false, 0, 1>(c);
            run_orb<3, 3, 3, 3, 10, 3, false, 0, 1, 2>(c);
            run_orb<3, 3, 3, 3, 10, 3, false, 4, 1>(c);
            run_orb<3, 3, 3, 3, 10, 3, false, 5, 1>(c);
            run_orb<3, 3, 3, 3, 10, 3, true, 5, 1>(c);
            run_orb<3, 3, 3, 2, 9, 2, false, 0, 1>(c);
            run_orb<3, 3, 2, 2, 8, 1, false, 0, 1>(c);
            if (n >= 128) {
                run_orb<3, 3, 3, 3, 10, 3, false, 0, 1, 3>(c);
                run_orb<4, 4, 2, 2, 10, 3, false, 0, 1>(c);     // half the boxes with 128-byte runs, half with 32
                run_orb<4, 3, 3, 2, 10, 3, false, 0, 1>(c);     // 128 / 64 / 64 / 32-byte runs
                run_orb<4, 4, 2, 2, 10, 3, false, 0, 1, 2>(c);
                run_orb<4, 3, 3, 2, 10, 3, false, 0, 1, 2>(c);
            }
        }
        CK(hipFree(c.dA));
        CK(hipFree(c.dB));
        CK(hipFree(c.dC));
        CK(hipFree(c.dstamps));
        CK(hipStreamDestroy(c.st));
        CK(hipStreamDestroy(c.st2));
    }
    return 0;
}
