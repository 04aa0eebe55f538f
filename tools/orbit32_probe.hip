// orbit32_probe.hip -- round 6: where do the 3.3 us of the 4-way permuted sum at 32^4 Float64 go, and what bounds them?
//   Stand-alone, no library.  C[i] = ((A[i] + A[r i]) + A[r^2 i]) + A[r^3 i], r = cyclic shift of the 4 indices.
//
//   part 1  what the memory system does with the PATTERN alone: workgroups of 128 lanes move 4 boxes of 256 elements (16 B per lane
//           and access) -- read only / write only / copy -- for boxes of 4x4x4x4 (32-byte runs, the orbit kernel's pattern),
//           8x4x4x2 (64 B), 16x4x2x2 (128 B), 32x2x2x2 (256 B) and contiguous; with 1/4, 1/2 and all of the workgroups; and on a
//           padded array (row strides 34 / 34*33 / ... elements: no power-of-two strides).
//   part 2  the sum itself in variants: origin row from a table in memory (the product kernel's form), from a table inside the
//           kernel-argument segment (one scalar round trip instead of two), from pure arithmetic (timing only: the covering is
//           wrong), one wave per orbit without a workgroup barrier, non-temporal / sc1 loads, store modes.
//   part 3  per-wave phase stamps (s_memtime) of the table form and the kernel-argument form.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/orbit32_probe.hip -o tools/bin/orbit32_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

typedef unsigned long long u64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

static constexpr int N = 32;

// ---------------------------------------------------------------------------------------------------------------------------
__global__ void k_empty() {}

// linear: 256 lanes, each 2 x 16 B
template <int MODE>
__global__ void __launch_bounds__(256) k_lin(const double* __restrict__ A, double* __restrict__ C, double* sink) {
    const size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    d2 x0 = {1.0, 2.0}, x1 = {3.0, 4.0};
    if (MODE != 2) {
        x0 = *reinterpret_cast<const d2*>(A + i);
        x1 = *reinterpret_cast<const d2*>(A + i + 512);
    }
    if (MODE == 1) {
        if (x0.x + x1.y == 12345.678) *sink = x0.x;
    } else {
        *reinterpret_cast<d2*>(C + i) = x0;
        *reinterpret_cast<d2*>(C + i + 512) = x1;
    }
}

// pattern: a workgroup of 128 lanes moves 4 boxes of R0 x R1 x R2 x R3 = 256 elements; box origins (element offsets) from a table
// LD: 0 plain, 1 nontemporal, 2 sc1 ; ST: 0 plain, 1 nt, 2 sc1 (write-through)
struct PatArgs {
    const double* A;
    double* C;
    const uint32_t* rows;  // 4 origins per workgroup
    double* sink;
    uint32_t s1, s2, s3;  // element strides of dims 1..3
};

__device__ __forceinline__ d2 ld16(const double* p, int mode) {
    if (mode == 1) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    if (mode == 2) {
        d2 r;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        return r;
    }
    return *reinterpret_cast<const d2*>(p);
}
__device__ __forceinline__ void st16(double* p, d2 v, int mode) {
    if (mode == 1) {
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
    } else if (mode == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<d2*>(p) = v;
    }
}

template <int R0, int R1, int R2, int R3, int MODE, int LD, int ST>
__global__ void __launch_bounds__(128) k_pat(const PatArgs a) {
    static_assert(R0 * R1 * R2 * R3 == 256, "256 elements per box");
    const uint4 row = reinterpret_cast<const uint4*>(a.rows)[blockIdx.x];
    const bool idle = row.x == 0xffffffffu;  // padding rows of the orbit list: move box 0 once more
    const uint32_t org[4] = {idle ? 0 : row.x, idle ? 0 : row.y, idle ? 0 : row.z, idle ? 0 : row.w};
    const uint32_t e = threadIdx.x * 2;
    const uint32_t c0 = e % R0, c1 = (e / R0) % R1, c2 = (e / (R0 * R1)) % R2, c3 = e / (R0 * R1 * R2);
    const uint32_t goff = c0 + c1 * a.s1 + c2 * a.s2 + c3 * a.s3;
    d2 x[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        x[g] = d2{1.0 + g, 2.0};
        if (MODE != 2) x[g] = ld16(a.A + org[g] + goff, LD);
    }
    if (MODE == 1) {
        if (x[0].x + x[1].y + x[2].x + x[3].y == 12345.678) *a.sink = x[0].x;
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) st16(a.C + org[g] + goff, x[g], ST);
}

// generic rotated-box pattern: a workgroup of LANES lanes moves 4 boxes of 2 * LANES elements whose shapes are the four rotations of
// (2^l0, 2^l1, 2^l2, 2^l3); 16 bytes per lane along dim 0
struct PatGArgs {
    const double* A;
    double* C;
    const uint32_t* rows;
    double* sink;
    int32_t lg[4];
};
template <int LANES, int MODE>
__global__ void __launch_bounds__(LANES) k_patg(const PatGArgs a) {
    const uint4 row = reinterpret_cast<const uint4*>(a.rows)[blockIdx.x];
    const uint32_t org[4] = {row.x, row.y, row.z, row.w};
    const uint32_t e = threadIdx.x * 2;
    d2 x[4];
    uint32_t off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t sh = 0, o = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t l = (uint32_t)a.lg[(g + d) & 3];
            o += ((e >> sh) & ((1u << l) - 1u)) << (5 * d);
            sh += l;
        }
        off[g] = org[g] + o;
        x[g] = d2{1.0 + g, 2.0};
        if (MODE != 2) x[g] = *reinterpret_cast<const d2*>(a.A + off[g]);
    }
    if (MODE == 1) {
        if (x[0].x + x[1].y + x[2].x + x[3].y == 12345.678) *a.sink = x[0].x;
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<d2*>(a.C + off[g]) = x[g];
}

// ---------------------------------------------------------------------------------------------------------------------------
// the sum.  Tile 4x4x4x4, slot g of an orbit rooted at tile t = (I,J,K,L) holds tile r^g t = (t[g], t[g+1], t[g+2], t[g+3]).
// An output of slot g at local (c0,c1,c2,c3) adds, for m = 0..3, slot (g+m)%4 at local (c[m], c[m+1], c[m+2], c[m+3]).
struct SumArgs {
    const double* A;
    double* C;
    const uint32_t* rows;  // VAR 0 / 3: 4 element origins per workgroup
    u64* stamps;           // PH builds: 8 words per wave
    uint32_t swz_s1, swz_s2, swz_mask, norb;
    uint32_t reps[528];    // VAR 1 / 4: packed root tile id (3 bits per coordinate), two workgroups per word; 0xffff = idle
};

__device__ __forceinline__ uint32_t tile_org(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
    return (t0 << 2) + (t1 << 7) + (t2 << 12) + (t3 << 17);  // 4 * (t0 + 32 t1 + 1024 t2 + 32768 t3)
}

// VAR: 0 table in memory, 128 lanes; 1 table in the kernel arguments, 128 lanes; 2 arithmetic (timing only), 128 lanes;
//      3 table in memory, ONE wave per orbit (64 lanes, two repeats), no workgroup barrier; 4 = 1 + one wave
// LD / ST as above.  PH: phase stamps.
struct Reps {
    uint32_t w[528];
};
struct SumRef {  // what the body reads: the same fields, however they were passed
    const double* A;
    double* C;
    const uint32_t* rows;
    u64* stamps;
    uint32_t swz_s1, swz_s2, swz_mask;
    const uint32_t* reps;  // points INTO the kernel-argument segment
};

template <int VAR, int LD, int ST, bool PH>
__device__ __forceinline__ void sum_body(const SumRef a) {
    constexpr bool ONEWAVE = VAR == 3 || VAR == 4;
    constexpr int NT = ONEWAVE ? 64 : 128;
    constexpr int NREP = ONEWAVE ? 2 : 1;
    __shared__ __attribute__((aligned(16))) double lds[4 * 256];
    u64 t[8], rt0 = 0;
    if (PH) {
        rt0 = wall_clock64();
        t[0] = __builtin_readcyclecounter();
    }
    uint32_t org[4];
    bool live = true;
    if (VAR == 0 || VAR == 3) {
        const uint4 row = reinterpret_cast<const uint4*>(a.rows)[blockIdx.x];
        live = row.x != 0xffffffffu;
        org[0] = live ? row.x : 0;
        org[1] = live ? row.y : 0;
        org[2] = live ? row.z : 0;
        org[3] = live ? row.w : 0;
    } else {
        uint32_t id;
        if (VAR == 2) {
            id = (blockIdx.x * 3u + (blockIdx.x >> 3)) & 4095u;  // some tile: the covering is wrong, the traffic is alike
        } else {
            id = (a.reps[blockIdx.x >> 1] >> ((blockIdx.x & 1) * 16)) & 0xffffu;  // a scalar load
            live = id != 0xffffu;
            id = live ? id : 0;
        }
        const uint32_t t0 = id & 7, t1 = (id >> 3) & 7, t2 = (id >> 6) & 7, t3 = (id >> 9) & 7;
        org[0] = tile_org(t0, t1, t2, t3);
        org[1] = tile_org(t1, t2, t3, t0);
        org[2] = tile_org(t2, t3, t0, t1);
        org[3] = tile_org(t3, t0, t1, t2);
    }
    if (PH) {
        asm volatile("" ::"s"(org[0]), "s"(org[1]), "s"(org[2]), "s"(org[3]));
        t[1] = __builtin_readcyclecounter();
    }
    const uint32_t tid = threadIdx.x;
    uint32_t goff[NREP], c[NREP][4];
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        const uint32_t e = (r * NT + tid) * 2;
        c[r][0] = e & 3;
        c[r][1] = (e >> 2) & 3;
        c[r][2] = (e >> 4) & 3;
        c[r][3] = (e >> 6) & 3;
        goff[r] = c[r][0] + (c[r][1] << 5) + (c[r][2] << 10) + (c[r][3] << 15);
    }
    d2 x[4][NREP];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const double* p = a.A + org[g] + goff[r];
            if (LD == 1) x[g][r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
            else if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[g][r]) : "v"(p) : "memory");
            else x[g][r] = *reinterpret_cast<const d2*>(p);
        }
    if (PH) {
        asm volatile("" ::: "memory");
        t[2] = __builtin_readcyclecounter();
    }
    const uint32_t s1 = a.swz_s1, s2 = a.swz_s2, mask = a.swz_mask;
    auto swz = [&](uint32_t i) { return i ^ (((i >> s1) ^ (i >> s2)) & mask); };
    // LDS read indices of the permuted views (m = 1..3), sub-element h moves along natural dim 0 = view position (4 - m) % 4
    uint32_t lr[4][NREP];
#pragma unroll
    for (int r = 0; r < NREP; ++r)
#pragma unroll
        for (int m = 1; m < 4; ++m)
            lr[m][r] = c[r][m] | (c[r][(m + 1) & 3] << 2) | (c[r][(m + 2) & 3] << 4) | (c[r][(m + 3) & 3] << 6);
    if (LD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PH) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(x[g][r]));
        t[3] = __builtin_readcyclecounter();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const uint32_t e = (r * NT + tid) * 2;
            lds[g * 256 + swz(e)] = x[g][r].x;
            lds[g * 256 + swz(e + 1)] = x[g][r].y;
        }
    if (ONEWAVE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS operations complete in order
    } else {
        __syncthreads();
    }
    if (PH) t[4] = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        double v[4][2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    // natural dim 0 sits at position (4 - m) & 3 of the view's coordinate list -> bit 2 * ((4 - m) & 3)
                    const uint32_t idx = lr[m][r] | ((uint32_t)h << (2 * ((4 - m) & 3)));
                    v[g][h][m] = lds[((g + m) & 3) * 256 + swz(idx)];
                }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            x[g][r].x = ((x[g][r].x + v[g][0][1]) + v[g][0][2]) + v[g][0][3];
            x[g][r].y = ((x[g][r].y + v[g][1][1]) + v[g][1][2]) + v[g][1][3];
        }
    }
    if (PH) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(x[g][r]));
        t[5] = __builtin_readcyclecounter();
    }
    if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < NREP; ++r) st16(a.C + org[g] + goff[r], x[g][r], ST);
    }
    if (PH) {
        asm volatile("" ::: "memory");
        t[6] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t[7] = __builtin_readcyclecounter();
        const u64 rt1 = wall_clock64();
        if ((tid & 63) == 0 && a.stamps) {
            u64* o = a.stamps + ((size_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 10;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = t[i];
            o[8] = rt0;
            o[9] = rt1;
        }
    } else if (ST == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int VAR, int LD, int ST, bool PH>
__global__ void __launch_bounds__(128) k_sum(const SumArgs a) {
    sum_body<VAR, LD, ST, PH>(SumRef{a.A, a.C, a.rows, a.stamps, a.swz_s1, a.swz_s2, a.swz_mask, a.reps});
}
// the same with leading SCALAR parameters: with -mllvm -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the wave (no
// scalar load of the kernel arguments before the table row / the first address can be formed)
template <int VAR, int LD, int ST, bool PH>
__global__ void __launch_bounds__(128) k_sumx(const double* A, double* C, const uint32_t* rows, u64* stamps, uint32_t s1, uint32_t s2, uint32_t mask,
                                              uint32_t pad, const Reps reps) {
    sum_body<VAR, LD, ST, PH>(SumRef{A, C, rows, stamps, s1, s2, mask, reps.w});
}

// ---------------------------------------------------------------------------------------------------------------------------
struct Ctx {
    hipStream_t st;
    hipEvent_t e0, e1;
};

// us per launch: hipGraph of `reps` launches, best of `tries` replays
static double time_graph(Ctx& c, const std::function<void(hipStream_t)>& launch, int reps = 200, int tries = 7) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch(c.st);
    CK(hipStreamEndCapture(c.st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, c.st));
    CK(hipStreamSynchronize(c.st));
    double best = 1e30;
    for (int t = 0; t < tries; ++t) {
        CK(hipEventRecord(c.e0, c.st));
        CK(hipGraphLaunch(ge, c.st));
        CK(hipEventRecord(c.e1, c.st));
        CK(hipEventSynchronize(c.e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, c.e0, c.e1));
        best = std::min(best, (double)ms * 1000.0 / reps);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return best;
}

struct Swz {
    uint32_t s1 = 31, s2 = 31, mask = 0;
    uint32_t operator()(uint32_t i) const { return i ^ (((i >> s1) ^ (i >> s2)) & mask); }
};

// bank-conflict count of the transposing reads (half-waves of 32 lanes over 32 slots of 8 bytes), lanes as in k_sum
static long swz_cost(const Swz& s, int nt, int nrep) {
    long cost = 0;
    for (int m = 1; m < 4; ++m)
        for (int r = 0; r < nrep; ++r)
            for (int h = 0; h < 2; ++h)
                for (int w0 = 0; w0 < nt; w0 += 32) {
                    uint32_t first[32];
                    int cnt[32] = {0}, worst = 0;
                    for (int lane = 0; lane < 32; ++lane) {
                        const uint32_t e = (uint32_t)(r * nt + w0 + lane) * 2;
                        uint32_t cc[4] = {e & 3, (e >> 2) & 3, (e >> 4) & 3, (e >> 6) & 3};
                        uint32_t idx = cc[m] | (cc[(m + 1) & 3] << 2) | (cc[(m + 2) & 3] << 4) | (cc[(m + 3) & 3] << 6);
                        idx |= (uint32_t)h << (2 * ((4 - m) & 3));
                        const uint32_t l = s(idx), q = l % 32;
                        if (cnt[q] == 0) {
                            first[q] = l;
                            cnt[q] = 1;
                        } else if (first[q] != l) {
                            ++cnt[q];
                        }
                    }
                    for (int q = 0; q < 32; ++q) worst = std::max(worst, cnt[q]);
                    cost += worst - 1;
                }
    return cost;
}
static Swz choose_swz(int nt, int nrep) {
    Swz best;
    long bc = swz_cost(best, nt, nrep);
    for (uint32_t s1 = 1; s1 <= 6 && bc > 0; ++s1)
        for (uint32_t s2 = s1 + 1; s2 <= 13 && bc > 0; ++s2) {
            Swz t;
            t.s1 = s1;
            t.s2 = s2 == 13 ? 31 : s2;
            t.mask = 31;
            const long cc = swz_cost(t, nt, nrep);
            if (cc < bc) {
                bc = cc;
                best = t;
            }
        }
    std::printf("   swizzle for %d lanes x %d: s1=%u s2=%u mask=%u, conflict cost %ld\n", nt, nrep, best.s1, best.s2, best.mask, bc);
    return best;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !std::strcmp(argv[1], "quick");
    Ctx c;
    CK(hipStreamCreate(&c.st));
    CK(hipEventCreate(&c.e0));
    CK(hipEventCreate(&c.e1));
    const size_t NE = (size_t)N * N * N * N;
    const size_t PADE = (size_t)34 * 33 * 35 * 33 + 64;  // padded layout: strides 34, 34*33, 34*33*35
    std::vector<double> hA(NE), hC(NE), want(NE);
    for (size_t i = 0; i < NE; ++i) hA[i] = (double)((i * 2654435761u) % 1000003u);
    for (int l = 0; l < N; ++l)
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) {
                    auto at = [&](int a, int b, int cc, int d) { return hA[a + N * (b + N * (cc + (size_t)N * d))]; };
                    want[i + N * (j + N * (k + (size_t)N * l))] = ((at(i, j, k, l) + at(j, k, l, i)) + at(k, l, i, j)) + at(l, i, j, k);
                }
    double *dA, *dC, *dsink;
    CK(hipMalloc(&dA, PADE * 8));
    CK(hipMalloc(&dC, PADE * 8));
    CK(hipMalloc(&dsink, 64));
    CK(hipMemset(dA, 0, PADE * 8));
    CK(hipMemset(dC, 0, PADE * 8));
    CK(hipMemcpy(dA, hA.data(), NE * 8, hipMemcpyHostToDevice));
    const double bytes = 2.0 * NE * 8;
    auto report = [&](const char* name, double us, double b = -1) {
        if (b < 0) b = bytes;
        std::printf("%-72s %7.3f us  %7.0f GB/s  frac %.3f\n", name, us, b / us * 1e-3, b / us * 1e-3 / 8000.0);
        std::fflush(stdout);
    };

    // ---- floor and linear --------------------------------------------------------------------------------------------------
    report("empty kernel 1048 x 128", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1048), dim3(128), 0, s); }), 0);
    report("linear copy", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(k_lin<0>, dim3(NE / 1024), dim3(256), 0, s, dA, dC, dsink); }));
    report("linear read only", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(k_lin<1>, dim3(NE / 1024), dim3(256), 0, s, dA, dC, dsink); }), bytes / 2);
    report("linear write only", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(k_lin<2>, dim3(NE / 1024), dim3(256), 0, s, dA, dC, dsink); }), bytes / 2);

    // ---- orbit list (roots in memory order, XCD-contiguous runs as the product deals them) -----------------------------------
    std::vector<uint32_t> roots;
    {
        std::vector<char> seen(4096, 0);
        // super-cells of 2 tiles along every dim, as plan_orbit lists them
        for (int cell = 0; cell < 256; ++cell)
            for (int q = 0; q < 16; ++q) {
                int cc[4] = {cell & 3, (cell >> 2) & 3, (cell >> 4) & 3, (cell >> 6) & 3};
                int t[4];
                for (int d = 0; d < 4; ++d) t[d] = cc[d] * 2 + ((q >> d) & 1);
                int root = 1 << 30;
                for (int g = 0; g < 4; ++g) {
                    const int id = t[g] | (t[(g + 1) & 3] << 3) | (t[(g + 2) & 3] << 6) | (t[(g + 3) & 3] << 9);
                    root = std::min(root, id);
                }
                if (seen[root]) continue;
                seen[root] = 1;
                roots.push_back((uint32_t)root);
            }
    }
    const size_t norb = roots.size();
    const size_t cs = (norb + 7) / 8, nwg = cs * 8;
    std::vector<uint32_t> list(nwg, 0xffffffffu);
    for (size_t x = 0; x < 8; ++x)
        for (size_t sl = 0; sl < cs; ++sl)
            if (x * cs + sl < norb) list[sl * 8 + x] = roots[x * cs + sl];
    std::printf("orbits %zu, workgroups %zu\n", norb, nwg);
    auto tile_org_h = [&](uint32_t id, int g, uint32_t s1, uint32_t s2, uint32_t s3) {
        uint32_t t[4] = {id & 7, (id >> 3) & 7, (id >> 6) & 7, (id >> 9) & 7};
        return 4 * (t[g] + t[(g + 1) & 3] * s1 + t[(g + 2) & 3] * s2 + t[(g + 3) & 3] * s3);
    };
    auto make_rows = [&](uint32_t s1, uint32_t s2, uint32_t s3) {
        std::vector<uint32_t> rows(nwg * 4, 0xffffffffu);
        for (size_t w = 0; w < nwg; ++w)
            if (list[w] != 0xffffffffu)
                for (int g = 0; g < 4; ++g) rows[w * 4 + g] = tile_org_h(list[w], g, s1, s2, s3);
        return rows;
    };
    auto upload = [&](const std::vector<uint32_t>& v) {
        uint32_t* d;
        CK(hipMalloc(&d, v.size() * 4));
        CK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        return d;
    };
    const std::vector<uint32_t> rows_h = make_rows(32, 1024, 32768);
    uint32_t* d_rows = upload(rows_h);
    // pattern rows for the run-length shapes: workgroup w of 1024 moves boxes w, w + 1024, w + 2048, w + 3072 (memory order), idle rows none
    auto box_rows = [&](int R0, int R1, int R2, int R3, uint32_t s1, uint32_t s2, uint32_t s3, bool orbit_like) {
        const int nb0 = N / R0, nb1 = N / R1, nb2 = N / R2;
        std::vector<uint32_t> rows(1024 * 4);
        for (uint32_t w = 0; w < 1024; ++w)
            for (uint32_t g = 0; g < 4; ++g) {
                // XCD-contiguous runs: workgroup w runs on XCD w % 8; give XCD x the x-th eighth of the boxes
                const uint32_t lin = (w % 8) * 128 + w / 8;
                uint32_t b = orbit_like ? (lin + g * 1024) : (lin * 4 + g);
                const uint32_t b0 = b % nb0, b1 = (b / nb0) % nb1, b2 = (b / (nb0 * nb1)) % nb2, b3 = b / (nb0 * nb1 * nb2);
                rows[w * 4 + g] = b0 * R0 + b1 * R1 * s1 + b2 * R2 * s2 + b3 * R3 * s3;
            }
        return rows;
    };

    // ---- part 1: the pattern alone ------------------------------------------------------------------------------------------
    std::printf("\n== part 1: pattern alone (128 lanes move 4 boxes of 256 elements; table row per workgroup) ==\n");
    {
        struct Shape {
            const char* name;
            int r[4];
        };
        auto run_shape = [&](auto kern_rd, auto kern_wr, auto kern_cp, const char* name, const std::vector<uint32_t>& rows, uint32_t s1, uint32_t s2, uint32_t s3,
                             unsigned grid) {
            uint32_t* d = upload(rows);
            PatArgs a{dA, dC, d, dsink, s1, s2, s3};
            char buf[160];
            const double frac = (double)grid / (double)(rows.size() / 4);
            std::snprintf(buf, sizeof buf, "%s read only, %u wgs", name, grid);
            report(buf, time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kern_rd, dim3(grid), dim3(128), 0, s, a); }), bytes / 2 * frac);
            std::snprintf(buf, sizeof buf, "%s write only, %u wgs", name, grid);
            report(buf, time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kern_wr, dim3(grid), dim3(128), 0, s, a); }), bytes / 2 * frac);
            std::snprintf(buf, sizeof buf, "%s copy, %u wgs", name, grid);
            report(buf, time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kern_cp, dim3(grid), dim3(128), 0, s, a); }), bytes * frac);
            CK(hipFree(d));
        };
#define SHAPE(R0, R1, R2, R3, NAME, ROWS, S1, S2, S3, GRID) \
    run_shape(k_pat<R0, R1, R2, R3, 1, 0, 0>, k_pat<R0, R1, R2, R3, 2, 0, 0>, k_pat<R0, R1, R2, R3, 0, 0, 0>, NAME, ROWS, S1, S2, S3, GRID)
        SHAPE(4, 4, 4, 4, "orbit list 4x4x4x4 (32 B runs)", rows_h, 32, 1024, 32768, (unsigned)nwg);
        SHAPE(4, 4, 4, 4, "orbit list 4x4x4x4 (32 B runs)", rows_h, 32, 1024, 32768, (unsigned)nwg / 2);
        SHAPE(4, 4, 4, 4, "orbit list 4x4x4x4 (32 B runs)", rows_h, 32, 1024, 32768, (unsigned)nwg / 4);
        SHAPE(4, 4, 4, 4, "boxes 4x4x4x4, far apart", box_rows(4, 4, 4, 4, 32, 1024, 32768, true), 32, 1024, 32768, 1024);
        SHAPE(4, 4, 4, 4, "boxes 4x4x4x4, neighbours", box_rows(4, 4, 4, 4, 32, 1024, 32768, false), 32, 1024, 32768, 1024);
        SHAPE(8, 4, 4, 2, "boxes 8x4x4x2 (64 B), far apart", box_rows(8, 4, 4, 2, 32, 1024, 32768, true), 32, 1024, 32768, 1024);
        SHAPE(8, 4, 4, 2, "boxes 8x4x4x2 (64 B), neighbours", box_rows(8, 4, 4, 2, 32, 1024, 32768, false), 32, 1024, 32768, 1024);
        SHAPE(16, 4, 2, 2, "boxes 16x4x2x2 (128 B), far apart", box_rows(16, 4, 2, 2, 32, 1024, 32768, true), 32, 1024, 32768, 1024);
        SHAPE(32, 2, 2, 2, "boxes 32x2x2x2 (256 B), far apart", box_rows(32, 2, 2, 2, 32, 1024, 32768, true), 32, 1024, 32768, 1024);
        SHAPE(32, 2, 2, 2, "boxes 32x2x2x2 (256 B), neighbours", box_rows(32, 2, 2, 2, 32, 1024, 32768, false), 32, 1024, 32768, 1024);
        if (!quick) {
            const uint32_t p1 = 34, p2 = 34 * 33, p3 = 34 * 33 * 35;
            SHAPE(4, 4, 4, 4, "PADDED strides: orbit list 4x4x4x4", make_rows(p1, p2, p3), p1, p2, p3, (unsigned)nwg);
            SHAPE(8, 4, 4, 2, "PADDED strides: boxes 8x4x4x2, far apart", box_rows(8, 4, 4, 2, p1, p2, p3, true), p1, p2, p3, 1024);
        }
        // load / store modes on the orbit list
        {
            PatArgs a{dA, dC, d_rows, dsink, 32, 1024, 32768};
            report("orbit list 4x4x4x4 read only, nt loads", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 1, 1, 0>), dim3(nwg), dim3(128), 0, s, a); }), bytes / 2);
            report("orbit list 4x4x4x4 write only, nt stores", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 2, 0, 1>), dim3(nwg), dim3(128), 0, s, a); }), bytes / 2);
            report("orbit list 4x4x4x4 write only, sc1 stores", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 2, 0, 2>), dim3(nwg), dim3(128), 0, s, a); }), bytes / 2);
            report("orbit list 4x4x4x4 copy, nt loads + nt stores", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 0, 1, 1>), dim3(nwg), dim3(128), 0, s, a); }));
        }
    }

    // ---- part 1b: arrangements of the same 4096 tiles over workgroups (pattern copy: 4 tiles per workgroup, each tile once) ----------
    std::printf("\n== part 1b: which tiles meet in a workgroup / an XCD, and how many workgroups (pattern copy, 4 tiles of 4x4x4x4 per workgroup) ==\n");
    {
        auto tid_org = [&](uint32_t id) { return 4 * ((id & 7) + ((id >> 3) & 7) * 32 + ((id >> 6) & 7) * 1024 + ((id >> 9) & 7) * 32768); };
        auto rot = [&](uint32_t id, int g) {
            uint32_t t[4] = {id & 7, (id >> 3) & 7, (id >> 6) & 7, (id >> 9) & 7};
            return t[g & 3] | (t[(g + 1) & 3] << 3) | (t[(g + 2) & 3] << 6) | (t[(g + 3) & 3] << 9);
        };
        typedef std::vector<std::array<uint32_t, 4>> WgList;  // 4 tile ids per workgroup
        auto deal_xcd = [&](const WgList& in) {  // XCD-contiguous runs, padded with copies of the first entry
            const size_t per = (in.size() + 7) / 8;
            WgList out(per * 8, in[0]);
            for (size_t x = 0; x < 8; ++x)
                for (size_t sl = 0; sl < per; ++sl)
                    if (x * per + sl < in.size()) out[sl * 8 + x] = in[x * per + sl];
            return out;
        };
        auto run = [&](const char* name, const WgList& wl) {
            std::vector<uint32_t> rows(wl.size() * 4);
            for (size_t w = 0; w < wl.size(); ++w)
                for (int g = 0; g < 4; ++g) rows[w * 4 + g] = tid_org(wl[w][g]);
            uint32_t* d = upload(rows);
            PatArgs a{dA, dC, d, dsink, 32, 1024, 32768};
            const unsigned grid = (unsigned)wl.size();
            const double rd = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 1, 0, 0>), dim3(grid), dim3(128), 0, s, a); });
            const double wr = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 2, 0, 0>), dim3(grid), dim3(128), 0, s, a); });
            const double cp = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL((k_pat<4, 4, 4, 4, 0, 0, 0>), dim3(grid), dim3(128), 0, s, a); });
            std::printf("%-78s %5u wgs | read %6.3f | write %6.3f | copy %6.3f us\n", name, grid, rd, wr, cp);
            std::fflush(stdout);
            CK(hipFree(d));
        };
        auto orbit_wgs = [&](const std::vector<uint32_t>& rts) {
            WgList wl;
            for (uint32_t r : rts) wl.push_back({rot(r, 0), rot(r, 1), rot(r, 2), rot(r, 3)});
            return wl;
        };
        // roots by orbit size
        std::vector<uint32_t> full, half, single;
        for (uint32_t r : roots) {
            if (rot(r, 1) == r) single.push_back(r);
            else if (rot(r, 2) == r) half.push_back(r);
            else full.push_back(r);
        }
        std::printf("orbits of size 4 / 2 / 1: %zu / %zu / %zu\n", full.size(), half.size(), single.size());
        run("A1 product list: supercells of 2, XCD-contiguous (degenerate orbits repeat tiles)", deal_xcd(orbit_wgs(roots)));
        {
            std::vector<uint32_t> r2(roots.begin(), roots.begin() + 1024);
            run("A2 the same, first 1024 orbits only (timing: is it the workgroup count?)", deal_xcd(orbit_wgs(r2)));
        }
        WgList packed;
        {
            // packed: full orbits as they come; two orbits of size 2 share a workgroup; four fixed tiles share one: 1008 + 14 + 2 = 1024
            std::vector<char> isfull(4096, 0);
            for (uint32_t r : full) isfull[r] = 1;
            std::vector<uint32_t> h = half, sgl = single;
            size_t hi = 0, si = 0;
            for (uint32_t r : roots) {
                if (isfull[r]) {
                    packed.push_back({rot(r, 0), rot(r, 1), rot(r, 2), rot(r, 3)});
                } else if (rot(r, 1) != r) {  // size 2: emit with its partner when the second of a couple comes by
                    if (hi % 2 == 1) packed.push_back({h[hi - 1], rot(h[hi - 1], 1), h[hi], rot(h[hi], 1)});
                    ++hi;
                } else {
                    if (si % 4 == 3) packed.push_back({sgl[si - 3], sgl[si - 2], sgl[si - 1], sgl[si]});
                    ++si;
                }
            }
            run("A3 packed: degenerate orbits share workgroups, every tile once", deal_xcd(packed));
        }
        {
            std::vector<uint32_t> r = roots;
            std::sort(r.begin(), r.end());
            run("A4 roots in memory order, XCD-contiguous", deal_xcd(orbit_wgs(r)));
            run("A5 roots in memory order, dealt round-robin over the XCDs", orbit_wgs(r));
        }
        run("A6 product list dealt round-robin over the XCDs", orbit_wgs(roots));
        {
            // far apart: workgroup w moves tiles lin, lin + 1024, lin + 2048, lin + 3072 in memory order (no orbit structure)
            WgList wl;
            for (uint32_t i = 0; i < 1024; ++i) wl.push_back({i, i + 1024, i + 2048, i + 3072});
            run("A7 no orbit structure: tiles lin + {0,1024,2048,3072}, XCD-contiguous", deal_xcd(wl));
            WgList wl2 = wl;
            for (uint32_t i = 0; i < 24; ++i) wl2.push_back(wl[i * 40]);
            run("A8 the same plus 24 repeated workgroups (1048)", deal_xcd(wl2));
            WgList wl3;
            for (uint32_t i = 0; i < 1024; ++i) wl3.push_back({i * 4, i * 4 + 1, i * 4 + 2, i * 4 + 3});
            run("A9 no orbit structure: four neighbours along dim 0 (whole 128-B lines per workgroup)", deal_xcd(wl3));
        }
        {
            // packed list, orbits sorted so that an XCD's run walks memory in order for slot 0 AND the workgroup count is 1024
            WgList p2 = packed;
            std::sort(p2.begin(), p2.end(), [](const std::array<uint32_t, 4>& x, const std::array<uint32_t, 4>& y) { return x[0] < y[0]; });
            run("A10 packed, sorted by the first tile (memory order), XCD-contiguous", deal_xcd(p2));
        }
    }

    // ---- part 1c: rotated NON-cubic boxes (what grouping 2 / 4 / 8 orbits along dims 0 / 0,1 / 0,1,2 into one workgroup would touch) ------
    std::printf("\n== part 1c: rotated boxes, slot g confined to the quarter l in [8g, 8g+8): every element once; workgroup count = 2^k * 256 ==\n");
    {
        auto runb = [&](auto krd, auto kwr, auto kcp, const char* name, int l0, int l1, int l2, int l3, int lanes) {
            const int lg[4] = {l0, l1, l2, l3};
            const unsigned nw = (unsigned)(NE / 4 / (2 * lanes));
            std::vector<uint32_t> rows(nw * 4);
            for (unsigned w = 0; w < nw; ++w) {
                const unsigned lin = (w % 8) * (nw / 8) + w / 8;  // XCD-contiguous
                for (int g = 0; g < 4; ++g) {
                    // box grid of slot g inside its quarter: extents 2^lg[(g+d)&3] along dim d; the quarter spans 8 along dim 3
                    unsigned b = lin, o = 0;
                    for (int d = 0; d < 4; ++d) {
                        const unsigned ext = 1u << lg[(g + d) & 3];
                        const unsigned nb = (d == 3 ? 8u : 32u) / ext;
                        o += (b % nb) * ext << (5 * d);
                        b /= nb;
                    }
                    rows[w * 4 + g] = o + ((unsigned)g * 8u << 15);
                }
            }
            uint32_t* d = upload(rows);
            PatGArgs a{dA, dC, d, dsink, {l0, l1, l2, l3}};
            const double rd = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(krd, dim3(nw), dim3(lanes), 0, s, a); });
            const double wr = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kwr, dim3(nw), dim3(lanes), 0, s, a); });
            const double cp = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kcp, dim3(nw), dim3(lanes), 0, s, a); });
            CK(hipMemset(dC, 0, NE * 8));
            hipLaunchKernelGGL(kcp, dim3(nw), dim3(lanes), 0, c.st, a);
            CK(hipMemcpy(hC.data(), dC, NE * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < NE; ++i) bad += hC[i] != hA[i];
            std::printf("%-58s %5u wgs x %4d | read %6.3f | write %6.3f | copy %6.3f us %s\n", name, nw, lanes, rd, wr, cp, bad ? "[cover WRONG]" : "[every element once]");
            std::fflush(stdout);
            CK(hipFree(d));
        };
#define RUNB(L, NAME, l0, l1, l2, l3) runb(k_patg<L, 1>, k_patg<L, 2>, k_patg<L, 0>, NAME, l0, l1, l2, l3, L)
        RUNB(128, "4x4x4x4 (one orbit)", 2, 2, 2, 2);
        RUNB(256, "8x4x4x4 and rotations (2 orbits paired along dim 0)", 3, 2, 2, 2);
        RUNB(512, "8x8x4x4 and rotations (4 orbits)", 3, 3, 2, 2);
        RUNB(512, "8x4x8x4 and rotations (4 orbits)", 3, 2, 3, 2);
        RUNB(1024, "8x8x8x4 and rotations (8 orbits)", 3, 3, 3, 2);
        RUNB(256, "16x4x4x2 and rotations", 4, 2, 2, 1);
        RUNB(512, "16x4x4x4 and rotations (4 orbits along dim 0)", 4, 2, 2, 2);
    }

    // ---- part 2: the sum ----------------------------------------------------------------------------------------------------
    std::printf("\n== part 2: the 4-way sum, variants ==\n");
    SumArgs sa;
    std::memset(&sa, 0, sizeof sa);
    sa.A = dA;
    sa.C = dC;
    sa.rows = d_rows;
    sa.stamps = nullptr;
    sa.norb = (uint32_t)norb;
    for (size_t w = 0; w < 1056; ++w) {
        const uint32_t v = (w < nwg && list[w] != 0xffffffffu) ? list[w] : 0xffffu;
        sa.reps[w >> 1] |= v << ((w & 1) * 16);
    }
    const Swz sw128 = choose_swz(128, 1), sw64 = choose_swz(64, 2);
    auto check = [&](const char* name) {
        CK(hipMemcpy(hC.data(), dC, NE * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < NE; ++i) bad += hC[i] != want[i];
        if (bad) std::printf("   !! %s: %zu wrong elements\n", name, bad);
        return bad == 0;
    };
    auto run_sum = [&](auto kern, const char* name, int lanes, const Swz& sw, bool verify) {
        SumArgs a = sa;
        a.swz_s1 = sw.s1;
        a.swz_s2 = sw.s2;
        a.swz_mask = sw.mask;
        CK(hipMemset(dC, 0, NE * 8));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, c.st, a);
        CK(hipStreamSynchronize(c.st));
        bool ok = true;
        if (verify) ok = check(name);
        const double us = time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, s, a); });
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s%s", name, verify ? (ok ? " [ok]" : " [WRONG]") : " [timing only]");
        report(buf, us);
    };
    run_sum(k_sum<0, 0, 0, false>, "sum: table in memory, 128 lanes (product form)", 128, sw128, true);
    run_sum(k_sum<1, 0, 0, false>, "sum: table in kernel arguments, 128 lanes", 128, sw128, true);
    run_sum(k_sum<2, 0, 0, false>, "sum: arithmetic origins, 128 lanes", 128, sw128, false);
    run_sum(k_sum<3, 0, 0, false>, "sum: table in memory, one wave per orbit", 64, sw64, true);
    run_sum(k_sum<4, 0, 0, false>, "sum: table in kernel arguments, one wave per orbit", 64, sw64, true);
    run_sum(k_sum<1, 1, 0, false>, "sum: kernarg table, nt loads", 128, sw128, true);
    run_sum(k_sum<1, 2, 0, false>, "sum: kernarg table, sc1 loads", 128, sw128, true);
    run_sum(k_sum<1, 0, 1, false>, "sum: kernarg table, nt stores", 128, sw128, true);
    run_sum(k_sum<1, 0, 2, false>, "sum: kernarg table, sc1 stores + wait", 128, sw128, true);
    run_sum(k_sum<0, 0, 2, false>, "sum: table in memory, sc1 stores + wait", 128, sw128, true);
    // explicit scalar parameters (preloaded into SGPRs when the binary is built with -mllvm -amdgpu-kernarg-preload-count=16)
    Reps reps_h;
    std::memcpy(reps_h.w, sa.reps, sizeof reps_h.w);
    auto run_sumx = [&](auto kern, const char* name, int lanes, const Swz& sw) {
        CK(hipMemset(dC, 0, NE * 8));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, c.st, (const double*)dA, dC, (const uint32_t*)d_rows, (u64*)nullptr, sw.s1, sw.s2, sw.mask, 0u, reps_h);
        CK(hipStreamSynchronize(c.st));
        const bool ok = check(name);
        const double us = time_graph(c, [&](hipStream_t s) {
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, s, (const double*)dA, dC, (const uint32_t*)d_rows, (u64*)nullptr, sw.s1, sw.s2, sw.mask, 0u, reps_h);
        });
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s%s", name, ok ? " [ok]" : " [WRONG]");
        report(buf, us);
    };
    run_sumx(k_sumx<0, 0, 0, false>, "sum (scalar params): table in memory, 128 lanes", 128, sw128);
    run_sumx(k_sumx<1, 0, 0, false>, "sum (scalar params): table in kernel arguments, 128 lanes", 128, sw128);
    run_sumx(k_sumx<4, 0, 0, false>, "sum (scalar params): kernarg table, one wave per orbit", 64, sw64);
    run_sumx(k_sumx<1, 0, 2, false>, "sum (scalar params): kernarg table, sc1 stores + wait", 128, sw128);

    // ---- part 3: phase stamps -----------------------------------------------------------------------------------------------
    std::printf("\n== part 3: per-wave phases (s_memtime cycles; mean / p50 / p95 over the waves of one launch, 20 launches back to back) ==\n");
    auto phases = [&](auto kern, const char* name, int lanes, const Swz& sw, bool scalar_params = false, unsigned grid_override = 0, const uint32_t* rows_override = nullptr) {
        const int waves = lanes / 64;
        const size_t nw = nwg * waves;
        const int L = 20;
        u64* dst;
        CK(hipMalloc(&dst, nw * 10 * sizeof(u64) * L));
        CK(hipMemset(dst, 0, nw * 10 * sizeof(u64) * L));
        SumArgs a = sa;
        a.swz_s1 = sw.s1;
        a.swz_s2 = sw.s2;
        a.swz_mask = sw.mask;
        (void)scalar_params;
        (void)grid_override;
        (void)rows_override;
        for (int rep = 0; rep < 2; ++rep)
            for (int l = 0; l < L; ++l) {
                a.stamps = dst + (size_t)l * nw * 10;
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, c.st, a);
            }
        CK(hipStreamSynchronize(c.st));
        std::vector<u64> h(nw * 10 * L);
        CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* ph[7] = {"origins known", "loads issued", "data arrived", "parked+barrier", "LDS reads+adds", "stores issued", "stores acked"};
        std::printf("%s\n", name);
        std::vector<double> acc[8];
        double span_sum = 0, life_cyc = 0, life_tick = 0;
        for (int l = 5; l < L; ++l) {
            u64 first = ~0ull, last = 0;  // device wall clock (100 MHz, one clock for the chip; s_memtime is per XCD)
            for (size_t w = 0; w < nw; ++w) {
                const u64* o = &h[((size_t)l * nw + w) * 10];
                if (o[0] == 0) continue;
                first = std::min(first, o[8]);
                last = std::max(last, o[9]);
            }
            span_sum += (double)(last - first);
            for (size_t w = 0; w < nw; ++w) {
                const u64* o = &h[((size_t)l * nw + w) * 10];
                if (o[0] == 0) continue;
                for (int i = 1; i < 8; ++i) acc[i - 1].push_back((double)(o[i] - o[i - 1]));
                acc[7].push_back((double)(o[7] - o[0]));
                life_cyc += (double)(o[7] - o[0]);
                life_tick += (double)(o[9] - o[8]);
            }
        }
        auto stat = [&](std::vector<double>& v, const char* nm) {
            std::sort(v.begin(), v.end());
            double m = 0;
            for (double x : v) m += x;
            m /= v.size();
            std::printf("   %-16s mean %7.0f  p50 %7.0f  p95 %7.0f  max %7.0f cycles\n", nm, m, v[v.size() / 2], v[v.size() * 95 / 100], v.back());
        };
        for (int i = 0; i < 7; ++i) stat(acc[i], ph[i]);
        stat(acc[7], "wave lifetime");
        std::printf("   s_memtime runs at %.0f MHz (lifetimes against the 100-MHz wall clock); launch span, first wave start -> last wave's stores acked: mean %.2f us\n",
                    life_cyc / (life_tick * 0.01), span_sum / (L - 5) * 0.01);
        CK(hipFree(dst));
    };
    auto phasesx = [&](auto kern, const char* name, int lanes, const Swz& sw, bool scalar_params = false, unsigned grid_override = 0, const uint32_t* rows_override = nullptr) {
        const int waves = lanes / 64;
        const size_t nw = nwg * waves;
        const int L = 20;
        u64* dst;
        CK(hipMalloc(&dst, nw * 10 * sizeof(u64) * L));
        CK(hipMemset(dst, 0, nw * 10 * sizeof(u64) * L));
        SumArgs a = sa;
        a.swz_s1 = sw.s1;
        a.swz_s2 = sw.s2;
        a.swz_mask = sw.mask;
        (void)scalar_params;
        (void)grid_override;
        (void)rows_override;
        for (int rep = 0; rep < 2; ++rep)
            for (int l = 0; l < L; ++l) {
                a.stamps = dst + (size_t)l * nw * 10;
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(lanes), 0, c.st, (const double*)dA, dC, (const uint32_t*)d_rows, a.stamps, sw.s1, sw.s2, sw.mask, 0u, reps_h);
            }
        CK(hipStreamSynchronize(c.st));
        std::vector<u64> h(nw * 10 * L);
        CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* ph[7] = {"origins known", "loads issued", "data arrived", "parked+barrier", "LDS reads+adds", "stores issued", "stores acked"};
        std::printf("%s\n", name);
        std::vector<double> acc[8];
        double span_sum = 0, life_cyc = 0, life_tick = 0;
        for (int l = 5; l < L; ++l) {
            u64 first = ~0ull, last = 0;  // device wall clock (100 MHz, one clock for the chip; s_memtime is per XCD)
            for (size_t w = 0; w < nw; ++w) {
                const u64* o = &h[((size_t)l * nw + w) * 10];
                if (o[0] == 0) continue;
                first = std::min(first, o[8]);
                last = std::max(last, o[9]);
            }
            span_sum += (double)(last - first);
            for (size_t w = 0; w < nw; ++w) {
                const u64* o = &h[((size_t)l * nw + w) * 10];
                if (o[0] == 0) continue;
                for (int i = 1; i < 8; ++i) acc[i - 1].push_back((double)(o[i] - o[i - 1]));
                acc[7].push_back((double)(o[7] - o[0]));
                life_cyc += (double)(o[7] - o[0]);
                life_tick += (double)(o[9] - o[8]);
            }
        }
        auto stat = [&](std::vector<double>& v, const char* nm) {
            std::sort(v.begin(), v.end());
            double m = 0;
            for (double x : v) m += x;
            m /= v.size();
            std::printf("   %-16s mean %7.0f  p50 %7.0f  p95 %7.0f  max %7.0f cycles\n", nm, m, v[v.size() / 2], v[v.size() * 95 / 100], v.back());
        };
        for (int i = 0; i < 7; ++i) stat(acc[i], ph[i]);
        stat(acc[7], "wave lifetime");
        std::printf("   s_memtime runs at %.0f MHz (lifetimes against the 100-MHz wall clock); launch span, first wave start -> last wave's stores acked: mean %.2f us\n",
                    life_cyc / (life_tick * 0.01), span_sum / (L - 5) * 0.01);
        CK(hipFree(dst));
    };
    phases(k_sum<0, 0, 0, true>, "table in memory, 128 lanes (product form)", 128, sw128);
    phases(k_sum<1, 0, 0, true>, "table in kernel arguments, 128 lanes", 128, sw128);
    phases(k_sum<4, 0, 0, true>, "table in kernel arguments, one wave per orbit", 64, sw64);
    phases(k_sum<1, 0, 2, true>, "table in kernel arguments, 128 lanes, sc1 stores", 128, sw128);
    phasesx(k_sumx<0, 0, 0, true>, "scalar parameters (preloaded into SGPRs when built with -amdgpu-kernarg-preload-count), table in memory, 128 lanes", 128, sw128);
    {
        // clock rate of s_memtime against the 100-MHz wall clock
        int dev = 0, khz = 0;
        CK(hipGetDevice(&dev));
        CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
        std::printf("device clock rate attribute: %d kHz\n", khz);
    }
    return 0;
}
