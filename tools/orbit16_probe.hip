// orbit16_probe.hip -- round 6, second design experiment for the 4-way permuted sum at 32^4 Float64.
// orbit32_probe.hip showed the pattern ceiling of 4x4x4x4 cubes (32-byte runs): reads cost by the 128-byte lines a workgroup touches,
// writes by the 64-byte pieces an INSTRUCTION's lane quads form.  Here a workgroup owns NCW = 4 / 8 / 16 cubes (any union of
// orbits, closed under the rotation), every cube with its own origin and its own 2-KiB LDS region; lane bit 1 selects one of two
// cubes, so that when the planner makes cubes 2i and 2i+1 neighbours along the unit axis the quad {c0 half, cube} moves one 64-byte
// run.  NCW = 16 with the orbits of {(2a+d, 2b+e, K, L)} = the rotations of an 8x8x4x4 box: half of all runs are 64 bytes, 256
// workgroups of 512 lanes, one per CU.  Tiles on the diagonals (blocks (a,b,a,b)) keep cube orbits, packed 16 cubes per workgroup.
// Every variant is verified bit-exactly.   Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

// the PRODUCT's device body of the PAIR form, compiled into this probe (device parts only): same kernel code, the probe's work list
#define SMR_JIT 1
#define SMR_CT 1
#include "../strided.jl_amd/csrc/smr_k_orbit.hip"
#undef SMR_JIT

typedef unsigned long long u64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u8v __attribute__((ext_vector_type(8)));

static constexpr int N = 32;

__global__ void k_empty() {}

// in-region LDS index of element i of a cube: fold the three high bits into bank bits 1..3 (lanes of a half-wave then hit 16 distinct
// 8-byte banks in the parking writes and in all three transposing reads, within the half {bit 0 ^ bit 4 = const}); regions of odd
// parity are XORed with 17, which moves them to the other half: the two cubes a lane pair reads never collide
template <int CPI>
__device__ __forceinline__ uint32_t swz(uint32_t i) {
    if (CPI == 1) return i ^ (((i >> 1) ^ (i >> 3)) & 31u);  // 128 lanes, one cube per half-wave: the product kernel's fold
    return i ^ (((i >> 5) & 7u) << 1);
}

// entry of 8 words per (workgroup, instruction j, pair p): orgA, orgB (element offsets of cubes CPI*j + 2p + {0,1}), own region words
// (16 bits each: (q << 8) | parity * 17), then for m = 1..3 the region word of the cube an output of cube {0,1} reads view m from
// STO: 0 all stores after all adds, 1 each cube's store right behind its adds
template <int NCW, int STO>
__global__ void __launch_bounds__(32 * NCW) k_sum16(const u8v* __restrict__ rows, const double* __restrict__ A, double* __restrict__ C) {
    constexpr int CPI = NCW / 4, P = CPI >= 2 ? CPI / 2 : 1;
    __shared__ __attribute__((aligned(16))) double lds[NCW * 256];
    const uint32_t tid = threadIdx.x;
    const uint32_t b = CPI >= 2 ? (tid >> 1) & 1u : 0u;
    const uint32_t p = CPI == 4 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 8)) : 0u;
    const uint32_t u = CPI >= 2 ? ((tid & 1u) | (((tid >> 2) & 63u) << 1)) : tid;
    const uint32_t e = u * 2;
    const uint32_t c0 = e & 3, c1 = (e >> 2) & 3, c2 = (e >> 4) & 3, c3 = (e >> 6) & 3;
    const uint32_t goff = c0 + (c1 << 5) + (c2 << 10) + (c3 << 15);
    u8v ent[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ent[j] = rows[((size_t)blockIdx.x * 4 + j) * P + p];
    uint32_t org[4];
    d2 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        org[j] = b ? ent[j][1] : ent[j][0];
        x[j] = *reinterpret_cast<const d2*>(A + org[j] + goff);
    }
    const uint32_t sh = b * 16;
    const uint32_t wi0 = swz<CPI>(e), wi1 = swz<CPI>(e + 1);
    uint32_t ri[4][2];
#pragma unroll
    for (int m = 1; m < 4; ++m) {
        const uint32_t c[4] = {c0, c1, c2, c3};
        const uint32_t lr = c[m] | (c[(m + 1) & 3] << 2) | (c[(m + 2) & 3] << 4) | (c[(m + 3) & 3] << 6);
#pragma unroll
        for (int h = 0; h < 2; ++h) ri[m][h] = swz<CPI>(lr | ((uint32_t)h << (2 * ((4 - m) & 3))));
    }
    uint32_t own[4], src[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        own[j] = (ent[j][2] >> sh) & 0xffffu;
#pragma unroll
        for (int m = 1; m < 4; ++m) src[j][m] = (ent[j][2 + m] >> sh) & 0xffffu;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lds[own[j] ^ wi0] = x[j].x;
        lds[own[j] ^ wi1] = x[j].y;
    }
    __syncthreads();
    if (STO == 0) {
        double v[4][2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 1; m < 4; ++m) v[j][h][m] = lds[src[j][m] ^ ri[m][h]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j].x = ((x[j].x + v[j][0][1]) + v[j][0][2]) + v[j][0][3];
            x[j].y = ((x[j].y + v[j][1][1]) + v[j][1][2]) + v[j][1][3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<d2*>(C + org[j] + goff) = x[j];
    } else if (STO == 2) {  // all LDS reads issued first, then per cube: adds (partial lgkmcnt waits) and its store
        double v[4][2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 1; m < 4; ++m) v[j][h][m] = lds[src[j][m] ^ ri[m][h]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j].x = ((x[j].x + v[j][0][1]) + v[j][0][2]) + v[j][0][3];
            x[j].y = ((x[j].y + v[j][1][1]) + v[j][1][2]) + v[j][1][3];
            *reinterpret_cast<d2*>(C + org[j] + goff) = x[j];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 1; m < 4; ++m) v[h][m] = lds[src[j][m] ^ ri[m][h]];
            x[j].x = ((x[j].x + v[0][1]) + v[0][2]) + v[0][3];
            x[j].y = ((x[j].y + v[1][1]) + v[1][2]) + v[1][3];
            *reinterpret_cast<d2*>(C + org[j] + goff) = x[j];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ void __launch_bounds__(256) k_product_pair(const uint32_t* list, const char* src, char* dst, uint32_t eshp, uint32_t elenp, uint32_t es0, uint32_t es1,
                                                      uint32_t es2, uint32_t es3, uint32_t ntlog, const smr::OrbitArgs a) {
    smr::OrbitHead h;
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
    smr::orbit_pair_body<double, smr::FAdd4<double>>(a, h, smr::FAdd4<double>{});
}

struct Ctx {
    hipStream_t st;
    hipEvent_t e0, e1;
};
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

// ---- host: cubes, orbits, workgroups ----------------------------------------------------------------------------------------------
static uint32_t rot(uint32_t id) { return (id >> 3) | ((id & 7u) << 9); }  // (t0,t1,t2,t3) -> (t1,t2,t3,t0)
static uint32_t cube_org(uint32_t id) { return 4 * ((id & 7) + 32 * ((id >> 3) & 7) + 1024 * ((id >> 6) & 7) + 32768 * ((id >> 9) & 7)); }
static uint32_t mk(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) { return t0 | (t1 << 3) | (t2 << 6) | (t3 << 9); }

struct Wg {
    std::vector<uint32_t> cube;  // NCW cube ids in instruction order (index CPI*j + 2p + b)
};

// pack loose cube orbits (each a list of distinct cubes closed under rot) into workgroups of ncw cubes; free places repeat cube 0
static void pack_orbits(std::vector<std::vector<uint32_t>> orbs, int ncw, std::vector<Wg>& out) {
    std::stable_sort(orbs.begin(), orbs.end(), [](const auto& a, const auto& b) { return a.size() > b.size(); });
    std::vector<char> used(orbs.size(), 0);
    for (size_t i = 0; i < orbs.size(); ++i) {
        if (used[i]) continue;
        Wg w;
        for (size_t k = i; k < orbs.size(); ++k)
            if (!used[k] && w.cube.size() + orbs[k].size() <= (size_t)ncw) {
                used[k] = 1;
                for (uint32_t q : orbs[k]) w.cube.push_back(q);
            }
        while ((int)w.cube.size() < ncw) w.cube.push_back(w.cube[0]);
        out.push_back(w);
    }
}

static std::vector<uint32_t> orbit_of(uint32_t id) {
    std::vector<uint32_t> o;
    uint32_t x = id;
    do {
        o.push_back(x);
        x = rot(x);
    } while (x != id);
    return o;
}

// workgroup lists.  mode 4: one cube orbit per workgroup (degenerate orbits packed), super-cell order as the product lists them.
// mode 8: orbits of (8,4,4,4) boxes (two cube orbits, partners along dim 0 in slot 0); mode 16: orbits of (8,8,4,4) boxes
static std::vector<Wg> make_wgs(int ncw) {
    std::vector<Wg> wgs;
    std::vector<char> seen(4096, 0);
    std::vector<std::vector<uint32_t>> loose;
    for (uint32_t blk = 0; blk < 256; ++blk) {
        const uint32_t B[4] = {blk & 3, (blk >> 2) & 3, (blk >> 4) & 3, (blk >> 6) & 3};
        // block orbit root = smallest rotation
        uint32_t minid = ~0u;
        bool sym = false;
        for (int g = 0; g < 4; ++g) {
            const uint32_t id = B[g] | (B[(g + 1) & 3] << 2) | (B[(g + 2) & 3] << 4) | (B[(g + 3) & 3] << 6);
            minid = std::min(minid, id);
            if (g > 0 && id == blk) sym = true;
        }
        if (sym || ncw == 4) {
            // cube orbits of this block, each once
            for (uint32_t q = 0; q < 16; ++q) {
                const uint32_t id = mk(2 * B[0] + (q & 1), 2 * B[1] + ((q >> 1) & 1), 2 * B[2] + ((q >> 2) & 1), 2 * B[3] + ((q >> 3) & 1));
                std::vector<uint32_t> o = orbit_of(id);
                const uint32_t root = *std::min_element(o.begin(), o.end());
                if (seen[root]) continue;
                seen[root] = 1;
                if (ncw == 4 && o.size() == 4) {
                    Wg w;
                    w.cube = o;
                    wgs.push_back(w);
                } else {
                    loose.push_back(o);
                }
            }
            continue;
        }
        if (minid != blk) continue;  // the block orbit is emitted from its root block
        if (ncw == 8) {
            for (uint32_t q = 0; q < 8; ++q) {  // (e, r, s)
                Wg w;
                w.cube.resize(8);
                for (uint32_t d = 0; d < 2; ++d) {
                    uint32_t id = mk(2 * B[0] + d, 2 * B[1] + (q & 1), 2 * B[2] + ((q >> 1) & 1), 2 * B[3] + ((q >> 2) & 1));
                    for (int j = 0; j < 4; ++j) {
                        w.cube[2 * j + d] = id;
                        seen[id] = 1;
                        id = rot(id);
                    }
                }
                wgs.push_back(w);
            }
        } else {
            for (uint32_t q = 0; q < 4; ++q) {  // (r, s)
                Wg w;
                w.cube.resize(16);
                for (uint32_t d = 0; d < 2; ++d)
                    for (uint32_t ee = 0; ee < 2; ++ee) {
                        uint32_t id = mk(2 * B[0] + d, 2 * B[1] + ee, 2 * B[2] + (q & 1), 2 * B[3] + ((q >> 1) & 1));
                        for (int j = 0; j < 4; ++j) {
                            // slot 1 = (2b+e, K, L, 2a+d): its unit-axis partner flips e
                            const uint32_t bb = j == 1 ? ee : d, pp = j == 1 ? d : ee;
                            w.cube[4 * j + 2 * pp + bb] = id;
                            seen[id] = 1;
                            id = rot(id);
                        }
                    }
                wgs.push_back(w);
            }
        }
    }
    pack_orbits(loose, ncw, wgs);
    return wgs;
}

int main(int argc, char** argv) {
    (void)argc;
    (void)argv;
    Ctx c;
    CK(hipStreamCreate(&c.st));
    CK(hipEventCreate(&c.e0));
    CK(hipEventCreate(&c.e1));
    const size_t NE = (size_t)N * N * N * N;
    std::vector<double> hA(NE), hC(NE), want(NE);
    for (size_t i = 0; i < NE; ++i) hA[i] = (double)((i * 2654435761u) % 1000003u) + 0.5 * (double)(i % 7);
    for (int l = 0; l < N; ++l)
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) {
                    auto at = [&](int a, int b, int cc, int d) { return hA[a + N * (b + N * (cc + (size_t)N * d))]; };
                    want[i + N * (j + N * (k + (size_t)N * l))] = ((at(i, j, k, l) + at(j, k, l, i)) + at(k, l, i, j)) + at(l, i, j, k);
                }
    double *dA, *dC;
    CK(hipMalloc(&dA, NE * 8));
    CK(hipMalloc(&dC, NE * 8));
    CK(hipMemcpy(dA, hA.data(), NE * 8, hipMemcpyHostToDevice));
    const double bytes = 2.0 * NE * 8;
    auto report = [&](const char* name, double us) {
        std::printf("%-96s %7.3f us  %7.0f GB/s  frac %.3f\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
        std::fflush(stdout);
    };
    report("empty kernel 256 x 512", time_graph(c, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); }));

    constexpr int NP = 45;
    std::vector<double*> cold_A(NP), cold_C(NP);
    for (int i = 0; i < NP; ++i) {
        CK(hipMalloc(&cold_A[i], NE * 8));
        CK(hipMalloc(&cold_C[i], NE * 8));
        CK(hipMemcpy(cold_A[i], dA, NE * 8, hipMemcpyDeviceToDevice));
    }
    auto run = [&](int ncw, int sto, int deal) {
        std::vector<Wg> wgs = make_wgs(ncw);
        // every cube exactly once (repeats only of a workgroup's own cube 0)?
        {
            std::vector<int> cnt(4096, 0);
            for (const Wg& w : wgs) {
                std::vector<uint32_t> d = w.cube;
                std::sort(d.begin(), d.end());
                d.erase(std::unique(d.begin(), d.end()), d.end());
                for (uint32_t q : d) ++cnt[q];
            }
            for (int q = 0; q < 4096; ++q)
                if (cnt[q] != 1) {
                    std::printf("cover wrong: cube %d x %d\n", q, cnt[q]);
                    std::exit(1);
                }
        }
        // deal: 0 list order (workgroup w on XCD w % 8), 1 one contiguous run of the list per XCD
        const size_t nw0 = wgs.size(), cs = (nw0 + 7) / 8, nwg = deal ? cs * 8 : nw0;
        std::vector<const Wg*> at(nwg, nullptr);
        for (size_t i = 0; i < nw0; ++i) {
            const size_t pos = deal ? (i % cs) * 8 + i / cs : i;
            at[pos] = &wgs[i];
        }
        const int CPI = ncw / 4, P = CPI >= 2 ? CPI / 2 : 1;
        std::vector<uint32_t> rows(nwg * 4 * P * 8, 0u);
        long runs64 = 0;
        for (size_t w = 0; w < nwg; ++w) {
            const Wg& g = at[w] ? *at[w] : wgs[0];  // an idle place repeats workgroup 0 (same values stored twice)
            std::map<uint32_t, int> index;
            for (int q = 0; q < ncw; ++q)
                if (!index.count(g.cube[q])) index[g.cube[q]] = q;
            auto parity = [&](int q) { return CPI >= 2 ? ((q ^ (q >> 1)) & 1) : 0; };
            auto word = [&](int q) { return (uint32_t)((q << 8) | (parity(q) ? 17 : 0)); };
            for (int j = 0; j < 4; ++j)
                for (int p = 0; p < P; ++p) {
                    uint32_t* en = &rows[((w * 4 + j) * P + p) * 8];
                    const int nb = CPI >= 2 ? 2 : 1;
                    for (int b = 0; b < nb; ++b) {
                        const int q = CPI * j + 2 * p + b;
                        const uint32_t id = g.cube[q];
                        en[b] = cube_org(id);
                        en[2] |= word(q) << (16 * b);
                        uint32_t s = id;
                        for (int m = 1; m < 4; ++m) {
                            s = rot(s);
                            en[2 + m] |= word(index.at(s)) << (16 * b);
                        }
                    }
                    if (nb == 2 && en[1] == en[0] + 4 && (en[0] % 8) == 0) ++runs64;
                }
        }
        uint32_t* d_rows;
        CK(hipMalloc(&d_rows, rows.size() * 4));
        CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        std::function<void(hipStream_t)> launch = [&](hipStream_t s) {
            const u8v* r = reinterpret_cast<const u8v*>(d_rows);
#define L(NCW, STO) hipLaunchKernelGGL((k_sum16<NCW, STO>), dim3(nwg), dim3(32 * NCW), 0, s, r, (const double*)dA, dC)
            if (ncw == 4) { if (sto == 2) L(4, 2); else if (sto) L(4, 1); else L(4, 0); }
            else if (ncw == 8) { if (sto == 2) L(8, 2); else if (sto) L(8, 1); else L(8, 0); }
            else { if (sto == 2) L(16, 2); else if (sto) L(16, 1); else L(16, 0); }
#undef L
        };
        // sto == 3 (8 cubes only): the product's orbit_pair_body on THIS work list (entries: orgA, orgB, own words, views 1..3)
        uint32_t* d_prows = nullptr;
        smr::OrbitArgs pa;
        std::memset(&pa, 0, sizeof pa);
        if (sto == 3) {
            std::vector<uint32_t> pr(nwg * 4 * 8, 0u);
            for (size_t w = 0; w < nwg; ++w)
                for (int j = 0; j < 4; ++j) {
                    const uint32_t* en = &rows[(w * 4 + j) * 8];
                    uint32_t* o = &pr[(w * 4 + j) * 8];
                    for (int q = 0; q < 6; ++q) o[q] = en[q];
                }
            CK(hipMalloc(&d_prows, pr.size() * 4));
            CK(hipMemcpy(d_prows, pr.data(), pr.size() * 4, hipMemcpyHostToDevice));
            pa.nin = 4;
            pa.tilelog = 8;
            pa.ntlog = 8;
            for (int j = 0; j < 4; ++j) {
                pa.esh[j] = 2 * j;
                pa.elen[j] = 2;
            }
            pa.estride[0] = 8; pa.estride[1] = 8 * 32; pa.estride[2] = 8 * 1024; pa.estride[3] = 8 * 32768;
            for (int k = 1; k < 4; ++k)
                for (int j = 0; j < 4; ++j) pa.lsh[k][j] = 2 * ((j - k + 4) & 3);
        }
        auto launch0 = launch;
        auto launchp = [&](hipStream_t s) {
            hipLaunchKernelGGL(k_product_pair, dim3(nwg), dim3(256), 16384, s, (const uint32_t*)d_prows, (const char*)dA, (char*)dC, 0x06040200u, 0x02020202u, 8u, 256u, 8192u,
                               262144u, 8u, pa);
        };
        if (sto == 3) launch = launchp;
        CK(hipMemset(dC, 0, NE * 8));
        launch(c.st);
        CK(hipStreamSynchronize(c.st));
        CK(hipMemcpy(hC.data(), dC, NE * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < NE; ++i) bad += hC[i] != want[i];
        const double us = time_graph(c, launch);
        // HBM-cold: every launch of the graph works on another pair of arrays (NP pairs = 45 x 16 MiB > the 256-MiB memory-side cache)
        int turn = 0;
        const double us_cold = time_graph(c, [&](hipStream_t s) {
            const u8v* r = reinterpret_cast<const u8v*>(d_rows);
            const double* a = cold_A[turn % NP];
            double* cc = cold_C[turn % NP];
            ++turn;
            if (sto == 3) {
                hipLaunchKernelGGL(k_product_pair, dim3(nwg), dim3(256), 16384, s, (const uint32_t*)d_prows, (const char*)a, (char*)cc, 0x06040200u, 0x02020202u, 8u, 256u,
                                   8192u, 262144u, 8u, pa);
                return;
            }
#define L(NCW, STO) hipLaunchKernelGGL((k_sum16<NCW, STO>), dim3(nwg), dim3(32 * NCW), 0, s, r, a, cc)
            if (ncw == 4) { if (sto == 2) L(4, 2); else if (sto) L(4, 1); else L(4, 0); }
            else if (ncw == 8) { if (sto == 2) L(8, 2); else if (sto) L(8, 1); else L(8, 0); }
            else { if (sto == 2) L(16, 2); else if (sto) L(16, 1); else L(16, 0); }
#undef L
        }, 180, 5);
        char buf[200];
        std::snprintf(buf, sizeof buf, "%2d cubes per workgroup (%4zu wgs x %3d lanes), %s, %s; cube pairs forming 64-B runs %ld of %zu [%s]", ncw, nwg,
                      32 * ncw, sto == 3 ? "THE PRODUCT'S orbit_pair_body on this list" : sto == 2 ? "reads first, store behind each cube's adds" : sto ? "per cube: reads, adds, store" : "stores last", deal ? "XCD-contiguous runs" : "list order", runs64,
                      nwg * 4 * P, bad ? "WRONG" : "ok");
        report(buf, us);
        std::printf("      the same, HBM-cold (rotating over %d pairs of arrays): %7.3f us  frac %.3f\n", NP, us_cold, bytes / us_cold * 1e-3 / 8000.0);
        CK(hipFree(d_rows));
    };
    for (int deal = 1; deal < 2; ++deal)
        for (int sto = 0; sto < 3; ++sto) {
            run(4, sto, deal);
            run(8, sto, deal);
            run(16, sto, deal);
        }
    run(8, 3, 1);
    return 0;
}
