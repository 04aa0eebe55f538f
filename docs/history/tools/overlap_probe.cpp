// overlap_probe -- do the two INDEPENDENT launches of the bench step (permutedims!(B, A, (4,3,2,1)) and C .= sum of 4 permuted
// views of A; both only read A) overlap on the device, and what does it take?  
//
// Drives the library's own kernels from plain C++ through the C ABI, in every launch form, and reports per form
//   * HIP-event and wall-clock time per step,
//   * with the stamp build of the library (make -C strided.jl_amd/csrc stamp): per-wave s_memrealtime stamps -> for every launch
//     first wave start / last wave end; device cadence per step; how far the sum's first wave starts BEFORE the permutedims!'s
//     last wave ends (overlap) -- the profiler-independent witness,
//   * a bit-exact check of both outputs after the timed region.
// Forms: in order on one stream | overlap window (dispatch without the AQL barrier bit when independent) | two streams |
//        the same three captured into a hipGraph.
//
// Build:  hipcc -O2 -std=c++17 tools/overlap_probe.cpp -Iinclude -ldl -o tools/bin/overlap_probe
// Run:    tools/bin/overlap_probe strided.jl_amd/libstrided_hip_stamp.so [n=32] [steps=200]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "strided_hip.h"

#define HC(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(2);                                                                      \
        }                                                                                      \
    } while (0)

static void* L;
template <class F> static F sym(const char* n) {
    void* p = dlsym(L, n);
    if (!p) {
        std::printf("missing symbol %s\n", n);
        std::exit(2);
    }
    return (F)p;
}
static decltype(&smr_init) p_init;
static decltype(&smr_plan_create) p_plan_create;
static decltype(&smr_plan_prepare) p_plan_prepare;
static decltype(&smr_plan_execute) p_plan_execute;
static decltype(&smr_plan_describe) p_plan_describe;
static decltype(&smr_set_option) p_set_option;
static decltype(&smr_get_option) p_get_option;
static decltype(&smr_last_error) p_last_error;
static decltype(&smr_overlap_begin) p_overlap_begin;
static decltype(&smr_overlap_end) p_overlap_end;
static decltype(&smr_seq_create) p_seq_create;
static decltype(&smr_seq_add) p_seq_add;
static decltype(&smr_seq_run) p_seq_run;
static decltype(&smr_seq_wait) p_seq_wait;
static decltype(&smr_seq_info) p_seq_info;
static decltype(&smr_seq_set) p_seq_set;
static decltype(&smr_seq_destroy) p_seq_destroy;

#define SC(x)                                                                                \
    do {                                                                                     \
        int rc_ = (x);                                                                       \
        if (rc_) {                                                                           \
            std::printf("smr error %d (%s) at %s:%d\n", rc_, p_last_error(), __FILE__, __LINE__); \
            std::exit(2);                                                                    \
        }                                                                                    \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Region {
    long a, b;
    int kind;  // 0 = permutedims!, 1 = sum
};

static int n = 32, R = 200;
static smr_plan *plan2, *plan3;
static bool stamped;
static unsigned long long* d_stamps;
static long stamp_words = 48l << 20;
static std::vector<Region> regions;
static double tick_us = 0.01;

static void exec(smr_plan* p, int kind, hipStream_t s) {
    const long before = stamped ? (long)p_get_option("stamp_used") : 0;
    SC(p_plan_execute(p, nullptr, (void*)s));
    if (stamped) regions.push_back({before, (long)p_get_option("stamp_used"), kind});
}

static double med(std::vector<double> v) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static std::vector<double> hA, hB, hC;
static double *dA, *dB, *dC;

static bool verify(const char* what) {
    const size_t N = (size_t)n * n * n * n;
    std::vector<double> gB(N), gC(N);
    HC(hipMemcpy(gB.data(), dB, N * 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(gC.data(), dC, N * 8, hipMemcpyDeviceToHost));
    const bool ok = std::memcmp(gB.data(), hB.data(), N * 8) == 0 && std::memcmp(gC.data(), hC.data(), N * 8) == 0;
    if (!ok) std::printf("    !! %s: outputs differ from the host truth\n", what);
    return ok;
}

// stamps of the recorded regions -> spans, cadence, overlap
static void analyse(const char* name, int skip_steps) {
    if (!stamped || regions.empty()) return;
    const long words = regions.back().b;
    std::vector<unsigned long long> h((size_t)words);
    HC(hipMemcpy(h.data(), d_stamps, (size_t)words * 8, hipMemcpyDeviceToHost));
    struct LS {
        double first, last;
        int kind;
    };
    std::vector<LS> ls;
    for (const Region& r : regions) {
        unsigned long long f = ~0ull, l = 0;
        for (long i = r.a; i + 1 < r.b; i += 2) {
            if (!h[i]) continue;
            f = std::min(f, h[i]);
            l = std::max(l, h[i + 1]);
        }
        if (l == 0) {
            std::printf("    (%s: a region without stamps)\n", name);
            return;
        }
        ls.push_back({(double)f * tick_us, (double)l * tick_us, r.kind});
    }
    std::vector<double> span[2], cad, ov_same, gap_ps, gap_sp;
    for (size_t i = 2 * (size_t)skip_steps; i + 3 < ls.size(); i += 2) {
        const LS &p = ls[i], &s = ls[i + 1], &pn = ls[i + 2];
        span[0].push_back(p.last - p.first);
        span[1].push_back(s.last - s.first);
        cad.push_back(pn.first - p.first);
        ov_same.push_back(std::max(0.0, std::min(p.last, s.last) - std::max(p.first, s.first)));
        gap_ps.push_back(s.first - p.last);   // < 0: the sum's first wave started before the permutedims!'s last wave ended
        gap_sp.push_back(pn.first - s.last);  // next step's permutedims! against this step's sum
    }
    std::printf("    stamps: span permutedims! %.2f us, sum %.2f us | sum starts %+.2f us after permutedims! ends | next permutedims! starts %+.2f us "
                "after sum ends | overlapped %.2f us | device cadence %.3f us/step\n",
                med(span[0]), med(span[1]), med(gap_ps), med(gap_sp), med(ov_same), med(cad));
}

template <class Fn> static void run_eager(const char* name, Fn step_all, hipStream_t s, hipStream_t s_other = nullptr) {
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    HC(hipMemset(dB, 0, hB.size() * 8));
    HC(hipMemset(dC, 0, hC.size() * 8));
    double best_ev = 1e30, best_wall = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        if (stamped) SC(p_set_option("stamp_used", 0));
        regions.clear();
        HC(hipDeviceSynchronize());
        const double t0 = now();
        HC(hipEventRecord(e0, s));
        step_all(R);
        HC(hipEventRecord(e1, s));
        const double t1 = now();
        HC(hipDeviceSynchronize());
        const double t2 = now();
        float ms = 0;
        HC(hipEventElapsedTime(&ms, e0, e1));
        best_ev = std::min(best_ev, (double)ms * 1e3 / R);
        best_wall = std::min(best_wall, (t2 - t0) * 1e6 / R);
        if (rep == 3) std::printf("%-34s events %.3f us/step | wall %.3f us/step (host enqueue %.3f)\n", name, best_ev, best_wall, (t1 - t0) * 1e6 / R);
    }
    (void)s_other;
    analyse(name, 20);
    verify(name);
    HC(hipEventDestroy(e0));
    HC(hipEventDestroy(e1));
}

template <class Fn> static void run_graph(const char* name, Fn capture_body, hipStream_t s) {
    HC(hipMemset(dB, 0, hB.size() * 8));
    HC(hipMemset(dC, 0, hC.size() * 8));
    if (stamped) SC(p_set_option("stamp_used", 0));
    regions.clear();
    hipGraph_t g;
    hipGraphExec_t ge;
    HC(hipDeviceSynchronize());
    HC(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    capture_body(R);
    HC(hipStreamEndCapture(s, &g));
    size_t nn = 0;
    HC(hipGraphGetNodes(g, nullptr, &nn));
    HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    HC(hipGraphLaunch(ge, s));
    HC(hipStreamSynchronize(s));
    double best_ev = 1e30, best_wall = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        const double t0 = now();
        HC(hipEventRecord(e0, s));
        HC(hipGraphLaunch(ge, s));
        HC(hipGraphLaunch(ge, s));
        HC(hipEventRecord(e1, s));
        HC(hipStreamSynchronize(s));
        const double t2 = now();
        float ms = 0;
        HC(hipEventElapsedTime(&ms, e0, e1));
        best_ev = std::min(best_ev, (double)ms * 1e3 / (2 * R));
        best_wall = std::min(best_wall, (t2 - t0) * 1e6 / (2 * R));
    }
    std::printf("%-34s events %.3f us/step | wall %.3f us/step | %zu graph nodes\n", name, best_ev, best_wall, nn);
    analyse(name, 20);
    verify(name);
    HC(hipGraphExecDestroy(ge));
    HC(hipGraphDestroy(g));
    HC(hipEventDestroy(e0));
    HC(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::printf("usage: overlap_probe <path to libstrided_hip[_stamp].so> [n] [steps]\n");
        return 2;
    }
    if (argc > 2) n = std::atoi(argv[2]);
    if (argc > 3) R = std::atoi(argv[3]);
    L = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!L) {
        std::printf("dlopen: %s\n", dlerror());
        return 2;
    }
    p_init = sym<decltype(p_init)>("smr_init");
    p_plan_create = sym<decltype(p_plan_create)>("smr_plan_create");
    p_plan_prepare = sym<decltype(p_plan_prepare)>("smr_plan_prepare");
    p_plan_execute = sym<decltype(p_plan_execute)>("smr_plan_execute");
    p_plan_describe = sym<decltype(p_plan_describe)>("smr_plan_describe");
    p_set_option = sym<decltype(p_set_option)>("smr_set_option");
    p_get_option = sym<decltype(p_get_option)>("smr_get_option");
    p_last_error = sym<decltype(p_last_error)>("smr_last_error");
    p_overlap_begin = sym<decltype(p_overlap_begin)>("smr_overlap_begin");
    p_overlap_end = sym<decltype(p_overlap_end)>("smr_overlap_end");
    p_seq_create = sym<decltype(p_seq_create)>("smr_seq_create");
    p_seq_add = sym<decltype(p_seq_add)>("smr_seq_add");
    p_seq_run = sym<decltype(p_seq_run)>("smr_seq_run");
    p_seq_wait = sym<decltype(p_seq_wait)>("smr_seq_wait");
    p_seq_info = sym<decltype(p_seq_info)>("smr_seq_info");
    p_seq_set = sym<decltype(p_seq_set)>("smr_seq_set");
    p_seq_destroy = sym<decltype(p_seq_destroy)>("smr_seq_destroy");
    SC(p_init(0));
    stamped = p_get_option("stamp_build") == 1;
    int khz = 0;
    HC(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    if (khz > 0) tick_us = 1e3 / khz;
    hipDeviceProp_t prop;
    HC(hipGetDeviceProperties(&prop, 0));
    std::printf("%s | %s build | n = %d (%.1f MiB per array) | %d steps per measurement | wall clock %d kHz\n", prop.name, stamped ? "STAMP" : "product", n,
                (double)n * n * n * n * 8 / 1048576.0, R, khz);
    if (const char* q = std::getenv("GPU_MAX_HW_QUEUES")) std::printf("GPU_MAX_HW_QUEUES=%s\n", q);

    const size_t N = (size_t)n * n * n * n;
    hA.resize(N);
    hB.resize(N);
    hC.resize(N);
    unsigned long long x = 1234;
    for (size_t i = 0; i < N; ++i) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        hA[i] = (double)(long long)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    const long st[4] = {1, n, (long)n * n, (long)n * n * n};
    const int perms[4][4] = {{0, 1, 2, 3}, {1, 2, 3, 0}, {2, 3, 0, 1}, {3, 0, 1, 2}};
    for (int i3 = 0; i3 < n; ++i3)
        for (int i2 = 0; i2 < n; ++i2)
            for (int i1 = 0; i1 < n; ++i1)
                for (int i0 = 0; i0 < n; ++i0) {
                    const int ix[4] = {i0, i1, i2, i3};
                    const size_t o = i0 * st[0] + i1 * st[1] + i2 * st[2] + i3 * st[3];
                    hB[o] = hA[i3 * st[0] + i2 * st[1] + i1 * st[2] + i0 * st[3]];
                    double acc = 0;
                    for (int k = 0; k < 4; ++k) {
                        size_t a = 0;
                        for (int d = 0; d < 4; ++d) a += (size_t)ix[d] * st[perms[k][d]];
                        acc = k == 0 ? hA[a] : acc + hA[a];
                    }
                    hC[o] = acc;
                }
    HC(hipMalloc(&dA, N * 8));
    HC(hipMalloc(&dB, N * 8));
    HC(hipMalloc(&dC, N * 8));
    HC(hipMemcpy(dA, hA.data(), N * 8, hipMemcpyHostToDevice));
    if (stamped) {
        HC(hipMalloc(&d_stamps, (size_t)stamp_words * 8));
        HC(hipMemset(d_stamps, 0, (size_t)stamp_words * 8));
        SC(p_set_option("stamp_base", (int64_t)(uintptr_t)d_stamps));
        SC(p_set_option("stamp_cap", stamp_words));
    }

    hipStream_t s1, s2;
    HC(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    HC(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

    smr_problem p;
    std::memset(&p, 0, sizeof p);
    p.N = 4;
    p.M = 2;
    for (int d = 0; d < 4; ++d) p.dims[d] = n;
    p.ops[0].base = dB;
    p.ops[0].dtype = SMR_F64;
    p.ops[1].base = dA;
    p.ops[1].dtype = SMR_F64;
    for (int d = 0; d < 4; ++d) {
        p.ops[0].strides[d] = st[d];
        p.ops[1].strides[d] = st[3 - d];
    }
    p.stream = s1;
    SC(p_plan_create(&p, &plan2));
    smr_problem q = p;
    q.M = 5;
    q.ops[0].base = dC;
    for (int k = 0; k < 4; ++k) {
        q.ops[1 + k].base = dA;
        q.ops[1 + k].dtype = SMR_F64;
        q.ops[1 + k].offset = 0;
        q.ops[1 + k].conj = 0;
        for (int d = 0; d < 4; ++d) q.ops[1 + k].strides[d] = st[perms[k][d]];
    }
    static const uint8_t prog[] = {SMR_OP_ARG, 1, SMR_OP_ARG, 2, SMR_OP_ADD, 0, SMR_OP_ARG, 3, SMR_OP_ADD, 0, SMR_OP_ARG, 4, SMR_OP_ADD, 0};
    q.fprog = prog;
    q.fprog_len = 7;
    SC(p_plan_create(&q, &plan3));
    SC(p_plan_prepare(plan2));
    SC(p_plan_prepare(plan3));
    char buf[512];
    p_plan_describe(plan2, buf, sizeof buf);
    std::printf("permutedims!: %s\n", buf);
    p_plan_describe(plan3, buf, sizeof buf);
    std::printf("4-way sum   : %s\n", buf);
    for (int i = 0; i < 50; ++i) {
        exec(plan2, 0, s1);
        exec(plan3, 1, s1);
    }
    HC(hipDeviceSynchronize());

    auto inorder = [&](int reps) {
        for (int i = 0; i < reps; ++i) {
            exec(plan2, 0, s1);
            exec(plan3, 1, s1);
        }
    };
    auto window = [&](int reps) {
        SC(p_overlap_begin(s1));
        for (int i = 0; i < reps; ++i) {
            exec(plan2, 0, s1);
            exec(plan3, 1, s1);
        }
        SC(p_overlap_end(s1));
    };
    auto two_streams = [&](int reps) {
        for (int i = 0; i < reps; ++i) {
            exec(plan2, 0, s1);
            exec(plan3, 1, s2);
        }
    };
    hipEvent_t fork, join;
    HC(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto two_chains_captured = [&](int reps) {
        HC(hipEventRecord(fork, s1));
        HC(hipStreamWaitEvent(s2, fork, 0));
        for (int i = 0; i < reps; ++i) {
            exec(plan2, 0, s1);
            exec(plan3, 1, s2);
        }
        HC(hipEventRecord(join, s2));
        HC(hipStreamWaitEvent(s1, join, 0));
    };

    // two host threads, one per stream (is the single-threaded two-stream form bound by the host's enqueue rate?)
    auto two_threads = [&](int reps) {
        std::thread t([&] {
            for (int i = 0; i < reps; ++i) SC(p_plan_execute(plan3, nullptr, (void*)s2));
        });
        for (int i = 0; i < reps; ++i) SC(p_plan_execute(plan2, nullptr, (void*)s1));
        t.join();
    };
    const long any0 = p_get_option("overlap_any"), ord0 = p_get_option("overlap_ordered");
    if (!stamped) run_eager("eager, two streams, two threads", two_threads, s1, s2);
    run_eager("eager, in order, one stream", inorder, s1);
    run_eager("eager, overlap window", window, s1);
    std::printf("    window decisions so far: %ld without the barrier bit, %ld ordered, %ld fences\n", (long)p_get_option("overlap_any") - any0,
                (long)p_get_option("overlap_ordered") - ord0, (long)p_get_option("overlap_fences"));
    run_eager("eager, two streams", two_streams, s1, s2);
    // the library's own replay: pre-built AQL packets on its own HSA queue
    long words2 = 0, words3 = 0;
    if (stamped && regions.size() >= 2) {
        words2 = regions[0].b - regions[0].a;
        words3 = regions[1].b - regions[1].a;
    }
    auto run_seq = [&](const char* name, int order, int fence, int steps_in_seq, int reps, int queues = 8, int slices = 1) {
        HC(hipMemset(dB, 0, hB.size() * 8));
        HC(hipMemset(dC, 0, hC.size() * 8));
        smr_seq* q = nullptr;
        SC(p_seq_create(&q));
        for (int i = 0; i < steps_in_seq; ++i) {
            SC(p_seq_add(q, plan2, nullptr));
            SC(p_seq_add(q, plan3, nullptr));
        }
        if (stamped) SC(p_set_option("stamp_used", 0));
        regions.clear();
        char info[512];
        if (order == 0) SC(p_seq_set(q, "order", 0));
        SC(p_seq_set(q, "fence_scope", fence));
        SC(p_seq_set(q, "queues", queues));
        SC(p_seq_set(q, "slices", slices));
        SC(p_seq_info(q, info, sizeof info));  // builds
        if (stamped) {
            long u = 0;
            for (int i = 0; i < steps_in_seq; ++i) {
                regions.push_back({u, u + words2, 0});
                u += words2;
                regions.push_back({u, u + words3, 1});
                u += words3;
            }
            if (u != (long)p_get_option("stamp_used")) std::printf("    (stamp regions do not add up: %ld vs %ld)\n", u, (long)p_get_option("stamp_used"));
        }
        hipEvent_t e0, e1;
        HC(hipEventCreate(&e0));
        HC(hipEventCreate(&e1));
        SC(p_seq_run(q, reps, s1));
        SC(p_seq_wait(q));
        HC(hipStreamSynchronize(s1));
        double best_ev = 1e30, best_wall = 1e30, best_host = 1e30;
        for (int rep = 0; rep < 6; ++rep) {
            HC(hipDeviceSynchronize());
            const double t0 = now();
            HC(hipEventRecord(e0, s1));
            SC(p_seq_run(q, reps, s1));
            const double t1 = now();
            HC(hipEventRecord(e1, s1));
            SC(p_seq_wait(q));
            const double t2 = now();
            HC(hipStreamSynchronize(s1));
            float ms = 0;
            HC(hipEventElapsedTime(&ms, e0, e1));
            const double steps = (double)steps_in_seq * reps;
            best_ev = std::min(best_ev, (double)ms * 1e3 / steps);
            best_wall = std::min(best_wall, (t2 - t0) * 1e6 / steps);
            best_host = std::min(best_host, (t1 - t0) * 1e6 / steps);
        }
        std::printf("%-34s events %.3f us/step | wall(run+wait) %.3f us/step | host submit %.3f us/step | %d steps x %d reps | %s\n", name, best_ev, best_wall,
                    best_host, steps_in_seq, reps, info);
        analyse(name, std::min(10, steps_in_seq / 4));
        verify(name);
        SC(p_seq_destroy(q));
        HC(hipEventDestroy(e0));
        HC(hipEventDestroy(e1));
    };
    const int RR = R / 50 > 0 ? R / 50 : 1;
    run_seq("seq AQL 1 queue, dependency-aware", 1, 1, 50, RR, 1);
    run_seq("seq AQL 1 queue, all ordered", 0, 1, 50, RR, 1);
    run_seq("seq AQL 1 queue, dep-aware, fence none", 1, 0, 50, RR, 1);
    run_seq("seq AQL 1 queue, fence system", 1, 2, 50, RR, 1);
    run_seq("seq AQL, queue per component", 1, 1, 50, RR);
    run_seq("seq AQL, queue per comp, 1 step x R", 1, 1, 1, R);
    run_seq("seq AQL, queue per comp, all ordered", 0, 1, 50, RR);
    run_seq("seq AQL, queue per comp, 1 step x 20", 1, 1, 1, 20);
    run_seq("seq AQL, 2 slices per component", 1, 1, 1, R, 8, 2);
    run_seq("seq AQL, 3 slices per component", 1, 1, 1, R, 8, 3);
    run_seq("seq AQL, 4 slices per component", 1, 1, 1, R, 8, 4);
    run_seq("seq AQL, 2 slices, 1 step x 20", 1, 1, 1, 20, 8, 2);
    run_seq("seq AQL, 4 slices, 1 step x 20", 1, 1, 1, 20, 8, 4);
    run_seq("seq AQL 1 queue, 1 step x 20", 1, 1, 1, 20, 1);
    run_graph("graph, in order", inorder, s1);
    run_graph("graph, overlap window", window, s1);
    run_graph("graph, two chains (fork/join once)", two_chains_captured, s1);
    return 0;
}
