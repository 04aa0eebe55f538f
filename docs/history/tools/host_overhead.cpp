// Host-side cost of the one-shot entry point smr_mapreduce (plan-cache hit path) and of
// smr_plan_execute, measured from plain C++ (no Python): launches per second when the GPU work
// is negligible.  Build: g++ -O2 tools/host_overhead.cpp -Iinclude -Lstrided.jl_amd -lstrided_hip
//   -Wl,-rpath,$PWD/strided.jl_amd -L/opt/rocm/lib -lamdhip64 -o /tmp/host_overhead
#include <chrono>
#include <cstdio>
#include <cstring>

#include "strided_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    if (smr_init(0) != SMR_OK) {
        std::printf("init failed: %s\n", smr_last_error());
        return 1;
    }
    const int n = 64;
    void *a = nullptr, *b = nullptr;
    smr_malloc(sizeof(double) * n * n, &a);
    smr_malloc(sizeof(double) * n * n, &b);
    smr_problem p;
    std::memset(&p, 0, sizeof p);
    p.N = 2;
    p.M = 2;
    p.dims[0] = p.dims[1] = n;
    p.ops[0].base = b; p.ops[0].strides[0] = 1; p.ops[0].strides[1] = n; p.ops[0].dtype = SMR_F64;
    p.ops[1].base = a; p.ops[1].strides[0] = n; p.ops[1].strides[1] = 1; p.ops[1].dtype = SMR_F64;  // transpose
    p.redop = SMR_RED_NONE;
    // round 4: the same two entry points on a library-owned stream (eager direct dispatch: the library submits AQL packets itself)
    void* lib_stream = nullptr;
    for (int variant = 0; variant < 4; ++variant) {
        if (variant == 2) {
            if (smr_stream_create(&lib_stream) != SMR_OK) {
                std::printf("smr_stream_create failed: %s\n", smr_last_error());
                return 1;
            }
            p.stream = lib_stream;
        }
        const bool use_plan = variant & 1;
        smr_plan* plan = nullptr;
        if (use_plan && smr_plan_create(&p, &plan) != SMR_OK) return 1;
        for (int i = 0; i < 100; ++i) use_plan ? smr_plan_execute(plan, nullptr, p.stream) : smr_mapreduce(&p);
        smr_stream_sync(p.stream);
        const int iters = 20000;
        const double t0 = now();
        for (int i = 0; i < iters; ++i) {
            const int rc = use_plan ? smr_plan_execute(plan, nullptr, p.stream) : smr_mapreduce(&p);
            if (rc) {
                std::printf("error: %s\n", smr_last_error());
                return 1;
            }
        }
        const double t1 = now();
        smr_stream_sync(p.stream);
        const double t2 = now();
        std::printf("%-18s %-16s host %.2f us/call (enqueue), %.2f us/call incl. drain\n", use_plan ? "smr_plan_execute" : "smr_mapreduce",
                    variant >= 2 ? "library stream" : "HIP null stream", (t1 - t0) / iters * 1e6, (t2 - t0) / iters * 1e6);
        if (plan) smr_plan_destroy(plan);
    }
    std::printf("direct launches %lld (argument-block hits %lld), through HIP instead %lld, argument blocks in device memory: %lld\n",
                (long long)smr_get_option("eager_launches"), (long long)smr_get_option("eager_arg_hits"), (long long)smr_get_option("eager_fallback"),
                (long long)smr_get_option("eager_kernarg_device"));
    return 0;
}
