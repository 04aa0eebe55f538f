// c5_proto.hip -- design experiment for BASELINE configs[4]: B .= A.*exp.(-2A) .+ sin.(A.*A), 8192^2 Float32.
// Which launch shape gets a transcendental-heavy streaming map closest to the copy rate?
//   U: 16-byte vectors per lane in flight, NT: lanes per workgroup, PERSIST: grid-stride over a fixed grid,
//   NTL / NTS: non-temporal loads / stores.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/c5_proto.hip -o tools/bin/c5_proto
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool COPY>
__device__ __forceinline__ float fexpr(float a) {
    if constexpr (COPY) return a;
    return a * expf(-2.0f * a) + sinf(a * a);
}

template <int U, int NT, bool PERSIST, bool NTL, bool NTS, bool COPY>
__global__ void __launch_bounds__(NT) k_map(const float* __restrict__ A, float* __restrict__ B, size_t nvec) {
    const size_t chunk = (size_t)NT * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nvec; base += PERSIST ? (size_t)gridDim.x * chunk : nvec) {
        f4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * NT + threadIdx.x;
            if (i < nvec) x[u] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(A) + i) : reinterpret_cast<const f4*>(A)[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * NT + threadIdx.x;
            f4 y;
            y.x = fexpr<COPY>(x[u].x);
            y.y = fexpr<COPY>(x[u].y);
            y.z = fexpr<COPY>(x[u].z);
            y.w = fexpr<COPY>(x[u].w);
            if (i < nvec) {
                if (NTS) __builtin_nontemporal_store(y, reinterpret_cast<f4*>(B) + i);
                else reinterpret_cast<f4*>(B)[i] = y;
            }
        }
    }
}

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

template <int U, int NT, bool PERSIST, bool NTL, bool NTS, bool COPY>
static void run(hipStream_t st, const float* A, float* B, size_t n, int wgs_per_cu) {
    const size_t nvec = n / 4;
    const size_t chunk = (size_t)NT * U;
    const unsigned grid = PERSIST ? (unsigned)(256 * wgs_per_cu) : (unsigned)((nvec + chunk - 1) / chunk);
    const float us = time_graph(st, 10, [&] { hipLaunchKernelGGL((k_map<U, NT, PERSIST, NTL, NTS, COPY>), dim3(grid), dim3(NT), 0, st, A, B, nvec); });
    printf("%-5s U=%d lanes=%4d %-10s ntl=%d nts=%d grid=%7u : %8.2f us %7.1f GB/s\n", COPY ? "copy" : "expr5", U, NT, PERSIST ? "persistent" : "one-shot", (int)NTL,
           (int)NTS, grid, us, 8.0 * n / us * 1e-3);
}

int main() {
    const size_t n = (size_t)8192 * 8192;
    float *A, *B;
    CK(hipMalloc(&A, n * 4));
    CK(hipMalloc(&B, n * 4));
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (float)(s >> 8) * (1.0f / 16777216.0f);
    }
    CK(hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    run<2, 256, false, false, false, true>(st, A, B, n, 0);
    run<4, 256, false, false, false, true>(st, A, B, n, 0);
    run<2, 256, true, false, false, true>(st, A, B, n, 8);
    run<2, 256, false, false, false, false>(st, A, B, n, 0);
    run<1, 256, false, false, false, false>(st, A, B, n, 0);
    run<4, 256, false, false, false, false>(st, A, B, n, 0);
    run<8, 256, false, false, false, false>(st, A, B, n, 0);
    run<2, 512, false, false, false, false>(st, A, B, n, 0);
    run<2, 1024, false, false, false, false>(st, A, B, n, 0);
    run<1, 1024, false, false, false, false>(st, A, B, n, 0);
    run<2, 128, false, false, false, false>(st, A, B, n, 0);
    run<2, 64, false, false, false, false>(st, A, B, n, 0);
    run<2, 256, false, true, false, false>(st, A, B, n, 0);
    run<2, 256, false, false, true, false>(st, A, B, n, 0);
    run<2, 256, false, true, true, false>(st, A, B, n, 0);
    run<2, 256, true, false, false, false>(st, A, B, n, 4);
    run<2, 256, true, false, false, false>(st, A, B, n, 8);
    run<2, 256, true, false, false, false>(st, A, B, n, 16);
    run<4, 256, true, false, false, false>(st, A, B, n, 8);
    run<1, 256, true, false, false, false>(st, A, B, n, 8);
    run<2, 512, true, false, false, false>(st, A, B, n, 4);
    run<2, 1024, true, false, false, false>(st, A, B, n, 2);
    run<2, 256, true, false, true, false>(st, A, B, n, 8);
    return 0;
}
