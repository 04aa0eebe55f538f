// xpose_proto.hip -- round 5 design experiment for HBM-sized transposing copies (permutedims!(B, A, (4,3,2,1)) at
// 128^4 Float64 runs at 0.66-0.71 of 8 TB/s with the library's 128 x 32 tiles; a contiguous copy of the same bytes reaches 0.79).
//
// B[i0,i1,i2,i3] = A[i3,i2,i1,i0], n^4 Float64, column-major.  A tile is TD0 elements of i0 (B's unit axis) x TD3 elements of i3
// (A's unit axis), for one (i1, i2).  A workgroup of 1024 lanes loads the tile along A's unit axis (16 B per lane, every load issued
// before the first LDS write), parks it in LDS (row pitch TD3 + 2: conflict-free transposed reads), reads it back along B's unit axis
// and stores 16 B per lane (non-temporal).  What varies: the tile shape (how long the contiguous runs are on each side: 128 x 32 =
// 256-byte reads / 1-KiB writes, 128 x 128 = 1 KiB on both sides, ...) and the order in which the grid walks (i3-tile, i0-tile, i1, i2).
// Build: hipcc -O3 --offload-arch=gfx950 tools/xpose_proto.hip -o tools/bin/xpose_proto;  run: tools/bin/xpose_proto [n=128]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// ORDER: which tile coordinate varies fastest with blockIdx.x
//   0: t3 (A-contiguous neighbours), then t0, then i1, then i2      1: t0, t3, i1, i2      2: t3, t0, i2, i1      3: i1 fastest (B-contiguous), then t3, t0, i2
template <int TD0, int TD3, int ORDER, bool NT>
__global__ void __launch_bounds__(1024) k_xpose(const double* __restrict__ A, double* __restrict__ B, int n) {
    constexpr int PITCH = TD3 + 2;
    extern __shared__ double lds[];
    const int nt0 = n / TD0, nt3 = n / TD3;
    unsigned b = blockIdx.x;
    int t0, t3, i1, i2;
    if (ORDER == 0) { t3 = b % nt3; b /= nt3; t0 = b % nt0; b /= nt0; i1 = b % n; i2 = b / n; }
    else if (ORDER == 1) { t0 = b % nt0; b /= nt0; t3 = b % nt3; b /= nt3; i1 = b % n; i2 = b / n; }
    else if (ORDER == 2) { t3 = b % nt3; b /= nt3; t0 = b % nt0; b /= nt0; i2 = b % n; i1 = b / n; }
    else { i1 = b % n; b /= n; t3 = b % nt3; b /= nt3; t0 = b % nt0; i2 = b / nt0; }
    const size_t n1 = (size_t)n, n2 = n1 * n1, n3 = n2 * n1;
    // A element (i3, i2, i1, i0): i3 + n*i2 + n^2*i1 + n^3*i0
    const double* a0 = A + (size_t)t3 * TD3 + n1 * (size_t)i2 + n2 * (size_t)i1 + n3 * (size_t)t0 * TD0;
    double* b0 = B + (size_t)t0 * TD0 + n1 * (size_t)i1 + n2 * (size_t)i2 + n3 * (size_t)t3 * TD3;
    const int tid = threadIdx.x;
    // ---- load: lanes along i3 (TD3/2 lanes per row of TD3), rows = i0
    constexpr int LPR = TD3 / 2, RPP = 1024 / LPR, NPASS = TD0 / RPP;
    const int lc = tid % LPR, lr = tid / LPR;
    d2 x[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) x[p] = *reinterpret_cast<const d2*>(a0 + n3 * (size_t)(p * RPP + lr) + 2 * lc);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        double* L = lds + (size_t)(p * RPP + lr) * PITCH + 2 * lc;  // row = i0, column = i3
        L[0] = x[p].x;
        L[1] = x[p].y;
    }
    __syncthreads();
    // ---- store: lanes along i0 (TD0/2 lanes per row of TD0), rows = i3
    constexpr int LPR2 = TD0 / 2, RPP2 = 1024 / LPR2, NPASS2 = TD3 / RPP2;
    const int sc = tid % LPR2, sr = tid / LPR2;
#pragma unroll
    for (int p = 0; p < NPASS2; ++p) {
        const int i3 = p * RPP2 + sr;
        d2 v;
        v.x = lds[(size_t)(2 * sc) * PITCH + i3];
        v.y = lds[(size_t)(2 * sc + 1) * PITCH + i3];
        d2* dst = reinterpret_cast<d2*>(b0 + n3 * (size_t)i3 + 2 * sc);
        if (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
    }
}

__global__ void k_copy(const d2* __restrict__ a, d2* __restrict__ b, size_t nv) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(a[i], b + i);
}

template <int TD0, int TD3, int ORDER, bool NT>
static void run(const char* name, const double* A, double* B, int n, const std::vector<double>& hA, std::vector<double>& hB, bool check) {
    if (n % TD0 || n % TD3) return;
    const size_t lds = (size_t)TD0 * (TD3 + 2) * sizeof(double);
    auto kern = k_xpose<TD0, TD3, ORDER, NT>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((size_t)(n / TD0) * (n / TD3) * n * n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, A, B, n);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, A, B, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 4 < best) best = ms / 4;
    }
    const double bytes = 16.0 * (double)n * n * n * n;
    long bad = -1;
    if (check) {
        CK(hipMemcpy(hB.data(), B, hB.size() * 8, hipMemcpyDeviceToHost));
        bad = 0;
        const size_t n1 = n, n2 = n1 * n1, n3 = n2 * n1;
        for (size_t s = 0; s < 200000; ++s) {  // sampled check
            const size_t e = (s * 2654435761ull) % hB.size();
            const size_t i0 = e % n1, i1 = (e / n1) % n1, i2 = (e / n2) % n1, i3 = e / n3;
            if (hB[e] != hA[i3 + n1 * i2 + n2 * i1 + n3 * i0]) ++bad;
        }
    }
    std::printf("%-34s tile %3d x %-3d order %d %s  %8.1f us  %6.0f GB/s  %.3f of 8 TB/s  lds %6zu  grid %8u  %s\n", name, TD0, TD3, ORDER, NT ? "nt" : "  ", best * 1e3,
                bytes / best / 1e6, bytes / best / 1e6 / 8000, lds, grid, bad < 0 ? "" : (bad ? "WRONG" : "ok"));
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 128;
    const size_t N = (size_t)n * n * n * n;
    double *A, *B;
    CK(hipMalloc(&A, N * 8));
    CK(hipMalloc(&B, N * 8));
    std::vector<double> hA(N), hB(N);
    for (size_t i = 0; i < N; ++i) hA[i] = (double)(i % 1000003) + 0.5;
    CK(hipMemcpy(A, hA.data(), N * 8, hipMemcpyHostToDevice));
    {   // the floor: a contiguous copy of the same bytes
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const d2*)A, (d2*)B, N / 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const d2*)A, (d2*)B, N / 2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("contiguous copy (nt stores), n = %d: %.1f us  %.0f GB/s\n", n, ms / 4 * 1e3, 16.0 * N / (ms / 4) / 1e6);
    }
    run<128, 32, 0, true>("library shape", A, B, n, hA, hB, true);
    run<128, 32, 1, true>("", A, B, n, hA, hB, false);
    run<128, 32, 3, true>("", A, B, n, hA, hB, false);
    run<32, 128, 0, true>("long reads, short writes", A, B, n, hA, hB, true);
    run<64, 64, 0, true>("", A, B, n, hA, hB, true);
    run<64, 128, 0, true>("", A, B, n, hA, hB, true);
    run<128, 64, 0, true>("", A, B, n, hA, hB, true);
    run<128, 64, 3, true>("", A, B, n, hA, hB, false);
    run<128, 128, 0, true>("1 KiB runs on both sides", A, B, n, hA, hB, true);
    run<128, 128, 2, true>("", A, B, n, hA, hB, false);
    run<128, 128, 3, true>("", A, B, n, hA, hB, false);
    run<128, 128, 0, false>("", A, B, n, hA, hB, false);
    run<64, 128, 3, true>("", A, B, n, hA, hB, false);
    return 0;
}
