// run_length_ceiling.hip -- round 5: what does the memory system deliver for the ACCESS PATTERN of the orbit kernel, with everything
// else removed?  (the 4-way permuted sum at 128^4 Float64 sits at 0.40 of 8 TB/s.)
//
// B = A for an n^4 Float64 array, no permutation, no LDS: a workgroup of 1024 lanes copies one box of r0 x r1 x r2 x r3 = 4096
// elements (32 KiB, what one cube of an orbit is): every lane moves 16 bytes at a time, the box's rows are r0 * 8 bytes long and lie
// n * 8, n^2 * 8, n^3 * 8 bytes apart -- exactly the addresses one 8^4 cube of the orbit kernel touches when r0 = 8.  Boxes are dealt
// to workgroups in plain order (box index fastest along dim 0).  What varies: the row length (64 B ... 1 KiB) and n (128: strides of
// 1 KiB / 128 KiB / 16 MiB; 136 and 144: the same rows at odd strides).  The orbit kernel cannot have rows longer than 64 B in all
// four of its cubes (a set closed under the cyclic shift of the coordinates with 4 x 4096 elements has average row length <= 8), so
// the 64-B line of this table is its ceiling as far as DRAM is concerned.
// Build: hipcc -O3 --offload-arch=gfx950 tools/run_length_ceiling.hip -o tools/bin/run_length_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// MODE 0: read + write (copy); 1: read only (sum into a sink); 2: write only
// ORDER: how box indices are dealt to workgroups.  0: in memory order (dim 0 fastest); 1: dim 3 fastest; 2: scattered (odd multiplier modulo the
// power-of-two box count: neighbours in time are far apart in memory); 3: dim 0 fastest, dims 1..3 scattered; 4 / 5: memory order, but
// the 2 / 4 boxes that share a 128-byte line / a 256-byte stretch run on the same XCD (same L2) one after the other
template <int R0, int R1, int R2, int R3, int MODE, bool NT, int ORDER = 0>
__global__ void __launch_bounds__(1024) k_box(const double* __restrict__ A, double* __restrict__ B, int n, double* sink) {
    static_assert(R0 * R1 * R2 * R3 == 4096, "32 KiB boxes");
    const int nb0 = n / R0, nb1 = n / R1, nb2 = n / R2;
    unsigned b = blockIdx.x;
    int b0, b1, b2, b3;
    if (ORDER == 2) b = (b * 40503u) & (gridDim.x - 1);  // (power-of-two grids only)
    if (ORDER == 4 || ORDER == 5) {  // 2 (4) boxes adjacent along dim 0 go to the SAME XCD in consecutive slots (workgroup x runs on XCD x % 8)
        constexpr unsigned G = ORDER == 4 ? 2 : 4;
        const unsigned xcd = b % 8, j = b / 8;
        b = G * ((j / G) * 8 + xcd) + j % G;
    }
    if (ORDER == 3) { const unsigned lo = b % nb0, hi = b / nb0; b = lo + nb0 * ((hi * 40503u) & (gridDim.x / nb0 - 1)); }
    if (ORDER == 1) {
        const int nb3 = n / R3;
        b3 = b % nb3; b /= nb3;
        b2 = b % nb2; b /= nb2;
        b1 = b % nb1;
        b0 = b / nb1;
    } else {
        b0 = b % nb0; b /= nb0;
        b1 = b % nb1; b /= nb1;
        b2 = b % nb2;
        b3 = b / nb2;
    }
    const size_t n1 = (size_t)n, n2 = n1 * n1, n3 = n2 * n1;
    const size_t base = (size_t)b0 * R0 + n1 * ((size_t)b1 * R1) + n2 * ((size_t)b2 * R2) + n3 * ((size_t)b3 * R3);
    constexpr int VPR = R0 / 2;  // 16-byte vectors per row
    double acc = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {  // 2048 vectors per box, 1024 lanes
        const int v = pass * 1024 + threadIdx.x;
        const int c = v % VPR;
        int r = v / VPR;
        const int i1 = r % R1; r /= R1;
        const int i2 = r % R2;
        const int i3 = r / R2;
        const size_t off = base + 2 * c + n1 * i1 + n2 * i2 + n3 * i3;
        d2 x = {1.0, 2.0};
        if (MODE != 2) x = *reinterpret_cast<const d2*>(A + off);
        if (MODE == 1) acc += x.x + x.y;
        if (MODE != 1) {
            if (NT) __builtin_nontemporal_store(x, reinterpret_cast<d2*>(B + off));
            else *reinterpret_cast<d2*>(B + off) = x;
        }
    }
    if (MODE == 1 && acc == 12345.678) *sink = acc;
}

template <int R0, int R1, int R2, int R3, int MODE, bool NT, int ORDER = 0>
static double run(const double* A, double* B, int n, double* sink) {
    if (n % R0 || n % R1 || n % R2 || n % R3) return 0;
    const unsigned grid = (unsigned)(((size_t)n * n * n * n) / 4096);
    auto kern = k_box<R0, R1, R2, R3, MODE, NT, ORDER>;
    if ((ORDER == 2 || ORDER == 3) && (grid & (grid - 1))) return 0;
    if (ORDER >= 4 && (grid % 32 || (n / R0) % 4)) return 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, A, B, n, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, A, B, n, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 3 < best) best = ms / 3;
    }
    const double bytes = (MODE == 0 ? 16.0 : 8.0) * (double)n * n * n * n;
    return bytes / best / 1e6;  // GB/s
}

template <int R0, int R1, int R2, int R3>
static void row(const double* A, double* B, int n, double* sink) {
    const double c = run<R0, R1, R2, R3, 0, false>(A, B, n, sink), cn = run<R0, R1, R2, R3, 0, true>(A, B, n, sink);
    const double r = run<R0, R1, R2, R3, 1, false>(A, B, n, sink), w = run<R0, R1, R2, R3, 2, false>(A, B, n, sink);
    if (c == 0) return;
    std::printf("  rows of %4d B (box %2d x %2d x %2d x %2d): copy %6.0f GB/s (%.2f of 8 TB/s), nt stores %6.0f (%.2f) | read only %6.0f | write only %6.0f\n", R0 * 8, R0,
                R1, R2, R3, c, c / 8000, cn, cn / 8000, r, w);
    const double o1 = run<R0, R1, R2, R3, 0, false, 1>(A, B, n, sink), o2 = run<R0, R1, R2, R3, 0, false, 2>(A, B, n, sink), o3 = run<R0, R1, R2, R3, 0, false, 3>(A, B, n, sink);
    const double o4 = run<R0, R1, R2, R3, 0, false, 4>(A, B, n, sink), o5 = run<R0, R1, R2, R3, 0, false, 5>(A, B, n, sink);
    std::printf("                 boxes dealt dim 3 fastest: copy %6.0f GB/s; scattered: %6.0f; dim 0 fastest, the rest scattered: %6.0f; pairs / quads along dim 0 on one XCD: %6.0f / %6.0f\n", o1, o2, o3, o4, o5);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
        const int n = argc > 1 ? std::atoi(argv[a]) : 128;
        const size_t N = (size_t)n * n * n * n;
        double *A, *B, *sink;
        CK(hipMalloc(&A, N * 8));
        CK(hipMalloc(&B, N * 8));
        CK(hipMalloc(&sink, 8));
        CK(hipMemset(A, 0, N * 8));
        CK(hipMemset(B, 0, N * 8));
        std::printf("n = %d (%.2f GiB per array; row strides %zu B, %zu KiB, %zu KiB)\n", n, N * 8 / 1073741824.0, (size_t)n * 8, (size_t)n * n * 8 / 1024,
                    (size_t)n * n * n * 8 / 1024);
        row<8, 8, 8, 8>(A, B, n, sink);
        row<16, 8, 8, 4>(A, B, n, sink);
        row<32, 8, 4, 4>(A, B, n, sink);
        row<64, 4, 4, 4>(A, B, n, sink);
        row<128, 4, 4, 2>(A, B, n, sink);
        CK(hipFree(A));
        CK(hipFree(B));
        CK(hipFree(sink));
    }
    return 0;
}
