// Micro-benchmark: what does one back-to-back launch cost on MI355X as a function of how the
// kernel gets its (wave-uniform) parameters?  (a) empty kernel, (b) big by-value kernarg
// struct, a few scattered fields read, (c) the same struct behind a device pointer.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { unsigned v[640]; };   // 2560 bytes
struct Small { unsigned v[16]; };

__global__ void k_empty(unsigned* out) { if (out == (unsigned*)1) out[0] = 1; }
template <class S>
__global__ void k_byval(const S a, unsigned* out) {
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S) / 4); i += 37) s += a.v[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;  // never true: keeps the loads alive
}
__global__ void k_byptr(const Big* __restrict__ a, unsigned* out) {
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 640; i += 37) s += a->v[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
// dependent chain: second load's address depends on the first (like slot -> descriptor)
__global__ void k_byval_dep(const Big a, unsigned* out) {
    unsigned i0 = a.v[0] & 511u;
    unsigned i1 = a.v[i0] & 511u;
    unsigned s = a.v[i1];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <class L>
static float time_graph(hipStream_t st, int reps, L launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < reps; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int t = 0; t < 5; ++t) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1000.f / reps;  // us per launch
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* out; CK(hipMalloc(&out, 4096));
    Big hb; memset(&hb, 0, sizeof hb); Small hs; memset(&hs, 0, sizeof hs);
    Big* db; CK(hipMalloc(&db, sizeof(Big))); CK(hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice));
    const int reps = 1000;
    for (int grid : {256, 1024, 4096}) {
        float a = time_graph(st, reps, [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, out); });
        float b = time_graph(st, reps, [&] { hipLaunchKernelGGL(k_byval<Small>, dim3(grid), dim3(256), 0, st, hs, out); });
        float c = time_graph(st, reps, [&] { hipLaunchKernelGGL(k_byval<Big>, dim3(grid), dim3(256), 0, st, hb, out); });
        float d = time_graph(st, reps, [&] { hipLaunchKernelGGL(k_byptr, dim3(grid), dim3(256), 0, st, db, out); });
        float e = time_graph(st, reps, [&] { hipLaunchKernelGGL(k_byval_dep, dim3(grid), dim3(256), 0, st, hb, out); });
        printf("grid=%5d x256  empty %.2f us | by-value 64B %.2f | by-value 2560B %.2f | device ptr 2560B %.2f | by-value dependent x3 %.2f\n",
               grid, a, b, c, d, e);
    }
    return 0;
}
