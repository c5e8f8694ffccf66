// Micro-benchmark behind DESIGN.md section 5 ("a persistent kernel for the coarse levels of the V-cycle?"): cost of a device-scope
// grid barrier between dependent phases on MI355X (8 XCDs with private L2s: a release / acquire pair at agent scope writes back and
// invalidates the L2), against the kernel boundary of a captured hipGraph.  Each phase is a tiny dependent sweep over a 28 703-row
// vector pair (the size of conforming level 1 at r=2): y[i] = 0.5 * (x[(i + 1) % n] + x[i]) + 1, then x <-> y.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ;  run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nblocks, unsigned& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++phase;
        __threadfence();                                            // release: this workgroup's writes (agent scope)
        atomicAdd(bar, 1u);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nblocks) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                            // acquire
    }
    __syncthreads();
}

__global__ void k_phase(int n, const double* __restrict__ x, double* __restrict__ y) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = 0.5 * (x[(i + 1) % n] + x[i]) + 1.0;
}

__global__ void k_persistent(int n, int nphases, double* a, double* b, unsigned* bar) {
    unsigned phase = 0;
    double *x = a, *y = b;
    for (int p = 0; p < nphases; ++p) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const double xn = __builtin_nontemporal_load(x + (i + 1) % n), xi = __builtin_nontemporal_load(x + i);
            y[i] = 0.5 * (xn + xi) + 1.0;
        }
        grid_sync(bar, gridDim.x, phase);
        double* t = x; x = y; y = t;
    }
}

int main() {
    const int n = 28703, nphases = 16, reps = 200;
    double *a, *b; unsigned* bar;
    CHK(hipMalloc(&a, n * 8)); CHK(hipMalloc(&b, n * 8)); CHK(hipMalloc(&bar, 4));
    CHK(hipMemset(a, 0, n * 8)); CHK(hipMemset(b, 0, n * 8));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    // (1) separate kernels in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < nphases; ++p) hipLaunchKernelGGL(k_phase, dim3((n + 255) / 256), dim3(256), 0, st, n, (const double*)((p & 1) ? b : a), (p & 1) ? a : b);
    CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of %d kernels: %.2f us per phase\n", nphases, 1e3 * ms / reps / nphases);
    std::vector<double> ref(n); CHK(hipMemcpy(ref.data(), a, n * 8, hipMemcpyDeviceToHost));
    // (2) persistent kernel with grid barriers, several grid sizes
    for (int G : {8, 16, 32, 64, 113, 128, 256}) {
        CHK(hipMemset(a, 0, n * 8)); CHK(hipMemset(b, 0, n * 8));
        for (int w = 0; w < 3; ++w) { CHK(hipMemsetAsync(bar, 0, 4, st)); hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, st, n, nphases, a, b, bar); }
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) { CHK(hipMemsetAsync(bar, 0, 4, st)); hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, st, n, nphases, a, b, bar); }
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("persistent, %3d workgroups: %.2f us per phase (incl. launch + memset / %d)\n", G, 1e3 * ms / reps / nphases, nphases);
    }
    // correctness of the last persistent run against the graph run: same number of phase applications per launch
    CHK(hipMemset(a, 0, n * 8)); CHK(hipMemset(b, 0, n * 8));
    CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
    CHK(hipMemcpy(ref.data(), a, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemset(a, 0, n * 8)); CHK(hipMemset(b, 0, n * 8));
    CHK(hipMemsetAsync(bar, 0, 4, st)); hipLaunchKernelGGL(k_persistent, dim3(128), dim3(256), 0, st, n, nphases, a, b, bar);
    CHK(hipStreamSynchronize(st));
    std::vector<double> got(n); CHK(hipMemcpy(got.data(), a, n * 8, hipMemcpyDeviceToHost));
    double d = 0; for (int i = 0; i < n; ++i) d = fmax(d, fabs(got[i] - ref[i]));
    printf("max difference persistent vs graph: %g\n", d);
    return 0;
}
