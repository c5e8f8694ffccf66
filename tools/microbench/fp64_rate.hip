// FP64 vector issue rate of MI355X as a function of the waves per SIMD: the operator applies of this repository are FP64-VALU work
// (P1: ~260 FP64 instructions per cell and species, P2: ~2 750 per cell), so their compute floor is set by this number, not by the
// 78.6 TFLOP/s of the data sheet alone.  Each lane runs 8 independent FMA / MUL+ADD chains; the clock the chip holds is measured
// with s_memtime (shader cycles) against s_memrealtime (100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip ;  run: ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_rate(int iters, double a, double* out, unsigned long long* clk) {
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) v[i] = fma(v[i], a, 1e-9);                      // v_fma_f64
                else if (MODE == 1) v[i] = v[i] * a;                           // v_mul_f64
                else v[i] = v[i] + a;                                          // v_add_f64
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, iters = 20000;
    double* out; unsigned long long* clk;
    CHK(hipMalloc(&out, sizeof(double) * ncu * 8 * 256)); CHK(hipMalloc(&clk, 16 * ncu * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[3] = {"v_fma_f64", "v_mul_f64", "v_add_f64"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wps : {1, 2, 4, 8}) {                      // waves per SIMD = 256-thread workgroups per CU
            const int grid = ncu * wps;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(grid), dim3(256), 0, 0, iters, 1.0000001, out, clk);
                else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(grid), dim3(256), 0, 0, iters, 1.0000001, out, clk);
                else hipLaunchKernelGGL(k_rate<2>, dim3(grid), dim3(256), 0, 0, iters, 1.0000001, out, clk);
            };
            launch(); CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0)); launch(); CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ninst = (double)iters * 32;                           // wave instructions per wave
            const double ghz = (double)h[0] / ((double)h[1] * 10.0) ;          // shader cycles per ns
            printf("%s, %d wave(s) per SIMD: %.2f shader cycles per wave instruction per wave, %.2f per SIMD; clock %.2f GHz; chip %.1f TFLOP/s (%s)\n",
                   names[mode], wps, (double)h[0] / ninst, (double)h[0] / ninst / wps, ghz,
                   (mode == 0 ? 2.0 : 1.0) * ninst * 64.0 * 4 * wps * ncu / (ms * 1e-3) / 1e12, mode == 0 ? "FMA = 2 flop" : "1 flop");
        }
    return 0;
}
