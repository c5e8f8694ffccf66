// Micro-benchmark behind the ring-staged operator applies (csrc/apply_ring.hip): a persistent workgroup per CU with
// LOADER wave(s) that stream a 256-cell block's records straight into LDS (global_load_lds_dwordx4: no VGPR destination, the
// per-lane SOURCE address makes the halo rows a gather) two blocks ahead of four CONSUMER waves that read LDS only.
// Checks on MI355X: (1) > 64 KB of dynamic LDS per workgroup, (2) M0 destinations beyond 64 KB, (3) lane -> LDS mapping
// (base + lane * 16), (4) counted vmcnt across raw s_barriers, (5) the rate one / two loader waves per CU sustain on the memory
// pattern of the KNP apply (two species rows + gphi row per cell, a gathered halo list per block, 64 B written per cell).
// build: hipcc --offload-arch=gfx950 -O3 -o glds_ring glds_ring.hip ;  run: ./glds_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// four consecutive 1 KiB pieces from one address register and one M0: the instruction offset moves BOTH the global and the LDS address
__device__ __forceinline__ void glds16_run4(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_nosave(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int BLK = 256, HMAX = 224, ENT = BLK + HMAX, NS = 2, SLOTS = 3;
constexpr int XBYTES = ENT * 32, GBYTES = BLK * 32, GHBYTES = 256 * 16, SLOT = NS * XBYTES + GBYTES + GHBYTES, LISTB = 1024;
constexpr int NDATA = NS * 8 + NS * 7 + 8 + 4;     // DMA instructions per block

// block of iteration n of this workgroup, or -1
__device__ __forceinline__ long blk_of(long n, long first, long last, long member, long members) {
    const long b = first + member + n * members;
    return b < last ? b : -1;
}

template <int NLOAD, bool LEAN = false>
__global__ __launch_bounds__(256 + 64 * NLOAD) void k_ring(long nc, const double* __restrict__ x, const double* __restrict__ g,
                                                           const int* __restrict__ list, int hs, double* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    const unsigned list0 = base + SLOTS * SLOT;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const long nblk = (nc + BLK - 1) / BLK, chunk = (nblk + 7) / 8;
    const long q = blockIdx.x & 7, first = q * chunk, last = first + chunk < nblk ? first + chunk : nblk;
    const long member = blockIdx.x >> 3, members = gridDim.x >> 3;
    if (blk_of(0, first, last, member, members) < 0) return;
    if (wave >= 4) {
        // ---------------- loader ----------------
        const int lw = wave - 4;
        auto list_dma = [&](long n) {
            const long b = blk_of(n, first, last, member, members);
            if (b < 0 || lw != 0) return;
            const int off = lane * 4 < hs ? lane * 4 : 0;
            glds16(list + b * hs + off, list0 + (unsigned)(n & 3) * LISTB);
        };
        auto data_dma = [&](long n) {
            const long b = blk_of(n, first, last, member, members);
            const unsigned slot = base + (unsigned)(n % SLOTS) * SLOT;
            const lds_int* L = (const lds_int*)(smem + SLOTS * SLOT + (n & 3) * LISTB);
            int i = 0;                                                   // running instruction number (round-robin over the loader waves)
            const long c0 = b * BLK;
            if (LEAN) {
                // own rows: runs of four pieces (no clamping: the arrays are padded); halo rows without save / restore of M0
                static_assert(!LEAN || NLOAD <= 2, "");
                const double* srcs[3] = {x, x + nc * 4, g};
                const unsigned dsts[3] = {slot, slot + XBYTES, slot + NS * XBYTES};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int r = 0; r < 2; ++r, ++i)
                        if (i % NLOAD == lw) glds16_run4(srcs[a] + (c0 + r * 128) * 4 + lane * 2, dsts[a] + r * 4096);
#pragma unroll
                for (int p = 0; p < 7; ++p) {
                    const int e = p * 32 + (lane >> 1);
                    const int src = e < hs ? L[e] : 0;
                    const long Kp = (src > 0 ? src : 0) >> 2;
#pragma unroll
                    for (int k = 0; k < NS; ++k, ++i)
                        if (i % NLOAD == lw) glds16_nosave(x + (long)k * nc * 4 + Kp * 4 + (lane & 1) * 2, slot + k * XBYTES + BLK * 32 + p * 1024);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p, ++i)
                    if (i % NLOAD == lw) {
                        const int e = p * 64 + lane;
                        const int src0 = e < hs ? L[e] : 0;
                        const int src = src0 > 0 ? src0 : 0;
                        glds16_nosave(g + (long)(src >> 2) * 4 + ((src & 3) >> 1) * 2, slot + NS * XBYTES + GBYTES + p * 1024);
                    }
                return;
            }
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int p = 0; p < 8; ++p, ++i)
                    if (i % NLOAD == lw) {
                        long row = c0 + p * 32 + (lane >> 1);
                        row = row < nc ? row : 0;
                        glds16(x + (long)k * nc * 4 + row * 4 + (lane & 1) * 2, slot + k * XBYTES + p * 1024);
                    }
#pragma unroll
            for (int p = 0; p < 8; ++p, ++i)
                if (i % NLOAD == lw) {
                    long row = c0 + p * 32 + (lane >> 1);
                    row = row < nc ? row : 0;
                    glds16(g + row * 4 + (lane & 1) * 2, slot + NS * XBYTES + p * 1024);
                }
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const int e = p * 32 + (lane >> 1);
                const int src = e < hs ? L[e] : -1;
                const long Kp = src >= 0 ? (src >> 2) : 0;
#pragma unroll
                for (int k = 0; k < NS; ++k, ++i)
                    if (i % NLOAD == lw) glds16(x + (long)k * nc * 4 + Kp * 4 + (lane & 1) * 2, slot + k * XBYTES + BLK * 32 + p * 1024);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p, ++i)
                if (i % NLOAD == lw) {
                    const int e = p * 64 + lane;
                    const int src = e < hs ? L[e] : -1;
                    const long Kp = src >= 0 ? (src >> 2) : 0;
                    const int j = src >= 0 ? (src & 3) : 0;
                    glds16(g + Kp * 4 + (j >> 1) * 2, slot + NS * XBYTES + GBYTES + p * 1024);
                }
        };
        list_dma(0); list_dma(1);
        WAIT_VM(0);
        wg_barrier();                           // loader waves see the lists (issued by loader 0)
        data_dma(0);
        list_dma(2);
        if (blk_of(1, first, last, member, members) >= 0) data_dma(1);
        WAIT_VM(0);
        wg_barrier();
        for (long n = 0; blk_of(n, first, last, member, members) >= 0; ++n) {
            list_dma(n + 3);
            if (blk_of(n + 2, first, last, member, members) >= 0) {
                data_dma(n + 2);
                if (NLOAD == 1) WAIT_VM(42); else if (!LEAN) WAIT_VM(21); else WAIT_VM(21);     // data(n + 2) may stay in flight; the list and data(n + 1) have landed (LEAN, 2 loaders: 3 runs x 4 + 9 = 21 each)
            } else {
                WAIT_VM(0);
            }
            wg_barrier();
        }
    } else {
        // ---------------- consumers ----------------
        const unsigned t = threadIdx.x;
        wg_barrier();
        wg_barrier();
        for (long n = 0;; ++n) {
            const long b = blk_of(n, first, last, member, members);
            if (b < 0) break;
            const long c = b * BLK + t;
            const char* slot = smem + (n % SLOTS) * SLOT;
            const lds_double* X0 = (const lds_double*)(slot);
            const lds_double* X1 = (const lds_double*)(slot + XBYTES);
            const lds_double* G = (const lds_double*)(slot + NS * XBYTES);
            const lds_double* GH = (const lds_double*)(slot + NS * XBYTES + GBYTES);
            const unsigned h = BLK + (t % (unsigned)hs);              // a halo entry
            double y0[4], y1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                y0[a] = X0[t * 4 + a] + G[t * 4 + a] + X0[h * 4 + a];
                y1[a] = X1[t * 4 + a] + X1[h * 4 + (3 - a)] + GH[(h - BLK) * 2 + (a & 1)];
            }
            if (c < nc) {
                *reinterpret_cast<double2*>(y + c * 4) = make_double2(y0[0], y0[1]);
                *reinterpret_cast<double2*>(y + c * 4 + 2) = make_double2(y0[2], y0[3]);
                *reinterpret_cast<double2*>(y + nc * 4 + c * 4) = make_double2(y1[0], y1[1]);
                *reinterpret_cast<double2*>(y + nc * 4 + c * 4 + 2) = make_double2(y1[2], y1[3]);
            }
            wg_barrier();
        }
    }
}

// the same work with register staging (one thread per cell, everything gathered from global memory): the lower bound of what the
// pattern costs without any staging logic
__global__ __launch_bounds__(256) void k_plain(long nc, const double* __restrict__ x, const double* __restrict__ g, const int* __restrict__ list,
                                               int hs, double* __restrict__ y) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const long b = c / 256;
    const int src = list[b * hs + (threadIdx.x % (unsigned)hs)];
    const long Kp = src >= 0 ? (src >> 2) : 0;
    const int j = src >= 0 ? (src & 3) : 0;
    double y0[4], y1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        y0[a] = x[c * 4 + a] + g[c * 4 + a] + x[Kp * 4 + a];
        y1[a] = x[nc * 4 + c * 4 + a] + x[nc * 4 + Kp * 4 + (3 - a)] + g[Kp * 4 + (j >> 1) * 2 + (a & 1)];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) { y[c * 4 + a] = y0[a]; y[nc * 4 + c * 4 + a] = y1[a]; }
}

int main(int argc, char** argv) {
    const int hs = 216;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", prop.gcnArchName, ncu, prop.sharedMemPerBlock,
           prop.maxSharedMemoryPerMultiProcessor);
    const size_t lds = SLOTS * SLOT + 4 * LISTB;
    printf("LDS per workgroup: %zu bytes, %d DMA instructions per block\n", lds, NDATA);
    CHK(hipFuncSetAttribute((const void*)k_ring<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)k_ring<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)k_ring<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void*)k_ring<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (long nc : {995328L, 7962624L}) {
        const long nblk = (nc + BLK - 1) / BLK;
        std::vector<double> hx(2 * nc * 4), hg(nc * 4);
        std::vector<int> hl(nblk * hs);
        srand(1);
        for (auto& v : hx) v = rand() / (double)RAND_MAX;
        for (auto& v : hg) v = rand() / (double)RAND_MAX;
        for (long b = 0; b < nblk; ++b) {
            const int len = 120 + rand() % (hs - 120 + 1);
            for (int e = 0; e < hs; ++e) {
                long r = b * BLK + (rand() % 4096) - 2048;
                r = r < 0 ? 0 : (r >= nc ? nc - 1 : r);
                hl[b * hs + e] = e < len ? (int)(r * 4 + rand() % 4) : -1;
            }
        }
        double *x, *g, *y, *y2; int* l;
        CHK(hipMalloc(&x, hx.size() * 8 + 16384)); CHK(hipMalloc(&g, hg.size() * 8 + 16384)); CHK(hipMalloc(&y, 2 * nc * 4 * 8)); CHK(hipMalloc(&y2, 2 * nc * 4 * 8));
        CHK(hipMalloc(&l, hl.size() * 4 + 1024));
        CHK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(g, hg.data(), hg.size() * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
        const double bytes = nc * (64.0 + 32.0 + 64.0) + nblk * hs * 4.0;      // own rows + outputs + lists (halo rows come from cache)
        float ms;
        const int reps = 50;
        // reference
        CHK(hipMemset(y2, 0, 2 * nc * 32));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_plain, dim3((unsigned)nblk), dim3(256), 0, st, nc, x, g, l, hs, y2);
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_plain, dim3((unsigned)nblk), dim3(256), 0, st, nc, x, g, l, hs, y2);
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("nc %ld  plain gather kernel        : %7.2f us  %.2f TB/s of own + output bytes\n", nc, 1e3 * ms / reps, bytes / (1e9 * ms / reps));
        std::vector<double> ref(2 * nc * 4), got(2 * nc * 4);
        CHK(hipMemcpy(ref.data(), y2, ref.size() * 8, hipMemcpyDeviceToHost));
        for (int nl = 1; nl <= 4; ++nl)
            for (int wgs : {ncu}) {
                CHK(hipMemset(y, 0, 2 * nc * 32));
                auto launch = [&]() {
                    if (nl == 1) hipLaunchKernelGGL(k_ring<1>, dim3(wgs), dim3(320), lds, st, nc, x, g, l, hs, y);
                    else if (nl == 2) hipLaunchKernelGGL(k_ring<2>, dim3(wgs), dim3(384), lds, st, nc, x, g, l, hs, y);
                    else if (nl == 3) hipLaunchKernelGGL((k_ring<1, true>), dim3(wgs), dim3(320), lds, st, nc, x, g, l, hs, y);
                    else hipLaunchKernelGGL((k_ring<2, true>), dim3(wgs), dim3(384), lds, st, nc, x, g, l, hs, y);
                };
                for (int w = 0; w < 3; ++w) launch();
                CHK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) launch();
                CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
                CHK(hipEventElapsedTime(&ms, e0, e1));
                CHK(hipMemcpy(got.data(), y, got.size() * 8, hipMemcpyDeviceToHost));
                double d = 0; long bad = 0;
                for (size_t i = 0; i < got.size(); ++i) { const double e = fabs(got[i] - ref[i]); if (e > 0) ++bad; d = fmax(d, e); }
                printf("nc %ld  ring, %d loader wave(s)%s, %3d WGs: %7.2f us  %.2f TB/s   max diff %g (%ld entries differ)\n", nc, (nl - 1) % 2 + 1, nl > 2 ? " LEAN" : "", wgs,
                       1e3 * ms / reps, bytes / (1e9 * ms / reps), d, bad);
            }
        hipFree(x); hipFree(g); hipFree(y); hipFree(y2); hipFree(l);
    }
    return 0;
}
