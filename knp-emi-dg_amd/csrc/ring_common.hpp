// Shared pieces of the ring-staged apply kernels (apply_ring.hip: structured meshes with geometry classes; apply_ring_u.hip: any
// 3D mesh, geometry from staged vertex coordinates): the LDS-DMA primitives, the LDS-only workgroup barrier, the block walk of a
// persistent workgroup and the swizzled 32-byte-row image of a nodal vector.
#pragma once
#include "cell_geom.hpp"

namespace ring {

constexpr int RB = 256;           // cells per block (= KNP_HALO_BLK: the halo tables are built for it)
static_assert(RB == KNP_HALO_BLK, "the halo tables are built for 256-cell blocks");

// one LDS-DMA instruction: lane l copies 16 bytes from its own source address to (lds_dst + 16 l); lds_dst is wave-uniform.  M0 is
// written in the statement that reads it and declared clobbered, so the compiler never assumes a value of its own survives.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// four consecutive 1 KiB pieces from ONE address register and ONE M0 value: the instruction offset moves both the global and the
// LDS address (checked by tools/microbench/glds_ring.hip)
__device__ __forceinline__ void glds16_run4(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// workgroup barrier that orders LDS traffic only: the loader's DMAs stay in flight across it (it counts them itself), the
// consumers' stores are never waited for
__device__ __forceinline__ void ring_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Blocks of one workgroup: the block range is cut into 8 contiguous chunks, one per XCD (XCD = blockIdx.x & 7 under round-robin
// dispatch: an XCD's workgroups walk adjacent blocks, so the halo rows come out of its own L2; speed only); inside a chunk the
// XCD's workgroups take the blocks round-robin.
struct RingWalk {
    int64_t b_lo, first, last, member, members;
    __device__ __forceinline__ explicit RingWalk(const MeshDev& m) {
        b_lo = m.c_begin / RB;
        const int64_t nblk = (m.c_end - 1) / RB - b_lo + 1;
        const int64_t chunk = (nblk + 7) / 8;
        first = (int64_t)(blockIdx.x & 7u) * chunk;
        last = first + chunk < nblk ? first + chunk : nblk;
        member = blockIdx.x >> 3;
        members = gridDim.x >> 3;
    }
    // absolute block (cells [256 b, 256 b + 256)) of iteration n, or -1
    __device__ __forceinline__ int64_t blk(int64_t n) const {
        const int64_t b = first + member + n * members;
        return b < last ? b_lo + b : -1;
    }
};

// LDS image of a nodal vector: 32-byte rows, read by the consumers with 16-byte ds_read_b128 -- lanes that read consecutive rows
// hit 16-byte bank groups (2 R + half) mod 16, i.e. rows 8 apart collide.  The two halves of every second group of 8 rows are
// therefore stored SWAPPED (the DMA writes lane-linear, so the swap is made on the per-lane source address): conflict-free.
__device__ __forceinline__ int swz_half(int lane) { return ((lane & 1) ^ ((lane >> 4) & 1)) * 2; }       // source half (in doubles) of lane 2 i + h
// 256 own rows of one nodal vector: 8 instructions, lanes 2 i / 2 i + 1 carry the two halves of a row; two runs of four pieces when
// the whole block lies inside the vector (every block but possibly the last one)
__device__ __forceinline__ void dma_own_rows(const double* __restrict__ v, int64_t c0, int64_t nc, unsigned dst, int lane) {
    if (c0 + RB <= nc) {
        const double* src = v + (c0 + (lane >> 1)) * 4 + swz_half(lane);
        glds16_run4(src, dst);
        glds16_run4(src + 128 * 4, dst + 4096);
        return;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        int64_t row = c0 + p * 32 + (lane >> 1);
        row = row < nc ? row : 0;
        glds16(v + row * 4 + swz_half(lane), dst + p * 1024);
    }
}
// consumer side: the four values of row R of such an image
__device__ __forceinline__ void lds_row(const lds_double* base, unsigned R, double* r) {
    typedef double __attribute__((ext_vector_type(2))) vdouble2;
    typedef __attribute__((address_space(3))) vdouble2 lds_vdouble2;
    const unsigned s = (R >> 3) & 1u;
    const vdouble2 a = *(const lds_vdouble2*)(base + 4 * R + 2 * s);
    const vdouble2 b = *(const lds_vdouble2*)(base + 4 * R + 2 * (s ^ 1u));
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
}
__device__ __forceinline__ int64_t list_cell(const lds_int* L, int e, int hs) {
    const int src = e < hs ? L[e] : -1;
    return src >= 0 ? (int64_t)(src >> 2) : 0;
}

}  // namespace ring
