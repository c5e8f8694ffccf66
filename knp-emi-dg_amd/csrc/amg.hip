// Device side of the auxiliary-space preconditioner  M^-1 = B^-1 + P Ac^+ P^T  (setup: knpemidg/amg.py).
//  * DG <-> conforming-P1 transfers (gather forms, deterministic),
//  * one V-cycle over a smoothed-aggregation hierarchy stored as CSR levels: Chebyshev-Jacobi polynomial
//    smoothing (fused SpMV + recurrence kernels, no sequential sweeps), CSR restriction / prolongation,
//    dense pseudo-inverse on the coarsest level.
// Stands in for hypre BoomerAMG (reference: src/knpemidg/solver.py:433, 505, 688, 767).
#include "../../include/knpemi_hip.h"
#include "knpemi_internal.hpp"
#include "amg.hpp"
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <thread>

namespace {

// number of right-hand-side columns of the hierarchy whose V-cycle is being enqueued (grid.y of every kernel): the KNP
// species share one hierarchy when their diffusion coefficients are close, and one chain of kernels then carries all of
// them instead of one concurrent chain per species
int s_ncol = 1;
#define GRIDX(nx) dim3((unsigned)(nx), (unsigned)s_ncol)

template <typename T> int up(knp_ctx* c, T** dst, const T* src, size_t n) {
    HIPCHK(c, hipMalloc((void**)dst, (n ? n : 1) * sizeof(T)));
    if (n) HIPCHK(c, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int up_csr(knp_ctx* c, CsrDev& M, int64_t nrows, int64_t ncols, const int32_t* rp, const int32_t* ci, const double* v) {
    M.nrows = nrows; M.ncols = ncols; M.nnz = rp[nrows];
    int rc = up(c, &M.rowptr, rp, (size_t)nrows + 1);
    rc |= up(c, &M.col, ci, (size_t)M.nnz);
    std::vector<float> v32((size_t)M.nnz);
    for (size_t i = 0; i < v32.size(); ++i) v32[i] = (float)v[i];
    rc |= up(c, &M.val, v32.data(), v32.size());
    return rc;
}

void free_csr(CsrDev& M) { hipFree(M.rowptr); hipFree(M.col); hipFree(M.val); M = CsrDev(); }

// rows with up to this many entries on average get 4 lanes (KNP_AMG_G4; 12 = never; 40: -1.3 % per step at r=2 P1, -1 % for P2 at r=1)
static double g4_limit() {
    static const double v = getenv("KNP_AMG_G4") ? atof(getenv("KNP_AMG_G4")) : 40.0;
    return v;
}

// Level vectors carry NC right-hand-side columns INTERLEAVED ([n][NC]; further column groups behind each other): a gather of entry
// `col` fetches both species' values from one 16-byte location.  With the columns stored one behind the other every gather pulled
// its own 64-byte sector out of L2 for 8 useful bytes, and the two-column kernels were bound by exactly that: k_csr-type kernels
// cost 24 bytes of useful data per entry but ran at the L2's sector rate (a 28-entry-per-row product took 1.9x the time of a
// 15-entry one -- linear in the entries, not in the launches; profiles/README.md, round 2).
// Partial dot products of CSR row `row` with the NC interleaved columns of x over this lane's entries (lane, lane+G, ...); the loop
// is unrolled so that U (col, val) pairs are in flight per lane.
template <int G, int NC>
__device__ __forceinline__ void row_dot(const CsrDev& A, int64_t row, int lane, const double* __restrict__ x, double* out) {
    constexpr int U = NC == 1 ? 4 : 2;                    // (col, val) pairs in flight per lane
    double s[U][NC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NC; ++j) s[u][j] = 0.0;
    if (row < A.nrows) {
        const int e = A.rowptr[row + 1];
        int k = A.rowptr[row] + lane;
        for (; k + (U - 1) * G < e; k += U * G) {
            int cc[U];
            double vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cc[u] = A.col[k + u * G]; vv[u] = A.val[k + u * G]; }      // float -> double
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NC; ++j) s[u][j] = fma(vv[u], x[(int64_t)cc[u] * NC + j], s[u][j]);
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
            if (k < e) {
                const int c0 = A.col[k];
                const double v0 = A.val[k];
#pragma unroll
                for (int j = 0; j < NC; ++j) s[u][j] = fma(v0, x[(int64_t)c0 * NC + j], s[u][j]);
                k += G;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        double t = U == 4 ? (s[0][j] + s[1][j]) + (s[2 % U][j] + s[3 % U][j]) : s[0][j] + s[1][j];
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) t += __shfl_down(t, off, G);
        out[j] = t;
    }
}

// y = A x (MODE 0) | y = b - A x (MODE 1) | y += A x (MODE 2); G lanes cooperate on one row; NC columns per thread, the rest of the
// right-hand-side columns along grid.y
template <int MODE, int G, int NC>
__global__ __launch_bounds__(256) void k_csr(CsrDev A, const double* __restrict__ x, const double* __restrict__ b,
                                             double* __restrict__ y) {
    x += (int64_t)blockIdx.y * NC * A.ncols;
    y += (int64_t)blockIdx.y * NC * A.nrows;
    if (MODE == 1) b += (int64_t)blockIdx.y * NC * A.nrows;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int lane = threadIdx.x % G;
    double s[NC];
    row_dot<G, NC>(A, row, lane, x, s);
    if (row < A.nrows && lane == 0) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int64_t o = row * NC + j;
            if (MODE == 0) y[o] = s[j];
            else if (MODE == 1) y[o] = b[o] - s[j];
            else y[o] += s[j];
        }
    }
}

// b_c = R r together with the first (zero-guess) Chebyshev update of the level that receives it:  rc = b_c ; d = dinv b_c / theta ;
// x = d  -- the separate k_cheb_first launch (5 us, three vector writes of a vector this kernel has in registers) goes away
template <int G, int NC>
__global__ __launch_bounds__(256) void k_csr_first(CsrDev R, const double* __restrict__ rfine, const double* __restrict__ dinv, double inv_theta,
                                                   double* __restrict__ bc, double* __restrict__ rc, double* __restrict__ d,
                                                   double* __restrict__ x) {
    rfine += (int64_t)blockIdx.y * NC * R.ncols;
    const int64_t o = (int64_t)blockIdx.y * NC * R.nrows;
    bc += o; rc += o; d += o; x += o;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int lane = threadIdx.x % G;
    double s[NC];
    row_dot<G, NC>(R, row, lane, rfine, s);
    if (row < R.nrows && lane == 0) {
        const double w = dinv[row] * inv_theta;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int64_t q = row * NC + j;
            const double v = w * s[j];
            bc[q] = s[j];
            rc[q] = s[j];
            d[q] = v;
            x[q] = v;
        }
    }
}

#define LAUNCH_BY_DENSITY(KERN, A, ...)                                                                                       \
    do {                                                                                                                      \
        const double avg_ = (A).nrows ? (double)(A).nnz / (double)(A).nrows : 0.0;                                            \
        const bool two_ = (s_ncol % 2) == 0;                                                                                  \
        const unsigned gy_ = (unsigned)(two_ ? s_ncol / 2 : s_ncol);                                                          \
        if (avg_ <= 12.0) {                                                                                                   \
            const dim3 g_((unsigned)(((A).nrows + 255) / 256), gy_);                                                          \
            if (two_) hipLaunchKernelGGL((KERN<1, 2>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                             \
            else hipLaunchKernelGGL((KERN<1, 1>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                                  \
        } else if (avg_ <= g4_limit()) {                                                                                      \
            const dim3 g_((unsigned)(((A).nrows * 4 + 255) / 256), gy_);                                                      \
            if (two_) hipLaunchKernelGGL((KERN<4, 2>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                             \
            else hipLaunchKernelGGL((KERN<4, 1>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                                  \
        } else if (avg_ <= 96.0) {                                                                                            \
            const dim3 g_((unsigned)(((A).nrows * 8 + 255) / 256), gy_);                                                      \
            if (two_) hipLaunchKernelGGL((KERN<8, 2>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                             \
            else hipLaunchKernelGGL((KERN<8, 1>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                                  \
        } else {                                                                                                              \
            const dim3 g_((unsigned)(((A).nrows * 64 + 255) / 256), gy_);                                                     \
            if (two_) hipLaunchKernelGGL((KERN<64, 2>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                            \
            else hipLaunchKernelGGL((KERN<64, 1>), g_, dim3(256), 0, c->stream, __VA_ARGS__);                                 \
        }                                                                                                                     \
    } while (0)

template <int MODE> struct CsrKern {
    template <int G, int NC> static void launch(knp_ctx* c, const dim3& g, const CsrDev& A, const double* x, const double* b, double* y, hipStream_t st) {
        hipLaunchKernelGGL((k_csr<MODE, G, NC>), g, dim3(256), 0, st, A, x, b, y);
    }
};

template <int MODE> void launch_csr_on(knp_ctx* c, const CsrDev& A, const double* x, const double* b, double* y, hipStream_t st) {
    const double avg = A.nrows ? (double)A.nnz / (double)A.nrows : 0.0;
    const bool two = (s_ncol % 2) == 0;
    const unsigned gy = (unsigned)(two ? s_ncol / 2 : s_ncol);
    if (avg <= 12.0) {
        const dim3 g((unsigned)((A.nrows + 255) / 256), gy);
        if (two) CsrKern<MODE>::template launch<1, 2>(c, g, A, x, b, y, st); else CsrKern<MODE>::template launch<1, 1>(c, g, A, x, b, y, st);
    } else if (avg <= g4_limit()) {
        const dim3 g((unsigned)((A.nrows * 4 + 255) / 256), gy);
        if (two) CsrKern<MODE>::template launch<4, 2>(c, g, A, x, b, y, st); else CsrKern<MODE>::template launch<4, 1>(c, g, A, x, b, y, st);
    } else if (avg <= 96.0) {
        const dim3 g((unsigned)((A.nrows * 8 + 255) / 256), gy);
        if (two) CsrKern<MODE>::template launch<8, 2>(c, g, A, x, b, y, st); else CsrKern<MODE>::template launch<8, 1>(c, g, A, x, b, y, st);
    } else {
        const dim3 g((unsigned)((A.nrows * 64 + 255) / 256), gy);
        if (two) CsrKern<MODE>::template launch<64, 2>(c, g, A, x, b, y, st); else CsrKern<MODE>::template launch<64, 1>(c, g, A, x, b, y, st);
    }
}

template <int MODE> void launch_csr(knp_ctx* c, const CsrDev& A, const double* x, const double* b, double* y) {
    launch_csr_on<MODE>(c, A, x, b, y, c->stream);
}

// first Chebyshev update:  d = dinv r / theta ;  x = d (zero guess) or x += d
__global__ void k_cheb_first(int64_t n, int nil, const double* __restrict__ dinv, const double* __restrict__ b, double inv_theta,
                             double* __restrict__ r, double* __restrict__ d, double* __restrict__ x) {
    // zero initial guess: r = b ; d = dinv r / theta ; x = d.   grid.y = column group, nil interleaved columns per row
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nil) return;
    const int64_t o = (int64_t)blockIdx.y * n * nil;
    b += o; r += o; d += o; x += o;
    const double bi = b[i];
    const double v = dinv[i / nil] * bi * inv_theta;
    r[i] = bi;
    d[i] = v;
    x[i] = v;
}

// non-zero guess, fused:  r = b - A x ;  d = dinv r / theta ;  xout = x + d     (xout != x: neighbours still read x)
template <int G, int NC>
__global__ __launch_bounds__(256) void k_cheb_first_res(CsrDev A, const double* __restrict__ dinv, const double* __restrict__ b,
                                                        const double* __restrict__ x, double inv_theta, double* __restrict__ r,
                                                        double* __restrict__ d, double* __restrict__ xout) {
    const int64_t o = (int64_t)blockIdx.y * NC * A.nrows;
    b += o; x += o; r += o; d += o; xout += o;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int lane = threadIdx.x % G;
    double s[NC];
    row_dot<G, NC>(A, row, lane, x, s);
    if (row < A.nrows && lane == 0) {
        const double di = dinv[row];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int64_t q = row * NC + j;
            const double rn = b[q] - s[j];
            const double v = di * rn * inv_theta;
            r[q] = rn;
            d[q] = v;
            xout[q] = x[q] + v;
        }
    }
}

// fused step:  r -= A d_in ;  d_out = c1 d_in + c2 dinv r ;  x += d_out        (G lanes per row)
template <int G, int NC>
__global__ __launch_bounds__(256) void k_cheb_step(CsrDev A, const double* __restrict__ dinv, const double* __restrict__ din,
                                                   double c1, double c2, double* __restrict__ r, double* __restrict__ dout,
                                                   double* __restrict__ x) {
    const int64_t o = (int64_t)blockIdx.y * NC * A.nrows;
    din += o; r += o; dout += o; x += o;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int lane = threadIdx.x % G;
    double s[NC];
    row_dot<G, NC>(A, row, lane, din, s);
    if (row < A.nrows && lane == 0) {
        const double di = dinv[row];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int64_t q = row * NC + j;
            const double rn = r[q] - s[j];
            const double dn = fma(c1, din[q], c2 * di * rn);
            r[q] = rn;
            dout[q] = dn;
            x[q] += dn;
        }
    }
}

void launch_cheb_step(knp_ctx* c, const CsrDev& A, const double* dinv, const double* din, double c1, double c2, double* r,
                      double* dout, double* x) {
    LAUNCH_BY_DENSITY(k_cheb_step, A, A, dinv, din, c1, c2, r, dout, x);
}

// dense y = M b on the coarsest level (n up to a few thousand): one workgroup per row, 4 independent loads in flight per
// lane; M is the pseudo-inverse stored in fp32 (preconditioner data: half the bytes, still an exactly symmetric
// operator because symmetric entries round identically), accumulation in fp64, fixed reduction tree.
constexpr int DENSE_ROWS = 2;
template <int NC>
__global__ __launch_bounds__(256) void k_dense_mv(int n, const float* __restrict__ M, const double* __restrict__ b,
                                                  double* __restrict__ y) {
    // DENSE_ROWS rows per workgroup (they share the loads of b: with one row per workgroup every workgroup pulls the whole of b through
    // its L1 -- 151 MB for two columns against 38 MB of matrix at n = 3 089), 4 independent row segments in flight per lane and row;
    // lanes read consecutive entries (a 16-byte-per-lane layout touches four times the cache lines per load of b and is slower);
    // NC right-hand-side columns share the pass over the fp32 rows (further columns along grid.y)
    __shared__ double part[4][DENSE_ROWS][NC];
    const int row0 = blockIdx.x * DENSE_ROWS;
    b += (int64_t)blockIdx.y * NC * n;
    y += (int64_t)blockIdx.y * NC * n;
    const float* __restrict__ Mr[DENSE_ROWS];
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; ++r) Mr[r] = M + (int64_t)(row0 + r < n ? row0 + r : n - 1) * n;      // (a duplicate row is not stored)
    double s[DENSE_ROWS][4][NC];
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NC; ++j) s[r][u][j] = 0.0;
    int k = threadIdx.x;
    for (; k + 768 < n; k += 1024) {
        float m[DENSE_ROWS][4];
#pragma unroll
        for (int r = 0; r < DENSE_ROWS; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) m[r][u] = Mr[r][k + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double bv = b[(int64_t)(k + 256 * u) * NC + j];
#pragma unroll
                for (int r = 0; r < DENSE_ROWS; ++r) s[r][u][j] = fma((double)m[r][u], bv, s[r][u][j]);
            }
    }
    for (; k < n; k += 256) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double bv = b[(int64_t)k * NC + j];
#pragma unroll
            for (int r = 0; r < DENSE_ROWS; ++r) s[r][0][j] = fma((double)Mr[r][k], bv, s[r][0][j]);
        }
    }
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            double v = (s[r][0][j] + s[r][1][j]) + (s[r][2][j] + s[r][3][j]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][r][j] = v;
        }
    __syncthreads();
    if ((int)threadIdx.x < DENSE_ROWS * NC) {
        const int r = threadIdx.x / NC, j = threadIdx.x % NC;
        if (row0 + r < n) y[(int64_t)(row0 + r) * NC + j] = (part[0][r][j] + part[1][r][j]) + (part[2][r][j] + part[3][r][j]);
    }
}

// rc[v] = sum over the DG dofs mapped to conforming dof v (CSR list, fixed order -> deterministic)
// G lanes per conforming dof (a vertex is shared by ~24 tets); fixed summation tree -> deterministic
template <int G>
__global__ __launch_bounds__(256) void k_dg_restrict(int64_t ncg, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                     const double* __restrict__ r, double* __restrict__ rc, int64_t r_stride, int nil,
                                                     const double* __restrict__ t, double ct) {
    r += (int64_t)blockIdx.y * r_stride;
    if (t) t += (int64_t)blockIdx.y * r_stride;
    rc += (int64_t)(blockIdx.y / nil) * nil * ncg + (blockIdx.y % nil);          // column blockIdx.y of the interleaved level vector
    const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int lane = threadIdx.x % G;
    double s = 0.0;
    if (v < ncg) {
        const int e = ptr[v + 1];
        for (int k = ptr[v] + lane; k < e; k += G) s += t ? fma(-ct, t[idx[k]], r[idx[k]]) : r[idx[k]];
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) s += __shfl_down(s, off, G);
    if (v < ncg && lane == 0) rc[v * nil] = s;
}

// Tile-wise restriction, stage 1: the tile's DG values (consecutive cells: one coalesced read of r) go to LDS; every slot (= one
// conforming dof touched by the tile) is summed by one thread in a fixed order.  The gather form above reads 8 bytes out of every
// 32-byte cell record four times over (once per vertex): 38 us at r=2 against 64 MB of input.
__global__ __launch_bounds__(256) void k_restrict_tiles(int64_t ndof_owned, int tile_dofs, const int32_t* __restrict__ tile_off,
                                                        const int32_t* __restrict__ slot_ptr, const uint16_t* __restrict__ slot_idx,
                                                        const double* __restrict__ r, int64_t r_stride, double* __restrict__ part,
                                                        int64_t nslots, const double* __restrict__ t, double ct) {
    // t != null: the restricted vector is r - ct t (the residual left by the first step of the DG-level Chebyshev smoother, whose
    // operator product t = A Binv r exists anyway: krylov.hip, hybrid two-level preconditioner)
    extern __shared__ double s_r[];
    r += (int64_t)blockIdx.y * r_stride;
    if (t) t += (int64_t)blockIdx.y * r_stride;
    part += (int64_t)blockIdx.y * nslots;
    const int64_t d0 = (int64_t)blockIdx.x * tile_dofs;
    const int n = (int)((ndof_owned - d0 < tile_dofs) ? (ndof_owned - d0) : tile_dofs);
    const double* src = r + d0;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {                        // 16-byte aligned column: pairs (tile_dofs is even)
        const double2* src2 = reinterpret_cast<const double2*>(src);
        if (t) {
            const double2* t2 = reinterpret_cast<const double2*>(t + d0);
            for (int i = threadIdx.x; 2 * i + 1 < n; i += 256) {
                const double2 v = src2[i], w = t2[i];
                s_r[2 * i] = fma(-ct, w.x, v.x);
                s_r[2 * i + 1] = fma(-ct, w.y, v.y);
            }
            if ((n & 1) && threadIdx.x == 0) s_r[n - 1] = fma(-ct, t[d0 + n - 1], src[n - 1]);
        } else {
            for (int i = threadIdx.x; 2 * i + 1 < n; i += 256) {
                const double2 v = src2[i];
                s_r[2 * i] = v.x;
                s_r[2 * i + 1] = v.y;
            }
            if ((n & 1) && threadIdx.x == 0) s_r[n - 1] = src[n - 1];
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) s_r[i] = t ? fma(-ct, t[d0 + i], src[i]) : src[i];
    }
    __syncthreads();
    const int p1 = tile_off[blockIdx.x + 1];
    for (int p = tile_off[blockIdx.x] + threadIdx.x; p < p1; p += 256) {
        double acc = 0.0;
        const int e = slot_ptr[p + 1];
        for (int k = slot_ptr[p]; k < e; ++k) acc += s_r[slot_idx[k]];
        part[p] = acc;
    }
}

// stage 2: rc[v] = sum of the slots of conforming dof v (fixed order); with dinv also the zero-guess first Chebyshev update of level 0
// (r = rc ; d = x = dinv rc / theta: one launch less per V-cycle; AmgHierarchy::fuse_first0)
__global__ __launch_bounds__(256) void k_restrict_sum(int64_t ncg, const int32_t* __restrict__ part_ptr, const int32_t* __restrict__ part_idx,
                                                      const double* __restrict__ part, int64_t nslots, double* __restrict__ rc, int nil,
                                                      const double* __restrict__ dinv, double inv_theta, double* __restrict__ r0,
                                                      double* __restrict__ d0, double* __restrict__ x0) {
    part += (int64_t)blockIdx.y * nslots;
    const int64_t o = (int64_t)(blockIdx.y / nil) * nil * ncg + (blockIdx.y % nil);   // column blockIdx.y of the interleaved level vector
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= ncg) return;
    double s = 0.0;
    const int e = part_ptr[v + 1];
    for (int k = part_ptr[v]; k < e; ++k) s += part[part_idx[k]];
    rc[o + v * nil] = s;
    if (dinv) {
        const double w = dinv[v] * s * inv_theta;
        r0[o + v * nil] = s;
        d0[o + v * nil] = w;
        x0[o + v * nil] = w;
    }
}

}  // namespace

// smoother on level lv: Chebyshev polynomial of D^-1 A on [lower*rho, rho]
static double level_theta(const AmgLevel& L) { return 0.5 * (L.rho + L.cheb_lower * L.rho); }

// first_done: the zero-guess first update was written by the kernel that produced L.b (k_csr_first)
static void smooth(knp_ctx* c, AmgLevel& L, bool zero_guess, bool first_done = false) {
    const double lmax = L.rho, lmin = L.cheb_lower * L.rho;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    const int nil = (s_ncol % 2 == 0) ? 2 : 1;
    if (zero_guess && first_done) {
    } else if (zero_guess) {
        hipLaunchKernelGGL(k_cheb_first, dim3((unsigned)((L.n * nil + 255) / 256), (unsigned)(s_ncol / nil)), dim3(256), 0, c->stream, L.n, nil,
                           L.dinv, L.b, 1.0 / theta, L.r, L.d0, L.x);
    } else {
        // x lives in L.x; the fused kernel writes the updated iterate to L.d1 (free at this point), then swap
        LAUNCH_BY_DENSITY(k_cheb_first_res, L.A, L.A, L.dinv, L.b, L.x, 1.0 / theta, L.r, L.d0, L.d1);
        double* t = L.x; L.x = L.d1; L.d1 = t;
    }
    double* din = L.d0;
    double* dout = L.d1;
    for (int k = 1; k < L.cheb_degree; ++k) {
        const double rho_new = 1.0 / (2.0 * sigma - rho);
        launch_cheb_step(c, L.A, L.dinv, din, rho_new * rho, 2.0 * rho_new / delta, L.r, dout, L.x);
        rho = rho_new;
        double* t = din; din = dout; dout = t;
    }
}

// x_0 = V(b_0) ; level vectors b/x of level 0 are filled / read by the caller
static int amg_vcycle_eager(knp_ctx* c, AmgHierarchy& H, int l0 = 0) {
    const int nl = (int)H.levels.size();
    s_ncol = H.ncol;
    struct Reset { ~Reset() { s_ncol = 1; } } reset_on_exit;
    bool first_done = l0 == 0 && H.fuse_first0;                     // level 0's first update came with the restriction (k_restrict_sum)
    for (int l = l0; l < nl - 1; ++l) {
        AmgLevel& L = H.levels[l];
        if (L.cheb_degree == 0) {                                    // transfer-only level: x = 0, r = b
            // level 0: amg_restrict_from_dg already went down to level 1 (outside the captured graph, see there)
            if (l > 0) launch_csr<0>(c, L.R, L.b, nullptr, H.levels[l + 1].b);
            first_done = false;
            continue;
        }
        smooth(c, L, true, first_done);
        launch_csr<1>(c, L.A, L.x, L.b, L.r);                        // r = b - A x
        AmgLevel& N = H.levels[l + 1];
        first_done = l + 1 < nl - 1 && N.cheb_degree > 0;
        if (first_done)                                              // b_{l+1} = R r and the next level's first update at once
            LAUNCH_BY_DENSITY(k_csr_first, L.R, L.R, L.r, N.dinv, 1.0 / level_theta(N), N.b, N.r, N.d0, N.x);
        else
            launch_csr<0>(c, L.R, L.r, nullptr, N.b);                // b_{l+1} = R r
    }
    AmgLevel& C = H.levels[nl - 1];
    if (H.ncol % 2 == 0)
        hipLaunchKernelGGL(k_dense_mv<2>, dim3((unsigned)((C.n + DENSE_ROWS - 1) / DENSE_ROWS), (unsigned)(H.ncol / 2)), dim3(256), 0, c->stream, (int)C.n, (const float*)H.pinv, C.b, C.x);
    else
        hipLaunchKernelGGL(k_dense_mv<1>, dim3((unsigned)((C.n + DENSE_ROWS - 1) / DENSE_ROWS), (unsigned)H.ncol), dim3(256), 0, c->stream, (int)C.n, (const float*)H.pinv, C.b, C.x);
    for (int l = nl - 2; l >= l0; --l) {
        AmgLevel& L = H.levels[l];
        if (L.cheb_degree == 0) {
            launch_csr<0>(c, L.P, H.levels[l + 1].x, nullptr, L.x);  // x = P x_{l+1}
            continue;
        }
        launch_csr<2>(c, L.P, H.levels[l + 1].x, nullptr, L.x);      // x += P x_{l+1}
        smooth(c, L, false);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// The V-cycle is ~40 tiny launch-bound kernels on fixed buffers: capture it once into a hipGraph and replay it
// (kernel boundaries ~1.5 us instead of ~5 us of eager launch latency each).
static int amg_vcycle_levels(knp_ctx* c, AmgHierarchy& H, int l0, hipStream_t on_stream) {
    static const bool use_graph = !(getenv("KNP_NO_GRAPH") && atoi(getenv("KNP_NO_GRAPH")));
    if (use_graph && !H.graph_tried) {
        H.graph_tried = true;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            H.entry_x = H.levels[0].x; H.entry_r = H.levels[0].r; H.entry_d0 = H.levels[0].d0;
            std::vector<std::pair<double*, double*>> keep;           // the smoother swaps (x, d1) of a level as it goes
            for (auto& L : H.levels) keep.emplace_back(L.x, L.d1);
            const int rc = amg_vcycle_eager(c, H, l0);
            const hipError_t e = hipStreamEndCapture(c->stream, &graph);
            if (rc == 0 && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess)
                H.graph_exec = exec;
            else                                                     // nothing ran: the eager V-cycle below starts from the buffers the
                for (size_t l = 0; l < H.levels.size(); ++l) { H.levels[l].x = keep[l].first; H.levels[l].d1 = keep[l].second; }   // restriction wrote
            if (graph) hipGraphDestroy(graph);
            if (getenv("KNP_DEBUG")) fprintf(stderr, "[knp] V-cycle graph capture: rc=%d end=%d exec=%p\n", rc, (int)e, (void*)H.graph_exec);
        } else if (getenv("KNP_DEBUG")) {
            fprintf(stderr, "[knp] hipStreamBeginCapture failed\n");
        }
        (void)hipGetLastError();
    }
    if (H.graph_exec) {
        HIPCHK(c, hipGraphLaunch((hipGraphExec_t)H.graph_exec, on_stream ? on_stream : c->stream));
        return 0;
    }
    return amg_vcycle_eager(c, H, l0);  // eager fallback always runs on the context's stream
}

// ---- row-distributed level 0 (partitioned runs) ---------------------------------------------------------------------------------------
// Vectors on level 0 come in two kinds (the usual bookkeeping of non-overlapping domain decomposition): ACCUMULATED -- every rank that
// has a shared dof holds its full value (x, the smoother's d, the accumulated residual) -- and DISTRIBUTED -- every rank holds the
// part its own cells / facets contributed (the restricted DG residual b, and every product with the sub-assembled matrix).  A product
// needs no communication; turning its result into an accumulated vector is one interface exchange (interface_accumulate); the
// restriction to the replicated level 1 is a per-rank partial sum behind one all-reduce of n_1 values (29 k at r=2, 6.5x fewer than the
// level-0 vector the replicated design all-reduced).  Arithmetic per dof is that of smooth() / amg_vcycle_eager on the global level.
__global__ void k_l0_first_nz(int64_t n, int nil, const double* __restrict__ dinv, const double* __restrict__ r, double inv_theta,
                              double* __restrict__ d, double* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nil) return;
    const int64_t o = (int64_t)blockIdx.y * n * nil;
    const double v = dinv[i / nil] * r[o + i] * inv_theta;
    d[o + i] = v;
    x[o + i] += v;
}

// r -= t (t = accumulated A d) ;  d = c1 d + c2 dinv r ;  x += d
__global__ void k_l0_step(int64_t n, int nil, const double* __restrict__ dinv, const double* __restrict__ t, double c1, double c2,
                          double* __restrict__ r, double* __restrict__ d, double* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nil) return;
    const int64_t o = (int64_t)blockIdx.y * n * nil;
    const double rn = r[o + i] - t[o + i];
    const double dn = fma(c1, d[o + i], c2 * dinv[i / nil] * rn);
    r[o + i] = rn;
    d[o + i] = dn;
    x[o + i] += dn;
}

static int dist0_smooth(knp_ctx* c, AmgHierarchy& H, bool zero_guess) {
    AmgLevel& L = H.levels[0];
    const int nil = (s_ncol % 2 == 0) ? 2 : 1;
    const dim3 g((unsigned)((L.n * nil + 255) / 256), (unsigned)(s_ncol / nil));
    const double lmax = L.rho, lmin = L.cheb_lower * L.rho;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    int rc;
    if (zero_guess) {                                                // r = acc(b) ; d = x = dinv r / theta
        HIPCHK(c, hipMemcpyAsync(H.t0, L.b, sizeof(double) * (size_t)L.n * s_ncol, hipMemcpyDeviceToDevice, c->stream));
        if ((rc = interface_accumulate(c, H.t0, L.n, s_ncol))) return rc;
        hipLaunchKernelGGL(k_cheb_first, g, dim3(256), 0, c->stream, L.n, nil, L.dinv, (const double*)H.t0, 1.0 / theta, L.r, L.d0, L.x);
    } else {                                                         // r = acc(b - A x) ; d = dinv r / theta ; x += d
        launch_csr<1>(c, L.A, L.x, L.b, L.r);
        if ((rc = interface_accumulate(c, L.r, L.n, s_ncol))) return rc;
        hipLaunchKernelGGL(k_l0_first_nz, g, dim3(256), 0, c->stream, L.n, nil, L.dinv, (const double*)L.r, 1.0 / theta, L.d0, L.x);
    }
    for (int k = 1; k < L.cheb_degree; ++k) {
        const double rho_new = 1.0 / (2.0 * sigma - rho);
        launch_csr<0>(c, L.A, L.d0, nullptr, H.t0);
        if ((rc = interface_accumulate(c, H.t0, L.n, s_ncol))) return rc;
        hipLaunchKernelGGL(k_l0_step, g, dim3(256), 0, c->stream, L.n, nil, L.dinv, (const double*)H.t0, rho_new * rho, 2.0 * rho_new / delta,
                           L.r, L.d0, L.x);
        rho = rho_new;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int amg_vcycle_dist0(knp_ctx* c, AmgHierarchy& H) {
    if (H.levels.size() < 2) { c->err = "amg: a distributed level 0 needs a coarser level"; return -1; }
    s_ncol = H.ncol;
    struct Reset { ~Reset() { s_ncol = 1; } } reset_on_exit;
    AmgLevel& L = H.levels[0];
    AmgLevel& N = H.levels[1];
    if (!H.t0) HIPCHK(c, hipMalloc((void**)&H.t0, sizeof(double) * (size_t)(L.n ? L.n : 1) * (size_t)H.ncol));
    int rc;
    if (L.cheb_degree > 0) {
        if ((rc = dist0_smooth(c, H, true))) return rc;
        launch_csr<1>(c, L.A, L.x, L.b, L.r);                        // distributed residual b - A x
        launch_csr<0>(c, L.R, L.r, nullptr, N.b);                    // this rank's part of the level-1 right-hand side
        if ((rc = allreduce_red(c, N.b, (int)(N.n * H.ncol)))) return rc;
    }                                                                // (transfer-only: amg_restrict_tail went down to level 1 already)
    if ((rc = amg_vcycle_levels(c, H, 1, nullptr))) return rc;
    s_ncol = H.ncol;
    if (L.cheb_degree == 0) {
        launch_csr<0>(c, L.P, N.x, nullptr, L.x);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    launch_csr<2>(c, L.P, N.x, nullptr, L.x);
    return dist0_smooth(c, H, false);
}

int amg_vcycle(knp_ctx* c, AmgHierarchy& H, hipStream_t on_stream) {
    if (!H.dist0) return amg_vcycle_levels(c, H, 0, on_stream);
    if (on_stream && on_stream != c->stream) { c->err = "amg: the distributed level 0 runs on the context's stream"; return -1; }
    return amg_vcycle_dist0(c, H);
}

int amg_restrict_tail(knp_ctx* c, AmgHierarchy& H, hipStream_t st);

int amg_restrict_from_dg(knp_ctx* c, AmgHierarchy& H, const double* r_dg, hipStream_t on_stream, int64_t r_stride, const double* t_dg,
                         double ct) {
    if (on_stream && c->dist) { c->err = "amg: the all-reduced restriction runs on the context's stream"; return -1; }
    s_ncol = H.ncol;
    struct Reset { ~Reset() { s_ncol = 1; } } reset_on_exit;
    hipStream_t st = on_stream ? on_stream : c->stream;
    if (H.ntiles > 0) {
        const int tile_dofs = H.tile_cells * c->nd;
        int rc = amg_restrict_tiles_prepare(c, H);
        if (rc) return rc;
        hipLaunchKernelGGL(k_restrict_tiles, dim3((unsigned)H.ntiles, (unsigned)H.ncol), dim3(256), sizeof(double) * tile_dofs, st,
                           c->m.nc_owned * c->nd, tile_dofs, H.tile_off, H.slot_ptr, H.slot_idx, r_dg, r_stride, H.part, H.nslots, t_dg, ct);
        return amg_restrict_finish(c, H, on_stream);
    }
    hipLaunchKernelGGL(k_dg_restrict<8>, GRIDX((H.ncg * 8 + 255) / 256), dim3(256), 0, st, H.ncg, H.cg_ptr, H.cg_idx, r_dg, H.levels[0].b, r_stride, (H.ncol % 2 == 0) ? 2 : 1, t_dg, ct);
    return amg_restrict_tail(c, H, st);
}

// the per-tile partial sums buffer [ncol][nslots] of the tile-wise restriction, sized at the first use
int amg_restrict_tiles_prepare(knp_ctx* c, AmgHierarchy& H) {
    if (H.ntiles <= 0) return -1;
    if (!H.part || H.part_cols != H.ncol) {
        hipFree(H.part);
        H.part = nullptr;
        HIPCHK(c, hipMalloc((void**)&H.part, sizeof(double) * (size_t)H.ncol * (size_t)(H.nslots ? H.nslots : 1)));
        H.part_cols = H.ncol;
    }
    return 0;
}

// everything behind stage 1 of the tile-wise restriction (H.part filled by k_restrict_tiles or by the fused Chebyshev / restriction
// kernel of krylov.hip): stage 2, the transfer-only finest level, the all-reduce of a partitioned run
int amg_restrict_finish(knp_ctx* c, AmgHierarchy& H, hipStream_t on_stream) {
    if (on_stream && c->dist) { c->err = "amg: the all-reduced restriction runs on the context's stream"; return -1; }
    s_ncol = H.ncol;
    struct Reset { ~Reset() { s_ncol = 1; } } reset_on_exit;
    hipStream_t st = on_stream ? on_stream : c->stream;
    AmgLevel& L0 = H.levels[0];
    const bool fuse = H.fuse_first0;                                // (amg_vcycle_eager then skips level 0's k_cheb_first)
    const bool baked = H.graph_exec != nullptr;                     // the captured V-cycle reads the buffers of its capture
    hipLaunchKernelGGL(k_restrict_sum, GRIDX((H.ncg + 255) / 256), dim3(256), 0, st, H.ncg, H.part_ptr, H.part_idx,
                       (const double*)H.part, H.nslots, L0.b, (H.ncol % 2 == 0) ? 2 : 1, fuse ? (const double*)L0.dinv : (const double*)nullptr,
                       1.0 / level_theta(L0), baked ? H.entry_r : L0.r, baked ? H.entry_d0 : L0.d0, baked ? H.entry_x : L0.x);
    return amg_restrict_tail(c, H, st);
}

static int amg_restrict_tail_impl(knp_ctx* c, AmgHierarchy& H, hipStream_t st) {
    // multi-GPU: the conforming hierarchy is replicated on every rank; the restricted residual is the sum of the
    // ranks' owned-cell contributions (one all-reduce), after which every rank runs the same V-cycle.  When the finest
    // level is transfer-only (EMI) the restriction to level 1 is linear in b, so it is applied to the LOCAL vector first
    // and the all-reduce moves the 6.5x shorter level-1 vector (29 k instead of 188 k doubles at r=2).
    if (H.levels.size() > 1 && H.levels[0].cheb_degree == 0) {
        AmgLevel& L = H.levels[0];
        launch_csr_on<0>(c, L.R, L.b, nullptr, H.levels[1].b, st);
        if (c->dist) return allreduce_red(c, H.levels[1].b, (int)(H.levels[1].n * H.ncol));
        return 0;
    }
    if (c->dist && !H.dist0) return allreduce_red(c, H.levels[0].b, (int)(H.ncg * H.ncol));
    return 0;                                                       // dist0: b stays the rank's partial sums (amg_vcycle_dist0)
}
int amg_restrict_tail(knp_ctx* c, AmgHierarchy& H, hipStream_t st) {
    const int keep = s_ncol;
    s_ncol = H.ncol;
    const int rc = amg_restrict_tail_impl(c, H, st);
    s_ncol = keep;
    return rc;
}

void amg_free(AmgHierarchy& H) {
    for (auto& L : H.levels) {
        free_csr(L.A); free_csr(L.P); free_csr(L.R);
        hipFree(L.dinv); hipFree(L.x); hipFree(L.b); hipFree(L.r); hipFree(L.d0); hipFree(L.d1);
    }
    H.levels.clear();
    if (H.graph_exec) hipGraphExecDestroy((hipGraphExec_t)H.graph_exec);
    H.graph_exec = nullptr;
    H.graph_tried = false;
    H.ncol = 1;
    hipFree(H.pinv); hipFree(H.dg2cg); hipFree(H.cg_ptr); hipFree(H.cg_idx);
    H.pinv = nullptr; H.dg2cg = nullptr; H.cg_ptr = nullptr; H.cg_idx = nullptr;
    hipFree(H.t0); H.t0 = nullptr; H.dist0 = false;
    H.fuse_first0 = false; H.entry_x = H.entry_r = H.entry_d0 = nullptr;
    hipFree(H.tile_off); hipFree(H.slot_ptr); hipFree(H.slot_idx); hipFree(H.part_ptr); hipFree(H.part_idx); hipFree(H.part);
    H.tile_off = H.slot_ptr = H.part_ptr = H.part_idx = nullptr; H.slot_idx = nullptr; H.part = nullptr;
    H.ntiles = H.nslots = 0; H.part_cols = 0;
    H.ready = false;
}

AmgHierarchy* amg_slot(knp_ctx* c, int which);   // abi.hip

extern "C" {

int knp_amg_begin(knp_ctx* c, int which, int64_t ncg, const int32_t* dg2cg, const int32_t* cg_ptr, const int32_t* cg_idx) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H) { c->err = "amg: bad slot"; return -1; }
    amg_free(*H);
    const int64_t ndof = c->m.nc * c->nd;
    const int64_t nmap = ndof;                              // dg2cg: conforming dof of every DG dof (injection)
    for (int64_t i = 0; i < nmap; ++i)
        if (dg2cg[i] < 0 || dg2cg[i] >= ncg) { c->err = "amg: dg2cg out of range"; return -1; }
    // cg_ptr / cg_idx = null: the inverse map (conforming dof -> the OWNED DG dofs that inject into it, ascending) is derived here by a
    // stable counting sort instead of the caller's argsort of nc nd keys
    std::vector<int32_t> own_ptr, own_idx;
    if (!cg_ptr || !cg_idx) {
        const int64_t nown = c->m.nc_owned * c->nd;
        own_ptr.assign((size_t)ncg + 1, 0);
        for (int64_t i = 0; i < nown; ++i) ++own_ptr[(size_t)dg2cg[i] + 1];
        for (int64_t v = 0; v < ncg; ++v) own_ptr[(size_t)v + 1] += own_ptr[(size_t)v];
        own_idx.resize((size_t)nown);
        std::vector<int32_t> fill(own_ptr.begin(), own_ptr.end() - 1);
        for (int64_t i = 0; i < nown; ++i) own_idx[(size_t)fill[(size_t)dg2cg[i]]++] = (int32_t)i;
        cg_ptr = own_ptr.data();
        cg_idx = own_idx.data();
    }
    if (cg_ptr[ncg] > ndof) { c->err = "amg: cg_ptr inconsistent"; return -1; }
    for (int64_t k = 0; k < cg_ptr[ncg]; ++k)
        if (cg_idx[k] < 0 || cg_idx[k] >= ndof) { c->err = "amg: cg_idx out of range"; return -1; }
    H->ncg = ncg;
    int rc = up(c, &H->dg2cg, dg2cg, (size_t)nmap);
    rc |= up(c, &H->cg_ptr, cg_ptr, (size_t)ncg + 1);
    rc |= up(c, &H->cg_idx, cg_idx, (size_t)cg_ptr[ncg]);
    // tables of the tile-wise restriction over the owned cells (device order = Morton: a tile is a compact patch)
    const int nd = c->nd;
    static const int tile_env = getenv("KNP_RESTRICT_TILE") ? atoi(getenv("KNP_RESTRICT_TILE")) : 0;
    const int tile_cells = tile_env > 0 ? tile_env : (nd <= 4 ? 512 : 256);      // 16 KB / 20 KB of LDS per workgroup (r=2: 512 -> 10.55, 1024 -> 10.71, 256 -> 10.69 ms/step)
    const int64_t n_own = c->m.nc_owned;
    const int64_t ntiles = (n_own + tile_cells - 1) / tile_cells;
    if (ntiles > 0 && (int64_t)tile_cells * nd <= 65536 && ((int64_t)tile_cells * nd) % 2 == 0) {
        std::vector<int32_t> tile_off((size_t)ntiles + 1, 0), slot_ptr(1, 0), slot_cg;
        std::vector<uint16_t> slot_idx;
        slot_idx.reserve((size_t)n_own * nd);
        // the per-tile sorts (2 048 keys each, ~2 000 tiles at 10^6 cells) run on a few host threads; the lists are stitched in tile order
        std::vector<std::vector<std::pair<int32_t, uint16_t>>> sorted((size_t)ntiles);
        {
            const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, ntiles / 64));
            std::vector<std::thread> pool;
            for (int w = 0; w < T; ++w)
                pool.emplace_back([&, w]() {
                    for (int64_t t = w; t < ntiles; t += T) {
                        const int64_t c0 = t * tile_cells, c1 = std::min<int64_t>(n_own, c0 + tile_cells);
                        auto& v = sorted[(size_t)t];
                        v.reserve((size_t)(c1 - c0) * nd);
                        for (int64_t i = c0 * nd; i < c1 * nd; ++i) v.emplace_back(dg2cg[i], (uint16_t)(i - c0 * nd));
                        std::sort(v.begin(), v.end());
                    }
                });
            for (auto& th : pool) th.join();
        }
        for (int64_t t = 0; t < ntiles; ++t) {
            const auto& tmp = sorted[(size_t)t];
            for (size_t k = 0; k < tmp.size(); ++k) {
                if (k == 0 || tmp[k].first != tmp[k - 1].first) {
                    if (k) slot_ptr.push_back((int32_t)slot_idx.size());
                    slot_cg.push_back(tmp[k].first);
                }
                slot_idx.push_back(tmp[k].second);
            }
            if (!tmp.empty()) slot_ptr.push_back((int32_t)slot_idx.size());
            tile_off[t + 1] = (int32_t)slot_cg.size();
        }
        const int64_t nslots = (int64_t)slot_cg.size();
        std::vector<int32_t> part_ptr((size_t)ncg + 1, 0), part_idx((size_t)nslots);
        for (int64_t p = 0; p < nslots; ++p) ++part_ptr[slot_cg[p] + 1];
        for (int64_t v = 0; v < ncg; ++v) part_ptr[v + 1] += part_ptr[v];
        std::vector<int32_t> fill(part_ptr.begin(), part_ptr.end() - 1);
        for (int64_t p = 0; p < nslots; ++p) part_idx[fill[slot_cg[p]]++] = (int32_t)p;
        H->tile_cells = tile_cells; H->ntiles = ntiles; H->nslots = nslots;
        rc |= up(c, &H->tile_off, tile_off.data(), tile_off.size());
        rc |= up(c, &H->slot_ptr, slot_ptr.data(), slot_ptr.size());
        rc |= up(c, &H->slot_idx, slot_idx.data(), slot_idx.size());
        rc |= up(c, &H->part_ptr, part_ptr.data(), part_ptr.size());
        rc |= up(c, &H->part_idx, part_idx.data(), part_idx.size());
    }
    return rc;
}

int knp_amg_level(knp_ctx* c, int which, int64_t n, const int32_t* rpA, const int32_t* ciA, const double* vA, const double* dinv,
                  double rho, int cheb_degree, double cheb_lower, int64_t ncoarse, const int32_t* rpP, const int32_t* ciP,
                  const double* vP, const int32_t* rpR, const int32_t* ciR, const double* vR) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H) { c->err = "amg: bad slot"; return -1; }
    if (H->levels.empty() ? (n != H->ncg) : (n != H->levels.back().ncoarse)) { c->err = "amg: level size mismatch"; return -1; }
    for (int64_t k = 0; k < rpA[n]; ++k)
        if (ciA[k] < 0 || ciA[k] >= n) { c->err = "amg: A column out of range"; return -1; }
    AmgLevel L;
    L.n = n; L.ncoarse = ncoarse; L.rho = rho; L.cheb_degree = cheb_degree; L.cheb_lower = cheb_lower;
    int rc = up_csr(c, L.A, n, n, rpA, ciA, vA);
    rc |= up(c, &L.dinv, dinv, (size_t)n);
    if (ncoarse > 0) {
        for (int64_t k = 0; k < rpP[n]; ++k)
            if (ciP[k] < 0 || ciP[k] >= ncoarse) { c->err = "amg: P column out of range"; return -1; }
        for (int64_t k = 0; k < rpR[ncoarse]; ++k)
            if (ciR[k] < 0 || ciR[k] >= n) { c->err = "amg: R column out of range"; return -1; }
        rc |= up_csr(c, L.P, n, ncoarse, rpP, ciP, vP);
        rc |= up_csr(c, L.R, ncoarse, n, rpR, ciR, vR);
    }
    double** w[] = {&L.x, &L.b, &L.r, &L.d0, &L.d1};
    const size_t nbuf = (size_t)(n ? n : 1) * (size_t)H->ncol;          // [ncol / nil][n][nil], nil = 2 interleaved columns when ncol is even
    for (auto p : w) {
        if (hipMalloc((void**)p, nbuf * sizeof(double)) != hipSuccess) rc = -2;
        else hipMemset(*p, 0, nbuf * sizeof(double));
    }
    H->levels.push_back(L);
    return rc;
}

static int amg_finish_impl(knp_ctx* c, int which, int64_t n, const double* pinv64, const float* pinv32) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H || H->levels.empty() || H->levels.back().n != n || H->levels.back().ncoarse != 0) {
        c->err = "amg: finish does not match the last level"; return -1;
    }
    if (!pinv64 && !pinv32) { c->err = "amg: finish without a coarse inverse"; return -1; }
    int rc;
    if (pinv32) rc = up(c, &H->pinv, pinv32, (size_t)n * n);
    else {
        std::vector<float> p32((size_t)n * n);
        for (size_t i = 0; i < p32.size(); ++i) p32[i] = (float)pinv64[i];
        rc = up(c, &H->pinv, p32.data(), p32.size());
    }
    const bool fuse_env = !(getenv("KNP_FUSE_FIRST0") && atoi(getenv("KNP_FUSE_FIRST0")) == 0);   // read per upload: tests switch it
    H->fuse_first0 = fuse_env && !c->dist && H->ntiles > 0 && H->levels.size() > 1 && H->levels[0].cheb_degree > 0;
    H->ready = (rc == 0);
    return rc;
}

int knp_amg_finish(knp_ctx* c, int which, int64_t n, const double* pinv) { return amg_finish_impl(c, which, n, pinv, nullptr); }
// the same with the inverse already rounded to fp32 (the device stores it in fp32 either way)
int knp_amg_finish_f32(knp_ctx* c, int which, int64_t n, const float* pinv) { return amg_finish_impl(c, which, n, nullptr, pinv); }

// Marks hierarchy `which` as carrying a ROW-DISTRIBUTED level 0 (between knp_amg_begin and the first knp_amg_level): ncg, dg2cg and the
// level-0 matrices are this rank's rows in local numbering, A sub-assembled from its own cells / facets (knp_amg_interface first).
int knp_amg_dist0(knp_ctx* c, int which) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H || !H->levels.empty()) { c->err = "amg: dist0 must be set after begin and before the first level"; return -1; }
    if (!c->dist) { c->err = "amg: a distributed level 0 needs a communicator"; return -1; }
    H->dist0 = true;
    return 0;
}

int knp_amg_columns(knp_ctx* c, int which, int ncol) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H || !H->levels.empty()) { c->err = "amg: columns must be set after begin and before the first level"; return -1; }
    if (ncol < 1 || ncol > KNP_MAX_SYS || (which == 0 && ncol != 1)) { c->err = "amg: bad column count"; return -1; }
    H->ncol = ncol;
    return 0;
}

int knp_amg_clear(knp_ctx* c, int which) {
    if (!c) return -1;
    AmgHierarchy* H = amg_slot(c, which);
    if (!H) return -1;
    amg_free(*H);
    return 0;
}

}  // extern "C"
