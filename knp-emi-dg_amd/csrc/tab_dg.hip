// DG-p path for p >= 2 ("tabulated" path): assembled cell-block operators.
//
// For P1 the operators are applied matrix-free (apply_p1.hip: 137 / 217 B per cell and apply, HBM-bound).  For P2 the
// forms carry quadratic coefficients (kappa, grad phi) and need 11..36-point quadrature on every facet, so a
// matrix-free apply would repeat ~60 kflop per cell in every Krylov iteration.  Here the quadrature runs ONCE per
// time step (exactly when the reference re-assembles: solver.py:477-479, 730-731) into dense cell blocks
//     blk[c][0]   : (nd x nd) diagonal block of cell c,
//     blk[c][1+i] : (nd x nd) coupling to the neighbour behind local facet i,
// kept in HBM (4 KB per cell and operator in 3D: 0.5 GB for the 124 416-tet mesh, 4 GB at 10^6 tets -- this is
// what 288 GB of HBM3E is for), and an apply is a block-sparse gather  y_c = sum_j blk[c][j] x_{nbr_j(c)}.
// The reference basis is tabulated on the host (knpemidg/dgtab.py) at the quadrature points of each integral class
// and uploaded once (knp_set_tabulation), so the kernels are generic in the polynomial degree.
//
// Reference forms restated here: a_emi / L_emi  src/knpemidg/solver.py:270-403, a_knp / L_knp :534-663,
// step III :808-845, facet projector utils.py:100-124.
#include "../../include/knpemi_hip.h"
#include "knpemi_internal.hpp"
#include "cell_geom.hpp"
#include <cstdlib>

namespace {

struct IonZ { int n; double z[KNP_MAX_IONS]; };
static IonZ ion_z(const knp_ctx* c) {
    IonZ a; a.n = c->p.n_ions;
    for (int i = 0; i < KNP_MAX_IONS; ++i) a.z[i] = i < c->p.n_ions ? c->p.z[i] : 1.0;
    return a;
}

template <int D> __device__ __forceinline__ void cell_geo(const MeshDev& m, int64_t c, CellGeom<D>& K) {
    int v[D + 1];
    load_cell_ints<D>(m.cells, c, v);
    double X[D + 1][D];
#pragma unroll
    for (int a = 0; a <= D; ++a) load_vertex<D>(m.coords, v[a], X[a]);
    gradients<D>(X, K);
}

// facet i of cell K: gn[l] = grad lambda_l . n (n = outward unit normal = -g_i/|g_i|), returns the facet area
template <int D> __device__ __forceinline__ double facet_frame(const CellGeom<D>& K, int i, double* nrm, double* gn) {
    const double gi2 = dotD<D>(K.g[i], K.g[i]);
    const double gl = sqrt(gi2);
#pragma unroll
    for (int k = 0; k < D; ++k) nrm[k] = -K.g[i][k] / gl;
#pragma unroll
    for (int l = 0; l <= D; ++l) gn[l] = dotD<D>(K.g[l], nrm);
    return (double)D * K.vol * gl;
}

template <int NV> __device__ __forceinline__ double dotNV(const double* a, const double* b) {
    double s = a[0] * b[0];
#pragma unroll
    for (int l = 1; l < NV; ++l) s = fma(a[l], b[l], s);
    return s;
}

template <int ND> __device__ __forceinline__ void ld(const double* __restrict__ p, int64_t c, double* v) {
#pragma unroll
    for (int a = 0; a < ND; ++a) v[a] = p[c * ND + a];
}

// ------------------------------------------------------------------------------------------------------------
// nodal coefficient fields (exact: linear combinations of DG-p functions with cell-wise constant factors)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_tab_kappa(int64_t nc, int nd, const double* __restrict__ cc, const double* __restrict__ celim,
                            const double* __restrict__ Dk, IonZ ia, double F, double psi, double* __restrict__ kappa) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc * nd) return;
    const int64_t c = t / nd;
    double s = 0.0;
    for (int i = 0; i < ia.n; ++i) {
        const double v = (i < ia.n - 1) ? cc[(int64_t)i * nc * nd + t] : celim[t];
        s += F * ia.z[i] * ia.z[i] * psi * Dk[(int64_t)i * nc + c] * v;
    }
    kappa[t] = s;
}

__global__ void k_tab_celim(int64_t nc, int nd, const double* __restrict__ cc, const double* __restrict__ rho, IonZ ia,
                            double* __restrict__ celim) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc * nd) return;
    const double zN = ia.z[ia.n - 1];
    double acc = 0.0;
    for (int i = 0; i < ia.n - 1; ++i) acc += -(1.0 / zN) * ia.z[i] * cc[(int64_t)i * nc * nd + t];
    acc += -(1.0 / zN) * rho[t / nd];
    celim[t] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// a_emi blocks: one thread per (cell, row a)
// ------------------------------------------------------------------------------------------------------------
template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_assemble_emi(MeshDev m, TabRule rc, TabRule rf, TabRule rm,
                                                          const double* __restrict__ kappa, double tau, double C_phi,
                                                          double* __restrict__ blk) {
    constexpr int NV = D + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.nc_owned * ND) return;
    const int64_t c = t / ND;
    const int a = (int)(t % ND);
    CellGeom<D> K;
    cell_geo<D>(m, c, K);
    double kap[ND];
    ld<ND>(kappa, c, kap);
    double diag[ND];
#pragma unroll
    for (int b = 0; b < ND; ++b) diag[b] = 0.0;

    // cells: int kappa grad u . grad v
    for (int q = 0; q < rc.nq; ++q) {
        const double* B = rc.B + (int64_t)q * ND;
        const double* dB = rc.dB + (int64_t)q * ND * NV;
        double kq = 0.0;
#pragma unroll
        for (int b = 0; b < ND; ++b) kq = fma(B[b], kap[b], kq);
        double ga[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < NV; ++l) s = fma(dB[a * NV + l], K.g[l][k], s);
            ga[k] = s;
        }
        const double w = rc.w[q] * K.vol * kq;
#pragma unroll
        for (int b = 0; b < ND; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double gb = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) gb = fma(dB[b * NV + l], K.g[l][k], gb);
                s = fma(ga[k], gb, s);
            }
            diag[b] = fma(w, s, diag[b]);
        }
    }

    const uint32_t flags = m.fflag[c];
    const double hc = m.h[c];
    for (int i = 0; i < NV; ++i) {
        const uint32_t fb = (flags >> (8 * i)) & 0xffu;
        const uint32_t kind = (fb >> 2) & 3u;
        const int j = (int)(fb & 3u);
        const int64_t nb = m.nbr[c * NV + i];
        double off[ND];
#pragma unroll
        for (int b = 0; b < ND; ++b) off[b] = 0.0;
        if (nb >= 0 && kind == FK_SIPG) {
            CellGeom<D> K2;
            cell_geo<D>(m, nb, K2);
            double kap2[ND];
            ld<ND>(kappa, nb, kap2);
            double nrm[D], gn[NV], gn2[NV];
            const double area = facet_frame<D>(K, i, nrm, gn);
#pragma unroll
            for (int l = 0; l < NV; ++l) gn2[l] = dotD<D>(K2.g[l], nrm);
            const double pen0 = tau / (0.5 * (hc + m.h[nb]));
            for (int q = 0; q < rf.nq; ++q) {
                const double* Bi = rf.B + ((int64_t)i * rf.nq + q) * ND;
                const double* dBi = rf.dB + ((int64_t)i * rf.nq + q) * ND * NV;
                const double* Bj = rf.B + ((int64_t)j * rf.nq + q) * ND;
                const double* dBj = rf.dB + ((int64_t)j * rf.nq + q) * ND * NV;
                double kq = 0.0, kq2 = 0.0;
#pragma unroll
                for (int b = 0; b < ND; ++b) { kq = fma(Bi[b], kap[b], kq); kq2 = fma(Bj[b], kap2[b], kq2); }
                const double w = rf.w[q] * area;
                const double pen = pen0 * 0.5 * (kq + kq2);
                const double Ba = Bi[a];
                const double dna = dotNV<NV>(dBi + a * NV, gn);
#pragma unroll
                for (int b = 0; b < ND; ++b) {
                    const double Bb = Bi[b], B2b = Bj[b];
                    const double dnb = dotNV<NV>(dBi + b * NV, gn);
                    const double dn2b = dotNV<NV>(dBj + b * NV, gn2);
                    diag[b] += w * (-0.5 * kq * dnb * Ba - 0.5 * kq * dna * Bb + pen * Ba * Bb);
                    off[b] += w * (-0.5 * kq2 * dn2b * Ba + 0.5 * kq * dna * B2b - pen * Ba * B2b);
                }
            }
        } else if (nb >= 0 && kind == FK_MEMBRANE) {
            const double area = (double)D * K.vol * sqrt(dotD<D>(K.g[i], K.g[i]));
            for (int q = 0; q < rm.nq; ++q) {
                const double* Bi = rm.B + ((int64_t)i * rm.nq + q) * ND;
                const double* Bj = rm.B + ((int64_t)j * rm.nq + q) * ND;
                const double w = rm.w[q] * area * C_phi * Bi[a];
#pragma unroll
                for (int b = 0; b < ND; ++b) {
                    diag[b] = fma(w, Bi[b], diag[b]);
                    off[b] = fma(-w, Bj[b], off[b]);
                }
            }
        }
        double* o = blk + ((c * (NV + 1) + 1 + i) * ND + a) * ND;
#pragma unroll
        for (int b = 0; b < ND; ++b) o[b] = off[b];
    }
    double* o = blk + ((c * (NV + 1)) * ND + a) * ND;
#pragma unroll
    for (int b = 0; b < ND; ++b) o[b] = diag[b];
}

// ------------------------------------------------------------------------------------------------------------
// a_knp blocks of species s = blockIdx.y
// ------------------------------------------------------------------------------------------------------------
struct KnpTabArgs { double inv_dt, psi, tau; double z[KNP_MAX_SYS]; };

template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_assemble_knp(MeshDev m, TabRule rc, TabRule rf, const double* __restrict__ phi,
                                                          const double* __restrict__ Dk, KnpTabArgs ka,
                                                          double* __restrict__ blk_all) {
    constexpr int NV = D + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.nc_owned * ND) return;
    const int s = blockIdx.y;
    const int64_t c = t / ND;
    const int a = (int)(t % ND);
    double* blk = blk_all + (int64_t)s * m.nc * (NV + 1) * ND * ND;
    const double z = ka.z[s];
    const double Dc = Dk[(int64_t)s * m.nc + c];
    CellGeom<D> K;
    cell_geo<D>(m, c, K);
    double ph[ND];
    ld<ND>(phi, c, ph);
    double diag[ND];
#pragma unroll
    for (int b = 0; b < ND; ++b) diag[b] = 0.0;

    // cells: 1/dt u v + D grad u . grad v + z psi D u grad(phi) . grad v     (row v = a, column u = b)
    for (int q = 0; q < rc.nq; ++q) {
        const double* B = rc.B + (int64_t)q * ND;
        const double* dB = rc.dB + (int64_t)q * ND * NV;
        double ga[D], gphi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s1 = 0.0;
#pragma unroll
            for (int l = 0; l < NV; ++l) s1 = fma(dB[a * NV + l], K.g[l][k], s1);
            ga[k] = s1;
            gphi[k] = 0.0;
        }
#pragma unroll
        for (int b = 0; b < ND; ++b)
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double gb = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) gb = fma(dB[b * NV + l], K.g[l][k], gb);
                gphi[k] = fma(ph[b], gb, gphi[k]);
            }
        const double w = rc.w[q] * K.vol;
        const double drift = z * ka.psi * Dc * dotD<D>(gphi, ga);
#pragma unroll
        for (int b = 0; b < ND; ++b) {
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double gb = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) gb = fma(dB[b * NV + l], K.g[l][k], gb);
                s2 = fma(ga[k], gb, s2);
            }
            diag[b] += w * (ka.inv_dt * B[a] * B[b] + Dc * s2 + drift * B[b]);
        }
    }

    const uint32_t flags = m.fflag[c];
    const double hc = m.h[c];
    for (int i = 0; i < NV; ++i) {
        const uint32_t fb = (flags >> (8 * i)) & 0xffu;
        const uint32_t kind = (fb >> 2) & 3u;
        const int j = (int)(fb & 3u);
        const int64_t nb = m.nbr[c * NV + i];
        double off[ND];
#pragma unroll
        for (int b = 0; b < ND; ++b) off[b] = 0.0;
        if (nb >= 0 && kind == FK_SIPG) {
            CellGeom<D> K2;
            cell_geo<D>(m, nb, K2);
            double ph2[ND];
            ld<ND>(phi, nb, ph2);
            const double D2 = Dk[(int64_t)s * m.nc + nb];
            double nrm[D], gn[NV], gn2[NV];
            const double area = facet_frame<D>(K, i, nrm, gn);
#pragma unroll
            for (int l = 0; l < NV; ++l) gn2[l] = dotD<D>(K2.g[l], nrm);
            const double pen = ka.tau / (0.5 * (hc + m.h[nb]));
            for (int q = 0; q < rf.nq; ++q) {
                const double* Bi = rf.B + ((int64_t)i * rf.nq + q) * ND;
                const double* dBi = rf.dB + ((int64_t)i * rf.nq + q) * ND * NV;
                const double* Bj = rf.B + ((int64_t)j * rf.nq + q) * ND;
                const double* dBj = rf.dB + ((int64_t)j * rf.nq + q) * ND * NV;
                // upwind speeds: own side with its outward normal n, neighbour side with -n
                double sp = 0.0, sm = 0.0;
#pragma unroll
                for (int b = 0; b < ND; ++b) {
                    sp = fma(ph[b], dotNV<NV>(dBi + b * NV, gn), sp);
                    sm = fma(ph2[b], dotNV<NV>(dBj + b * NV, gn2), sm);
                }
                sp *= Dc;
                sm *= -D2;
                const double un = 0.5 * (sp + fabs(sp)), un2 = 0.5 * (sm + fabs(sm));
                const double w = rf.w[q] * area;
                const double Ba = Bi[a];
                const double dna = dotNV<NV>(dBi + a * NV, gn);
#pragma unroll
                for (int b = 0; b < ND; ++b) {
                    const double Bb = Bi[b], B2b = Bj[b];
                    const double dnb = dotNV<NV>(dBi + b * NV, gn);
                    const double dn2b = dotNV<NV>(dBj + b * NV, gn2);
                    diag[b] += w * (-0.5 * Dc * dnb * Ba - 0.5 * Dc * dna * Bb + pen * Dc * Ba * Bb - z * ka.psi * un * Ba * Bb);
                    off[b] += w * (-0.5 * D2 * dn2b * Ba + 0.5 * Dc * dna * B2b - pen * D2 * Ba * B2b + z * ka.psi * un2 * Ba * B2b);
                }
            }
        }
        double* o = blk + ((c * (NV + 1) + 1 + i) * ND + a) * ND;
#pragma unroll
        for (int b = 0; b < ND; ++b) o[b] = off[b];
    }
    double* o = blk + ((c * (NV + 1)) * ND + a) * ND;
#pragma unroll
    for (int b = 0; b < ND; ++b) o[b] = diag[b];
}

// ------------------------------------------------------------------------------------------------------------
// y_c = sum_j blk[c][j] x_{nbr_j(c)}     one thread per (cell, row); blockIdx.y = system
// ------------------------------------------------------------------------------------------------------------
template <int D, int ND>
__global__ __launch_bounds__(256) void k_tab_apply(MeshDev m, const double* __restrict__ blk_all, const double* __restrict__ x_all,
                                                   double* __restrict__ y_all) {
    constexpr int NV = D + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.nc_owned * ND) return;
    const int s = blockIdx.y;
    const int64_t c = t / ND;
    const int a = (int)(t % ND);
    const double* blk = blk_all + (int64_t)s * m.nc * (NV + 1) * ND * ND;
    const double* x = x_all + (int64_t)s * m.nc * ND;
    const double* row = blk + ((c * (NV + 1)) * ND + a) * ND;
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < ND; ++b) acc = fma(row[b], x[c * ND + b], acc);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int64_t nb = m.nbr[c * NV + i];
        if (nb < 0) continue;
        const double* r2 = row + (int64_t)(1 + i) * ND * ND;
        double s2 = 0.0;
#pragma unroll
        for (int b = 0; b < ND; ++b) s2 = fma(r2[b], x[nb * ND + b], s2);
        acc += s2;
    }
    y_all[(int64_t)s * m.nc * ND + t] = acc;
}

// inverse of the diagonal blocks (block-Jacobi): one thread per cell, its block in LDS (column index strided by the
// 64 lanes -> conflict free), in-place Gauss-Jordan without pivoting (blocks are SPD resp. mass-dominated)
template <int D, int ND>
__global__ __launch_bounds__(64) void k_tab_block_inverse(MeshDev m, const double* __restrict__ blk_all, bjreal* __restrict__ binv_all, int symmetric) {
    constexpr int NV = D + 1;
    __shared__ double M[ND * ND * 64];
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int s = blockIdx.y;
    const int tid = threadIdx.x;
    if (c >= m.nc_owned) return;
    const double* src = blk_all + (int64_t)s * m.nc * (NV + 1) * ND * ND + c * (NV + 1) * ND * ND;
    for (int k = 0; k < ND * ND; ++k) M[k * 64 + tid] = src[k];
    for (int p = 0; p < ND; ++p) {
        const double piv = 1.0 / M[(p * ND + p) * 64 + tid];
        M[(p * ND + p) * 64 + tid] = 1.0;
        for (int k = 0; k < ND; ++k) M[(p * ND + k) * 64 + tid] *= piv;
        for (int r = 0; r < ND; ++r) {
            if (r == p) continue;
            const double f = M[(r * ND + p) * 64 + tid];
            M[(r * ND + p) * 64 + tid] = 0.0;
            for (int k = 0; k < ND; ++k) M[(r * ND + k) * 64 + tid] -= f * M[(p * ND + k) * 64 + tid];
        }
    }
    bjreal* dst = binv_all + (int64_t)s * m.nc * ND * ND + c * ND * ND;
    if (symmetric) {                                   // EMI: exactly symmetric after rounding to fp32
        for (int a = 0; a < ND; ++a)
            for (int b = 0; b < ND; ++b) dst[a * ND + b] = (bjreal)(0.5 * (M[(a * ND + b) * 64 + tid] + M[(b * ND + a) * 64 + tid]));
    } else {
        for (int k = 0; k < ND * ND; ++k) dst[k] = (bjreal)M[k * 64 + tid];
    }
}

// ------------------------------------------------------------------------------------------------------------
// L_emi: one thread per CELL, all ND rows (round 3; rounds 1-2 ran one thread per (cell, row): every row recomputed the cell's
// geometry, the nodal combination S, the neighbour's data and every quadrature-point flux -- 419 us per step for 124 416 P2 cells)
// ------------------------------------------------------------------------------------------------------------
template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_emi_rhs(MeshDev m, TabRule rc, TabRule rf, TabRule rm, const double* __restrict__ cc,
                                                     const double* __restrict__ celim, const double* __restrict__ Dk,
                                                     const double* __restrict__ phiM, const double* __restrict__ Ich, IonZ ia,
                                                     double F, double C_phi, int splitting, const double* __restrict__ extra,
                                                     double* __restrict__ out) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m.nc_owned) return;
    CellGeom<D> K;
    cell_geo<D>(m, c, K);
    // S = sum_k F z_k D_k c_k is a DG-p function (D_k is cell-wise constant): nodal combination
    double S[ND], acc[ND];
#pragma unroll
    for (int b = 0; b < ND; ++b) { S[b] = 0.0; acc[b] = 0.0; }
    for (int i = 0; i < ia.n; ++i) {
        const double* src = (i < ia.n - 1) ? cc + (int64_t)i * m.nc * ND : celim;
        const double f = F * ia.z[i] * Dk[(int64_t)i * m.nc + c];
#pragma unroll
        for (int b = 0; b < ND; ++b) S[b] = fma(f, src[c * ND + b], S[b]);
    }
    for (int q = 0; q < rc.nq; ++q) {
        const double* dB = rc.dB + (int64_t)q * ND * NV;
        double gS[D];
#pragma unroll
        for (int k = 0; k < D; ++k) gS[k] = 0.0;
#pragma unroll
        for (int l = 0; l < NV; ++l) {
            double sl = 0.0;
#pragma unroll
            for (int b = 0; b < ND; ++b) sl = fma(S[b], dB[b * NV + l], sl);
#pragma unroll
            for (int k = 0; k < D; ++k) gS[k] = fma(sl, K.g[l][k], gS[k]);
        }
        // grad(S) . grad(lambda_l), then each row's gradient is a combination of those
        double gl[NV];
#pragma unroll
        for (int l = 0; l < NV; ++l) gl[l] = dotD<D>(gS, K.g[l]);
        const double wv = rc.w[q] * K.vol;
#pragma unroll
        for (int a = 0; a < ND; ++a) acc[a] -= wv * dotNV<NV>(dB + a * NV, gl);
    }
    const uint32_t flags = m.fflag[c];
    for (int i = 0; i < NV; ++i) {
        const uint32_t fb = (flags >> (8 * i)) & 0xffu;
        const uint32_t kind = (fb >> 2) & 3u;
        const int j = (int)(fb & 3u);
        const int64_t nb = m.nbr[c * NV + i];
        if (nb < 0) continue;
        if (kind == FK_SIPG) {
            CellGeom<D> K2;
            cell_geo<D>(m, nb, K2);
            double S2[ND];
#pragma unroll
            for (int b = 0; b < ND; ++b) S2[b] = 0.0;
            for (int k = 0; k < ia.n; ++k) {
                const double* src = (k < ia.n - 1) ? cc + (int64_t)k * m.nc * ND : celim;
                const double f = F * ia.z[k] * Dk[(int64_t)k * m.nc + nb];
#pragma unroll
                for (int b = 0; b < ND; ++b) S2[b] = fma(f, src[nb * ND + b], S2[b]);
            }
            double nrm[D], gn[NV], gn2[NV];
            const double area = facet_frame<D>(K, i, nrm, gn);
#pragma unroll
            for (int l = 0; l < NV; ++l) gn2[l] = dotD<D>(K2.g[l], nrm);
            for (int q = 0; q < rf.nq; ++q) {
                const double* Bi = rf.B + ((int64_t)i * rf.nq + q) * ND;
                const double* dBi = rf.dB + ((int64_t)i * rf.nq + q) * ND * NV;
                const double* dBj = rf.dB + ((int64_t)j * rf.nq + q) * ND * NV;
                double flux = 0.0;
#pragma unroll
                for (int b = 0; b < ND; ++b) {
                    flux = fma(S[b], dotNV<NV>(dBi + b * NV, gn), flux);
                    flux = fma(S2[b], dotNV<NV>(dBj + b * NV, gn2), flux);
                }
                const double wf = rf.w[q] * area * 0.5 * flux;
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[a] = fma(wf, Bi[a], acc[a]);
            }
        } else if (kind == FK_MEMBRANE && splitting != 2) {    // MMS: Robin data arrives as host-integrated extra RHS
            const int64_t f = m.cfacet[c * NV + i];
            double g = phiM[f];
            if (!splitting) {
                double It = 0.0;
                for (int k = 0; k < ia.n; ++k) It += Ich[(int64_t)k * m.nf + f];
                g -= It / C_phi;
            }
            const double sgn = ((fb >> 4) & 1u) ? -1.0 : 1.0;      // e (plus) side carries -v_e
            const double area = (double)D * K.vol * sqrt(dotD<D>(K.g[i], K.g[i]));
            const double wm = sgn * C_phi * g * area;
            for (int q = 0; q < rm.nq; ++q) {
                const double* Bi = rm.B + ((int64_t)i * rm.nq + q) * ND;
                const double wq = wm * rm.w[q];
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[a] = fma(wq, Bi[a], acc[a]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < ND; ++a) out[c * ND + a] = acc[a] + (extra ? extra[c * ND + a] : 0.0);
}

// ------------------------------------------------------------------------------------------------------------
// L_knp of species s = blockIdx.y: one thread per CELL, all ND rows (round 3, as above)
// ------------------------------------------------------------------------------------------------------------
struct KnpRhsTab { double F, C_M, dt; int splitting; const double* mms_C; const double* extra; };

template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_knp_rhs(MeshDev m, TabRule rc, TabRule rm, const double* __restrict__ cc,
                                                     const double* __restrict__ cprev, const double* __restrict__ celim,
                                                     const double* __restrict__ phi, const double* __restrict__ Dk,
                                                     const double* __restrict__ phiM, const double* __restrict__ Ich,
                                                     const double* __restrict__ fsrc, IonZ ia, KnpRhsTab ra,
                                                     double* __restrict__ out_all) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m.nc_owned) return;
    const int s = blockIdx.y;
    CellGeom<D> K;
    cell_geo<D>(m, c, K);
    const double z = ia.z[s];
    const double Dc = Dk[(int64_t)s * m.nc + c];
    double cp[ND], acc[ND];
    ld<ND>(cprev + (int64_t)s * m.nc * ND, c, cp);
#pragma unroll
    for (int a = 0; a < ND; ++a) acc[a] = 0.0;
    const double fs = fsrc ? fsrc[(int64_t)s * m.nc + c] : 0.0;
    for (int q = 0; q < rc.nq; ++q) {
        const double* B = rc.B + (int64_t)q * ND;
        double cq = 0.0;
#pragma unroll
        for (int b = 0; b < ND; ++b) cq = fma(B[b], cp[b], cq);
        const double wq = rc.w[q] * K.vol * (cq / ra.dt + fs);
#pragma unroll
        for (int a = 0; a < ND; ++a) acc[a] = fma(wq, B[a], acc[a]);
    }
    const uint32_t flags = m.fflag[c];
    for (int i = 0; i < NV; ++i) {
        const uint32_t fb = (flags >> (8 * i)) & 0xffu;
        if (((fb >> 2) & 3u) != FK_MEMBRANE) continue;
        const int64_t nb = m.nbr[c * NV + i];
        if (nb < 0) continue;
        const int j = (int)(fb & 3u);
        const int64_t f = m.cfacet[c * NV + i];
        const bool is_e = (fb >> 4) & 1u;
        // own-side nodal fields: c_k (current iterate), alpha_sum = sum_k D_k z_k^2 c_k, phi on both sides
        double ck[ND], as[ND], po[ND], pn[ND];
        ld<ND>(cc + (int64_t)s * m.nc * ND, c, ck);
        ld<ND>(phi, c, po);
        ld<ND>(phi, nb, pn);
#pragma unroll
        for (int b = 0; b < ND; ++b) as[b] = 0.0;
        for (int k = 0; k < ia.n; ++k) {
            const double* src = (k < ia.n - 1) ? cc + (int64_t)k * m.nc * ND : celim;
            const double fz = ia.z[k] * ia.z[k] * Dk[(int64_t)k * m.nc + c];
#pragma unroll
            for (int b = 0; b < ND; ++b) as[b] = fma(fz, src[c * ND + b], as[b]);
        }
        const double area = (double)D * K.vol * sqrt(dotD<D>(K.g[i], K.g[i]));
        const double sgn = is_e ? -1.0 : 1.0;
        if (ra.splitting == 2) {
            // MMS (solver.py:649-650): -(phi_i - phi_e)(C_i v_i - C_e v_e) with the DG0 coupling coefficient C
            const double Cown = ra.mms_C[(int64_t)s * m.nc + c];
            for (int q = 0; q < rm.nq; ++q) {
                const double* Bi = rm.B + ((int64_t)i * rm.nq + q) * ND;
                const double* Bj = rm.B + ((int64_t)j * rm.nq + q) * ND;
                double pq = 0.0, pq2 = 0.0;
#pragma unroll
                for (int b = 0; b < ND; ++b) { pq = fma(Bi[b], po[b], pq); pq2 = fma(Bj[b], pn[b], pq2); }
                const double jump = is_e ? (pq2 - pq) : (pq - pq2);
                const double wq = rm.w[q] * area * sgn * Cown * jump;
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[a] -= wq * Bi[a];
            }
            continue;
        }
        const double I_k = Ich[(int64_t)s * m.nf + f];
        double I_tot = 0.0;
        for (int k = 0; k < ia.n; ++k) I_tot += Ich[(int64_t)k * m.nf + f];
        const double pM = phiM[f];
        for (int q = 0; q < rm.nq; ++q) {
            const double* Bi = rm.B + ((int64_t)i * rm.nq + q) * ND;
            const double* Bj = rm.B + ((int64_t)j * rm.nq + q) * ND;
            double cq = 0.0, aq = 0.0, pq = 0.0, pq2 = 0.0;
#pragma unroll
            for (int b = 0; b < ND; ++b) {
                cq = fma(Bi[b], ck[b], cq);
                aq = fma(Bi[b], as[b], aq);
                pq = fma(Bi[b], po[b], pq);
                pq2 = fma(Bj[b], pn[b], pq2);
            }
            const double alpha = Dc * z * z * cq / aq;
            const double C = alpha * ra.C_M / (ra.F * z * ra.dt);
            double g = pM - ra.dt / (ra.C_M * alpha) * I_k;
            if (ra.splitting) g += (ra.dt / ra.C_M) * I_tot;
            const double jump = is_e ? (pq2 - pq) : (pq - pq2);       // phi_i - phi_e
            const double wq = rm.w[q] * area * sgn * C * (g - jump);
#pragma unroll
            for (int a = 0; a < ND; ++a) acc[a] = fma(wq, Bi[a], acc[a]);
        }
    }
    double* out = out_all + (int64_t)s * m.nc * ND;
    const double* ex = ra.extra ? ra.extra + (int64_t)s * m.nc * ND : nullptr;
#pragma unroll
    for (int a = 0; a < ND; ++a) out[c * ND + a] = acc[a] + (ex ? ex[c * ND + a] : 0.0);
}

// ------------------------------------------------------------------------------------------------------------
// step III per membrane facet: phi_M = avg(phi_i - phi_e), E_k = RT/(F z_k) avg ln(c_e / c_i)
// ------------------------------------------------------------------------------------------------------------
template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_facet_updates(MeshDev m, TabRule ra, TabRule rn, const double* __restrict__ cc,
                                                           const double* __restrict__ celim, const double* __restrict__ phi,
                                                           double* __restrict__ phiM, double* __restrict__ E, IonZ ia,
                                                           double RT_over_F) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.nmf) return;
    const int32_t* row = m.mf + 6 * t;
    if (!row[5]) return;
    const int64_t ce = row[0], ci = row[1];
    const int le = row[2], li = row[3];
    const int64_t f = row[4];
    double ve[ND], vi[ND];
    if (phi) {
        ld<ND>(phi, ce, ve);
        ld<ND>(phi, ci, vi);
        double s = 0.0;
        for (int q = 0; q < ra.nq; ++q) {
            const double* Be = ra.B + ((int64_t)le * ra.nq + q) * ND;
            const double* Bi = ra.B + ((int64_t)li * ra.nq + q) * ND;
            double pe = 0.0, pi = 0.0;
#pragma unroll
            for (int b = 0; b < ND; ++b) { pe = fma(Be[b], ve[b], pe); pi = fma(Bi[b], vi[b], pi); }
            s += ra.w[q] * (pi - pe);
        }
        phiM[f] = s;
    }
    for (int k = 0; k < ia.n; ++k) {
        const double* src = (k < ia.n - 1) ? cc + (int64_t)k * m.nc * ND : celim;
        ld<ND>(src, ce, ve);
        ld<ND>(src, ci, vi);
        double s = 0.0;
        for (int q = 0; q < rn.nq; ++q) {
            const double* Be = rn.B + ((int64_t)le * rn.nq + q) * ND;
            const double* Bi = rn.B + ((int64_t)li * rn.nq + q) * ND;
            double xe = 0.0, xi = 0.0;
#pragma unroll
            for (int b = 0; b < ND; ++b) { xe = fma(Be[b], ve[b], xe); xi = fma(Bi[b], vi[b], xi); }
            s += rn.w[q] * log(xe / xi);
        }
        E[(int64_t)k * m.nf + f] = RT_over_F / ia.z[k] * s;
    }
}

template <int D, int ND>
__global__ __launch_bounds__(128) void k_tab_facet_trace(MeshDev m, TabRule ra, const double* __restrict__ nodal, int side,
                                                         double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.nmf) return;
    const int32_t* row = m.mf + 6 * t;
    const int64_t cell = row[side];
    const int lf = row[2 + side];
    double v[ND];
    ld<ND>(nodal, cell, v);
    double s = 0.0;
    for (int q = 0; q < ra.nq; ++q) {
        const double* B = ra.B + ((int64_t)lf * ra.nq + q) * ND;
        double x = 0.0;
#pragma unroll
        for (int b = 0; b < ND; ++b) x = fma(B[b], v[b], x);
        s += ra.w[q] * x;
    }
    out[row[4]] = s;
}

}  // namespace

// ---- launchers (called from the degree dispatch in apply_p1.hip / rhs_p1.hip) ---------------------------------
static int need_tabs(knp_ctx* c, std::initializer_list<int> slots) {
    for (int s : slots)
        if (!c->tab[s].w) { c->err = "DG-p path: tabulation slot " + std::to_string(s) + " not set (knp_set_tabulation)"; return -1; }
    return 0;
}

#define TAB_DISPATCH(c, KERN, grid, block, ...)                                                              \
    do {                                                                                                     \
        if ((c)->m.dim == 3) hipLaunchKernelGGL((KERN<3, 10>), grid, block, 0, (c)->stream, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERN<2, 6>), grid, block, 0, (c)->stream, __VA_ARGS__);                       \
        HIPCHK(c, hipGetLastError());                                                                        \
    } while (0)

bool p2_assembled() {
    static const bool on = getenv("KNP_P2_ASSEMBLED") && atoi(getenv("KNP_P2_ASSEMBLED")) == 1;
    return on;
}

static int ensure_blocks(knp_ctx* c) {
    const size_t per = (size_t)c->m.nc * (c->m.dim + 2) * c->nd * c->nd;
    if (!c->blk_emi) HIPCHK(c, hipMalloc((void**)&c->blk_emi, sizeof(double) * per));
    if (!c->blk_knp) HIPCHK(c, hipMalloc((void**)&c->blk_knp, sizeof(double) * per * c->p.n_sys));
    return 0;
}

static dim3 cells_grid(knp_ctx* c, int block, int ny = 1) {
    return dim3((unsigned)((c->m.nc_owned + block - 1) / block), (unsigned)ny);
}
static dim3 rows_grid(knp_ctx* c, int block, int ny = 1) {
    return dim3((unsigned)((c->m.nc_owned * c->nd + block - 1) / block), (unsigned)ny);
}

int tab_kappa(knp_ctx* c, const double* cc, const double* celim, double* kappa) {
    const int64_t n = c->m.nc * c->nd;
    hipLaunchKernelGGL(k_tab_kappa, dim3((unsigned)grid_for(n)), dim3(KNP_BLOCK), 0, c->stream, c->m.nc, c->nd, cc, celim,
                       (const double*)c->D, ion_z(c), c->p.F, c->p.psi, kappa);
    HIPCHK(c, hipGetLastError());
    if (!p2_assembled()) return 0;                       // matrix-free applies read kappa directly (apply_p2.hip)
    if (need_tabs(c, {KNP_TAB_CELL_STIFF, KNP_TAB_FACET_EMI, KNP_TAB_FACET_MEM}) || ensure_blocks(c)) return -1;
    TAB_DISPATCH(c, k_tab_assemble_emi, rows_grid(c, 128), dim3(128), c->m, c->tab[KNP_TAB_CELL_STIFF], c->tab[KNP_TAB_FACET_EMI],
                 c->tab[KNP_TAB_FACET_MEM], (const double*)kappa, c->p.tau_emi, c->p.C_phi, c->blk_emi);
    return 0;
}

int tab_assemble_knp(knp_ctx* c, const double* phi) {
    if (need_tabs(c, {KNP_TAB_CELL_STIFF, KNP_TAB_FACET_KNP}) || ensure_blocks(c)) return -1;
    KnpTabArgs ka;
    ka.inv_dt = 1.0 / c->p.dt; ka.psi = c->p.psi; ka.tau = c->p.tau_knp;
    for (int k = 0; k < KNP_MAX_SYS; ++k) ka.z[k] = k < c->p.n_sys ? c->p.z[k] : 0.0;
    TAB_DISPATCH(c, k_tab_assemble_knp, rows_grid(c, 128, c->p.n_sys), dim3(128), c->m, c->tab[KNP_TAB_CELL_STIFF],
                 c->tab[KNP_TAB_FACET_KNP], phi, (const double*)c->D, ka, c->blk_knp);
    return 0;
}

int tab_apply(knp_ctx* c, int which, const double* x, double* y) {
    const double* blk = which == 0 ? c->blk_emi : c->blk_knp;
    if (!blk) { c->err = "DG-p path: operator blocks not assembled (update_kappa / update_dnphi first)"; return -1; }
    TAB_DISPATCH(c, k_tab_apply, rows_grid(c, 256, which == 0 ? 1 : c->p.n_sys), dim3(256), c->m, blk, x, y);
    return 0;
}

int tab_block_inverse(knp_ctx* c, int which, bjreal* binv) {
    const double* blk = which == 0 ? c->blk_emi : c->blk_knp;
    if (!blk) { c->err = "DG-p path: operator blocks not assembled"; return -1; }
    const dim3 g((unsigned)((c->m.nc_owned + 63) / 64), (unsigned)(which == 0 ? 1 : c->p.n_sys));
    TAB_DISPATCH(c, k_tab_block_inverse, g, dim3(64), c->m, blk, binv, which == 0 ? 1 : 0);
    return 0;
}

int tab_emi_rhs(knp_ctx* c, const double* cc, const double* celim, const double* phiM, const double* Ich, double* b) {
    if (need_tabs(c, {KNP_TAB_CELL_RHS_EMI, KNP_TAB_FACET_RHS_EMI, KNP_TAB_FACET_MEM_LIN})) return -1;
    TAB_DISPATCH(c, k_tab_emi_rhs, cells_grid(c, 128), dim3(128), c->m, c->tab[KNP_TAB_CELL_RHS_EMI], c->tab[KNP_TAB_FACET_RHS_EMI],
                 c->tab[KNP_TAB_FACET_MEM_LIN], cc, celim, (const double*)c->D, phiM, Ich, ion_z(c), c->p.F, c->p.C_phi,
                 c->p.splitting, (const double*)c->extra_emi, b);
    return 0;
}

int tab_knp_rhs(knp_ctx* c, const double* cc, const double* cprev, const double* celim, const double* phi, const double* phiM,
                const double* Ich, double* b) {
    if (need_tabs(c, {KNP_TAB_CELL_MASS, KNP_TAB_FACET_MEM_KNP})) return -1;
    KnpRhsTab ra{c->p.F, c->p.C_M, c->p.dt, c->p.splitting, (const double*)c->mms_C, (const double*)c->extra_knp};
    TAB_DISPATCH(c, k_tab_knp_rhs, cells_grid(c, 128, c->p.n_sys), dim3(128), c->m, c->tab[KNP_TAB_CELL_MASS],
                 c->tab[KNP_TAB_FACET_MEM_KNP], cc, cprev, celim, phi, (const double*)c->D, phiM, Ich, (const double*)c->fsrc,
                 ion_z(c), ra, b);
    return 0;
}

int tab_step_updates(knp_ctx* c, const double* cc, double* celim, const double* phi, double* phiM, double* E, bool do_celim) {
    if (need_tabs(c, {KNP_TAB_FACET_AVG, KNP_TAB_FACET_NERNST})) return -1;
    if (do_celim) {
        const int64_t n = c->m.nc * c->nd;
        hipLaunchKernelGGL(k_tab_celim, dim3((unsigned)grid_for(n)), dim3(KNP_BLOCK), 0, c->stream, c->m.nc, c->nd, cc,
                           (const double*)c->rho, ion_z(c), celim);
    }
    if (c->m.nmf > 0)
        TAB_DISPATCH(c, k_tab_facet_updates, dim3((unsigned)((c->m.nmf + 127) / 128)), dim3(128), c->m, c->tab[KNP_TAB_FACET_AVG],
                     c->tab[KNP_TAB_FACET_NERNST], cc, (const double*)celim, phi, phiM, E, ion_z(c), c->p.R * c->p.T / c->p.F);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int tab_facet_trace(knp_ctx* c, const double* nodal, int side, double* out) {
    if (need_tabs(c, {KNP_TAB_FACET_AVG})) return -1;
    if (c->m.nmf > 0)
        TAB_DISPATCH(c, k_tab_facet_trace, dim3((unsigned)((c->m.nmf + 127) / 128)), dim3(128), c->m, c->tab[KNP_TAB_FACET_AVG], nodal,
                     side, out);
    return 0;
}

void tab_free(knp_ctx* c) {
    for (int s = 0; s < KNP_TAB_COUNT; ++s) {
        hipFree(c->tab_mem[s]);
        c->tab_mem[s] = nullptr;
        c->tab[s] = TabRule();
    }
    hipFree(c->blk_emi); hipFree(c->blk_knp);
    c->blk_emi = c->blk_knp = nullptr;
}

extern "C" int knp_set_tabulation(knp_ctx* c, int slot, int nloc, int nq, const double* w, const double* B, const double* dB) {
    if (!c || slot < 0 || slot >= KNP_TAB_COUNT || !w || !B || !dB) return -1;
    const int NV = c->m.dim + 1;
    const bool facet = slot >= KNP_TAB_FACET_EMI;
    if (nq < 1 || nq > 256 || nloc != (facet ? NV : 1)) { c->err = "set_tabulation: nloc must be 1 (cell rules) or dim+1 (facet rules)"; return -1; }
    const size_t nB = (size_t)nloc * nq * c->nd, ndB = nB * NV;
    hipFree(c->tab_mem[slot]);
    c->tab_mem[slot] = nullptr;
    HIPCHK(c, hipMalloc((void**)&c->tab_mem[slot], sizeof(double) * (nq + nB + ndB)));
    double* base = c->tab_mem[slot];
    HIPCHK(c, hipMemcpy(base, w, sizeof(double) * nq, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(base + nq, B, sizeof(double) * nB, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(base + nq + nB, dB, sizeof(double) * ndB, hipMemcpyHostToDevice));
    TabRule r;
    r.nq = nq; r.w = base; r.B = base + nq; r.dB = base + nq + nB;
    c->tab[slot] = r;
    return 0;
}
