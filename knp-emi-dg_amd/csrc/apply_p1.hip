// Matrix-free P1 SIPG operator applies (EMI potential operator, batched KNP species operator),
// cell-based gather: one thread owns one cell, computes the cell's volume integral and the
// contribution of each of its D+1 facets to ITS OWN test functions, reading the neighbour's
// DoFs / coefficients / apex vertex.  No atomics, bitwise reproducible.
//
// Replaces: dolfin.assemble(a_emi) + PETSc MatMult        (reference: src/knpemidg/solver.py:325-328,346,477,509)
//           dolfin.assemble(A_knp) + PETSc MatMult        (reference: src/knpemidg/solver.py:586-594,730,771)
// P1 facet integrals are closed forms (mass / triple-product matrices of a (D-1)-simplex).
#include "cell_geom.hpp"

template <int D> struct FacetConst;
template <> struct FacetConst<3> { static constexpr double mass = 1.0 / 12.0, trip = 1.0 / 60.0; };
template <> struct FacetConst<2> { static constexpr double mass = 1.0 / 6.0, trip = 1.0 / 24.0; };

// ------------------------------------------------------------------------------------------
// EMI:  y = A(kappa) x
//   A(u,v) = int kappa grad u.grad v - int_dS0 avg(kappa grad u).n jump(v) - int_dS0 avg(kappa grad v).n jump(u)
//          + int_dS0 tau/avg(h) avg(kappa) jump(u) jump(v) + C_phi int_dS(mem) jump(u) jump(v)
// ------------------------------------------------------------------------------------------
template <int D, int I, bool DIAG>
__device__ __forceinline__ void emi_facet(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                          const double* xv, const double* kv,
                                          const double* __restrict__ x, const double* __restrict__ kappa,
                                          double C_phi, double tau, double* y) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    FacetGeom<D> F;
    facet_own<D, I>(K, F);
    double xn[NV];
    if (DIAG) {
#pragma unroll
        for (int a = 0; a < NV; ++a) xn[a] = 0.0;
    } else {
        load_nodal<D>(x, Kp, xn);
    }
    double du[D], sdu = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        du[mm] = xv[mm + (mm >= I)] - pick_facet<D>(xn, mm, j);
        sdu += du[mm];
    }
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * F.area * FacetConst<D>::mass;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[mm + (mm >= I)] += w * (sdu + du[mm]);
        return;
    }
    double kn[NV];
    load_nodal<D>(kappa, Kp, kn);
    double Xo[D];
    load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
    facet_neighbour<D, I>(K, Xo, F);
    double dnu_own = 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) dnu_own += xv[a] * F.dn[a];
    double foot = 0.0, kf[D], knf[D], sk = 0.0, skn = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        foot += F.beta[mm] * pick_facet<D>(xn, mm, j);
        kf[mm] = kv[mm + (mm >= I)];
        knf[mm] = pick_facet<D>(kn, mm, j);
        sk += kf[mm];
        skn += knf[mm];
    }
    const double dnu_nb = (pick_apex<D>(xn, j) - foot) / F.hp;
    const double am = F.area * FacetConst<D>::mass;
    // consistency term on own test functions (jump(v) = +v on this side)
    double q = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        y[mm + (mm >= I)] -= 0.5 * am * (dnu_own * (sk + kf[mm]) + dnu_nb * (skn + knf[mm]));
        q += kf[mm] * (sdu + du[mm]);
    }
    // adjoint consistency: -1/2 (grad v.n) int kappa_K jump(u)
    q *= 0.5 * am;
#pragma unroll
    for (int a = 0; a < NV; ++a) y[a] -= F.dn[a] * q;
    // penalty
    const double hbar = 0.5 * (sqrt(K.h2) + sqrt(F.hN2));
    const double pw = tau / hbar * F.area * FacetConst<D>::trip;
    double kb[D], skb = 0.0, skd = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kb[mm] = 0.5 * (kf[mm] + knf[mm]);
        skb += kb[mm];
        skd += kb[mm] * du[mm];
    }
#pragma unroll
    for (int mm = 0; mm < D; ++mm)
        y[mm + (mm >= I)] += pw * (skb * sdu + kb[mm] * sdu + du[mm] * skb + skd + 2.0 * kb[mm] * du[mm]);
}

template <int D, bool DIAG>
__device__ __forceinline__ void emi_cell(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                         const double* xv, const double* kv,
                                         const double* __restrict__ x, const double* __restrict__ kappa,
                                         double C_phi, double tau, double* y) {
    constexpr int NV = D + 1;
    double gu[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        gu[k] = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) gu[k] += xv[a] * K.g[a][k];
    }
    double kbar = 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) kbar += kv[a];
    kbar *= K.vol / (double)NV;
#pragma unroll
    for (int a = 0; a < NV; ++a) y[a] = kbar * dotD<D>(gu, K.g[a]);
    emi_facet<D, 0, DIAG>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, y);
    emi_facet<D, 1, DIAG>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, y);
    emi_facet<D, 2, DIAG>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, y);
    if (D == 3) emi_facet<D, (D == 3 ? 3 : 0), DIAG>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, y);
}

template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_emi_apply(MeshDev m, const double* __restrict__ x,
                                                         const double* __restrict__ kappa, double* __restrict__ y,
                                                         double C_phi, double tau) {
    constexpr int NV = D + 1;
    const int64_t c = xcd_block(blockIdx.x, gridDim.x) * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc_owned) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double xv[NV], kv[NV], yv[NV];
    load_nodal<D>(x, c, xv);
    load_nodal<D>(kappa, c, kv);
    emi_cell<D, false>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, yv);
    store_nodal<D>(y, c, yv);
}

// in-register inverse of a small dense matrix (Gauss-Jordan, no pivoting: the blocks are SPD
// for EMI and diagonally dominant M/dt + diffusion blocks for KNP)
template <int N> __device__ __forceinline__ void invert_small(double (*A)[N]) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const double ip = 1.0 / A[p][p];
        A[p][p] = 1.0;
#pragma unroll
        for (int k = 0; k < N; ++k) A[p][k] *= ip;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r == p) continue;
            const double f = A[r][p];
            A[r][p] = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) A[r][k] -= f * A[p][k];
        }
    }
}

// inverse of the cell-diagonal block of A_emi (block-Jacobi preconditioner), stored [c][row][col]
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_emi_blockjacobi(MeshDev m, const double* __restrict__ kappa,
                                                               double* __restrict__ binv, double C_phi, double tau,
                                                               double shift) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc_owned) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double kv[NV];
    load_nodal<D>(kappa, c, kv);
    double A[NV][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        double e[NV], col[NV];
#pragma unroll
        for (int a = 0; a < NV; ++a) e[a] = (a == b) ? 1.0 : 0.0;
        emi_cell<D, true>(m, K, nb, flags, e, kv, nullptr, kappa, C_phi, tau, col);
#pragma unroll
        for (int a = 0; a < NV; ++a) A[a][b] = col[a];
    }
    // B_emi's mass shift kappa/Lp^2 int u v (reference: solver.py:390-395), lumped with mean kappa
    if (shift != 0.0) {
        double kbar = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) kbar += kv[a];
        kbar /= (double)NV;
        const double w = shift * kbar * K.vol / (double)((D + 1) * (D + 2));
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int b = 0; b < NV; ++b) A[a][b] += w * ((a == b) ? 2.0 : 1.0);
    }
    invert_small<NV>(A);
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int b = 0; b < NV; ++b) binv[(c * NV + a) * NV + b] = A[a][b];
}

// ------------------------------------------------------------------------------------------
// KNP: y_k = A_k x_k for all solved species k at once (shared mesh / geometry / phi data)
//   A_k(u,v) = 1/dt int u v + int D grad u.grad v - int_dS0 avg(D grad u).n jump(v)
//            - int_dS0 avg(D grad v).n jump(u) + int_dS0 tau/avg(h) jump(D u) jump(v)
//            + z psi int D u grad(phi).grad v - z psi int_dS0 jump(v) jump(un u),
//   un = max(D grad(phi).n_own, 0).   `dnphi[c][i]` = grad(phi)_c . n_i (outward) is precomputed
//   once per KNP solve (phi is frozen during the solve).
// ------------------------------------------------------------------------------------------
struct KnpArgs {
    int ns;
    double inv_dt, psi, tau;
    double z[KNP_MAX_SYS];
};

template <int D, int NS, int I, bool DIAG>
__device__ __forceinline__ void knp_facet(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                          const double (*xv)[D + 1], const double* sv, const double* Dk,
                                          const double* __restrict__ x, const double* __restrict__ dnphi,
                                          const double* __restrict__ Dall, int64_t c, const KnpArgs& ka,
                                          double (*y)[D + 1]) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind != FK_SIPG) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    FacetGeom<D> F;
    facet_own<D, I>(K, F);
    double Xo[D];
    load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
    facet_neighbour<D, I>(K, Xo, F);
    const double s_nb = dnphi[Kp * NV + j];
    const double sp_own = fmax(sv[I], 0.0), sp_nb = fmax(s_nb, 0.0);
    const double hbar = 0.5 * (sqrt(K.h2) + sqrt(F.hN2));
    const double pen = ka.tau / hbar;
    const double am = F.area * FacetConst<D>::mass;
    const double aD = F.area / (double)D;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double Dn = Dall[(int64_t)k * m.nc + Kp];
        double xn[NV];
        if (DIAG) {
#pragma unroll
            for (int a = 0; a < NV; ++a) xn[a] = 0.0;
        } else {
            load_nodal<D>(x + (int64_t)k * m.nc * NV, Kp, xn);
        }
        double dnu_own = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) dnu_own += xv[k][a] * F.dn[a];
        double foot = 0.0, sdu = 0.0, w[D], sw = 0.0, u[D], su = 0.0;
        const double un = Dk[k] * sp_own, unn = Dn * sp_nb;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double xo = xv[k][mm + (mm >= I)];
            const double xnb = pick_facet<D>(xn, mm, j);
            foot += F.beta[mm] * xnb;
            sdu += xo - xnb;
            w[mm] = Dk[k] * xo - Dn * xnb;
            sw += w[mm];
            u[mm] = un * xo - unn * xnb;
            su += u[mm];
        }
        const double dnu_nb = (pick_apex<D>(xn, j) - foot) / F.hp;
        const double t1 = 0.5 * (Dk[k] * dnu_own + Dn * dnu_nb) * aD;
        const double t2 = 0.5 * Dk[k] * aD * sdu;
        const double zp = ka.z[k] * ka.psi;
#pragma unroll
        for (int a = 0; a < NV; ++a) y[k][a] -= F.dn[a] * t2;
#pragma unroll
        for (int mm = 0; mm < D; ++mm)
            y[k][mm + (mm >= I)] += -t1 + am * (pen * (sw + w[mm]) - zp * (su + u[mm]));
    }
}

template <int D, int NS, bool DIAG>
__device__ __forceinline__ void knp_cell(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                         const double (*xv)[D + 1], const double* sv, const double* Dk,
                                         const double* __restrict__ x, const double* __restrict__ dnphi,
                                         const double* __restrict__ Dall, int64_t c, const KnpArgs& ka,
                                         double (*y)[D + 1]) {
    constexpr int NV = D + 1;
    // grad(phi).grad(lambda_a) = -|g_a| * (grad(phi).n_a)
    double gphi[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) gphi[a] = -sqrt(dotD<D>(K.g[a], K.g[a])) * sv[a];
    const double mw = ka.inv_dt * K.vol / (double)((D + 1) * (D + 2));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double gu[D], sx = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            gu[kk] = 0.0;
#pragma unroll
            for (int a = 0; a < NV; ++a) gu[kk] += xv[k][a] * K.g[a][kk];
        }
#pragma unroll
        for (int a = 0; a < NV; ++a) sx += xv[k][a];
        const double drift = ka.z[k] * ka.psi * Dk[k] * K.vol * sx / (double)NV;
#pragma unroll
        for (int a = 0; a < NV; ++a)
            y[k][a] = mw * (sx + xv[k][a]) + Dk[k] * K.vol * dotD<D>(gu, K.g[a]) + drift * gphi[a];
    }
    knp_facet<D, NS, 0, DIAG>(m, K, nb, flags, xv, sv, Dk, x, dnphi, Dall, c, ka, y);
    knp_facet<D, NS, 1, DIAG>(m, K, nb, flags, xv, sv, Dk, x, dnphi, Dall, c, ka, y);
    knp_facet<D, NS, 2, DIAG>(m, K, nb, flags, xv, sv, Dk, x, dnphi, Dall, c, ka, y);
    if (D == 3) knp_facet<D, NS, (D == 3 ? 3 : 0), DIAG>(m, K, nb, flags, xv, sv, Dk, x, dnphi, Dall, c, ka, y);
}

template <int D, int NS>
__global__ __launch_bounds__(KNP_BLOCK) void k_knp_apply(MeshDev m, const double* __restrict__ x,
                                                         const double* __restrict__ dnphi,
                                                         const double* __restrict__ Dall, double* __restrict__ yout,
                                                         KnpArgs ka) {
    constexpr int NV = D + 1;
    const int64_t c = xcd_block(blockIdx.x, gridDim.x) * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc_owned) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double xv[NS][NV], y[NS][NV], sv[NV], Dk[NS];
    load_nodal<D>(dnphi, c, sv);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        load_nodal<D>(x + (int64_t)k * m.nc * NV, c, xv[k]);
        Dk[k] = Dall[(int64_t)k * m.nc + c];
    }
    knp_cell<D, NS, false>(m, K, nb, flags, xv, sv, Dk, x, dnphi, Dall, c, ka, y);
#pragma unroll
    for (int k = 0; k < NS; ++k) store_nodal<D>(yout + (int64_t)k * m.nc * NV, c, y[k]);
}

// one species per launch dimension (setup only, once per KNP solve)
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_knp_blockjacobi(MeshDev m, const double* __restrict__ dnphi,
                                                               const double* __restrict__ Dall,
                                                               double* __restrict__ binv, KnpArgs ka) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= m.nc_owned) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double sv[NV], Dk[1];
    load_nodal<D>(dnphi, c, sv);
    Dk[0] = Dall[(int64_t)k * m.nc + c];
    KnpArgs k1 = ka;
    k1.z[0] = ka.z[k];
    double A[NV][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        double e[1][NV], col[1][NV];
#pragma unroll
        for (int a = 0; a < NV; ++a) e[0][a] = (a == b) ? 1.0 : 0.0;
        knp_cell<D, 1, true>(m, K, nb, flags, e, sv, Dk, nullptr, dnphi, Dall + (int64_t)k * m.nc, c, k1, col);
#pragma unroll
        for (int a = 0; a < NV; ++a) A[a][b] = col[0][a];
    }
    invert_small<NV>(A);
    double* out = binv + ((int64_t)k * m.nc + c) * NV * NV;
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int b = 0; b < NV; ++b) out[a * NV + b] = A[a][b];
}

// dnphi[c][i] = grad(phi)_c . n_i
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_dnphi(MeshDev m, const double* __restrict__ phi, double* __restrict__ out) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc) return;
    int verts[NV];
    load_cell_ints<D>(m.cells, c, verts);
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double pv[NV], gp[D], s[NV];
    load_nodal<D>(phi, c, pv);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        gp[k] = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) gp[k] += pv[a] * K.g[a][k];
    }
#pragma unroll
    for (int a = 0; a < NV; ++a) s[a] = -dotD<D>(gp, K.g[a]) / sqrt(dotD<D>(K.g[a], K.g[a]));
    store_nodal<D>(out, c, s);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int64_t grid_for(int64_t n) { return (n + KNP_BLOCK - 1) / KNP_BLOCK; }
static inline int64_t grid8(int64_t n) { return ((grid_for(n) + 7) / 8) * 8; }

int launch_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y) {
    if (c->degree != 1) { c->err = "P1 kernels only"; return -1; }
    const dim3 g((unsigned)grid8(c->m.nc_owned)), b(KNP_BLOCK);
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_emi_apply<3>, g, b, 0, c->stream, c->m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    else
        hipLaunchKernelGGL(k_emi_apply<2>, g, b, 0, c->stream, c->m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static KnpArgs make_knp_args(knp_ctx* c) {
    KnpArgs ka;
    ka.ns = c->p.n_sys;
    ka.inv_dt = 1.0 / c->p.dt;
    ka.psi = c->p.psi;
    ka.tau = c->p.tau_knp;
    for (int k = 0; k < KNP_MAX_SYS; ++k) ka.z[k] = (k < c->p.n_sys) ? c->p.z[k] : 0.0;
    return ka;
}

template <int D> static int knp_apply_dispatch(knp_ctx* c, const double* x, const double* dnphi, double* y) {
    const dim3 g((unsigned)grid8(c->m.nc_owned)), b(KNP_BLOCK);
    const KnpArgs ka = make_knp_args(c);
    switch (c->p.n_sys) {
        case 1: hipLaunchKernelGGL((k_knp_apply<D, 1>), g, b, 0, c->stream, c->m, x, dnphi, c->D, y, ka); break;
        case 2: hipLaunchKernelGGL((k_knp_apply<D, 2>), g, b, 0, c->stream, c->m, x, dnphi, c->D, y, ka); break;
        case 3: hipLaunchKernelGGL((k_knp_apply<D, 3>), g, b, 0, c->stream, c->m, x, dnphi, c->D, y, ka); break;
        case 4: hipLaunchKernelGGL((k_knp_apply<D, 4>), g, b, 0, c->stream, c->m, x, dnphi, c->D, y, ka); break;
        default: c->err = "knp_apply supports 1..4 solved species"; return -1;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int launch_knp_apply(knp_ctx* c, const double* x, const double* dnphi, double* y) {
    if (c->degree != 1) { c->err = "P1 kernels only"; return -1; }
    return c->m.dim == 3 ? knp_apply_dispatch<3>(c, x, dnphi, y) : knp_apply_dispatch<2>(c, x, dnphi, y);
}

int launch_emi_blockjacobi(knp_ctx* c, const double* kappa, double* binv) {
    const dim3 g((unsigned)grid_for(c->m.nc_owned)), b(KNP_BLOCK);
    const double shift = 0.0;
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_emi_blockjacobi<3>, g, b, 0, c->stream, c->m, kappa, binv, c->p.C_phi, c->p.tau_emi, shift);
    else
        hipLaunchKernelGGL(k_emi_blockjacobi<2>, g, b, 0, c->stream, c->m, kappa, binv, c->p.C_phi, c->p.tau_emi, shift);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int launch_knp_blockjacobi(knp_ctx* c, const double* dnphi, double* binv) {
    const dim3 g((unsigned)grid_for(c->m.nc_owned), (unsigned)c->p.n_sys), b(KNP_BLOCK);
    const KnpArgs ka = make_knp_args(c);
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_knp_blockjacobi<3>, g, b, 0, c->stream, c->m, dnphi, c->D, binv, ka);
    else
        hipLaunchKernelGGL(k_knp_blockjacobi<2>, g, b, 0, c->stream, c->m, dnphi, c->D, binv, ka);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int launch_dnphi(knp_ctx* c, const double* phi, double* dnphi) {
    const dim3 g((unsigned)grid_for(c->m.nc)), b(KNP_BLOCK);
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_dnphi<3>, g, b, 0, c->stream, c->m, phi, dnphi);
    else
        hipLaunchKernelGGL(k_dnphi<2>, g, b, 0, c->stream, c->m, phi, dnphi);
    HIPCHK(c, hipGetLastError());
    return 0;
}
