// Matrix-free P1 SIPG operator applies (EMI potential operator, batched KNP species operator),
// cell-based gather: one thread owns one cell, computes the cell's volume integral and the
// contribution of each of its D+1 facets to ITS OWN test functions, reading the neighbour's
// DoFs / coefficients / apex vertex.  No atomics, bitwise reproducible.
//
// Replaces: dolfin.assemble(a_emi) + PETSc MatMult        (reference: src/knpemidg/solver.py:325-328,346,477,509)
//           dolfin.assemble(A_knp) + PETSc MatMult        (reference: src/knpemidg/solver.py:586-594,730,771)
// P1 facet integrals are closed forms (mass / triple-product matrices of a (D-1)-simplex); the
// geometry enters only through the own cell's Gram matrix and the neighbour apex's barycentric
// coordinates (cell_geom.hpp).
#include "cell_geom.hpp"
#include <cstdlib>


// grad(w') . g_i for the neighbour's P1 function w' (values wn[], neighbour local facet j)
template <int D, int I>
__device__ __forceinline__ double nb_grad_dot(const CellGeom<D>& K, const double* L, double rLi, const double* wn, int j) {
    const double gr = K.G[I][I] * rLi;
    double s = pick_apex<D>(wn, j) * gr;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const int a = mm + (mm >= I);
        s = fma(pick_facet<D>(wn, mm, j), fma(-L[a], gr, K.G[a][I]), s);
    }
    return s;
}

// ------------------------------------------------------------------------------------------
// EMI:  y = A(kappa) x
//   A(u,v) = int kappa grad u.grad v - int_dS0 avg(kappa grad u).n jump(v) - int_dS0 avg(kappa grad v).n jump(u)
//          + int_dS0 tau/avg(h) avg(kappa) jump(u) jump(v) + C_phi int_dS(mem) jump(u) jump(v)
// With s(w) = grad w . g_i:   area * (grad w . n_i) = -D vol s(w),   area = sqrt(G_ii) D vol.
// ------------------------------------------------------------------------------------------
template <int D, int I, int MODE>
__device__ __forceinline__ void emi_facet(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                          const double* xv, const double* kv, double hK,
                                          const double* __restrict__ x, const double* __restrict__ kappa,
                                          double C_phi, double tau, double* y) {
    // MODE 0: apply (neighbour data gathered from global memory); MODE 1: cell-diagonal block (neighbour values 0)
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    double xn[NV];
    if (MODE == 1) {
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
    const double DV = (double)D * K.vol;
    const double sqG = fast_sqrt(K.G[I][I]);
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * sqG * DV * FacetConst<D>::mass;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[mm + (mm >= I)] = fma(w, sdu + du[mm], y[mm + (mm >= I)]);
        return;
    }
    double kn[NV], Xo[D], L[NV];
    load_nodal<D>(kappa, Kp, kn);
    const double hN = m.h[Kp];
    load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
    apex_bary<D>(K, Xo, L);
    const double rLi = fast_rcp(L[I]);
    // s = grad u . g_i on both sides
    double s_own = 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) s_own = fma(xv[a], K.G[a][I], s_own);
    const double s_nb = nb_grad_dot<D, I>(K, L, rLi, xn, j);
    double kf[D], knf[D], sk = 0.0, skn = 0.0, q = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kf[mm] = kv[mm + (mm >= I)];
        knf[mm] = pick_facet<D>(kn, mm, j);
        sk += kf[mm];
        skn += knf[mm];
        q = fma(kf[mm], sdu + du[mm], q);
    }
    const double hm = 0.5 * DV * FacetConst<D>::mass;
    // consistency: -1/2 int (k grad u.n + k' grad u'.n) v   ->  +hm (s_own (sk+kf_m) + s_nb (skn+knf_m))
    // adjoint consistency: -1/2 (grad v_a.n) int k jump(u)  ->  +hm G_ai q
    q *= hm;
#pragma unroll
    for (int a = 0; a < NV; ++a) y[a] = fma(K.G[a][I], q, y[a]);
    // penalty: tau/avg(h) int avg(k) jump(u) v
    const double pw = tau * fast_rcp(0.5 * (hK + hN)) * sqG * DV * FacetConst<D>::trip;
    double kb[D], skb = 0.0, skd = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kb[mm] = 0.5 * (kf[mm] + knf[mm]);
        skb += kb[mm];
        skd = fma(kb[mm], du[mm], skd);
    }
    const double base = fma(skb, sdu, skd);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const double t1 = hm * fma(s_own, sk + kf[mm], s_nb * (skn + knf[mm]));
        const double t3 = pw * (base + fma(kb[mm], sdu, du[mm] * fma(2.0, kb[mm], skb)));
        y[mm + (mm >= I)] += t1 + t3;
    }
}

// EMI facet for the geometry-class + LDS-staged kernel: in-block neighbours are read from LDS with the facet-vertex
// permutation folded into the per-lane address (no register selects); out-of-block lanes overwrite from global.
template <int D, int I>
__device__ __forceinline__ void emi_facet_cls(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                              const double* xv, const double* kv,
                                              const double* __restrict__ x, const double* __restrict__ kappa,
                                              double C_phi, double tau, const StageView<D>& st, double* y) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    const unsigned loc0 = (unsigned)(Kp - st.c0);
    const bool in_block = loc0 < st.nvalid;
    const unsigned loc = in_block ? loc0 : 0u;
    const lds_double* xl = st.x + loc * NV;
    const lds_double* kl = st.k + loc * NV;
    double xf[D], knf[D], xap;
    xap = xl[in_block ? j : 0];
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const int o = mm + ((in_block && mm >= j) ? 1 : 0);
        xf[mm] = xl[o];
        knf[mm] = kl[o];
    }
    if (!in_block) {
        static_assert(D == 3, "classed kernels are 3D only");
        const double* px = x + Kp * NV;
        const double* pk = kappa + Kp * NV;
        const double2 q0 = *reinterpret_cast<const double2*>(px), q1 = *reinterpret_cast<const double2*>(px + 2);
        const double2 r0 = *reinterpret_cast<const double2*>(pk), r1 = *reinterpret_cast<const double2*>(pk + 2);
        xf[0] = (j == 0) ? q0.y : q0.x;  knf[0] = (j == 0) ? r0.y : r0.x;
        xf[1] = (j <= 1) ? q1.x : q0.y;  knf[1] = (j <= 1) ? r1.x : r0.y;
        xf[2] = (j <= 2) ? q1.y : q1.x;  knf[2] = (j <= 2) ? r1.y : r1.x;
        xap = (j & 2) ? ((j & 1) ? q1.y : q1.x) : ((j & 1) ? q0.y : q0.x);
    }
    double du[D], sdu = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        du[mm] = xv[mm + (mm >= I)] - xf[mm];
        sdu += du[mm];
    }
    const double DV = (double)D * K.vol;
    const lds_double* ft = st.lext + 8 * I;                                // class-level coefficients (MeshDev::cls_ext)
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * ft[6] * FacetConst<D>::mass;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[mm + (mm >= I)] = fma(w, sdu + du[mm], y[mm + (mm >= I)]);
        return;
    }
    double s_own = 0.0, s_nb = xap * ft[0];
#pragma unroll
    for (int a = 0; a < NV; ++a) s_own = fma(xv[a], K.G[a][I], s_own);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], ft[1 + mm], s_nb);
    double kf[D], sk = 0.0, skn = 0.0, q = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kf[mm] = kv[mm + (mm >= I)];
        sk += kf[mm];
        skn += knf[mm];
        q = fma(kf[mm], sdu + du[mm], q);
    }
    const double hm = 0.5 * DV * FacetConst<D>::mass;
    q *= hm;
#pragma unroll
    for (int a = 0; a < NV; ++a) y[a] = fma(K.G[a][I], q, y[a]);
    const double pw = tau * ft[4] * FacetConst<D>::trip;
    double kb[D], skb = 0.0, skd = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kb[mm] = 0.5 * (kf[mm] + knf[mm]);
        skb += kb[mm];
        skd = fma(kb[mm], du[mm], skd);
    }
    const double base = fma(skb, sdu, skd);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const double t1 = hm * fma(s_own, sk + kf[mm], s_nb * (skn + knf[mm]));
        const double t3 = pw * (base + fma(kb[mm], sdu, du[mm] * fma(2.0, kb[mm], skb)));
        y[mm + (mm >= I)] += t1 + t3;
    }
}

template <int D, int MODE>
__device__ __forceinline__ void emi_cell(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                         const double* xv, const double* kv, double hK,
                                         const double* __restrict__ x, const double* __restrict__ kappa,
                                         double C_phi, double tau, double* y) {
    constexpr int NV = D + 1;
    double kbar = 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) kbar += kv[a];
    kbar *= K.vol / (double)NV;
#pragma unroll
    for (int a = 0; a < NV; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < NV; ++b) s = fma(K.G[a][b], xv[b], s);
        y[a] = kbar * s;
    }
    emi_facet<D, 0, MODE>(m, K, nb, flags, xv, kv, hK, x, kappa, C_phi, tau, y);
    emi_facet<D, 1, MODE>(m, K, nb, flags, xv, kv, hK, x, kappa, C_phi, tau, y);
    emi_facet<D, 2, MODE>(m, K, nb, flags, xv, kv, hK, x, kappa, C_phi, tau, y);
    if (D == 3) emi_facet<D, (D == 3 ? 3 : 0), MODE>(m, K, nb, flags, xv, kv, hK, x, kappa, C_phi, tau, y);
}

// classed + LDS-staged: the class table and the workgroup's own x / kappa live in LDS, so in-block neighbours
// (~83 % under the Morton ordering) cost ds_reads instead of per-lane L1 gathers (the texture addresser, not HBM,
// is what saturates first in the direct variants: TA_BUSY ~75-90 %).
#define CLS_MAX_LDS 32
template <int D, int BLK>
__global__ __launch_bounds__(BLK) void k_emi_apply_cls_staged(MeshDev m, const double* __restrict__ x,
                                                              const double* __restrict__ kappa, double* __restrict__ y,
                                                              double C_phi, double tau) {
    constexpr int NV = D + 1;
    __shared__ __attribute__((aligned(16))) double s_x[BLK * NV];
    __shared__ __attribute__((aligned(16))) double s_k[BLK * NV];
    __shared__ __attribute__((aligned(16))) double s_tab[CLS_MAX_LDS * 11];                 // vol + Gram per class
    __shared__ __attribute__((aligned(16))) double s_ext[CLS_MAX_LDS * (KNP_CLS_EXT + 1)];  // derived facet coefficients, odd stride
    const int64_t c0 = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * BLK;
    if (c0 >= m.c_end) return;
    const int64_t c = c0 + threadIdx.x;
    const bool valid = c < m.c_end;
    for (int i = threadIdx.x; i < m.ncls * 11; i += BLK) s_tab[i] = m.cls_table[(i / 11) * KNP_CLS_STRIDE + (i % 11)];
    for (int i = threadIdx.x; i < m.ncls * KNP_CLS_EXT; i += BLK) s_ext[(i / KNP_CLS_EXT) * (KNP_CLS_EXT + 1) + (i % KNP_CLS_EXT)] = m.cls_ext[i];
    int nb[NV];
    uint32_t flags = 0;
    unsigned cls = 0;
    double xv[NV], kv[NV], yv[NV];
    if (valid) {
        load_cell_ints<D>(m.nbr, c, nb);
        flags = m.fflag[c];
        cls = m.cls[c];
        load_nodal<D>(x, c, xv);
        load_nodal<D>(kappa, c, kv);
        const unsigned t = threadIdx.x;
#pragma unroll
        for (int a = 0; a < NV; ++a) { s_x[t * NV + a] = xv[a]; s_k[t * NV + a] = kv[a]; }
    }
    __syncthreads();
    if (!valid) return;
    const lds_double* rec = TO_LDS(s_tab) + cls * 11;
    CellGeom<D> K;
    K.vol = rec[0];
    {
        int q = 1;
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int b = a; b < NV; ++b) { K.G[a][b] = rec[q]; K.G[b][a] = rec[q]; ++q; }
    }
    StageView<D> st{TO_LDS(s_x), TO_LDS(s_k), nullptr, nullptr, c0,
                    (unsigned)((m.c_end - c0 < BLK) ? (m.c_end - c0) : BLK), nullptr, rec, TO_LDS(s_ext) + cls * (KNP_CLS_EXT + 1)};
    {
        double kbar = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) kbar += kv[a];
        kbar *= K.vol / (double)NV;
#pragma unroll
        for (int a = 0; a < NV; ++a) {
            double sa = 0.0;
#pragma unroll
            for (int b = 0; b < NV; ++b) sa = fma(K.G[a][b], xv[b], sa);
            yv[a] = kbar * sa;
        }
    }
    emi_facet_cls<D, 0>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, st, yv);
    emi_facet_cls<D, 1>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, st, yv);
    emi_facet_cls<D, 2>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, st, yv);
    emi_facet_cls<D, 3>(m, K, nb, flags, xv, kv, x, kappa, C_phi, tau, st, yv);
    store_nodal<D>(y, c, yv);
}

// direct variant: every neighbour access is a global (L1/L2) gather
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_emi_apply(MeshDev m, const double* __restrict__ x, const double* __restrict__ kappa, double* __restrict__ y,
                 double C_phi, double tau) {
    constexpr int NV = D + 1;
    const int64_t c = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * KNP_BLOCK + threadIdx.x;
    if (c >= m.c_end) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    double xv[NV], kv[NV], yv[NV];
    load_nodal<D>(x, c, xv);
    load_nodal<D>(kappa, c, kv);
    const double hK = m.h[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    emi_cell<D, 0>(m, K, nb, flags, xv, kv, hK, x, kappa, C_phi, tau, yv);
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
                                                               bjreal* __restrict__ binv, double C_phi, double tau,
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
    const double hK = m.h[c];
    double A[NV][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        double e[NV], col[NV];
#pragma unroll
        for (int a = 0; a < NV; ++a) e[a] = (a == b) ? 1.0 : 0.0;
        emi_cell<D, 1>(m, K, nb, flags, e, kv, hK, nullptr, kappa, C_phi, tau, col);
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
        for (int b = 0; b < NV; ++b) binv[(c * NV + a) * NV + b] = (bjreal)(0.5 * (A[a][b] + A[b][a]));   // exactly symmetric in fp32
}

// ------------------------------------------------------------------------------------------
// KNP: y_k = A_k x_k for all solved species k at once (shared mesh / geometry / phi data)
//   A_k(u,v) = 1/dt int u v + int D grad u.grad v - int_dS0 avg(D grad u).n jump(v)
//            - int_dS0 avg(D grad v).n jump(u) + int_dS0 tau/avg(h) jump(D u) jump(v)
//            + z psi int D u grad(phi).grad v - z psi int_dS0 jump(v) jump(un u),
//   un = max(D grad(phi).n_own, 0).   `gphi[c][a]` = grad(phi)_c . grad(lambda_a) is precomputed
//   once per KNP solve (phi is frozen during the solve):  area * grad(phi).n_i = -D vol gphi_i.
// ------------------------------------------------------------------------------------------

template <int D, int NS, int I, bool DIAG>
__device__ __forceinline__ void knp_facet(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                          const double (*xv)[D + 1], const double* gp, const double* Dk, double hK,
                                          const double* __restrict__ x, const double* __restrict__ gphi,
                                          const double* __restrict__ Dall, const KnpArgs& ka, double (*y)[D + 1]) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind != FK_SIPG) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    double Xo[D], L[NV];
    load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
    const double hN = m.h[Kp];
    const double gp_nb = gphi[Kp * NV + j];
    apex_bary<D>(K, Xo, L);
    const double rLi = fast_rcp(L[I]);
    const double DV = (double)D * K.vol;
    // upwind speeds times area: un*area = D_k max(-gphi_i, 0) D vol ; neighbour: vol' = -L_i vol
    const double up_own = fmax(-gp[I], 0.0) * DV;
    const double up_nb = fmax(-gp_nb, 0.0) * DV * (-L[I]);
    const double penA = ka.tau * fast_rcp(0.5 * (hK + hN)) * fast_sqrt(K.G[I][I]) * DV;
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
        double s_own = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) s_own = fma(xv[k][a], K.G[a][I], s_own);
        const double s_nb = nb_grad_dot<D, I>(K, L, rLi, xn, j);
        const double zp = ka.z[k] * ka.psi;
        // per facet-vertex weight of the mass-like terms:  pen (D u - D' u') - z psi (un u - un' u')
        const double c_own = penA * Dk[k] - zp * Dk[k] * up_own;
        const double c_nb = penA * Dn - zp * Dn * up_nb;
        double sdu = 0.0, w[D], sw = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double xo = xv[k][mm + (mm >= I)];
            const double xnb = pick_facet<D>(xn, mm, j);
            sdu += xo - xnb;
            w[mm] = fma(c_own, xo, -c_nb * xnb);
            sw += w[mm];
        }
        // consistency: +1/2 vol (D s_own + D' s_nb) ; adjoint: +1/2 D G_ai vol sum(du)
        const double t1 = 0.5 * K.vol * fma(Dk[k], s_own, Dn * s_nb);
        const double t2 = 0.5 * Dk[k] * K.vol * sdu;
#pragma unroll
        for (int a = 0; a < NV; ++a) y[k][a] = fma(K.G[a][I], t2, y[k][a]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm)
            y[k][mm + (mm >= I)] += t1 + FacetConst<D>::mass * (sw + w[mm]);
    }
}

template <int D, int NS, bool DIAG>
__device__ __forceinline__ void knp_cell(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                         const double (*xv)[D + 1], const double* gp, const double* Dk, double hK,
                                         const double* __restrict__ x, const double* __restrict__ gphi,
                                         const double* __restrict__ Dall, const KnpArgs& ka, double (*y)[D + 1]) {
    constexpr int NV = D + 1;
    const double mw = ka.inv_dt * K.vol / (double)((D + 1) * (D + 2));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double sx = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) sx += xv[k][a];
        const double drift = ka.z[k] * ka.psi * Dk[k] * K.vol * sx / (double)NV;
        const double dv = Dk[k] * K.vol;
#pragma unroll
        for (int a = 0; a < NV; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < NV; ++b) s = fma(K.G[a][b], xv[k][b], s);
            y[k][a] = fma(mw, sx + xv[k][a], fma(dv, s, drift * gp[a]));
        }
    }
    knp_facet<D, NS, 0, DIAG>(m, K, nb, flags, xv, gp, Dk, hK, x, gphi, Dall, ka, y);
    knp_facet<D, NS, 1, DIAG>(m, K, nb, flags, xv, gp, Dk, hK, x, gphi, Dall, ka, y);
    knp_facet<D, NS, 2, DIAG>(m, K, nb, flags, xv, gp, Dk, hK, x, gphi, Dall, ka, y);
    if (D == 3) knp_facet<D, NS, (D == 3 ? 3 : 0), DIAG>(m, K, nb, flags, xv, gp, Dk, hK, x, gphi, Dall, ka, y);
}

template <int D, int NS>
__global__ __launch_bounds__(KNP_BLOCK) void k_knp_apply(MeshDev m, const double* __restrict__ x,
                                                         const double* __restrict__ gphi,
                                                         const double* __restrict__ Dall, double* __restrict__ yout,
                                                         KnpArgs ka) {
    constexpr int NV = D + 1;
    const int64_t c = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * KNP_BLOCK + threadIdx.x;
    if (c >= m.c_end) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    double xv[NS][NV], y[NS][NV], gp[NV], Dk[NS];
    load_nodal<D>(gphi, c, gp);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        load_nodal<D>(x + (int64_t)k * m.nc * NV, c, xv[k]);
        Dk[k] = Dall[(int64_t)k * m.nc + c];
    }
    const double hK = m.h[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    knp_cell<D, NS, false>(m, K, nb, flags, xv, gp, Dk, hK, x, gphi, Dall, ka, y);
#pragma unroll
    for (int k = 0; k < NS; ++k) store_nodal<D>(yout + (int64_t)k * m.nc * NV, c, y[k]);
}

// ---- geometry-class + LDS-staged KNP variant (structured meshes), see k_emi_apply_cls_staged ----
template <int D, int NS, int BLK> struct KnpStage {
    const lds_double* x;     // [NS][BLK][NV]
    const lds_double* g;     // [BLK][NV]   gphi
    const lds_double* Dd;    // [NS][BLK]
    const lds_double* rec;   // class record of this cell
    int64_t c0;
    unsigned nvalid;
};

template <int D, int NS, int BLK, int I>
__device__ __forceinline__ void knp_facet_cls(const MeshDev& m, const CellGeom<D>& K, const int* nb, uint32_t flags,
                                              const double (*xv)[D + 1], const double* gp, const double* Dk,
                                              const double* __restrict__ x, const double* __restrict__ gphi,
                                              const double* __restrict__ Dall, const KnpArgs& ka,
                                              const KnpStage<D, NS, BLK>& st, double (*y)[D + 1]) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind != FK_SIPG) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    const unsigned loc0 = (unsigned)(Kp - st.c0);
    const bool in_block = loc0 < st.nvalid;
    const unsigned loc = in_block ? loc0 : 0u;
    double L[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) L[a] = st.rec[11 + 6 * I + a];
    const double sqG = st.rec[11 + 6 * I + 4], hinv = st.rec[11 + 6 * I + 5];
    const double gl = st.g[loc * NV + (in_block ? j : 0)];
    double gg = 0.0;
    if (!in_block) gg = gphi[Kp * NV + j];
    const double gp_nb = in_block ? gl : gg;
    const double rLi = fast_rcp(L[I]);
    const double DV = (double)D * K.vol;
    const double up_own = fmax(-gp[I], 0.0) * DV;
    const double up_nb = fmax(-gp_nb, 0.0) * DV * (-L[I]);
    const double penA = ka.tau * hinv * sqG * DV;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        // in-block neighbours: LDS is addressed per lane, so the facet-vertex permutation costs nothing;
        // out-of-block: exec-masked global gather + register selects
        const lds_double* xl = st.x + ((unsigned)k * BLK + loc) * NV;
        double xf[D], xap, Dn;
        xap = xl[in_block ? j : 0];
#pragma unroll
        for (int mm = 0; mm < D; ++mm) xf[mm] = xl[mm + ((in_block && mm >= j) ? 1 : 0)];
        Dn = st.Dd[k * BLK + loc];
        if (!in_block) {
            // conditional overwrite with scalars (no arrays): the runtime-j selects stay v_cndmask
            const double* px = x + (int64_t)k * m.nc * NV + Kp * NV;
            const double2 q0 = *reinterpret_cast<const double2*>(px);
            const double2 q1 = *reinterpret_cast<const double2*>(px + 2);
            const double g0 = q0.x, g1 = q0.y, g2 = q1.x, g3 = q1.y;
            Dn = Dall[(int64_t)k * m.nc + Kp];
            xf[0] = (j == 0) ? g1 : g0;
            xf[1] = (j <= 1) ? g2 : g1;
            xf[2] = (j <= 2) ? g3 : g2;
            xap = (j & 2) ? ((j & 1) ? g3 : g2) : ((j & 1) ? g1 : g0);
        }
        double s_own = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) s_own = fma(xv[k][a], K.G[a][I], s_own);
        const double gr = K.G[I][I] * rLi;
        double s_nb = xap * gr;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], fma(-L[mm + (mm >= I)], gr, K.G[mm + (mm >= I)][I]), s_nb);
        const double zp = ka.z[k] * ka.psi;
        const double c_own = penA * Dk[k] - zp * Dk[k] * up_own;
        const double c_nb = penA * Dn - zp * Dn * up_nb;
        double sdu = 0.0, w[D], sw = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double xo = xv[k][mm + (mm >= I)];
            sdu += xo - xf[mm];
            w[mm] = fma(c_own, xo, -c_nb * xf[mm]);
            sw += w[mm];
        }
        const double t1 = 0.5 * K.vol * fma(Dk[k], s_own, Dn * s_nb);
        const double t2 = 0.5 * Dk[k] * K.vol * sdu;
#pragma unroll
        for (int a = 0; a < NV; ++a) y[k][a] = fma(K.G[a][I], t2, y[k][a]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm)
            y[k][mm + (mm >= I)] += t1 + FacetConst<D>::mass * (sw + w[mm]);
    }
}

template <int D, int NS, int BLK>
__global__ __launch_bounds__(BLK) void k_knp_apply_cls_staged(MeshDev m, const double* __restrict__ x,
                                                              const double* __restrict__ gphi,
                                                              const double* __restrict__ Dall, double* __restrict__ yout,
                                                              KnpArgs ka) {
    constexpr int NV = D + 1;
    __shared__ __attribute__((aligned(16))) double s_x[NS * BLK * NV];
    __shared__ __attribute__((aligned(16))) double s_g[BLK * NV];
    __shared__ double s_D[NS * BLK];
    __shared__ __attribute__((aligned(16))) double s_tab[CLS_MAX_LDS * KNP_CLS_STRIDE];
    const int64_t c0 = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * BLK;
    if (c0 >= m.c_end) return;
    const int64_t c = c0 + threadIdx.x;
    const bool valid = c < m.c_end;
    for (int i = threadIdx.x; i < m.ncls * KNP_CLS_STRIDE; i += BLK) s_tab[i] = m.cls_table[i];
    int nb[NV];
    uint32_t flags = 0;
    unsigned cls = 0;
    double xv[NS][NV], y[NS][NV], gp[NV], Dk[NS];
    if (valid) {
        load_cell_ints<D>(m.nbr, c, nb);
        flags = m.fflag[c];
        cls = m.cls[c];
        load_nodal<D>(gphi, c, gp);
        const unsigned t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            load_nodal<D>(x + (int64_t)k * m.nc * NV, c, xv[k]);
            Dk[k] = Dall[(int64_t)k * m.nc + c];
            s_D[k * BLK + t] = Dk[k];
#pragma unroll
            for (int a = 0; a < NV; ++a) s_x[(k * BLK + t) * NV + a] = xv[k][a];
        }
#pragma unroll
        for (int a = 0; a < NV; ++a) s_g[t * NV + a] = gp[a];
    }
    __syncthreads();
    if (!valid) return;
    const lds_double* rec = TO_LDS(s_tab) + cls * KNP_CLS_STRIDE;
    CellGeom<D> K;
    K.vol = rec[0];
    {
        int q = 1;
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int b = a; b < NV; ++b) { K.G[a][b] = rec[q]; K.G[b][a] = rec[q]; ++q; }
    }
    KnpStage<D, NS, BLK> st{TO_LDS(s_x), TO_LDS(s_g), TO_LDS(s_D), rec, c0,
                            (unsigned)((m.c_end - c0 < BLK) ? (m.c_end - c0) : BLK)};
    // volume terms (same as knp_cell)
    const double mw = ka.inv_dt * K.vol / (double)((D + 1) * (D + 2));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double sx = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) sx += xv[k][a];
        const double drift = ka.z[k] * ka.psi * Dk[k] * K.vol * sx / (double)NV;
        const double dv = Dk[k] * K.vol;
#pragma unroll
        for (int a = 0; a < NV; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < NV; ++b) s = fma(K.G[a][b], xv[k][b], s);
            y[k][a] = fma(mw, sx + xv[k][a], fma(dv, s, drift * gp[a]));
        }
    }
    knp_facet_cls<D, NS, BLK, 0>(m, K, nb, flags, xv, gp, Dk, x, gphi, Dall, ka, st, y);
    knp_facet_cls<D, NS, BLK, 1>(m, K, nb, flags, xv, gp, Dk, x, gphi, Dall, ka, st, y);
    knp_facet_cls<D, NS, BLK, 2>(m, K, nb, flags, xv, gp, Dk, x, gphi, Dall, ka, st, y);
    if (D == 3) knp_facet_cls<D, NS, BLK, (D == 3 ? 3 : 0)>(m, K, nb, flags, xv, gp, Dk, x, gphi, Dall, ka, st, y);
#pragma unroll
    for (int k = 0; k < NS; ++k) store_nodal<D>(yout + (int64_t)k * m.nc * NV, c, y[k]);
}

// ---- halo-staged persistent variants (3D P1, structured meshes) -------------------------------------------------------
// Measured on the staged kernels above (tools/pmc_apply.sh, profiles/r02_pmc_apply_halo.md): 60 % of the wave cycles are
// spent parked, and a probe with the facet arithmetic removed still takes 85 % of the time -- the kernels are bound by their
// memory phase, which is a CHAIN of dependent round trips (topology -> neighbour rows, inside the facet loop for the 17 %
// of the facets whose neighbour lies outside the workgroup's 256 cells).  For the KNP operator (the EMI twin of this kernel
// measured equal to k_emi_apply_cls_staged, 34.1 vs 33.9 us at r=2 and 277 vs 280 us at r=3, and was removed)
//   * the out-of-block neighbours of every 256-cell block are known in advance (MeshDev::hb_src / hb_loc, built once from
//     the topology): own records and halo records are loaded before the single barrier and the facet loop reads LDS only
//     (one uniform path, the facet-vertex permutation folded into the per-lane LDS address, no register selects);
//   * a workgroup walks several blocks of its XCD's chunk and fetches the NEXT block's halo list while it works on the
//     current one, so that a block's loads -- own and halo -- are one round trip;
//   * LDS is component-major ([component][entry]: consecutive cells on consecutive banks; the row-major layout of the
//     staged kernels spends 70 % of its LDS cycles in bank conflicts), the class table has an odd stride;
//   * D is read through a material table when the cells carry few distinct coefficient tuples (knp_set_params).
// LDS entries [0,256) = the block's cells, [256, 256+nh) = halo entries (one per out-of-block coupled facet).
#define HALO_FT KNP_CLS_EXT   // per-class facet record kept in LDS: 4 x 8 derived coefficients (MeshDev::cls_ext)
#define HALO_FTS 33      // its LDS stride (odd: lanes of different classes land on different banks)

// vol + Gram matrix of the cell's class, straight from the (L1/L2-resident) table into registers
__device__ __forceinline__ void load_class_gram(const double* __restrict__ table, unsigned cls, CellGeom<3>& K) {
    const double* rec = table + (size_t)cls * KNP_CLS_STRIDE;
    K.vol = rec[0];
    int q = 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) { K.G[a][b] = rec[q]; K.G[b][a] = rec[q]; ++q; }
}

// The blocks of one workgroup.  The block range is cut into nq contiguous chunks, nq/8 per XCD (XCD = blockIdx.x & 7,
// round-robin dispatch; an XCD's chunks are adjacent, so facet neighbours stay in its L2); workgroup w serves chunk queue w % nq
// with the other `members` workgroups of that queue.  Full rounds are strided (block = first + member + round * members); the
// remainder (< members blocks) goes to whichever workgroups get there first, through the queue's counter.  A workgroup knows
// its next block one iteration ahead, because that block's halo list is fetched while the current block is worked on.
// counters[2][HALO_NQ] (one 128-byte line each): this launch draws from set `flip` (zero on entry) and zeroes the other one --
// the previous launch's, which the next launch will draw from (launches of one context are ordered on its stream).
#define HALO_NQ 64            // counters per set (upper bound of the queue count nq, a multiple of 8)
#define HALO_CPAD 32          // ints between two counters: atomics on ONE line retire at ~13 ns chip-wide (measured with a draw per
                              // block: 31 104 draws on one line = 392 us, more than the whole kernel)
struct HaloWalk {
    int64_t b_lo, first, last, members, member, dyn0, cur, nxt;
    int n, rounds;
    int* ctr;
    __device__ __forceinline__ int64_t strided(int k) const { return first + member + (int64_t)k * members; }
    __device__ __forceinline__ HaloWalk(const MeshDev& m, int* counters, int flip_nq) {
        const int flip = flip_nq & 1;
        const unsigned nq = (unsigned)flip_nq >> 2;
        b_lo = m.c_begin / KNP_HALO_BLK;
        const int64_t nblk = (m.c_end - 1) / KNP_HALO_BLK - b_lo + 1;
        const int64_t chunk = (nblk + nq - 1) / nq;
        const unsigned q = blockIdx.x % nq;
        first = (int64_t)((q & 7u) * (nq >> 3) + (q >> 3)) * chunk;
        last = first + chunk < nblk ? first + chunk : nblk;
        members = gridDim.x / nq;
        member = blockIdx.x / nq;
        rounds = last > first ? (int)((last - first) / members) : 0;
        if (rounds < 2) rounds = 1 << 30;                                    // short chunks: strided throughout, no draws
        else if (flip_nq & 2) rounds = 2;                                    // default (KNP_HALO_DYN=0 turns it off): every block after the first two is drawn
        dyn0 = first + (int64_t)rounds * members;
        n = 0;
        cur = strided(0);
        nxt = strided(1);
        ctr = counters + (HALO_NQ * flip + q) * HALO_CPAD;
        if (blockIdx.x == 0 && threadIdx.x < HALO_NQ) counters[(HALO_NQ * (1 - flip) + threadIdx.x) * HALO_CPAD] = 0;
    }
    // thread 0, at the top of iteration n: the block after next
    __device__ __forceinline__ int64_t after_next() const { return n + 2 < rounds ? strided(n + 2) : dyn0 + atomicAdd(ctr, 1); }
    // every thread, after the iteration's second barrier
    __device__ __forceinline__ void advance(int64_t nn) { cur = nxt; nxt = nn; ++n; }
};

// s_D: MAT ? [NS][KNP_MAX_MAT] coefficient table indexed by the neighbour's material id dsel : [NS][ent] staged values
template <int NS, bool MAT, int I>
__device__ __forceinline__ void knp_facet_halo(const CellGeom<3>& K, uint32_t flags, unsigned loc, unsigned dsel, const double (*xv)[4],
                                               const double* gp, const double* Dk, const KnpArgs& ka, const lds_double* s_x,
                                               const lds_double* s_g, const lds_double* s_D, const lds_double* ft, unsigned ent,
                                               double (*y)[4]) {
    constexpr int D = 3, NV = 4;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    if (((fb >> 2) & 3u) != FK_SIPG) return;
    const unsigned j = fb & 3u;
    // class-level coefficients (cls_ext): nothing geometric is recomputed per lane
    const double gr = ft[8 * I], pen_geo = ft[8 * I + 4], nLI_DV = ft[8 * I + 5];
    double cf[D];
#pragma unroll
    for (int mm = 0; mm < D; ++mm) cf[mm] = ft[8 * I + 1 + mm];
    const double gp_nb = s_g[loc < KNP_HALO_BLK ? j * KNP_HALO_BLK + loc : loc + (KNP_HALO_BLK * NV - KNP_HALO_BLK)];
    const double DV = (double)D * K.vol;
    const double up_own = fmax(-gp[I], 0.0) * DV;
    const double up_nb = fmax(-gp_nb, 0.0) * nLI_DV;
    const double penA = ka.tau * pen_geo;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const lds_double* xl = s_x + (unsigned)k * NV * ent + loc;                  // component-major: [k][a][entry]
        const double xap = xl[j * ent];
        double xf[D];
#pragma unroll
        for (int mm = 0; mm < D; ++mm) xf[mm] = xl[(mm + (mm >= (int)j ? 1 : 0)) * ent];
        const double Dn = MAT ? s_D[(unsigned)k * KNP_MAX_MAT + dsel] : s_D[(unsigned)k * ent + loc];
        double s_own = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) s_own = fma(xv[k][a], K.G[a][I], s_own);
        double s_nb = xap * gr;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], cf[mm], s_nb);
        const double zp = ka.z[k] * ka.psi;
        const double c_own = penA * Dk[k] - zp * Dk[k] * up_own;
        const double c_nb = penA * Dn - zp * Dn * up_nb;
        double sdu = 0.0, w[D], sw = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double xo = xv[k][mm + (mm >= I)];
            sdu += xo - xf[mm];
            w[mm] = fma(c_own, xo, -c_nb * xf[mm]);
            sw += w[mm];
        }
        const double t1 = 0.5 * K.vol * fma(Dk[k], s_own, Dn * s_nb);
        const double t2 = 0.5 * Dk[k] * K.vol * sdu;
#pragma unroll
        for (int a = 0; a < NV; ++a) y[k][a] = fma(K.G[a][I], t2, y[k][a]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm)
            y[k][mm + (mm >= I)] += t1 + FacetConst<D>::mass * (sw + w[mm]);
    }
}

template <int NS, bool MAT>
__global__ __launch_bounds__(KNP_HALO_BLK) void k_knp_apply_halo(MeshDev m, const double* __restrict__ x,
                                                                 const double* __restrict__ gphi,
                                                                 const double* __restrict__ Dall, double* __restrict__ yout,
                                                                 KnpArgs ka, unsigned ent, const uint8_t* __restrict__ mat,
                                                                 const uint8_t* __restrict__ nmat4, const double* __restrict__ dtab,
                                                                 int* __restrict__ counters, int flip_nq) {
    constexpr int NV = 4, BLK = KNP_HALO_BLK;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_x = smem;                                   // [NS][4][ent]
    double* s_g = s_x + NS * ent * NV;                    // [4][256] own gphi, then [ent - 256] the halo's one component
    double* s_D = s_g + BLK * NV + (ent - BLK);           // MAT: [NS][KNP_MAX_MAT] coefficient table ; else [NS][ent]
    double* s_ft = s_D + (MAT ? NS * KNP_MAX_MAT : NS * ent);   // [ncls][25]
    int* s_draw = reinterpret_cast<int*>(s_ft + m.ncls * HALO_FTS);
    const unsigned t = threadIdx.x;
    HaloWalk w(m, counters, flip_nq);
    if (w.cur >= w.last) return;
    for (int i = t; i < m.ncls * HALO_FT; i += BLK) s_ft[(i / HALO_FT) * HALO_FTS + (i % HALO_FT)] = m.cls_ext[i];
    if (MAT && t < NS * KNP_MAX_MAT) s_D[t] = dtab[t];
    const bool hl = (int)t < m.hb_stride;
    int src = hl ? m.hb_src[(w.b_lo + w.cur) * m.hb_stride + t] : -1;
    while (w.cur < w.last) {
        const int64_t c = (w.b_lo + w.cur) * BLK + t;
        const bool valid = c >= m.c_begin && c < m.c_end;
        const bool stage = c < m.nc;
        double xv[NS][NV], y[NS][NV], gp[NV], Dk[NS];
        if (stage) {
            load_nodal<3>(gphi, c, gp);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                load_nodal<3>(x + (int64_t)k * m.nc * NV, c, xv[k]);
                if (!MAT) Dk[k] = Dall[(int64_t)k * m.nc + c];
            }
        }
        // this thread's halo entry: the list was fetched while the previous block was being worked on
        double2 hq[NS][2];
        double hg = 0.0, hD[NS];
        if (src >= 0) {
            const int64_t Kp = src >> 2;
            hg = gphi[Kp * NV + (src & 3)];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const double2* px = reinterpret_cast<const double2*>(x + (int64_t)k * m.nc * NV + Kp * NV);
                hq[k][0] = px[0];
                hq[k][1] = px[1];
                if (!MAT) hD[k] = Dall[(int64_t)k * m.nc + Kp];
            }
        }
        const int src_next = (hl && w.nxt < w.last) ? m.hb_src[(w.b_lo + w.nxt) * m.hb_stride + t] : -1;
        uint32_t flags = 0, nm = 0;
        unsigned cls = 0, mymat = 0;
        uint2 lw = make_uint2(0u, 0u);
        CellGeom<3> K;
        if (valid) {
            flags = m.fflag[c];
            cls = m.cls[c];
            lw = *reinterpret_cast<const uint2*>(m.hb_loc + c * NV);
            if (MAT) {
                mymat = mat[c];
                nm = *reinterpret_cast<const uint32_t*>(nmat4 + c * NV);
            }
            load_class_gram(m.cls_table, cls, K);
        }
        int64_t drawn = 0;
        if (t == 0) drawn = w.after_next();          // behind the iteration's loads: its return does not gate them (in-order counter)
        if (stage) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (!MAT) s_D[(unsigned)k * ent + t] = Dk[k];
#pragma unroll
                for (int a = 0; a < NV; ++a) s_x[((unsigned)k * NV + a) * ent + t] = xv[k][a];
            }
#pragma unroll
            for (int a = 0; a < NV; ++a) s_g[a * BLK + t] = gp[a];
        }
        if (src >= 0) {
            s_g[BLK * NV + t] = hg;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double* dst = s_x + (unsigned)k * NV * ent + BLK + t;
                dst[0] = hq[k][0].x; dst[ent] = hq[k][0].y; dst[2 * ent] = hq[k][1].x; dst[3 * ent] = hq[k][1].y;
                if (!MAT) s_D[(unsigned)k * ent + BLK + t] = hD[k];
            }
        }
        __syncthreads();
        if (valid) {
            if (MAT) {
#pragma unroll
                for (int k = 0; k < NS; ++k) Dk[k] = TO_LDS(s_D)[k * KNP_MAX_MAT + mymat];
            }
            const lds_double* ft = TO_LDS(s_ft) + cls * HALO_FTS;
            const double mw = ka.inv_dt * K.vol / 20.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double sx = 0.0;
#pragma unroll
                for (int a = 0; a < NV; ++a) sx += xv[k][a];
                const double drift = ka.z[k] * ka.psi * Dk[k] * K.vol * sx / (double)NV;
                const double dv = Dk[k] * K.vol;
#pragma unroll
                for (int a = 0; a < NV; ++a) {
                    double s = 0.0;
#pragma unroll
                    for (int bb = 0; bb < NV; ++bb) s = fma(K.G[a][bb], xv[k][bb], s);
                    y[k][a] = fma(mw, sx + xv[k][a], fma(dv, s, drift * gp[a]));
                }
            }
            knp_facet_halo<NS, MAT, 0>(K, flags, lw.x & 0xffffu, nm & 0xffu, xv, gp, Dk, ka, TO_LDS(s_x), TO_LDS(s_g), TO_LDS(s_D), ft, ent, y);
            knp_facet_halo<NS, MAT, 1>(K, flags, lw.x >> 16, (nm >> 8) & 0xffu, xv, gp, Dk, ka, TO_LDS(s_x), TO_LDS(s_g), TO_LDS(s_D), ft, ent, y);
            knp_facet_halo<NS, MAT, 2>(K, flags, lw.y & 0xffffu, (nm >> 16) & 0xffu, xv, gp, Dk, ka, TO_LDS(s_x), TO_LDS(s_g), TO_LDS(s_D), ft, ent, y);
            knp_facet_halo<NS, MAT, 3>(K, flags, lw.y >> 16, nm >> 24, xv, gp, Dk, ka, TO_LDS(s_x), TO_LDS(s_g), TO_LDS(s_D), ft, ent, y);
#pragma unroll
            for (int k = 0; k < NS; ++k) store_nodal<3>(yout + (int64_t)k * m.nc * NV, c, y[k]);
        }
        if (t == 0) *s_draw = (int)drawn;
        __syncthreads();                   // the next block overwrites the staging
        src = src_next;
        w.advance(*s_draw);
    }
}

// material id of the neighbour behind every facet (once per knp_set_params)
__global__ void k_neighbour_materials(int64_t nc, const int32_t* __restrict__ nbr, const uint8_t* __restrict__ mat, uint8_t* __restrict__ nmat4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc * 4) return;
    const int32_t nb = nbr[i];
    nmat4[i] = nb >= 0 ? mat[nb] : (uint8_t)0;
}

// one species per launch dimension (setup only, once per KNP solve)
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_knp_blockjacobi(MeshDev m, const double* __restrict__ gphi,
                                                               const double* __restrict__ Dall,
                                                               bjreal* __restrict__ binv, KnpArgs ka) {
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
    double gp[NV], Dk[1];
    load_nodal<D>(gphi, c, gp);
    Dk[0] = Dall[(int64_t)k * m.nc + c];
    const double hK = m.h[c];
    KnpArgs k1 = ka;
    k1.z[0] = ka.z[k];
    double A[NV][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        double e[1][NV], col[1][NV];
#pragma unroll
        for (int a = 0; a < NV; ++a) e[0][a] = (a == b) ? 1.0 : 0.0;
        knp_cell<D, 1, true>(m, K, nb, flags, e, gp, Dk, hK, nullptr, gphi, Dall + (int64_t)k * m.nc, k1, col);
#pragma unroll
        for (int a = 0; a < NV; ++a) A[a][b] = col[0][a];
    }
    invert_small<NV>(A);
    bjreal* out = binv + ((int64_t)k * m.nc + c) * NV * NV;
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int b = 0; b < NV; ++b) out[a * NV + b] = (bjreal)A[a][b];
}

// gphi[c][a] = grad(phi)_c . grad(lambda_a) = sum_b phi_b G_ab
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_gphi(MeshDev m, const double* __restrict__ phi, double* __restrict__ out) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc) return;
    int verts[NV];
    load_cell_ints<D>(m.cells, c, verts);
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double pv[NV], s[NV];
    load_nodal<D>(phi, c, pv);
#pragma unroll
    for (int a = 0; a < NV; ++a) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < NV; ++b) t = fma(K.G[a][b], pv[b], t);
        s[a] = t;
    }
    store_nodal<D>(out, c, s);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int64_t grid_for(int64_t n) { return (n + KNP_BLOCK - 1) / KNP_BLOCK; }
static inline int64_t grid8(int64_t n) { return ((grid_for(n) + 7) / 8) * 8; }

// event pair around one apply launch while knp_apply_timing is on (the in-solver figure next to knp_bench_apply's)
struct ApplyTimerScope {
    knp_ctx* c; int which; bool on;
    ApplyTimerScope(knp_ctx* ctx, int w) : c(ctx), which(w), on(ctx->time_applies) {
        if (!on) return;
        auto& pool = c->tev[which];
        if (c->tev_used[which] == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            pool.push_back({a, b});
        }
        hipEventRecord(pool[c->tev_used[which]].first, c->stream);
    }
    ~ApplyTimerScope() {
        if (!on) return;
        hipEventRecord(c->tev[which][c->tev_used[which]].second, c->stream);
        ++c->tev_used[which];
    }
};

static int emi_apply_impl(knp_ctx* c, const double* x, const double* kappa, double* y);
static int knp_apply_impl(knp_ctx* c, const double* x, const double* gphi, double* y);

int launch_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y) {
    ApplyTimerScope t(c, 0);
    return emi_apply_impl(c, x, kappa, y);
}

int launch_knp_apply(knp_ctx* c, const double* x, const double* gphi, double* y) {
    ApplyTimerScope t(c, 1);
    return knp_apply_impl(c, x, gphi, y);
}

// halo-staged kernels: usable when the class and halo tables exist and the block's LDS footprint stays below 64 KB;
// KNP_APPLY_HALO=0 selects the previous staged kernels (A/B runs)
static int env_int(const char* name, int dflt) {            // read per launch (tests switch variants inside one process)
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static bool halo_enabled() { return env_int("KNP_APPLY_HALO", 1) != 0; }
static unsigned halo_entries(const knp_ctx* c) { return (unsigned)(KNP_HALO_BLK + c->m.hb_stride); }
// block counters of the persistent kernels: per operator two sets of 8 that swap roles at every launch (the kernel zeroes the
// set of the launch before it; launches of one context are ordered on its stream)
static int halo_queues() {
    const int v = (env_int("KNP_HALO_NQ", 8) / 8) * 8;
    return v < 8 ? 8 : (v > HALO_NQ ? HALO_NQ : v);
}
static int* halo_counters(knp_ctx* c, int which, int* flip_nq) {
    c->halo_flip[which] ^= 1;
    const int dyn = env_int("KNP_HALO_DYN", 1) ? 2 : 0;          // default: drawn (measured: -3..7 % at 8 M cells)
    *flip_nq = c->halo_flip[which] | dyn | (halo_queues() << 2);
    return c->halo_ctr + which * 2 * HALO_NQ * HALO_CPAD;
}
// persistent grid: as many workgroups as fit on the chip at once (a multiple of the 64 chunk queues), at most one per block;
// KNP_HALO_WG_PER_CU overrides the occupancy query (tuning)
// reserve_cus: with an active communicator the interior launch runs next to the halo exchange (pack kernel + RCCL's send / receive
// kernels on the high-priority halo stream, comm.hip): a persistent grid that occupies every CU would leave them nothing to run on
// until its first workgroups retire, so it is sized for (CUs - reserve_cus).
template <typename KernelT> static dim3 halo_grid(const MeshDev& m, int device, KernelT kernel, size_t lds, int reserve_cus) {
    const int64_t nb = (m.c_end - 1) / KNP_HALO_BLK - m.c_begin / KNP_HALO_BLK + 1;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    int per_cu = env_int("KNP_HALO_WG_PER_CU", 0);
    if (per_cu <= 0) {
        static std::map<std::pair<const void*, size_t>, int> cache;            // one occupancy query per kernel instance and LDS size
        const auto key = std::make_pair((const void*)kernel, lds);
        auto it = cache.find(key);
        if (it == cache.end()) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, KNP_HALO_BLK, lds) != hipSuccess || n < 1) n = 2;
            it = cache.emplace(key, n).first;
        }
        per_cu = it->second;
    }
    if (per_cu < 1) per_cu = 1;
    const int64_t nq = halo_queues();
    const int cus = std::max(ncu - std::max(reserve_cus, 0), ncu / 2);
    int64_t g = std::min<int64_t>(((nb + nq - 1) / nq) * nq, (int64_t)per_cu * cus);
    g = std::max<int64_t>(nq, (g / nq) * nq);
    return dim3((unsigned)g);
}

static int emi_apply_impl(knp_ctx* c, const double* x, const double* kappa, double* y) {
    if (c->degree != 1) return p2_assembled() ? tab_apply(c, 0, x, y) : p2_emi_apply(c, x, kappa, y);
    if (c->m.c_end - c->m.c_begin <= 0) return 0;
    MeshDev m = c->m;                          // the cell range of this launch (the ring-staged kernel may take only its front part)
    if (ring_usable(c, 0) && m.c_begin < c->m.hb_long0 * KNP_HALO_BLK) {
        m.c_end = std::min<int64_t>(c->m.c_end, c->m.hb_long0 * KNP_HALO_BLK);
        const int rc = ring_emi_apply(c, m, x, kappa, y, (c->dist && c->halo_stream) ? env_int("KNP_HALO_RESERVE_CU", 8) : 0);
        if (rc || m.c_end >= c->m.c_end) return rc;
        m.c_begin = m.c_end;                   // blocks with long neighbour lists (a partition's cut cells): LDS-staged kernel below
        m.c_end = c->m.c_end;
    } else if (const int64_t ucells = ring_u_cells(c, 0); ucells > m.c_begin) {
        m.c_end = std::min<int64_t>(c->m.c_end, ucells);          // meshes without geometry classes: ring-staged, geometry from staged coordinates
        const int rc = ring_u_emi_apply(c, m, x, kappa, y, (c->dist && c->halo_stream) ? env_int("KNP_HALO_RESERVE_CU", 8) : 0);
        if (rc || m.c_end >= c->m.c_end) return rc;
        m.c_begin = m.c_end;                   // blocks beyond the staging limits: coordinate-path kernel below
        m.c_end = c->m.c_end;
    }
    const int64_t n = m.c_end - m.c_begin;
    const dim3 g((unsigned)grid8(n)), b(KNP_BLOCK);
    if (c->m.cls && c->m.dim == 3 && c->m.ncls <= CLS_MAX_LDS)
        hipLaunchKernelGGL((k_emi_apply_cls_staged<3, 256>), g, b, 0, c->stream, m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    else if (c->m.dim == 3)
        hipLaunchKernelGGL(k_emi_apply<3>, g, b, 0, c->stream, m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    else
        hipLaunchKernelGGL(k_emi_apply<2>, g, b, 0, c->stream, m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// the halo-staged persistent KNP kernel is usable when the class and halo tables exist, at most two species are solved and
// the block's LDS footprint stays below 64 KB
static bool knp_halo_usable(const knp_ctx* c, size_t* lds_bytes, bool* with_materials) {
    if (c->degree != 1 || c->m.dim != 3 || !c->m.cls || !c->m.hb_stride || !c->halo_ctr || c->p.n_sys > 2 || !halo_enabled()) return false;
    const bool matp = env_int("KNP_APPLY_MAT", 1) != 0 && c->nmat > 0;
    const size_t ns = (size_t)c->p.n_sys, ent = halo_entries(c);
    const size_t lds = sizeof(double) * (ns * ent * 4 + KNP_HALO_BLK * 4 + (ent - KNP_HALO_BLK) + (matp ? ns * KNP_MAX_MAT : ns * ent) +
                                         (size_t)c->m.ncls * HALO_FTS + 1);
    if (lds_bytes) *lds_bytes = lds;
    if (with_materials) *with_materials = matp;
    return lds <= 65536;
}

// which kernel an operator apply runs (bench.py / tests name the kernel they measured): 0 coordinate path, 1 geometry classes +
// LDS staging, 2 halo-staged persistent (+ 4 when D comes from the material table), 3 ring-staged (EMI) / 7 ring-staged (KNP,
// material table), 8 matrix-free P2, 9 assembled P2 blocks, 10 ring-staged without geometry classes (apply_ring_u.hip)
int apply_variant(knp_ctx* c, int which) {
    if (c->degree != 1) return p2_assembled() ? 9 : 8;
    bool matp = false;
    if ((which == 0 || which == 1) && ring_usable(c, which)) return which == 1 ? 7 : 3;
    if ((which == 0 || which == 1) && ring_u_cells(c, which) > 0) return 10;
    if (which == 1 && knp_halo_usable(c, nullptr, &matp)) return matp ? 6 : 2;
    if (c->m.dim == 3 && c->m.cls && c->m.ncls <= CLS_MAX_LDS && (which == 0 || c->p.n_sys <= 3)) return 1;
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

template <int D> static int knp_apply_dispatch(knp_ctx* c, const double* x, const double* gphi, double* y) {
    if (c->m.c_end - c->m.c_begin <= 0) return 0;
    const dim3 b(KNP_BLOCK);
    const KnpArgs ka = make_knp_args(c);
    size_t lds = 0;
    bool matp = false;
    MeshDev m = c->m;                          // the cell range of this launch (the halo-staged kernel may take only its front part)
    if (D == 3 && ring_usable(c, 1) && m.c_begin < c->m.hb_long0 * KNP_HALO_BLK) {
        m.c_end = std::min<int64_t>(c->m.c_end, c->m.hb_long0 * KNP_HALO_BLK);
        const int reserve = (c->dist && c->halo_stream) ? env_int("KNP_HALO_RESERVE_CU", 8) : 0;
        const int rc = ring_knp_apply(c, m, x, gphi, y, ka, reserve);
        if (rc || m.c_end >= c->m.c_end) return rc;
        m.c_begin = m.c_end;
        m.c_end = c->m.c_end;
    } else if (const int64_t ucells = (D == 3 ? ring_u_cells(c, 1) : 0); ucells > m.c_begin) {
        m.c_end = std::min<int64_t>(c->m.c_end, ucells);
        const int reserve = (c->dist && c->halo_stream) ? env_int("KNP_HALO_RESERVE_CU", 8) : 0;
        const int rc = ring_u_knp_apply(c, m, x, gphi, y, ka, reserve);
        if (rc || m.c_end >= c->m.c_end) return rc;
        m.c_begin = m.c_end;
        m.c_end = c->m.c_end;
    } else if (D == 3 && knp_halo_usable(c, &lds, &matp) && m.c_begin < c->m.hb_long0 * KNP_HALO_BLK) {
        const unsigned ent = halo_entries(c);
        const dim3 hb(KNP_HALO_BLK);
        int flip_nq = 0;
        int* ctr = halo_counters(c, 1, &flip_nq);
        m.c_end = std::min<int64_t>(c->m.c_end, c->m.hb_long0 * KNP_HALO_BLK);
        const int reserve = (c->dist && c->halo_stream) ? env_int("KNP_HALO_RESERVE_CU", 8) : 0;
#define KNP_HALO_LAUNCH(NS_, MAT_)                                                                                                   \
    hipLaunchKernelGGL((k_knp_apply_halo<NS_, MAT_>), halo_grid(m, c->device, k_knp_apply_halo<NS_, MAT_>, lds, reserve), hb, lds, c->stream, m, x,   \
                       gphi, c->D, y, ka, ent, (const uint8_t*)c->mat, (const uint8_t*)c->nmat4, (const double*)c->dtab, ctr, flip_nq)
        if (c->p.n_sys == 1) { if (matp) KNP_HALO_LAUNCH(1, true); else KNP_HALO_LAUNCH(1, false); }
        else if (matp) KNP_HALO_LAUNCH(2, true);
        else KNP_HALO_LAUNCH(2, false);
#undef KNP_HALO_LAUNCH
        HIPCHK(c, hipGetLastError());
        if (m.c_end >= c->m.c_end) return 0;
        m.c_begin = m.c_end;                   // blocks with long neighbour lists (a partition's cut cells): LDS-staged kernel below
        m.c_end = c->m.c_end;
    }
    const int64_t nrest = m.c_end - m.c_begin;
    const dim3 gr((unsigned)grid8(nrest));
    if (D == 3 && c->m.cls && c->m.ncls <= CLS_MAX_LDS && c->p.n_sys <= 3) {
        switch (c->p.n_sys) {
            case 1: hipLaunchKernelGGL((k_knp_apply_cls_staged<3, 1, 256>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
            case 2: hipLaunchKernelGGL((k_knp_apply_cls_staged<3, 2, 256>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
            default: hipLaunchKernelGGL((k_knp_apply_cls_staged<3, 3, 256>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
        }
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    switch (c->p.n_sys) {
        case 1: hipLaunchKernelGGL((k_knp_apply<D, 1>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
        case 2: hipLaunchKernelGGL((k_knp_apply<D, 2>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
        case 3: hipLaunchKernelGGL((k_knp_apply<D, 3>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
        case 4: hipLaunchKernelGGL((k_knp_apply<D, 4>), gr, b, 0, c->stream, m, x, gphi, c->D, y, ka); break;
        default: c->err = "knp_apply supports 1..4 solved species"; return -1;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int knp_apply_impl(knp_ctx* c, const double* x, const double* gphi, double* y) {
    if (c->degree != 1) return p2_assembled() ? tab_apply(c, 1, x, y) : p2_knp_apply(c, x, gphi, y);     // P2: gphi holds phi (launch_dnphi)
    return c->m.dim == 3 ? knp_apply_dispatch<3>(c, x, gphi, y) : knp_apply_dispatch<2>(c, x, gphi, y);
}

int launch_emi_blockjacobi(knp_ctx* c, const double* kappa, bjreal* binv) {
    if (c->degree != 1) return p2_assembled() ? tab_block_inverse(c, 0, binv) : p2_block_inverse(c, 0, kappa, binv);
    const dim3 g((unsigned)grid_for(c->m.nc_owned)), b(KNP_BLOCK);
    const double shift = 0.0;
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_emi_blockjacobi<3>, g, b, 0, c->stream, c->m, kappa, binv, c->p.C_phi, c->p.tau_emi, shift);
    else
        hipLaunchKernelGGL(k_emi_blockjacobi<2>, g, b, 0, c->stream, c->m, kappa, binv, c->p.C_phi, c->p.tau_emi, shift);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int launch_knp_blockjacobi(knp_ctx* c, const double* gphi, bjreal* binv) {
    if (c->degree != 1) return p2_assembled() ? tab_block_inverse(c, 1, binv) : p2_block_inverse(c, 1, gphi, binv);
    const dim3 g((unsigned)grid_for(c->m.nc_owned), (unsigned)c->p.n_sys), b(KNP_BLOCK);
    const KnpArgs ka = make_knp_args(c);
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_knp_blockjacobi<3>, g, b, 0, c->stream, c->m, gphi, c->D, binv, ka);
    else
        hipLaunchKernelGGL(k_knp_blockjacobi<2>, g, b, 0, c->stream, c->m, gphi, c->D, binv, ka);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int launch_neighbour_materials(knp_ctx* c) {
    const int64_t n = c->m.nc * 4;
    hipLaunchKernelGGL(k_neighbour_materials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->m.nc, c->m.nbr, c->mat, c->nmat4);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int launch_dnphi(knp_ctx* c, const double* phi, double* gphi) {
    if (c->degree != 1) {
        // P2: the matrix-free apply evaluates the drift from phi itself; the "derived" field keeps the potential the KNP
        // solve is frozen at (the assembled variant integrates it into the cell blocks instead)
        if (p2_assembled()) return tab_assemble_knp(c, phi);
        HIPCHK(c, hipMemcpyAsync(gphi, phi, sizeof(double) * c->m.nc * c->nd, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    const dim3 g((unsigned)grid_for(c->m.nc)), b(KNP_BLOCK);
    if (c->m.dim == 3)
        hipLaunchKernelGGL(k_gphi<3>, g, b, 0, c->stream, c->m, phi, gphi);
    else
        hipLaunchKernelGGL(k_gphi<2>, g, b, 0, c->stream, c->m, phi, gphi);
    HIPCHK(c, hipGetLastError());
    return 0;
}
