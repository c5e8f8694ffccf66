// Per-step (not per-Krylov-iteration) P1 kernels: coefficient update, right-hand sides,
// facet projections and the eliminated-ion update.
//
// Replaces: dolfin.assemble(L_emi)   reference: src/knpemidg/solver.py:309-310,330-344,478
//           dolfin.assemble(L_knp)   reference: src/knpemidg/solver.py:597-629,731
//           kappa / alpha_sum        reference: src/knpemidg/solver.py:303-306
//           pcws_constant_project    reference: src/knpemidg/utils.py:100-124
//           step-III updates         reference: src/knpemidg/solver.py:808-845
#include "cell_geom.hpp"

// Facet quadrature for the two NON-polynomial integrands of the path.  Same published rules
// (and the same 15-digit constants) as oracle/quadrature.py: Strang-Fix 6-point (degree 4) and
// 7-point (degree 5) rules on triangles, 3-point Gauss-Legendre on intervals.
template <int D, int DEG> struct FacetRule;
template <> struct FacetRule<3, 5> {
    static constexpr int nq = 7;
    __device__ static void get(int q, double* mu, double& w) {
        constexpr double a1 = 0.797426985353087, b1 = 0.101286507323456, w1 = 0.125939180544827;
        constexpr double a2 = 0.059715871789770, b2 = 0.470142064105115, w2 = 0.132394152788506;
        if (q == 0) { mu[0] = mu[1] = mu[2] = 1.0 / 3.0; w = 0.225; return; }
        const int r = (q - 1) % 3;
        const bool first = q <= 3;
        const double a = first ? a1 : a2, b = first ? b1 : b2;
        w = first ? w1 : w2;
        mu[0] = (r == 0) ? a : b; mu[1] = (r == 1) ? a : b; mu[2] = (r == 2) ? a : b;
    }
};
template <> struct FacetRule<3, 4> {
    static constexpr int nq = 6;
    __device__ static void get(int q, double* mu, double& w) {
        constexpr double a1 = 0.816847572980459, b1 = 0.091576213509771, w1 = 0.109951743655322;
        constexpr double a2 = 0.108103018168070, b2 = 0.445948490915965, w2 = 0.223381589678011;
        const int r = q % 3;
        const bool first = q < 3;
        const double a = first ? a1 : a2, b = first ? b1 : b2;
        w = first ? w1 : w2;
        mu[0] = (r == 0) ? a : b; mu[1] = (r == 1) ? a : b; mu[2] = (r == 2) ? a : b;
    }
};
template <int DEG> struct FacetRule<2, DEG> {       // degree 4 and 5 both need 3 Gauss points
    static constexpr int nq = 3;
    __device__ static void get(int q, double* mu, double& w) {
        constexpr double xi = 0.7745966692414834;
        const double x = (q == 0) ? 0.5 * (1.0 - xi) : ((q == 1) ? 0.5 : 0.5 * (1.0 + xi));
        w = (q == 1) ? 0.5 * 0.8888888888888888 : 0.5 * 0.5555555555555556;
        mu[0] = 1.0 - x; mu[1] = x;
    }
};

struct IonArgs {
    int n;                    // total species (last eliminated)
    double z[KNP_MAX_IONS];
};

// kappa = sum_k F z_k^2 psi D_k c_k ; asum = sum_k D_k z_k^2 c_k      (all N ions)
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_kappa(MeshDev m, const double* __restrict__ cc,
                                                     const double* __restrict__ celim, const double* __restrict__ Dall,
                                                     double* __restrict__ kappa, IonArgs ia, double F, double psi) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc) return;
    double k[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) k[a] = 0.0;
    for (int i = 0; i < ia.n; ++i) {
        double cv[NV];
        load_nodal<D>((i < ia.n - 1) ? cc + (int64_t)i * m.nc * NV : celim, c, cv);
        const double f = F * ia.z[i] * ia.z[i] * psi * Dall[(int64_t)i * m.nc + c];
#pragma unroll
        for (int a = 0; a < NV; ++a) k[a] += f * cv[a];
    }
    store_nodal<D>(kappa, c, k);
}

// ------------------------------------------------------------------------------------------
// L_emi
// ------------------------------------------------------------------------------------------
template <int D, int I>
__device__ __forceinline__ void emi_rhs_facet(const MeshDev& m, const CellGeom<D>& K, int64_t c, const int* nb,
                                              uint32_t flags, const double* __restrict__ cc,
                                              const double* __restrict__ celim, const double* __restrict__ Dall,
                                              const double* __restrict__ phiM, const double* __restrict__ Ich,
                                              const IonArgs& ia, double F, double C_phi, int splitting, double* b) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    if (kind == FK_MEMBRANE) {
        const int64_t f = m.cfacet[c * NV + I];
        if (splitting == 2) return;                                 // MMS: Robin data arrives as host-integrated extra RHS
        double g = phiM[f];
        if (!splitting) {
            double It = 0.0;
            for (int i = 0; i < ia.n; ++i) It += Ich[(int64_t)i * m.nf + f];
            g -= It / C_phi;
        }
        const double sgn = ((fb >> 4) & 1u) ? -1.0 : 1.0;           // JUMP(v) = v_i - v_e
        const double w = sgn * C_phi * g * fast_sqrt(K.G[I][I]) * K.vol;   // area / D = sqrt(G_ii) vol
#pragma unroll
        for (int mm = 0; mm < D; ++mm) b[mm + (mm >= I)] += w;
        return;
    }
    double Xo[D], L[NV];
    load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
    apex_bary<D>(K, Xo, L);
    const double rLi = fast_rcp(L[I]);
    const double gr = K.G[I][I] * rLi;
    double flux = 0.0;
    for (int i = 0; i < ia.n; ++i) {
        const double* src = (i < ia.n - 1) ? cc + (int64_t)i * m.nc * NV : celim;
        double cv[NV], cn[NV];
        load_nodal<D>(src, c, cv);
        load_nodal<D>(src, Kp, cn);
        double s_own = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) s_own = fma(cv[a], K.G[a][I], s_own);
        double s_nb = pick_apex<D>(cn, j) * gr;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const int a = mm + (mm >= I);
            s_nb = fma(pick_facet<D>(cn, mm, j), fma(-L[a], gr, K.G[a][I]), s_nb);
        }
        flux += F * ia.z[i] * 0.5 * (Dall[(int64_t)i * m.nc + c] * s_own + Dall[(int64_t)i * m.nc + Kp] * s_nb);
    }
    // area/D * (grad c . n) = -vol (grad c . g_i)
    const double w = -flux * K.vol;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) b[mm + (mm >= I)] += w;
}

template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_emi_rhs(MeshDev m, const double* __restrict__ cc,
                                                       const double* __restrict__ celim, const double* __restrict__ Dall,
                                                       const double* __restrict__ phiM, const double* __restrict__ Ich,
                                                       double* __restrict__ bout, IonArgs ia, double F, double C_phi,
                                                       int splitting, const double* __restrict__ extra) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc_owned) return;
    int verts[NV], nb[NV];
    load_cell_ints<D>(m.cells, c, verts);
    load_cell_ints<D>(m.nbr, c, nb);
    const uint32_t flags = m.fflag[c];
    CellGeom<D> K;
    load_cell_geometry<D>(m, verts, K);
    double b[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) b[a] = 0.0;
    for (int i = 0; i < ia.n; ++i) {
        double cv[NV], gc[D];
        load_nodal<D>((i < ia.n - 1) ? cc + (int64_t)i * m.nc * NV : celim, c, cv);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            gc[k] = 0.0;
#pragma unroll
            for (int a = 0; a < NV; ++a) gc[k] += cv[a] * K.g[a][k];
        }
        const double f = -F * ia.z[i] * Dall[(int64_t)i * m.nc + c] * K.vol;
#pragma unroll
        for (int a = 0; a < NV; ++a) b[a] += f * dotD<D>(gc, K.g[a]);
    }
    emi_rhs_facet<D, 0>(m, K, c, nb, flags, cc, celim, Dall, phiM, Ich, ia, F, C_phi, splitting, b);
    emi_rhs_facet<D, 1>(m, K, c, nb, flags, cc, celim, Dall, phiM, Ich, ia, F, C_phi, splitting, b);
    emi_rhs_facet<D, 2>(m, K, c, nb, flags, cc, celim, Dall, phiM, Ich, ia, F, C_phi, splitting, b);
    if (D == 3) emi_rhs_facet<D, (D == 3 ? 3 : 0)>(m, K, c, nb, flags, cc, celim, Dall, phiM, Ich, ia, F, C_phi, splitting, b);
    if (extra) {
        double ev[NV];
        load_nodal<D>(extra, c, ev);
#pragma unroll
        for (int a = 0; a < NV; ++a) b[a] += ev[a];
    }
    store_nodal<D>(bout, c, b);
}

// ------------------------------------------------------------------------------------------
// L_knp   (species k = blockIdx.y)
// ------------------------------------------------------------------------------------------
struct KnpRhsArgs {
    double F, C_M, dt;
    int splitting;
    const double* mms_C;      // [n_sys][nc] in MMS mode
    const double* extra;      // [n_sys][nc*nd] or null
};

template <int D, int I>
__device__ __forceinline__ void knp_rhs_facet(const MeshDev& m, const CellGeom<D>& K, int64_t c, const int* nb,
                                              uint32_t flags, int k, double zk, double Dk, const double* ck,
                                              const double* asum, const double* pv, const double* __restrict__ phi,
                                              const double* __restrict__ phiM, const double* __restrict__ Ich,
                                              const IonArgs& ia, const KnpRhsArgs& ra, double* b) {
    constexpr int NV = D + 1;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind != FK_MEMBRANE) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    const bool is_e = (fb >> 4) & 1u;
    const double area = fast_sqrt(K.G[I][I]) * (double)D * K.vol;
    const int64_t f = m.cfacet[c * NV + I];
    const double pM = phiM[f];
    const double Ik = Ich[(int64_t)k * m.nf + f];
    double It = 0.0;
    for (int i = 0; i < ia.n; ++i) It += Ich[(int64_t)i * m.nf + f];
    double pn[NV];
    load_nodal<D>(phi, Kp, pn);
    double cf[D], af[D], pf[D], pnf[D];
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const int a = mm + (mm >= I);
        cf[mm] = ck[a];
        af[mm] = asum[a];
        pf[mm] = pv[a];
        pnf[mm] = pick_facet<D>(pn, mm, j);
    }
    const double sgn = is_e ? -1.0 : 1.0;
    if (ra.splitting == 2) {
        // MMS (solver.py:649-650): -(phi_i - phi_e)(C_i v_i - C_e v_e) with DG0 C; P1 phi -> exact facet mass matrix
        const double Cown = ra.mms_C[(int64_t)k * m.nc + c];
        double dph[D], sd = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) { dph[mm] = is_e ? (pnf[mm] - pf[mm]) : (pf[mm] - pnf[mm]); sd += dph[mm]; }
        const double w = -sgn * Cown * area * ((D == 3) ? 1.0 / 12.0 : 1.0 / 6.0);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) b[mm + (mm >= I)] += w * (sd + dph[mm]);
        return;
    }
    using Rule = FacetRule<D, 5>;
#pragma unroll
    for (int q = 0; q < Rule::nq; ++q) {
        double mu[D], w;
        Rule::get(q, mu, w);
        double cq = 0.0, aq = 0.0, po = 0.0, pb = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            cq += mu[mm] * cf[mm];
            aq += mu[mm] * af[mm];
            po += mu[mm] * pf[mm];
            pb += mu[mm] * pnf[mm];
        }
        const double alpha = Dk * zk * zk * cq / aq;
        const double C = alpha * ra.C_M / (ra.F * zk * ra.dt);
        double g = pM - ra.dt / (ra.C_M * alpha) * Ik;
        if (ra.splitting) g += (ra.dt / ra.C_M) * It;
        const double dphi = is_e ? (pb - po) : (po - pb);            // phi_i - phi_e
        const double val = w * area * sgn * C * (g - dphi);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) b[mm + (mm >= I)] += val * mu[mm];
    }
}

template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_knp_rhs(MeshDev m, const double* __restrict__ cc,
                                                       const double* __restrict__ cprev, const double* __restrict__ celim,
                                                       const double* __restrict__ phi, const double* __restrict__ Dall,
                                                       const double* __restrict__ phiM, const double* __restrict__ Ich,
                                                       const double* __restrict__ fsrc, double* __restrict__ bout,
                                                       IonArgs ia, KnpRhsArgs ra) {
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
    double cp[NV], b[NV], sp = 0.0;
    load_nodal<D>(cprev + (int64_t)k * m.nc * NV, c, cp);
#pragma unroll
    for (int a = 0; a < NV; ++a) sp += cp[a];
    const double mw = K.vol / (ra.dt * (double)((D + 1) * (D + 2)));
    const double fs = fsrc ? fsrc[(int64_t)k * m.nc + c] * K.vol / (double)NV : 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) b[a] = mw * (sp + cp[a]) + fs;
    // any membrane facet?
    bool has_mem = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) has_mem |= (((flags >> (8 * i + 2)) & 3u) == FK_MEMBRANE);
    if (has_mem) {
        double ck[NV], asum[NV], pv[NV];
        load_nodal<D>(cc + (int64_t)k * m.nc * NV, c, ck);
        load_nodal<D>(phi, c, pv);
#pragma unroll
        for (int a = 0; a < NV; ++a) asum[a] = 0.0;
        for (int i = 0; i < ia.n; ++i) {
            double cv[NV];
            load_nodal<D>((i < ia.n - 1) ? cc + (int64_t)i * m.nc * NV : celim, c, cv);
            const double f = ia.z[i] * ia.z[i] * Dall[(int64_t)i * m.nc + c];
#pragma unroll
            for (int a = 0; a < NV; ++a) asum[a] += f * cv[a];
        }
        const double zk = ia.z[k], Dk = Dall[(int64_t)k * m.nc + c];
        knp_rhs_facet<D, 0>(m, K, c, nb, flags, k, zk, Dk, ck, asum, pv, phi, phiM, Ich, ia, ra, b);
        knp_rhs_facet<D, 1>(m, K, c, nb, flags, k, zk, Dk, ck, asum, pv, phi, phiM, Ich, ia, ra, b);
        knp_rhs_facet<D, 2>(m, K, c, nb, flags, k, zk, Dk, ck, asum, pv, phi, phiM, Ich, ia, ra, b);
        if (D == 3) knp_rhs_facet<D, (D == 3 ? 3 : 0)>(m, K, c, nb, flags, k, zk, Dk, ck, asum, pv, phi, phiM, Ich, ia, ra, b);
    }
    if (ra.extra) {
        double ev[NV];
        load_nodal<D>(ra.extra + (int64_t)k * m.nc * NV, c, ev);
#pragma unroll
        for (int a = 0; a < NV; ++a) b[a] += ev[a];
    }
    store_nodal<D>(bout + (int64_t)k * m.nc * NV, c, b);
}

// ------------------------------------------------------------------------------------------
// step III
// ------------------------------------------------------------------------------------------
// c_N = -(sum_k z_k c_k + rho)/z_N
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_celim(MeshDev m, const double* __restrict__ cc, const double* __restrict__ rho,
                                                     double* __restrict__ celim, IonArgs ia) {
    constexpr int NV = D + 1;
    const int64_t c = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (c >= m.nc) return;
    const double zN = ia.z[ia.n - 1];
    double acc[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) acc[a] = 0.0;
    for (int i = 0; i < ia.n - 1; ++i) {
        double cv[NV];
        load_nodal<D>(cc + (int64_t)i * m.nc * NV, c, cv);
        const double f = -(1.0 / zN) * ia.z[i];
#pragma unroll
        for (int a = 0; a < NV; ++a) acc[a] += f * cv[a];
    }
    const double r = -(1.0 / zN) * rho[c];
#pragma unroll
    for (int a = 0; a < NV; ++a) acc[a] += r;
    store_nodal<D>(celim, c, acc);
}

// per membrane facet: phi_M = avg(phi_i - phi_e); E_k = RT/(F z_k) avg ln(c_e/c_i) for all N ions
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_facet_updates(MeshDev m, const double* __restrict__ cc,
                                                             const double* __restrict__ celim, const double* __restrict__ phi,
                                                             double* __restrict__ phiM, double* __restrict__ E,
                                                             IonArgs ia, double RT_over_F) {
    constexpr int NV = D + 1;
    const int64_t t = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (t >= m.nmf) return;
    const int32_t* row = m.mf + 6 * t;
    if (!row[5]) return;                                   // facet owned by another rank
    const int64_t ce = row[0], ci = row[1];
    const int le = row[2], li = row[3];
    const int64_t f = row[4];
    double ve[NV], vi[NV];
    if (phi) {
        load_nodal<D>(phi, ce, ve);
        load_nodal<D>(phi, ci, vi);
        double s = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) s += pick_facet<D>(vi, mm, li) - pick_facet<D>(ve, mm, le);
        phiM[f] = s / (double)D;
    }
    using Rule = FacetRule<D, 4>;
    for (int i = 0; i < ia.n; ++i) {
        const double* src = (i < ia.n - 1) ? cc + (int64_t)i * m.nc * NV : celim;
        load_nodal<D>(src, ce, ve);
        load_nodal<D>(src, ci, vi);
        double fe[D], fi[D];
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            fe[mm] = pick_facet<D>(ve, mm, le);
            fi[mm] = pick_facet<D>(vi, mm, li);
        }
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < Rule::nq; ++q) {
            double mu[D], w;
            Rule::get(q, mu, w);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int mm = 0; mm < D; ++mm) { a += mu[mm] * fe[mm]; b += mu[mm] * fi[mm]; }
            acc += w * log(a / b);
        }
        E[(int64_t)i * m.nf + f] = RT_over_F / ia.z[i] * acc;
    }
}

// facet average of the plus (side=0, ECS-like) or minus (side=1) trace of a nodal field
template <int D>
__global__ __launch_bounds__(KNP_BLOCK) void k_facet_trace(MeshDev m, const double* __restrict__ nodal, int side,
                                                           double* __restrict__ out) {
    constexpr int NV = D + 1;
    const int64_t t = (int64_t)blockIdx.x * KNP_BLOCK + threadIdx.x;
    if (t >= m.nmf) return;
    const int32_t* row = m.mf + 6 * t;
    if (!row[5]) return;
    const int64_t cell = row[side];
    const int lf = row[2 + side];
    double v[NV];
    load_nodal<D>(nodal, cell, v);
    double s = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) s += pick_facet<D>(v, mm, lf);
    out[row[4]] = s / (double)D;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static IonArgs ion_args(knp_ctx* c) {
    IonArgs ia;
    ia.n = c->p.n_ions;
    for (int i = 0; i < KNP_MAX_IONS; ++i) ia.z[i] = c->p.z[i];
    return ia;
}

#define DISPATCH_DIM(c, KERN, grid, ...)                                                         \
    do {                                                                                         \
        if ((c)->m.dim == 3) hipLaunchKernelGGL(KERN<3>, grid, dim3(KNP_BLOCK), 0, (c)->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERN<2>, grid, dim3(KNP_BLOCK), 0, (c)->stream, __VA_ARGS__);    \
        HIPCHK(c, hipGetLastError());                                                            \
    } while (0)

int launch_kappa(knp_ctx* c, const double* cc, const double* celim, double* kappa) {
    if (c->degree != 1) return tab_kappa(c, cc, celim, kappa);
    DISPATCH_DIM(c, k_kappa, dim3((unsigned)grid_for(c->m.nc)), c->m, cc, celim, c->D, kappa, ion_args(c), c->p.F, c->p.psi);
    return 0;
}

int launch_emi_rhs(knp_ctx* c, const double* cc, const double* celim, const double* phiM, const double* Ich, double* b) {
    if (c->degree != 1) return tab_emi_rhs(c, cc, celim, phiM, Ich, b);
    DISPATCH_DIM(c, k_emi_rhs, dim3((unsigned)grid_for(c->m.nc_owned)), c->m, cc, celim, c->D, phiM, Ich, b,
                 ion_args(c), c->p.F, c->p.C_phi, c->p.splitting, (const double*)c->extra_emi);
    return 0;
}

int launch_knp_rhs(knp_ctx* c, const double* cc, const double* cprev, const double* celim, const double* phi,
                   const double* phiM, const double* Ich, double* b) {
    if (c->degree != 1) return tab_knp_rhs(c, cc, cprev, celim, phi, phiM, Ich, b);
    KnpRhsArgs ra{c->p.F, c->p.C_M, c->p.dt, c->p.splitting, (const double*)c->mms_C, (const double*)c->extra_knp};
    DISPATCH_DIM(c, k_knp_rhs, dim3((unsigned)grid_for(c->m.nc_owned), (unsigned)c->p.n_sys), c->m, cc, cprev, celim, phi,
                 c->D, phiM, Ich, (const double*)c->fsrc, b, ion_args(c), ra);
    return 0;
}

int launch_step_updates(knp_ctx* c, const double* cc, double* celim, const double* phi, double* phiM, double* E) {
    if (c->degree != 1) return tab_step_updates(c, cc, celim, phi, phiM, E, true);
    DISPATCH_DIM(c, k_celim, dim3((unsigned)grid_for(c->m.nc)), c->m, cc, c->rho, celim, ion_args(c));
    if (c->m.nmf > 0)
        DISPATCH_DIM(c, k_facet_updates, dim3((unsigned)grid_for(c->m.nmf)), c->m, cc, (const double*)celim, phi, phiM, E,
                     ion_args(c), c->p.R * c->p.T / c->p.F);
    return 0;
}

int launch_facet_trace(knp_ctx* c, const double* nodal, int side, double* out) {
    if (c->degree != 1) return tab_facet_trace(c, nodal, side, out);
    if (c->m.nmf > 0)
        DISPATCH_DIM(c, k_facet_trace, dim3((unsigned)grid_for(c->m.nmf)), c->m, nodal, side, out);
    return 0;
}

int launch_nernst_only(knp_ctx* c, const double* cc, const double* celim, double* E) {
    if (c->degree != 1) return tab_step_updates(c, cc, const_cast<double*>(celim), nullptr, nullptr, E, false);
    if (c->m.nmf > 0)
        DISPATCH_DIM(c, k_facet_updates, dim3((unsigned)grid_for(c->m.nmf)), c->m, cc, celim, (const double*)nullptr,
                     (double*)nullptr, E, ion_args(c), c->p.R * c->p.T / c->p.F);
    return 0;
}
