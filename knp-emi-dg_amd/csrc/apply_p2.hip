// Matrix-free DG-P2 operator applies (EMI potential operator, KNP species operator) and their cell-diagonal blocks.
//
// Replaces, for Solver(degree_emi=2, degree_knp=2) (reference: tests/run_MMS_space.py:194-195, src/knpemidg/solver.py:157-175):
//   dolfin.assemble(a_emi) + PETSc MatMult   src/knpemidg/solver.py:325-328, 346, 477, 509
//   dolfin.assemble(A_knp) + PETSc MatMult   src/knpemidg/solver.py:586-594, 730, 771
// Round 1 integrated these forms into dense cell blocks once per time step (4 KB per cell and operator, 12 GB at 10^6 tets)
// and streamed them in every Krylov iteration: HBM-bound on 16-20x the algorithmic bytes.  Here nothing is assembled:
//
//  * cells:  grad u is P1, so  int coef grad u . grad v  =  vol sum_{ll'} G_ll' sum_{vv'} W_vv' U_l(v) V_l'(v')  with the nodal
//            gradient values U_l(v) = d u / d lambda_l at vertex v and  W_vv' = sum_b coef_b M3[b][vv']  (reference tensor
//            M3 = int phi_b lambda_v lambda_v', p2_tables.hpp): exact, 300 FMAs per cell instead of a 27-point rule;
//  * facets: everything is evaluated in the FACET FRAME [apex | facet vertices | facet edges | apex edges]: the own cell's
//            frame is a compile-time permutation of its registers, the neighbour's 10 dofs are gathered through a nibble-
//            packed permutation selected by its local facet index, after which both sides share the same facet basis:
//            traces are P2 on the facet (D(D+1)/2 nodes), normal derivatives are P1 (D vertex values), and the integrals
//            run over the facet rule of p2_tables.hpp (degree 6 for a_emi: exact; degree 5 = FIAT's points for a_knp, whose
//            upwind speed |D grad(phi).n| is not a polynomial);
//  * geometry in Gram form (cell_geom.hpp): class records on (block-)structured meshes, vertex coordinates otherwise.
//
// One thread per cell (EMI) / per (cell, species) (KNP); the workgroup's own x and coefficient live in LDS, where ~83 % of the
// facet neighbours of a Morton-ordered block are found.  FP64 vector FMAs: on MI355X the FP64 matrix rate equals the vector
// rate, the facet contractions are 12 x 6 / 7 x 6 (padding to 16 x 16 x 4 MFMA tiles wastes > 40 %), and the sparse P2
// gradient structure (4 of 40 entries per point) is exploited here; see DESIGN.md section 4b for the measured comparison.
// The numpy restatement of exactly this formulation is tests/p2_formulation.py (checked against the oracle on the CPU).
#include "../../include/knpemi_hip.h"
#include "cell_geom.hpp"
#include "p2_tables.hpp"
#include "p2_mono_tables.hpp"
#include <cstdlib>

namespace {

template <int D> struct P2 {
    static constexpr int NV = D + 1, ND = NV * (NV + 1) / 2, NFE = D * (D - 1) / 2, NF = D + NFE;
    static constexpr int edge(int a, int b) {                 // cell dof of edge (a, b)
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        int k = NV;
        for (int i = 0; i < lo; ++i) k += NV - 1 - i;
        return k + (hi - lo - 1);
    }
    static constexpr int fe(int m, int mp) {                  // frame slot of the facet edge (m, mp)
        const int lo = m < mp ? m : mp, hi = m < mp ? mp : m;
        int k = 1 + D;
        for (int i = 0; i < lo; ++i) k += D - 1 - i;
        return k + (hi - lo - 1);
    }
    static constexpr int ae(int m) { return 1 + D + NFE + m; } // frame slot of the edge (apex, facet vertex m)
    // cell dof behind frame slot s of local facet I (= P2Tab<D>::FRAME_SLOTS[I][s], as arithmetic so that it folds to a
    // register index at compile time; loads from the table would be run-time loads + dynamic register indexing)
    static constexpr int fs(int I, int s) {
        if (s == 0) return I;
        if (s <= D) return (s - 1) + ((s - 1) >= I ? 1 : 0);
        if (s <= D + NFE) {
            int k = 1 + D;
            for (int a = 0; a < D; ++a)
                for (int b = a + 1; b < D; ++b) {
                    if (k == s) return edge(a + (a >= I ? 1 : 0), b + (b >= I ? 1 : 0));
                    ++k;
                }
        }
        const int mm = s - 1 - D - NFE;
        return edge(I, mm + (mm >= I ? 1 : 0));
    }
    static constexpr uint64_t packed(int j) {                 // nibble s = fs(j, s)
        uint64_t p = 0;
        for (int s = 0; s < ND; ++s) p |= (uint64_t)fs(j, s) << (4 * s);
        return p;
    }
    static constexpr int pair(int a, int b) {                 // index of the unordered vertex pair in M3
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        int k = 0;
        for (int i = 0; i < lo; ++i) k += NV - i;
        return k + (hi - lo);
    }
};

#define CLS_MAX_P2 32

// U[l][v] = d u / d lambda_l at vertex v
template <int D> __device__ __forceinline__ void nodal_gradients(const double* x, double (*U)[D + 1]) {
#pragma unroll
    for (int v = 0; v <= D; ++v)
#pragma unroll
        for (int l = 0; l <= D; ++l) U[l][v] = (l == v) ? 3.0 * x[v] : fma(4.0, x[P2<D>::edge(v, l)], -x[l]);
}

// y[a] (+)= sum_{l, v} H[l][v] * d phi_a / d lambda_l (v)
template <int D, bool ADD> __device__ __forceinline__ void project_gradients(const double (*H)[D + 1], double* y) {
    constexpr int NV = D + 1;
#pragma unroll
    for (int a = 0; a < NV; ++a) {
        double s = 3.0 * H[a][a];
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (v != a) s -= H[a][v];
        y[a] = ADD ? y[a] + s : s;
    }
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int b = a + 1; b < NV; ++b) {
            const double s = 4.0 * (H[a][b] + H[b][a]);
            y[P2<D>::edge(a, b)] = ADD ? y[P2<D>::edge(a, b)] + s : s;
        }
}

// W[v][v'] = sum_b coef[b] M3[b][pair(v, v')]
template <int D> __device__ __forceinline__ void pair_matrix(const double* coef, double (*W)[D + 1]) {
    constexpr int NV = D + 1, ND = P2<D>::ND;
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int b = a; b < NV; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < ND; ++k) s = fma(coef[k], P2Tab<D>::M3[k][P2<D>::pair(a, b)], s);
            W[a][b] = s;
            W[b][a] = s;
        }
}

// normal derivative at the D facet vertices from a frame vector F = [apex | fv | fe | ae]
template <int D> __device__ __forceinline__ void dn_vertices(const double* F, double gnA, const double* gnV, double* out) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
        double s = 3.0 * gnV[m] * F[1 + m];
        s = fma(gnA, fma(4.0, F[P2<D>::ae(m)], -F[0]), s);
#pragma unroll
        for (int mp = 0; mp < D; ++mp)
            if (mp != m) s = fma(gnV[mp], fma(4.0, F[P2<D>::fe(m, mp)], -F[1 + mp]), s);
        out[m] = s;
    }
}

// y (cell dofs) += sum_m T[m] * (normal derivative of the own basis functions at facet vertex m); frame of local facet I
template <int D, int I> __device__ __forceinline__ void back_project(const double* T, double gnA, const double* gnV, double* y) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y[P2<D>::fs(I, 1 + m)] = fma(3.0 * gnV[m], T[m], y[P2<D>::fs(I, 1 + m)]);
        y[P2<D>::fs(I, 0)] = fma(-gnA, T[m], y[P2<D>::fs(I, 0)]);
        y[P2<D>::fs(I, P2<D>::ae(m))] = fma(4.0 * gnA, T[m], y[P2<D>::fs(I, P2<D>::ae(m))]);
#pragma unroll
        for (int mp = 0; mp < D; ++mp)
            if (mp != m) {
                y[P2<D>::fs(I, 1 + mp)] = fma(-gnV[mp], T[m], y[P2<D>::fs(I, 1 + mp)]);
                y[P2<D>::fs(I, P2<D>::fe(m, mp))] = fma(4.0 * gnV[mp], T[m], y[P2<D>::fs(I, P2<D>::fe(m, mp))]);
            }
    }
}

template <int D> __device__ __forceinline__ uint64_t frame_packed(int j) {
    constexpr uint64_t p0 = P2<D>::packed(0), p1 = P2<D>::packed(1), p2 = P2<D>::packed(2), p3 = P2<D>::packed(D == 3 ? 3 : 0);
    static_assert(p0 == P2Tab<D>::FRAME_PACKED[0] && p1 == P2Tab<D>::FRAME_PACKED[1] && p2 == P2Tab<D>::FRAME_PACKED[2] &&
                  p3 == P2Tab<D>::FRAME_PACKED[D == 3 ? 3 : 0], "frame permutation disagrees with the generated table");
    uint64_t p = p0;
    if (j == 1) p = p1;
    if (j == 2) p = p2;
    if (D == 3 && j == 3) p = p3;
    return p;
}

// Workgroup staging: the block's own nodal vectors in LDS ([cell][ND]); vectors A and B (x and kappa | x and phi)
template <int D> struct StageP2 {
    const lds_double* a;
    const lds_double* b;
    int64_t c0;
    unsigned nvalid;        // 0: nothing staged (setup kernels)
};

// neighbour's vector in its facet frame, slots [S0, S1): LDS for in-block neighbours, exec-masked global gather otherwise
template <int D, int S0, int S1>
__device__ __forceinline__ void load_frame(const lds_double* l, const double* __restrict__ g, bool in_block, uint64_t packed, double* out) {
    double vg[S1 - S0];
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const unsigned dof = (unsigned)(packed >> (4 * s)) & 15u;
        out[s - S0] = l[dof];
        vg[s - S0] = 0.0;
    }
    if (!in_block) {
#pragma unroll
        for (int s = S0; s < S1; ++s) vg[s - S0] = g[(unsigned)(packed >> (4 * s)) & 15u];
    }
#pragma unroll
    for (int s = S0; s < S1; ++s) out[s - S0] = in_block ? out[s - S0] : vg[s - S0];
}

// per-facet geometry: own barycentric coordinates of the neighbour's apex, |g_i|, 2 / (h + h')
template <int D, int I, bool CLS>
__device__ __forceinline__ void facet_geometry(const MeshDev& m, const CellGeom<D>& K, const lds_double* rec, int64_t c, int64_t Kp, int j,
                                               double* L, double& sqG, double& hinv) {
    constexpr int NV = D + 1;
    if (CLS) {
#pragma unroll
        for (int a = 0; a < NV; ++a) L[a] = rec[11 + 6 * I + a];
        sqG = rec[11 + 6 * I + 4];
        hinv = rec[11 + 6 * I + 5];
    } else {
        double Xo[D];
        load_vertex<D>(m.coords, m.cells[Kp * NV + j], Xo);
        apex_bary<D>(K, Xo, L);
        sqG = fast_sqrt(K.G[I][I]);
        hinv = fast_rcp(0.5 * (m.h[c] + m.h[Kp]));
    }
}

// ---- monomial-product form of the triangle-facet integrals (p2_mono_tables.hpp; tools/gen_p2_tables.py: mono_tables) --------------
// a P2 trace given by its six facet-node values (v0 v1 v2 e01 e02 e12) as a polynomial in the facet's barycentric coordinates:
// coefficients of l0^2 l1^2 l2^2 l0l1 l0l2 l1l2  (Lagrange P2 with 1 = l0 + l1 + l2 folded in)
__device__ __forceinline__ void p2_to_mono(const double* u, double* a) {
    a[0] = u[0]; a[1] = u[1]; a[2] = u[2];
    a[3] = fma(4.0, u[3], -(u[0] + u[1]));
    a[4] = fma(4.0, u[4], -(u[0] + u[2]));
    a[5] = fma(4.0, u[5], -(u[1] + u[2]));
}
// q (degree 4, 15 coefficients) = a * b for two degree-2 polynomials; the index map is a compile-time table, so every q[.] is a register
__device__ __forceinline__ void mono_mul22(const double* a, const double* b, double* q) {
#pragma unroll
    for (int k = 0; k < P2Mono::N4; ++k) q[k] = 0.0;
#pragma unroll
    for (int i = 0; i < P2Mono::N2; ++i)
#pragma unroll
        for (int j = 0; j < P2Mono::N2; ++j) q[P2Mono::IDX22[i][j]] = fma(a[i], b[j], q[P2Mono::IDX22[i][j]]);
}
// c (degree 3, 10 coefficients) += a (degree 2) * sum_m d[m] l_m
__device__ __forceinline__ void mono_mul21_add(const double* a, const double* d, double* c) {
#pragma unroll
    for (int i = 0; i < P2Mono::N2; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) c[P2Mono::IDX21[i][m]] = fma(a[i], d[m], c[P2Mono::IDX21[i][m]]);
}

// ------------------------------------------------------------------------------------------------------------------
// EMI:  y = A(kappa) x     (forms: apply_p1.hip header; reference solver.py:325-328, 346)
// ------------------------------------------------------------------------------------------------------------------
// MODE 0: apply; MODE 1: cell-diagonal block (neighbour values 0, neighbour coefficient from global memory)
template <int D, int I, bool CLS, int MODE>
__device__ __forceinline__ void emi_facet_p2(const MeshDev& m, const CellGeom<D>& K, const lds_double* rec, int64_t c, const int* nb,
                                             uint32_t flags, const double* xv, const double* kv, const double* __restrict__ x,
                                             const double* __restrict__ kappa, const StageP2<D>& st, double C_phi, double tau,
                                             double* y) {
    constexpr int NV = D + 1, ND = P2<D>::ND, NF = P2<D>::NF;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    const uint64_t packed = frame_packed<D>(j);
    const unsigned loc0 = (unsigned)(Kp - st.c0);
    const bool in_block = MODE == 0 && loc0 < st.nvalid;
    const unsigned loc = in_block ? loc0 : 0u;
    double Fn[ND], Kn[NF];
    if (MODE == 0) {
        load_frame<D, 0, ND>(st.a + loc * ND, x + Kp * ND, in_block, packed, Fn);
        load_frame<D, 1, 1 + NF>(st.b + loc * ND, kappa + Kp * ND, in_block, packed, Kn);
    } else {
#pragma unroll
        for (int s = 0; s < ND; ++s) Fn[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NF; ++s) Kn[s] = kappa[Kp * ND + ((unsigned)(packed >> (4 * (1 + s))) & 15u)];
    }
    double ju[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) ju[n] = xv[P2<D>::fs(I, 1 + n)] - Fn[1 + n];
    double L[NV], sqG, hinv;
    facet_geometry<D, I, CLS>(m, K, rec, c, Kp, j, L, sqG, hinv);
    const double area = sqG * (double)D * K.vol;
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * area;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NF; ++k) s = fma(P2Tab<D>::FMASS[n][k], ju[k], s);
            y[P2<D>::fs(I, 1 + n)] = fma(w, s, y[P2<D>::fs(I, 1 + n)]);
        }
        return;
    }
    // n = -g_I / |g_I|:  grad(lambda_l) . n = -G_lI / |g_I|;  neighbour basis through the apex coordinates L (cell_geom.hpp)
    const double rs = fast_rcp(sqG);
    const double gnA = -sqG;
    double gnV[D], gnVn[D];
    const double gr = gnA * fast_rcp(L[I]);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        gnV[mm] = -K.G[mm + (mm >= I)][I] * rs;
        gnVn[mm] = fma(-L[mm + (mm >= I)], gr, gnV[mm]);
    }
    double Fo[ND];
#pragma unroll
    for (int s = 0; s < ND; ++s) Fo[s] = xv[P2<D>::fs(I, s)];
    double dno[D], dnn[D];
    dn_vertices<D>(Fo, gnA, gnV, dno);
    dn_vertices<D>(Fn, gr, gnVn, dnn);
    const double pen = 0.5 * tau * hinv;
    double r[NF], T[D];
#pragma unroll
    for (int n = 0; n < NF; ++n) r[n] = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) T[mm] = 0.0;
    if constexpr (D == 3) {
        // Round 4: every integrand of a_emi is a polynomial of degree <= 6 on the facet (kappa, [u] P2; d_n u P1; test function P2 / P1), so
        // the traces are multiplied AS POLYNOMIALS in the facet's barycentric coordinates and the products integrated exactly
        // against the test functions (tables of p2_mono_tables.hpp) instead of being sampled at the 12 points of the degree-6 rule:
        //   flux = pen (ko + kn) [u] - 1/2 (ko d_n u + kn d_n u')    degree 4 (the cubic part times 1 = l0 + l1 + l2)
        //   t    = -1/2 ko [u]                                          degree 4
        // 327 instead of 480 FP64 instructions per facet (profiles/r04_p2_mono.txt); same numbers to rounding (any exact rule gives them).
        double kon[NF], kom[NF], knm[NF], jm[NF], sm[NF];
#pragma unroll
        for (int n = 0; n < NF; ++n) kon[n] = kv[P2<D>::fs(I, 1 + n)];
        p2_to_mono(kon, kom);
        p2_to_mono(Kn, knm);
        p2_to_mono(ju, jm);
#pragma unroll
        for (int n = 0; n < NF; ++n) sm[n] = kom[n] + knm[n];
        double q4[P2Mono::N4], t4[P2Mono::N4], c3[P2Mono::N3];
        mono_mul22(sm, jm, q4);
        mono_mul22(kom, jm, t4);
        double hdo[D], hdn[D];
#pragma unroll
        for (int mm = 0; mm < D; ++mm) { hdo[mm] = -0.5 * dno[mm]; hdn[mm] = -0.5 * dnn[mm]; }
#pragma unroll
        for (int k = 0; k < P2Mono::N3; ++k) c3[k] = 0.0;
        mono_mul21_add(kom, hdo, c3);
        mono_mul21_add(knm, hdn, c3);
        // g = pen q4 + (l0 + l1 + l2) c3, held in q4's registers
#pragma unroll
        for (int k = 0; k < P2Mono::N4; ++k) q4[k] *= pen;
#pragma unroll
        for (int k = 0; k < P2Mono::N3; ++k)
#pragma unroll
            for (int mm = 0; mm < 3; ++mm) q4[P2Mono::IDX31[k][mm]] += c3[k];
#pragma unroll
        for (int k = 0; k < P2Mono::N4; ++k) {
#pragma unroll
            for (int n = 0; n < NF; ++n) r[n] = fma(q4[k], P2Mono::I4[k][n], r[n]);
#pragma unroll
            for (int mm = 0; mm < D; ++mm) T[mm] = fma(t4[k], P2Mono::I4L[k][mm], T[mm]);
        }
        const double ht = -0.5 * area;
#pragma unroll
        for (int n = 0; n < NF; ++n) r[n] *= area;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) T[mm] *= ht;
    } else {
#pragma unroll
    for (int q = 0; q < P2Tab<D>::NQE; ++q) {
        double ko = 0.0, kn = 0.0, jq = 0.0, do_ = 0.0, dn_ = 0.0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const double p = P2Tab<D>::PSIE[q][n];
            ko = fma(kv[P2<D>::fs(I, 1 + n)], p, ko);
            kn = fma(Kn[n], p, kn);
            jq = fma(ju[n], p, jq);
        }
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            do_ = fma(dno[mm], P2Tab<D>::LAME[q][mm], do_);
            dn_ = fma(dnn[mm], P2Tab<D>::LAME[q][mm], dn_);
        }
        const double w = P2Tab<D>::WE[q] * area;
        const double flux = w * fma(pen * (ko + kn), jq, -0.5 * fma(ko, do_, kn * dn_));
        const double t = -0.5 * w * ko * jq;
#pragma unroll
        for (int n = 0; n < NF; ++n) r[n] = fma(flux, P2Tab<D>::PSIE[q][n], r[n]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) T[mm] = fma(t, P2Tab<D>::LAME[q][mm], T[mm]);
    }
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) y[P2<D>::fs(I, 1 + n)] += r[n];
    back_project<D, I>(T, gnA, gnV, y);
}

template <int D, bool CLS, int MODE>
__device__ __forceinline__ void emi_cell_p2(const MeshDev& m, const CellGeom<D>& K, const lds_double* rec, int64_t c, const int* nb,
                                            uint32_t flags, const double* xv, const double* kv, const double* __restrict__ x,
                                            const double* __restrict__ kappa, const StageP2<D>& st, double C_phi, double tau,
                                            double* y) {
    constexpr int NV = D + 1;
    {
        double U[NV][NV], W[NV][NV], H[NV][NV];
        nodal_gradients<D>(xv, U);
        pair_matrix<D>(kv, W);
        // T = G U (in U's place would need a copy: H holds T, then H <- vol T W)
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) s = fma(K.G[k][l], U[l][v], s);
                H[k][v] = s * K.vol;
            }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double row[NV];
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                double s = 0.0;
#pragma unroll
                for (int v = 0; v < NV; ++v) s = fma(H[k][v], W[v][w], s);
                row[w] = s;
            }
#pragma unroll
            for (int w = 0; w < NV; ++w) U[k][w] = row[w];
        }
        project_gradients<D, false>(U, y);
    }
    emi_facet_p2<D, 0, CLS, MODE>(m, K, rec, c, nb, flags, xv, kv, x, kappa, st, C_phi, tau, y);
    emi_facet_p2<D, 1, CLS, MODE>(m, K, rec, c, nb, flags, xv, kv, x, kappa, st, C_phi, tau, y);
    emi_facet_p2<D, 2, CLS, MODE>(m, K, rec, c, nb, flags, xv, kv, x, kappa, st, C_phi, tau, y);
    if (D == 3) emi_facet_p2<D, (D == 3 ? 3 : 0), CLS, MODE>(m, K, rec, c, nb, flags, xv, kv, x, kappa, st, C_phi, tau, y);
}

template <int ND> __device__ __forceinline__ void load_cellvec(const double* __restrict__ p, int64_t c, double* v) {
    const double2* q = reinterpret_cast<const double2*>(p + (int64_t)ND * c);
#pragma unroll
    for (int k = 0; k < ND / 2; ++k) { const double2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}
template <int ND> __device__ __forceinline__ void store_cellvec(double* __restrict__ p, int64_t c, const double* v) {
    double2* q = reinterpret_cast<double2*>(p + (int64_t)ND * c);
#pragma unroll
    for (int k = 0; k < ND / 2; ++k) q[k] = make_double2(v[2 * k], v[2 * k + 1]);
}

template <int D, bool CLS> __device__ __forceinline__ void cell_geometry_p2(const MeshDev& m, int64_t c, const lds_double* rec, CellGeom<D>& K) {
    constexpr int NV = D + 1;
    if (CLS) {
        K.vol = rec[0];
        int q = 1;
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int b = a; b < NV; ++b) { K.G[a][b] = rec[q]; K.G[b][a] = rec[q]; ++q; }
    } else {
        int verts[NV];
        load_cell_ints<D>(m.cells, c, verts);
        load_cell_geometry<D>(m, verts, K);
    }
}

template <int D, int BLK, bool CLS>
__global__ __launch_bounds__(BLK) void k_emi_apply_p2(MeshDev m, const double* __restrict__ x, const double* __restrict__ kappa,
                                                      double* __restrict__ y, double C_phi, double tau) {
    constexpr int NV = D + 1, ND = P2<D>::ND;
    __shared__ __attribute__((aligned(16))) double s_x[BLK * ND];
    __shared__ __attribute__((aligned(16))) double s_k[BLK * ND];
    __shared__ __attribute__((aligned(16))) double s_tab[CLS ? CLS_MAX_P2 * KNP_CLS_STRIDE : 2];
    const int64_t c0 = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * BLK;
    if (c0 >= m.c_end) return;
    const int64_t c = c0 + threadIdx.x;
    const bool valid = c < m.c_end;
    if (CLS)
        for (int i = threadIdx.x; i < m.ncls * KNP_CLS_STRIDE; i += BLK) s_tab[i] = m.cls_table[i];
    int nb[NV];
    uint32_t flags = 0;
    unsigned cls = 0;
    double xv[ND], kv[ND], yv[ND];
    if (valid) {
        load_cell_ints<D>(m.nbr, c, nb);
        flags = m.fflag[c];
        if (CLS) cls = m.cls[c];
        load_cellvec<ND>(x, c, xv);
        load_cellvec<ND>(kappa, c, kv);
        const unsigned t = threadIdx.x;
#pragma unroll
        for (int a = 0; a < ND; ++a) { s_x[t * ND + a] = xv[a]; s_k[t * ND + a] = kv[a]; }
    }
    __syncthreads();
    if (!valid) return;
    const lds_double* rec = TO_LDS(s_tab) + cls * KNP_CLS_STRIDE;
    CellGeom<D> K;
    cell_geometry_p2<D, CLS>(m, c, rec, K);
    StageP2<D> st{TO_LDS(s_x), TO_LDS(s_k), c0, (unsigned)((m.c_end - c0 < BLK) ? (m.c_end - c0) : BLK)};
    emi_cell_p2<D, CLS, 0>(m, K, rec, c, nb, flags, xv, kv, x, kappa, st, C_phi, tau, yv);
    store_cellvec<ND>(y, c, yv);
}

// ------------------------------------------------------------------------------------------------------------------
// KNP:  y_k = A_k x_k, one species per blockIdx.y    (reference solver.py:586-594)
//   cells:  1/dt M + D K + z psi D int u grad(phi).grad(v);   facets dS(0): consistency, adjoint consistency, penalty on
//   jump(D u), upwind drift  -z psi jump(v) jump(un u),  un = max(D grad(phi).n_own, 0) sampled at the degree-5 facet points
// ------------------------------------------------------------------------------------------------------------------
struct KnpP2Args { double inv_dt, psi, tau; double z[KNP_MAX_SYS]; };

template <int D, int I, bool CLS, int MODE>
__device__ __forceinline__ void knp_facet_p2(const MeshDev& m, const CellGeom<D>& K, const lds_double* rec, int64_t c, const int* nb,
                                             uint32_t flags, const double* xv, const double* pv, double Dc, double zp,
                                             const double* __restrict__ x, const double* __restrict__ phi,
                                             const double* __restrict__ Dspec, const StageP2<D>& st, double tau, double* y) {
    constexpr int NV = D + 1, ND = P2<D>::ND, NF = P2<D>::NF;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    if (((fb >> 2) & 3u) != FK_SIPG) return;
    const int j = (int)(fb & 3u);
    const int64_t Kp = nb[I];
    const uint64_t packed = frame_packed<D>(j);
    const unsigned loc0 = (unsigned)(Kp - st.c0);
    const bool in_block = MODE == 0 && loc0 < st.nvalid;
    const unsigned loc = in_block ? loc0 : 0u;
    double L[NV], sqG, hinv;
    facet_geometry<D, I, CLS>(m, K, rec, c, Kp, j, L, sqG, hinv);
    const double area = sqG * (double)D * K.vol;
    const double rs = fast_rcp(sqG);
    const double gnA = -sqG;
    double gnV[D], gnVn[D];
    const double gr = gnA * fast_rcp(L[I]);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        gnV[mm] = -K.G[mm + (mm >= I)][I] * rs;
        gnVn[mm] = fma(-L[mm + (mm >= I)], gr, gnV[mm]);
    }
    const double D2 = Dspec[Kp];
    // upwind speeds at the facet vertices (P1 along the facet): own side with n, neighbour side with -n
    double sp[D], sm[D];
    {
        double Po[ND], Pn[ND];
#pragma unroll
        for (int s = 0; s < ND; ++s) Po[s] = pv[P2<D>::fs(I, s)];
        if (MODE == 0) load_frame<D, 0, ND>(st.b + loc * ND, phi + Kp * ND, in_block, packed, Pn);
        else {
#pragma unroll
            for (int s = 0; s < ND; ++s) Pn[s] = phi[Kp * ND + ((unsigned)(packed >> (4 * s)) & 15u)];
        }
        dn_vertices<D>(Po, gnA, gnV, sp);
        dn_vertices<D>(Pn, gr, gnVn, sm);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) { sp[mm] *= Dc; sm[mm] *= -D2; }
    }
    double Fo[ND], Fn[ND];
#pragma unroll
    for (int s = 0; s < ND; ++s) Fo[s] = xv[P2<D>::fs(I, s)];
    if (MODE == 0) load_frame<D, 0, ND>(st.a + loc * ND, x + Kp * ND, in_block, packed, Fn);
    else {
#pragma unroll
        for (int s = 0; s < ND; ++s) Fn[s] = 0.0;
    }
    double dno[D], dnn[D];
    dn_vertices<D>(Fo, gnA, gnV, dno);
    dn_vertices<D>(Fn, gr, gnVn, dnn);
    const double pen = tau * hinv;
    double r[NF], T[D];
#pragma unroll
    for (int n = 0; n < NF; ++n) r[n] = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) T[mm] = 0.0;
#pragma unroll
    for (int q = 0; q < P2Tab<D>::NQK; ++q) {
        double uo = 0.0, un = 0.0, do_ = 0.0, dn_ = 0.0, so = 0.0, sn = 0.0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const double p = P2Tab<D>::PSIK[q][n];
            uo = fma(Fo[1 + n], p, uo);
            un = fma(Fn[1 + n], p, un);
        }
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double l = P2Tab<D>::LAMK[q][mm];
            do_ = fma(dno[mm], l, do_);
            dn_ = fma(dnn[mm], l, dn_);
            so = fma(sp[mm], l, so);
            sn = fma(sm[mm], l, sn);
        }
        const double upo = fmax(so, 0.0), upn = fmax(sn, 0.0);
        const double w = P2Tab<D>::WK[q] * area;
        // -1/2 (D dn u + D' dn u') + pen (D u - D' u') - z psi (un u - un' u')
        const double flux = w * (fma(fma(pen, Dc, -zp * upo), uo, -fma(pen, D2, -zp * upn) * un) - 0.5 * fma(Dc, do_, D2 * dn_));
        const double t = -0.5 * w * Dc * (uo - un);
#pragma unroll
        for (int n = 0; n < NF; ++n) r[n] = fma(flux, P2Tab<D>::PSIK[q][n], r[n]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) T[mm] = fma(t, P2Tab<D>::LAMK[q][mm], T[mm]);
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) y[P2<D>::fs(I, 1 + n)] += r[n];
    back_project<D, I>(T, gnA, gnV, y);
}

template <int D, bool CLS, int MODE>
__device__ __forceinline__ void knp_cell_p2(const MeshDev& m, const CellGeom<D>& K, const lds_double* rec, int64_t c, const int* nb,
                                            uint32_t flags, const double* xv, const double* pv, double Dc, double zp, double inv_dt,
                                            const double* __restrict__ x, const double* __restrict__ phi,
                                            const double* __restrict__ Dspec, const StageP2<D>& st, double tau, double* y) {
    constexpr int NV = D + 1, ND = P2<D>::ND;
    {
        // H[k][w] = D vol ( sum_l G_kl Z_l(w) + z psi sum_v (G P)_k(v) Q[v][w] ),  Z = M1 U,  Q = pair matrix of x
        double U[NV][NV], Q[NV][NV], H[NV][NV];
        nodal_gradients<D>(xv, U);
        constexpr double m1 = 1.0 / (double)((D + 1) * (D + 2));
#pragma unroll
        for (int l = 0; l < NV; ++l) {
            double su = 0.0;
#pragma unroll
            for (int v = 0; v < NV; ++v) su += U[l][v];
#pragma unroll
            for (int v = 0; v < NV; ++v) U[l][v] = m1 * (U[l][v] + su);
        }
        const double dv = Dc * K.vol;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) s = fma(K.G[k][l], U[l][w], s);
                H[k][w] = dv * s;
            }
        pair_matrix<D>(xv, Q);
        nodal_gradients<D>(pv, U);                      // U <- nodal gradients of phi
        const double dz = dv * zp;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double gp[NV];                              // grad(phi) . grad(lambda_k) at the vertices
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NV; ++l) s = fma(K.G[k][l], U[l][v], s);
                gp[v] = s * dz;
            }
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                double s = H[k][w];
#pragma unroll
                for (int v = 0; v < NV; ++v) s = fma(gp[v], Q[v][w], s);
                H[k][w] = s;
            }
        }
        project_gradients<D, false>(H, y);
        const double mw = inv_dt * K.vol;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < ND; ++b) s = fma(P2Tab<D>::MASS[a][b], xv[b], s);
            y[a] = fma(mw, s, y[a]);
        }
    }
    knp_facet_p2<D, 0, CLS, MODE>(m, K, rec, c, nb, flags, xv, pv, Dc, zp, x, phi, Dspec, st, tau, y);
    knp_facet_p2<D, 1, CLS, MODE>(m, K, rec, c, nb, flags, xv, pv, Dc, zp, x, phi, Dspec, st, tau, y);
    knp_facet_p2<D, 2, CLS, MODE>(m, K, rec, c, nb, flags, xv, pv, Dc, zp, x, phi, Dspec, st, tau, y);
    if (D == 3) knp_facet_p2<D, (D == 3 ? 3 : 0), CLS, MODE>(m, K, rec, c, nb, flags, xv, pv, Dc, zp, x, phi, Dspec, st, tau, y);
}

template <int D, int BLK, bool CLS>
__global__ __launch_bounds__(BLK) void k_knp_apply_p2(MeshDev m, const double* __restrict__ x_all, const double* __restrict__ phi,
                                                      const double* __restrict__ Dall, double* __restrict__ y_all, KnpP2Args ka) {
    constexpr int NV = D + 1, ND = P2<D>::ND;
    __shared__ __attribute__((aligned(16))) double s_x[BLK * ND];
    __shared__ __attribute__((aligned(16))) double s_p[BLK * ND];
    __shared__ __attribute__((aligned(16))) double s_tab[CLS ? CLS_MAX_P2 * KNP_CLS_STRIDE : 2];
    const int64_t c0 = m.c_begin + xcd_block(blockIdx.x, gridDim.x) * BLK;
    if (c0 >= m.c_end) return;
    const int sp = blockIdx.y;
    const double* x = x_all + (int64_t)sp * m.nc * ND;
    const double* Dspec = Dall + (int64_t)sp * m.nc;
    const int64_t c = c0 + threadIdx.x;
    const bool valid = c < m.c_end;
    if (CLS)
        for (int i = threadIdx.x; i < m.ncls * KNP_CLS_STRIDE; i += BLK) s_tab[i] = m.cls_table[i];
    int nb[NV];
    uint32_t flags = 0;
    unsigned cls = 0;
    double xv[ND], pv[ND], yv[ND], Dc = 0.0;
    if (valid) {
        load_cell_ints<D>(m.nbr, c, nb);
        flags = m.fflag[c];
        if (CLS) cls = m.cls[c];
        load_cellvec<ND>(x, c, xv);
        load_cellvec<ND>(phi, c, pv);
        Dc = Dspec[c];
        const unsigned t = threadIdx.x;
#pragma unroll
        for (int a = 0; a < ND; ++a) { s_x[t * ND + a] = xv[a]; s_p[t * ND + a] = pv[a]; }
    }
    __syncthreads();
    if (!valid) return;
    const lds_double* rec = TO_LDS(s_tab) + cls * KNP_CLS_STRIDE;
    CellGeom<D> K;
    cell_geometry_p2<D, CLS>(m, c, rec, K);
    StageP2<D> st{TO_LDS(s_x), TO_LDS(s_p), c0, (unsigned)((m.c_end - c0 < BLK) ? (m.c_end - c0) : BLK)};
    knp_cell_p2<D, CLS, 0>(m, K, rec, c, nb, flags, xv, pv, Dc, ka.z[sp] * ka.psi, ka.inv_dt, x, phi, Dspec, st, ka.tau, yv);
    store_cellvec<ND>(y_all + (int64_t)sp * m.nc * ND, c, yv);
}

// ------------------------------------------------------------------------------------------------------------------
// Cell-diagonal blocks and their inverses (block-Jacobi).  ND threads per cell: thread (cell, b) applies the cell-local
// operator to the unit vector e_b (neighbour values 0), the block is transposed through LDS so that thread r holds row r,
// and the ND threads of a cell run an in-register Gauss-Jordan with the pivot row broadcast through LDS.
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool EMI, int CPB>
__global__ __launch_bounds__(CPB * P2<D>::ND) void k_p2_blockjacobi(MeshDev m, const double* __restrict__ coef, const double* __restrict__ Dall,
                                                                    bjreal* __restrict__ binv_all, double C_phi, double tau, KnpP2Args ka) {
    constexpr int NV = D + 1, ND = P2<D>::ND;
    __shared__ double M[CPB][ND][ND + 1];
    __shared__ double prow[CPB][ND];
    const int ci = threadIdx.x / ND, b = threadIdx.x % ND;
    const int64_t c = (int64_t)blockIdx.x * CPB + ci;
    const int sp = blockIdx.y;
    const bool valid = c < m.nc_owned;
    double col[ND];
    if (valid) {
        int nb[NV];
        load_cell_ints<D>(m.nbr, c, nb);
        const uint32_t flags = m.fflag[c];
        CellGeom<D> K;
        cell_geometry_p2<D, false>(m, c, nullptr, K);
        double e[ND], cv[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) e[a] = (a == b) ? 1.0 : 0.0;
        load_cellvec<ND>(coef, c, cv);
        StageP2<D> st{nullptr, nullptr, 0, 0u};
        if (EMI) {
            emi_cell_p2<D, false, 1>(m, K, nullptr, c, nb, flags, e, cv, nullptr, coef, st, C_phi, tau, col);
        } else {
            const double* Dspec = Dall + (int64_t)sp * m.nc;
            knp_cell_p2<D, false, 1>(m, K, nullptr, c, nb, flags, e, cv, Dspec[c], ka.z[sp] * ka.psi, ka.inv_dt, nullptr, coef, Dspec, st,
                                     ka.tau, col);
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) M[ci][a][b] = col[a];
    }
    __syncthreads();
    double row[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) row[k] = valid ? M[ci][b][k] : (k == b ? 1.0 : 0.0);
#pragma unroll
    for (int p = 0; p < ND; ++p) {
        if (b == p) {
            const double inv = 1.0 / row[p];
            row[p] = 1.0;
#pragma unroll
            for (int k = 0; k < ND; ++k) { row[k] *= inv; prow[ci][k] = row[k]; }
        }
        __syncthreads();
        if (b != p) {
            const double f = row[p];
            row[p] = 0.0;
#pragma unroll
            for (int k = 0; k < ND; ++k) row[k] = fma(-f, prow[ci][k], row[k]);
        }
        __syncthreads();
    }
    bjreal* dst = binv_all + (int64_t)sp * m.nc * ND * ND + c * ND * ND + b * ND;
    if (EMI) {                                         // exactly symmetric after rounding to fp32 (PCG needs an SPD preconditioner)
#pragma unroll
        for (int k = 0; k < ND; ++k) M[ci][b][k] = row[k];
        __syncthreads();
        if (valid) {
#pragma unroll
            for (int k = 0; k < ND; ++k) dst[k] = (bjreal)(0.5 * (row[k] + M[ci][k][b]));
        }
    } else if (valid) {
#pragma unroll
        for (int k = 0; k < ND; ++k) dst[k] = (bjreal)row[k];
    }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------------------------------------
static inline unsigned grid8_for(int64_t n, int blk) { return (unsigned)((((n + blk - 1) / blk + 7) / 8) * 8); }

static KnpP2Args knp_p2_args(knp_ctx* c) {
    KnpP2Args ka;
    ka.inv_dt = 1.0 / c->p.dt; ka.psi = c->p.psi; ka.tau = c->p.tau_knp;
    for (int k = 0; k < KNP_MAX_SYS; ++k) ka.z[k] = k < c->p.n_sys ? c->p.z[k] : 0.0;
    return ka;
}

#define P2_BLK 256

int p2_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y) {
    const bool cls = c->m.cls && c->m.dim == 3 && c->m.ncls <= CLS_MAX_P2;
    if (c->m.c_end <= c->m.c_begin) return 0;
    const dim3 g(grid8_for(c->m.c_end - c->m.c_begin, P2_BLK)), b(P2_BLK);
    if (c->m.dim == 3) {
        if (cls) hipLaunchKernelGGL((k_emi_apply_p2<3, P2_BLK, true>), g, b, 0, c->stream, c->m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
        else hipLaunchKernelGGL((k_emi_apply_p2<3, P2_BLK, false>), g, b, 0, c->stream, c->m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    } else {
        hipLaunchKernelGGL((k_emi_apply_p2<2, P2_BLK, false>), g, b, 0, c->stream, c->m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int p2_knp_apply(knp_ctx* c, const double* x, const double* phi, double* y) {
    const bool cls = c->m.cls && c->m.dim == 3 && c->m.ncls <= CLS_MAX_P2;
    if (c->m.c_end <= c->m.c_begin) return 0;
    const dim3 g(grid8_for(c->m.c_end - c->m.c_begin, P2_BLK), (unsigned)c->p.n_sys), b(P2_BLK);
    const KnpP2Args ka = knp_p2_args(c);
    if (c->m.dim == 3) {
        if (cls) hipLaunchKernelGGL((k_knp_apply_p2<3, P2_BLK, true>), g, b, 0, c->stream, c->m, x, phi, (const double*)c->D, y, ka);
        else hipLaunchKernelGGL((k_knp_apply_p2<3, P2_BLK, false>), g, b, 0, c->stream, c->m, x, phi, (const double*)c->D, y, ka);
    } else {
        hipLaunchKernelGGL((k_knp_apply_p2<2, P2_BLK, false>), g, b, 0, c->stream, c->m, x, phi, (const double*)c->D, y, ka);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// which 0: EMI blocks from kappa; 1: KNP blocks (all species) from phi
int p2_block_inverse(knp_ctx* c, int which, const double* coef, bjreal* binv) {
    const KnpP2Args ka = knp_p2_args(c);
    if (c->m.dim == 3) {
        constexpr int CPB = 25;
        const dim3 g((unsigned)((c->m.nc_owned + CPB - 1) / CPB), (unsigned)(which == 0 ? 1 : c->p.n_sys)), b(CPB * 10);
        if (which == 0) hipLaunchKernelGGL((k_p2_blockjacobi<3, true, CPB>), g, b, 0, c->stream, c->m, coef, (const double*)c->D, binv, c->p.C_phi, c->p.tau_emi, ka);
        else hipLaunchKernelGGL((k_p2_blockjacobi<3, false, CPB>), g, b, 0, c->stream, c->m, coef, (const double*)c->D, binv, c->p.C_phi, c->p.tau_emi, ka);
    } else {
        constexpr int CPB = 42;
        const dim3 g((unsigned)((c->m.nc_owned + CPB - 1) / CPB), (unsigned)(which == 0 ? 1 : c->p.n_sys)), b(CPB * 6);
        if (which == 0) hipLaunchKernelGGL((k_p2_blockjacobi<2, true, CPB>), g, b, 0, c->stream, c->m, coef, (const double*)c->D, binv, c->p.C_phi, c->p.tau_emi, ka);
        else hipLaunchKernelGGL((k_p2_blockjacobi<2, false, CPB>), g, b, 0, c->stream, c->m, coef, (const double*)c->D, binv, c->p.C_phi, c->p.tau_emi, ka);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// FP64 MFMA probe: the one genuinely dense, cell-independent contraction of the P2 path -- the facet quadrature
//   forward  [NQ x NF] . {jump(u), kappa, kappa'}  and  [NQ x D] . {dn u, dn u'},   pointwise flux,
//   backward [NF x NQ] . flux  and  [D x NQ] . t
// -- once as v_mfma_f64_16x16x4_f64 tiles (columns = 16 cell-facets per wave, the 4 lane groups hold the K index) and once
// as the per-thread FMA chain the product kernel uses, on identical synthetic inputs [ncol][26] -> outputs [ncol][9].
// knp_probe_facet_contraction times both; tests check that they agree.  Measured (profiles/r02_mfma_probe.md): the MFMA
// form pads 12 x 6 / 6 x 12 / 3 x 12 to 16 x 16 x 4 tiles and runs at the same FP64 rate as the vector pipe, so it loses.
// ------------------------------------------------------------------------------------------------------------------
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct ProbeTabs { double psi[16][8]; double lam[16][4]; double w[16]; };   // zero padded: [q][n], [q][m], [q]

__global__ __launch_bounds__(256) void k_probe_valu(int64_t ncol, const double* __restrict__ in, double* __restrict__ out) {
    constexpr int D = 3, NF = 6, NQ = P2Tab<3>::NQE;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const double* p = in + col * 26;
    double ju[NF], ko[NF], kn[NF], dno[D], dnn[D];
#pragma unroll
    for (int n = 0; n < NF; ++n) { ju[n] = p[n]; ko[n] = p[6 + n]; kn[n] = p[12 + n]; }
#pragma unroll
    for (int m = 0; m < D; ++m) { dno[m] = p[18 + m]; dnn[m] = p[21 + m]; }
    const double area = p[24], pen = p[25];
    double r[NF] = {0, 0, 0, 0, 0, 0}, T[D] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double a = 0.0, b = 0.0, j = 0.0, d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const double ps = P2Tab<3>::PSIE[q][n];
            a = fma(ko[n], ps, a); b = fma(kn[n], ps, b); j = fma(ju[n], ps, j);
        }
#pragma unroll
        for (int m = 0; m < D; ++m) { d0 = fma(dno[m], P2Tab<3>::LAME[q][m], d0); d1 = fma(dnn[m], P2Tab<3>::LAME[q][m], d1); }
        const double w = P2Tab<3>::WE[q] * area;
        const double flux = w * fma(pen * (a + b), j, -0.5 * fma(a, d0, b * d1));
        const double t = -0.5 * w * a * j;
#pragma unroll
        for (int n = 0; n < NF; ++n) r[n] = fma(flux, P2Tab<3>::PSIE[q][n], r[n]);
#pragma unroll
        for (int m = 0; m < D; ++m) T[m] = fma(t, P2Tab<3>::LAME[q][m], T[m]);
    }
    double* o = out + col * 9;
#pragma unroll
    for (int n = 0; n < NF; ++n) o[n] = r[n];
#pragma unroll
    for (int m = 0; m < D; ++m) o[6 + m] = T[m];
}

// one wave = 16 columns; lane (g = lane >> 4, n = lane & 15): B operand element [k = g][col n], A operand element [row n][k = g],
// C/D registers i = 0..3 hold rows g + 4 i of column n (cdna_hip_programming.md section 3, f64 layout)
__global__ __launch_bounds__(256) void k_probe_mfma(int64_t ncol, const double* __restrict__ in, const ProbeTabs* __restrict__ tabs,
                                                    double* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int64_t col0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    if (col0 >= ncol) return;                                   // whole wave
    const int64_t col = (col0 + n < ncol) ? col0 + n : ncol - 1;
    const double* p = in + col * 26;
    // A operands: forward  A[q = n][k]  with k = g (step 0), 4 + g (step 1);  backward  A[m = n][q = g + 4 i]
    const double aP0 = tabs->psi[n][g], aP1 = tabs->psi[n][4 + g], aL = tabs->lam[n][g];
    double bP[3], bL[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { bP[i] = n < 8 ? tabs->psi[g + 4 * i][n] : 0.0; bL[i] = n < 4 ? tabs->lam[g + 4 * i][n] : 0.0; }
    const double4_t z = {0.0, 0.0, 0.0, 0.0};
    double4_t Dj = z, Da = z, Db = z, D0 = z, D1 = z;
    // B operands: nodal values k = g and 4 + g of this column (k >= 6 / k >= 3 are padding)
    Dj = __builtin_amdgcn_mfma_f64_16x16x4f64(aP0, p[g], Dj, 0, 0, 0);
    Da = __builtin_amdgcn_mfma_f64_16x16x4f64(aP0, p[6 + g], Da, 0, 0, 0);
    Db = __builtin_amdgcn_mfma_f64_16x16x4f64(aP0, p[12 + g], Db, 0, 0, 0);
    const bool k1 = g < 2;
    Dj = __builtin_amdgcn_mfma_f64_16x16x4f64(aP1, k1 ? p[4 + g] : 0.0, Dj, 0, 0, 0);
    Da = __builtin_amdgcn_mfma_f64_16x16x4f64(aP1, k1 ? p[10 + g] : 0.0, Da, 0, 0, 0);
    Db = __builtin_amdgcn_mfma_f64_16x16x4f64(aP1, k1 ? p[16 + g] : 0.0, Db, 0, 0, 0);
    const bool k3 = g < 3;
    D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aL, k3 ? p[18 + g] : 0.0, D0, 0, 0, 0);
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aL, k3 ? p[21 + g] : 0.0, D1, 0, 0, 0);
    const double area = p[24], pen = p[25];
    double4_t R = z, T = z;
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                // quadrature points q = g + 4 i of this column
        const double w = tabs->w[g + 4 * i] * area;
        const double a = Da[i], b = Db[i], j = Dj[i];
        const double flux = w * fma(pen * (a + b), j, -0.5 * fma(a, D0[i], b * D1[i]));
        const double t = -0.5 * w * a * j;
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(bP[i], flux, R, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(bL[i], t, T, 0, 0, 0);
    }
    if (col0 + n < ncol) {
        double* o = out + (col0 + n) * 9;
        o[g] = R[0];                                            // rows m = g
        if (g < 2) o[4 + g] = R[1];                             // rows m = 4 + g
        if (g < 3) o[6 + g] = T[0];
    }
}

}  // namespace

extern "C" int knp_probe_facet_contraction(knp_ctx* c, int variant, int64_t ncol, int reps, const double* in_host, double* out_host,
                                           float* avg_ms) {
    if (!c || ncol < 1 || reps < 1 || !in_host || !out_host || !avg_ms || (variant != 0 && variant != 1)) return -1;
    double *in = nullptr, *out = nullptr;
    ProbeTabs h{};
    for (int q = 0; q < P2Tab<3>::NQE; ++q) {
        for (int n = 0; n < 6; ++n) h.psi[q][n] = P2Tab<3>::PSIE[q][n];
        for (int m = 0; m < 3; ++m) h.lam[q][m] = P2Tab<3>::LAME[q][m];
        h.w[q] = P2Tab<3>::WE[q];
    }
    ProbeTabs* tabs = nullptr;
    HIPCHK(c, hipMalloc((void**)&in, sizeof(double) * 26 * ncol));
    HIPCHK(c, hipMalloc((void**)&out, sizeof(double) * 9 * ncol));
    HIPCHK(c, hipMalloc((void**)&tabs, sizeof(ProbeTabs)));
    HIPCHK(c, hipMemcpy(in, in_host, sizeof(double) * 26 * ncol, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(tabs, &h, sizeof(ProbeTabs), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(out, 0, sizeof(double) * 9 * ncol));
    auto launch = [&]() {
        if (variant == 0)
            hipLaunchKernelGGL(k_probe_valu, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, c->stream, ncol, (const double*)in, out);
        else
            hipLaunchKernelGGL(k_probe_mfma, dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, c->stream, ncol, (const double*)in,
                               (const ProbeTabs*)tabs, out);
    };
    launch();
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) launch();
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipGetLastError());
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / (float)reps;
    HIPCHK(c, hipMemcpy(out_host, out, sizeof(double) * 9 * ncol, hipMemcpyDeviceToHost));
    hipFree(in); hipFree(out); hipFree(tabs);
    return 0;
}
