// Ring-staged P1 SIPG operator applies for ANY 3D mesh (no geometry classes: the reference's production runs are unstructured tissue
// reconstructions, examples/emix-simulations/run_EMIx_simulation.py:162-195, examples/rat-neuron/run_rat_neuron.py:156-207).
// Same operators as apply_p1.hip / apply_ring.hip (reference: src/knpemidg/solver.py:325-328, 346 for a_emi; :586-594 for A_knp) and
// the same structure as apply_ring.hip -- one persistent workgroup per CU, four LOADER waves that copy a 256-cell block's records
// into LDS with global_load_lds_dwordx4 while four CONSUMER waves (one cell per lane) work on the block before it, consumers read
// LDS only -- with the geometry taken from VERTEX COORDINATES instead of class records:
//
//   * per block the host lists the vertices its cells touch plus the apex vertices of the facet neighbours outside the block
//     (<= 256 on the Morton-ordered tissue meshes: 176-185 on average), the loaders gather their coordinates (32 B each, out of the
//     L2: the whole coordinate array of the 973 k-tet mesh is 5.4 MB) next to the nodal rows, and a cell carries eight 1-byte
//     positions in that list (own vertices, neighbour apexes);
//   * the consumers form the Gram-form geometry of cell_geom.hpp in registers (gradients, Gram matrix, volume: ~95 FP64
//     instructions per cell; per facet the apex coordinates L, G_ii / L_i, the neighbour-gradient weights and one square root: ~40);
//     2 / (h + h') per facet is streamed as a fourth "nodal" row (its two diameters would cost two more gathers or ~100 instructions);
//   * out-of-block neighbour lists reach 318 entries on these meshes (structured: <= 224), a slot is 58 KB (EMI) / 72 KB (KNP, two
//     species): TWO slots, the loaders run one block ahead and every interval ends with a full vmcnt(0).
//
// Streaming a precomputed per-cell geometry record instead (Gram matrix + per-facet coefficients: 248-344 B per cell) would triple the
// bytes of a 137-byte operator; recomputing costs FP64 issue slots, which the loaders' stream leaves free.
#include "cell_geom.hpp"
#include "ring_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

using namespace ring;
constexpr int UH = 320;           // halo entries staged per block (a multiple of 32)
constexpr int UHX = UH / 32;      // DMA instructions per halo row set (two lanes per row)
constexpr int UV = 256;           // vertices staged per block (positions fit one byte)
constexpr int UENT = RB + UH;
constexpr int USLOTS = 2;
constexpr int ULIST_H = 2048, ULIST_V = 1024, ULISTB = ULIST_H + ULIST_V;   // bytes of one list buffer: halo list (2 DMA pieces), vertex list (1)
constexpr int ULOADERS = 4;

struct RingUTables {
    int32_t* hb_src = nullptr;    // [nblk][UH]   4 * neighbour cell + its local facet, -1 padded
    uint16_t* hb_loc = nullptr;   // [nc_owned][4] LDS entry of the neighbour behind facet i
    int32_t* vb_src = nullptr;    // [nblk][UV]   vertex ids staged for the block, 0 padded
    uint8_t* vloc = nullptr;      // [nc_owned][8] positions of the own vertices (0..3) and of the facet neighbours' apexes (4..7)
    double* hinv4 = nullptr;      // [nc_owned][4] 2 / (h + h') per facet (0 on the boundary)
    int64_t long0 = 0;            // blocks [0, long0) fit the staging limits
    int hs = 0;                   // longest halo list
};

// slot images (bytes)
struct EmiU {
    static constexpr int XB = UENT * 32;                 // x rows, then kappa rows
    static constexpr int HI = 2 * XB;                    // 2 / (h + h') rows [256][4]
    static constexpr int CO = HI + RB * 32;              // vertex coordinates [UV][4]
    static constexpr int VL = CO + UV * 32;              // vloc [256][8]
    static constexpr int FL = VL + RB * 8;               // flag bytes [256] u32
    static constexpr int HL = FL + RB * 4;               // hb_loc [256][4] u16
    static constexpr int SLOT = HL + RB * 8;
};
template <int NS> struct KnpU {
    static constexpr int XB = UENT * 32;                 // one species' rows
    static constexpr int G0 = NS * XB;                   // own gphi rows [256][4]
    static constexpr int GH0 = G0 + RB * 32;             // halo gphi halves [UH][2]
    static constexpr int HI = GH0 + UH * 16;
    static constexpr int CO = HI + RB * 32;
    static constexpr int VL = CO + UV * 32;
    static constexpr int FL = VL + RB * 8;
    static constexpr int HL = FL + RB * 4;
    static constexpr int NM = HL + RB * 8;               // neighbour materials [256] u32
    static constexpr int MA = NM + RB * 4;               // own material [256] u8 (the DMA piece that brings it is 1 KiB)
    static constexpr int SLOT = MA + 1024;
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

// ---- consumer side: geometry from the staged coordinates -------------------------------------------------------------------------
__device__ __forceinline__ void lds_vertex(const lds_double* co, unsigned v, double* X) {
    typedef double __attribute__((ext_vector_type(2))) vdouble2;
    typedef __attribute__((address_space(3))) vdouble2 lds_vdouble2;
    const vdouble2 a = *(const lds_vdouble2*)(co + 4 * v);
    X[0] = a.x; X[1] = a.y; X[2] = co[4 * v + 2];
}

// what the facet terms need from the geometry of facet I (MeshDev::cls_ext holds the same numbers per class on structured meshes):
// gr = G_II / L_I, cf = the neighbour-gradient weights G_{a_m I} - L_{a_m} gr, the penalty and neighbour-volume factors
struct FacetCoef { double gr, cf[3], pen_geo, nLI_DV, sqG_DV; };
template <int I> __device__ __forceinline__ void facet_coef(const CellGeom<3>& K, const lds_double* co, unsigned vapex, double hinv, FacetCoef& f) {
    double Xo[3], L[4];
    lds_vertex(co, vapex, Xo);
    apex_bary<3>(K, Xo, L);
    const double gr = K.G[I][I] * fast_rcp(L[I]);
    f.gr = gr;
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) f.cf[mm] = fma(-L[mm + (mm >= I)], gr, K.G[mm + (mm >= I)][I]);
    const double DV = 3.0 * K.vol;
    f.sqG_DV = fast_sqrt(K.G[I][I]) * DV;
    f.pen_geo = hinv * f.sqG_DV;
    f.nLI_DV = -L[I] * DV;
}

// ================================================================================================================================
// EMI:  y = A(kappa) x      (forms and notation: apply_p1.hip, emi_facet_cls; arithmetic of apply_ring.hip: emi_facet_ring)
// ================================================================================================================================
template <int I>
__device__ __forceinline__ void emi_facet_u(const CellGeom<3>& K, uint32_t flags, unsigned loc, unsigned vapex, double hinv, const double* xv,
                                            const double* gx, const double* kv, double C_phi, double tau, const lds_double* X,
                                            const lds_double* KA, const lds_double* co, double* y) {
    constexpr int D = 3, NV = 4;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    const uint32_t kind = (fb >> 2) & 3u;
    if (kind >= FK_EXTERIOR) return;
    const unsigned j = fb & 3u;
    double xr[NV], kr[NV], xf[D], knf[D];
    lds_row(X, loc, xr);
    lds_row(KA, loc, kr);
    const double xap = pick_apex<D>(xr, (int)j);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        xf[mm] = pick_facet<D>(xr, mm, (int)j);
        knf[mm] = pick_facet<D>(kr, mm, (int)j);
    }
    double du[D], sdu = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        du[mm] = xv[mm + (mm >= I)] - xf[mm];
        sdu += du[mm];
    }
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * (fast_sqrt(K.G[I][I]) * (3.0 * K.vol)) * FacetConst<D>::mass;       // facet area = sqrt(G_II) D vol
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[mm + (mm >= I)] = fma(w, sdu + du[mm], y[mm + (mm >= I)]);
        return;
    }
    FacetCoef fc;
    facet_coef<I>(K, co, vapex, hinv, fc);
    const double s_own = gx[I];                                            // (G x)_I from the cell term
    double s_nb = xap * fc.gr;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], fc.cf[mm], s_nb);
    double kf[D], sk = 0.0, skn = 0.0, q = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kf[mm] = kv[mm + (mm >= I)];
        sk += kf[mm];
        skn += knf[mm];
        q = fma(kf[mm], sdu + du[mm], q);
    }
    const double hm = 0.5 * (double)D * K.vol * FacetConst<D>::mass;
    q *= hm;
#pragma unroll
    for (int a = 0; a < NV; ++a) y[a] = fma(K.G[a][I], q, y[a]);
    const double pw = tau * fc.pen_geo * FacetConst<D>::trip;
    double kb[D], skb = 0.0, skd = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        kb[mm] = 0.5 * (kf[mm] + knf[mm]);
        skb += kb[mm];
        skd = fma(kb[mm], du[mm], skd);
    }
    const double bs = fma(skb, sdu, skd);
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const double t1 = hm * fma(s_own, sk + kf[mm], s_nb * (skn + knf[mm]));
        const double t3 = pw * (bs + fma(kb[mm], sdu, du[mm] * fma(2.0, kb[mm], skb)));
        y[mm + (mm >= I)] += t1 + t3;
    }
}

// loaders of both kernels: the lists of block n (halo list: two pieces, vertex list: one) into list buffer n & 1
__device__ __forceinline__ void dma_lists_u(const RingUTables& T, int64_t b, unsigned dst, int lane) {
    glds16(T.hb_src + b * UH + lane * 4, dst);
    glds16(T.hb_src + b * UH + 256 + (lane * 4 < UH - 256 ? lane * 4 : 0), dst + 1024);
    glds16(T.vb_src + b * UV + lane * 4, dst + ULIST_H);
}
// coordinates of the block's vertices: four pieces, lanes 2 i / 2 i + 1 carry the two halves of a padded 32-byte vertex
__device__ __forceinline__ void dma_coords_u(const double* __restrict__ coords, const lds_int* V, unsigned dst, int lane, int p0, int p1) {
    for (int p = p0; p < p1; ++p) {
        const int v = V[p * 32 + (lane >> 1)];
        glds16(coords + (int64_t)v * 4 + (lane & 1) * 2, dst + p * 1024);
    }
}

__global__ __launch_bounds__(RB + 64 * ULOADERS) void k_emi_apply_ring_u(MeshDev m, RingUTables T, const double* __restrict__ x,
                                                                       const double* __restrict__ kappa, double* __restrict__ yout, double C_phi,
                                                                       double tau) {
    typedef EmiU R;
    constexpr int NV = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_list = smem + USLOTS * R::SLOT;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    const unsigned list0 = base + USLOTS * R::SLOT;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const RingWalk w(m);
    if (w.blk(0) < 0) return;
    if (wave >= 4) {
        // loaders:  0: x rows + the lists     1: kappa rows + flag / vloc bytes     2: x halo rows + hinv rows     3: kappa halo rows + coordinates + hb_loc
        const int lw = wave - 4;
        auto list_dma = [&](int64_t n) {
            const int64_t b = w.blk(n);
            if (lw == 0 && b >= 0) dma_lists_u(T, b, list0 + (unsigned)(n & 1) * ULISTB, lane);
        };
        auto data_dma = [&](int64_t n) {
            const int64_t c0 = w.blk(n) * RB;
            const unsigned slot = base + (unsigned)(n & 1) * R::SLOT;
            const lds_int* L = (const lds_int*)(s_list + (n & 1) * ULISTB);
            const lds_int* V = (const lds_int*)(s_list + (n & 1) * ULISTB + ULIST_H);
            if (lw == 0) {
                dma_own_rows(x, c0, m.nc, slot, lane);
                dma_coords_u(m.coords, V, slot + R::CO, lane, 0, 4);
            } else if (lw == 1) {
                dma_own_rows(kappa, c0, m.nc, slot + R::XB, lane);
                glds16(m.fflag + c0 + 4 * lane, slot + R::FL);
                glds16(T.vloc + (c0 + 2 * lane) * 8, slot + R::VL);
                glds16(T.vloc + (c0 + 128 + 2 * lane) * 8, slot + R::VL + 1024);
                dma_coords_u(m.coords, V, slot + R::CO, lane, 4, 8);
            } else {
                const double* v = lw == 2 ? x : kappa;
                const unsigned dst = slot + (lw == 2 ? 0 : R::XB) + RB * 32;
#pragma unroll
                for (int p = 0; p < UHX; ++p) glds16(v + list_cell(L, p * 32 + (lane >> 1), T.hs) * NV + swz_half(lane), dst + p * 1024);
                if (lw == 2) dma_own_rows(T.hinv4, c0, m.nc_owned, slot + R::HI, lane);
                else {
                    glds16(T.hb_loc + (c0 + 2 * lane) * 4, slot + R::HL);
                    glds16(T.hb_loc + (c0 + 128 + 2 * lane) * 4, slot + R::HL + 1024);
                }
            }
        };
        list_dma(0); list_dma(1);
        wait_vm<0>();
        ring_barrier();                                                // A: every loader sees lists 0 and 1
        data_dma(0);
        wait_vm<0>();
        ring_barrier();                                                // B: block 0 has landed
        for (int64_t n = 0; w.blk(n) >= 0; ++n) {
            list_dma(n + 2);                                           // into the buffer block n's gathers read in the last interval
            if (w.blk(n + 1) >= 0) data_dma(n + 1);
            wait_vm<0>();
            ring_barrier();
        }
        return;
    }
    const unsigned t = threadIdx.x & (RB - 1);
    ring_barrier();
    ring_barrier();
    for (int64_t n = 0;; ++n) {
        const int64_t b = w.blk(n);
        if (b < 0) break;
        const int64_t c = b * RB + t;
        if (c >= m.c_begin && c < m.c_end) {
            const char* slot = smem + (n & 1) * R::SLOT;
            const lds_double* X = (const lds_double*)slot;
            const lds_double* KA = (const lds_double*)(slot + R::XB);
            const lds_double* co = (const lds_double*)(slot + R::CO);
            const uint32_t flags = ((const lds_u32*)(slot + R::FL))[t];
            const uint32_t vl0 = ((const lds_u32*)(slot + R::VL))[2 * t], vl1 = ((const lds_u32*)(slot + R::VL))[2 * t + 1];
            const uint32_t lw0 = ((const lds_u32*)(slot + R::HL))[2 * t], lw1 = ((const lds_u32*)(slot + R::HL))[2 * t + 1];
            CellGeom<3> K;
            {
                double Xc[4][3];
                lds_vertex(co, vl0 & 0xffu, Xc[0]);
                lds_vertex(co, (vl0 >> 8) & 0xffu, Xc[1]);
                lds_vertex(co, (vl0 >> 16) & 0xffu, Xc[2]);
                lds_vertex(co, vl0 >> 24, Xc[3]);
                cell_geometry_from<3>(Xc, K);
            }
            double xv[NV], kv[NV], yv[NV], gx[NV], hi[NV];
            lds_row(X, t, xv);
            lds_row(KA, t, kv);
            lds_row((const lds_double*)(slot + R::HI), t, hi);
            double kbar = 0.0;
#pragma unroll
            for (int a = 0; a < NV; ++a) kbar += kv[a];
            kbar *= K.vol / (double)NV;
#pragma unroll
            for (int a = 0; a < NV; ++a) {
                double sa = 0.0;
#pragma unroll
                for (int bb = 0; bb < NV; ++bb) sa = fma(xv[bb], K.G[bb][a], sa);
                gx[a] = sa;
                yv[a] = kbar * sa;
            }
            emi_facet_u<0>(K, flags, lw0 & 0xffffu, vl1 & 0xffu, hi[0], xv, gx, kv, C_phi, tau, X, KA, co, yv);
            emi_facet_u<1>(K, flags, lw0 >> 16, (vl1 >> 8) & 0xffu, hi[1], xv, gx, kv, C_phi, tau, X, KA, co, yv);
            emi_facet_u<2>(K, flags, lw1 & 0xffffu, (vl1 >> 16) & 0xffu, hi[2], xv, gx, kv, C_phi, tau, X, KA, co, yv);
            emi_facet_u<3>(K, flags, lw1 >> 16, vl1 >> 24, hi[3], xv, gx, kv, C_phi, tau, X, KA, co, yv);
            store_nodal<3>(yout, c, yv);
        }
        ring_barrier();                                                // the loaders refill this slot in their next interval
    }
}

// ================================================================================================================================
// KNP:  y_k = A_k x_k for all solved species (forms and notation: apply_p1.hip, k_knp_apply_halo; arithmetic of knp_facet_ring)
// ================================================================================================================================
template <int NS, int I>
__device__ __forceinline__ void knp_facet_u(const CellGeom<3>& K, uint32_t flags, unsigned loc, unsigned vapex, double hinv, unsigned dsel,
                                            const double (*xv)[4], const double (*gx)[4], const double* gp, const double* Dk, const double* hvD,
                                            const double* zpsi, double tau, const lds_double* X, const lds_double* G, const lds_double* sD,
                                            const lds_double* co, double (*y)[4]) {
    constexpr int D = 3, NV = 4;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    if (((fb >> 2) & 3u) != FK_SIPG) return;
    const unsigned j = fb & 3u;
    FacetCoef fc;
    facet_coef<I>(K, co, vapex, hinv, fc);
    double gp_nb;
    {   // the 16-byte half that holds component j of the neighbour's gphi row: own rows (swizzled image) or the halo's [entry][2]
        typedef double __attribute__((ext_vector_type(2))) vdouble2;
        typedef __attribute__((address_space(3))) vdouble2 lds_vdouble2;
        const unsigned idx = loc < (unsigned)RB ? loc * NV + 2u * (((j >> 1) ^ (loc >> 3)) & 1u) : (unsigned)(RB * NV) + (loc - RB) * 2u;
        const vdouble2 g2 = *(const lds_vdouble2*)(G + idx);
        gp_nb = (j & 1u) ? g2.y : g2.x;
    }
    const double DV = (double)D * K.vol;
    const double up_own = fmax(-gp[I], 0.0) * DV;
    const double up_nb = fmax(-gp_nb, 0.0) * fc.nLI_DV;
    const double penA = tau * fc.pen_geo;
    const double hv = 0.5 * K.vol;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double xr[NV], xf[D];
        lds_row(X + (unsigned)k * (UENT * NV), loc, xr);
        const double xap = pick_apex<D>(xr, (int)j);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) xf[mm] = pick_facet<D>(xr, mm, (int)j);
        const double Dn = sD[(unsigned)k * KNP_MAX_MAT + dsel];
        const double s_own = gx[k][I];
        double s_nb = xap * fc.gr;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], fc.cf[mm], s_nb);
        const double zp = zpsi[k];
        const double c_own = Dk[k] * fma(-zp, up_own, penA);
        const double c_nb = Dn * fma(-zp, up_nb, penA);
        double sdu = 0.0, w[D], sw = 0.0;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) {
            const double xo = xv[k][mm + (mm >= I)];
            sdu += xo - xf[mm];
            w[mm] = fma(c_own, xo, -c_nb * xf[mm]);
            sw += w[mm];
        }
        const double t1m = fma(FacetConst<D>::mass, sw, hv * fma(Dk[k], s_own, Dn * s_nb));
        const double t2 = hvD[k] * sdu;
#pragma unroll
        for (int a = 0; a < NV; ++a) y[k][a] = fma(K.G[a][I], t2, y[k][a]);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[k][mm + (mm >= I)] += fma(FacetConst<D>::mass, w[mm], t1m);
    }
}

template <int NS>
__global__ __launch_bounds__(RB + 64 * ULOADERS) void k_knp_apply_ring_u(MeshDev m, RingUTables T, const double* __restrict__ x,
                                                                       const double* __restrict__ gphi, double* __restrict__ yout, KnpArgs ka,
                                                                       const uint8_t* __restrict__ mat, const uint8_t* __restrict__ nmat4,
                                                                       const double* __restrict__ dtab) {
    typedef KnpU<NS> R;
    constexpr int NV = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_list = smem + USLOTS * R::SLOT;
    double* s_D = reinterpret_cast<double*>(s_list + 2 * ULISTB);                // [NS][KNP_MAX_MAT]
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    const unsigned list0 = base + USLOTS * R::SLOT;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const RingWalk w(m);
    if (w.blk(0) < 0) return;
    if (threadIdx.x < NS * KNP_MAX_MAT) s_D[threadIdx.x] = dtab[threadIdx.x];
    if (wave >= 4) {
        //   loader 0: species 0 rows, flag + vloc bytes, the lists, half of the coordinates     loader 1: species 1 rows, materials, hb_loc, the other half
        //   loader 2: gphi rows + the halo's gphi halves + hinv rows                            loader 3: the halo rows of all species
        const int lw = wave - 4;
        auto list_dma = [&](int64_t n) {
            const int64_t b = w.blk(n);
            if (lw == 0 && b >= 0) dma_lists_u(T, b, list0 + (unsigned)(n & 1) * ULISTB, lane);
        };
        auto data_dma = [&](int64_t n) {
            const int64_t c0 = w.blk(n) * RB;
            const unsigned slot = base + (unsigned)(n & 1) * R::SLOT;
            const lds_int* L = (const lds_int*)(s_list + (n & 1) * ULISTB);
            const lds_int* V = (const lds_int*)(s_list + (n & 1) * ULISTB + ULIST_H);
            if (lw == 0) {
                dma_own_rows(x, c0, m.nc, slot, lane);
                glds16(m.fflag + c0 + 4 * lane, slot + R::FL);
                glds16(T.vloc + (c0 + 2 * lane) * 8, slot + R::VL);
                glds16(T.vloc + (c0 + 128 + 2 * lane) * 8, slot + R::VL + 1024);
                dma_coords_u(m.coords, V, slot + R::CO, lane, 0, 4);
            } else if (lw == 1) {
                if (NS > 1) dma_own_rows(x + m.nc * NV, c0, m.nc, slot + R::XB, lane);
                glds16(nmat4 + (c0 + 4 * lane) * 4, slot + R::NM);
                glds16(mat + c0 + 16 * lane, slot + R::MA);
                glds16(T.hb_loc + (c0 + 2 * lane) * 4, slot + R::HL);
                glds16(T.hb_loc + (c0 + 128 + 2 * lane) * 4, slot + R::HL + 1024);
                dma_coords_u(m.coords, V, slot + R::CO, lane, 4, 8);
            } else if (lw == 2) {
                dma_own_rows(gphi, c0, m.nc, slot + R::G0, lane);
                dma_own_rows(T.hinv4, c0, m.nc_owned, slot + R::HI, lane);
#pragma unroll
                for (int p = 0; p < UH / 64; ++p) {
                    const int e = p * 64 + lane;
                    const int src = e < T.hs ? L[e] : -1;
                    const int64_t Kp = src >= 0 ? (int64_t)(src >> 2) : 0;
                    const int j = src >= 0 ? (src & 3) : 0;
                    glds16(gphi + Kp * NV + (j >> 1) * 2, slot + R::GH0 + p * 1024);
                }
            } else {
#pragma unroll
                for (int p = 0; p < UHX; ++p) {
                    const int64_t Kp = list_cell(L, p * 32 + (lane >> 1), T.hs);
#pragma unroll
                    for (int k = 0; k < NS; ++k) glds16(x + (int64_t)k * m.nc * NV + Kp * NV + swz_half(lane), slot + k * R::XB + RB * 32 + p * 1024);
                }
            }
        };
        list_dma(0); list_dma(1);
        wait_vm<0>();
        ring_barrier();                                                // A
        data_dma(0);
        wait_vm<0>();
        ring_barrier();                                                // B
        for (int64_t n = 0; w.blk(n) >= 0; ++n) {
            list_dma(n + 2);
            if (w.blk(n + 1) >= 0) data_dma(n + 1);
            wait_vm<0>();
            ring_barrier();
        }
        return;
    }
    const unsigned t = threadIdx.x & (RB - 1);
    double zpsi[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) zpsi[k] = ka.z[k] * ka.psi;
    ring_barrier();                                                    // A: material table staged
    ring_barrier();                                                    // B
    for (int64_t n = 0;; ++n) {
        const int64_t b = w.blk(n);
        if (b < 0) break;
        const int64_t c = b * RB + t;
        if (c >= m.c_begin && c < m.c_end) {
            const char* slot = smem + (n & 1) * R::SLOT;
            const lds_double* X = (const lds_double*)slot;
            const lds_double* G = (const lds_double*)(slot + R::G0);
            const lds_double* co = (const lds_double*)(slot + R::CO);
            const lds_double* sD = TO_LDS(s_D);
            const uint32_t flags = ((const lds_u32*)(slot + R::FL))[t];
            const uint32_t vl0 = ((const lds_u32*)(slot + R::VL))[2 * t], vl1 = ((const lds_u32*)(slot + R::VL))[2 * t + 1];
            const uint32_t lw0 = ((const lds_u32*)(slot + R::HL))[2 * t], lw1 = ((const lds_u32*)(slot + R::HL))[2 * t + 1];
            const uint32_t nm = ((const lds_u32*)(slot + R::NM))[t];
            const unsigned mymat = ((const lds_u8*)(slot + R::MA))[t];
            CellGeom<3> K;
            {
                double Xc[4][3];
                lds_vertex(co, vl0 & 0xffu, Xc[0]);
                lds_vertex(co, (vl0 >> 8) & 0xffu, Xc[1]);
                lds_vertex(co, (vl0 >> 16) & 0xffu, Xc[2]);
                lds_vertex(co, vl0 >> 24, Xc[3]);
                cell_geometry_from<3>(Xc, K);
            }
            double xv[NS][NV], y[NS][NV], gp[NV], hi[NV], Dk[NS], gx[NS][NV], hvD[NS];
            lds_row(G, t, gp);
            lds_row((const lds_double*)(slot + R::HI), t, hi);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                lds_row(X + k * (UENT * NV), t, xv[k]);
                Dk[k] = sD[k * KNP_MAX_MAT + mymat];
                hvD[k] = 0.5 * K.vol * Dk[k];
            }
            const double mw = ka.inv_dt * K.vol / 20.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double sx = 0.0;
#pragma unroll
                for (int a = 0; a < NV; ++a) sx += xv[k][a];
                const double drift = zpsi[k] * Dk[k] * K.vol * sx / (double)NV;
                const double dv = Dk[k] * K.vol;
#pragma unroll
                for (int a = 0; a < NV; ++a) {
                    double sacc = 0.0;
#pragma unroll
                    for (int bb = 0; bb < NV; ++bb) sacc = fma(xv[k][bb], K.G[bb][a], sacc);
                    gx[k][a] = sacc;
                    y[k][a] = fma(mw, sx + xv[k][a], fma(dv, sacc, drift * gp[a]));
                }
            }
            knp_facet_u<NS, 0>(K, flags, lw0 & 0xffffu, vl1 & 0xffu, hi[0], nm & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, co, y);
            knp_facet_u<NS, 1>(K, flags, lw0 >> 16, (vl1 >> 8) & 0xffu, hi[1], (nm >> 8) & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, co, y);
            knp_facet_u<NS, 2>(K, flags, lw1 & 0xffffu, (vl1 >> 16) & 0xffu, hi[2], (nm >> 16) & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, co, y);
            knp_facet_u<NS, 3>(K, flags, lw1 >> 16, vl1 >> 24, hi[3], nm >> 24, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, co, y);
#pragma unroll
            for (int k = 0; k < NS; ++k) store_nodal<3>(yout + (int64_t)k * m.nc * NV, c, y[k]);
        }
        ring_barrier();
    }
}

int env_int_u(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int device_cus_u(int device) {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

template <typename KernelT> bool grant_lds_u(KernelT kernel, size_t lds) {
    static std::map<const void*, size_t> granted;
    auto it = granted.find((const void*)kernel);
    if (it != granted.end() && it->second >= lds) return true;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    granted[(const void*)kernel] = lds;
    return true;
}

dim3 grid_u(const MeshDev& m, int device, int reserve_cus) {
    const int64_t nblk = (m.c_end - 1) / RB - m.c_begin / RB + 1;
    const int cus = std::max(device_cus_u(device) - std::max(reserve_cus, 0), 8);
    int64_t per_xcd = std::max<int64_t>(1, std::min<int64_t>(cus / 8, (nblk + 7) / 8));
    const int wg = env_int_u("KNP_RING_WG", 0);                        // tests only (apply_ring.hip: ring_grid)
    if (wg >= 8) per_xcd = std::min<int64_t>(per_xcd, wg / 8);
    return dim3((unsigned)(8 * per_xcd));
}

struct RingUState {
    RingUTables T;
    bool tried = false, usable = false;
};

template <typename Tp> int up_u(knp_ctx* c, Tp** dst, const std::vector<Tp>& src) {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(Tp);
    HIPCHK(c, hipMalloc((void**)dst, bytes + 4096));                   // the DMA pieces read whole blocks past the last cell
    HIPCHK(c, hipMemset((char*)*dst + bytes, 0, 4096));
    if (!src.empty()) HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return 0;
}

// per-block tables of the unstructured ring: built once per context from the device's own mesh tables (cells, neighbours, flag bytes,
// diameters are downloaded: the context does not keep host copies)
int build_tables_u(knp_ctx* c, RingUState& S) {
    const MeshDev& m = c->m;
    const int64_t nc = m.nc, no = m.nc_owned, B = RB, nblk = (no + B - 1) / B;
    std::vector<int32_t> cells((size_t)nc * 4), nbr((size_t)nc * 4);
    std::vector<uint32_t> fflag((size_t)nc);
    std::vector<double> h((size_t)nc);
    HIPCHK(c, hipMemcpy(cells.data(), m.cells, sizeof(int32_t) * cells.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nbr.data(), m.nbr, sizeof(int32_t) * nbr.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(fflag.data(), m.fflag, sizeof(uint32_t) * fflag.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.data(), m.h, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> hsrc((size_t)nblk * UH, -1), vsrc((size_t)nblk * UV, 0);
    std::vector<uint16_t> hloc((size_t)no * 4, 0);
    std::vector<uint8_t> vloc((size_t)no * 8, 0);
    std::vector<double> hinv((size_t)no * 4, 0.0);
    std::vector<int32_t> vpos((size_t)m.nv, -1), touched;
    int64_t long0 = nblk;
    int hs = 0;
    for (int64_t b = 0; b < nblk && long0 == nblk; ++b) {
        const int64_t k0 = b * B, k1 = std::min(no, k0 + B);
        int nh = 0, nvb = 0;
        touched.clear();
        bool fits = true;
        auto vertex = [&](int32_t v) -> int {
            if (vpos[(size_t)v] < 0) {
                if (nvb >= UV) { fits = false; return 0; }
                vpos[(size_t)v] = nvb;
                vsrc[(size_t)b * UV + nvb] = v;
                touched.push_back(v);
                ++nvb;
            }
            return vpos[(size_t)v];
        };
        for (int64_t k = k0; k < k1; ++k)
            for (int a = 0; a < 4; ++a) vloc[(size_t)k * 8 + a] = (uint8_t)vertex(cells[(size_t)k * 4 + a]);
        for (int64_t k = k0; k < k1 && fits; ++k)
            for (int a = 0; a < 4; ++a) {
                const uint32_t fb = (fflag[(size_t)k] >> (8 * a)) & 0xffu;
                const uint32_t kind = (fb >> 2) & 3u;
                const int64_t nbk = nbr[(size_t)k * 4 + a];
                if ((kind != FK_SIPG && kind != FK_MEMBRANE) || nbk < 0) continue;
                const int j = (int)(fb & 3u);
                vloc[(size_t)k * 8 + 4 + a] = (uint8_t)vertex(cells[(size_t)nbk * 4 + j]);
                hinv[(size_t)k * 4 + a] = 2.0 / (h[(size_t)k] + h[(size_t)nbk]);
                if (nbk / B == b && nbk < no) { hloc[(size_t)k * 4 + a] = (uint16_t)(nbk - k0); continue; }
                if (nh >= UH) { fits = false; break; }
                hloc[(size_t)k * 4 + a] = (uint16_t)(B + nh);
                hsrc[(size_t)b * UH + nh] = (int32_t)(nbk * 4 + j);
                ++nh;
            }
        for (int32_t v : touched) vpos[(size_t)v] = -1;
        if (!fits) { long0 = b; break; }
        hs = std::max(hs, nh);
    }
    if (long0 == 0) return 0;                                          // nothing fits: the thread-per-cell kernels stay
    int rc = 0;
    rc |= up_u(c, &S.T.hb_src, hsrc);
    rc |= up_u(c, &S.T.hb_loc, hloc);
    rc |= up_u(c, &S.T.vb_src, vsrc);
    rc |= up_u(c, &S.T.vloc, vloc);
    rc |= up_u(c, &S.T.hinv4, hinv);
    if (rc) return rc;
    S.T.long0 = long0;
    S.T.hs = std::max(hs, 1);
    S.usable = true;
    if (getenv("KNP_DEBUG")) fprintf(stderr, "[knp] unstructured ring tables: %lld of %lld blocks, longest halo list %d\n", (long long)long0, (long long)nblk, hs);
    return 0;
}

// per-context tables.  The map is guarded (contexts may live on different host threads; std::map keeps references to other entries
// valid across insert / erase), a context's own entry is only touched by the thread that drives that context.
std::mutex g_states_mu;
std::map<const knp_ctx*, RingUState>& states_locked() {
    static std::map<const knp_ctx*, RingUState> s;
    return s;
}
RingUState& state_u(const knp_ctx* c) {
    std::lock_guard<std::mutex> lock(g_states_mu);
    return states_locked()[c];
}

}  // namespace

void ring_u_free(knp_ctx* c) {
    RingUTables T;
    {
        std::lock_guard<std::mutex> lock(g_states_mu);
        auto& st = states_locked();
        auto it = st.find(c);
        if (it == st.end()) return;
        T = it->second.T;
        st.erase(it);
    }
    hipFree(T.hb_src); hipFree(T.hb_loc); hipFree(T.vb_src); hipFree(T.vloc); hipFree(T.hinv4);
}

static size_t ring_u_lds(const knp_ctx* c, int which) {
    if (which == 0) return (size_t)USLOTS * EmiU::SLOT + 2 * ULISTB;
    const size_t slot = c->p.n_sys == 1 ? KnpU<1>::SLOT : KnpU<2>::SLOT;
    return (size_t)USLOTS * slot + 2 * ULISTB + sizeof(double) * (size_t)c->p.n_sys * KNP_MAX_MAT;
}

// which 0: EMI, 1: KNP.  3D P1 meshes WITHOUT geometry classes whose blocks fit the staging limits (320 out-of-block neighbours, 256
// vertices) and, for KNP, a material table for at most two solved species.  Returns the number of leading cells the ring covers (0: not
// usable); KNP_APPLY_RING_U=0 selects the thread-per-cell coordinate kernels (A/B runs).  Builds the tables at the first call.
int64_t ring_u_cells(knp_ctx* c, int which) {
    if (c->degree != 1 || c->m.dim != 3 || c->m.cls || c->m.nc_owned < RB || c->m.nv >= (int64_t(1) << 31)) return 0;
    if (env_int_u("KNP_APPLY_RING_U", 1) == 0 || env_int_u("KNP_APPLY_RING", 1) == 0) return 0;
    if (which == 1 && !(c->nmat > 0 && c->p.n_sys <= 2 && env_int_u("KNP_APPLY_MAT", 1) != 0)) return 0;
    if (ring_u_lds(c, which) > 160 * 1024) return 0;
    RingUState& S = state_u(c);
    if (!S.tried) {
        S.tried = true;
        if (build_tables_u(c, S)) { S.usable = false; return 0; }
    }
    return S.usable ? std::min<int64_t>(S.T.long0 * RB, c->m.nc_owned) : 0;
}

int ring_u_emi_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* kappa, double* y, int reserve_cus) {
    const RingUTables& T = state_u(c).T;
    const size_t lds = ring_u_lds(c, 0);
    if (!grant_lds_u(k_emi_apply_ring_u, lds)) { c->err = "hipFuncSetAttribute(k_emi_apply_ring_u) failed"; return -2; }
    hipLaunchKernelGGL(k_emi_apply_ring_u, grid_u(m, c->device, reserve_cus), dim3(RB + 64 * ULOADERS), lds, c->stream, m, T, x, kappa, y, c->p.C_phi,
                       c->p.tau_emi);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int ring_u_knp_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* gphi, double* y, const KnpArgs& ka, int reserve_cus) {
    const RingUTables& T = state_u(c).T;
    const size_t lds = ring_u_lds(c, 1);
    const dim3 g = grid_u(m, c->device, reserve_cus);
    if (c->p.n_sys == 1) {
        if (!grant_lds_u(k_knp_apply_ring_u<1>, lds)) { c->err = "hipFuncSetAttribute(k_knp_apply_ring_u) failed"; return -2; }
        hipLaunchKernelGGL((k_knp_apply_ring_u<1>), g, dim3(RB + 64 * ULOADERS), lds, c->stream, m, T, x, gphi, y, ka, (const uint8_t*)c->mat,
                           (const uint8_t*)c->nmat4, (const double*)c->dtab);
    } else {
        if (!grant_lds_u(k_knp_apply_ring_u<2>, lds)) { c->err = "hipFuncSetAttribute(k_knp_apply_ring_u) failed"; return -2; }
        hipLaunchKernelGGL((k_knp_apply_ring_u<2>), g, dim3(RB + 64 * ULOADERS), lds, c->stream, m, T, x, gphi, y, ka, (const uint8_t*)c->mat,
                           (const uint8_t*)c->nmat4, (const double*)c->dtab);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}
