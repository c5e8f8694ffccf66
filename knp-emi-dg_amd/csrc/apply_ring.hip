// Ring-staged P1 SIPG operator applies on (block-)structured 3D meshes: the same operators as apply_p1.hip
// (reference: src/knpemidg/solver.py:325-328, 346, 477, 509 for a_emi;  :586-594, 730, 771 for A_knp), restructured around the
// memory system of MI355X instead of around the thread:
//
//   * ONE persistent workgroup per CU = consumer waves (one cell per lane, 256-cell blocks) + 4 LOADER waves, one per SIMD;
//   * the loaders stream a block's records -- its own x / coefficient rows, the rows of the facet neighbours outside the block
//     (the per-block lists hb_src of knp_ctx_create, a gather through the per-lane SOURCE address) -- straight into LDS with
//     global_load_lds_dwordx4 (no VGPR destination, 1 KiB per wave instruction), TWO blocks ahead of the consumers, into a
//     three-slot (KNP) / four-slot (EMI) ring; a counted s_waitcnt vmcnt(N) retires exactly the block the consumers need next and leaves the following
//     one in flight across the workgroup barrier;
//   * the consumers read LDS only (own rows, neighbour rows, per-cell topology bytes, class records, material table) -- they issue
//     no global load at all, so nothing ever makes them wait for their y stores (loads and stores share the in-order vmcnt
//     counter): one raw s_barrier per block, no vmcnt drain.
//
// The thread-per-cell kernels of apply_p1.hip keep 33-50 KB of loads in flight per CU (three workgroups that alternate between a
// load phase and a compute phase) and reach 3.3 TB/s of moved bytes; the loader keeps two whole blocks (2 x 39 KB) in flight all
// the time and the same memory pattern runs at 5.3 TB/s (tools/microbench/glds_ring.hip, profiles/r03_glds_ring_probe.txt).
// LDS image of a slot: row-major 32-byte records [entry][4] (halves swizzled, dma_own_rows), entries [0, 256) = the block's cells,
// [256, 480) = halo list entries.
#include "cell_geom.hpp"
#include "ring_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <map>

namespace {

using namespace ring;
constexpr int RH = 224;           // halo entries staged per block (a multiple of 32; the BoxMesh lists have <= 224 entries)
constexpr int RHX = RH / 32;      // DMA instructions per halo row set (two lanes per row)
constexpr int RENT = RB + RH;
constexpr int RSLOTS = 3;
constexpr int RLISTB = 1024;      // bytes of one list buffer = one DMA instruction
constexpr int RNLIST = 4;
constexpr int RLOADERS = 4;       // loader waves per workgroup, one per SIMD
constexpr int META_F = 0, META_L = 1024, META_N = 3072, META_C = 4096, META_M = 5120, META_KNP = 6144, META_EMI = 5120;   // topology bytes in a slot (dma_meta)
constexpr int RCLS = 43;          // LDS stride of a geometry-class record (odd: lanes of different classes on different banks): vol + Gram (11) + cls_ext (32)

// ---- loader pieces -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma_list(const MeshDev& m, int64_t b, unsigned dst, int lane) {
    const int off = lane * 4 < m.hb_stride ? lane * 4 : 0;
    glds16(m.hb_src + b * m.hb_stride + off, dst);
}
// ---- geometry-class record in LDS: [0] vol, [1..10] Gram (upper triangle), [11 + 8 i ..] the derived facet coefficients of cls_ext ----
__device__ __forceinline__ void class_gram(const lds_double* rec, CellGeom<3>& K) {
    K.vol = rec[0];
    int q = 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) { K.G[a][b] = rec[q]; K.G[b][a] = rec[q]; ++q; }
}

// ================================================================================================================================
// KNP:  y_k = A_k x_k for all solved species (forms and notation: apply_p1.hip, k_knp_apply_halo)
// ================================================================================================================================
template <int NS> struct KnpRing {
    static constexpr int XB = RENT * 32;            // bytes of one species' rows
    static constexpr int G0 = NS * XB;              // own gphi rows [256][4]
    static constexpr int GH0 = G0 + RB * 32;        // halo gphi: [256][2], the half row that holds the neighbour's component j
    static constexpr int META0 = GH0 + 4096;        // topology bytes
    static constexpr int SLOT = META0 + META_KNP;
    static constexpr int NDATA = NS * 8 + NS * RHX + 8 + 4 + 6;
    static_assert(NDATA <= 63, "vmcnt is a 6-bit counter");
};

template <int NS, int I>
__device__ __forceinline__ void knp_facet_ring(const CellGeom<3>& K, uint32_t flags, unsigned loc, unsigned dsel, const double (*xv)[4],
                                               const double (*gx)[4], const double* gp, const double* Dk, const double* hvD,
                                               const double* zpsi, double tau,
                                               const lds_double* X, const lds_double* G, const lds_double* sD, const lds_double* ft,
                                               double (*y)[4]) {
    constexpr int D = 3, NV = 4;
    const uint32_t fb = (flags >> (8 * I)) & 0xffu;
    if (((fb >> 2) & 3u) != FK_SIPG) return;
    const unsigned j = fb & 3u;
    // class-level coefficients (MeshDev::cls_ext): gr = G_II / L_I, cf = neighbour-gradient weights, penalty and upwind factors
    const double gr = ft[8 * I], pen_geo = ft[8 * I + 4], nLI_DV = ft[8 * I + 5];
    double cf[D];
#pragma unroll
    for (int mm = 0; mm < D; ++mm) cf[mm] = ft[8 * I + 1 + mm];
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
    const double up_nb = fmax(-gp_nb, 0.0) * nLI_DV;
    const double penA = tau * pen_geo;
    const double hv = 0.5 * K.vol;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double xr[NV], xf[D];
        lds_row(X + (unsigned)k * (RENT * NV), loc, xr);
        const double xap = pick_apex<D>(xr, (int)j);
#pragma unroll
        for (int mm = 0; mm < D; ++mm) xf[mm] = pick_facet<D>(xr, mm, (int)j);
        const double Dn = sD[(unsigned)k * KNP_MAX_MAT + dsel];
        const double s_own = gx[k][I];                                             // (G x)_I = grad(u) . g_I, from the cell term
        double s_nb = xap * gr;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], cf[mm], s_nb);
        // penalty and upwind weights of the two traces:  D (pen - z psi un),  written so that each costs one FMA and one product
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

// per-cell topology bytes of a block, DMA'd by the loaders next to the nodal rows: flag bytes (4 B / cell), hb_loc (8 B), neighbour
// materials (4 B), class (2 B), material (1 B).  The device arrays are padded by 4 KB (abi.hip), so whole blocks can be read past nc.
struct CellMeta { uint32_t flags, nm; uint2 lw; unsigned cls, mymat; };
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ __forceinline__ CellMeta read_meta(const char* meta, unsigned t, bool knp) {
    CellMeta q;
    q.flags = ((const lds_u32*)(meta + META_F))[t];
    q.lw.x = ((const lds_u32*)(meta + META_L))[2 * t];
    q.lw.y = ((const lds_u32*)(meta + META_L))[2 * t + 1];
    q.cls = ((const lds_u16*)(meta + META_C))[t];
    q.nm = 0; q.mymat = 0;
    if (knp) {
        q.nm = ((const lds_u32*)(meta + META_N))[t];
        q.mymat = ((const lds_u8*)(meta + META_M))[t];
    }
    return q;
}

// NG consumer groups of four waves work on the SAME block, each on NS / NG of the species (A_knp has no cross-ion coupling): with
// NG = 2 every SIMD holds two consumer waves that hide each other's LDS and FMA latencies (one wave per SIMD leaves the consumers,
// not the loader, as the bottleneck: 54 us at 10^6 cells against 31 us for the loader alone).
template <int NS, int NG>
__global__ __launch_bounds__(RB * NG + 64 * RLOADERS) void k_knp_apply_ring(MeshDev m, const double* __restrict__ x, const double* __restrict__ gphi,
                                                                 double* __restrict__ yout, KnpArgs ka, const uint8_t* __restrict__ mat,
                                                                 const uint8_t* __restrict__ nmat4, const double* __restrict__ dtab, int dbg) {
    typedef KnpRing<NS> R;
    constexpr int NV = 4, KS = NS / NG, NTHREADS = RB * NG + 64 * RLOADERS;
    static_assert(KS * NG == NS, "species split evenly over the consumer groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_list = smem + RSLOTS * R::SLOT;
    double* s_cls = reinterpret_cast<double*>(s_list + RNLIST * RLISTB);        // [ncls][RCLS]: vol, Gram (11), then the 32 derived coefficients
    double* s_D = s_cls + m.ncls * RCLS;                                         // [NS][KNP_MAX_MAT]
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    const unsigned list0 = base + RSLOTS * R::SLOT;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const RingWalk w(m);
    if (w.blk(0) < 0) return;
    for (int i = threadIdx.x; i < m.ncls * 11; i += NTHREADS) s_cls[(i / 11) * RCLS + (i % 11)] = m.cls_table[(i / 11) * KNP_CLS_STRIDE + (i % 11)];
    for (int i = threadIdx.x; i < m.ncls * KNP_CLS_EXT; i += NTHREADS) s_cls[(i / KNP_CLS_EXT) * RCLS + 11 + (i % KNP_CLS_EXT)] = m.cls_ext[i];
    if (threadIdx.x < NS * KNP_MAX_MAT) s_D[threadIdx.x] = dtab[threadIdx.x];
    if (wave >= 4 * NG) {
        // ------------------------------------------------ loaders ------------------------------------------------
        // four loader waves, one per SIMD (a workgroup's waves are dealt round-robin over the SIMDs), each with its own share of a
        // block's pieces and its own vmcnt: a DMA piece costs its wave ~60 issue cycles, which on one wave is a quarter of a
        // consumer wave's time on that SIMD -- spread, it is the same small tax on every SIMD.
        //   loader 0: species 0 rows, flag + hb_loc bytes, the lists      loader 1: species 1 rows, class / material bytes
        //   loader 2: gphi rows + the halo's gphi halves                  loader 3: the halo rows of all species
        const int lw = wave - 4 * NG;
        constexpr int N0 = 8 + 3, N1 = (NS > 1 ? 8 : 0) + 3, N2 = 8 + 4, N3 = NS * RHX;
        auto list_dma = [&](int64_t n) {
            const int64_t b = w.blk(n);
            if (lw == 0 && b >= 0) dma_list(m, b, list0 + (unsigned)(n & (RNLIST - 1)) * RLISTB, lane);
        };
        auto data_dma = [&](int64_t n) {
            const int64_t b = w.blk(n);
            if (dbg & 4) return;                                       // timing probe: consumers alone (the waits below then retire at once)
            const unsigned slot = base + (unsigned)(n % RSLOTS) * R::SLOT;
            const lds_int* L = (const lds_int*)(s_list + (n & (RNLIST - 1)) * RLISTB);
            const int64_t c0 = b * RB;
            if (lw == 0) {
                dma_own_rows(x, c0, m.nc, slot, lane);
                glds16(m.fflag + c0 + 4 * lane, slot + R::META0 + META_F);
                glds16(m.hb_loc + (c0 + 2 * lane) * 4, slot + R::META0 + META_L);
                glds16(m.hb_loc + (c0 + 128 + 2 * lane) * 4, slot + R::META0 + META_L + 1024);
            } else if (lw == 1) {
                if (NS > 1) dma_own_rows(x + m.nc * NV, c0, m.nc, slot + R::XB, lane);
                glds16(m.cls + c0 + 8 * lane, slot + R::META0 + META_C);
                glds16(nmat4 + (c0 + 4 * lane) * 4, slot + R::META0 + META_N);
                glds16(mat + c0 + 16 * lane, slot + R::META0 + META_M);
            } else if (lw == 2) {
                dma_own_rows(gphi, c0, m.nc, slot + R::G0, lane);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int e = p * 64 + lane;
                    const int src = e < m.hb_stride ? L[e] : -1;
                    int64_t Kp = src >= 0 ? (int64_t)(src >> 2) : 0;
                    const int j = src >= 0 ? (src & 3) : 0;
                    if (dbg & 1) Kp = c0 + e < m.nc ? c0 + e : 0;
                    glds16(gphi + Kp * NV + (j >> 1) * 2, slot + R::GH0 + p * 1024);
                }
            } else {
#pragma unroll
                for (int p = 0; p < RHX; ++p) {
                    int64_t Kp = list_cell(L, p * 32 + (lane >> 1), m.hb_stride);
                    if (dbg & 1) Kp = c0 + p * 32 + (lane >> 1) < m.nc ? c0 + p * 32 + (lane >> 1) : 0;     // timing probe: halo rows = own rows (coalesced, cached)
#pragma unroll
                    for (int k = 0; k < NS; ++k) glds16(x + (int64_t)k * m.nc * NV + Kp * NV + swz_half(lane), slot + k * R::XB + RB * 32 + p * 1024);
                }
            }
        };
        // all but this loader's share of the youngest block have landed (its list piece, issued before that share, included)
        auto wait_older = [&]() {
            if (lw == 0) wait_vm<N0>(); else if (lw == 1) wait_vm<N1>(); else if (lw == 2) wait_vm<N2>(); else wait_vm<N3>();
        };
        list_dma(0); list_dma(1);
        wait_vm<0>();
        ring_barrier();                                                // A: every loader sees lists 0 and 1
        data_dma(0);
        list_dma(2);
        if (w.blk(1) >= 0) { data_dma(1); wait_older(); } else wait_vm<0>();
        ring_barrier();                                                // B: block 0 and list 2 have landed
        for (int64_t n = 0; w.blk(n) >= 0; ++n) {
            list_dma(n + 3);
            if (w.blk(n + 2) >= 0) { data_dma(n + 2); wait_older(); }  // block n + 2 stays in flight; list n + 3, block n + 1 have landed
            else wait_vm<0>();
            ring_barrier();
        }
        return;
    }
    // ------------------------------------------------ consumers ------------------------------------------------
    const unsigned t = threadIdx.x & (RB - 1);
    const int k0 = (wave >> 2) * KS;                                   // first species of this consumer group
    double zpsi[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) zpsi[k] = ka.z[k0 + k] * ka.psi;
    ring_barrier();                                                    // A: class / material tables staged
    ring_barrier();                                                    // B
    for (int64_t n = 0;; ++n) {
        const int64_t b = w.blk(n);
        if (b < 0) break;
        const int64_t c = b * RB + t;
        if (c >= m.c_begin && c < m.c_end) {
            const char* slot = smem + (n % RSLOTS) * R::SLOT;
            const CellMeta cur = read_meta(slot + R::META0, t, true);
            const lds_double* X = (const lds_double*)slot + k0 * (RENT * NV);
            const lds_double* G = (const lds_double*)(slot + R::G0);
            const lds_double* rec = TO_LDS(s_cls) + cur.cls * RCLS;
            const lds_double* sD = TO_LDS(s_D) + k0 * KNP_MAX_MAT;
            CellGeom<3> K;
            class_gram(rec, K);
            double xv[KS][NV], y[KS][NV], gp[NV], Dk[KS];
            lds_row(G, t, gp);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                lds_row(X + k * (RENT * NV), t, xv[k]);
                Dk[k] = sD[k * KNP_MAX_MAT + cur.mymat];
            }
            const double mw = ka.inv_dt * K.vol / 20.0;
            double gx[KS][NV], hvD[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) hvD[k] = 0.5 * K.vol * Dk[k];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
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
            const lds_double* ft = rec + 11;
            if (!(dbg & 2)) {
            knp_facet_ring<KS, 0>(K, cur.flags, cur.lw.x & 0xffffu, cur.nm & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, ft, y);
            knp_facet_ring<KS, 1>(K, cur.flags, cur.lw.x >> 16, (cur.nm >> 8) & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, ft, y);
            knp_facet_ring<KS, 2>(K, cur.flags, cur.lw.y & 0xffffu, (cur.nm >> 16) & 0xffu, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, ft, y);
            knp_facet_ring<KS, 3>(K, cur.flags, cur.lw.y >> 16, cur.nm >> 24, xv, gx, gp, Dk, hvD, zpsi, ka.tau, X, G, sD, ft, y);
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) store_nodal<3>(yout + (int64_t)(k0 + k) * m.nc * NV, c, y[k]);
        }
        ring_barrier();                                                // the loader refills this slot in its next iteration
    }
}

// ================================================================================================================================
// EMI:  y = A(kappa) x      (forms and notation: apply_p1.hip, emi_facet_cls)
// ================================================================================================================================
struct EmiRing {
    static constexpr int XB = RENT * 32;            // x rows, then kappa rows
    static constexpr int META0 = 2 * XB;
    static constexpr int SLOT = META0 + META_EMI;
    static constexpr int NDATA = 8 + 8 + 2 * RHX + 4;
};

template <int I>
__device__ __forceinline__ void emi_facet_ring(const CellGeom<3>& K, uint32_t flags, unsigned loc, const double* xv, const double* gx,
                                               const double* kv, double C_phi, double tau, const lds_double* X, const lds_double* KA,
                                               const lds_double* ft, double* y) {
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
    const double sqG_DV = ft[8 * I + 6];                                  // sqrt(G_II) D vol = facet area
    if (kind == FK_MEMBRANE) {
        const double w = C_phi * sqG_DV * FacetConst<D>::mass;
#pragma unroll
        for (int mm = 0; mm < D; ++mm) y[mm + (mm >= I)] = fma(w, sdu + du[mm], y[mm + (mm >= I)]);
        return;
    }
    const double gr = ft[8 * I];
    const double s_own = gx[I];                                            // (G x)_I from the cell term
    double s_nb = xap * gr;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) s_nb = fma(xf[mm], ft[8 * I + 1 + mm], s_nb);
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
    const double pw = tau * ft[8 * I + 4] * FacetConst<D>::trip;
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

// Two consumer groups of four waves, STAGGERED by half a block: a block is worked on during two barrier intervals (first half: own
// rows, cell term, facets 0 and 1; second half: facets 2 and 3, store), group n % 2 starts block n in interval n, so every SIMD
// holds one wave in each half and the loader's schedule (one block issued per interval, two in flight) is unchanged.  A block's
// slot lives for four intervals (issued in n - 2, read in n and n + 1): four 32 KB slots.
constexpr int EMI_SLOTS = 4;
__global__ __launch_bounds__(2 * RB + 64 * RLOADERS) void k_emi_apply_ring(MeshDev m, const double* __restrict__ x, const double* __restrict__ kappa,
                                                                double* __restrict__ yout, double C_phi, double tau) {
    typedef EmiRing R;
    constexpr int NV = 4, NTHREADS = 2 * RB + 64 * RLOADERS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_list = smem + EMI_SLOTS * R::SLOT;
    double* s_cls = reinterpret_cast<double*>(s_list + RNLIST * RLISTB);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    const unsigned list0 = base + EMI_SLOTS * R::SLOT;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const RingWalk w(m);
    if (w.blk(0) < 0) return;
    for (int i = threadIdx.x; i < m.ncls * 11; i += NTHREADS) s_cls[(i / 11) * RCLS + (i % 11)] = m.cls_table[(i / 11) * KNP_CLS_STRIDE + (i % 11)];
    for (int i = threadIdx.x; i < m.ncls * KNP_CLS_EXT; i += NTHREADS) s_cls[(i / KNP_CLS_EXT) * RCLS + 11 + (i % KNP_CLS_EXT)] = m.cls_ext[i];
    if (wave >= 8) {
        // loaders (see k_knp_apply_ring):  0: x rows + the lists   1: kappa rows   2: x halo rows + flag / class bytes   3: kappa halo rows + hb_loc
        const int lw = wave - 8;
        constexpr int N0 = 8, N1 = 8, N2 = RHX + 2, N3 = RHX + 2;
        auto list_dma = [&](int64_t n) {
            const int64_t b = w.blk(n);
            if (lw == 0 && b >= 0) dma_list(m, b, list0 + (unsigned)(n & (RNLIST - 1)) * RLISTB, lane);
        };
        auto data_dma = [&](int64_t n) {
            const int64_t c0 = w.blk(n) * RB;
            const unsigned slot = base + (unsigned)(n & (EMI_SLOTS - 1)) * R::SLOT;
            const lds_int* L = (const lds_int*)(s_list + (n & (RNLIST - 1)) * RLISTB);
            if (lw == 0) dma_own_rows(x, c0, m.nc, slot, lane);
            else if (lw == 1) dma_own_rows(kappa, c0, m.nc, slot + R::XB, lane);
            else {
                const double* v = lw == 2 ? x : kappa;
                const unsigned dst = slot + (lw == 2 ? 0 : R::XB) + RB * 32;
#pragma unroll
                for (int p = 0; p < RHX; ++p) glds16(v + list_cell(L, p * 32 + (lane >> 1), m.hb_stride) * NV + swz_half(lane), dst + p * 1024);
                if (lw == 2) {
                    glds16(m.fflag + c0 + 4 * lane, slot + R::META0 + META_F);
                    glds16(m.cls + c0 + 8 * lane, slot + R::META0 + META_C);
                } else {
                    glds16(m.hb_loc + (c0 + 2 * lane) * 4, slot + R::META0 + META_L);
                    glds16(m.hb_loc + (c0 + 128 + 2 * lane) * 4, slot + R::META0 + META_L + 1024);
                }
            }
        };
        auto wait_older = [&]() {
            if (lw == 0) wait_vm<N0>(); else if (lw == 1) wait_vm<N1>(); else if (lw == 2) wait_vm<N2>(); else wait_vm<N3>();
        };
        list_dma(0); list_dma(1);
        wait_vm<0>();
        ring_barrier();
        data_dma(0);
        list_dma(2);
        if (w.blk(1) >= 0) { data_dma(1); wait_older(); } else wait_vm<0>();
        ring_barrier();
        for (int64_t n = 0;; ++n) {                                    // interval n: block n's first half, block n - 1's second half
            if (w.blk(n) < 0) { wait_vm<0>(); ring_barrier(); break; } // the last interval has second halves only
            list_dma(n + 3);
            if (w.blk(n + 2) >= 0) { data_dma(n + 2); wait_older(); }
            else wait_vm<0>();
            ring_barrier();
        }
        return;
    }
    const unsigned t = threadIdx.x & (RB - 1);
    const int g = wave >> 2;
    ring_barrier();
    ring_barrier();
    CellGeom<3> K;
    double xv[NV], kv[NV], yv[NV], gx[NV];
    const lds_double *X = nullptr, *KA = nullptr, *ft = nullptr;
    CellMeta cur;
    cur.flags = 0; cur.lw = make_uint2(0u, 0u); cur.cls = 0; cur.nm = 0; cur.mymat = 0;
    int64_t ccur = -1;
    bool valid = false;
    for (int64_t n = 0;; ++n) {
        const int64_t b = w.blk(n);
        if ((n & 1) == g) {
            if (b >= 0) {                                              // first half of block n
                ccur = b * RB + t;
                valid = ccur >= m.c_begin && ccur < m.c_end;
                if (valid) {
                    const char* slot = smem + (n & (EMI_SLOTS - 1)) * R::SLOT;
                    cur = read_meta(slot + R::META0, t, false);
                    X = (const lds_double*)slot;
                    KA = (const lds_double*)(slot + R::XB);
                    const lds_double* rec = TO_LDS(s_cls) + cur.cls * RCLS;
                    ft = rec + 11;
                    class_gram(rec, K);
                    lds_row(X, t, xv);
                    lds_row(KA, t, kv);
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
                    emi_facet_ring<0>(K, cur.flags, cur.lw.x & 0xffffu, xv, gx, kv, C_phi, tau, X, KA, ft, yv);
                    emi_facet_ring<1>(K, cur.flags, cur.lw.x >> 16, xv, gx, kv, C_phi, tau, X, KA, ft, yv);
                }
            }
        } else if (n >= 1 && valid) {                                  // second half of block n - 1
            emi_facet_ring<2>(K, cur.flags, cur.lw.y & 0xffffu, xv, gx, kv, C_phi, tau, X, KA, ft, yv);
            emi_facet_ring<3>(K, cur.flags, cur.lw.y >> 16, xv, gx, kv, C_phi, tau, X, KA, ft, yv);
            store_nodal<3>(yout, ccur, yv);
        }
        ring_barrier();
        if (b < 0) break;
    }
}

int env_int_ring(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int device_cus(int device) {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

// more than 64 KB of dynamic LDS per workgroup has to be granted per kernel, once
template <typename KernelT> bool ring_grant_lds(KernelT kernel, size_t lds) {
    static std::map<const void*, size_t> granted;
    auto it = granted.find((const void*)kernel);
    if (it != granted.end() && it->second >= lds) return true;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    granted[(const void*)kernel] = lds;
    return true;
}

// one workgroup per CU (the ring takes most of a CU's LDS), a multiple of 8, not more than 8 per 8 blocks
dim3 ring_grid(const MeshDev& m, int device, int reserve_cus) {
    const int64_t nblk = (m.c_end - 1) / RB - m.c_begin / RB + 1;
    const int cus = std::max(device_cus(device) - std::max(reserve_cus, 0), 8);
    int64_t per_xcd = std::max<int64_t>(1, std::min<int64_t>(cus / 8, (nblk + 7) / 8));
    // tests only: KNP_RING_WG = workgroups of the launch (rounded down to a multiple of 8), so that small oracle-sized meshes put many
    // blocks on a workgroup (slot reuse, list-buffer wrap, counted vmcnt with two blocks in flight: the steady state of the r=2 runs)
    const int wg = env_int_ring("KNP_RING_WG", 0);
    if (wg >= 8) per_xcd = std::min<int64_t>(per_xcd, wg / 8);
    return dim3((unsigned)(8 * per_xcd));
}

}  // namespace

static size_t ring_lds_bytes(const knp_ctx* c, int which) {
    const size_t tables = RNLIST * RLISTB + sizeof(double) * (size_t)c->m.ncls * RCLS;
    if (which == 0) return (size_t)EMI_SLOTS * EmiRing::SLOT + tables;
    const size_t slot = c->p.n_sys == 1 ? KnpRing<1>::SLOT : KnpRing<2>::SLOT;
    return (size_t)RSLOTS * slot + tables + sizeof(double) * (size_t)c->p.n_sys * KNP_MAX_MAT;
}

// which 0: EMI, 1: KNP.  Structured 3D P1 meshes with class records, halo lists of at most 224 entries and (KNP) a material table for
// at most two solved species; KNP_APPLY_RING=0 selects the thread-per-cell kernels of apply_p1.hip (A/B runs)
bool ring_usable(const knp_ctx* c, int which) {
    if (c->degree != 1 || c->m.dim != 3 || !c->m.cls || c->m.ncls > 32 || !c->m.hb_src || c->m.hb_stride <= 0 || c->m.hb_stride > RH) return false;
    if (env_int_ring("KNP_APPLY_RING", 1) == 0 || ring_lds_bytes(c, which) > 160 * 1024) return false;
    if (which == 0) return env_int_ring("KNP_EMI_RING", 1) != 0;
    return c->nmat > 0 && c->p.n_sys <= 2 && env_int_ring("KNP_APPLY_MAT", 1) != 0 && env_int_ring("KNP_APPLY_HALO", 1) != 0;
}

int ring_emi_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* kappa, double* y, int reserve_cus) {
    const size_t lds = ring_lds_bytes(c, 0);
    if (!ring_grant_lds(k_emi_apply_ring, lds)) { c->err = "hipFuncSetAttribute(k_emi_apply_ring) failed"; return -2; }
    hipLaunchKernelGGL(k_emi_apply_ring, ring_grid(m, c->device, reserve_cus), dim3(2 * RB + 64 * RLOADERS), lds, c->stream, m, x, kappa, y, c->p.C_phi, c->p.tau_emi);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int ring_knp_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* gphi, double* y, const KnpArgs& ka, int reserve_cus) {
    const int ns = c->p.n_sys;
    const size_t lds = ring_lds_bytes(c, 1);
    const dim3 g = ring_grid(m, c->device, reserve_cus);
    const bool split = ns == 2 && env_int_ring("KNP_RING_SPLIT", 0) != 0;       // two consumer groups, one species each (measured equal to one group: 47.6 vs 46.3 us)
#define RING_KNP_LAUNCH(NS_, NG_)                                                                                                     \
    do {                                                                                                                              \
        if (!ring_grant_lds(k_knp_apply_ring<NS_, NG_>, lds)) { c->err = "hipFuncSetAttribute(k_knp_apply_ring) failed"; return -2; }  \
        hipLaunchKernelGGL((k_knp_apply_ring<NS_, NG_>), g, dim3(RB * NG_ + 64 * RLOADERS), lds, c->stream, m, x, gphi, y, ka, (const uint8_t*)c->mat,  \
                           (const uint8_t*)c->nmat4, (const double*)c->dtab, env_int_ring("KNP_RING_DEBUG", 0));                     \
    } while (0)
    if (ns == 1) RING_KNP_LAUNCH(1, 1);
    else if (split) RING_KNP_LAUNCH(2, 2);
    else RING_KNP_LAUNCH(2, 1);
#undef RING_KNP_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
}
