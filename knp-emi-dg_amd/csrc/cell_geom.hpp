// Per-thread affine geometry of one P1 simplex cell in "Gram form".
//
// Everything the SIPG forms need from the geometry of a cell K and of the neighbour K' behind
// facet i is expressed through
//   g_a = grad lambda_a (own cell),  G_ab = g_a . g_b,  vol,
//   L_a = lambda_a(X_o)   (own barycentric coordinates of the NEIGHBOUR's apex vertex X_o; L_i < 0),
// because on all of space  lambda'_apex = lambda_i / L_i  and  lambda'_b = lambda_b - L_b lambda'_apex:
//   grad u' . g_i = sum_m x'_m (G_{a_m i} - L_{a_m} G_ii / L_i) + x'_apex G_ii / L_i,
//   vol' = -L_i vol,   area_i = sqrt(G_ii) D vol,   area_i * (grad w . n_i) = -D vol (grad w . g_i).
// So no normal vector, height or foot point is ever formed, and only the penalty / membrane terms
// need one square root per facet.  Geometry is recomputed from vertex coordinates every launch
// (coordinates stay cache resident: 5.6 MB at 10^6 tets) instead of streaming >= 96 B/cell of Jacobians.
//
// Conventions shared with the host tables and the oracle:
//  * cells hold ascending vertex ids, local facet i is opposite local vertex i;
//  * facet vertex m (m = 0..D-1) is the cell's local vertex  m + (m >= i);
//    for the neighbour (whose local facet index is j) it is  m + (m >= j).
#pragma once
#include "knpemi_internal.hpp"

// ---- fast reciprocal / square root (hardware seed + 2 Newton steps; operands are normal, positive
//      or negative, never 0/inf/nan on valid meshes) ------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    g = fma(fma(-g, g, x), h, g);
    return g;
}

template <int D> struct CellGeom {
    static constexpr int NV = D + 1;
    double X0[D];      // coordinates of local vertex 0
    double g[NV][D];   // grad lambda_a
    double G[NV][NV];  // Gram matrix (symmetric, both triangles filled)
    double vol;
};

template <int D> __device__ __forceinline__ void load_vertex(const double* __restrict__ coords, int v, double* out);
template <> __device__ __forceinline__ void load_vertex<3>(const double* __restrict__ coords, int v, double* out) {
    const double2 a = *reinterpret_cast<const double2*>(coords + 4 * (int64_t)v);
    const double b = coords[4 * (int64_t)v + 2];
    out[0] = a.x; out[1] = a.y; out[2] = b;
}
template <> __device__ __forceinline__ void load_vertex<2>(const double* __restrict__ coords, int v, double* out) {
    const double2 a = *reinterpret_cast<const double2*>(coords + 2 * (int64_t)v);
    out[0] = a.x; out[1] = a.y;
}

template <int D> __device__ __forceinline__ double dotD(const double* a, const double* b) {
    double s = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < D; ++k) s = fma(a[k], b[k], s);
    return s;
}

template <int D> __device__ __forceinline__ void load_nodal(const double* __restrict__ p, int64_t c, double* v);
template <> __device__ __forceinline__ void load_nodal<3>(const double* __restrict__ p, int64_t c, double* v) {
    const double2 a = *reinterpret_cast<const double2*>(p + 4 * c);
    const double2 b = *reinterpret_cast<const double2*>(p + 4 * c + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <> __device__ __forceinline__ void load_nodal<2>(const double* __restrict__ p, int64_t c, double* v) {
    v[0] = p[3 * c]; v[1] = p[3 * c + 1]; v[2] = p[3 * c + 2];
}
template <int D> __device__ __forceinline__ void store_nodal(double* __restrict__ p, int64_t c, const double* v);
template <> __device__ __forceinline__ void store_nodal<3>(double* __restrict__ p, int64_t c, const double* v) {
    *reinterpret_cast<double2*>(p + 4 * c) = make_double2(v[0], v[1]);
    *reinterpret_cast<double2*>(p + 4 * c + 2) = make_double2(v[2], v[3]);
}
template <> __device__ __forceinline__ void store_nodal<2>(double* __restrict__ p, int64_t c, const double* v) {
    p[3 * c] = v[0]; p[3 * c + 1] = v[1]; p[3 * c + 2] = v[2];
}

template <int D> __device__ __forceinline__ void load_cell_ints(const int32_t* __restrict__ p, int64_t c, int* v);
template <> __device__ __forceinline__ void load_cell_ints<3>(const int32_t* __restrict__ p, int64_t c, int* v) {
    const int4 a = *reinterpret_cast<const int4*>(p + 4 * c);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void load_cell_ints<2>(const int32_t* __restrict__ p, int64_t c, int* v) {
    v[0] = p[3 * c]; v[1] = p[3 * c + 1]; v[2] = p[3 * c + 2];
}

// gradients + volume from vertex coordinates X[a][:]
template <int D> __device__ __forceinline__ void gradients(const double (*X)[D], CellGeom<D>& K);

template <> __device__ __forceinline__ void gradients<3>(const double (*X)[3], CellGeom<3>& K) {
    double e1[3], e2[3], e3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e1[k] = X[1][k] - X[0][k];
        e2[k] = X[2][k] - X[0][k];
        e3[k] = X[3][k] - X[0][k];
        K.X0[k] = X[0][k];
    }
    const double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    const double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    const double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
    const double inv = fast_rcp(det);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        K.g[1][k] = c23[k] * inv;
        K.g[2][k] = c31[k] * inv;
        K.g[3][k] = c12[k] * inv;
        K.g[0][k] = -(K.g[1][k] + K.g[2][k] + K.g[3][k]);
    }
    K.vol = fabs(det) * (1.0 / 6.0);
}

template <> __device__ __forceinline__ void gradients<2>(const double (*X)[2], CellGeom<2>& K) {
    const double e1[2] = {X[1][0] - X[0][0], X[1][1] - X[0][1]};
    const double e2[2] = {X[2][0] - X[0][0], X[2][1] - X[0][1]};
    K.X0[0] = X[0][0]; K.X0[1] = X[0][1];
    const double det = e1[0] * e2[1] - e1[1] * e2[0];
    const double inv = fast_rcp(det);
    K.g[1][0] = e2[1] * inv;  K.g[1][1] = -e2[0] * inv;
    K.g[2][0] = -e1[1] * inv; K.g[2][1] = e1[0] * inv;
    K.g[0][0] = -(K.g[1][0] + K.g[2][0]);
    K.g[0][1] = -(K.g[1][1] + K.g[2][1]);
    K.vol = fabs(det) * 0.5;
}

template <int D> __device__ __forceinline__ void cell_geometry_from(const double (*X)[D], CellGeom<D>& K) {
    gradients<D>(X, K);
#pragma unroll
    for (int a = 0; a <= D; ++a)
#pragma unroll
        for (int b = a; b <= D; ++b) {
            const double v = dotD<D>(K.g[a], K.g[b]);
            K.G[a][b] = v;
            K.G[b][a] = v;
        }
}

template <int D> __device__ __forceinline__ void load_cell_geometry(const MeshDev& m, const int* verts, CellGeom<D>& K) {
    double X[D + 1][D];
#pragma unroll
    for (int a = 0; a <= D; ++a) load_vertex<D>(m.coords, verts[a], X[a]);
    cell_geometry_from<D>(X, K);
}

// Workgroup-local staging of what facet neighbours need from a cell (x, coefficient, diameter, vertex
// coordinates): ~85 % of the neighbours of a Morton-ordered block live in the same block and are served
// from LDS instead of per-lane L1 gathers.
typedef __attribute__((address_space(3))) double lds_double;   // explicit LDS pointers: ds_read, never flat_load
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) int lds_int;
#define TO_LDS(p) ((const lds_double*)(p))

template <int D> struct StageView {
    const lds_double* x;    // [nvalid][NV]      (per species: + k * xstride)
    const lds_double* k;    // [nvalid][NV]      coefficient (kappa | gphi)
    const lds_double* h;    // [nvalid]
    const lds_double* X;    // [nvalid][NV][D]
    int64_t c0;
    unsigned nvalid;        // 0 disables staging (setup kernels)
    const double* rec;      // geometry-class record of the cell (MODE 3), else null
    const lds_double* lrec; // the same record in LDS (MODE 4)
    const lds_double* lext; // derived facet coefficients of the class in LDS (MeshDev::cls_ext), or null
};

template <int D> __device__ __forceinline__ void lds_nodal(const lds_double* base, unsigned idx, double* v) {
    if (D == 3) {
        typedef double __attribute__((ext_vector_type(2))) vdouble2;
        typedef __attribute__((address_space(3))) vdouble2 lds_vdouble2;
        const vdouble2 a = *(const lds_vdouble2*)(base + 4 * idx);
        const vdouble2 b = *(const lds_vdouble2*)(base + 4 * idx + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
        v[0] = base[3 * idx]; v[1] = base[3 * idx + 1]; v[2] = base[3 * idx + 2];
    }
}

// L_a = lambda_a(Xo) for all own vertices a
template <int D> __device__ __forceinline__ void apex_bary(const CellGeom<D>& K, const double* Xo, double* L) {
    double dv[D];
#pragma unroll
    for (int k = 0; k < D; ++k) dv[k] = Xo[k] - K.X0[k];
#pragma unroll
    for (int a = 0; a <= D; ++a) L[a] = dotD<D>(K.g[a], dv) + (a == 0 ? 1.0 : 0.0);
}

// pick entry (m + (m >= j)) of a (D+1)-vector held in registers; j is a runtime 2-bit value.
// Written as explicit selects so that the compiler keeps the vector in registers.
template <int D> __device__ __forceinline__ double pick_facet(const double* v, int mm, int j) {
    return (mm >= j) ? v[mm + 1] : v[mm];
}
template <int D> __device__ __forceinline__ double pick_apex(const double* v, int j);
template <> __device__ __forceinline__ double pick_apex<3>(const double* v, int j) {
    const double lo = (j & 1) ? v[1] : v[0];
    const double hi = (j & 1) ? v[3] : v[2];
    return (j & 2) ? hi : lo;
}
template <> __device__ __forceinline__ double pick_apex<2>(const double* v, int j) {
    const double lo = (j & 1) ? v[1] : v[0];
    return (j & 2) ? v[2] : lo;
}

// XCD-aware block remap: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so give
// each XCD one contiguous chunk of cells to keep facet-neighbour gathers in its own L2.
// The grid is a multiple of 8 blocks; the mapping is a bijection; speed only, never correctness.
__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nblocks) {
    const int64_t chunk = nblocks >> 3;
    return (b & 7) * chunk + (b >> 3);
}
