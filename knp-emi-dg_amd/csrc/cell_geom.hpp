// Per-thread affine geometry of one P1 simplex cell and of the neighbour across one facet,
// recomputed from vertex coordinates on every operator apply (coordinates stay cache
// resident: 4.2 MB at 10^6 tets) instead of streaming ~130 B/cell of stored Jacobians.
//
// Conventions shared with the host tables (knpemidg/tables.py) and the oracle:
//  * cells hold ascending vertex ids, local facet i is opposite local vertex i;
//  * facet vertex m (m = 0..D-1) is the cell's local vertex  m + (m >= i);
//    for the neighbour (whose local facet index is j) it is  m + (m >= j).
#pragma once
#include "knpemi_internal.hpp"

template <int D> struct CellGeom {
    static constexpr int NV = D + 1;
    double X[NV][D];   // vertex coordinates
    double g[NV][D];   // grad lambda_a
    double vol;
    double h2;         // squared cell diameter (longest edge)
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
    for (int k = 1; k < D; ++k) s += a[k] * b[k];
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

// gradients, volume, diameter from the vertex coordinates already in K.X
template <int D> __device__ __forceinline__ void cell_geometry(CellGeom<D>& K);

template <> __device__ __forceinline__ void cell_geometry<3>(CellGeom<3>& K) {
    double e1[3], e2[3], e3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e1[k] = K.X[1][k] - K.X[0][k];
        e2[k] = K.X[2][k] - K.X[0][k];
        e3[k] = K.X[3][k] - K.X[0][k];
    }
    double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
    const double inv = 1.0 / det;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        K.g[1][k] = c23[k] * inv;
        K.g[2][k] = c31[k] * inv;
        K.g[3][k] = c12[k] * inv;
        K.g[0][k] = -(K.g[1][k] + K.g[2][k] + K.g[3][k]);
    }
    K.vol = fabs(det) * (1.0 / 6.0);
    double e23[3], e13[3], e12[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e12[k] = K.X[2][k] - K.X[1][k];
        e13[k] = K.X[3][k] - K.X[1][k];
        e23[k] = K.X[3][k] - K.X[2][k];
    }
    double h2 = fmax(dotD<3>(e1, e1), dotD<3>(e2, e2));
    h2 = fmax(h2, dotD<3>(e3, e3));
    h2 = fmax(h2, dotD<3>(e12, e12));
    h2 = fmax(h2, dotD<3>(e13, e13));
    h2 = fmax(h2, dotD<3>(e23, e23));
    K.h2 = h2;
}

template <> __device__ __forceinline__ void cell_geometry<2>(CellGeom<2>& K) {
    double e1[2] = {K.X[1][0] - K.X[0][0], K.X[1][1] - K.X[0][1]};
    double e2[2] = {K.X[2][0] - K.X[0][0], K.X[2][1] - K.X[0][1]};
    const double det = e1[0] * e2[1] - e1[1] * e2[0];
    const double inv = 1.0 / det;
    K.g[1][0] = e2[1] * inv;  K.g[1][1] = -e2[0] * inv;
    K.g[2][0] = -e1[1] * inv; K.g[2][1] = e1[0] * inv;
    K.g[0][0] = -(K.g[1][0] + K.g[2][0]);
    K.g[0][1] = -(K.g[1][1] + K.g[2][1]);
    K.vol = fabs(det) * 0.5;
    double e12[2] = {K.X[2][0] - K.X[1][0], K.X[2][1] - K.X[1][1]};
    K.h2 = fmax(fmax(dotD<2>(e1, e1), dotD<2>(e2, e2)), dotD<2>(e12, e12));
}

template <int D> __device__ __forceinline__ void load_cell_geometry(const MeshDev& m, const int* verts, CellGeom<D>& K) {
#pragma unroll
    for (int a = 0; a <= D; ++a) load_vertex<D>(m.coords, verts[a], K.X[a]);
    cell_geometry<D>(K);
}

// Geometry of facet i of cell K and of the neighbour cell behind it.
template <int D> struct FacetGeom {
    double n[D];        // unit normal, outward from K
    double area;
    double dn[D + 1];   // grad lambda_a . n  for the own cell
    double hp;          // height of the neighbour's apex above the facet
    double beta[D];     // barycentric coordinates (facet vertices) of the apex's foot point
    double hN2;         // squared diameter of the neighbour
};

// own part (no neighbour needed)
template <int D, int I> __device__ __forceinline__ void facet_own(const CellGeom<D>& K, FacetGeom<D>& F) {
    const double gi2 = dotD<D>(K.g[I], K.g[I]);
    const double gin = sqrt(gi2);
    const double rin = 1.0 / gin;
    F.area = gin * (double)D * K.vol;
#pragma unroll
    for (int k = 0; k < D; ++k) F.n[k] = -K.g[I][k] * rin;
#pragma unroll
    for (int a = 0; a <= D; ++a) F.dn[a] = dotD<D>(K.g[a], F.n);
}

// neighbour part from its apex vertex Xo
template <int D, int I> __device__ __forceinline__ void facet_neighbour(const CellGeom<D>& K, const double* Xo, FacetGeom<D>& F) {
    constexpr int a0 = (0 >= I) ? 1 : 0;
    double d0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) d0[k] = Xo[k] - K.X[a0][k];
    F.hp = dotD<D>(d0, F.n);
    double hN2 = 0.0;
#pragma unroll
    for (int mm = 0; mm < D; ++mm) {
        const int a = mm + (mm >= I);
        double dv[D];
#pragma unroll
        for (int k = 0; k < D; ++k) dv[k] = Xo[k] - K.X[a][k];
        hN2 = fmax(hN2, dotD<D>(dv, dv));
        F.beta[mm] = 1.0 + dotD<D>(K.g[a], dv) - F.hp * F.dn[a];
        // facet edges belong to the neighbour too
#pragma unroll
        for (int m2 = mm + 1; m2 < D; ++m2) {
            const int b = m2 + (m2 >= I);
            double ev[D];
#pragma unroll
            for (int k = 0; k < D; ++k) ev[k] = K.X[b][k] - K.X[a][k];
            hN2 = fmax(hN2, dotD<D>(ev, ev));
        }
    }
    F.hN2 = hN2;
}

// pick entry (m + (m >= j)) of a (D+1)-vector held in registers, j is a runtime value
template <int D> __device__ __forceinline__ double pick_facet(const double* v, int mm, int j) {
    return (mm >= j) ? v[mm + 1] : v[mm];
}
template <int D> __device__ __forceinline__ double pick_apex(const double* v, int j) {
    double r = v[0];
#pragma unroll
    for (int a = 1; a <= D; ++a) r = (j == a) ? v[a] : r;
    return r;
}

// XCD-aware block remap: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so give
// each XCD one contiguous chunk of cells to keep facet-neighbour gathers in its own L2.
// The grid is a multiple of 8 blocks; the mapping is a bijection; speed only, never correctness.
__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nblocks) {
    const int64_t chunk = nblocks >> 3;
    return (b & 7) * chunk + (b >> 3);
}
