// Threaded host kernels of the preconditioner SETUP (knpemidg/amg.py): the sparse matrix products of the smoothed-aggregation
// hierarchy (prolongator smoothing A P, Galerkin products R (A P)) were > 80 % of the non-LAPACK setup time in single-threaded
// scipy.  Row-parallel Gustavson SpGEMM in two passes (symbolic row counts, prefix sum, numeric fill with sorted columns).
// The reference builds BoomerAMG inside PETSc at every solve (src/knpemidg/solver.py:433, 505, 688, 767); this is the
// corresponding setup work of this build's auxiliary-space hierarchy.  Plain C ABI, no device code.
#include "../../include/knpemi_hip.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

template <typename F> void parallel_rows(int64_t n, int nthreads, F f) {
    if (nthreads < 1) nthreads = 1;
    if (n < 4096 || nthreads == 1) { f(0, n, 0); return; }
    std::vector<std::thread> pool;
    const int64_t chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([=]() { f(lo, hi, t); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// C = A * B for CSR matrices with int32 indices / fp64 values; A is [n x k], B is [k x m].
// Cp must hold n + 1 entries; *Cj / *Cx are malloc'ed here (release with knp_host_free).  Columns of every row come out sorted.
int knp_host_spgemm(int64_t n, int64_t m, const int32_t* Ap, const int32_t* Aj, const double* Ax, const int32_t* Bp, const int32_t* Bj,
                    const double* Bx, int32_t* Cp, int32_t** Cj, double** Cx, int nthreads) {
    if (!Ap || !Bp || !Cp || !Cj || !Cx || n < 0 || m < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    // one pass: every thread forms its contiguous chunk of rows (sorted columns) into its own buffer, a prefix sum over the row lengths places
    // the buffers (the symbolic pass of the first version walked every product twice)
    const int nt = n < 4096 ? 1 : nthreads;
    const int64_t chunk = (n + nt - 1) / nt;
    std::vector<std::vector<int32_t>> bj((size_t)nt);
    std::vector<std::vector<double>> bx((size_t)nt);
    std::vector<int32_t> len((size_t)n, 0);
    auto work = [&](int t) {
        const int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
        std::vector<int32_t> pos((size_t)m, -1);
        std::vector<std::pair<int32_t, double>> row;
        auto& oj = bj[(size_t)t];
        auto& ox = bx[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) {
            row.clear();
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int32_t k = Aj[p];
                const double a = Ax[p];
                for (int32_t q = Bp[k]; q < Bp[k + 1]; ++q) {
                    const int32_t j = Bj[q];
                    if (pos[j] < 0) { pos[j] = (int32_t)row.size(); row.emplace_back(j, a * Bx[q]); }
                    else row[pos[j]].second += a * Bx[q];
                }
            }
            for (auto& e : row) pos[e.first] = -1;
            std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& u, const std::pair<int32_t, double>& v) { return u.first < v.first; });
            for (auto& e : row) { oj.push_back(e.first); ox.push_back(e.second); }
            len[(size_t)i] = (int32_t)row.size();
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back([&, t]() { work(t); });
        for (auto& th : pool) th.join();
    }
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (nnz > 2147483647LL) return -3;
        Cp[i] = (int32_t)nnz;
        nnz += len[(size_t)i];
    }
    if (nnz > 2147483647LL) return -3;
    Cp[n] = (int32_t)nnz;
    *Cj = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    *Cx = (double*)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
    if (!*Cj || !*Cx) return -2;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * chunk;
        if (lo >= n) break;
        std::memcpy(*Cj + Cp[lo], bj[(size_t)t].data(), sizeof(int32_t) * bj[(size_t)t].size());
        std::memcpy(*Cx + Cp[lo], bx[(size_t)t].data(), sizeof(double) * bx[(size_t)t].size());
    }
    return 0;
}

void knp_host_free(void* p) { std::free(p); }

// y = A x, row-parallel (power iterations of the spectral-radius estimates)
int knp_host_spmv(int64_t n, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* x, double* y, int nthreads) {
    if (!Ap || !x || !y) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            double s = 0.0;
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p) s += Ax[p] * x[Aj[p]];
            y[i] = s;
        }
    });
    return 0;
}


// Gram-form geometry of every simplex cell (the host setup's copy of csrc/cell_geom.hpp): vol[c] and G[c][a][b] = grad lambda_a . grad
// lambda_b, from vertex coordinates.  Replaces a batched numpy inverse + determinant + einsum over the cells (0.6 s at 10^6 tets).
int knp_host_cell_gram(int64_t nc, int dim, const double* coords, const int32_t* cells, double* vol, double* G, int nthreads) {
    if (!coords || !cells || !vol || !G || (dim != 2 && dim != 3) || nc < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    const int nv = dim + 1;
    parallel_rows(nc, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t c = lo; c < hi; ++c) {
            const int32_t* cv = cells + c * nv;
            double g[4][3] = {{0.0}};
            double det;
            const double* X0 = coords + (int64_t)cv[0] * dim;
            if (dim == 3) {
                double e[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int k = 0; k < 3; ++k) e[a][k] = coords[(int64_t)cv[a + 1] * 3 + k] - X0[k];
                const double c23[3] = {e[1][1] * e[2][2] - e[1][2] * e[2][1], e[1][2] * e[2][0] - e[1][0] * e[2][2], e[1][0] * e[2][1] - e[1][1] * e[2][0]};
                const double c31[3] = {e[2][1] * e[0][2] - e[2][2] * e[0][1], e[2][2] * e[0][0] - e[2][0] * e[0][2], e[2][0] * e[0][1] - e[2][1] * e[0][0]};
                const double c12[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
                det = e[0][0] * c23[0] + e[0][1] * c23[1] + e[0][2] * c23[2];
                const double inv = 1.0 / det;
                for (int k = 0; k < 3; ++k) {
                    g[1][k] = c23[k] * inv; g[2][k] = c31[k] * inv; g[3][k] = c12[k] * inv;
                    g[0][k] = -(g[1][k] + g[2][k] + g[3][k]);
                }
                vol[c] = std::abs(det) / 6.0;
            } else {
                const double e1[2] = {coords[(int64_t)cv[1] * 2] - X0[0], coords[(int64_t)cv[1] * 2 + 1] - X0[1]};
                const double e2[2] = {coords[(int64_t)cv[2] * 2] - X0[0], coords[(int64_t)cv[2] * 2 + 1] - X0[1]};
                det = e1[0] * e2[1] - e1[1] * e2[0];
                const double inv = 1.0 / det;
                g[1][0] = e2[1] * inv; g[1][1] = -e2[0] * inv;
                g[2][0] = -e1[1] * inv; g[2][1] = e1[0] * inv;
                g[0][0] = -(g[1][0] + g[2][0]); g[0][1] = -(g[1][1] + g[2][1]);
                vol[c] = std::abs(det) / 2.0;
            }
            double* Gc = G + c * nv * nv;
            for (int a = 0; a < nv; ++a)
                for (int b = 0; b < nv; ++b) {
                    double v = 0.0;
                    for (int k = 0; k < dim; ++k) v += g[a][k] * g[b][k];
                    Gc[a * nv + b] = v;
                }
        }
    });
    return 0;
}

// out[p] = sum of src[order[k]] over k in [starts[p], starts[p + 1])  (starts has nseg + 1 entries): the scatter-add of per-cell blocks
// into the values of a CSR matrix whose pattern (sort order of the entry keys) is known -- numpy's gather + add.reduceat, threaded.
// Summation order inside a segment = ascending k, as in the numpy version.
int knp_host_segment_sum(int64_t nseg, const int64_t* starts, const int64_t* order, const double* src, double* out, int nthreads) {
    if (!starts || !order || !src || !out || nseg < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    parallel_rows(nseg, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t p = lo; p < hi; ++p) {
            double acc = 0.0;
            for (int64_t k = starts[p]; k < starts[p + 1]; ++k) acc += src[order[k]];
            out[p] = acc;
        }
    });
    return 0;
}

// CSR pattern of a matrix assembled from per-cell dense blocks scattered through a dof map: entry e = (c, a, b) of blocks [nc][nd][nd]
// lands in row dof[c][a], column dof[c][b].  Returns what numpy's stable argsort of the keys (row * n + col) gives -- order[] (entries
// grouped by (row, col), ties in ascending e), starts[] (first position of every distinct key; nseg + 1 entries), cols[] / indptr[] of the
// CSR matrix -- by a stable counting sort over the rows and a stable sort by column inside each row, threaded over the rows
// (numpy's argsort of 16 M int64 keys was the largest single item of a first assembly: 0.3 s).
int knp_host_block_pattern(int64_t nc, int nd, int64_t n, const int32_t* dof, int64_t* order, int64_t* starts, int32_t* cols, int32_t* indptr,
                           int64_t* nseg_out, int nthreads) {
    if (!dof || !order || !starts || !cols || !indptr || !nseg_out || nc < 0 || nd < 1 || n < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    const int64_t nent = nc * nd * nd;
    for (int64_t i = 0; i < nc * nd; ++i)
        if (dof[i] < 0 || dof[i] >= n) return -1;
    // stable counting sort over the rows, threaded: thread t owns a contiguous range of cells and its own histogram; its entries of row r
    // go behind those of the threads before it, so every row lists its entries in ascending e
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, nc / 65536 + 1));
    const int64_t cchunk = (nc + T - 1) / T;
    std::vector<std::vector<int32_t>> hist((size_t)T, std::vector<int32_t>((size_t)n, 0));
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t]() {
                std::vector<int32_t>& h = hist[(size_t)t];
                for (int64_t i = t * cchunk * nd; i < std::min(nc, (t + 1) * cchunk) * nd; ++i) h[(size_t)dof[i]] += nd;
            });
        for (auto& th : pool) th.join();
    }
    std::vector<int64_t> rowptr((size_t)n + 1, 0);
    std::vector<std::vector<int64_t>> base((size_t)T, std::vector<int64_t>((size_t)n, 0));
    for (int64_t r = 0; r < n; ++r) {
        int64_t acc = rowptr[(size_t)r];
        for (int t = 0; t < T; ++t) { base[(size_t)t][(size_t)r] = acc; acc += hist[(size_t)t][(size_t)r]; }
        rowptr[(size_t)r + 1] = acc;
    }
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t]() {
                std::vector<int64_t>& f = base[(size_t)t];
                for (int64_t c = t * cchunk; c < std::min(nc, (t + 1) * cchunk); ++c)
                    for (int a = 0; a < nd; ++a) {
                        int64_t& q = f[(size_t)dof[c * nd + a]];
                        for (int b = 0; b < nd; ++b) order[q++] = (c * nd + a) * nd + b;
                    }
            });
        for (auto& th : pool) th.join();
    }
    // per row: stable sort by column, count the distinct columns
    std::vector<int32_t> rowcnt((size_t)n, 0);
    auto col_of = [&](int64_t e) { return dof[(e / ((int64_t)nd * nd)) * nd + (e % nd)]; };
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t r = lo; r < hi; ++r) {
            int64_t* b = order + rowptr[(size_t)r];
            int64_t* e = order + rowptr[(size_t)r + 1];
            std::stable_sort(b, e, [&](int64_t x, int64_t y) { return col_of(x) < col_of(y); });
            int32_t cnt = 0, last = -1;
            for (int64_t* p = b; p < e; ++p) { const int32_t cj = col_of(*p); if (p == b || cj != last) { ++cnt; last = cj; } }
            rowcnt[(size_t)r] = cnt;
        }
    });
    int64_t nseg = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (nseg > 2147483647LL) return -3;
        indptr[r] = (int32_t)nseg;
        nseg += rowcnt[(size_t)r];
    }
    if (nseg > 2147483647LL) return -3;
    indptr[n] = (int32_t)nseg;
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t r = lo; r < hi; ++r) {
            int64_t q = indptr[r];
            int32_t last = -1;
            for (int64_t p = rowptr[(size_t)r]; p < rowptr[(size_t)r + 1]; ++p) {
                const int32_t cj = col_of(order[p]);
                if (p == rowptr[(size_t)r] || cj != last) { starts[q] = p; cols[q] = cj; ++q; last = cj; }
            }
        }
    });
    starts[nseg] = nent;
    *nseg_out = nseg;
    return 0;
}

// Facet table of a simplicial mesh (knpemidg/mesh.py: Mesh._build_facets): facets are numbered by first appearance in (cell, local facet)
// order, local facet i is the one opposite local vertex i, its vertices are the cell's other vertices in the cell's (ascending) order.
//   cell_facets[nc][nv], facets[<= nc nv][d], facet_cells[..][2] (side 0 = the lower cell, -1 on the boundary), facet_local[..][2]
// Returns the number of facets, or -1 (bad input) / -2 (a facet shared by more than two cells).
int64_t knp_host_build_facets(int64_t nc, int nv, const int32_t* cells, int32_t* cell_facets, int32_t* facets, int32_t* facet_cells,
                              int8_t* facet_local) {
    if (!cells || !cell_facets || !facets || !facet_cells || !facet_local || nc < 0 || (nv != 3 && nv != 4)) return -1;
    const int d = nv - 1;
    // flat open-addressing table on a packed 64-bit key (three 21-bit vertex ids, or two 32-bit ones): 4 nc look-ups, most of them
    // first insertions -- a node-based std::unordered_map spent 0.5 s on them at 10^6 tets
    int64_t vmax = 0;
    for (int64_t i = 0; i < nc * nv; ++i) { if (cells[i] < 0) return -1; vmax = std::max<int64_t>(vmax, cells[i]); }
    if (nv == 4 && vmax >= (int64_t(1) << 21)) return -1;                      // the caller falls back to its sort-based builder
    int bits = 4;
    while ((int64_t(1) << bits) < 3 * nc * nv / 2 + 16) ++bits;
    const uint64_t mask = (uint64_t(1) << bits) - 1;
    auto facet_key = [&](int64_t c, int i, int32_t* v) {
        int q = 0;
        v[0] = v[1] = v[2] = -1;
        for (int a = 0; a < nv; ++a)
            if (a != i) v[q++] = cells[c * nv + a];
        const uint64_t key = nv == 4 ? ((uint64_t)v[0] << 42) | ((uint64_t)v[1] << 21) | (uint64_t)v[2] : ((uint64_t)v[0] << 32) | (uint64_t)(uint32_t)v[1];
        uint64_t h = key * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        return std::make_pair(key, h & mask);
    };
    int nthreads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* ev = std::getenv("KNP_SETUP_THREADS")) nthreads = std::max(1, std::atoi(ev));
    if (nc * nv >= (int64_t(1) << 21) && nthreads > 1 && nc * nv < 2147483647LL) {
        // PARALLEL build with the same numbering (facets by first appearance in (cell, local facet) order e = c nv + i, side 0 = the first
        // appearance): (1) every e claims / finds its key's slot with a compare-and-swap and leaves e in one of the slot's two places;
        // (2) e is a facet's first appearance if it is the smaller of the two; (3) prefix sum over the first appearances = facet ids;
        // (4) the first appearance of every facet writes its rows (both sides), every e its cell_facets entry.  At 8 x 10^6 tets the
        // serial loop below spends 1.5 s in 32 M cache-missing probes; sixteen threads keep sixteen misses in flight.
        const size_t nslot = (size_t)1 << bits;
        const int64_t ne = nc * nv;
        std::vector<uint64_t> keys(nslot);
        std::vector<int32_t> ea(nslot), eb(nslot), slot_of((size_t)ne);
        parallel_rows((int64_t)nslot, nthreads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t k = lo; k < hi; ++k) { keys[(size_t)k] = ~uint64_t(0); ea[(size_t)k] = -1; eb[(size_t)k] = -1; }
        });
        std::vector<int> bad((size_t)nthreads, 0);
        parallel_rows(ne, nthreads, [&](int64_t lo, int64_t hi, int t) {
            int32_t v[3];
            for (int64_t e = lo; e < hi; ++e) {
                const auto ks = facet_key(e / nv, (int)(e % nv), v);
                uint64_t slot = ks.second;
                for (;;) {
                    uint64_t cur = __atomic_load_n(&keys[slot], __ATOMIC_RELAXED);
                    if (cur == ~uint64_t(0)) {
                        uint64_t expect = ~uint64_t(0);
                        if (__atomic_compare_exchange_n(&keys[slot], &expect, ks.first, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                        cur = expect;
                    }
                    if (cur == ks.first) break;
                    slot = (slot + 1) & mask;
                }
                slot_of[(size_t)e] = (int32_t)slot;
                int32_t expect = -1;
                if (!__atomic_compare_exchange_n(&ea[slot], &expect, (int32_t)e, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    expect = -1;
                    if (!__atomic_compare_exchange_n(&eb[slot], &expect, (int32_t)e, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) bad[(size_t)t] = 1;   // a third cell on one facet
                }
            }
        });
        for (int b : bad) if (b) return -2;
        auto first_of = [&](int32_t slot) { return eb[(size_t)slot] < 0 ? ea[(size_t)slot] : std::min(ea[(size_t)slot], eb[(size_t)slot]); };
        std::vector<int32_t> e0_of((size_t)ne);                               // first appearance of the facet behind e
        parallel_rows(ne, nthreads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t e = lo; e < hi; ++e) e0_of[(size_t)e] = first_of(slot_of[(size_t)e]);
        });
        std::vector<int32_t> fid((size_t)ne);                                 // id of the facet whose first appearance is e (valid there only)
        int64_t nfp = 0;
        for (int64_t e = 0; e < ne; ++e) {                                    // sequential reads: the prefix sum is not worth threads
            fid[(size_t)e] = (int32_t)nfp;
            nfp += e0_of[(size_t)e] == (int32_t)e ? 1 : 0;
        }
        parallel_rows(ne, nthreads, [&](int64_t lo, int64_t hi, int) {
            int32_t v[3];
            for (int64_t e = lo; e < hi; ++e) {
                const int32_t slot = slot_of[(size_t)e];
                const int32_t e0 = e0_of[(size_t)e];
                const int64_t f = fid[(size_t)e0];
                cell_facets[e] = (int32_t)f;
                if (e0 != (int32_t)e) continue;
                facet_key(e / nv, (int)(e % nv), v);
                for (int a = 0; a < d; ++a) facets[f * d + a] = v[a];
                const int32_t e1 = eb[(size_t)slot] < 0 ? -1 : std::max(ea[(size_t)slot], eb[(size_t)slot]);
                facet_cells[2 * f] = (int32_t)(e / nv);
                facet_local[2 * f] = (int8_t)(e % nv);
                facet_cells[2 * f + 1] = e1 < 0 ? -1 : (int32_t)(e1 / nv);
                facet_local[2 * f + 1] = e1 < 0 ? (int8_t)-1 : (int8_t)(e1 % nv);
            }
        });
        return nfp;
    }
    std::vector<uint64_t> keys((size_t)1 << bits, ~uint64_t(0));
    std::vector<int32_t> vals((size_t)1 << bits, -1);
    int64_t nf = 0;
    for (int64_t c = 0; c < nc; ++c)
        for (int i = 0; i < nv; ++i) {
            int32_t v[3] = {-1, -1, -1};
            int q = 0;
            for (int a = 0; a < nv; ++a)
                if (a != i) v[q++] = cells[c * nv + a];
            const uint64_t key = nv == 4 ? ((uint64_t)v[0] << 42) | ((uint64_t)v[1] << 21) | (uint64_t)v[2] : ((uint64_t)v[0] << 32) | (uint64_t)(uint32_t)v[1];
            uint64_t h = key * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            uint64_t slot = h & mask;
            while (keys[slot] != key && keys[slot] != ~uint64_t(0)) slot = (slot + 1) & mask;
            int32_t f;
            if (keys[slot] != key) {
                f = (int32_t)nf++;
                keys[slot] = key;
                vals[slot] = f;
                for (int a = 0; a < d; ++a) facets[(int64_t)f * d + a] = v[a];
                facet_cells[2 * (int64_t)f] = (int32_t)c; facet_cells[2 * (int64_t)f + 1] = -1;
                facet_local[2 * (int64_t)f] = (int8_t)i; facet_local[2 * (int64_t)f + 1] = -1;
            } else {
                f = vals[slot];
                if (facet_cells[2 * (int64_t)f + 1] >= 0) return -2;
                facet_cells[2 * (int64_t)f + 1] = (int32_t)c;
                facet_local[2 * (int64_t)f + 1] = (int8_t)i;
            }
            cell_facets[c * nv + i] = f;
        }
    return nf;
}

// Geometry classes of a 3D mesh (knpemidg/_abi.py: geometry_classes): cells whose own shape, neighbour-apex positions and diameters
// coincide to `quantum` (an absolute length: tol x median diameter) share a class.  feat = 26 quantised numbers per cell (9 edge-vector
// components, 12 apex offsets, own diameter, four neighbour diameters), hashed; classes are numbered by first appearance and every
// member is compared with its class representative number by number (a hash collision returns -4).
//   nbr[nc][4] neighbour cell or -1, nbj[nc][4] its local facet; cls[nc] out, first[max_classes] out (representative cell of each class)
// Returns the class count, or -3 when there are more than max_classes.
int64_t knp_host_geometry_classes(int64_t nc, const double* coords, const int32_t* cells, const int32_t* nbr, const int8_t* nbj, double quantum,
                                  int64_t max_classes, int32_t* cls, int64_t* first, int nthreads) {
    if (!coords || !cells || !nbr || !nbj || !cls || !first || nc < 0 || !(quantum > 0.0)) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    constexpr int NF = 26;
    std::vector<double> h((size_t)nc);
    parallel_rows(nc, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t c = lo; c < hi; ++c) {
            double h2 = 0.0;
            for (int a = 0; a < 4; ++a)
                for (int b = a + 1; b < 4; ++b) {
                    double d2 = 0.0;
                    for (int k = 0; k < 3; ++k) { const double d = coords[(int64_t)cells[c * 4 + a] * 3 + k] - coords[(int64_t)cells[c * 4 + b] * 3 + k]; d2 += d * d; }
                    h2 = std::max(h2, d2);
                }
            h[(size_t)c] = std::sqrt(h2);
        }
    });
    std::vector<int64_t> feat((size_t)nc * NF);
    std::vector<uint64_t> hash((size_t)nc);
    const double inv = 1.0 / quantum;
    parallel_rows(nc, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t c = lo; c < hi; ++c) {
            int64_t* q = feat.data() + c * NF;
            const double* X0 = coords + (int64_t)cells[c * 4] * 3;
            int n = 0;
            for (int a = 1; a < 4; ++a)
                for (int k = 0; k < 3; ++k) q[n++] = (int64_t)std::nearbyint((coords[(int64_t)cells[c * 4 + a] * 3 + k] - X0[k]) * inv);
            for (int i = 0; i < 4; ++i) {
                const int32_t nb = nbr[c * 4 + i];
                for (int k = 0; k < 3; ++k)
                    q[n++] = nb >= 0 ? (int64_t)std::nearbyint((coords[(int64_t)cells[(int64_t)nb * 4 + nbj[c * 4 + i]] * 3 + k] - X0[k]) * inv) : 0;
            }
            q[n++] = (int64_t)std::nearbyint(h[(size_t)c] * inv);
            for (int i = 0; i < 4; ++i) q[n++] = nbr[c * 4 + i] >= 0 ? (int64_t)std::nearbyint(h[(size_t)nbr[c * 4 + i]] * inv) : 0;
            uint64_t hv = 0xcbf29ce484222325ull;
            for (int k = 0; k < NF; ++k) { hv ^= (uint64_t)q[k] + 0x9E3779B97F4A7C15ull + (hv << 6) + (hv >> 2); hv *= 0x100000001b3ull; }
            hash[(size_t)c] = hv;
        }
    });
    std::unordered_map<uint64_t, int32_t> ids;
    int64_t ncls = 0;
    for (int64_t c = 0; c < nc; ++c) {
        auto it = ids.find(hash[(size_t)c]);
        if (it == ids.end()) {
            if (ncls >= max_classes) return -3;
            ids.emplace(hash[(size_t)c], (int32_t)ncls);
            first[ncls] = c;
            cls[c] = (int32_t)ncls++;
        } else {
            cls[c] = it->second;
            if (std::memcmp(feat.data() + c * NF, feat.data() + first[it->second] * NF, sizeof(int64_t) * NF) != 0) return -4;
        }
    }
    return ncls;
}
// The dense coarsest-level inverse after LAPACK's potri: `a` [n x n] row-major holds valid numbers in its UPPER triangle only (a Fortran-order
// lower triangle seen through the C view).  out = the full symmetric matrix rounded to fp32, *maxabs = the largest magnitude (NaN / Inf if
// any entry is not finite) -- numpy's tril + add + transpose + astype + isfinite + abs().max(), one pass over 64 x 64 tiles, threaded.
int knp_host_sym_to_f32(int64_t n, const double* a, float* out, double* maxabs, int nthreads) {
    if (!a || !out || !maxabs || n < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    constexpr int64_t T = 64;
    const int64_t nt = (n + T - 1) / T;
    std::vector<double> part((size_t)std::max(1, nthreads), 0.0);
    auto body = [&](int64_t lo, int64_t hi, int t) {
        double mx = 0.0;
        bool bad = false;
        for (int64_t bi = lo; bi < hi; ++bi)
            for (int64_t bj = bi; bj < nt; ++bj) {
                const int64_t i1 = std::min(n, (bi + 1) * T), j1 = std::min(n, (bj + 1) * T);
                for (int64_t i = bi * T; i < i1; ++i)
                    for (int64_t j = std::max(i, bj * T); j < j1; ++j) {
                        const double v = a[i * n + j];
                        const float f = (float)v;
                        out[i * n + j] = f;
                        out[j * n + i] = f;
                        const double av = std::fabs(v);
                        if (!(av <= 1.7976931348623157e308)) bad = true;
                        if (av > mx) mx = av;
                    }
            }
        part[(size_t)t] = bad ? std::nan("") : mx;
    };
    // tile rows near the top hold more tiles: deal them round-robin instead of in contiguous chunks
    if (nthreads == 1 || nt < 4) body(0, nt, 0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&, t]() {
                double mx = 0.0;
                bool bad = false;
                for (int64_t bi = t; bi < nt; bi += nthreads) {
                    body(bi, bi + 1, t);
                    if (part[(size_t)t] != part[(size_t)t]) bad = true; else mx = std::max(mx, part[(size_t)t]);
                }
                part[(size_t)t] = bad ? std::nan("") : mx;
            });
        for (auto& th : pool) th.join();
    }
    double mx = 0.0;
    for (double v : part) { if (v != v) { mx = v; break; } mx = std::max(mx, v); }
    *maxabs = mx;
    return 0;
}
// Distance-2 maximal-independent-set aggregation of a strength graph (CSR pattern, symmetric, no diagonal): the vectorised Luby rounds of
// knpemidg/amg.py (mis2_aggregate) pass for pass -- same keys, same max-reductions, hence the same aggregates -- without numpy's gather +
// reduceat temporaries (4 of them per round).  key[n]: distinct positive priorities; agg[n]: aggregate of every node; returns their number.
int64_t knp_host_mis2_aggregate(int64_t n, const int32_t* indptr, const int32_t* indices, const double* key, int64_t* agg, int nthreads) {
    if (!indptr || !indices || !key || !agg || n < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    const double NINF = -HUGE_VAL;
    std::vector<int8_t> state((size_t)n, 0);                 // 0 undecided, 1 root, 2 covered
    std::vector<double> a((size_t)n), b((size_t)n), c((size_t)n);
    int64_t undecided = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (indptr[i + 1] == indptr[i]) state[(size_t)i] = 1;
        else ++undecided;
    }
    // out[i] = max(in[i], max over neighbours in[j])
    auto spread = [&](const std::vector<double>& in, std::vector<double>& out) {
        parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                double m = in[(size_t)i];
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) m = std::max(m, in[(size_t)indices[k]]);
                out[(size_t)i] = m;
            }
        });
    };
    while (undecided > 0) {
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = state[(size_t)i] == 0 ? key[i] : NINF;
        spread(a, b);
        spread(b, c);
        for (int64_t i = 0; i < n; ++i)
            if (state[(size_t)i] == 0 && a[(size_t)i] >= c[(size_t)i]) state[(size_t)i] = 1;
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = state[(size_t)i] == 1 ? 1.0 : NINF;
        spread(a, b);
        spread(b, c);
        undecided = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (state[(size_t)i] == 0 && c[(size_t)i] > 0.0) state[(size_t)i] = 2;
            if (state[(size_t)i] == 0) ++undecided;
        }
    }
    int64_t nroot = 0;
    for (int64_t i = 0; i < n; ++i) agg[i] = state[(size_t)i] == 1 ? nroot++ : -1;
    // distance-1 members take the neighbouring root with the largest number, distance-2 members follow a neighbour (two synchronous rounds)
    std::vector<int64_t> next((size_t)n);
    for (int round = 0; round < 2; ++round) {
        parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                int64_t best = agg[i];
                if (best < 0)
                    for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) best = std::max(best, agg[indices[k]]);
                next[(size_t)i] = best;
            }
        });
        std::memcpy(agg, next.data(), sizeof(int64_t) * (size_t)n);
    }
    for (int64_t i = 0; i < n; ++i)
        if (agg[i] < 0) agg[i] = nroot++;                    // safety: singletons
    return nroot;
}
// median over the cells of the per-axis extent (max - min over the cell's vertices): the length scale of the Morton curve below
// (numpy: np.median(xc.max(axis=1) - xc.min(axis=1), axis=0) on the gathered [nc][nv][d] coordinates)
int knp_host_cell_extent_median(int64_t nc, int nv, int d, const double* coords, const int32_t* cells, double* out) {
    if (!coords || !cells || !out || nc < 1 || nv < 2 || d < 1 || d > 3) return -1;
    std::vector<double> ext((size_t)nc);
    for (int k = 0; k < d; ++k) {
        for (int64_t c = 0; c < nc; ++c) {
            double lo = coords[(int64_t)cells[c * nv] * d + k], hi = lo;
            for (int a = 1; a < nv; ++a) {
                const double v = coords[(int64_t)cells[c * nv + a] * d + k];
                lo = std::min(lo, v);
                hi = std::max(hi, v);
            }
            ext[(size_t)c] = hi - lo;
        }
        const int64_t h = nc / 2;
        std::nth_element(ext.begin(), ext.begin() + h, ext.end());
        double med = ext[(size_t)h];
        if (nc % 2 == 0) med = (*std::max_element(ext.begin(), ext.begin() + h) + med) / 2.0;
        out[k] = med;
    }
    return 0;
}

// Stable argsort of n points along a Morton (Z-order) curve, bit for bit what knpemidg/_abi.py: morton_order does in ~70 numpy passes:
// p = x / scale (scale may be NULL), q = trunc((p - min) / largest span * (2^bits - 1)) with bits = 21 (3D) / 31 (otherwise), code = bits of
// the axes interleaved (axis 0 lowest), order = stable sort by code.  With conn != NULL the points are the midpoints
// ((x_0 + x_1) + ... + x_{nv-1}) / nv of the rows of conn [n][nv] (cells -> their midpoints) instead of pts itself.
int knp_host_morton_order(int64_t n, int d, const double* pts, const int32_t* conn, int nv, const double* scale, int64_t* order, int nthreads) {
    if (!pts || !order || n < 0 || d < 1 || d > 3 || (conn && nv < 1)) return -1;
    if (n == 0) return 0;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<double> p((size_t)n * d);
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i)
            for (int k = 0; k < d; ++k) {
                double v;
                if (conn) {
                    v = pts[(int64_t)conn[i * nv] * d + k];
                    for (int a = 1; a < nv; ++a) v += pts[(int64_t)conn[i * nv + a] * d + k];
                    v /= (double)nv;
                } else v = pts[i * d + k];
                p[(size_t)(i * d + k)] = scale ? v / scale[k] : v;
            }
    });
    double lo[3], hi[3];
    for (int k = 0; k < d; ++k) { lo[k] = p[(size_t)k]; hi[k] = p[(size_t)k]; }
    for (int64_t i = 1; i < n; ++i)
        for (int k = 0; k < d; ++k) {
            lo[k] = std::min(lo[k], p[(size_t)(i * d + k)]);
            hi[k] = std::max(hi[k], p[(size_t)(i * d + k)]);
        }
    double span = hi[0] - lo[0];
    for (int k = 1; k < d; ++k) span = std::max(span, hi[k] - lo[k]);
    const int bits = d == 3 ? 21 : 31;
    const uint64_t qmax = (uint64_t(1) << bits) - 1;
    const double M = (double)qmax;
    std::vector<uint64_t> code((size_t)n, 0), code2((size_t)n);
    if (span != 0.0)
        parallel_rows(n, nthreads, [&](int64_t a, int64_t b, int) {
            for (int64_t i = a; i < b; ++i) {
                uint64_t cd = 0;
                for (int k = 0; k < d; ++k) {
                    const double t = (p[(size_t)(i * d + k)] - lo[k]) / span * M;
                    uint64_t q = (uint64_t)t;
                    if (q > qmax) q = qmax;
                    for (int bb = 0; bb < bits; ++bb) cd |= ((q >> bb) & 1ull) << (bb * d + k);
                }
                code[(size_t)i] = cd;
            }
        });
    // LSD radix sort of (code, index), 8 bits per pass; passes whose byte is the same everywhere are skipped
    std::vector<int64_t> idx2((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[i] = i;
    uint64_t* ck = code.data();
    uint64_t* ck2 = code2.data();
    int64_t* ix = order;
    int64_t* ix2 = idx2.data();
    // per pass: every thread counts the bytes of its contiguous chunk, the chunks' counts are turned into start positions bucket by bucket
    // (bucket-major, chunk-minor: the stable order), every thread scatters its chunk
    const int T = n >= (int64_t(1) << 18) ? std::max(1, std::min(nthreads, 64)) : 1;
    const int64_t chunk = (n + T - 1) / T;
    std::vector<int64_t> hist((size_t)T * 256);
    for (int pass = 0; pass < 8; ++pass) {
        const int sh = 8 * pass;
        std::fill(hist.begin(), hist.end(), 0);
        auto each_chunk = [&](auto fn) {
            if (T == 1) { fn(0); return; }
            std::vector<std::thread> pool;
            for (int t = 0; t < T; ++t) pool.emplace_back([=]() { fn(t); });
            for (auto& th : pool) th.join();
        };
        each_chunk([&](int t) {
            int64_t* h = hist.data() + (size_t)t * 256;
            const int64_t lo_ = t * chunk, hi_ = std::min(n, lo_ + chunk);
            for (int64_t i = lo_; i < hi_; ++i) ++h[(ck[i] >> sh) & 0xffu];
        });
        bool single = false;
        int64_t run = 0;
        for (int v = 0; v < 256; ++v) {
            int64_t tot = 0;
            for (int t = 0; t < T; ++t) {
                const int64_t c = hist[(size_t)t * 256 + v];
                hist[(size_t)t * 256 + v] = run + tot;
                tot += c;
            }
            if (tot == n) single = true;
            run += tot;
        }
        if (single) continue;
        each_chunk([&](int t) {
            int64_t* h = hist.data() + (size_t)t * 256;
            const int64_t lo_ = t * chunk, hi_ = std::min(n, lo_ + chunk);
            for (int64_t i = lo_; i < hi_; ++i) {
                const int64_t pos = h[(ck[i] >> sh) & 0xffu]++;
                ck2[pos] = ck[i];
                ix2[pos] = ix[i];
            }
        });
        std::swap(ck, ck2);
        std::swap(ix, ix2);
    }
    if (ix != order) std::memcpy(order, ix, sizeof(int64_t) * (size_t)n);
    return 0;
}
// cell -> (neighbour cell behind local facet i or -1, the neighbour's local index of that facet) from the facet tables
// (numpy: two take_along_axis gathers over [nc][nv][2] temporaries)
int knp_host_cell_neighbours(int64_t nc, int nv, const int32_t* cell_facets, const int32_t* facet_cells, const int8_t* facet_local, int32_t* nb,
                             int8_t* nj, int nthreads) {
    if (!cell_facets || !facet_cells || !facet_local || !nb || !nj || nc < 0 || nv < 2) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    parallel_rows(nc, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t c = lo; c < hi; ++c)
            for (int i = 0; i < nv; ++i) {
                const int64_t f = cell_facets[c * nv + i];
                const int other = facet_cells[2 * f] != c ? 0 : 1;          // the side of the facet that is NOT this cell
                nb[c * nv + i] = facet_cells[2 * f + other];
                nj[c * nv + i] = facet_local[2 * f + other];
            }
    });
    return 0;
}
// Box tests on the midpoints ((x_0 + x_1) + ... ) / nv of the rows of conn (mesh builders: make_mesh_3D.py:15-50 of the reference marks the
// cells whose midpoint lies in [a, b] and the facets on the box surface).  mode 0: out = 1 where a <= m <= b on every axis; mode 1: out = 1 where
// the midpoint lies on the surface of the box -- within [a - eps, b + eps] on the other axes and closer than eps to a or b on one axis.
int knp_host_box_marks(int64_t n, int d, const double* coords, const int32_t* conn, int nv, const double* a, const double* b, double eps, int mode,
                       uint8_t* out, int nthreads) {
    if (!coords || !conn || !a || !b || !out || n < 0 || d < 1 || d > 3 || nv < 1 || (mode != 0 && mode != 1)) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            double m[3];
            for (int k = 0; k < d; ++k) {
                double v = coords[(int64_t)conn[i * nv] * d + k];
                for (int q = 1; q < nv; ++q) v += coords[(int64_t)conn[i * nv + q] * d + k];
                m[k] = v / (double)nv;
            }
            bool r;
            if (mode == 0) {
                r = true;
                for (int k = 0; k < d; ++k) r = r && m[k] >= a[k] && m[k] <= b[k];
            } else {
                r = false;
                for (int ax = 0; ax < d; ++ax) {
                    bool within = true;
                    for (int o = 0; o < d; ++o)
                        if (o != ax) within = within && m[o] >= a[o] - eps && m[o] <= b[o] + eps;
                    r = r || (within && (std::fabs(m[ax] - a[ax]) < eps || std::fabs(m[ax] - b[ax]) < eps));
                }
            }
            out[i] = r ? 1 : 0;
        }
    });
    return 0;
}
// ---- smoothed-aggregation passes of knpemidg/amg.py: build_hierarchy, row-parallel, same arithmetic as the numpy / scipy lines they replace ----

// Strength graph: S keeps the off-diagonal entries of row i with |a_ij| >= theta sqrt(|a_ii a_jj|).  Sp[n + 1] by the caller, *Sj allocated here.
int knp_host_strength(int64_t n, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* diag, double theta, int32_t* Sp, int32_t** Sj,
                      int nthreads) {
    if (!Ap || !Aj || !Ax || !diag || !Sp || !Sj || n < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    auto strong = [&](int64_t i, int32_t p) {
        const int32_t j = Aj[p];
        return j != i && std::fabs(Ax[p]) >= theta * std::sqrt(std::fabs(diag[i] * diag[j]));
    };
    std::vector<int32_t> cnt((size_t)n, 0);
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            int32_t c = 0;
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p) c += strong(i, p) ? 1 : 0;
            cnt[(size_t)i] = c;
        }
    });
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) { Sp[i] = (int32_t)nnz; nnz += cnt[(size_t)i]; if (nnz > 2147483647LL) return -3; }
    Sp[n] = (int32_t)nnz;
    *Sj = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    if (!*Sj) return -2;
    int32_t* sj = *Sj;
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            int32_t o = Sp[i];
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p)
                if (strong(i, p)) sj[o++] = Aj[p];
        }
    });
    return 0;
}

// One damped-Jacobi step on a prolongator:  C = P - diag(v) A P  (v = omega / a_ii), entries that come out exactly 0 dropped (scipy's sparse
// subtraction does), columns sorted.  The products are formed as (v_i a_ik) p_kj and summed in the order of knp_host_spgemm, then subtracted from
// p_ij -- the numbers of  (P - spgemm(scale_rows(A, v), P)).  A is [n x n], P is [n x m].
int knp_host_smooth_prolongator(int64_t n, int64_t m, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* v, const int32_t* Pp,
                                const int32_t* Pj, const double* Px, int32_t* Cp, int32_t** Cj, double** Cx, int nthreads) {
    if (!Ap || !Aj || !Ax || !v || !Pp || !Pj || !Px || !Cp || !Cj || !Cx || n < 0 || m < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    // rows are formed once (their values decide which entries vanish) into ONE buffer per thread, row after row; a prefix sum over the row
    // lengths then places the threads' buffers in the result
    if (nthreads < 1) nthreads = 1;
    const int nt = n < 4096 ? 1 : nthreads;
    const int64_t chunk = (n + nt - 1) / nt;
    std::vector<std::vector<int32_t>> bj((size_t)nt);
    std::vector<std::vector<double>> bx((size_t)nt);
    std::vector<int32_t> len((size_t)n, 0);
    auto work = [&](int t) {
        const int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
        std::vector<int32_t> pos((size_t)m, -1);
        std::vector<std::pair<int32_t, double>> row;
        std::vector<uint8_t> inP;
        auto& oj = bj[(size_t)t];
        auto& ox = bx[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) {
            row.clear();
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int32_t k = Aj[p];
                const double a = Ax[p] * v[i];
                for (int32_t q = Pp[k]; q < Pp[k + 1]; ++q) {
                    const int32_t j = Pj[q];
                    if (pos[j] < 0) { pos[j] = (int32_t)row.size(); row.emplace_back(j, a * Px[q]); }
                    else row[pos[j]].second += a * Px[q];
                }
            }
            const size_t nprod = row.size();
            inP.assign(nprod, 0);
            for (int32_t q = Pp[i]; q < Pp[i + 1]; ++q) {                 // P - (product): entries of P without a product partner stay as they are
                const int32_t j = Pj[q];
                if (pos[j] < 0) { pos[j] = (int32_t)row.size(); row.emplace_back(j, Px[q]); }
                else { row[pos[j]].second = Px[q] - row[pos[j]].second; inP[(size_t)pos[j]] = 1; }
            }
            for (size_t e = 0; e < nprod; ++e)
                if (!inP[e]) row[e].second = -row[e].second;               // 0 - product
            for (auto& e : row) pos[e.first] = -1;
            std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& a_, const std::pair<int32_t, double>& b_) { return a_.first < b_.first; });
            int32_t kept = 0;
            for (auto& e : row)
                if (e.second != 0.0) { oj.push_back(e.first); ox.push_back(e.second); ++kept; }
            len[(size_t)i] = kept;
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back([&, t]() { work(t); });
        for (auto& th : pool) th.join();
    }
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) { Cp[i] = (int32_t)nnz; nnz += len[(size_t)i]; if (nnz > 2147483647LL) return -3; }
    Cp[n] = (int32_t)nnz;
    *Cj = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    *Cx = (double*)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
    if (!*Cj || !*Cx) return -2;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * chunk;
        if (lo >= n) break;
        std::memcpy(*Cj + Cp[lo], bj[(size_t)t].data(), sizeof(int32_t) * bj[(size_t)t].size());
        std::memcpy(*Cx + Cp[lo], bx[(size_t)t].data(), sizeof(double) * bx[(size_t)t].size());
    }
    return 0;
}

// Prolongator truncation: per row keep the entries with |p_ij| >= trunc * max_j |p_ij| and rescale the row so that it still interpolates the
// coarse vector Bc to the same fine value ((P Bc)_i, sums in storage order as scipy's csr_matvec).  Tp[n + 1] by the caller, *Tj / *Tx allocated here.
int knp_host_truncate_prolongator(int64_t n, const int32_t* Pp, const int32_t* Pj, const double* Px, double trunc, const double* Bc, int32_t* Tp,
                                  int32_t** Tj, double** Tx, int nthreads) {
    if (!Pp || !Pj || !Px || !Bc || !Tp || !Tj || !Tx || n < 0) return -1;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<double> rowmax((size_t)n, 0.0);
    std::vector<int32_t> cnt((size_t)n, 0);
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            double mx = 0.0;
            for (int32_t p = Pp[i]; p < Pp[i + 1]; ++p) mx = std::max(mx, std::fabs(Px[p]));
            rowmax[(size_t)i] = mx;
            int32_t c = 0;
            for (int32_t p = Pp[i]; p < Pp[i + 1]; ++p) c += std::fabs(Px[p]) >= trunc * mx ? 1 : 0;
            cnt[(size_t)i] = c;
        }
    });
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) { Tp[i] = (int32_t)nnz; nnz += cnt[(size_t)i]; }
    Tp[n] = (int32_t)nnz;
    *Tj = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    *Tx = (double*)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
    if (!*Tj || !*Tx) return -2;
    int32_t* tj = *Tj;
    double* tx = *Tx;
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            double tgt = 0.0, got = 0.0;
            int64_t o = Tp[i];
            const double thr = trunc * rowmax[(size_t)i];
            for (int32_t p = Pp[i]; p < Pp[i + 1]; ++p) {
                tgt += Px[p] * Bc[Pj[p]];
                if (std::fabs(Px[p]) >= thr) { tj[o] = Pj[p]; tx[o] = Px[p]; got += Px[p] * Bc[Pj[p]]; ++o; }
            }
            const double scale = std::fabs(got) > 1e-300 ? tgt / (got == 0.0 ? 1.0 : got) : 1.0;
            for (int64_t q = Tp[i]; q < o; ++q) tx[q] = scale * tx[q];
        }
    });
    return 0;
}
}  // extern "C"
