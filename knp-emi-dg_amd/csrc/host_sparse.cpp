// Threaded host kernels of the preconditioner SETUP (knpemidg/amg.py): the sparse matrix products of the smoothed-aggregation
// hierarchy (prolongator smoothing A P, Galerkin products R (A P)) were > 80 % of the non-LAPACK setup time in single-threaded
// scipy.  Row-parallel Gustavson SpGEMM in two passes (symbolic row counts, prefix sum, numeric fill with sorted columns).
// The reference builds BoomerAMG inside PETSc at every solve (src/knpemidg/solver.py:433, 505, 688, 767); this is the
// corresponding setup work of this build's auxiliary-space hierarchy.  Plain C ABI, no device code.
#include "../../include/knpemi_hip.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
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
    std::vector<int64_t> cnt((size_t)n, 0);
    // pass 1: row sizes
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> mark((size_t)m, -1);
        for (int64_t i = lo; i < hi; ++i) {
            int64_t c = 0;
            for (int32_t p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int32_t k = Aj[p];
                for (int32_t q = Bp[k]; q < Bp[k + 1]; ++q)
                    if (mark[Bj[q]] != (int32_t)i) { mark[Bj[q]] = (int32_t)i; ++c; }
            }
            cnt[i] = c;
        }
    });
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (nnz > 2147483647LL) return -3;
        Cp[i] = (int32_t)nnz;
        nnz += cnt[i];
    }
    if (nnz > 2147483647LL) return -3;
    Cp[n] = (int32_t)nnz;
    *Cj = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    *Cx = (double*)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
    if (!*Cj || !*Cx) return -2;
    int32_t* cj = *Cj;
    double* cx = *Cx;
    // pass 2: numeric, sorted columns
    parallel_rows(n, nthreads, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> pos((size_t)m, -1);
        std::vector<std::pair<int32_t, double>> row;
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
            int64_t o = Cp[i];
            for (auto& e : row) { cj[o] = e.first; cx[o] = e.second; ++o; }
        }
    });
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

}  // extern "C"
