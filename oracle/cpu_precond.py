"""CPU application of the auxiliary-space preconditioner for bench.py's `cpu_baseline` leg (TEST / BENCH INFRASTRUCTURE: never
imported by the product).  Same operator as the device applies (csrc/krylov.hip, csrc/amg.hip):

    M^-1 r = B^-1 r + P V(P^T r),   B = cell-block diagonal of the DG matrix, P = injection of the conforming space,
    V = one V-cycle of the smoothed-aggregation hierarchy with Chebyshev-Jacobi smoothing and a dense coarse pseudo-inverse,

built from the hierarchy LEVELS the product's setup produced (passed in by the caller), so that CPU and GPU iteration counts are
comparable.  The reference preconditions with hypre BoomerAMG (src/knpemidg/solver.py:433, 688); this stands in for it on the
CPU exactly as it does on the GPU."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _cheb(A, dinv, rho, degree, lower, b, x=None):
    """`degree` steps of the Chebyshev iteration for D^-1 A on [lower rho, rho] (Saad, Alg. 12.1)."""
    if degree <= 0:
        return np.zeros_like(b) if x is None else x
    lmax, lmin = rho, lower * rho
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rk = 1.0 / sigma
    r = b if x is None else b - A @ x
    d = dinv * r / theta
    x = d.copy() if x is None else x + d
    for _ in range(1, degree):
        rn = 1.0 / (2.0 * sigma - rk)
        r = r - A @ d
        d = rn * rk * d + (2.0 * rn / delta) * (dinv * r)
        x = x + d
        rk = rn
    return x


def vcycle(levels, b):
    nl = len(levels)
    bs, xs = [b], []
    for l in range(nl - 1):
        L = levels[l]
        x = _cheb(L.A, L.dinv, L.rho, L.cheb_degree, L.cheb_lower, bs[l])
        xs.append(x)
        r = bs[l] - L.A @ x if L.cheb_degree > 0 else bs[l]
        bs.append(L.R @ r)
    x = levels[-1].pinv @ bs[-1]
    for l in range(nl - 2, -1, -1):
        L = levels[l]
        x = xs[l] + L.P @ x
        if L.cheb_degree > 0:
            x = _cheb(L.A, L.dinv, L.rho, L.cheb_degree, L.cheb_lower, bs[l], x)
    return x


def aux_space_preconditioner(A, nd, dg2cg, levels):
    """LinearOperator r -> B^-1 r + P V(P^T r) for the DG matrix A (csr) with nd dofs per cell."""
    n = A.shape[0]
    nb = n // nd
    Ab = A.tobsr(blocksize=(nd, nd))
    Ab.sort_indices()
    rowid = np.repeat(np.arange(nb), np.diff(Ab.indptr))
    sel = Ab.indices == rowid
    diag = np.zeros((nb, nd, nd))
    diag[rowid[sel]] = Ab.data[sel]
    inv = np.linalg.inv(diag)
    P = sp.csr_matrix((np.ones(n), (np.arange(n), np.asarray(dg2cg).ravel())), shape=(n, levels[0].A.shape[0]))
    PT = P.T.tocsr()

    def mv(r):
        r = np.asarray(r).ravel()
        return np.einsum("bij,bj->bi", inv, r.reshape(nb, nd)).ravel() + P @ vcycle(levels, PT @ r)
    return spla.LinearOperator((n, n), matvec=mv, dtype=np.float64)
