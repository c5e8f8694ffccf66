"""CPU experiment: what the solves on the EMIx tissue reconstruction (121 617 unstructured tets) are waiting for.  PCG / BiCGStab iteration
counts of the oracle's EMI / KNP matrices with EXACT auxiliary-space solves, for different DG-level smoothers and auxiliary spaces:
  BJ          cell-block-Jacobi                      ChebK      K-step Chebyshev on the block-Jacobi-preconditioned operator
  +C          conforming (membrane-broken) P1 space  +P0        an additional piecewise-constant space
  sGS         symmetric block Gauss-Seidel over the cells (Morton order)
usage: python tools/emix_smoother_experiment.py [emix|r1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "examples", "emix_simulations")]
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import knpemi_oracle as ko
from knpemidg import amg, _abi
from common import synthetic_state

which = sys.argv[1] if len(sys.argv) > 1 else "emix"
if which == "emix":
    from emix_common import load_mesh
    m, s, f = load_mesh()
    pb = ko.build_emix(m, s.array(), f.array())
else:
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(1)
    pb = ko.build_idealized(m, s.array(), f.array())
synthetic_state(pb)
nd = pb.nd
xc = m.coords[m.cells]
scale = np.median(xc.max(axis=1) - xc.min(axis=1), axis=0)
order = _abi.morton_order(m.cell_midpoints(), scale)
perm = (order[:, None] * nd + np.arange(nd)[None, :]).ravel()
cs = amg.ConformingSpace(m, f.array(), (1, 2))
P = sp.csr_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), cs.dof.ravel())), shape=(pb.ndof, cs.n))[perm]
nb = pb.ndof // nd
R0 = sp.csr_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), np.repeat(np.arange(nb), nd))), shape=(pb.ndof, nb))


def prep(A):
    A = A.tocsr()[perm][:, perm].tocsr()
    Ab = A.tobsr(blocksize=(nd, nd)); Ab.sort_indices()
    rowid = np.repeat(np.arange(nb), np.diff(Ab.indptr))
    dsel = Ab.indices == rowid
    Dblk = np.zeros((nb, nd, nd)); Dblk[rowid[dsel]] = Ab.data[dsel]
    return A, np.linalg.inv(Dblk), Ab, rowid, Dblk


def tri(Ab, rowid, keep):
    return sp.bsr_matrix((Ab.data[keep], Ab.indices[keep], np.concatenate([[0], np.cumsum(np.bincount(rowid[keep], minlength=nb))])),
                         shape=Ab.shape).tocsc()


def bj(Dinv, r):
    return np.einsum("bij,bj->bi", Dinv, r.reshape(-1, nd)).ravel()


def exact(Ac):
    Ac = Ac.tocsc()
    Ac = Ac + 1e-8 * sp.identity(Ac.shape[0]) * abs(Ac.diagonal()).mean()
    return spla.splu(Ac)


def count(A, M, b, solver, tol):
    it = [0]
    Mop = spla.LinearOperator(A.shape, matvec=M)
    fn = spla.cg if solver == "cg" else spla.bicgstab
    x, info = fn(A, b, rtol=tol, atol=0, maxiter=400, M=Mop, callback=lambda xk: it.__setitem__(0, it[0] + 1))
    return it[0] if info == 0 else "%d (no conv.)" % it[0]


def run(name, Araw, b, solver, tol):
    A, Dinv, Ab, rowid, Dblk = prep(Araw)
    luC = exact(P.T @ A @ P)
    lu0 = exact(R0.T @ A @ R0)
    C = lambda r: P @ luC.solve(P.T @ r)
    C0 = lambda r: R0 @ lu0.solve(R0.T @ r)
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    for _ in range(40):
        y = bj(Dinv, A @ x); lam = np.linalg.norm(y) / np.linalg.norm(x); x = y / np.linalg.norm(y)
    lmax = 1.1 * lam

    def cheb(k, lower=0.05):
        lmin = lower * lmax
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta

        def apply(r):                                            # Saad Alg. 12.1 on Binv A, zero initial guess
            rho = 1.0 / sigma
            d = bj(Dinv, r) / theta
            x = d.copy()
            for _ in range(k - 1):
                res = r - A @ x
                rho_n = 1.0 / (2.0 * sigma - rho)
                d = rho_n * rho * d + 2.0 * rho_n / delta * bj(Dinv, res)
                x = x + d
                rho = rho_n
            return x
        return apply
    Lw = spla.splu(tri(Ab, rowid, Ab.indices <= rowid), permc_spec="NATURAL", diag_pivot_thresh=0)
    Up = spla.splu(tri(Ab, rowid, Ab.indices >= rowid), permc_spec="NATURAL", diag_pivot_thresh=0)
    Dm = sp.bsr_matrix((Dblk, np.arange(nb), np.arange(nb + 1)), shape=A.shape).tocsr()
    sgs = lambda r: Up.solve(Dm @ Lw.solve(r))
    b = b[perm]
    print("%s: %d DoFs, lambda_max(Binv A) = %.2f" % (name, A.shape[0], lmax / 1.1))
    V = {"BJ + C": lambda r: bj(Dinv, r) + C(r)}
    for k in (2, 3, 4, 6):
        V["Cheb%d + C" % k] = (lambda ch: (lambda r: ch(r) + C(r)))(cheb(k))
    V["Cheb2(lower 0.2) + C"] = (lambda ch: (lambda r: ch(r) + C(r)))(cheb(2, 0.2))
    V["Cheb2 + C + P0"] = (lambda ch: (lambda r: ch(r) + C(r) + C0(r)))(cheb(2))
    V["BJ + C + P0"] = lambda r: bj(Dinv, r) + C(r) + C0(r)
    V["sGS + C"] = lambda r: sgs(r) + C(r)
    V["sGS * C (multiplicative)"] = lambda r: (lambda z: z + sgs(r - A @ z))(C(r))
    for k, M in V.items():
        t0 = time.perf_counter()
        print("   %-28s %s" % (k, count(A, M, b, solver, tol)), flush=True)


Ae, be, _ = ko.assemble_emi(pb, want_B=False)
run("EMI", Ae, be - be.mean(), "cg", 1e-8)
run("KNP species 0", ko.assemble_knp(pb, 0), ko.knp_rhs(pb, 0), "bicgstab", 1e-9)
