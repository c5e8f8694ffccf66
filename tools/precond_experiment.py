"""CPU experiment (round 3): variants of the two-level preconditioner that cost the SAME per application as the shipped additive one
(one operator apply inside the two-step Chebyshev block-Jacobi smoother + one coarse solve), on the oracle's KNP and EMI matrices.
usage: python tools/precond_experiment.py [n_axons] [resolution]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import knpemi_oracle as ko
from knpemidg import amg, _abi
from knpemidg.mesh import make_mesh_3D
from common import synthetic_state
n_ax = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m, s, f = make_mesh_3D(res, n_axons=n_ax)
mt = (1,) if n_ax == 1 else (1, 2)
pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=mt)
synthetic_state(pb)
nd = pb.nd
cs = amg.ConformingSpace(m, f.array(), mt)
P = sp.csr_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), cs.dof.ravel())), shape=(pb.ndof, cs.n))

def bj_of(A):
    Ab = A.tobsr(blocksize=(nd, nd)); Ab.sort_indices()
    nb = A.shape[0] // nd
    rowid = np.repeat(np.arange(nb), np.diff(Ab.indptr))
    dsel = Ab.indices == rowid
    Dblk = np.zeros((nb, nd, nd)); Dblk[rowid[dsel]] = Ab.data[dsel]
    Dinv = np.linalg.inv(Dblk)
    return lambda r: np.einsum("bij,bj->bi", Dinv, r.reshape(-1, nd)).ravel()

def run(name, A, b, solver, tol, singular, A_for_B=None):
    A = A.tocsr()
    bj = bj_of(A if A_for_B is None else A_for_B.tocsr())
    Ac = (P.T @ A @ P).tocsc()
    if singular:
        Ac = Ac + 1e-8 * sp.identity(Ac.shape[0]) * abs(Ac.diagonal()).mean()
    lu = spla.splu(Ac)
    coarse = lambda r: P @ lu.solve(P.T @ r)
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    for _ in range(30):
        y = bj(A @ x); lam = np.linalg.norm(y) / np.linalg.norm(x); x = y / np.linalg.norm(y)
    lmax = 1.1 * lam
    napply = [0]
    def Aop(v):
        napply[0] += 1
        return A @ v
    def mk(lfrac):
        lmin = lfrac * lmax
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin); sigma = theta / delta; rho0 = 1 / sigma; rho1 = 1 / (2 * sigma - rho0)
        ca, cb = (1 + rho1 * rho0) / theta, 2 * rho1 / delta
        def additive(r):
            y0 = bj(r); t = Aop(y0)
            return ca * y0 + cb * bj(r - t / theta) + coarse(r)
        def hybrid(r):                       # coarse correction of the residual left by the FIRST Chebyshev step (same apply)
            y0 = bj(r); t = Aop(y0)
            r1 = r - t / theta
            return ca * y0 + cb * bj(r1) + coarse(r1)
        def hybrid2(r):                      # coarse first-step residual, weighted consistently: x = y0/theta + coarse(r1) + (second cheb step on r1)
            y0 = bj(r); t = Aop(y0)
            r1 = r - t / theta
            return y0 / theta + coarse(r1) + (rho1 * rho0) * y0 / theta + cb * bj(r1)
        def mult_pre(r):                     # one damped BJ step, then coarse on the true residual (1 apply)
            y0 = bj(r); t = Aop(y0)
            om = 1.0 / theta
            return om * y0 + coarse(r - om * t)
        def cheb3(r, where):                 # three Chebyshev steps (2 applies), coarse correction of the residual after step `where`
            rho2 = 1 / (2 * sigma - rho1)
            y0 = bj(r); x = y0 / theta; d = x.copy()
            rr = r - Aop(d)
            xc = coarse(rr) if where == 1 else 0.0
            d = rho1 * rho0 * d + 2 * rho1 / delta * bj(rr); x = x + d
            rr = rr - Aop(d)
            if where == 2:
                xc = coarse(rr)
            d = rho2 * rho1 * d + 2 * rho2 / delta * bj(rr); x = x + d
            return x + xc
        return {"additive": additive, "hybrid": hybrid, "cheb3_c1": lambda r: cheb3(r, 1), "cheb3_c2": lambda r: cheb3(r, 2)}
    out = {}
    for lfrac in (0.03, 0.07, 0.1, 0.15):
        for k, M in mk(lfrac).items():
            it = [0]; napply[0] = 0
            Mop = spla.LinearOperator(A.shape, matvec=M)
            if solver == "cg":
                xs, info = spla.cg(A, b, rtol=tol, atol=0, maxiter=500, M=Mop, callback=lambda xk: it.__setitem__(0, it[0] + 1))
            elif solver == "gmres":
                xs, info = spla.gmres(A, b, rtol=tol, atol=0, restart=30, maxiter=20, M=Mop, callback=lambda rk: it.__setitem__(0, it[0] + 1), callback_type="pr_norm")
            else:
                xs, info = spla.bicgstab(A, b, rtol=tol, atol=0, maxiter=500, M=Mop, callback=lambda xk: it.__setitem__(0, it[0] + 1))
            print("%s lmin=%.2f %-10s its %3d info %d  precond applications %d" % (name, lfrac, k, it[0], info, napply[0]), flush=True)

# the r=0 mesh with dt scaled by 16 has the diffusion number D dt / h^2 of the benchmarked r=2 mesh (h is 4x smaller there)
pb.dt *= float(os.environ.get("DT_SCALE", 16.0))
pb.C_phi = pb.C_M / pb.dt
ko.solve_emi(pb, direct=True)
Ak = ko.assemble_knp(pb, 0); bk = ko.knp_rhs(pb, 0)
sc = 1.0 / abs(Ak.diagonal()).mean()
run("KNP", Ak * sc, bk * sc / np.linalg.norm(bk * sc), "bicgstab", 1e-6, False)
phi_keep = pb.phi.copy(); pb.phi = np.zeros_like(pb.phi)
A0 = ko.assemble_knp(pb, 0); pb.phi = phi_keep
print("relative size of the drift part of A:", abs(Ak - A0).max() / abs(Ak).max())
run("KNP-driftfreeB", Ak * sc, bk * sc / np.linalg.norm(bk * sc), "bicgstab", 1e-6, False, A_for_B=A0 * sc)
