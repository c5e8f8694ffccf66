"""CPU experiment behind DESIGN.md section 5 ("why the two-level preconditioner is what it is"): PCG iteration counts of the EMI system
of the r=0 single-axon mesh (62 k DoFs, oracle matrix, Morton cell order, EXACT coarse solves) for different DG-level smoothers and
auxiliary spaces.  Result (rtol 1e-8):  block-Jacobi + conforming 64 | two-step Chebyshev block-Jacobi + conforming 40 (shipped) |
symmetric block Gauss-Seidel + conforming 37 | the same, multiplicative 29 | smoothed DG<-conforming transfers 38-39 |
Chebyshev + conforming + an additional piecewise-constant space 27.   usage: python tools/smoother_experiment.py"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import knpemi_oracle as ko
from knpemidg import amg, _abi
from knpemidg.mesh import make_mesh_3D
from common import synthetic_state
m, s, f = make_mesh_3D(0, n_axons=1)
pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
synthetic_state(pb)
nd = pb.nd
# Morton order of cells (as on the device)
xc = m.coords[m.cells]; scale = np.median(xc.max(axis=1)-xc.min(axis=1), axis=0)
order = _abi.morton_order(m.cell_midpoints(), scale)
perm = (order[:, None]*nd + np.arange(nd)[None, :]).ravel()
cs = amg.ConformingSpace(m, f.array(), (1,))
P = sp.csr_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), cs.dof.ravel())), shape=(pb.ndof, cs.n))[perm]
def prep(A):
    A = A.tocsr()[perm][:, perm].tocsr()
    Ab = A.tobsr(blocksize=(nd, nd)); Ab.sort_indices()
    nb = A.shape[0]//nd
    rowid = np.repeat(np.arange(nb), np.diff(Ab.indptr))
    dsel = Ab.indices == rowid
    Dblk = np.zeros((nb, nd, nd)); Dblk[rowid[dsel]] = Ab.data[dsel]
    Dinv = np.linalg.inv(Dblk)
    lower = sp.bsr_matrix((Ab.data[Ab.indices <= rowid], Ab.indices[Ab.indices <= rowid], np.concatenate([[0], np.cumsum(np.bincount(rowid[Ab.indices <= rowid], minlength=nb))])), shape=A.shape).tocsr()
    upper = sp.bsr_matrix((Ab.data[Ab.indices >= rowid], Ab.indices[Ab.indices >= rowid], np.concatenate([[0], np.cumsum(np.bincount(rowid[Ab.indices >= rowid], minlength=nb))])), shape=A.shape).tocsr()
    Dm = sp.bsr_matrix((Dblk, np.arange(nb), np.arange(nb+1)), shape=A.shape).tocsr()
    return A, Dinv, lower.tocsc(), upper.tocsc(), Dm
def bj(Dinv, r): return np.einsum("bij,bj->bi", Dinv, r.reshape(-1, nd)).ravel()
def variants(A, singular):
    A, Dinv, Lw, Up, Dm = prep(A)
    Ac = (P.T @ A @ P).tocsc()
    if singular:
        Ac = Ac + 1e-8*sp.identity(Ac.shape[0])*abs(Ac.diagonal()).mean()
    lu = spla.splu(Ac)
    coarse = lambda r: P @ lu.solve(P.T @ r)
    # lambda max of Binv A
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    for _ in range(30):
        y = bj(Dinv, A @ x); lam = np.linalg.norm(y)/np.linalg.norm(x); x = y/np.linalg.norm(y)
    lmax = 1.1*lam; lmin = 0.05*lmax
    theta, delta = 0.5*(lmax+lmin), 0.5*(lmax-lmin); sigma = theta/delta; rho0 = 1/sigma; rho1 = 1/(2*sigma-rho0)
    def cheb2(r):
        y0 = bj(Dinv, r); t = A @ y0
        return (1+rho1*rho0)/theta*y0 + 2*rho1/delta*bj(Dinv, r - t/theta)
    Ll = spla.splu(Lw, permc_spec="NATURAL", diag_pivot_thresh=0); Uu = spla.splu(Up, permc_spec="NATURAL", diag_pivot_thresh=0)
    fgs = lambda r: Ll.solve(r)
    sgs = lambda r: Uu.solve(Dm @ Ll.solve(r))
    return A, {"BJ+C": lambda r: bj(Dinv, r) + coarse(r), "Cheb2+C": lambda r: cheb2(r) + coarse(r), "fGS+C": lambda r: fgs(r) + coarse(r),
               "sGS+C": lambda r: sgs(r) + coarse(r),
               "sGS*C (mult)": lambda r: (lambda z: z + sgs(r - A @ z))(coarse(r))}
def count(A, M, b, solver, tol):
    it = [0]
    Mop = spla.LinearOperator(A.shape, matvec=M)
    if solver == "cg":
        x, info = spla.cg(A, b, rtol=tol, atol=0, maxiter=500, M=Mop, callback=lambda xk: it.__setitem__(0, it[0]+1))
    else:
        x, info = spla.bicgstab(A, b, rtol=tol, atol=0, maxiter=500, M=Mop, callback=lambda xk: it.__setitem__(0, it[0]+1))
    return it[0], info
Ae, be, _ = ko.assemble_emi(pb, want_B=False)
be = (be - be.mean())[perm]
A, V = variants(Ae, True)
for k, M in V.items():
    if "fGS" in k: continue
    print("EMI", k, count(A, M, be, "cg", 1e-8))
Ak = ko.assemble_knp(pb, 0); bk = ko.knp_rhs(pb, 0)[perm]
A, V = variants(Ak, False)
x0 = None
for k, M in V.items():
    print("KNP", k, count(A, M, bk, "bicgstab", 1e-9))

print("---- smoothed transfer experiment (EMI) ----")
A, Dinv, Lw, Up, Dm = prep(Ae)
x = np.random.default_rng(0).standard_normal(A.shape[0])
for _ in range(30):
    y = bj(Dinv, A @ x); lam = np.linalg.norm(y)/np.linalg.norm(x); x = y/np.linalg.norm(y)
lmax = 1.1*lam
Binv = sp.bsr_matrix((Dinv, np.arange(A.shape[0]//nd), np.arange(A.shape[0]//nd+1)), shape=A.shape).tocsr()
for omega_f in (0.5, 1.0, 4.0/3.0):
    om = omega_f/lmax
    Ps = (P - om*(Binv @ (A @ P))).tocsr()
    Acs = (Ps.T @ A @ Ps).tocsc()
    Acs = Acs + 1e-8*sp.identity(Acs.shape[0])*abs(Acs.diagonal()).mean()
    lu = spla.splu(Acs)
    coarse = lambda r: Ps @ lu.solve(Ps.T @ r)
    lmin = 0.05*lmax
    theta, delta = 0.5*(lmax+lmin), 0.5*(lmax-lmin); sigma = theta/delta; rho0 = 1/sigma; rho1 = 1/(2*sigma-rho0)
    def cheb2(r):
        y0 = bj(Dinv, r); t = A @ y0
        return (1+rho1*rho0)/theta*y0 + 2*rho1/delta*bj(Dinv, r - t/theta)
    print("omega", omega_f, "nnz(Ac_s)/nnz(Ac)", Acs.nnz/(P.T@A@P).nnz,
          "BJ+Cs", count(A, lambda r: bj(Dinv, r)+coarse(r), be, "cg", 1e-8), "Cheb2+Cs", count(A, lambda r: cheb2(r)+coarse(r), be, "cg", 1e-8))

print("---- additional P0 auxiliary space (EMI) ----")
nb = A.shape[0]//nd
R0 = sp.csr_matrix((np.ones(A.shape[0]), (np.arange(A.shape[0]), np.repeat(np.arange(nb), nd))), shape=(A.shape[0], nb))
A0 = (R0.T @ A @ R0).tocsc(); A0 = A0 + 1e-8*sp.identity(nb)*abs(A0.diagonal()).mean()
lu0 = spla.splu(A0)
Ac = (P.T @ A @ P).tocsc(); Ac = Ac + 1e-8*sp.identity(Ac.shape[0])*abs(Ac.diagonal()).mean(); luc = spla.splu(Ac)
lmin = 0.05*lmax
theta, delta = 0.5*(lmax+lmin), 0.5*(lmax-lmin); sigma = theta/delta; rho0 = 1/sigma; rho1 = 1/(2*sigma-rho0)
def cheb2(r):
    y0 = bj(Dinv, r); t = A @ y0
    return (1+rho1*rho0)/theta*y0 + 2*rho1/delta*bj(Dinv, r - t/theta)
print("Cheb2 + C + P0", count(A, lambda r: cheb2(r) + P @ luc.solve(P.T @ r) + R0 @ lu0.solve(R0.T @ r), be, "cg", 1e-8))
print("BJ + C + P0", count(A, lambda r: bj(Dinv, r) + P @ luc.solve(P.T @ r) + R0 @ lu0.solve(R0.T @ r), be, "cg", 1e-8))
print("Cheb2 + P0 only", count(A, lambda r: cheb2(r) + R0 @ lu0.solve(R0.T @ r), be, "cg", 1e-8))
