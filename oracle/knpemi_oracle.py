"""TEST INFRASTRUCTURE (oracle) -- a CPU restatement of the reference's DG
assemble-and-solve path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path never does.

PARITY UNPINNED: the reference (adajel/KNP-EMI-DG) cannot be imported here
(dolfin / petsc4py are absent, plain ModuleNotFoundError) and its tests hold no
golden vectors for this path (tests print errors and rates without asserting,
/root/reference/tests/run_MMS_space.py:297-329).  This restatement is therefore
pinned only indirectly: by manufactured solutions restated from
/root/reference/tests/mms_space.py (oracle/mms.py), by exactness checks of the
quadrature and by physical invariants.

What is restated (all citations relative to /root/reference):

* geometry, oriented interface normal, plus/minus traces
      src/knpemidg/solver.py:85-121, src/knpemidg/utils.py:61-98
* EMI forms a_emi, L_emi, B_emi              src/knpemidg/solver.py:270-403
* KNP forms A_knp, L_knp                     src/knpemidg/solver.py:534-663
* facet-average projector                    src/knpemidg/utils.py:100-124
* step-III updates                           src/knpemidg/solver.py:808-845
* solver semantics (CG / GMRES, nullspace)   src/knpemidg/solver.py:406-531, 665-791

Unlike the HIP path (matrix-free, cell-based gather, closed-form P1 facet
integrals) this oracle is deliberately FEniCS-shaped: it loops over cells and
interior facets, builds macro-element blocks with numerical quadrature for any
polynomial degree, and assembles global CSR matrices.

DoF layout (build-defined): scalar dof(c, j) = c*nd + j; P1 node
j = cell vertex j (cells hold ascending vertex ids); P2 adds one node per edge
(a<b) in lexicographic order after the vertices.  Facet fields hold one value
per facet of `Mesh.facets`.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from quadrature import simplex_rule


# ----------------------------------------------------------------------------
# reference-element basis in barycentric coordinates
# ----------------------------------------------------------------------------
def edges_of(nv):
    return [(a, b) for a in range(nv) for b in range(a + 1, nv)]


def ndofs(p, d):
    return d + 1 if p == 1 else (d + 1) * (d + 2) // 2


def tabulate(p, bary):
    """Values B[q, j] and barycentric derivatives dB[q, j, l] = d phi_j / d lambda_l."""
    bary = np.asarray(bary)
    nq, nv = bary.shape
    if p == 1:
        B = bary.copy()
        dB = np.broadcast_to(np.eye(nv), (nq, nv, nv)).copy()
        return B, dB
    if p == 2:
        ed = edges_of(nv)
        nd = nv + len(ed)
        B = np.zeros((nq, nd))
        dB = np.zeros((nq, nd, nv))
        for a in range(nv):
            B[:, a] = bary[:, a] * (2 * bary[:, a] - 1)
            dB[:, a, a] = 4 * bary[:, a] - 1
        for e, (a, b) in enumerate(ed):
            B[:, nv + e] = 4 * bary[:, a] * bary[:, b]
            dB[:, nv + e, a] = 4 * bary[:, b]
            dB[:, nv + e, b] = 4 * bary[:, a]
        return B, dB
    raise ValueError("degree 1 or 2")


def node_bary(p, d):
    """Barycentric coordinates of the Lagrange nodes."""
    nv = d + 1
    pts = [np.eye(nv)[a] for a in range(nv)]
    if p == 2:
        for a, b in edges_of(nv):
            v = np.zeros(nv)
            v[a] = v[b] = 0.5
            pts.append(v)
    return np.array(pts)


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
class Geometry:
    """Affine geometry of every cell and facet.

    vol[c], glam[c, l, :] = grad lambda_l, h[c] = CellDiameter,
    farea[f], fnormal[f, :] = unit normal pointing out of facet_cells[f, 0].
    """

    def __init__(self, mesh):
        self.mesh = mesh
        d = mesh.gdim
        x = mesh.coords[mesh.cells]                      # [Nc, d+1, d]
        J = (x[:, 1:, :] - x[:, :1, :]).transpose(0, 2, 1)   # columns = edge vectors
        detJ = np.linalg.det(J)
        fact = {2: 2.0, 3: 6.0}[d]
        self.vol = np.abs(detJ) / fact
        Jinv = np.linalg.inv(J)                          # rows = grad of lambda_1..d
        glam = np.empty((x.shape[0], d + 1, d))
        glam[:, 1:, :] = Jinv
        glam[:, 0, :] = -Jinv.sum(axis=1)
        self.glam = glam
        h = np.zeros(x.shape[0])
        for a in range(d + 1):
            for b in range(a + 1, d + 1):
                h = np.maximum(h, np.linalg.norm(x[:, a] - x[:, b], axis=1))
        self.h = h
        # facets
        fx = mesh.coords[mesh.facets]                    # [Nf, d, d]
        if d == 2:
            t = fx[:, 1] - fx[:, 0]
            self.farea = np.linalg.norm(t, axis=1)
            n = np.stack([t[:, 1], -t[:, 0]], axis=1)
        else:
            n = np.cross(fx[:, 1] - fx[:, 0], fx[:, 2] - fx[:, 0])
            self.farea = 0.5 * np.linalg.norm(n, axis=1)
        n = n / np.linalg.norm(n, axis=1)[:, None]
        # orient: outward from side-0 cell == along -grad lambda_i of that cell
        c0 = mesh.facet_cells[:, 0]
        l0 = mesh.facet_local[:, 0].astype(np.int64)
        g = glam[c0, l0]
        sgn = -np.sign(np.einsum("fd,fd->f", n, g))
        self.fnormal = n * sgn[:, None]


def facet_orientation(mesh, cell_tags):
    """Which side of each interior facet is the `plus` (normal-leaving, lower tag)
    side of the oriented interface normal n_g.

    reference: src/knpemidg/utils.py:61-98.  n_g points from the lower cell tag to
    the higher one; on equal tags the reference picks n('-') (utils.py:80), i.e.
    plus = the '-' restriction.  This build calls facet side 0 (lower cell index)
    '+', so on equal tags plus = side 1.
    Returns e_side[f] in {0,1} (side index of the plus / ECS-like cell); -1 on the boundary.
    """
    fc = mesh.facet_cells
    e = np.full(fc.shape[0], -1, dtype=np.int8)
    it = fc[:, 1] >= 0
    t0 = cell_tags[fc[it, 0]].astype(np.int64)
    t1 = cell_tags[fc[it, 1]].astype(np.int64)
    # chi('+') >= chi('-')  -> n_g = n('-') -> normal leaves side 1 -> plus = side 1
    e[it] = np.where(t0 >= t1, 1, 0)
    return e


# ----------------------------------------------------------------------------
# discretisation context
# ----------------------------------------------------------------------------
class Space:
    """DG-p tabulations shared by all forms."""

    def __init__(self, mesh, geom, p):
        self.mesh, self.geom, self.p = mesh, geom, p
        d = mesh.gdim
        self.d = d
        self.nd = ndofs(p, d)
        self._cell_tabs = {}
        self._facet_tabs = {}

    def cell_tab(self, degree):
        if degree not in self._cell_tabs:
            bary, w = simplex_rule(self.d, degree)
            B, dB = tabulate(self.p, bary)
            self._cell_tabs[degree] = (bary, w, B, dB)
        return self._cell_tabs[degree]

    def facet_tab(self, degree):
        """Tabulation on each local facet: because cells and facets both hold ascending
        vertex ids, facet vertex j maps to the cell's local vertices with `i` removed, in order."""
        if degree not in self._facet_tabs:
            mu, w = simplex_rule(self.d - 1, degree)
            Bs, dBs, barys = [], [], []
            for i in range(self.d + 1):
                bary = np.insert(mu, i, 0.0, axis=1)
                B, dB = tabulate(self.p, bary)
                Bs.append(B)
                dBs.append(dB)
                barys.append(bary)
            self._facet_tabs[degree] = (mu, w, np.array(Bs), np.array(dBs), np.array(barys))
        return self._facet_tabs[degree]

    # physical gradients of basis at cell quadrature points: [Nc, q, j, d]
    def cell_grads(self, dB):
        return np.einsum("qjl,cld->cqjd", dB, self.geom.glam)

    def interp_nodes(self):
        """Physical coordinates of the Lagrange nodes [Nc, nd, d]."""
        nb = node_bary(self.p, self.d)
        x = self.mesh.coords[self.mesh.cells]
        return np.einsum("jl,cld->cjd", nb, x)


def _coo(nrows, rows, cols, vals):
    return sp.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())), shape=(nrows, nrows)).tocsr()


class FacetSide:
    """Basis values / normal derivatives of one side of a set of facets at facet quad points."""

    def __init__(self, space, fids, side, degree):
        mesh, geom = space.mesh, space.geom
        mu, w, Bs, dBs, barys = space.facet_tab(degree)
        self.cells = mesh.facet_cells[fids, side].astype(np.int64)
        li = mesh.facet_local[fids, side].astype(np.int64)
        self.B = Bs[li]                                       # [F, q, j]
        gl = geom.glam[self.cells]                            # [F, l, d]
        self.grad = np.einsum("fqjl,fld->fqjd", dBs[li], gl)  # [F, q, j, d]
        self.bary = barys[li]                                 # [F, q, l]
        self.w = w
        self.mu = mu

    def val(self, nodal):
        """Evaluate a DG-p nodal field [Nc, nd] at the facet quadrature points -> [F, q]."""
        return np.einsum("fqj,fj->fq", self.B, nodal[self.cells])

    def gradval(self, nodal):
        return np.einsum("fqjd,fj->fqd", self.grad, nodal[self.cells])


# ----------------------------------------------------------------------------
# the problem: parameters + state, array-backed
# ----------------------------------------------------------------------------
class Problem:
    """Everything `Solver` holds after setup_domain/parameters/FEM_spaces
    (reference: src/knpemidg/solver.py:85-225), on arrays.

    ions: list of dicts with 'z', 'D' f64[Nc], 'name'; the LAST ion is eliminated
    (solver.py:69, 189-191).  State: c [N_ions, Nc, nd] (solved species), c_elim [Nc, nd],
    phi [Nc, nd], phi_M [Nf], I_ch[name] [Nf].
    """

    def __init__(self, mesh, cell_tags, facet_tags, p, ions, params, membrane_tags, rho=None):
        self.mesh = mesh
        self.cell_tags = np.asarray(cell_tags).astype(np.int64)
        self.facet_tags = np.asarray(facet_tags).astype(np.int64)
        self.p = p
        self.geom = Geometry(mesh)
        self.space = Space(mesh, self.geom, p)
        self.nd = self.space.nd
        self.d = mesh.gdim
        self.ions = ions
        self.N_ions = len(ions) - 1
        self.F = float(params["F"])
        self.R = float(params["R"])
        self.T = float(params["temperature"])
        self.C_M = float(params["C_M"])
        self.dt = float(params["dt"])
        self.C_phi = float(params["C_phi"])
        self.psi = self.F / (self.R * self.T)
        self.tau = 20.0 * self.d * p                   # solver.py:109-111
        self.membrane_tags = list(membrane_tags)
        nc = mesh.num_cells()
        self.rho = np.zeros(nc) if rho is None else np.asarray(rho, float)
        self.e_side = facet_orientation(mesh, self.cell_tags)
        it = mesh.facet_cells[:, 1] >= 0
        self.int0 = np.nonzero(it & (self.facet_tags == 0))[0]          # dS(0)
        self.mem = np.nonzero(it & np.isin(self.facet_tags, self.membrane_tags))[0]
        self.ext = np.nonzero(~it)[0]
        self.ndof = nc * self.nd
        nf = mesh.num_facets()
        self.c = np.zeros((self.N_ions, nc, self.nd))
        self.c_prev_n = np.zeros_like(self.c)
        self.c_elim = np.zeros((nc, self.nd))
        self.phi = np.zeros((nc, self.nd))
        self.phi_M = np.zeros(nf)
        self.I_ch = {ion["name"]: np.zeros(nf) for ion in ions}
        self.splitting = True
        self.f_source = [0.0] * self.N_ions              # constants on dx(0) (solver.py:599)
        self.mms = None
        self.Lp = float((mesh.coords.max(axis=0) - mesh.coords.min(axis=0)).max())  # solver.py:383-391

    # -- coefficient fields ---------------------------------------------------
    def all_c(self):
        """c_k of all N ions incl. eliminated (solver.py:289-296): [N, Nc, nd]."""
        return np.concatenate([self.c, self.c_elim[None]], axis=0)

    def kappa(self):
        """kappa = sum_k F z_k^2 D_k psi c_k  (solver.py:306), DG-p nodal."""
        cc = self.all_c()
        k = np.zeros_like(self.c_elim)
        for ion, ck in zip(self.ions, cc):
            k += self.F * ion["z"] ** 2 * self.psi * ion["D"][:, None] * ck
        return k

    def alpha_sum(self):
        """alpha_sum = sum_k D_k z_k^2 c_k (solver.py:303)."""
        cc = self.all_c()
        s = np.zeros_like(self.c_elim)
        for ion, ck in zip(self.ions, cc):
            s += ion["z"] ** 2 * ion["D"][:, None] * ck
        return s


# ----------------------------------------------------------------------------
# EMI assembly  (solver.py:270-403)
# ----------------------------------------------------------------------------
def _sipg_blocks(pb, fids, coefP, coefM, pen_of, deg):
    """Macro-element blocks of
        - avg(k grad u).n+ jump(v) - avg(k grad v).n+ jump(u) + pen * jump(u) jump(v)
    on facets `fids` with side-wise coefficient values coefP/coefM [F,q] and penalty
    weights pen_of(P, M) -> [F,q].  Returns rows, cols, vals of the 2nd x 2nd blocks."""
    sp_, geom, nd = pb.space, pb.geom, pb.nd
    P = FacetSide(sp_, fids, 0, deg)
    M = FacetSide(sp_, fids, 1, deg)
    n = geom.fnormal[fids]
    area = geom.farea[fids]
    wq = P.w[None, :] * area[:, None]                            # [F,q]
    kP = coefP(P)
    kM = coefM(M)
    dnP = np.einsum("fqjd,fd->fqj", P.grad, n)
    dnM = np.einsum("fqjd,fd->fqj", M.grad, n)
    J = np.concatenate([P.B, -M.B], axis=2)                      # jump operator [F,q,2nd]
    G = 0.5 * np.concatenate([kP[:, :, None] * dnP, kM[:, :, None] * dnM], axis=2)
    pen = pen_of(P, M, kP, kM)
    blk = -np.einsum("fq,fqv,fqu->fvu", wq, J, G) - np.einsum("fq,fqv,fqu->fvu", wq, G, J)
    dofs = np.concatenate([P.cells[:, None] * nd + np.arange(nd)[None, :],
                           M.cells[:, None] * nd + np.arange(nd)[None, :]], axis=1)
    rows = np.repeat(dofs[:, :, None], 2 * nd, axis=2)
    cols = np.repeat(dofs[:, None, :], 2 * nd, axis=1)
    return rows, cols, blk, (P, M, wq, J, pen)


def assemble_emi(pb, want_B=True):
    """Returns (A, b, B): a_emi, L_emi and the preconditioner form B_emi."""
    mesh, geom, S = pb.mesh, pb.geom, pb.space
    nd, p, d = pb.nd, pb.p, pb.d
    ndof = pb.ndof
    kap = pb.kappa()
    rows, cols, vals = [], [], []

    # --- cells: int kappa grad u . grad v
    deg = max(1, 3 * p - 2)
    bary, w, B, dB = S.cell_tab(deg)
    G = S.cell_grads(dB)                                         # [c,q,j,d]
    kq = np.einsum("qj,cj->cq", B, kap)
    wq = w[None, :] * geom.vol[:, None]
    blk = np.einsum("cq,cq,cqud,cqvd->cvu", wq, kq, G, G)
    dofs = np.arange(mesh.num_cells())[:, None] * nd + np.arange(nd)[None, :]
    rows.append(np.repeat(dofs[:, :, None], nd, axis=2))
    cols.append(np.repeat(dofs[:, None, :], nd, axis=1))
    vals.append(blk)

    # --- dS(0): SIPG terms
    fdeg = 3 * p
    hbar = None
    if len(pb.int0):
        f0 = pb.int0
        hbar = 0.5 * (geom.h[mesh.facet_cells[f0, 0]] + geom.h[mesh.facet_cells[f0, 1]])

        def pen_of(P, M, kP, kM):
            return (pb.tau / hbar)[:, None] * 0.5 * (kP + kM)
        r, c, blk, (P, M, wqf, J, pen) = _sipg_blocks(
            pb, f0, lambda s: s.val(kap), lambda s: s.val(kap), pen_of, fdeg)
        blk = blk + np.einsum("fq,fq,fqv,fqu->fvu", wqf, pen, J, J)
        rows.append(r); cols.append(c); vals.append(blk)

    # --- membrane: C_phi jump(u) jump(v) on dS(tag)
    if len(pb.mem):
        fm = pb.mem
        P = FacetSide(S, fm, 0, 2 * p)
        M = FacetSide(S, fm, 1, 2 * p)
        wqf = P.w[None, :] * geom.farea[fm][:, None]
        J = np.concatenate([P.B, -M.B], axis=2)
        blk = pb.C_phi * np.einsum("fq,fqv,fqu->fvu", wqf, J, J)
        dofs = np.concatenate([P.cells[:, None] * nd + np.arange(nd)[None, :],
                               M.cells[:, None] * nd + np.arange(nd)[None, :]], axis=1)
        rows.append(np.repeat(dofs[:, :, None], 2 * nd, axis=2))
        cols.append(np.repeat(dofs[:, None, :], 2 * nd, axis=1))
        vals.append(blk)

    A = _coo(ndof, np.concatenate([r.ravel() for r in rows]),
             np.concatenate([c.ravel() for c in cols]),
             np.concatenate([v.ravel() for v in vals]))

    b = emi_rhs(pb)

    Bm = None
    if want_B:
        # B = a + kappa/Lp^2 * int u v   (solver.py:376-395)
        deg = 3 * p
        bary, w, Bt, dB = S.cell_tab(deg)
        kq = np.einsum("qj,cj->cq", Bt, kap)
        wq = w[None, :] * geom.vol[:, None]
        blk = np.einsum("cq,cq,qu,qv->cvu", wq, kq, Bt, Bt) / pb.Lp ** 2
        dofs = np.arange(mesh.num_cells())[:, None] * nd + np.arange(nd)[None, :]
        Mk = _coo(ndof, np.repeat(dofs[:, :, None], nd, axis=2),
                  np.repeat(dofs[:, None, :], nd, axis=1), blk)
        Bm = A + Mk
    return A, b, Bm


def emi_rhs(pb):
    """L_emi (solver.py:309-310, 330-344; MMS extras 349-374 via pb.mms)."""
    mesh, geom, S = pb.mesh, pb.geom, pb.space
    nd, p = pb.nd, pb.p
    b = np.zeros((mesh.num_cells(), nd))
    cc = pb.all_c()

    # - F z_k int D_k grad c_k . grad v dx
    deg = max(1, 2 * p - 2)
    bary, w, B, dB = S.cell_tab(deg)
    G = S.cell_grads(dB)
    wq = w[None, :] * geom.vol[:, None]
    for ion, ck in zip(pb.ions, cc):
        gc = np.einsum("cqjd,cj->cqd", G, ck)
        b += -pb.F * ion["z"] * ion["D"][:, None] * np.einsum("cq,cqd,cqvd->cv", wq, gc, G)

    # + F z_k int_dS(0) avg(D_k grad c_k).n+ jump(v)
    if len(pb.int0):
        f0 = pb.int0
        fdeg = max(1, 2 * p - 1)
        P = FacetSide(S, f0, 0, fdeg)
        M = FacetSide(S, f0, 1, fdeg)
        n = geom.fnormal[f0]
        wqf = P.w[None, :] * geom.farea[f0][:, None]
        flux = np.zeros((len(f0), len(P.w)))
        for ion, ck in zip(pb.ions, cc):
            gP = np.einsum("fqd,fd->fq", P.gradval(ck), n) * ion["D"][P.cells][:, None]
            gM = np.einsum("fqd,fd->fq", M.gradval(ck), n) * ion["D"][M.cells][:, None]
            flux += pb.F * ion["z"] * 0.5 * (gP + gM)
        np.add.at(b, P.cells, np.einsum("fq,fq,fqv->fv", wqf, flux, P.B))
        np.add.at(b, M.cells, -np.einsum("fq,fq,fqv->fv", wqf, flux, M.B))

    # + C_phi int_dS(tag) avg(g) JUMP(v, n_g),  JUMP = minus - plus = v_i - v_e
    if len(pb.mem) and pb.mms is None:
        fm = pb.mem
        g = pb.phi_M[fm].copy()
        if not pb.splitting:                                       # solver.py:337
            I_tot = sum(pb.I_ch[ion["name"]][fm] for ion in pb.ions)
            g = g - I_tot / pb.C_phi
        _add_membrane_linear(pb, b, fm, pb.C_phi * g, 2 * p - 1 if p > 1 else 1)
    if pb.mms is not None:
        pb.mms.add_emi_rhs(pb, b)
    return b.ravel()


def _add_membrane_linear(pb, b, fids, coef, deg):
    """b += int_F coef[f] * (v_i - v_e) with e = plus (lower-tag) side."""
    S, geom, mesh = pb.space, pb.geom, pb.mesh
    es = pb.e_side[fids].astype(np.int64)
    for side in (0, 1):
        Sd = FacetSide(S, fids, side, deg)
        wqf = Sd.w[None, :] * geom.farea[fids][:, None]
        sign = np.where(es == side, -1.0, 1.0)                   # e side gets -v_e
        np.add.at(b, Sd.cells, np.einsum("f,fq,fqv->fv", sign * coef, wqf, Sd.B))


# ----------------------------------------------------------------------------
# KNP assembly  (solver.py:534-663)
# ----------------------------------------------------------------------------
def knp_facet_degree(p):
    # UFL's estimate for the merged dS(0) integrand: jump(v) jump(un u) with
    # un ~ grad(phi) -> p + (p-1) + p ; P1: exact, P2: kink of |.| sampled at degree 5
    return max(2, 3 * p - 1)


def assemble_knp(pb, idx):
    """A_knp block of solved species `idx` (the mixed operator is block diagonal over
    species, solver.py:550-594) and nothing else."""
    mesh, geom, S = pb.mesh, pb.geom, pb.space
    nd, p = pb.nd, pb.p
    ion = pb.ions[idx]
    z, D = ion["z"], ion["D"]
    rows, cols, vals = [], [], []
    nc = mesh.num_cells()

    # cells: 1/dt u v + D grad u.grad v + z psi D u grad(phi).grad v
    deg = max(2, 3 * p - 2, 2 * p)
    bary, w, B, dB = S.cell_tab(deg)
    G = S.cell_grads(dB)
    wq = w[None, :] * geom.vol[:, None]
    gphi = np.einsum("cqjd,cj->cqd", G, pb.phi)
    blk = np.einsum("cq,qu,qv->cvu", wq, B, B) / pb.dt
    blk += np.einsum("c,cq,cqud,cqvd->cvu", D, wq, G, G)
    blk += z * pb.psi * np.einsum("c,cq,qu,cqd,cqvd->cvu", D, wq, B, gphi, G)
    dofs = np.arange(nc)[:, None] * nd + np.arange(nd)[None, :]
    rows.append(np.repeat(dofs[:, :, None], nd, axis=2))
    cols.append(np.repeat(dofs[:, None, :], nd, axis=1))
    vals.append(blk)

    if len(pb.int0):
        f0 = pb.int0
        fdeg = knp_facet_degree(p)
        hbar = 0.5 * (geom.h[mesh.facet_cells[f0, 0]] + geom.h[mesh.facet_cells[f0, 1]])
        r, c, blk, (P, M, wqf, J, _) = _sipg_blocks(
            pb, f0, lambda s: np.broadcast_to(D[s.cells][:, None], (len(f0), len(s.w))),
            lambda s: np.broadcast_to(D[s.cells][:, None], (len(f0), len(s.w))),
            lambda P, M, kP, kM: None, fdeg)
        n = geom.fnormal[f0]
        # tau/avg(h) jump(D u) jump(v)
        JD = np.concatenate([D[P.cells][:, None, None] * P.B, -D[M.cells][:, None, None] * M.B], axis=2)
        blk = blk + np.einsum("f,fq,fqv,fqu->fvu", pb.tau / hbar, wqf, J, JD)
        # - z psi jump(v) jump(un u), un = 0.5 (D grad(phi).n + |D grad(phi).n|), n = own outward normal
        sP = np.einsum("fqd,fd->fq", P.gradval(pb.phi), n) * D[P.cells][:, None]
        sM = -np.einsum("fqd,fd->fq", M.gradval(pb.phi), n) * D[M.cells][:, None]
        unP = 0.5 * (sP + np.abs(sP))
        unM = 0.5 * (sM + np.abs(sM))
        JU = np.concatenate([unP[:, :, None] * P.B, -unM[:, :, None] * M.B], axis=2)
        blk = blk - z * pb.psi * np.einsum("fq,fqv,fqu->fvu", wqf, J, JU)
        rows.append(r); cols.append(c); vals.append(blk)

    A = _coo(pb.ndof, np.concatenate([r.ravel() for r in rows]),
             np.concatenate([c.ravel() for c in cols]),
             np.concatenate([v.ravel() for v in vals]))
    return A


def membrane_rhs_degree(p):
    # C*g*v with alpha rational in c: UFL sums degrees of numerator and denominator:
    # alpha ~ 2p, C g ~ 4p, times v -> 5p  
    return 5 * p


def knp_rhs(pb, idx):
    """L_knp for solved species idx (solver.py:597-629; MMS 632-657 via pb.mms)."""
    mesh, geom, S = pb.mesh, pb.geom, pb.space
    nd, p = pb.nd, pb.p
    ion = pb.ions[idx]
    z, D = ion["z"], ion["D"]
    b = np.zeros((mesh.num_cells(), nd))

    deg = 2 * p
    bary, w, B, dB = S.cell_tab(deg)
    wq = w[None, :] * geom.vol[:, None]
    cq = np.einsum("qj,cj->cq", B, pb.c_prev_n[idx])
    b += np.einsum("cq,cq,qv->cv", wq, cq, B) / pb.dt
    fs = pb.f_source[idx]                                         # ion['f_source'] * v_c * dx(0), solver.py:599
    ecs = (pb.cell_tags == 0)
    if callable(fs):
        # any UFL coefficient in the reference (e.g. the Expression(degree=4) of run_tortuosity.py:180-200): evaluated at the
        # points of a degree-8 rule (t = pb.t for time-dependent sources)
        bary8, w8, B8, _ = S.cell_tab(8)
        X = np.einsum("ql,cld->cqd", bary8, mesh.coords[mesh.cells[ecs]])
        fq = np.broadcast_to(np.asarray(fs(X, getattr(pb, "t", 0.0)), dtype=float), X.shape[:-1])
        b[ecs] += np.einsum("q,c,cq,qv->cv", w8, geom.vol[ecs], fq, B8)
    elif isinstance(fs, np.ndarray):
        b[ecs] += fs[ecs, None] * np.einsum("cq,qv->cv", wq[ecs], B)
    elif fs != 0.0:
        b[ecs] += fs * np.einsum("cq,qv->cv", wq[ecs], B)

    if len(pb.mem) and pb.mms is None:
        fm = pb.mem
        deg = membrane_rhs_degree(p)
        es = pb.e_side[fm].astype(np.int64)
        sides = [FacetSide(S, fm, 0, deg), FacetSide(S, fm, 1, deg)]
        wqf = sides[0].w[None, :] * geom.farea[fm][:, None]
        asum = pb.alpha_sum()
        ck = pb.c[idx]                                           # c_prev_k == current c
        I_k = pb.I_ch[ion["name"]][fm][:, None]
        I_tot = sum(pb.I_ch[i["name"]][fm] for i in pb.ions)[:, None]
        phiM = pb.phi_M[fm][:, None]
        Cs, gs, phis = [], [], []
        for sd in sides:
            alpha = D[sd.cells][:, None] * z * z * sd.val(ck) / sd.val(asum)
            C = alpha * pb.C_M / (pb.F * z * pb.dt)
            g = phiM - pb.dt / (pb.C_M * alpha) * I_k
            if pb.splitting:
                g = g + (pb.dt / pb.C_M) * I_tot
            Cs.append(C); gs.append(g); phis.append(sd.val(pb.phi))
        # JUMP(C g v) - jump(phi) jump(C) avg(v) - jump(phi) avg(C) jump(v)
        #   = sum over sides s of  sgn_s * C_s * (g_s - (phi_i - phi_e)) * v_s,  sgn = +1 on i, -1 on e
        for side, sd in enumerate(sides):
            is_e = (es == side)
            phi_i = np.where(is_e[:, None], phis[1 - side], phis[side])
            phi_e = np.where(is_e[:, None], phis[side], phis[1 - side])
            sgn = np.where(is_e, -1.0, 1.0)[:, None]
            integrand = sgn * Cs[side] * (gs[side] - (phi_i - phi_e))
            np.add.at(b, sd.cells, np.einsum("fq,fq,fqv->fv", wqf, integrand, sd.B))
    if pb.mms is not None:
        pb.mms.add_knp_rhs(pb, idx, b)
    return b.ravel()


# ----------------------------------------------------------------------------
# facet projector and step-III updates (utils.py:100-124, solver.py:808-845)
# ----------------------------------------------------------------------------
def facet_average(pb, fids, fun, deg):
    """(1/|F|) int_F fun(plus_vals, minus_vals) -- `fun` receives callables that evaluate a
    nodal field on the plus (e) / minus (i) side at the facet quadrature points."""
    S = pb.space
    es = pb.e_side[fids].astype(np.int64)
    sides = [FacetSide(S, fids, 0, deg), FacetSide(S, fids, 1, deg)]

    def plus(nodal):
        v0, v1 = sides[0].val(nodal), sides[1].val(nodal)
        return np.where((es == 0)[:, None], v0, v1)

    def minus(nodal):
        v0, v1 = sides[0].val(nodal), sides[1].val(nodal)
        return np.where((es == 0)[:, None], v1, v0)
    vals = fun(plus, minus)
    return np.einsum("q,fq->f", sides[0].w, vals)


def nernst_degree(p):
    # ln(f): deg(f)+2, f = c_e/c_i -> 2p  
    return 2 * p + 2


def update_phi_M(pb, fids=None):
    """phi_M = facet-avg(JUMP(phi)) = avg(phi_i - phi_e) (solver.py:813-814)."""
    fids = pb.mem if fids is None else fids
    out = facet_average(pb, fids, lambda plus, minus: minus(pb.phi) - plus(pb.phi), max(1, pb.p))
    pb.phi_M[fids] = out
    return out


def nernst(pb, k, fids=None):
    """E_k = RT/(F z_k) facet-avg ln(c_e / c_i) on membrane facets (solver.py:827-828, 841-842)."""
    fids = pb.mem if fids is None else fids
    ck = pb.all_c()[k]
    z = pb.ions[k]["z"]
    val = facet_average(pb, fids, lambda plus, minus: np.log(plus(ck) / minus(ck)), nernst_degree(pb.p))
    return pb.R * pb.T / (pb.F * z) * val


def update_c_elim(pb):
    """c_N = -(sum_k z_k c_k + rho)/z_N (solver.py:831-838); all terms live in DG-p/DG0 so the
    L2 projection is the nodal combination."""
    zN = pb.ions[-1]["z"]
    acc = np.zeros_like(pb.c_elim)
    for ion, ck in zip(pb.ions[:-1], pb.c):
        acc += -(1.0 / zN) * ion["z"] * ck
    acc += -(1.0 / zN) * pb.rho[:, None]
    pb.c_elim = acc
    return acc


# ----------------------------------------------------------------------------
# solves (solver.py:406-531, 665-791)
# ----------------------------------------------------------------------------
def solve_emi(pb, direct=True, rtol=1e-5, x0=None, stats=None):
    A, b, Bm = assemble_emi(pb, want_B=not direct)
    n = A.shape[0]
    if direct:
        # singular (null space = constants, and 1^T A = 0): with the constant removed from b (Z_.remove(b),
        # solver.py:489-490) the system is consistent, so dropping the last equation/unknown (x[n-1] = 0) is exact;
        # the solution is then shifted to zero mean.  (A bordered system [A 1; 1^T 0] gives the same x but its dense
        # border fills the sparse LU: 113 s instead of 6 s on the 15 552-tet mesh.)
        b = b - b.mean()
        Ar = A.tocsc()[:n - 1, :n - 1]
        lu = spla.splu(Ar, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        x = np.concatenate([lu.solve(b[:n - 1]), [0.0]])
        x -= x.mean()
    else:
        # CG with a Jacobi-type stand-in for BoomerAMG(B): block-diagonal of B
        Minv = block_jacobi(Bm, pb.nd)
        it = [0]
        x, info = spla.cg(A, b, x0=x0, rtol=rtol, atol=0.0, maxiter=20000, M=Minv,
                          callback=lambda xk: it.__setitem__(0, it[0] + 1))
        if stats is not None:
            stats["emi_iters"] = it[0]
        assert info == 0, "EMI CG did not converge"
    pb.phi = x.reshape(-1, pb.nd)
    return pb.phi


def block_jacobi(A, nd):
    """LinearOperator applying the inverse of the nd x nd cell-diagonal blocks of A."""
    n = A.shape[0]
    nb = n // nd
    Ab = A.tobsr(blocksize=(nd, nd))
    Ab.sort_indices()
    diag = np.zeros((nb, nd, nd))
    indptr, indices, data = Ab.indptr, Ab.indices, Ab.data
    rowid = np.repeat(np.arange(nb), np.diff(indptr))
    sel = indices == rowid
    diag[rowid[sel]] = data[sel]
    inv = np.linalg.inv(diag)

    def mv(x):
        return np.einsum("bij,bj->bi", inv, x.reshape(nb, nd)).ravel()
    return spla.LinearOperator((n, n), matvec=mv)


def solve_knp(pb, direct=True, rtol=1e-7, stats=None):
    out = np.zeros_like(pb.c)
    its = []
    for idx in range(pb.N_ions):
        A = assemble_knp(pb, idx)
        b = knp_rhs(pb, idx)
        if direct:
            x = spla.splu(A.tocsc(), permc_spec="MMD_AT_PLUS_A").solve(b)      # structurally symmetric: AMD on A + A^T
        else:
            Minv = block_jacobi(A, pb.nd)
            it = [0]
            x, info = spla.gmres(A, b, x0=pb.c[idx].ravel(), rtol=rtol, atol=0.0, restart=30,
                                 maxiter=2000, M=Minv, callback=lambda r: it.__setitem__(0, it[0] + 1),
                                 callback_type="pr_norm")
            its.append(it[0])
            assert info == 0, "KNP GMRES did not converge"
        out[idx] = x.reshape(-1, pb.nd)
    if stats is not None:
        stats["knp_iters"] = its
    pb.c = out
    return out


def solve_for_time_step(pb, direct=True, rtol_emi=1e-5, rtol_knp=1e-7, stats=None):
    """One PDE step I -> II -> III (solver.py:794-847).  Returns dict of Nernst potentials."""
    x0 = pb.phi.ravel().copy()
    solve_emi(pb, direct=direct, rtol=rtol_emi, x0=x0, stats=stats)
    solve_knp(pb, direct=direct, rtol=rtol_knp, stats=stats)
    pb.c_prev_n = pb.c.copy()
    update_phi_M(pb)
    update_c_elim(pb)
    E = {ion["name"]: nernst(pb, k) for k, ion in enumerate(pb.ions)}
    return E


# ----------------------------------------------------------------------------
# the reference's idealized-geometry configurations, on arrays
# ----------------------------------------------------------------------------
def idealized_params():
    """Physical parameters and initial values of examples/idealized-geometries/run_3D.py:60-90
    (identical in run_2D.py:60-90)."""
    dt = 1.0e-4
    C_M = 0.02
    return dict(dt=dt, C_M=C_M, temperature=300.0, F=96485.0, R=8.314, C_phi=C_M / dt,
                D=dict(Na=1.33e-9, K=1.96e-9, Cl=2.03e-9),
                z=dict(Na=1.0, K=1.0, Cl=-1.0),
                init=dict(Na=(12.838513108648856, 100.71925900027354),        # (ICS, ECS)
                          K=(124.15397583491901, 3.3236967382705265)),
                phi_M_init=-0.07438609374462003)


def build_idealized(mesh, subdomains, surfaces, p=1, membrane_tags=(1, 2)):
    """Problem of run_3D.py / run_2D.py: ion_list = [K, Cl, Na] (Na eliminated, run_3D.py:142),
    tag-wise constant initial concentrations, phi_M = phi_M_init on membrane facets."""
    P = idealized_params()
    tags = np.asarray(subdomains).astype(np.int64)
    nc = mesh.num_cells()
    init = dict(P["init"])
    init["Cl"] = (init["Na"][0] + init["K"][0], init["Na"][1] + init["K"][1])   # run_3D.py:84-85
    ions = [dict(name=n, z=P["z"][n], D=np.full(nc, P["D"][n])) for n in ("K", "Cl", "Na")]
    pb = Problem(mesh, tags, np.asarray(surfaces), p, ions, P, membrane_tags=membrane_tags)
    ics = (tags >= 1)[:, None]
    ones = np.ones((nc, pb.nd))
    for i, n in enumerate(("K", "Cl")):
        pb.c[i] = np.where(ics, init[n][0], init[n][1]) * ones
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = np.where(ics, init["Na"][0], init["Na"][1]) * ones
    pb.phi_M[pb.mem] = P["phi_M_init"]
    return pb


def tortuosity_params():
    """Physical parameters and initial values of examples/local-astrocyte-depolarization/run_tortuosity.py:78-165 (cm / ms / mV /
    mM), variant M2 (lambda_i = 12.8, lambda_e = 6.4): the one shipped configuration with a NON-ZERO background charge rho_sub
    (:116-121) and diffusion coefficients that differ per subdomain (D / lambda^2, :154-156).  Tuples by subdomain
    0 ECS / 1 neuronal / 2 glial."""
    dt = 0.1
    C_M = 1.0
    init = dict(K=(3.092970607490389, 124.13988964240784, 99.3100014897692),
                Na=(144.60625137617149, 12.850454639128186, 15.775818906083778),
                Cl=(133.62525154406637, 5.0, 5.203660274163705))
    lam = (1.6 * 4, 3.2 * 4, 3.2 * 4)
    return dict(dt=dt, C_M=C_M, temperature=307e3, F=96500e3, R=8.315e3, C_phi=C_M / dt,
                D=dict(Na=1.33e-8, K=1.96e-8, Cl=2.03e-8), lam=lam,
                z=dict(Na=1.0, K=1.0, Cl=-1.0), init=init,
                rho=tuple(-(init["Na"][s] + init["K"][s] - init["Cl"][s]) for s in range(3)),
                phi_M_init={1: -83.08511451850003, 2: -74.3848784437955})


def build_tortuosity(mesh, subdomains, surfaces, p=1, membrane_tags=(1, 2)):
    """Problem of run_tortuosity.py:78-230 on any mesh with subdomain tags 0 / 1 / 2 (coordinates in cm): ion_list = [K, Na, Cl]
    with Cl (z = -1) ELIMINATED (:229), D_k / lambda_sub^2 per subdomain, rho_sub != 0 entering the eliminated concentration
    (solver.py:831-838)."""
    P = tortuosity_params()
    tags = np.asarray(subdomains).astype(np.int64)
    nc = mesh.num_cells()
    lam2 = np.asarray(P["lam"])[tags] ** 2
    ions = [dict(name=n, z=P["z"][n], D=P["D"][n] / lam2) for n in ("K", "Na", "Cl")]
    pb = Problem(mesh, tags, np.asarray(surfaces), p, ions, P, membrane_tags=membrane_tags, rho=np.asarray(P["rho"])[tags])
    ones = np.ones((nc, pb.nd))
    for i, n in enumerate(("K", "Na")):
        pb.c[i] = np.asarray(P["init"][n])[tags][:, None] * ones
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = np.asarray(P["init"]["Cl"])[tags][:, None] * ones
    for tag, v in P["phi_M_init"].items():
        if tag in membrane_tags:
            pb.phi_M[pb.mem[pb.facet_tags[pb.mem] == tag]] = v
    return pb


def emix_params():
    """Physical parameters and initial values of examples/emix-simulations/run_EMIx_simulation.py:56-91 (cm / ms / mV / mM:
    temperature in mK, F in mC/mol, R in mJ/(K mol), D in cm^2/ms).  init = (ECS, glial, neuronal) by subdomain 0 / 1 / 2."""
    dt = 0.1
    C_M = 2.0
    return dict(dt=dt, C_M=C_M, temperature=300e3, F=96485e3, R=8.314e3, C_phi=C_M / dt,
                D=dict(Na=1.33e-8, K=1.96e-8, Cl=2.03e-8),
                z=dict(Na=1.0, K=1.0, Cl=-1.0),
                init=dict(K=(3.3236967382613933, 102.75563828644862, 124.15397583492471),
                          Na=(100.71925900028181, 12.39731187972181, 12.838513108606818)),
                # initial membrane potentials = the ODE models' initial V per membrane tag (mm_glial.py:11, mm_hh.py:14)
                phi_M_init={1: -83.08511451850003, 2: -74.3848784437955})


def build_emix(mesh, subdomains, surfaces, p=1, membrane_tags=(1, 2)):
    """Problem of run_EMIx_simulation.py:56-170: ion_list = [K, Cl, Na] (Na eliminated, :147), concentrations constant per
    subdomain (0 ECS, 1 glial, 2 neuronal; :118-124), Cl = Na + K in every subdomain (:86-89), phi_M = the membrane models'
    initial potentials on their facets (glial tag 1, neuronal tag 2; :249)."""
    P = emix_params()
    tags = np.asarray(subdomains).astype(np.int64)
    nc = mesh.num_cells()
    init = dict(P["init"])
    init["Cl"] = tuple(a + b for a, b in zip(init["Na"], init["K"]))
    ions = [dict(name=n, z=P["z"][n], D=np.full(nc, P["D"][n])) for n in ("K", "Cl", "Na")]
    pb = Problem(mesh, tags, np.asarray(surfaces), p, ions, P, membrane_tags=membrane_tags)
    ones = np.ones((nc, pb.nd))
    for i, n in enumerate(("K", "Cl")):
        pb.c[i] = np.asarray(init[n])[tags][:, None] * ones
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = np.asarray(init["Na"])[tags][:, None] * ones
    for tag, v in P["phi_M_init"].items():
        if tag in membrane_tags:
            pb.phi_M[pb.mem[pb.facet_tags[pb.mem] == tag]] = v
    return pb
